# usage: ab_sf.sh KEY v1 v2 ...  : SegFormer forward ms with the tuning switch at each value (same box)
KEY=$1; shift
for v in "$@"; do
  echo -n "$KEY=$v "; env $KEY=$v python3 scripts/bench_segformer.py || exit 1
done
