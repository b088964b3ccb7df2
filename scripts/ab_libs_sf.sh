# usage: ab_libs_sf.sh tag1 tag2 ... : SegFormer forward under each variant library (flair-1_amd/flair_amd/libflair_hip_<tag>.so; "base" = the shipped one)
for v in "$@"; do
  if [ "$v" == "base" ]; then L=$(pwd)/flair-1_amd/flair_amd/libflair_hip.so; else L=$(pwd)/flair-1_amd/flair_amd/libflair_hip_$v.so; fi
  echo -n "$v "; FLAIR_HIP_LIB=$L python3 scripts/bench_segformer.py || exit 1
done
