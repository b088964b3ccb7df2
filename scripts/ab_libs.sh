#!/bin/bash
# same-box A/B of two builds: $1 = variant lib path (relative to repo root)
V=$1
for i in 1 2 3; do
  for lib in "" "$V"; do
    if [ -z "$lib" ]; then tag=base; unset FLAIR_HIP_LIB; else tag=var; export FLAIR_HIP_LIB=$PWD/$lib; fi
    python bench.py --steps 30 --warmup 10 --no-alone --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$tag', d['ms_per_step'], d['roofline']['achieved'])"
  done
done
