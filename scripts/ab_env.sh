#!/bin/bash
# same-box A/B of environment settings: each argument is one "VAR=value[,VAR2=value2]" setting ("-" = defaults); 3 interleaved rounds
for i in 1 2 3; do
  for setting in "$@"; do
    ( if [ "$setting" != "-" ]; then OLDIFS=$IFS; IFS=, ; for kv in $setting; do export "$kv"; done; IFS=$OLDIFS; fi
      python bench.py --steps ${STEPS:-30} --warmup ${WARMUP:-10} --no-alone --no-cpu-baseline $BENCH_ARGS 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$setting', d['ms_per_step'], d['roofline']['achieved'])" )
  done
done
