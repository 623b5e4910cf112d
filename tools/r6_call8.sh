#!/bin/bash
# round 6, GPU call 8: scatter with parity slots (NSIM_SCATTER_PARITY) x issue grouping (NSIM_SCATTER_GROUP), alternated
OUT=gpurun_out/r6_call8
mkdir -p $OUT
python -m pytest tests/test_field.py tests/test_distant.py tests/test_batched.py -q -m gpu -p no:cacheprovider > $OUT/tests_default.log 2>&1; echo "tests(default) rc=$?"; tail -1 $OUT/tests_default.log
NSIM_SCATTER_GROUP=1 python -m pytest tests/test_field.py tests/test_distant.py -q -m gpu -p no:cacheprovider > $OUT/tests_group1.log 2>&1; echo "tests(group1) rc=$?"; tail -1 $OUT/tests_group1.log
for rep in 1 2; do
  for cfg in "0 0" "1 0" "1 1"; do
    set -- $cfg
    NSIM_SCATTER_PARITY=$1 NSIM_SCATTER_GROUP=$2 python bench.py --steps 64 --warmup 16 --no-cpu-baseline --no-variants --no-parity > $OUT/bench_p$1_g$2_$rep.json 2> $OUT/bench_p$1_g$2_$rep.err
  done
done
for cfg in "0 0" "1 0" "1 1"; do
  set -- $cfg
  NSIM_SCATTER_PARITY=$1 NSIM_SCATTER_GROUP=$2 python bench.py --config street --steps 12 --warmup 6 > $OUT/street_p$1_g$2.json 2> $OUT/street_p$1_g$2.err
  NSIM_SCATTER_PARITY=$1 NSIM_SCATTER_GROUP=$2 python bench.py --distant --steps 32 --warmup 8 --no-cpu-baseline --no-variants --no-parity > $OUT/distant_p$1_g$2.json 2> $OUT/distant_p$1_g$2.err
done
for f in $OUT/bench_p*.json $OUT/street_p*.json $OUT/distant_p*.json; do python -c "
import json
d=json.loads(open('$f').read().strip().splitlines()[-1]); k=d.get('kernels') or {}; print('$f'.split('/')[-1], d['ms_per_step'], d.get('ms_per_step_p50'), 'scatter', (k.get('nsim_lotd_scatter') or {}).get('avg_ms'), 'scatter4', (k.get('nsim_lotd4_scatter') or {}).get('avg_ms'))
"; done
