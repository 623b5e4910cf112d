#!/bin/bash
# round 6, GPU call 7: scatter with consecutive issue grouping + cross-corner fold (NSIM_SCATTER_GROUP=1) against the default
OUT=gpurun_out/r6_call7
mkdir -p $OUT
NSIM_SCATTER_GROUP=1 python -m pytest tests/test_field.py tests/test_distant.py -q -m gpu -p no:cacheprovider > $OUT/tests_group1.log 2>&1; echo "tests(group1) rc=$?"; tail -2 $OUT/tests_group1.log
for rep in 1 2; do
  for g in 0 1; do
    NSIM_SCATTER_GROUP=$g python bench.py --steps 64 --warmup 16 --no-cpu-baseline --no-variants --no-parity > $OUT/bench_group${g}_$rep.json 2> $OUT/bench_group${g}_$rep.err
  done
done
for g in 0 1; do
  NSIM_SCATTER_GROUP=$g python bench.py --config street --steps 12 --warmup 6 > $OUT/street_group${g}.json 2> $OUT/street_group${g}.err
done
for f in $OUT/bench_group*.json $OUT/street_group*.json; do python -c "
import json
d=json.loads(open('$f').read().strip().splitlines()[-1]); r=d.get('roofline',{}); print('$f'.split('/')[-1], d['ms_per_step'], d.get('ms_per_step_p50'), r.get('avg_launch_ms'), (d.get('kernels') or {}).get('nsim_lotd_scatter'))
"; done
