#!/bin/bash
# round 6, GPU call 6: the API path (NSIM_FUSED_STEP=0) with the zero arena on / off, alternated; street / multi steps likewise
OUT=gpurun_out/r6_call6
mkdir -p $OUT
for rep in 1 2; do
  for a in 1 0; do
    NSIM_ZERO_ARENA=$a NSIM_FUSED_STEP=0 python bench.py --steps 64 --warmup 16 --no-cpu-baseline --no-variants --no-parity > $OUT/api_arena${a}_$rep.json 2> $OUT/api_arena${a}_$rep.err
  done
done
python bench.py --steps 64 --warmup 16 --no-cpu-baseline --no-variants --no-parity > $OUT/fused.json 2> $OUT/fused.err
for a in 1 0; do
  NSIM_ZERO_ARENA=$a python bench.py --config street --steps 12 --warmup 6 > $OUT/street_arena$a.json 2> $OUT/street_arena$a.err
  NSIM_ZERO_ARENA=$a python bench.py --config multi --steps 12 --warmup 6 > $OUT/multi_arena$a.json 2> $OUT/multi_arena$a.err
done
python -m pytest tests/test_trainer.py tests/test_ray_query.py tests/test_renderer.py tests/test_convergence.py -q -m gpu -p no:cacheprovider > $OUT/tests.log 2>&1; echo "tests rc=$?"; tail -3 $OUT/tests.log
for f in $OUT/api_arena*.json $OUT/fused.json $OUT/street_arena*.json $OUT/multi_arena*.json; do python -c "
import json,sys
d=json.loads(open('$f').read().strip().splitlines()[-1]); print('$f'.split('/')[-1], d['ms_per_step'], d.get('ms_per_step_p50'), d.get('host_wait_ms_per_step'), d['config'].get('launch_chain'))
"; done
