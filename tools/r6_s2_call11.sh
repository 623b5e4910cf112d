#!/bin/bash
# round 6 (session 2), GPU call 11: the compose renderer's collect + sort as one launch (nsim_compose_collect_sort): parity tests on
# the device, the multi-object step (before this change: 12.76-12.89 ms, 89 C-ABI calls per step)
OUT=gpurun_out/r6_s2_call11
mkdir -p $OUT
for rep in 1 2 3; do
  python bench.py --config multi --steps 24 --warmup 8 > $OUT/multi_fused_$rep.json 2> $OUT/multi_fused_$rep.err
  NSIM_COMPOSE_FUSED=0 python bench.py --config multi --steps 24 --warmup 8 > $OUT/multi_steps3_$rep.json 2> $OUT/multi_steps3_$rep.err
done
for f in $OUT/multi_*.json; do python -c "
import json
d=json.loads(open('$f').read().strip().splitlines()[-1]); k=d.get('kernels') or {}
print('$f'.split('/')[-1], d['ms_per_step'], d.get('ms_per_step_p50'), d.get('abi_calls_per_step'), d.get('host_wait_ms_per_step'))
"; done
