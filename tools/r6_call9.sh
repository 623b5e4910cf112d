#!/bin/bash
# round 6, GPU call 9: early exit of the scatter's segmented scan (product) against all six rounds (variant "noexit"), alternated
OUT=gpurun_out/r6_call9
mkdir -p $OUT
for rep in 1 2; do
  python bench.py --steps 64 --warmup 16 --no-cpu-baseline --no-variants --no-parity > $OUT/bench_exit_$rep.json 2> $OUT/bench_exit_$rep.err
  python tools/variant.py run noexit --steps 64 --warmup 16 --no-cpu-baseline --no-variants --no-parity > $OUT/bench_noexit_$rep.json 2> $OUT/bench_noexit_$rep.err
done
python bench.py --config street --steps 12 --warmup 6 > $OUT/street_exit.json 2> $OUT/street_exit.err
python tools/variant.py run noexit --config street --steps 12 --warmup 6 > $OUT/street_noexit.json 2> $OUT/street_noexit.err
for f in $OUT/bench_*.json $OUT/street_*.json; do python -c "
import json
d=json.loads(open('$f').read().strip().splitlines()[-1]); k=d.get('kernels') or {}; print('$f'.split('/')[-1], d['ms_per_step'], d.get('ms_per_step_p50'), 'scatter', (k.get('nsim_lotd_scatter') or {}).get('avg_ms'), 'scatter4', (k.get('nsim_lotd4_scatter') or {}).get('avg_ms'))
"; done
