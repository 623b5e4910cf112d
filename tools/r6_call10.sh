#!/bin/bash
# round 6, GPU call 10: (a) scatter scan early exit (product) vs all six rounds (variant "noexit"); (b) gather parity probe
OUT=gpurun_out/r6_call10
mkdir -p $OUT
B="--steps 64 --warmup 16 --no-cpu-baseline --no-variants --no-parity"
for rep in 1 2; do
  python bench.py $B > $OUT/bench_exit_$rep.json 2> $OUT/bench_exit_$rep.err
  python tools/variant.py run noexit $B > $OUT/bench_noexit_$rep.json 2> $OUT/bench_noexit_$rep.err
  NSIM_GATHER_PARITY=1 python bench.py $B > $OUT/bench_gpar_$rep.json 2> $OUT/bench_gpar_$rep.err
done
python bench.py --config street --steps 12 --warmup 6 > $OUT/street_exit.json 2> $OUT/street_exit.err
python tools/variant.py run noexit --config street --steps 12 --warmup 6 > $OUT/street_noexit.json 2> $OUT/street_noexit.err
NSIM_GATHER_PARITY=1 python bench.py --config street --steps 12 --warmup 6 > $OUT/street_gpar.json 2> $OUT/street_gpar.err
for f in $OUT/bench_*.json $OUT/street_*.json; do python -c "
import json
d=json.loads(open('$f').read().strip().splitlines()[-1]); k=d.get('kernels') or {}; g=lambda n:(k.get(n) or {}).get('avg_ms'); print('$f'.split('/')[-1], d['ms_per_step'], d.get('ms_per_step_p50'), 'scatter', g('nsim_lotd_scatter'), 'scatter4', g('nsim_lotd4_scatter'), 'gather_lm', g('nsim_lotd_gather_lm'), 'field_fwd', g('nsim_field_fwd'))
"; done
