#!/bin/bash
# round 6 (session 2), GPU call 16: the sampling decoder k_field_sdf at 4 / 3 waves per SIMD (128 / 149 registers) vs the product (149, 3);
# the forward decoder's grid cap (NSIM_FWD_GRID, run time)
OUT=gpurun_out/r6_s2_call16
mkdir -p $OUT
B="--steps 64 --warmup 16 --no-cpu-baseline --no-variants --no-parity"
for rep in 1 2; do
  python bench.py $B > $OUT/bench_def_$rep.json 2> $OUT/bench_def_$rep.err
  for v in sdf4 sdf3; do
    python tools/variant.py run $v $B > $OUT/bench_${v}_$rep.json 2> $OUT/bench_${v}_$rep.err
  done
  for g in 256 1024; do
    NSIM_FWD_GRID=$g python bench.py $B > $OUT/bench_fwdgrid${g}_$rep.json 2> $OUT/bench_fwdgrid${g}_$rep.err
  done
done
for f in $OUT/bench_*.json; do python -c "
import json
d=json.loads(open('$f').read().strip().splitlines()[-1]); k=d.get('kernels') or {}
print('$f'.split('/')[-1], d['ms_per_step'], d.get('ms_per_step_p50'), ' '.join(f\"{n.replace('nsim_','')}={v['avg_ms']}\" for n,v in k.items() if 'sdf' in n or 'fwd' in n))
"; done
