#!/bin/bash
# round 6 (session 2), GPU call 13: the with-grad level-major gather (k_lotd_gather_lm<., true>, 68 us, ~57 % VALU-busy by count) with
# fewer points per lane (more waves per SIMD) and / or the parity enumeration through operand tables: default (4 points, corner indices)
# vs wj2 (2 points) vs wj2s / wj1s / wj3s (2 / 1 / 3 points + slot tables)
OUT=gpurun_out/r6_s2_call13
mkdir -p $OUT
B="--steps 64 --warmup 16 --no-cpu-baseline --no-variants --no-parity"
for rep in 1 2; do
  python bench.py $B > $OUT/bench_def_$rep.json 2> $OUT/bench_def_$rep.err
  for v in wj2 wj2s wj1s wj3s; do
    python tools/variant.py run $v $B > $OUT/bench_${v}_$rep.json 2> $OUT/bench_${v}_$rep.err
  done
done
python bench.py --config street --steps 12 --warmup 6 > $OUT/street_def.json 2> $OUT/street_def.err
for v in wj2 wj2s; do python tools/variant.py run $v --config street --steps 12 --warmup 6 > $OUT/street_$v.json 2> $OUT/street_$v.err; done
for f in $OUT/bench_*.json $OUT/street_*.json; do python -c "
import json
d=json.loads(open('$f').read().strip().splitlines()[-1]); k=d.get('kernels') or {}
print('$f'.split('/')[-1], d['ms_per_step'], d.get('ms_per_step_p50'), ' '.join(f\"{n.replace('nsim_','')}={v['avg_ms']}\" for n,v in k.items() if 'fwd' in n or 'gather' in n))
"; done
