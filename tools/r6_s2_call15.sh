#!/bin/bash
# round 6 (session 2), GPU call 15: the no-grad level-major gathers (sampling pass, refresh) with 1 / 2 / 4 points per lane and the operand tables
OUT=gpurun_out/r6_s2_call15
mkdir -p $OUT
B="--steps 64 --warmup 16 --no-cpu-baseline --no-variants --no-parity"
for rep in 1 2; do
  python bench.py $B > $OUT/bench_def_$rep.json 2> $OUT/bench_def_$rep.err
  for v in p2 p2s p4s p1s; do
    python tools/variant.py run $v $B > $OUT/bench_${v}_$rep.json 2> $OUT/bench_${v}_$rep.err
  done
done
for f in $OUT/bench_*.json; do python -c "
import json
d=json.loads(open('$f').read().strip().splitlines()[-1]); k=d.get('kernels') or {}
print('$f'.split('/')[-1], d['ms_per_step'], d.get('ms_per_step_p50'), ' '.join(f\"{n.replace('nsim_','')}={v['avg_ms']}\" for n,v in k.items() if 'gather' in n or 'fwd' in n))
"; done
