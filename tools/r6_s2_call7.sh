#!/bin/bash
# round 6 (session 2), GPU call 7: level-major gathers with the parity enumeration through per-axis operand tables (lotd_slots):
# default (slots, WJ at 5 waves) vs slots0 (runtime corner indices, as committed before) vs slots1w1 (slots, WJ at 99 registers / 4 waves)
OUT=gpurun_out/r6_s2_call7
mkdir -p $OUT
python -m pytest tests/test_field.py tests/test_sampling.py tests/test_ray_query.py tests/test_fullsize_parity.py -q -m gpu -p no:cacheprovider -x > $OUT/tests.log 2>&1
echo "tests rc=$?"; tail -2 $OUT/tests.log
B="--steps 64 --warmup 16 --no-cpu-baseline --no-variants --no-parity"
for rep in 1 2 3; do
  python bench.py $B > $OUT/bench_def_$rep.json 2> $OUT/bench_def_$rep.err
  python tools/variant.py run slots0 $B > $OUT/bench_slots0_$rep.json 2> $OUT/bench_slots0_$rep.err
  python tools/variant.py run slots1w1 $B > $OUT/bench_slots1w1_$rep.json 2> $OUT/bench_slots1w1_$rep.err
done
python bench.py --config street --steps 12 --warmup 6 > $OUT/street_def.json 2> $OUT/street_def.err
python tools/variant.py run slots0 --config street --steps 12 --warmup 6 > $OUT/street_slots0.json 2> $OUT/street_slots0.err
for f in $OUT/bench_*.json $OUT/street_*.json; do python -c "
import json
d=json.loads(open('$f').read().strip().splitlines()[-1]); k=d.get('kernels') or {}
print('$f'.split('/')[-1], d['ms_per_step'], d.get('ms_per_step_p50'), ' '.join(f\"{n.replace('nsim_','')}={v['avg_ms']}\" for n,v in k.items()))
"; done
