#!/bin/bash
# round 6 (session 2), GPU call 5: k_field_bwd_j with two alternating staging sets (4 barriers per group) vs one set (8): bench x3, street
OUT=gpurun_out/r6_s2_call5
mkdir -p $OUT
python -m pytest tests/test_field.py -q -m gpu -p no:cacheprovider -x > $OUT/tests.log 2>&1
echo "tests rc=$?"; tail -2 $OUT/tests.log
B="--steps 64 --warmup 16 --no-cpu-baseline --no-variants --no-parity"
for rep in 1 2 3; do
  python bench.py $B > $OUT/bench_db_$rep.json 2> $OUT/bench_db_$rep.err
  python tools/variant.py run nodb $B > $OUT/bench_nodb_$rep.json 2> $OUT/bench_nodb_$rep.err
done
python bench.py --config street --steps 12 --warmup 6 > $OUT/street_db.json 2> $OUT/street_db.err
python tools/variant.py run nodb --config street --steps 12 --warmup 6 > $OUT/street_nodb.json 2> $OUT/street_nodb.err
for f in $OUT/bench_*.json $OUT/street_*.json; do python -c "
import json
d=json.loads(open('$f').read().strip().splitlines()[-1]); k=d.get('kernels') or {}
print('$f'.split('/')[-1], d['ms_per_step'], d.get('ms_per_step_p50'), ' '.join(f\"{n.replace('nsim_','')}={v['avg_ms']}\" for n,v in k.items()))
"; done
