#!/bin/bash
# round 6 (session 2), GPU call 14: the 4-D level-major gather of the distant model with one point per lane (G4_PTS=1) vs two;
# + the unit tests of the field / distant families on the tree with the one-point with-grad 3-D gather
OUT=gpurun_out/r6_s2_call14
mkdir -p $OUT
python -m pytest tests/test_field.py tests/test_distant.py tests/test_batched.py -q -m gpu -p no:cacheprovider -x > $OUT/tests.log 2>&1
echo "tests rc=$?"; tail -2 $OUT/tests.log
for rep in 1 2; do
  python bench.py --config street --steps 12 --warmup 6 > $OUT/street_def_$rep.json 2> $OUT/street_def_$rep.err
  python tools/variant.py run g4p1 --config street --steps 12 --warmup 6 > $OUT/street_g4p1_$rep.json 2> $OUT/street_g4p1_$rep.err
done
python bench.py --distant --steps 32 --warmup 8 --no-cpu-baseline --no-variants --no-parity > $OUT/distant_def.json 2> $OUT/distant_def.err
python tools/variant.py run g4p1 --distant --steps 32 --warmup 8 --no-cpu-baseline --no-variants --no-parity > $OUT/distant_g4p1.json 2> $OUT/distant_g4p1.err
for f in $OUT/street_*.json $OUT/distant_*.json; do python -c "
import json
d=json.loads(open('$f').read().strip().splitlines()[-1]); k=d.get('kernels') or {}
print('$f'.split('/')[-1], d['ms_per_step'], d.get('ms_per_step_p50'), ' '.join(f\"{n.replace('nsim_','')}={v['avg_ms']}\" for n,v in k.items() if 'distant' in n or 'fwd' in n))
"; done
