#!/bin/bash
# round 6 (session 2), GPU call 9: bf16 staging of the weight-gradient operands with two values per v_cvt_pk_bf16_f32
# (ds_write_b16 + ds_write_b16_d16_hi) vs one conversion per value (variant nopairs)
OUT=gpurun_out/r6_s2_call9
mkdir -p $OUT
python -m pytest tests/test_field.py tests/test_distant.py tests/test_sky.py -q -m gpu -p no:cacheprovider -x > $OUT/tests.log 2>&1
echo "tests rc=$?"; tail -2 $OUT/tests.log
B="--steps 64 --warmup 16 --no-cpu-baseline --no-variants --no-parity"
for rep in 1 2 3; do
  python bench.py $B > $OUT/bench_def_$rep.json 2> $OUT/bench_def_$rep.err
  python tools/variant.py run nopairs $B > $OUT/bench_nopairs_$rep.json 2> $OUT/bench_nopairs_$rep.err
done
python bench.py --config street --steps 12 --warmup 6 > $OUT/street_def.json 2> $OUT/street_def.err
python tools/variant.py run nopairs --config street --steps 12 --warmup 6 > $OUT/street_nopairs.json 2> $OUT/street_nopairs.err
for f in $OUT/bench_*.json $OUT/street_*.json; do python -c "
import json
d=json.loads(open('$f').read().strip().splitlines()[-1]); k=d.get('kernels') or {}
print('$f'.split('/')[-1], d['ms_per_step'], d.get('ms_per_step_p50'), ' '.join(f\"{n.replace('nsim_','')}={v['avg_ms']}\" for n,v in k.items()))
"; done
