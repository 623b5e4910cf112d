#!/bin/bash
# round 6 (session 2), GPU call 3: loads behind the plane-image waits (k_field JDIR, k_field_bwd_j BJD), DPP integer scans of the
# single-workgroup rank / pack kernels: unit tests of the touched families, then the bench twice
OUT=gpurun_out/r6_s2_call3
mkdir -p $OUT
python -m pytest tests/test_pack_ops.py tests/test_sampling.py tests/test_field.py tests/test_ray_query.py -q -m gpu -p no:cacheprovider -x > $OUT/tests.log 2>&1
echo "tests rc=$?"; tail -2 $OUT/tests.log
B="--steps 64 --warmup 16 --no-cpu-baseline --no-variants --no-parity"
for rep in 1 2 3; do
  python bench.py $B > $OUT/bench_$rep.json 2> $OUT/bench_$rep.err
  python tools/variant.py run sc5 $B > $OUT/bench_sc5_$rep.json 2> $OUT/bench_sc5_$rep.err
  python tools/variant.py run sc6 $B > $OUT/bench_sc6_$rep.json 2> $OUT/bench_sc6_$rep.err
done
for f in $OUT/bench_*.json; do python -c "
import json
d=json.loads(open('$f').read().strip().splitlines()[-1]); k=d.get('kernels') or {}
print('$f'.split('/')[-1], d['ms_per_step'], d.get('ms_per_step_p50'), ' '.join(f\"{n.replace('nsim_','')}={v['avg_ms']}\" for n,v in k.items()))
"; done
