#!/bin/bash
# round 6 (session 2), GPU call 10: field.hip with -mllvm -amdgpu-mfma-vgpr-form=1 (MFMA results in VGPRs where the allocator can:
# 594 -> 358 v_accvgpr moves in k_field_bwd_j<0,2,1>) vs the product build
OUT=gpurun_out/r6_s2_call10
mkdir -p $OUT
B="--steps 64 --warmup 16 --no-cpu-baseline --no-variants --no-parity"
for rep in 1 2 3; do
  python bench.py $B > $OUT/bench_def_$rep.json 2> $OUT/bench_def_$rep.err
  python tools/variant.py run vform $B > $OUT/bench_vform_$rep.json 2> $OUT/bench_vform_$rep.err
done
python bench.py --config street --steps 12 --warmup 6 > $OUT/street_def.json 2> $OUT/street_def.err
python tools/variant.py run vform --config street --steps 12 --warmup 6 > $OUT/street_vform.json 2> $OUT/street_vform.err
for f in $OUT/bench_*.json $OUT/street_*.json; do python -c "
import json
d=json.loads(open('$f').read().strip().splitlines()[-1]); k=d.get('kernels') or {}
print('$f'.split('/')[-1], d['ms_per_step'], d.get('ms_per_step_p50'), ' '.join(f\"{n.replace('nsim_','')}={v['avg_ms']}\" for n,v in k.items()))
"; done
