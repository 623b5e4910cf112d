#!/bin/bash
# round 4, GPU call 7: k_field MODE 3 with the features-only LDS image + direct dh/dx loads (two workgroups per CU): parity,
# then A/B against the 16 KB-image build (variant "nojdir") on the headline step, interleaved
R=/root/repo; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_field.py tests/test_trainer.py tests/test_fullsize_parity.py -m gpu -x -q > $O/c7_tests.log 2>&1; tail -2 $O/c7_tests.log
B="--steps 64 --warmup 24 --no-cpu-baseline --no-variants --no-parity"
for i in 1 2; do
  timeout 300 python tools/variant.py run nojdir $B > $O/c7_base_$i.json 2> $O/c7_base_$i.err
  timeout 300 python bench.py $B > $O/c7_jdir_$i.json 2> $O/c7_jdir_$i.err
done
NSIM_FWD_GRID=256 timeout 300 python bench.py $B > $O/c7_jdir_g256.json 2> $O/c7_jdir_g256.err
NSIM_FWD_GRID=768 timeout 300 python bench.py $B > $O/c7_jdir_g768.json 2> $O/c7_jdir_g768.err
python - <<'PY'
import json
for n in ("base_1","jdir_1","base_2","jdir_2","jdir_g256","jdir_g768"):
    try:
        d=json.loads(open(f"/root/repo/gpurun_out/c7_{n}.json").read().strip().splitlines()[-1])
        k=d.get("kernels",{})
        print(n, d["ms_per_step"], d["step_ms"]["p50"], k["nsim_field_fwd"]["avg_ms"], k["nsim_field_bwd_sdf"]["avg_ms"])
    except Exception as e:
        print(n, "ERR", e)
PY
