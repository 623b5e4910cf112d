#!/bin/bash
# round 4, GPU call 5: GPU parity of the fused launches (render head, merge + draw) and their A/B on the headline step
R=/root/repo; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_pack_ops.py tests/test_sampling.py tests/test_sampling_fuzz.py tests/test_trainer.py tests/test_ray_query.py tests/test_fullsize_parity.py -m gpu -x -q > $O/c5_tests.log 2>&1; tail -2 $O/c5_tests.log
B="--steps 64 --warmup 24 --no-cpu-baseline --no-variants --no-parity"
NSIM_RENDER_HEAD=0 NSIM_FUSE_MERGE_UPSAMPLE=0 timeout 300 python bench.py $B > $O/c5_base.json 2> $O/c5_base.err
NSIM_RENDER_HEAD=1 NSIM_FUSE_MERGE_UPSAMPLE=0 timeout 300 python bench.py $B > $O/c5_head.json 2> $O/c5_head.err
NSIM_RENDER_HEAD=0 NSIM_FUSE_MERGE_UPSAMPLE=1 timeout 300 python bench.py $B > $O/c5_merge.json 2> $O/c5_merge.err
timeout 300 python bench.py $B > $O/c5_both.json 2> $O/c5_both.err
NSIM_RENDER_HEAD=0 NSIM_FUSE_MERGE_UPSAMPLE=0 timeout 300 python bench.py $B > $O/c5_base_b.json 2> $O/c5_base_b.err
timeout 300 python bench.py $B > $O/c5_both_b.json 2> $O/c5_both_b.err
python - <<'PY'
import json
for n in ("base","head","merge","both","base_b","both_b"):
    try:
        d=json.loads(open(f"/root/repo/gpurun_out/c5_{n}.json").read().strip().splitlines()[-1])
        print(n, d["ms_per_step"], d["step_ms"]["p50"], d.get("abi_calls_per_step"))
    except Exception as e:
        print(n, "ERR", e)
PY
