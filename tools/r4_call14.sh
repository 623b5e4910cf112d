#!/bin/bash
R=/root/repo; O=$R/gpurun_out; mkdir -p $O; cd $R
B="--steps 64 --warmup 24 --no-cpu-baseline --no-variants --no-parity"
timeout 300 python bench.py $B > $O/c14_base.json 2>/dev/null
timeout 300 python tools/variant.py run noj $B > $O/c14_noj.json 2>/dev/null
timeout 300 python tools/variant.py run pts2 $B > $O/c14_pts2.json 2>/dev/null
timeout 300 python bench.py $B > $O/c14_base2.json 2>/dev/null
python - <<'PY'
import json
for n in ("base","noj","pts2","base2"):
    try:
        d=json.loads(open(f"/root/repo/gpurun_out/c14_{n}.json").read().strip().splitlines()[-1]); k=d["kernels"]
        print(n, d["ms_per_step"], d["step_ms"]["p50"], "fwd", k["nsim_field_fwd"]["avg_ms"], "gather", k["nsim_lotd_gather_lm"]["avg_ms"], "bwd_sdf", k["nsim_field_bwd_sdf"]["avg_ms"])
    except Exception as e: print(n, "ERR", e)
PY
