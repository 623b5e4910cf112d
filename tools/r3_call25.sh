#!/bin/bash
R=/root/repo; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_fullsize_parity.py tests/test_permuto.py -m gpu -q -k "permuto" > $O/c25_tests.log 2>&1
grep -E "passed|failed|^FAILED" $O/c25_tests.log
python bench.py --no-cpu-baseline > $O/c25_bench.json 2>$O/c25.err
python - <<PY
import json
d=json.loads(open("$O/c25_bench.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d.get("step_ms"), json.dumps(d.get("variants")))
for f in ("parity_fullsize_permuto_api_f32_compressed.json","parity_fullsize_permuto_api_fp16_compressed.json"):
    r=json.load(open("$O/"+f)); print(f, {k:r[k] for k in ("rays_with_other_count","img_depth_volume","img_normals_volume","fix_nablas","fix_grad_grid","fix_grad_h_appear","psnr_rgb_db","sdf_nograd_max")})
PY
