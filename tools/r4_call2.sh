#!/bin/bash
# round 4, GPU call 2: atomic scope probe, GPU parity of the LDS-DMA sampling decoder, A/B of its variants on the headline step
R=/root/repo; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 120 ./tools/atomic_bench4 > $O/atomic_bench4.txt 2>&1; tail -9 $O/atomic_bench4.txt
timeout 600 python -m pytest tests/test_field.py tests/test_sampling.py tests/test_ray_query.py tests/test_fullsize_parity.py -m gpu -x -q > $O/c2_tests.log 2>&1; tail -2 $O/c2_tests.log
B="--steps 64 --warmup 24 --no-cpu-baseline --no-variants --no-parity"
NSIM_SDF_GLDS=0 timeout 300 python bench.py $B > $O/c2_base.json 2> $O/c2_base.err
timeout 300 python bench.py $B > $O/c2_glds2.json 2> $O/c2_glds2.err
timeout 300 python tools/variant.py run sdf_nbuf1 $B > $O/c2_nbuf1.json 2> $O/c2_nbuf1.err
timeout 300 python tools/variant.py run sdf_nbuf1_w3 $B > $O/c2_nbuf1_w3.json 2> $O/c2_nbuf1_w3.err
NSIM_SDF_GLDS=0 timeout 300 python bench.py $B > $O/c2_base_b.json 2> $O/c2_base_b.err
python - <<'PY'
import json
for n in ("base","glds2","nbuf1","nbuf1_w3","base_b"):
    try:
        d=json.loads(open(f"/root/repo/gpurun_out/c2_{n}.json").read().strip().splitlines()[-1])
        k=d.get("kernels",{})
        print(n, d["ms_per_step"], {a:(b.get("avg_ms") if isinstance(b,dict) else b) for a,b in k.items() if "sdf" in a or "gather" in a})
    except Exception as e:
        print(n, "ERR", e)
PY
