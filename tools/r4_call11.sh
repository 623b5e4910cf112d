#!/bin/bash
# round 4, GPU call 11: the drop-in API path (renderer + autograd functions) as the timed loop: host wait + cProfile
R=/root/repo; O=$R/gpurun_out; mkdir -p $O; cd $R
export NSIM_FUSED_STEP=0 NSIM_AUTOGRAD_MT=0
B="--steps 128 --warmup 24 --no-cpu-baseline --no-variants --no-parity"
timeout 300 python bench.py $B > $O/c11_api.json 2> $O/c11_api.err
timeout 300 python -m cProfile -o /tmp/p.prof bench.py --steps 256 --warmup 24 --no-cpu-baseline --no-variants --no-parity > /dev/null 2> $O/c11_prof.err
python - > $O/c11_cprofile.txt <<'PY'
import pstats
p = pstats.Stats('/tmp/p.prof')
p.sort_stats('tottime').print_stats(60)
p.sort_stats('cumtime').print_stats(60)
PY
python - <<'PY'
import json
d=json.loads(open("/root/repo/gpurun_out/c11_api.json").read().strip().splitlines()[-1])
print(d["ms_per_step"], d["step_ms"], d["abi_calls_per_step"], d["host_wait_ms_per_step"], d["config"].get("launch_chain"))
PY
