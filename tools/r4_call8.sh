#!/bin/bash
# round 4, GPU call 8: how host-bound is the headline step?  host_wait_ms_per_step of the bench line + a cProfile of the loop
R=/root/repo; O=$R/gpurun_out; mkdir -p $O; cd $R
B="--steps 128 --warmup 24 --no-cpu-baseline --no-variants --no-parity"
timeout 300 python bench.py $B > $O/c8_bench.json 2> $O/c8_bench.err
timeout 300 python -m cProfile -o /tmp/p.prof bench.py --steps 256 --warmup 24 --no-cpu-baseline --no-variants --no-parity > $O/c8_prof_bench.json 2> $O/c8_prof.err
python - > $O/c8_cprofile.txt <<'PY'
import pstats
p = pstats.Stats('/tmp/p.prof')
p.sort_stats('tottime').print_stats(45)
PY
python - <<'PY'
import json
d=json.loads(open("/root/repo/gpurun_out/c8_bench.json").read().strip().splitlines()[-1])
print(d["ms_per_step"], d["step_ms"], d["abi_calls_per_step"], d["host_wait_ms_per_step"])
PY
head -70 $O/c8_cprofile.txt | tail -60
