#!/bin/bash
# GPU call 16: distant forward grid sweep; sdf grid 768; street / multi re-check
R=/root/repo; O=$R/gpurun_out; mkdir -p $O; cd $R
A="--steps 16 --warmup 8 --no-cpu-baseline --no-parity --no-variants"
for g in 1024 768 512 256; do
  NSIM_NERF_FWD_GRID=$g python bench.py --distant $A > $O/c16_nerfgrid_$g.json 2>>$O/c16.err
done
NSIM_SDF_GRID=768 python bench.py --steps 32 --warmup 16 --no-cpu-baseline --no-parity --no-variants > $O/c16_sdf768.json 2>>$O/c16.err
python bench.py --config street --steps 8 --warmup 4 --no-cpu-baseline --no-parity --no-variants > $O/c16_street.json 2>>$O/c16.err
for f in c16_nerfgrid_1024 c16_nerfgrid_768 c16_nerfgrid_512 c16_nerfgrid_256 c16_sdf768 c16_street; do python - <<PY
import json
d=json.loads(open("$O/$f.json").read().strip().splitlines()[-1])
print("$f", d["value"], d["ms_per_step"], {k:v["avg_ms"] for k,v in d["kernels"].items()})
PY
done
