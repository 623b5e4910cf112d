mkdir -p gpurun_out
run() { # name, env...
  name=$1; shift
  env "$@" timeout 200 python bench.py --no-cpu-baseline --no-variants --no-parity > gpurun_out/c13_$name.json 2>/dev/null
  python - <<PY
import json
d=json.loads(open("gpurun_out/c13_$name.json").read().strip().splitlines()[-1])
k=d["kernels"]
print("%-28s %10.1f %7.3f %7.3f  sdf %.4f gather %.4f fwd %.4f bwd_sdf %.4f bwd_rad %.4f" % ("$name", d["value"], d["ms_per_step"], d["ms_per_step_p50"], k["nsim_field_sdf"]["avg_ms"], k["nsim_lotd_gather_lm"]["avg_ms"], k["nsim_field_fwd"]["avg_ms"], k["nsim_field_bwd_sdf"]["avg_ms"], k["nsim_field_bwd_rad"]["avg_ms"]))
PY
}
for rep in 1 2; do
run base_$rep A=1
run fuse_mu0_$rep NSIM_FUSE_MERGE_UPSAMPLE=0
run sdfgrid512_$rep NSIM_SDF_GRID=512
run sdfgrid1024_$rep NSIM_SDF_GRID=1024
run sdfgrid1536_$rep NSIM_SDF_GRID=1536
run fwdgrid512_$rep NSIM_FWD_GRID=512
run specfwd0_$rep NSIM_SPEC_FORWARD=0
run prefetch0_$rep NSIM_PREFETCH_STREAM=0
run bwdgrid512_$rep NSIM_SDF_BWD_GRID=512
done
