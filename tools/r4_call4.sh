#!/bin/bash
# round 4, GPU call 4: gather de-duplication microbenchmark; host-side profile (cProfile) of the reference's trainer on the MI355X
R=/root/repo; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 120 ./tools/gather_dedup_bench > $O/gather_dedup_bench.txt 2>&1; cat $O/gather_dedup_bench.txt
export NSIM_REFERENCE_ROOT=$R/gpurun_scratch/reference PYTHONWARNINGS=ignore
cat > /tmp/prof_ref.py <<'PY'
import sys, cProfile, pstats
sys.path.insert(0, '/root/repo/tools'); sys.path.insert(0, '/root/repo')
import torch, time
import run_reference_train
pr = cProfile.Profile()
import tqdm as _tq
_orig = _tq.tqdm.update
state = dict(n=0, t0=None)
def upd(self, n=1):
    state["n"] += 1
    if state["n"] == 100:
        torch.cuda.synchronize(); pr.enable(); state["t0"] = time.perf_counter()
    if state["n"] == 300:
        torch.cuda.synchronize(); pr.disable(); print("[prof] 200 iterations in", time.perf_counter() - state["t0"], "s", flush=True)
    return _orig(self, n)
_tq.tqdm.update = upd
try:
    run_reference_train.main(sys.argv[1:])
finally:
    st = pstats.Stats(pr, stream=sys.stdout)
    st.sort_stats("tottime").print_stats(70)
    st.sort_stats("cumulative").print_stats(90)
PY
timeout 600 python /tmp/prof_ref.py --config $NSIM_REFERENCE_ROOT/code_single/configs/object_centric/lotd_neus.dtu.230814.yaml --exp_dir /tmp/ref_prof \
  --dataset_cfg.target=neuralsim_amd.dataio.SyntheticObjectDataset --dataset_cfg.param.n_frames=24 --dataset_cfg.param.image_hw=256 \
  --num_rays=8192 --num_iters=320 --training.i_val=-1 --training.i_save=-1 --training.i_backup=-1 --training.i_log=1000 > $O/c4_ref_cprofile.txt 2>&1
grep "\[prof\]" $O/c4_ref_cprofile.txt
