#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): the REFERENCE's own trainers, source unchanged, on this repository's kernels through the
# nr3d_lib shim.  The reference tree is not part of this repository: it rides along as git-ignored scratch
# (cp -r /root/reference gpurun_scratch/reference in the authoring container before the call).  Logs -> gpurun_out/ref_*.log.
#   object (8192 rays) / street / multi-object: iterations per second of the reference's loop
#   ddp2: code_single/tools/train.py --ddp as TWO ranks sharing the one GPU of the box (gloo; RCCL refuses duplicate devices),
#         constant learning rate, every rank dumps its replica (tools/run_reference_train.py --dump-replica-state)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
export NSIM_REFERENCE_ROOT=$R/gpurun_scratch/reference PYTHONWARNINGS=ignore
[ -d $NSIM_REFERENCE_ROOT ] || { echo "no reference tree under gpurun_scratch/"; exit 1; }
REFC=$NSIM_REFERENCE_ROOT/code_single/configs
COMMON="--training.i_val=-1 --training.i_save=-1 --training.i_backup=-1 --training.i_log=100"
OBJ="--config $REFC/object_centric/lotd_neus.dtu.230814.yaml --dataset_cfg.target=neuralsim_amd.dataio.SyntheticObjectDataset --dataset_cfg.param.n_frames=24 --dataset_cfg.param.image_hw=256 --num_rays=8192"
timeout 600 python tools/run_reference_train.py $OBJ --exp_dir /tmp/ref_obj8k --num_iters=300 $COMMON > $O/ref_object8k.log 2>&1; echo "object8k rc=$?" >> $O/ref_object8k.log; tail -c 600 $O/ref_object8k.log | tr '\r' '\n' | tail -3
# two ranks on the one GPU: LOCAL_RANK 0 for both (one device), gloo
rm -rf /tmp/ref_ddp /tmp/ref_ddp_dump
for r in 0 1; do
  RANK=$r LOCAL_RANK=0 WORLD_SIZE=2 MASTER_ADDR=127.0.0.1 MASTER_PORT=29655 NSIM_DIST_BACKEND=gloo timeout 600 python tools/run_reference_train.py --dump-replica-state /tmp/ref_ddp_dump --ddp $OBJ --exp_dir /tmp/ref_ddp --num_iters=120 --warmup_steps=0 --min_factor=1.0 $COMMON > $O/ref_ddp2_rank$r.log 2>&1 &
done
wait
python - <<'PY' > $O/ref_ddp2_compare.json 2>&1
import json, torch
a, b = (torch.load(f"/tmp/ref_ddp_dump/rank{k}.pt") for k in (0, 1))
params = [k for k in a["state"] if k.startswith("param:")]
diff = {k: float((a["state"][k].double() - b["state"][k].double()).abs().max()) for k in params}
print(json.dumps(dict(world=a["world"], find_unused_parameters=a["find_unused_parameters"], ddp_params=a["ddp_params"], n_param_tensors=len(params),
                      max_abs_difference_between_replicas=max(diff.values()), bit_identical=all(v == 0.0 for v in diff.values()),
                      moved=float(max(a["state"][k].abs().max() for k in params)))))
PY
cat $O/ref_ddp2_compare.json; tail -c 300 $O/ref_ddp2_rank0.log | tr '\r' '\n' | tail -2
timeout 900 python tools/run_reference_train.py --config $REFC/waymo/streetsurf/withmask_withlidar_joint.240219.yaml --exp_dir /tmp/ref_street \
  --dataset_cfg.target=neuralsim_amd.dataio.SyntheticStreetDataset --dataset_cfg.param.n_frames=16 --dataset_cfg.param.image_h=160 --dataset_cfg.param.image_w=240 \
  --dataset_cfg.param.lidar_beams=16384 "--scenebank_cfg.scenarios=[synthetic_street]" "--lidar_list=[lidar_TOP]" "--lidar_weight=[1.0]" \
  --assetbank_cfg.LearnableParams.model_params.enable_after=50 "--training.error_map.error_map_hw=[16,24]" \
  --num_iters=150 $COMMON > $O/ref_street.log 2>&1; echo "street rc=$?" >> $O/ref_street.log; tail -c 600 $O/ref_street.log | tr '\r' '\n' | tail -3
timeout 900 python tools/run_reference_train.py --script code_multi/tools/train.py --config "$NSIM_REFERENCE_ROOT/code_multi/configs/exps/fg_neus=permuto/all_occ.240201.yaml" --exp_dir /tmp/ref_multi \
  --dataset_cfg.target=neuralsim_amd.dataio.SyntheticStreetDataset --dataset_cfg.param.n_frames=16 --dataset_cfg.param.image_h=160 --dataset_cfg.param.image_w=240 \
  --dataset_cfg.param.lidar_beams=16384 --dataset_cfg.param.n_vehicles=8 "--scenebank_cfg.scenarios=[synthetic_street]" "--scenebank_cfg.load_class_names=[Street,Vehicle]" \
  "--lidar_list=[lidar_TOP]" "--lidar_weight=[1.0]" --assetbank_cfg.Vehicle.asset_params.initialize_cfg.num_iters=300 \
  --num_iters=100 $COMMON > $O/ref_multi.log 2>&1; echo "multi rc=$?" >> $O/ref_multi.log; tail -c 600 $O/ref_multi.log | tr '\r' '\n' | tail -3
