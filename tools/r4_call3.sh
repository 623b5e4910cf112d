#!/bin/bash
# round 4, GPU call 3: (a) the REFERENCE's own trainers, source unchanged, on the MI355X -- its sources ride along as git-ignored
# scratch (gpurun_scratch/reference, copied right before the call and deleted right after: the GPU box has no /root/reference);
# (b) grid size A/B of the sampling decoder (LDS-DMA form: 128 VGPR, 42 KB LDS -> three workgroups fit a CU).
R=/root/repo; O=$R/gpurun_out; mkdir -p $O; cd $R
export NSIM_REFERENCE_ROOT=$R/gpurun_scratch/reference PYTHONWARNINGS=ignore
REFC=$NSIM_REFERENCE_ROOT/code_single/configs
COMMON="--training.i_val=-1 --training.i_save=-1 --training.i_backup=-1 --training.i_log=100"
# object-centric config at the YAML's own sizes (4096 rays / iteration, L = 16 T = 2^19 main model + NeRF++ distant model + image embeddings)
timeout 600 python tools/run_reference_train.py --config $REFC/object_centric/lotd_neus.dtu.230814.yaml --exp_dir /tmp/ref_obj \
  --dataset_cfg.target=neuralsim_amd.dataio.SyntheticObjectDataset --dataset_cfg.param.n_frames=24 --dataset_cfg.param.image_hw=256 \
  --num_iters=400 $COMMON > $O/c3_ref_object.log 2>&1; echo "object rc=$?" >> $O/c3_ref_object.log; tail -c 600 $O/c3_ref_object.log | tr '\r' '\n' | tail -3
# same, 8192 rays per iteration (BASELINE configs[1]'s ray count)
timeout 600 python tools/run_reference_train.py --config $REFC/object_centric/lotd_neus.dtu.230814.yaml --exp_dir /tmp/ref_obj8k \
  --dataset_cfg.target=neuralsim_amd.dataio.SyntheticObjectDataset --dataset_cfg.param.n_frames=24 --dataset_cfg.param.image_hw=256 \
  --num_rays=8192 --num_iters=300 $COMMON > $O/c3_ref_object8k.log 2>&1; echo "object8k rc=$?" >> $O/c3_ref_object8k.log; tail -c 600 $O/c3_ref_object8k.log | tr '\r' '\n' | tail -3
# street config at the YAML's own model sizes (32 Mi-parameter cuboid LoTD, distant, sky, lidar step, pose refinement after 50 it)
timeout 900 python tools/run_reference_train.py --config $REFC/waymo/streetsurf/withmask_withlidar_joint.240219.yaml --exp_dir /tmp/ref_street \
  --dataset_cfg.target=neuralsim_amd.dataio.SyntheticStreetDataset --dataset_cfg.param.n_frames=16 --dataset_cfg.param.image_h=160 --dataset_cfg.param.image_w=240 \
  --dataset_cfg.param.lidar_beams=16384 "--scenebank_cfg.scenarios=[synthetic_street]" "--lidar_list=[lidar_TOP]" "--lidar_weight=[1.0]" \
  --assetbank_cfg.LearnableParams.model_params.enable_after=50 "--training.error_map.error_map_hw=[16,24]" \
  --num_iters=150 $COMMON > $O/c3_ref_street.log 2>&1; echo "street rc=$?" >> $O/c3_ref_street.log; tail -c 600 $O/c3_ref_street.log | tr '\r' '\n' | tail -3
# multi-object config (code_multi trainer): street + 8 vehicles on the shared conditional permutohedral model
timeout 900 python tools/run_reference_train.py --script code_multi/tools/train.py --config "$NSIM_REFERENCE_ROOT/code_multi/configs/exps/fg_neus=permuto/all_occ.240201.yaml" --exp_dir /tmp/ref_multi \
  --dataset_cfg.target=neuralsim_amd.dataio.SyntheticStreetDataset --dataset_cfg.param.n_frames=16 --dataset_cfg.param.image_h=160 --dataset_cfg.param.image_w=240 \
  --dataset_cfg.param.lidar_beams=16384 --dataset_cfg.param.n_vehicles=8 "--scenebank_cfg.scenarios=[synthetic_street]" "--scenebank_cfg.load_class_names=[Street,Vehicle]" \
  "--lidar_list=[lidar_TOP]" "--lidar_weight=[1.0]" --assetbank_cfg.Vehicle.asset_params.initialize_cfg.num_iters=300 \
  --num_iters=100 $COMMON > $O/c3_ref_multi.log 2>&1; echo "multi rc=$?" >> $O/c3_ref_multi.log; tail -c 600 $O/c3_ref_multi.log | tr '\r' '\n' | tail -3
B="--steps 64 --warmup 24 --no-cpu-baseline --no-variants --no-parity"
for G in 512 768 1024; do
  NSIM_SDF_GRID=$G timeout 300 python bench.py $B > $O/c3_grid$G.json 2> $O/c3_grid$G.err
done
python - <<'PY'
import json
for n in ("grid512","grid768","grid1024"):
    try:
        d=json.loads(open(f"/root/repo/gpurun_out/c3_{n}.json").read().strip().splitlines()[-1])
        print(n, d["ms_per_step"], d["kernels"]["nsim_field_sdf"]["avg_ms"])
    except Exception as e:
        print(n, "ERR", e)
PY
