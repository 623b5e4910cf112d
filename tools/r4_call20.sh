#!/bin/bash
# round 4, GPU call 20: the reference's trainers, unchanged, on the final tree (backward on the calling thread, run-aware sort)
R=/root/repo; O=$R/gpurun_out; mkdir -p $O; cd $R
export NSIM_REFERENCE_ROOT=$R/gpurun_scratch/reference PYTHONWARNINGS=ignore
REFC=$NSIM_REFERENCE_ROOT/code_single/configs
COMMON="--training.i_val=-1 --training.i_save=-1 --training.i_backup=-1 --training.i_log=100"
timeout 600 python tools/run_reference_train.py --config $REFC/object_centric/lotd_neus.dtu.230814.yaml --exp_dir /tmp/ref_obj8k \
  --dataset_cfg.target=neuralsim_amd.dataio.SyntheticObjectDataset --dataset_cfg.param.n_frames=24 --dataset_cfg.param.image_hw=256 \
  --num_rays=8192 --num_iters=300 $COMMON > $O/c20_ref_object8k.log 2>&1; echo "object8k rc=$?" >> $O/c20_ref_object8k.log; tail -c 600 $O/c20_ref_object8k.log | tr '\r' '\n' | tail -3
timeout 900 python tools/run_reference_train.py --config $REFC/waymo/streetsurf/withmask_withlidar_joint.240219.yaml --exp_dir /tmp/ref_street \
  --dataset_cfg.target=neuralsim_amd.dataio.SyntheticStreetDataset --dataset_cfg.param.n_frames=16 --dataset_cfg.param.image_h=160 --dataset_cfg.param.image_w=240 \
  --dataset_cfg.param.lidar_beams=16384 "--scenebank_cfg.scenarios=[synthetic_street]" "--lidar_list=[lidar_TOP]" "--lidar_weight=[1.0]" \
  --assetbank_cfg.LearnableParams.model_params.enable_after=50 "--training.error_map.error_map_hw=[16,24]" \
  --num_iters=150 $COMMON > $O/c20_ref_street.log 2>&1; echo "street rc=$?" >> $O/c20_ref_street.log; tail -c 600 $O/c20_ref_street.log | tr '\r' '\n' | tail -3
timeout 900 python tools/run_reference_train.py --script code_multi/tools/train.py --config "$NSIM_REFERENCE_ROOT/code_multi/configs/exps/fg_neus=permuto/all_occ.240201.yaml" --exp_dir /tmp/ref_multi \
  --dataset_cfg.target=neuralsim_amd.dataio.SyntheticStreetDataset --dataset_cfg.param.n_frames=16 --dataset_cfg.param.image_h=160 --dataset_cfg.param.image_w=240 \
  --dataset_cfg.param.lidar_beams=16384 --dataset_cfg.param.n_vehicles=8 "--scenebank_cfg.scenarios=[synthetic_street]" "--scenebank_cfg.load_class_names=[Street,Vehicle]" \
  "--lidar_list=[lidar_TOP]" "--lidar_weight=[1.0]" --assetbank_cfg.Vehicle.asset_params.initialize_cfg.num_iters=300 \
  --num_iters=100 $COMMON > $O/c20_ref_multi.log 2>&1; echo "multi rc=$?" >> $O/c20_ref_multi.log; tail -c 600 $O/c20_ref_multi.log | tr '\r' '\n' | tail -3
