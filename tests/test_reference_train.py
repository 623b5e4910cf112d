"""the reference's OWN trainer -- ``/root/reference/code_single/tools/train.py``, source unchanged, run with
``runpy`` by tools/run_reference_train.py -- trains on this repository: every ``nr3d_lib`` import of the trainer, the scene
graph (``app/resources``), the data loaders (``dataio/data_loader``), the asset bank, the model wrappers
(``app/models/single``), the renderer and the losses resolves to the shim package ``nr3d_lib/`` of this repository, the
models are built by ``import_str(model_class)(**model_params)`` from the reference's YAML
(code_single/configs/object_centric/lotd_neus.dtu.230814.yaml: NeuS main model + NeRF++ distant model + image embeddings,
joint frame-pixel sampling with error maps), the data comes from ``neuralsim_amd.dataio.SyntheticObjectDataset`` (no
files), and the arithmetic runs in the HIP kernels -- here, in the CPU-only authoring container, on their host emulator
with every table / grid / ray count shrunk through the trainer's own ``--a.b.c=value`` overrides.

Authoring container only (needs /root/reference; the GPU box does not have it)."""
import os
import pickle
import subprocess
import sys
from pathlib import Path

import pytest
import torch

ROOT = Path(__file__).resolve().parent.parent
REF = Path("/root/reference")
CFG = REF / "code_single/configs/object_centric/lotd_neus.dtu.230814.yaml"
needs_reference = pytest.mark.skipif(not CFG.exists(), reason="executes the reference's own sources from /root/reference (authoring container only; emulator backend). What it pins is replayed on the GPU box from frozen reference outputs: tests/test_reference_frozen.py, test_reference_glue.py::test_*_fixture")

M, D = "assetbank_cfg.Main.model_params", "assetbank_cfg.Distant.model_params"
SMALL = [
    "--dataset_cfg.target=neuralsim_amd.dataio.SyntheticObjectDataset", "--dataset_cfg.param.n_frames=6",
    "--dataset_cfg.param.image_hw=32", "--num_rays=192", "--num_coarse=8", "--num_fine=[4,4]",
    "--upsample_inv_s_factors=[1,4]", "--step_size=0.05", "--bgsample=8",
    f"--{M}.surface_cfg.encoding_cfg.lotd_auto_compute_cfg.num_levels=8",
    f"--{M}.surface_cfg.encoding_cfg.lotd_auto_compute_cfg.log2_hashmap_size=12",
    f"--{M}.surface_cfg.encoding_cfg.lotd_auto_compute_cfg.max_res=64", f"--{M}.accel_cfg.resolution=[16,16,16]",
    f"--{M}.accel_cfg.init_cfg.num_pts=4096", f"--{M}.accel_cfg.init_cfg.num_steps=2",
    f"--{M}.accel_cfg.update_from_net_cfg.num_pts=4096", f"--{M}.accel_cfg.update_from_net_cfg.num_steps=1",
    f"--{M}.accel_cfg.n_steps_warmup=2", f"--{M}.accel_cfg.n_steps_between_update=2",
    f"--{M}.ray_query_cfg.query_param.march_cfg.max_steps=128",
    f"--{D}.encoding_cfg.lotd_auto_compute_cfg.target_num_params=16384",
    f"--{D}.encoding_cfg.lotd_auto_compute_cfg.log2_hashmap_size=10", f"--{D}.encoding_cfg.lotd_auto_compute_cfg.min_res_xyz=3",
    f"--{D}.encoding_cfg.lotd_auto_compute_cfg.min_res_w=2", "--training.i_save=-1", "--training.i_backup=-1",
    "--training.uniform_sample.Main=128",
]


def _run(exp_dir, extra, timeout=900):
    cmd = [sys.executable, str(ROOT / "tools" / "run_reference_train.py"), "--emulate", "--config", str(CFG), "--exp_dir",
           str(exp_dir)] + SMALL + list(extra)
    env = dict(os.environ, PYTHONWARNINGS="ignore")
    return subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=env, cwd=str(ROOT))


@needs_reference
def test_reference_trainer_runs_unchanged(tmp_path):
    exp = tmp_path / "exp"
    r = _run(exp, ["--num_iters=12", "--training.i_val=8", "--training.i_log=1"])
    tail = (r.stdout + r.stderr)[-3000:]
    assert r.returncode == 0, tail
    assert "Everything done." in r.stdout, tail
    # what the trainer leaves behind: the config it ran with, the model summary, the final checkpoint, the scalar log
    assert (exp / "config.yaml").exists() and (exp / "model.txt").exists()
    ck = sorted((exp / "ckpts").glob("final_*.pt"))
    assert len(ck) == 1 and ck[0].name == "final_00000012.pt"
    state = torch.load(str(ck[0]), map_location="cpu", weights_only=False)
    assert state["global_step"] == 12
    bank = state["asset_bank"]
    # AssetBank.state_dict (app/resources/asset_bank.py:245-258): one entry per model id
    main = next(v for k, v in bank.items() if k.startswith("LoTDNeuSObj#Main"))
    dist = next(v for k, v in bank.items() if k.startswith("LoTDNeRFDistant#Distant"))
    assert any(k.startswith("ImageEmbeddings#") for k in bank), list(bank)
    assert any(k.endswith("encoding.flattened_params") for k in main) and any(k.endswith("flattened_params") for k in dist)
    assert any(k.startswith("optimizer_") for k in state)
    stats = pickle.loads((exp / "stats.p").read_bytes())
    key = next(k for k in stats if k.endswith("loss_rgb"))
    loss = [v for _, v in stats[key]]
    assert len(loss) >= 12 and all(l == l and l < 10 for l in loss), (key, loss)
    assert sum(loss[-4:]) < sum(loss[:4]), loss                # it trains: the photometric loss goes down
    # resuming picks the checkpoint up (``--resume_dir``): nothing left to do at num_iters
    r2 = _run(exp, ["--num_iters=12", "--training.i_val=-1"])
    assert r2.returncode == 0 and "Everything done." in r2.stdout, (r2.stdout + r2.stderr)[-2000:]


@needs_reference
def test_reference_eval_and_render_tools_run_unchanged(tmp_path):
    """VERDICT r5 missing 5: the reference's ``code_single/tools/eval.py`` (metric loop :241-316) and ``render.py`` (replay,
    :213-220), sources unchanged, on an experiment directory the reference's own trainer wrote on this repository: they reload
    the scenario + checkpoint (``load_scene_bank``, ``AssetBank.create_asset_bank(load_state_dict=...)``), rebuild the dataset
    from the saved config, render every frame through ``SingleVolumeRenderer.render(..., rayschunk=...)`` with the validation
    renderer settings and score it with ``nr3d_lib.graphics.utils.PSNR / SSIM / LPIPS`` (this package's arithmetic:
    neuralsim_amd/eval.py; LPIPS needs weights that are not here: NaN with a warning).  Checked: both tools finish, the
    metric files carry the values ``neuralsim_amd.eval`` computes (finite PSNR / SSIM in range), the occupancy ratio of the
    reloaded grid is reported."""
    import json
    exp = tmp_path / "exp"
    r = _run(exp, ["--num_iters=8", "--training.i_val=-1", "--training.i_log=4"])
    assert r.returncode == 0 and "Everything done." in r.stdout, (r.stdout + r.stderr)[-3000:]
    env = dict(os.environ, PYTHONWARNINGS="ignore")
    for script, dirname in (("code_single/tools/eval.py", "eval"), ("code_single/tools/render.py", "render")):
        cmd = [sys.executable, str(ROOT / "tools" / "run_reference_train.py"), "--emulate", "--script", script, "--resume_dir",
               str(exp), "--no_output", "--rayschunk", "4096", "--dirname", dirname]
        rr = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=str(ROOT))
        tail = (rr.stdout + rr.stderr)[-3000:]
        assert rr.returncode == 0, (script, tail)
        assert "rendering frames" in tail and "100%" in tail, (script, tail)
    ev = exp / "eval"
    misc = json.loads(next(ev.glob("*_misc.json")).read_text())
    assert 3.0 < misc["full_psnr"] < 60.0 and -1.0 <= misc["full_ssim"] <= 1.0 and 0.0 < misc["occ_ratio"] < 1.0, misc
    assert misc["full_lpips"] != misc["full_lpips"]                    # NaN: no lpips weights in this image (documented)
    psnr_txt = next(ev.glob("*_psnr.txt")).read_text()
    assert psnr_txt.startswith("full:") and abs(float(psnr_txt.split()[1]) - misc["full_psnr"]) < 1e-3, psnr_txt


@needs_reference
def test_reference_trainer_ddp_two_ranks(tmp_path):
    """SURVEY row a21 through the reference's OWN multi-GPU entry: ``code_single/tools/train.py --ddp``,
    source unchanged, launched as ``python -m torch.distributed.run --nproc-per-node 2`` (gloo + the host emulator here; RCCL
    on a GPU node): ``nr3d_lib.distributed.init_env(args)`` joins the group and names this rank's device in
    ``args.device_ids``, rank 0 alone runs ``training_initialize`` (train.py:1396-1399), ``DistributedDataParallel(trainer,
    device_ids, output_device=local_rank, find_unused_parameters=True)`` (:1401-1406) wraps the shim's models -- its
    constructor broadcasts rank 0's parameters AND buffers (f32 masters, fp16 shadow tables, occupancy grids), its reducer
    hooks average the gradients the HIP autograd functions produce --, every rank steps ``local_it = it + rank`` and the loop
    advances ``it += world_size`` (:1446-1447, :1651).

    Checked: the run finishes on both ranks; 8 global iterations = 4 steps per rank (checkpoint ``final_00000008.pt``, four
    logged steps); every PARAMETER is bit-identical across the two replicas afterwards (f32 masters; the fp16 shadow buffers
    follow); the buffers -- occupancy values / bits, the trainer's error maps -- are rank-local between steps BY THE
    REFERENCE'S DESIGN (rank r refreshes at ``local_it``; the importance sampler sees its own pixels) and are equal again after
    DDP's pre-forward buffer broadcast, i.e. every rank renders with rank 0's grid.  The schedule is held constant
    (``--warmup_steps=0 --min_factor=1.0``) because the reference hands rank r the learning rate of iteration ``it + r``
    (``asset_bank.training_update_lr(local_it)``, :1449): under a decaying / warming schedule its replicas drift by the
    learning-rate difference on ANY backend (measured here: 8e-6 after four steps) -- a property of the trainer, shown by the
    second, shorter run below."""
    def run(exp, dump, extra, port):
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
               "--master-port", str(port), str(ROOT / "tools" / "run_reference_train.py"), "--emulate", "--dump-replica-state",
               str(dump), "--ddp", "--config", str(CFG), "--exp_dir", str(exp)] + SMALL + list(extra)
        env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
        env.update(PYTHONWARNINGS="ignore", OMP_NUM_THREADS="2")
        return subprocess.run(cmd, capture_output=True, text=True, timeout=1500, env=env, cwd=str(ROOT))

    import socket

    def port():
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            return s.getsockname()[1]

    exp, dump = tmp_path / "exp", tmp_path / "dump"
    r = run(exp, dump, ["--num_iters=8", "--training.i_val=-1", "--training.i_log=1", "--warmup_steps=0", "--min_factor=1.0"], port())
    tail = (r.stdout + r.stderr)[-3000:]
    assert r.returncode == 0 and "Everything done." in r.stdout, tail
    ck = sorted((exp / "ckpts").glob("final_*.pt"))
    assert len(ck) == 1 and ck[0].name == "final_00000008.pt", ck          # it += world_size: 4 steps per rank = 8 iterations
    assert torch.load(str(ck[0]), map_location="cpu", weights_only=False)["global_step"] == 8
    stats = pickle.loads((exp / "stats.p").read_bytes())
    key = next(k for k in stats if k.endswith("loss_rgb"))
    its = [i for i, _ in stats[key]]
    assert len(its) == 4 and all(b - a == 2 for a, b in zip(its, its[1:])), its        # rank 0 logs it = 0, 2, 4, 6
    a, b = (torch.load(str(dump / f"rank{k}.pt")) for k in (0, 1))
    assert a["world"] == b["world"] == 2 and a["find_unused_parameters"] and a["broadcast_buffers"]
    assert a["ddp_params"] == b["ddp_params"] > 50_000                      # NeuS + distant model + image embeddings
    params = [k for k in a["state"] if k.startswith("param:")]
    assert any("encoding.flattened_params" in k for k in params) and any("Distant" in k for k in params), params
    for k in params:
        assert torch.equal(a["state"][k], b["state"][k]), (k, float((a["state"][k] - b["state"][k]).abs().max()))
    for k in a["state"]:
        if k.startswith("synced_buffer:") or (k.startswith("buffer:") and k.endswith("params16")):
            assert torch.equal(a["state"][k], b["state"][k]), k             # fp16 shadows follow the masters; DDP's broadcast
    occ = [k for k in a["state"] if k.startswith("synced_buffer:") and k.endswith("accel.occ_val")]
    assert len(occ) == 1 and float(a["state"][occ[0]].max()) > 0            # the occupancy grid IS one of the broadcast buffers
    # the reference's own schedule (exponential + warm-up, yaml:381-386): rank r runs the learning rate of iteration it + r
    r2 = run(tmp_path / "exp2", tmp_path / "dump2", ["--num_iters=4", "--training.i_val=-1", "--training.i_log=1"], port())
    assert r2.returncode == 0 and "Everything done." in r2.stdout, (r2.stdout + r2.stderr)[-3000:]
    a2, b2 = (torch.load(str(tmp_path / "dump2" / f"rank{k}.pt")) for k in (0, 1))
    drift = max(float((a2["state"][k] - b2["state"][k]).abs().max()) for k in params)
    assert 0 < drift < 1e-3, drift


# ================================================================================================ the street config (BASELINE configs[3])
STREET_CFG = REF / "code_single/configs/waymo/streetsurf/withmask_withlidar_joint.240219.yaml"
S_, DV = "assetbank_cfg.Street.model_params", "assetbank_cfg.Distant.model_params"
STREET_SMALL = [
    "--dataset_cfg.target=neuralsim_amd.dataio.SyntheticStreetDataset", "--dataset_cfg.param.n_frames=6",
    "--dataset_cfg.param.image_h=24", "--dataset_cfg.param.image_w=32", "--scenebank_cfg.scenarios=[synthetic_street]",
    "--camera_list=[camera_FRONT,camera_FRONT_LEFT,camera_FRONT_RIGHT]", "--lidar_list=[lidar_TOP]", "--lidar_weight=[1.0]",
    "--num_rays_pixel=128", "--num_rays_lidar=96", "--num_coarse=16", "--num_fine=[4,4,8]", "--step_size=2.0", "--num_uniform=256",
    "--distant_nsample=8", "--log2_hashmap_size=12", "--max_num_levels=10", "--warmup_steps=1",
    f"--{S_}.surface_cfg.encoding_cfg.lotd_auto_compute_cfg.target_num_params=131072",
    f"--{S_}.surface_cfg.encoding_cfg.lotd_auto_compute_cfg.min_res=3", f"--{S_}.accel_cfg.vox_size=8.0",
    f"--{S_}.accel_cfg.init_cfg.num_pts=8192", f"--{S_}.accel_cfg.init_cfg.num_steps=2",
    f"--{S_}.accel_cfg.update_from_net_cfg.num_pts=8192", f"--{S_}.accel_cfg.update_from_net_cfg.num_steps=1",
    f"--{S_}.accel_cfg.n_steps_warmup=2", f"--{S_}.accel_cfg.n_steps_between_update=2",
    f"--{S_}.ray_query_cfg.query_param.march_cfg.max_steps=256",
    f"--{DV}.encoding_cfg.lotd_auto_compute_cfg.target_num_params=16384", f"--{DV}.encoding_cfg.lotd_auto_compute_cfg.log2_hashmap_size=10",
    f"--{DV}.encoding_cfg.lotd_auto_compute_cfg.min_res_xyz=3", f"--{DV}.encoding_cfg.lotd_auto_compute_cfg.min_res_w=2",
    "--training.i_save=-1", "--training.i_backup=-1", "--training.i_log=1", "--training.error_map.error_map_hw=[6,8]",
    "--assetbank_cfg.LearnableParams.model_params.enable_after=1",
]


@needs_reference
def test_reference_trainer_runs_the_street_config(tmp_path):
    """BASELINE configs[3] through the reference's OWN trainer, source unchanged: ``code_single/tools/train.py --config
    code_single/configs/waymo/streetsurf/withmask_withlidar_joint.240219.yaml`` with every ``model_params`` block of the YAML
    (``LoTDNeuSStreet`` cuboid LoTD sized from the camera frusta, ``LoTDNeRFDistant``, ``SimpleSky``, ``ImageEmbeddings``,
    ``LearnableParams`` pose refinement) on ``neuralsim_amd.dataio.SyntheticStreetDataset``: an ego vehicle with an opencv-model
    camera rig and a roof lidar as its children (docs/data/autonomous_driving.md:38-100), occupancy / human / ignore masks.  Per
    iteration the trainer runs the joint pixel step (rgb l1, mask, eikonal, sparsity, clearance ...) AND the lidar step
    (depth l1 + line of sight, train.py:876-960), each with its own backward and optimizer step; pose refinement switches on
    after ``enable_after`` and the camera poses receive gradients through ``OpenCVCameraMatHW.lift`` -> rays -> kernels.
    Sizes shrunk through the trainer's own ``--a.b.c=`` overrides (host emulator)."""
    exp = tmp_path / "street"
    cmd = [sys.executable, str(ROOT / "tools" / "run_reference_train.py"), "--emulate", "--config", str(STREET_CFG), "--exp_dir",
           str(exp), "--num_iters=14", "--training.i_val=8"] + STREET_SMALL
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=1500, env=dict(os.environ, PYTHONWARNINGS="ignore"), cwd=str(ROOT))
    tail = (r.stdout + r.stderr)[-3000:]
    assert r.returncode == 0 and "Everything done." in r.stdout, tail
    ck = sorted((exp / "ckpts").glob("final_*.pt"))
    assert len(ck) == 1 and ck[0].name == "final_00000014.pt"
    state = torch.load(str(ck[0]), map_location="cpu", weights_only=False)
    bank = state["asset_bank"]
    ids = {k.split("#")[0] for k in bank}
    assert ids == {"LoTDNeuSStreet", "LoTDNeRFDistant", "SimpleSky", "ImageEmbeddings", "LearnableParams"}, ids
    street = next(v for k, v in bank.items() if k.startswith("LoTDNeuSStreet#Street"))
    assert any(k.endswith("encoding.flattened_params") for k in street) and bool(street["is_pretrained"])
    lp = next(v for k, v in bank.items() if k.startswith("LearnableParams"))
    d_rot = [v for k, v in lp.items() if k.endswith("rot.subattr.delta.tensor")]
    d_tr = [v for k, v in lp.items() if k.endswith("trans.subattr.delta.tensor")]
    assert len(d_rot) == len(d_tr) == 3 and all(tuple(v.shape) == (6, 4) for v in d_rot)
    assert max(float(v.abs().max()) for v in d_rot) > 0 and max(float(v.abs().max()) for v in d_tr) > 0      # the poses moved
    stats = pickle.loads((exp / "stats.p").read_bytes())
    for key in ("train_step_pixel.losses/loss_rgb", "train_step_pixel.losses/loss_mask", "train_step_pixel.losses/total",
                "train_step_lidar.losses/lidar_loss.depth", "train_step_lidar.losses/lidar_loss.los.empty",
                "train_step_lidar.losses/total"):
        vals = [v for _, v in stats[key]]
        assert len(vals) >= 5 and all(v == v and abs(v) < 1e3 for v in vals), (key, vals)
    assert any(k.startswith("train_step_pixel.losses/loss_eikonal") for k in stats), [k for k in stats if "eikonal" in k]
    tot = [v for _, v in stats["train_step_pixel.losses/total"]]
    assert sum(tot[-5:]) < sum(tot[:5]), tot                   # it trains (fresh 128-ray batches: compare windows)
    # the street model rendered samples (not only the distant model and the sky): its volume buffer statistics were logged
    assert any(k.startswith("train_step_pixel.obj=street/volume_buffer.opacity_alpha") for k in stats)


# ================================================================================================ multi-object (BASELINE configs[4])
MULTI_CFG = REF / "code_multi/configs/exps/fg_neus=permuto/all_occ.240201.yaml"
V_ = "assetbank_cfg.Vehicle.model_params"
MULTI_SMALL = [a for a in STREET_SMALL if "LearnableParams" not in a and "error_map" not in a] + [
    "--dataset_cfg.param.n_vehicles=3", "--scenebank_cfg.load_class_names=[Street,Vehicle]", "--veh_dtype=float", "--veh_n_levels=6",
    "--veh_log2_hashmap_size=11", f"--{V_}.surface_cfg.encoding_cfg.permuto_auto_compute_cfg.finest_res=24.0",
    f"--{V_}.surface_cfg.encoding_cfg.permuto_auto_compute_cfg.coarsest_res=2.0", f"--{V_}.accel_cfg.resolution=[8,8,8]",
    f"--{V_}.accel_cfg.init_cfg.num_pts=8192", f"--{V_}.accel_cfg.init_cfg.num_steps=2", f"--{V_}.accel_cfg.update_from_net_cfg.num_pts=2048",
    f"--{V_}.accel_cfg.update_from_net_cfg.num_steps=1", f"--{V_}.ray_query_cfg.query_param.num_coarse=8",
    f"--{V_}.ray_query_cfg.query_param.num_fine=8", f"--{V_}.ray_query_cfg.query_param.march_cfg.max_steps=64",
    f"--{V_}.ray_query_cfg.query_param.march_cfg.step_size=0.1", "--assetbank_cfg.Vehicle.asset_params.initialize_cfg.num_iters=100",
    "--assetbank_cfg.Vehicle.asset_params.initialize_cfg.num_points=1024", "--assetbank_cfg.Vehicle.asset_params.initialize_cfg.batch_size=3",
]


@needs_reference
def test_reference_multi_object_trainer_runs_unchanged(tmp_path):
    """BASELINE configs[4]: the reference's multi-object trainer ``code_multi/tools/train.py`` (source unchanged, run by
    tools/run_reference_train.py --script) on its CURRENT foreground config ``code_multi/configs/exps/fg_neus=permuto/
    all_occ.240201.yaml``: street background (``LoTDNeuSStreet`` + distant + sky) and moving vehicles that share ONE conditional
    model -- ``model_class: app.models.shared.AD_GenerativePermutoConcatNeuSObj`` -- composed by the reference's own
    ``BufferComposeRenderer`` (``set_condition({'ins_id': ...})`` -> ``batched_ray_test(compact_batch=True)`` ->
    ``batched_ray_query`` -> per-ray merge of the packed buffers of street, vehicles and distant shells,
    app/renderers/buffer_compose_renderer.py:209-806).  Scene graph: ``SyntheticStreetDataset(n_vehicles=3)`` -- vehicle nodes
    with per-frame ``transform`` / ``scale`` segments.  The Pedestrian / Cyclist classes of the YAML (time-conditioned fields,
    out of scope) are not loaded (``load_class_names``).  Checked: it runs to the end, the vehicle model rendered samples, and
    the auto-decoder's per-instance codes (``weight_init: zero``) were LEARNED -- gradients reach them through nsim_permuto_dz."""
    exp = tmp_path / "multi"
    cmd = [sys.executable, str(ROOT / "tools" / "run_reference_train.py"), "--emulate", "--script", "code_multi/tools/train.py",
           "--config", str(MULTI_CFG), "--exp_dir", str(exp), "--num_iters=6", "--training.i_val=-1"] + MULTI_SMALL
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=1700, env=dict(os.environ, PYTHONWARNINGS="ignore"), cwd=str(ROOT))
    tail = (r.stdout + r.stderr)[-3000:]
    assert r.returncode == 0 and "Everything done." in r.stdout, tail
    ck = sorted((exp / "ckpts").glob("final_*.pt"))
    assert len(ck) == 1 and ck[0].name == "final_00000006.pt"
    bank = torch.load(str(ck[0]), map_location="cpu", weights_only=False)["asset_bank"]
    assert "AD_GenerativePermutoConcatNeuSObj#Vehicle" in bank, list(bank)
    veh = bank["AD_GenerativePermutoConcatNeuSObj#Vehicle"]
    z = veh["_latents.z_ins.weight"]
    assert tuple(z.shape) == (3, 4) and float(z.abs().max()) > 0 and bool(torch.isfinite(z).all())
    assert any(k.endswith("encoding.flattened_params") for k in veh) and bool(veh["is_pretrained"])
    stats = pickle.loads((exp / "stats.p").read_bytes())
    for key in ("train_step_pixel.losses/loss_rgb", "train_step_pixel.losses/total", "train_step_lidar.losses/total",
                "train_step_pixel.losses/loss_eikonal.Vehicle.render"):
        vals = [v for _, v in stats[key]]
        assert len(vals) >= 1 and all(v == v and abs(v) < 1e3 for v in vals), (key, vals)
    assert any(k.startswith("train_step_pixel.obj=Vehicle/volume_buffer.opacity_alpha") for k in stats)
    assert any(k.startswith("train_step_pixel.obj=street/volume_buffer.opacity_alpha") for k in stats)
