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
needs_reference = pytest.mark.skipif(not CFG.exists(), reason="/root/reference is not present")

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
