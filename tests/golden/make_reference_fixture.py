"""Generate tests/golden/reference_fixture.pt: outputs of the REFERENCE's own sources (loaded unchanged from /root/reference by
tests/ref_glue.py) on seeded inputs, frozen so that the GPU box -- where /root/reference does not exist -- can replay the
reference-pinned checks of tests/test_reference_glue.py / tests/test_reference_configs.py:

  cameras             ``Camera._get_selected_rays_from_ixy`` / ``get_all_rays`` (app/resources/observers/cameras.py:281-310,
                      332-360) on pinhole, OpenCV (distorted) and fisheye rigs
  volume_integration  ``SingleVolumeRenderer._volume_integration`` (app/renderers/single_volume_renderer.py:73-102), both depth
                      modes, train / eval normals
  lidar               ``LineOfSightLoss`` (app/loss/lidar.py:57-206; nerf / neus_urban / neus_unisim) and the masked l1 depth term
                      on a frozen lidar volume buffer
  configs             the resolved ``model_params`` blocks of the three YAMLs of the hot path (lotd_neus.dtu / replica,
                      withmask_withlidar_joint: Street, Distant, Sky) as ``nr3d_lib.config.load_config`` of the authoring
                      container resolves them -- hyper-parameters only -- and nothing else of the files

Authoring container only:   python tests/golden/make_reference_fixture.py
The replay lives in tests/test_reference_frozen.py (runs on both backends, no /root/reference)."""
import ctypes
import sys
from pathlib import Path

import torch

HERE = Path(__file__).resolve().parent
ROOT = HERE.parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "tests"), str(ROOT / "tests" / "emu")]

import build_emu  # noqa: E402
import ref_glue  # noqa: E402
from neuralsim_amd import _lib  # noqa: E402


def cameras():
    from util import look_at_cameras
    intr, c2w, WH = look_at_cameras(V=5, seed=9, H=40, W=56, f=61.3)
    intr[:, 0, 2] += 0.37                                 # principal point off the pixel grid
    intr[:, 1, 1] *= 1.03
    g = torch.Generator().manual_seed(2)
    N = 600
    xy = torch.rand(N, 2, generator=g).clamp(1e-6, 1 - 1e-6)
    xy[:4] = torch.tensor([[0.0, 0.0], [1.0, 1.0], [1.0, 0.0], [0.999999, 0.5]])      # the clamp at the image border
    fidx = torch.randint(0, 5, (N,), generator=g)
    dist_cv = torch.tensor([[0.043, -0.36, 0.0008, -0.0006, 0.01]]).repeat(5, 1) * torch.linspace(0.5, 1.5, 5)[:, None]
    # fisheye rig: f = 14 px on a 56 x 40 image -> rays up to ~70 degrees off axis (fisheye.py:22's coefficients)
    intr_f = intr.clone()
    intr_f[:, 0, 0], intr_f[:, 1, 1] = 14.0, 14.4
    dist_fe = torch.tensor([[0.18198802503702904, -0.04198598106075817, 0.010013633995507613, -0.0025294664427881705]]).repeat(5, 1)
    out = dict(intr=intr, c2w=c2w, WH=WH, xy=xy, fidx=fidx, dist_opencv=dist_cv, intr_fisheye=intr_f, dist_fisheye=dist_fe)
    with ref_glue.reference_camera_class() as Camera:
        cam = ref_glue.FakeCamera(intr, c2w, WH.float())
        for snap in (True, False):
            o, d = Camera._get_selected_rays_from_ixy(cam, fidx, xy, snap_to_pixel_centers=snap)
            out[f"pinhole_snap{int(snap)}"] = (o, d)
        cam_d = ref_glue.FakeCamera(intr, c2w, WH.float(), distortion=dist_cv)
        out["opencv"] = Camera._get_selected_rays_from_ixy(cam_d, fidx, xy, snap_to_pixel_centers=True)
        cam_f = ref_glue.FakeCamera(intr_f, c2w, WH.float(), distortion=dist_fe)
        out["fisheye"] = Camera._get_selected_rays_from_ixy(cam_f, fidx, xy, snap_to_pixel_centers=True)
        out["all_rays_frame2"] = Camera.get_all_rays(ref_glue.FakeCamera(intr[2], c2w[2], WH[2].float()))
    return out


def volume_integration():
    from oracle import pack_ops as opo
    g = torch.Generator().manual_seed(3)
    n = torch.tensor([5, 1, 0, 9, 3, 70, 0, 130])
    pi = opo.get_pack_infos_from_n(n)
    S = int(n.sum())
    alpha, t = torch.rand(S, generator=g) * 0.7, torch.rand(S, generator=g).cumsum(0)
    rgb, nab = torch.rand(S, 3, generator=g), torch.randn(S, 3, generator=g) * 1.5
    keep = n > 0
    pi_hit, rih = pi[keep], keep.nonzero()[:, 0]
    out = dict(n=n, alpha=alpha, t=t, rgb=rgb, nablas=nab, pack_infos_hit=pi_hit, rays_inds_hit=rih, N=int(n.shape[0]), cases={})
    with ref_glue.reference_renderer_modules() as mods:
        m = mods["app.renderers.single_volume_renderer"]
        saved = (m.packed_alpha_to_vw, m.packed_sum, m.packed_div)
        m.packed_alpha_to_vw = lambda a, pack_infos: opo.packed_alpha_to_vw(a, pack_infos)
        m.packed_sum, m.packed_div = opo.packed_sum, opo.packed_div
        try:
            for training in (True, False):
                for norm_depth in (True, False):
                    r = ref_glue.make_reference_renderer(mods, dict(depth_use_normalized_vw=norm_depth), training=training)
                    rendered = mods["app.renderers.utils"].prepare_empty_rendered([out["N"]], with_rgb=True, with_normal=True)
                    vb = dict(type="packed", rays_inds_hit=rih, pack_infos_hit=pi_hit, opacity_alpha=alpha.clone(), t=t, rgb=rgb,
                              nablas_in_world=nab.clone())
                    r._volume_integration(vb, rendered)
                    out["cases"][(training, norm_depth)] = dict(vw=vb["vw"].clone(), **{k: v.clone() for k, v in rendered.items()})
        finally:
            m.packed_alpha_to_vw, m.packed_sum, m.packed_div = saved
    return out


def lidar():
    """a lidar volume buffer of the mirror renderer (``with_rgb=False``) frozen WITH the reference's losses on it"""
    from neuralsim_amd.renderers.single_volume_renderer import SingleVolumeRenderer
    from renderer_scenario import build_scenario
    dev = torch.device("cpu")
    sc = build_scenario("main_train", dev)
    r = SingleVolumeRenderer(dict(with_rgb=False, with_normal=False, near=0.01, depth_use_normalized_vw=True, perturb=False)).train()
    with torch.no_grad():
        ret = r.ray_query(sc["rays_o"], sc["rays_d"], model=sc["model"], return_buffer=True)
    vb = ret["volume_buffer"]
    N = sc["N"]
    g = torch.Generator().manual_seed(8)
    depth = ret["rendered"]["depth_volume"].detach().clone()
    ranges = (depth + torch.randn(N, generator=g) * 0.05).clamp_min(0.1)
    ranges[::3] += 0.5                                  # returns BEHIND the rendered surface: visibility mass in the "empty" zone
    ranges[2::4] -= 0.3                                 # ... and in front of it
    ranges = ranges.clamp_min(0.1)
    ranges[::5] = 0.0                                   # beams without a return
    ranges[1::7] = 3.0                                  # ... and beyond discard_toofar
    mask = (ranges > 0) & (ranges < 2.5)
    frozen = dict(N=N, vw=vb["vw"].detach().clone(), t=vb["t"].detach().clone(), pack_infos_hit=vb["pack_infos_hit"].clone(),
                  rays_inds_hit=vb["rays_inds_hit"].clone(), depth_volume=depth, ranges=ranges, mask=mask, losses={})
    ret_f = dict(volume_buffer=dict(type="packed", vw=frozen["vw"], t=frozen["t"], pack_infos_hit=frozen["pack_infos_hit"],
                                    rays_inds_hit=frozen["rays_inds_hit"]), rendered=dict(depth_volume=depth))
    with ref_glue.reference_lidar_loss_module() as lidar_mod:
        for fn_type, param in (("nerf", dict(sigma=0.1)), ("neus_urban", dict(sigma=0.1)), ("neus_unisim", dict(epsilon=0.15))):
            los = lidar_mod.LineOfSightLoss(w=0.1, fn_type=fn_type, fn_param=param)
            for k, v in los(None, ret_f, None, dict(ranges=ranges), it=0, mask=mask).items():
                frozen["losses"][f"{fn_type}.{k}"] = float(v)
    from nr3d_lib.models.loss.recon import l1_loss
    frozen["losses"]["depth_l1_w0.02"] = float(0.02 * l1_loss(depth, ranges, mask, reduction="mean"))
    return frozen


def configs():
    from nr3d_lib.config import load_config
    CFG = Path("/root/reference/code_single/configs")
    out = {}
    for key, rel, blocks in (("dtu", "object_centric/lotd_neus.dtu.230814.yaml", ("Main", "Distant")),
                             ("replica", "indoor/lotd_neus.replica.230814.yaml", ("Main",)),
                             ("street", "waymo/streetsurf/withmask_withlidar_joint.240219.yaml", ("Street", "Distant", "Sky"))):
        c = load_config(str(CFG / rel))
        out[key] = dict(source=f"code_single/configs/{rel}", num_iters=int(c.training.num_iters),
                        distant_nsample=int(c.get("distant_nsample", 0) or 0),
                        blocks={b: dict(model_params=c.assetbank_cfg[b].model_params.to_dict(),
                                        initialize_cfg=(c.assetbank_cfg[b].get("asset_params", {}) or {}).get("initialize_cfg", None))
                                for b in blocks})
        for b in out[key]["blocks"].values():
            if b["initialize_cfg"] is not None and hasattr(b["initialize_cfg"], "to_dict"):
                b["initialize_cfg"] = b["initialize_cfg"].to_dict()
    return out


def main():
    assert ref_glue.reference_available(), "needs /root/reference"
    lib = _lib.bind(ctypes.CDLL(str(build_emu.build())))
    _lib.get_lib = lambda: lib                      # the emulator backend, as tests/conftest.py installs it
    _lib.stream_handle = lambda: 0
    _lib.require_device = lambda t, name="tensor": None
    out = dict(cameras=cameras(), volume_integration=volume_integration(), lidar=lidar(), configs=configs())

    def own(v):
        if torch.is_tensor(v):
            return v.detach().clone().contiguous()
        if isinstance(v, dict):
            return {k: own(x) for k, x in v.items()}
        if isinstance(v, (list, tuple)):
            return type(v)(own(x) for x in v)
        return v
    torch.save(own(out), HERE / "reference_fixture.pt")
    print("wrote", HERE / "reference_fixture.pt", (HERE / "reference_fixture.pt").stat().st_size, "bytes")
    print("lidar losses:", out["lidar"]["losses"])


if __name__ == "__main__":
    main()
