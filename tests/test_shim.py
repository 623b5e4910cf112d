"""The reference's hot-path import lines resolve against the nr3d_lib shim and hit the HIP-backed implementation."""
import pytest
import torch


def test_reference_import_lines_resolve():
    # app/renderers/single_volume_renderer.py:19-20
    from nr3d_lib.graphics.nerf import packed_alpha_to_vw, ray_alpha_to_vw  # noqa: F401
    from nr3d_lib.graphics.pack_ops import get_pack_infos_from_n, merge_two_packs_sorted, packed_div, packed_sum  # noqa: F401
    # app/renderers/buffer_compose_renderer.py:33 ; app/renderers/utils.py:15 ; app/loss/lidar.py:17
    from nr3d_lib.graphics.pack_ops import interleave_linstep, packed_geq, packed_leq, packed_lt, packed_matmul, packed_sort  # noqa: F401
    from nr3d_lib.models.utils import batchify_query  # noqa: F401
    from nr3d_lib.graphics.pack_ops.pack_ops import packed_sum as ps2          # app/loss/eikonal.py:22
    from nr3d_lib.profile import profile                                       # single_volume_renderer.py:15
    from nr3d_lib.config import ConfigDict, parse_device_ids                   # single_volume_renderer.py:17
    assert ps2 is packed_sum and parse_device_ids("0,1") == [0, 1]
    c = ConfigDict(a=dict(b=1), **dict(with_rgb=True))
    assert c.a.b == 1 and c.get("missing", 5) == 5 and c.copy().a.b == 1
    with profile("phase"):
        pass
    assert profile(lambda x: x + 1)(1) == 2
    from nr3d_lib.models.fields.neus import LoTDNeuSModel
    from nr3d_lib.models.accelerations import OccGridAccel, OccGridEma  # noqa: F401
    from nr3d_lib.models.grid_encodings.lotd import LoTDEncoding  # noqa: F401
    from nr3d_lib.distributed import get_rank, get_world_size, init_env, is_master  # noqa: F401
    import neuralsim_amd.fields.neus as impl
    assert LoTDNeuSModel is impl.LoTDNeuSModel and get_world_size() == 1 and is_master()


def test_occupied_sampling_and_ray_conversion(backend):
    from neuralsim_amd.fields.neus import LoTDNeuSModel
    from util import SMALL_RES
    m = LoTDNeuSModel(lod_res=SMALL_RES, log2_hashmap_size=10, sdf_D=1, precision="f32",
                      accel_cfg=dict(resolution=(8, 8, 8))).to(backend)
    m.geometric_init_sphere(0.5)
    m.accel.occ_val.zero_()
    m.accel.occ_val[5 + 8 * (2 + 8 * 7)] = 1.0          # voxel (5,2,7)
    m.accel.pack_bits()
    out = m.sample_pts_in_occupied(64)
    x = out["net_x"].cpu()
    lo = torch.tensor([5, 2, 7]) / 8 * 2 - 1
    assert ((x >= lo - 1e-6) & (x <= lo + 0.25 + 1e-6)).all() and out["nablas"].shape == (64, 3)
    th = 0.3
    R = torch.tensor([[1.0, 0, 0], [0, torch.cos(torch.tensor(th)), -torch.sin(torch.tensor(th))],
                      [0, torch.sin(torch.tensor(th)), torch.cos(torch.tensor(th))]])
    t = torch.tensor([0.1, -0.2, 0.3])
    o, d = torch.randn(5, 3), torch.randn(5, 3)
    oo, dd = LoTDNeuSModel.convert_rays_in_node(o, d, R, t, 2.0)
    assert torch.allclose(oo, (o - t) @ R / 2.0, atol=1e-6) and torch.allclose(dd, d @ R / 2.0, atol=1e-6)


def test_config_eval_is_arithmetic_only():
    """``${eval:...}`` evaluates arithmetic through an AST walk -- no names, attributes or calls beyond min / max / int /
    float / round / abs -- so a YAML file cannot execute code when it is loaded."""
    from nr3d_lib.config import _safe_arith, resolve_config
    assert _safe_arith("2**20") == 1048576 and _safe_arith("8*(2**20)") == 8388608
    assert _safe_arith("int(1.5*4096)+max(1, 2)") == 6146 and _safe_arith("-3 // 2") == -2
    c = resolve_config(dict(a=4, b='${eval:"${a}*2+1"}', c="${b}"))
    assert c.b == 9 and c.c == 9
    for bad in ("__import__('os').system('true')", "().__class__", "open('x')", "a.b", "[x for x in (1,)]", "2**99999"):
        with pytest.raises((ValueError, SyntaxError)):
            _safe_arith(bad)


def test_masked_reductions_broadcast_a_per_ray_mask():
    """``fn(rgb_pred [N,3], rgb_gt [N,3], mask=remain [N], reduction='mean' | 'none')`` (app/loss/photometric.py:108-142;
    every street config with ``respect_ignore_mask``): the per-ray mask is broadcast over the channels; 'mean' averages
    ``loss * mask`` over ALL elements, 'mean_in_mask' over the masked-in ones."""
    from nr3d_lib.models.loss.recon import l1_loss, mse_loss, relative_l2_loss
    from nr3d_lib.models.loss.utils import reduce
    g = torch.Generator().manual_seed(0)
    p, t = torch.rand(8, 3, generator=g), torch.rand(8, 3, generator=g)
    m = torch.tensor([1, 0, 1, 1, 0, 0, 1, 0], dtype=torch.bool)
    e = (p - t) ** 2
    assert torch.allclose(mse_loss(p, t, mask=m, reduction="mean"), (e * m[:, None]).mean())
    assert torch.allclose(mse_loss(p, t, mask=m, reduction="mean_in_mask"), e[m].mean())
    assert mse_loss(p, t, mask=m, reduction="none").shape == (8, 3)
    assert float(mse_loss(p, t, mask=m, reduction="none")[1].abs().sum()) == 0.0
    assert torch.allclose(l1_loss(p, t, mask=m.float(), reduction="mean"), ((p - t).abs() * m[:, None]).mean())
    assert torch.allclose(relative_l2_loss(p, t), (e / (t ** 2 + 1e-2)).mean())
    x = torch.rand(5, generator=g)
    mk = torch.tensor([1.0, 1, 0, 0, 1])
    assert torch.allclose(reduce(x, mask=mk, reduction="mean"), (x * mk).mean())
    assert torch.allclose(reduce(x, mask=mk, reduction="sum"), (x * mk).sum())
    with pytest.raises(ValueError):
        reduce(x, reduction="median")


def test_lotd_model_can_run_the_reference_pretraining_procedure(backend):
    """``initialize_cfg{num_iters, lr}`` honoured as an optimisation (``geo_init_impl: pretrain`` / NSIM_GEO_INIT=pretrain)
    instead of the default deterministic write of the sphere: zero-out, Adam steps through the
    model's own kernels, then the occupancy grid from the network."""
    import torch
    from neuralsim_amd.fields.neus import LoTDNeuSModel
    m = LoTDNeuSModel(lod_res=[4, 6, 8, 12, 16, 24], log2_hashmap_size=10, sdf_D=1, precision="f32", seed=2,
                      accel_cfg=dict(resolution=(8, 8, 8), init_cfg=dict(num_steps=1, num_pts=1024),
                                     update_from_net_cfg=dict(num_steps=1, num_pts=1024), update_from_samples_cfg={})).to(backend)
    assert not bool(m.is_pretrained)
    assert m.training_initialize(dict(geo_init_impl="pretrain", num_iters=50, lr=1e-2, num_pts=1024)) is True
    assert bool(m.is_pretrained) and m.training_initialize(dict(geo_init_impl="pretrain", num_iters=80)) is False
    x = (torch.rand(512, 3, generator=torch.Generator().manual_seed(0)) * 1.6 - 0.8).to(backend)
    err = (m.query_sdf(x).cpu() - (x.cpu().norm(dim=-1) - 0.5)).abs().mean()
    assert float(err) < 0.08, float(err)
    assert 0.0 < m.accel.frac_occupied() < 1.0


def test_sorted_ckpts_puts_the_final_state_last(tmp_path):
    """``sorted_ckpts(dir)[-1]`` is what render.py:59 / extract_mesh.py:38 / the resume path load: numbered < latest < final
    (``latest.pt`` is only rewritten every ``i_save`` seconds, ``final_*.pt`` once after the last iteration, train.py:1684)."""
    import os
    from nr3d_lib.checkpoint import CheckpointIO, sorted_ckpts
    lin = torch.nn.Linear(2, 2)
    io = CheckpointIO(checkpoint_dir=str(tmp_path))
    io.register_modules(net=lin)
    for name, it in (("00000100.pt", 100), ("latest.pt", 150), ("00000200.pt", 200), ("final_00000250.pt", 250)):
        with torch.no_grad():
            lin.weight.fill_(float(it))
        io.save(filename=name, global_step=it)
    names = [os.path.basename(p) for p in sorted_ckpts(str(tmp_path))]
    assert names == ["00000100.pt", "00000200.pt", "latest.pt", "final_00000250.pt"]
    assert io.load_file(None)["global_step"] == 250 and float(lin.weight[0, 0]) == 250.0
    os.remove(tmp_path / "final_00000250.pt")           # an interrupted run: the most recent latest.pt wins
    assert io.load_file(None)["global_step"] == 150
    assert sorted_ckpts(str(tmp_path / "missing")) == []
