"""The reference's own YAML ``model_params`` blocks (code_single/configs/...) constructing this package's models the
way the reference does: ``import_str(model_class)(**model_params, device=device)`` on a class derived from the
nr3d_lib model, then ``populate`` / ``training_initialize`` / ``training_before_per_step``
(app/resources/asset_bank.py:129-138, 291-298; app/models/single/neus.py:30-64, 125-236).

The YAML files are read from /root/reference (authoring container only); the GPU box runs the excerpt-free tests."""
from pathlib import Path

import pytest
import torch

CFG = Path("/root/reference/code_single/configs")
needs_reference = pytest.mark.skipif(not CFG.exists(), reason="executes the reference's own sources from /root/reference (authoring container only; emulator backend). What it pins is replayed on the GPU box from frozen reference outputs: tests/test_reference_frozen.py, test_reference_glue.py::test_*_fixture")


def _load(rel):
    from nr3d_lib.config import load_config
    return load_config(str(CFG / rel))


@needs_reference
def test_config_interpolation_on_the_reference_yaml():
    c = _load("object_centric/lotd_neus.dtu.230814.yaml")
    m = c.assetbank_cfg.Main.model_params
    assert m.var_ctrl_cfg.stop_it == c.training.num_iters == 7500 and m.var_ctrl_cfg.final_inv_s == 2000.0
    assert m.accel_cfg.init_cfg.num_pts == 2 ** 20                       # ${eval:"2**20"}
    assert m.ray_query_cfg.query_param.num_fine == [8, 8, 32] and m.ray_query_cfg.query_param.march_cfg.step_size == 0.005
    assert c.assetbank_cfg.Distant.model_params.encoding_cfg.lotd_auto_compute_cfg.target_num_params == 8 * 2 ** 20
    assert c.assetbank_cfg.Distant.model_params.ray_query_cfg.query_param.march_cfg.max_steps == 64


@needs_reference
@pytest.mark.parametrize("rel,inside_out,final_inv_s", [("object_centric/lotd_neus.dtu.230814.yaml", False, 2000.0),
                                                        ("indoor/lotd_neus.replica.230814.yaml", True, 1200.0)])
def test_object_centric_block_builds_the_model(backend, rel, inside_out, final_inv_s):
    """``class LoTDNeuSObj(AssetMixin, LoTDNeuSModel)`` + ``cls(**model_params, device=device)`` with the YAML block
    verbatim (only the table is shrunk so that the emulator run stays small)."""
    from nr3d_lib.models.fields.neus import LoTDNeuSModel

    class LoTDNeuSObj(LoTDNeuSModel):              # what app/models/single/neus.py:30 derives
        assigned_to, is_ray_query_supported = "OBJECT", True

        def asset_populate(self, scene=None, obj=None, config=None, device=None, **kw):
            super().populate(device=device)

        def asset_training_initialize(self, scene, obj, config, logger=None, log_prefix=None):
            return super().training_initialize(config, logger=logger, log_prefix=log_prefix)

    c = _load(rel)
    mp = c.assetbank_cfg.Main.model_params
    full = LoTDNeuSObj(**mp.to_dict(), device=None)                       # the real 16-level / 2^19 block (host only)
    cfg = full.encoding.cfg
    assert cfg.lod_res == [16, 23, 31, 43, 59, 81, 112, 154, 213, 295, 407, 562, 777, 1073, 1483, 2048]  # yaml :97
    assert cfg.lod_types == ["Dense"] * 5 + ["Hash"] * 11 and cfg.hashmap_size == 2 ** 19 and cfg.n_params == 12196216
    assert full.sdf_D == 1 and full.inside_out == inside_out and full.field_meta.precision == 0          # dtype: half
    assert full.ray_query_cfg["query_mode"] == "march_occ_multi_upsample_compressed"
    assert full.accel.resolution == [64, 64, 64] and full.accel.n_steps_warmup == 256
    assert full._var_ctrl == dict(start_it=2000, stop_it=7500, final_inv_s=final_inv_s)
    # a small pyramid of the same block on the test backend: life cycle of the reference's trainer
    small = mp.to_dict()
    small["surface_cfg"]["encoding_cfg"]["lotd_auto_compute_cfg"].update(num_levels=8, log2_hashmap_size=12, max_res=64)
    small["accel_cfg"].update(resolution=[16, 16, 16], init_cfg=dict(num_steps=2, num_pts=2 ** 12),
                              update_from_net_cfg=dict(num_steps=2, num_pts=2 ** 12), n_steps_warmup=2,
                              n_steps_between_update=2)
    m = LoTDNeuSObj(**small, device=backend)
    m.asset_populate(device=backend)
    assert m.asset_training_initialize(None, None, c.assetbank_cfg.Main.asset_params.initialize_cfg) is True
    assert 0.0 < m.accel.frac_occupied() < 0.8                            # the radius_init sphere is in the grid
    x = torch.tensor([[0.0, 0.0, 0.0], [0.9, 0.0, 0.0]], device=backend)
    s = m.query_sdf(x).cpu()
    assert (s[0] > 0 > s[1]) if inside_out else (s[0] < 0 < s[1])         # radius_init 0.5, sign by inside_out
    m.training_before_per_step(0)
    assert m.encoding.cfg.meta.n_active_levels == 3                       # hardmask: start_level 2 -> levels 0..2
    m.training_before_per_step(1000)
    assert m.encoding.cfg.meta.n_active_levels == 0                       # stop_it 1000: all levels
    m.training_before_per_step(4750)
    assert abs(m._ctrl_mix - 0.5) < 1e-6                                  # var_ctrl: half way from 2000 to 7500
    m.training_after_per_step(4750)


@needs_reference
def test_street_block_is_built_at_populate(backend):
    """``LoTDNeuSStreet``: cuboid pyramid and ``vox_size`` occupancy grid are sized from the AABB that
    ``asset_populate`` computes from the camera frusta and hands to ``populate(aabb=...)`` (neus.py:152-196)."""
    from nr3d_lib.models.fields.neus import LoTDNeuSModel
    c = _load("waymo/streetsurf/withmask_withlidar_joint.240219.yaml")
    mp = c.assetbank_cfg.Street.model_params.to_dict()
    m = LoTDNeuSModel(**mp, device=None)
    with pytest.raises(AssertionError):
        m.populate(device=None)                                           # no AABB yet
    aabb = torch.tensor([[-60.0, -20.0, -4.0], [60.0, 20.0, 12.0]])
    m.populate(aabb=aabb)
    cfg = m.encoding.cfg
    assert cfg.hashmap_size == 2 ** 20 and 30 * 2 ** 20 < cfg.n_params < 36 * 2 ** 20                      # "32 Mi params"
    assert cfg.lod_res3[0][0] > cfg.lod_res3[0][1] > cfg.lod_res3[0][2] == 16                              # per-axis
    assert m.accel.resolution == [120, 40, 16] and m.sdf_scale == 25.0 and m.sdf_D == 1                    # vox_size 1.0
    assert m.ray_query_cfg["query_param"]["num_coarse"] == 128
    assert m.ray_query_cfg["query_param"]["upsample_use_estimate_alpha"] is False
    # Distant (street variant) and Sky blocks
    from neuralsim_amd.env import SimpleSky
    from neuralsim_amd.fields.nerf_distant import LoTDNeRFDistantModel
    dp = c.assetbank_cfg.Distant.model_params.to_dict()
    dp["encoding_cfg"]["lotd_auto_compute_cfg"].update(target_num_params=2 ** 14, log2_hashmap_size=10, min_res_xyz=3,
                                                       min_res_w=2)
    d = LoTDNeRFDistantModel(**dp, device=backend).populate(aabb=aabb, device=backend)
    assert d.include_inf is False and d.use_view_dirs is False and d.K == c.distant_nsample
    assert torch.equal(d.aabb.cpu(), aabb)
    assert d.cfg.cuboid and d.cfg.res3[0] == [23, 8, 3]                   # lotd_use_cuboid: per-axis 4-D pyramid (120:40:16)
    sky = SimpleSky(**c.assetbank_cfg.Sky.model_params.to_dict(), device=backend)
    assert sky.n_frequencies == 10 and sky.n_appear == 4


@needs_reference
def test_distant_block_object_centric():
    from neuralsim_amd.fields.nerf_distant import LoTDNeRFDistantModel
    c = _load("object_centric/lotd_neus.dtu.230814.yaml")
    d = LoTDNeRFDistantModel(**c.assetbank_cfg.Distant.model_params.to_dict())
    assert d.include_inf and d.use_view_dirs and d.K == 64 and (d.r_min, d.r_max) == (1.0, 1000.0)
    assert d.cfg.n_params >= 8 * 2 ** 20 and d.cfg.num_levels == 12


def test_unsupported_options_fail_loudly():
    from neuralsim_amd.fields.neus import LoTDNeuSModel
    base = dict(dtype="half", surface_cfg=dict(encoding_cfg=dict(lotd_cfg=dict(lod_res=[4, 8], lod_n_feats=[2, 2],
                                                                               hashmap_size=1024)),
                                               decoder_cfg=dict(type="mlp", D=1, W=64)))
    LoTDNeuSModel(**base)
    for bad, key in ((dict(use_tcnn_backend=True), "use_tcnn_backend"), (dict(dtype="bfloat16"), "dtype"),
                     (dict(cos_anneal_cfg=dict(stop_it=10)), "cos_anneal_cfg")):
        with pytest.raises(NotImplementedError, match=key):
            LoTDNeuSModel(**dict(base, **bad))
    wide = dict(base, surface_cfg=dict(base["surface_cfg"], decoder_cfg=dict(type="mlp", D=1, W=128)))
    with pytest.raises(NotImplementedError, match="decoder_cfg.W"):
        LoTDNeuSModel(**wide)
    with pytest.raises(TypeError, match="unexpected"):
        LoTDNeuSModel(**dict(base, not_a_key=1))


def test_street_pretrain_targets(backend):
    """``pretrain_sdf_road_surface`` / ``pretrain_sdf_capsule`` (shim of nr3d_lib.models.fields.sdf) through the
    reference's ``LoTDNeuSStreet.asset_training_initialize`` flow (app/models/single/neus.py:198-236): the table
    starts as the road surface ``ego_height`` below the track, ``is_pretrained`` flips, the occupancy grid is built."""
    from nr3d_lib.models.fields.neus import LoTDNeuSModel
    from nr3d_lib.models.fields.sdf import pretrain_sdf_capsule, pretrain_sdf_road_surface
    aabb = torch.tensor([[-8.0, -4.0, -2.0], [8.0, 4.0, 2.0]])
    m = LoTDNeuSModel(lod_res=[[9, 5, 3], [17, 9, 5], [33, 17, 9]], log2_hashmap_size=13, sdf_D=1, precision="f32",
                      aabb=aabb, sdf_scale=25.0, accel_cfg=dict(resolution=(16, 8, 4), update_from_net_cfg=dict(num_steps=2, num_pts=2 ** 13)))
    m = m.to(backend)
    m.geo_init_method = "pretrain"
    tracks = torch.stack([torch.linspace(-7, 7, 29), torch.zeros(29), torch.full([29], 0.5)], dim=-1)   # camera 0.5 above z = 0 ...
    # --- what asset_training_initialize does
    assert not m.implicit_surface.is_pretrained
    pretrain_sdf_road_surface(m.implicit_surface, tracks, lr=1e-3, num_iters=1000, num_points=262144, w_eikonal=3e-3,
                              floor_dim="z", floor_up_sign=1, ego_height=2.0, logger=None, log_prefix="street")
    m.implicit_surface.is_pretrained = ~m.implicit_surface.is_pretrained
    m.accel.init(m.query_sdf)
    assert bool(m.is_pretrained)
    x = torch.tensor([[0.0, 0.0, 0.5], [3.0, 1.0, -1.5], [-5.0, -2.0, -1.9], [2.0, 0.5, 1.5]], device=backend)
    sdf = m.query_sdf(x).cpu()
    want = (x.cpu()[:, 2] - (0.5 - 2.0)) / 25.0                           # ... so the road is at z = -1.5; one SDF unit = 25 m
    assert (sdf - want).abs().max() < 0.05 / 25.0
    occ = m.accel.occ_grid.cpu()                                          # [X, Y, Z]: only the slab around z = -1.5
    assert bool(occ[:, :, 0].any()) and not bool(occ[:, :, 2:].any())
    pretrain_sdf_capsule(m.implicit_surface, tracks, surface_distance=1.0)
    s2 = m.query_sdf(torch.tensor([[0.0, 0.0, 0.5], [0.0, 3.0, 0.5]], device=backend)).cpu()
    assert s2[0] > 0.9 / 25.0 and s2[1] < -1.5 / 25.0                      # free on the track, solid 3 m beside it


@needs_reference
def test_scalar_num_fine_has_one_reading():
    """the multi-object YAMLs give ``num_fine`` as ONE number next to two ``upsample_inv_s_factors``
    (no_fg_occ.221218.yaml:378-390: 8 with [1, 4]; all_occ.240201.yaml:481-484: 16 with [1, 4]).  ``fields.neus.fine_list`` reads
    it as the total, dealt evenly to the stages -- and the BASELINE configs[4] workload (``scenarios.vehicle_model``) passes the
    YAML's literal through the same function instead of holding a second reading ([8, 8] in rounds 3-4)."""
    import yaml
    from neuralsim_amd.fields.neus import fine_list
    for path, want_nf, want in (("code_multi/configs/exps/fg_neus=hyper_lotd/no_fg_occ.221218.yaml", 8, [4, 4]),
                                ("code_multi/configs/exps/fg_neus=permuto/all_occ.240201.yaml", 16, [8, 8])):
        text = (CFG.parent.parent / path).read_text()
        tree = yaml.safe_load(text)
        qp = tree["assetbank_cfg"]["Vehicle"]["model_params"]["ray_query_cfg"]["query_param"]
        assert qp["num_fine"] == want_nf and list(qp["upsample_inv_s_factors"]) == [1, 4]
        assert fine_list(qp) == want
    assert fine_list(dict(num_fine=[8, 8, 32], upsample_inv_s_factors=[1, 4, 16])) == [8, 8, 32]      # a list names the stages
    import inspect
    from neuralsim_amd import scenarios
    src = inspect.getsource(scenarios.vehicle_model)
    assert "num_coarse=32, num_fine=8," in src          # the yaml's literal, read by fine_list inside the model
