"""Translation of the reference's ``model_params`` blocks into the constructors of this package.

The reference builds every model as ``import_str(cfg.model_class)(**cfg.model_params, device=device)``
(app/resources/asset_bank.py:129-138) where the class derives from an nr3d_lib model
(``class LoTDNeuSObj(AssetMixin, LoTDNeuSModel)``, app/models/single/neus.py:30), i.e. the nr3d_lib constructors
receive the YAML block verbatim.  ``LoTDNeuSModel`` / ``LoTDNeRFDistantModel`` accept exactly those keyword sets
(code_single/configs/object_centric/lotd_neus.dtu.230814.yaml:80-166 and :186-247, indoor/lotd_neus.replica.230814.yaml,
waymo/streetsurf/withmask_withlidar_joint.240219.yaml:146-303) next to their native ones; this module is the mapping.

Error behaviour: an option the gfx950 kernels do not cover raises ``NotImplementedError`` naming the key (never a
silent fallback); unknown keys raise ``TypeError`` like any unexpected keyword.
"""
import math
from typing import Dict, Optional, Tuple

NEUS_REFERENCE_KEYS = {"dtype", "var_ctrl_cfg", "cos_anneal_cfg", "use_tcnn_backend", "surface_cfg", "radiance_cfg",
                       "shrink_milestones"}
DISTANT_REFERENCE_KEYS = {"dtype", "encoding_cfg", "extra_pos_embed_cfg", "density_decoder_cfg", "radiance_decoder_cfg",
                          "n_extra_feat_from_output", "use_tcnn_backend", "cr_obj_classname"}


def _unsupported(key, value, why):
    raise NotImplementedError(f"{key}={value!r}: {why}")


def _precision(dtype) -> str:
    s = str(dtype).replace("torch.", "")
    if s in ("half", "float16", "fp16"):
        return "fp16"
    if s in ("float", "float32", "f32"):
        return "f32"
    _unsupported("dtype", dtype, "the field kernels compute in fp16-MFMA ('half') or f32-MFMA ('float') mode")


def _check_decoder(prefix: str, c: dict, D_allowed):
    if c.get("type", "mlp") != "mlp":
        _unsupported(f"{prefix}.type", c.get("type"), "only the fused MLP decoder is built")
    if int(c.get("W", 64)) != 64:
        _unsupported(f"{prefix}.W", c.get("W"), "the fused MFMA decoders are 64 wide")
    if int(c.get("D", min(D_allowed))) not in D_allowed:
        _unsupported(f"{prefix}.D", c.get("D"), f"hidden layers supported: {sorted(D_allowed)}")
    if c.get("select_n_levels") is not None:
        _unsupported(f"{prefix}.select_n_levels", c["select_n_levels"], "level selection is done by anneal_cfg here")


def ngp_num_levels(target_num_params: int, min_res: int, log2_hashmap_size: int, per_level_scale: float,
                   aspect=(1.0, 1.0, 1.0), max_num_levels: Optional[int] = None, n_feats: int = 2) -> int:
    """``lotd_auto_compute_cfg{type: ngp, target_num_params}``: the shortest pyramid (growth ``per_level_scale`` from
    ``min_res`` on the shortest axis) whose parameter count reaches the target.  The generator lives in the absent
    nr3d_lib; this rule is fixed here (street config: 32 Mi parameters at T = 2^20 -> 19 levels)."""
    T = 2 ** int(log2_hashmap_size)
    mn = min(aspect)
    total, L = 0, 0
    cap = int(max_num_levels) if max_num_levels else 32
    while L < cap:
        r = min_res * per_level_scale ** L
        nvert = 1
        for a in aspect:
            nvert *= int(math.ceil(r * a / mn - 1e-6))
        total += min(nvert, T) * n_feats
        L += 1
        if total >= target_num_params * (1.0 - 1.0 / 64):
            break
    return L


def lod_res_from_encoding_cfg(enc: dict, aabb=None) -> Tuple[list, int]:
    """-> (lod_res per level -- ints or per-axis triples --, log2_hashmap_size)."""
    from ..grid_encodings.lotd import cuboid_ngp_res, gen_ngp_res
    cuboid = bool(enc.get("lotd_use_cuboid", False))
    aspect = (1.0, 1.0, 1.0)
    if cuboid:
        assert aabb is not None, "lotd_use_cuboid needs the AABB (populate(aabb=...))"
        ext = [float(aabb[1][i] - aabb[0][i]) for i in range(3)]
        aspect = tuple(e / min(ext) for e in ext)
    if enc.get("lotd_cfg") is not None:
        c = enc["lotd_cfg"]
        nf = c.get("lod_n_feats", 2)
        if any(int(f) != 2 for f in (nf if isinstance(nf, (list, tuple)) else [nf])):
            _unsupported("lotd_cfg.lod_n_feats", nf, "the kernels are specialised for 2 features per level")
        hs = int(c.get("hashmap_size", 2 ** 19))
        assert hs & (hs - 1) == 0, "hashmap_size must be a power of two"
        res = list(c["lod_res"])
        log2_T = int(math.log2(hs))
        types = c.get("lod_types")
        if types is not None:       # the Dense/Hash split is implied by the table size here: refuse a different one
            for r, t in zip(res, types):
                r3 = [r] * 3 if not isinstance(r, (list, tuple)) else list(r)
                implied = "Dense" if r3[0] * r3[1] * r3[2] <= hs else "Hash"
                if t != implied:
                    _unsupported("lotd_cfg.lod_types", types, f"level res {r}: this layout stores it as {implied}")
        return res, log2_T
    c = enc["lotd_auto_compute_cfg"]
    if int(c.get("n_feats", 2)) != 2:
        _unsupported("lotd_auto_compute_cfg.n_feats", c.get("n_feats"), "2 features per level")
    log2_T = int(c.get("log2_hashmap_size", 19))
    mn = int(c.get("min_res", 16))
    s = float(c.get("per_level_scale", 1.382))
    if c["type"] == "gen_ngp":
        L = int(c["num_levels"])
        mx = c.get("max_res")
        if mx is None:      # 16 * 1.382^15 = 2048 (the list quoted in lotd_neus.dtu.230814.yaml:97)
            mx = int(round(mn * s ** (L - 1) / 32.0)) * 32
    elif c["type"] == "ngp":
        L = ngp_num_levels(int(c["target_num_params"]), mn, log2_T, s, aspect, c.get("max_num_levels"))
        mx = mn * s ** (L - 1)
    else:
        _unsupported("lotd_auto_compute_cfg.type", c["type"], "gen_ngp and ngp are built")
    return (cuboid_ngp_res(aspect, mn, mx, L) if cuboid else gen_ngp_res(mn, mx, L)), log2_T


def neus_needs_aabb(params: dict) -> bool:
    """Street-style blocks size their pyramid / occupancy grid from the AABB handed to ``populate`` later."""
    enc = params.get("surface_cfg", {}).get("encoding_cfg", {})
    acc = params.get("accel_cfg") or {}
    return bool(enc.get("lotd_use_cuboid", False)) or ("vox_size" in acc and "resolution" not in acc)


def neus_native_kwargs(params: dict, aabb=None) -> Tuple[Dict, Dict]:
    """Reference ``model_params`` of a LoTD NeuS model -> (native constructor kwargs, post-construction settings)."""
    p = dict(params)
    unknown = set(p) - NEUS_REFERENCE_KEYS - {"accel_cfg", "ray_query_cfg"}
    if unknown:
        raise TypeError(f"LoTDNeuSModel: unexpected model_params {sorted(unknown)}")
    if p.get("use_tcnn_backend", False):
        _unsupported("use_tcnn_backend", True, "there is no tiny-cuda-nn here; the decoders are HIP MFMA kernels")
    if p.get("cos_anneal_cfg") is not None:
        _unsupported("cos_anneal_cfg", p["cos_anneal_cfg"], "cosine annealing of the alpha estimate is not built")
    if p.get("shrink_milestones"):
        _unsupported("shrink_milestones", p["shrink_milestones"], "AABB shrinking is not built")
    kw: Dict = dict(precision=_precision(p.get("dtype", "half")))
    post: Dict = {}
    vc = dict(p.get("var_ctrl_cfg") or {})
    kw["ln_inv_s_init"] = float(vc.get("ln_inv_s_init", 0.1))
    kw["ln_inv_s_factor"] = float(vc.get("ln_inv_s_factor", 10.0))
    if vc.get("ctrl_type") is not None:
        if vc["ctrl_type"] != "mix_linear":
            _unsupported("var_ctrl_cfg.ctrl_type", vc["ctrl_type"], "mix_linear is built")
        post["var_ctrl"] = dict(ctrl_type="mix_linear", start_it=int(vc.get("start_it", 0)),
                                stop_it=int(vc.get("stop_it", 1)), final_inv_s=float(vc.get("final_inv_s", 2000.0)))
    sc = dict(p.get("surface_cfg") or {})
    enc = dict(sc.get("encoding_cfg") or {})
    dec = dict(sc.get("decoder_cfg") or {})
    _check_decoder("surface_cfg.decoder_cfg", dec, {1, 2})
    act = dec.get("activation") or dict(type="softplus", beta=100.0)
    act = dict(type=act) if isinstance(act, str) else dict(act)            # ``activation: relu`` (no_fg_occ.221218.yaml:357)
    if act.get("type", "softplus") == "relu":
        act = dict(type="relu", beta=-1.0)
    elif act.get("type", "softplus") != "softplus":
        _unsupported("decoder_cfg.activation.type", act.get("type"), "softplus(beta) and relu are what the kernels evaluate")
    if int(sc.get("n_extra_feat_from_output", 0)) != 0:
        _unsupported("surface_cfg.n_extra_feat_from_output", sc["n_extra_feat_from_output"],
                     "the radiance net reads position / normal / view direction / appearance only")
    kw.update(sdf_D=int(dec.get("D", 1)), W=int(dec.get("W", 64)), softplus_beta=float(act.get("beta", 100.0)),
              sdf_scale=float(sc.get("sdf_scale", 1.0)), inside_out=bool(sc.get("inside_out", False)),
              bounding_size=float(sc.get("bounding_size", 2.0)),
              param_bound=float((enc.get("param_init_cfg") or {}).get("bound", 1e-4)))
    kw["lod_res"], kw["log2_hashmap_size"] = lod_res_from_encoding_cfg(enc, aabb)
    an = enc.get("anneal_cfg")
    if an is not None:
        if an.get("type", "hardmask") != "hardmask":
            _unsupported("encoding_cfg.anneal_cfg.type", an.get("type"), "hardmask level annealing is built")
        post["anneal"] = dict(start_it=int(an.get("start_it", 0)), stop_it=int(an.get("stop_it", 1000)),
                              start_level=int(an.get("start_level", 2)))
    for k in ("clip_level_grad_ema_factor",):
        if float(sc.get(k, enc.get(k, 0)) or 0) != 0:
            _unsupported(k, sc.get(k, enc.get(k)), "per-level gradient clipping is not built")
    post["radius_init"] = float(sc.get("radius_init", 0.5))
    post["geo_init_method"] = sc.get("geo_init_method", "pretrain_after_zero_out")
    rc = dict(p.get("radiance_cfg") or {})
    _check_decoder("radiance_cfg", rc, {2})
    de = dict(rc.get("dir_embed_cfg") or dict(type="spherical", degree=4))
    if not rc.get("use_view_dirs", True) or de.get("type") != "spherical" or int(de.get("degree", 4)) != 4:
        _unsupported("radiance_cfg.dir_embed_cfg", de, "view directions enter through spherical harmonics of degree 4")
    if not rc.get("use_nablas", True):
        _unsupported("radiance_cfg.use_nablas", False, "the radiance kernel reads the sample normal")
    if not rc.get("use_pos", True):
        _unsupported("radiance_cfg.use_pos", False, "the radiance kernel reads the sample position")
    if int(rc.get("n_appear_embedding", 4)) != 4:
        _unsupported("radiance_cfg.n_appear_embedding", rc.get("n_appear_embedding"), "4 appearance channels")
    acc = dict(p.get("accel_cfg") or {})
    if acc.get("type", "occ_grid") != "occ_grid":
        _unsupported("accel_cfg.type", acc.get("type"), "occ_grid (and occ_grid_batched on the batched model)")
    if (acc.get("occ_val_fn_cfg") or {}).get("type", "sdf") != "sdf":
        _unsupported("accel_cfg.occ_val_fn_cfg.type", acc["occ_val_fn_cfg"]["type"], "sdf")
    ic = acc.get("init_cfg")
    if ic is not None and ic.get("mode", "from_net") != "from_net":
        _unsupported("accel_cfg.init_cfg.mode", ic.get("mode"), "from_net")
    if acc.get("update_from_samples_cfg"):
        _unsupported("accel_cfg.update_from_samples_cfg", acc["update_from_samples_cfg"], "{} (every sampling-pass sample is used)")
    if "vox_size" in acc and "resolution" not in acc:
        assert aabb is not None, "accel_cfg.vox_size needs the AABB (populate(aabb=...))"
        # the AABB is in object units; a street node's scale makes one unit several metres -- the caller's populate
        # hands the AABB in the units vox_size is quoted in (StreetSurf: metres before the node scale is applied)
        acc["resolution"] = [max(1, int(math.ceil(float(aabb[1][i] - aabb[0][i]) / float(acc["vox_size"]) - 1e-6)))
                             for i in range(3)]
    kw["accel_cfg"] = acc
    if p.get("ray_query_cfg") is not None:
        kw["ray_query_cfg"] = _plain(p["ray_query_cfg"])
        cs = (kw["ray_query_cfg"].get("query_param") or {}).get("coarse_step_cfg")
        if cs is not None and cs.get("step_mode", "linear") != "linear":
            _unsupported("ray_query_cfg.query_param.coarse_step_cfg.step_mode", cs.get("step_mode"), "linear")
    if aabb is not None:
        kw["aabb"] = aabb
    return kw, post


def _plain(v):
    if isinstance(v, dict):
        return {k: _plain(x) for k, x in v.items()}
    if isinstance(v, (list, tuple)):
        return [_plain(x) for x in v]
    return v


def distant_native_kwargs(params: dict) -> Dict:
    """Reference ``model_params`` of ``LoTDNeRFDistant`` (lotd_neus.dtu.230814.yaml:186-247; street variant
    withmask_withlidar_joint.240219.yaml:250-303) -> native kwargs of ``LoTDNeRFDistantModel``."""
    p = dict(params)
    if p.get("use_tcnn_backend", False):
        _unsupported("use_tcnn_backend", True, "there is no tiny-cuda-nn here")
    kw: Dict = dict(precision=_precision(p.get("dtype", "half")))
    enc = dict(p.get("encoding_cfg") or {})
    if int(enc.get("input_ch", 4)) != 4:
        _unsupported("encoding_cfg.input_ch", enc.get("input_ch"), "the distant model encodes (x, y, z, 1/r)")
    ac = dict(enc.get("lotd_auto_compute_cfg") or {})
    if ac.get("type", "ngp4d") != "ngp4d":
        _unsupported("encoding_cfg.lotd_auto_compute_cfg.type", ac.get("type"), "ngp4d")
    if int(ac.get("n_feats", 2)) != 2:
        _unsupported("lotd_auto_compute_cfg.n_feats", ac.get("n_feats"), "2 features per level")
    kw["lotd_auto_compute_cfg"] = {k: ac[k] for k in ("target_num_params", "min_res_xyz", "min_res_w", "log2_hashmap_size",
                                                      "per_level_scale") if k in ac}
    kw["param_bound"] = float((enc.get("param_init_cfg") or {}).get("bound", 1e-4))
    kw["lotd_use_cuboid"] = bool(enc.get("lotd_use_cuboid", False))
    if (p.get("extra_pos_embed_cfg") or {}).get("type", "identity") != "identity":
        _unsupported("extra_pos_embed_cfg.type", p["extra_pos_embed_cfg"]["type"], "identity")
    dd = dict(p.get("density_decoder_cfg") or {})
    _check_decoder("density_decoder_cfg", dd, {1})
    if dd.get("output_activation", "softplus") != "softplus":
        _unsupported("density_decoder_cfg.output_activation", dd.get("output_activation"), "softplus")
    rd = dict(p.get("radiance_decoder_cfg") or {})
    _check_decoder("radiance_decoder_cfg", rd, {2})
    if rd.get("use_pos", False) or rd.get("use_nablas", False):
        _unsupported("radiance_decoder_cfg.use_pos/use_nablas", True, "the distant radiance net has no position / normal input")
    kw["use_view_dirs"] = bool(rd.get("use_view_dirs", True))
    if kw["use_view_dirs"]:
        de = dict(rd.get("dir_embed_cfg") or dict(type="spherical", degree=4))
        if de.get("type") != "spherical" or int(de.get("degree", 4)) != 4:
            _unsupported("radiance_decoder_cfg.dir_embed_cfg", de, "spherical harmonics of degree 4")
    if int(rd.get("n_appear_embedding", 4)) != 4:
        _unsupported("radiance_decoder_cfg.n_appear_embedding", rd.get("n_appear_embedding"), "4 appearance channels")
    if int(p.get("n_extra_feat_from_output", 0)) != 0:
        _unsupported("n_extra_feat_from_output", p["n_extra_feat_from_output"], "0")
    for k in ("include_inf_distance", "radius_scale_min", "radius_scale_max"):
        if k in p:
            kw[k] = p[k]
    rq = _plain(p.get("ray_query_cfg") or {})
    mc = ((rq.get("query_param") or {}).get("march_cfg") or {})
    if rq.get("query_mode", "march") != "march":
        _unsupported("ray_query_cfg.query_mode", rq.get("query_mode"), "march")
    if mc.get("sample_mode", "box") not in ("box", "fixed_cuboid_shells"):
        _unsupported("march_cfg.sample_mode", mc.get("sample_mode"), "box / fixed_cuboid_shells (AABB-shaped shells)")
    if mc.get("interval_type", "inverse_proportional") != "inverse_proportional":
        _unsupported("march_cfg.interval_type", mc.get("interval_type"), "inverse_proportional (uniform in 1/r)")
    if "max_steps" in mc:
        kw["max_steps"] = int(mc["max_steps"])
    return kw
