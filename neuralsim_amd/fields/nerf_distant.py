"""NeRF++ distant-view model -- host side (csrc/nerf_field.hip).

Mirrors ``nr3d_lib.models.fields_distant.nerf.LoTDNeRFDistantModel`` as wrapped and driven by the reference
(app/models/single/nerf.py:145-196 ``LoTDNeRFDistant``; call site app/renderers/single_volume_renderer.py:281-309):
the model is queried on ALL rays with ``near`` := the close-range object's ``far`` on the rays that hit its AABB,
``ray_query_cfg{query_mode: march, march_cfg{sample_mode: box, max_steps: 64}}``, ``radius_scale_min/max = 1/1000``,
``include_inf_distance: true`` and returns a *batched* volume buffer [N, K].
Hyper-parameters: code_single/configs/object_centric/lotd_neus.dtu.230814.yaml:186-247.  Spec: oracle/distant.py.
"""
import ctypes as C
import math
from typing import Dict

import torch
import torch.nn as nn

from .. import _lib
from ..model_base import ModelMixin


class LoTD4Config:
    """``lotd_auto_compute_cfg{type: ngp4d, target_num_params, min_res_xyz, min_res_w, n_feats 2, log2_hashmap_size,
    per_level_scale}`` (yaml :193-200)."""

    def __init__(self, target_num_params=8 * 2 ** 20, min_res_xyz=8, min_res_w=4, log2_hashmap_size=19,
                 per_level_scale=1.382, max_levels=16, aspect=None):
        """``aspect``: the AABB's (x, y, z) extents when ``lotd_use_cuboid: true``
        (withmask_withlidar_joint.240219.yaml:256) -- the shortest axis follows the ``min_res_xyz`` progression, the
        other two are stretched by their extent ratio (the 3-D pyramid's convention, grid_encodings/lotd.py); None =
        cubic levels."""
        T = 2 ** log2_hashmap_size
        self.res_xyz, self.res_w, self.types, self.sizes, self.offsets, self.res3 = [], [], [], [], [], []
        asp = [1.0, 1.0, 1.0] if aspect is None else [float(a) / min(float(b) for b in aspect) for a in aspect]
        self.cuboid = aspect is not None
        off = 0
        for l in range(max_levels):
            Rx = int(math.ceil(min_res_xyz * per_level_scale ** l - 1e-6))
            Rw = int(math.ceil(min_res_w * per_level_scale ** l - 1e-6))
            R3 = [int(math.ceil(min_res_xyz * per_level_scale ** l * a - 1e-6)) for a in asp]
            n = R3[0] * R3[1] * R3[2] * Rw
            dense = n <= T
            self.res_xyz.append(Rx)
            self.res3.append(R3)
            self.res_w.append(Rw)
            self.types.append("Dense" if dense else "Hash")
            self.sizes.append(n if dense else T)
            self.offsets.append(off)
            off += self.sizes[-1] * 2
            if off >= target_num_params:
                break
        self.n_params, self.num_levels = off, len(self.res_xyz)
        self.out_features = 2 * self.num_levels
        m = _lib.Lotd4Meta()
        m.num_levels = self.num_levels
        for l in range(self.num_levels):
            m.res_xyz[l], m.res_y[l], m.res_z[l], m.res_w[l] = self.res3[l][0], self.res3[l][1], self.res3[l][2], self.res_w[l]
            m.type[l] = 0 if self.types[l] == "Dense" else 1
            m.size[l], m.offset[l] = self.sizes[l], self.offsets[l]
        self.meta = m


class _DistantFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, model, grid, den_w, den_b, rad_w, rad_b, h_appear, u4, rays_d, valid, K, holder=None):
        """holder: a dict the renderer may fill AFTER the joint compositing with ``keep`` [S] uint8 -- the shells whose
        transmittance in the joint ray is above the renderer's threshold; the backward then treats the others as
        invalid (their visibility weight and everything behind them is negligible: an opaque foreground, or the shells
        behind a dense one) and neither runs their MLP backward nor scatters their table gradient."""
        S = u4.shape[0]
        dev = u4.device
        grid16, wpack = model._shadow()
        sigma = torch.empty([S], dtype=torch.float32, device=dev)
        rgb = torch.empty([S, 3], dtype=torch.float32, device=dev)
        need_bwd = any(ctx.needs_input_grad)
        if not model.use_view_dirs:       # the SH columns of the first radiance layer are zero: any direction will do
            rays_d = rays_d.detach()
        h_pl = torch.empty([16, S, 2], dtype=torch.float32, device=dev) if need_bwd else None
        ha = h_appear.detach().float().contiguous() if h_appear is not None else None
        _lib.call("nsim_distant_fwd", model.meta, _lib.ptr(grid16), _lib.ptr(wpack), _lib.ptr(u4), _lib.ptr(rays_d),
                  _lib.ptr(ha), S, K, _lib.ptr(sigma), _lib.ptr(rgb), _lib.ptr(h_pl))
        if _lib.TIMER is not None:
            _lib.TIMER.note_units("nsim_distant_fwd", S)
            _lib.TIMER.note_units("nsim_distant_bwd", S)
            _lib.TIMER.note_units("nsim_lotd4_scatter", S)
        ctx.model, ctx.S, ctx.K = model, S, K
        ctx.set_materialize_grads(False)       # an unused output arrives as None (not zeros): see backward
        ctx.holder = holder        # a plain dict of the caller's (no tensor of this node inside: no reference cycle)
        # save_for_backward, not a ctx attribute: sigma / rgb are OUTPUTS (output -> grad_fn -> ctx -> output would be a
        # reference cycle that only the cyclic collector frees: 80 MB per step at 8192 rays x 64 shells)
        ctx.save_for_backward(u4, rays_d, valid, ha, h_pl, sigma, rgb)
        ctx.ha_shape = h_appear.shape if h_appear is not None else None
        return sigma, rgb

    @staticmethod
    def backward(ctx, g_sigma, g_rgb):
        model = ctx.model
        u4, rays_d, valid, ha, h_pl, sigma, rgb = ctx.saved_tensors
        dev = u4.device
        _, wpack = model._shadow()
        need = ctx.needs_input_grad
        F = model.cfg.out_features
        z = lambda n: torch.zeros([n], dtype=torch.float32, device=dev)   # noqa: E731
        dden_w, dden_b, drad_w, drad_b = z(64 * F + 64), z(65), z(64 * (F + 20) + 4096 + 192), z(131)
        dha = torch.zeros(ctx.ha_shape, dtype=torch.float32, device=dev) if (ha is not None and need[6]) else None
        dgrid = z(model.cfg.n_params) if need[1] else None
        dh_pl = torch.empty([16, ctx.S, 2], dtype=torch.float32, device=dev) if dgrid is not None else None
        gs = g_sigma.float().contiguous() if g_sigma is not None else None
        gr = g_rgb.float().contiguous() if g_rgb is not None else None
        if gs is None and gr is None:
            return (None,) * 12
        keep = ctx.holder.get("keep") if ctx.holder is not None else None
        if keep is not None:
            valid = valid & keep.reshape(-1)
        _lib.call("nsim_distant_bwd", model.meta, _lib.ptr(wpack), _lib.ptr(h_pl), _lib.ptr(sigma.detach()),
                  _lib.ptr(rgb.detach()), _lib.ptr(rays_d), _lib.ptr(ha), _lib.ptr(valid), ctx.S, ctx.K, _lib.ptr(gs),
                  _lib.ptr(gr), _lib.ptr(dh_pl), _lib.ptr(dden_w), _lib.ptr(dden_b), _lib.ptr(drad_w), _lib.ptr(drad_b),
                  _lib.ptr(dha))
        if dgrid is not None:
            _lib.call("nsim_lotd4_scatter", model.cfg.meta, _lib.ptr(u4), _lib.ptr(valid), ctx.S, _lib.ptr(dh_pl),
                      _lib.ptr(dgrid))
        if gr is None:      # the colour output has no consumer (a lidar render, with_rgb=False): the radiance branch was not
            # differentiated -- no gradient rather than a zero one, as autograd reports an unused sub-network
            return (None, dgrid, dden_w, dden_b, None, None, None, None, None, None, None, None)
        return (None, dgrid, dden_w, dden_b, model._contract_rad_w(drad_w), drad_b, dha, None, None, None, None, None)


class _DensityAlphaFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, sigma, t, valid, N, K, include_inf=True):
        sigma = sigma.float().contiguous()
        alpha = torch.empty_like(sigma)
        _lib.call("nsim_density_alpha_fwd", _lib.ptr(sigma), _lib.ptr(t), _lib.ptr(valid), N, K, int(include_inf),
                  _lib.ptr(alpha))
        ctx.save_for_backward(sigma, t, valid)
        ctx.N, ctx.K, ctx.inf = N, K, int(include_inf)
        return alpha

    @staticmethod
    def backward(ctx, g):
        sigma, t, valid = ctx.saved_tensors
        dsigma = torch.empty_like(sigma)
        _lib.call("nsim_density_alpha_bwd", _lib.ptr(sigma), _lib.ptr(t), _lib.ptr(valid), _lib.ptr(g.float().contiguous()),
                  ctx.N, ctx.K, ctx.inf, _lib.ptr(dsigma))
        return dsigma, None, None, None, None, None


class LoTDNeRFDistantModel(ModelMixin, nn.Module):
    is_ray_query_supported = True

    def __init__(self, aabb: torch.Tensor = None, precision: str = "fp16", radius_scale_min: float = 1.0,
                 radius_scale_max: float = 1000.0, max_steps: int = 64, include_inf_distance: bool = True,
                 use_view_dirs: bool = True, lotd_auto_compute_cfg: dict = None, param_bound: float = 1e-4,
                 seed: int = 7, device=None, ray_query_cfg: dict = None, lotd_use_cuboid: bool = False,
                 **reference_params):
        """``include_inf_distance`` / ``radiance_decoder_cfg.use_view_dirs``: true / true in the object-centric configs
        (lotd_neus.dtu.230814.yaml:221-236), false / false in the street config, which has a sky model and feeds the
        radiance net features + appearance only (withmask_withlidar_joint.240219.yaml:281-294).  The street config's
        ``sample_mode: fixed_cuboid_shells`` + ``interval_type: inverse_proportional`` is what the shells kernel does:
        cuboid shells = the AABB scaled about its centre, uniform in 1/r."""
        if reference_params:        # the reference's model_params block verbatim (fields/ref_config.py)
            from . import ref_config
            kw = ref_config.distant_native_kwargs(dict(
                reference_params, include_inf_distance=include_inf_distance, radius_scale_min=radius_scale_min,
                radius_scale_max=radius_scale_max, ray_query_cfg=ray_query_cfg))
            kw.setdefault("max_steps", max_steps)
            LoTDNeRFDistantModel.__init__(self, aabb=aabb, seed=seed, device=device, **kw)
            return
        super().__init__()
        self._ctor = dict(precision=precision, radius_scale_min=radius_scale_min, radius_scale_max=radius_scale_max,
                          max_steps=max_steps, include_inf_distance=include_inf_distance, use_view_dirs=use_view_dirs,
                          lotd_auto_compute_cfg=lotd_auto_compute_cfg, param_bound=param_bound, seed=seed,
                          ray_query_cfg=ray_query_cfg, lotd_use_cuboid=lotd_use_cuboid)
        self.lotd_use_cuboid = bool(lotd_use_cuboid)
        # ``include_inf_distance: None`` = decided at populate time from the scene (no Sky node -> the last shell reaches
        # infinity; app/models/single/nerf.py:180-182 assigns the attribute)
        self.include_inf_distance = include_inf_distance
        self.use_view_dirs = bool(use_view_dirs)
        c = dict(lotd_auto_compute_cfg or {})
        aspect = None
        if self.lotd_use_cuboid and aabb is not None:
            ext = (torch.as_tensor(aabb, dtype=torch.float32).reshape(2, 3)[1]
                   - torch.as_tensor(aabb, dtype=torch.float32).reshape(2, 3)[0]).tolist()
            if max(ext) / min(ext) > 1.0 + 1e-6:
                aspect = ext
        self.cfg = LoTD4Config(c.get("target_num_params", 8 * 2 ** 20), c.get("min_res_xyz", 8), c.get("min_res_w", 4),
                               c.get("log2_hashmap_size", 19), c.get("per_level_scale", 1.382), aspect=aspect)
        F = self.cfg.out_features
        g = torch.Generator().manual_seed(seed)
        p = ((torch.rand(self.cfg.n_params, generator=g) * 2 - 1) * param_bound).half().float()
        self.flattened_params = nn.Parameter(p)
        self.register_buffer("params16", p.half(), persistent=False)
        self._shadow_version = self.flattened_params._version

        def lin(o, i):
            b = 1.0 / math.sqrt(i)
            return (torch.rand(o, i, generator=g) * 2 - 1) * b, (torch.rand(o, generator=g) * 2 - 1) * b
        dw1, db1 = lin(64, F)
        dw2, db2 = lin(1, 64)
        rw1, rb1 = lin(64, F + (20 if self.use_view_dirs else 4))
        rw2, rb2 = lin(64, 64)
        rw3, rb3 = lin(3, 64)
        self.den_w = nn.Parameter(torch.cat([dw1.reshape(-1), dw2.reshape(-1)]))
        self.den_b = nn.Parameter(torch.cat([db1, db2]))
        self.rad_w = nn.Parameter(torch.cat([rw1.reshape(-1), rw2.reshape(-1), rw3.reshape(-1)]))
        self.rad_b = nn.Parameter(torch.cat([rb1, rb2, rb3]))
        if aabb is None:
            aabb = torch.tensor([[-1.0, -1, -1], [1.0, 1, 1]])
        self.register_buffer("aabb", aabb.float().reshape(2, 3).clone())
        self.r_min, self.r_max, self.K = float(radius_scale_min), float(radius_scale_max), int(max_steps)
        m = _lib.DistantMeta()
        m.lotd = self.cfg.meta
        m.precision = {"fp16": 0, "f32": 1}[precision]
        self.meta = m
        self._wpack, self._wpack_versions = None, None
        self.ray_query_cfg = dict(query_mode="march", query_param=dict(march_cfg=dict(sample_mode="box", max_steps=self.K)))
        if device is not None:
            self.to(device)

    def populate(self, aabb: torch.Tensor = None, device=None, **unused):
        """The reference hands the close-range object's AABB over at populate time (``populate_cfg.cr_obj_classname``,
        lotd_neus.dtu.230814.yaml:239-241; app/models/single/nerf.py:145-196)."""
        if aabb is not None:
            a = torch.as_tensor(aabb, dtype=torch.float32).reshape(2, 3).cpu()
            ext = (a[1] - a[0]).tolist()
            cur = (self.aabb[1] - self.aabb[0]).cpu().tolist()
            same_shape = all(abs(e / min(ext) - c / min(cur)) < 1e-6 for e, c in zip(ext, cur))
            if self.lotd_use_cuboid and not same_shape:
                # per-axis pyramid: the table is sized from the AABB's aspect, which is only known now
                dev = device if device is not None else self.den_w.device
                LoTDNeRFDistantModel.__init__(self, aabb=a, **self._ctor)
                device = dev
            else:
                self.aabb.copy_(a.to(self.aabb.device))
        if device is not None:
            self.to(device)
        return self

    def training_initialize(self, config=None, logger=None, log_prefix=None) -> bool:
        return False

    @property
    def include_inf(self) -> bool:
        return True if self.include_inf_distance is None else bool(self.include_inf_distance)

    # ------------------------------------------------------------------ optimizer (model_base.ModelMixin)
    def _param_groups(self, cfg: dict):
        """``training_cfg{lr: bglr, eps, betas}`` (lotd_neus.dtu.230814.yaml:249-254)."""
        self._shadow()
        return [dict(name="encoding", params=[self.flattened_params], shadow16=lambda: self._shadow()[0]),
                dict(name="density_decoder", params=[self.den_w, self.den_b]),
                dict(name="radiance_decoder", params=[self.rad_w, self.rad_b])]

    def _after_optimizer_step(self):
        self._wpack_versions = None

    def _weight_reg_tensors(self):
        return [self.den_w, self.rad_w]

    @property
    def space(self):
        """The close-range object's box this model surrounds (``populate(aabb=cr_obj.model.space.aabb)``,
        app/models/single/nerf.py:170-177)."""
        from ..spatial import AABBSpace
        a = self.aabb
        key = (a.data_ptr(), a._version, str(a.device))       # no value comparison: that would be a device->host sync
        sp = getattr(self, "_space", None)
        if sp is None or sp[0] != key:
            from ..spatial import AABBSpace
            sp = (key, AABBSpace(aabb=a.detach().clone(), device=a.device))
            object.__setattr__(self, "_space", sp)          # a view of the model's box, not a registered sub-module
        return sp[1]

    def training_before_per_step(self, it: int, logger=None):
        pass

    def training_after_per_step(self, it: int, logger=None):
        pass

    def _shadow(self):
        p = self.flattened_params
        if self._shadow_version != p._version or self.params16.device != p.device:
            self.params16 = p.detach().half()
            self._shadow_version = p._version
        vers = (self.den_w._version, self.den_b._version, self.rad_w._version, self.rad_b._version, self.meta.precision,
                str(self.den_w.device))
        if self._wpack is None or self._wpack_versions != vers:
            nbytes = int(_lib.get_lib().nsim_distant_wpack_bytes(self.meta))
            if self._wpack is None or self._wpack.numel() != nbytes or self._wpack.device != self.den_w.device:
                self._wpack = torch.zeros([nbytes], dtype=torch.uint8, device=self.den_w.device)
            _lib.call("nsim_distant_pack_weights", self.meta, _lib.ptr(self.den_w.detach()), _lib.ptr(self.den_b.detach()),
                      _lib.ptr(self._expand_rad_w(self.rad_w.detach())), _lib.ptr(self.rad_b.detach()),
                      _lib.ptr(self._wpack))
            self._wpack_versions = vers
        return self.params16, self._wpack

    # The kernels' first radiance layer is [64 x (F + 16 SH + 4 appearance)]; without view directions the parameter is
    # [64 x (F + 4)] (the reference's shape) and the SH columns of the packed matrix are zeros.
    def _expand_rad_w(self, rad_w: torch.Tensor) -> torch.Tensor:
        if self.use_view_dirs:
            return rad_w
        F = self.cfg.out_features
        q1 = rad_w[:64 * (F + 4)].view(64, F + 4)
        full = torch.zeros([64, F + 20], dtype=rad_w.dtype, device=rad_w.device)
        full[:, :F] = q1[:, :F]
        full[:, F + 16:] = q1[:, F:]
        return torch.cat([full.reshape(-1), rad_w[64 * (F + 4):]]).contiguous()

    def _contract_rad_w(self, d_full: torch.Tensor) -> torch.Tensor:
        if self.use_view_dirs:
            return d_full
        F = self.cfg.out_features
        q1 = d_full[:64 * (F + 20)].view(64, F + 20)
        return torch.cat([torch.cat([q1[:, :F], q1[:, F + 16:]], dim=1).reshape(-1), d_full[64 * (F + 20):]])

    def ray_query(self, *, ray_input: dict = None, ray_tested: dict, config=None, return_buffer=True,
                  return_details=False, render_per_obj_individual=False) -> Dict:
        """``ray_tested`` carries ALL rays: rays_o, rays_d [N,3], near [N] (cr ``far`` on rays that hit the close-range
        AABB), optional rays_h_appear [N,4] (single_volume_renderer.py:286-300)."""
        cfg = dict(config or {})
        o = ray_tested["rays_o"].detach().float().contiguous()
        d = ray_tested["rays_d"].detach().float().contiguous()
        near = ray_tested["near"].detach().float().contiguous()
        N, K, dev = o.shape[0], self.K, o.device
        jitter = cfg.get("_jitter_dv", None)
        if jitter is None and cfg.get("perturb", False):
            jitter = torch.rand([N, K], device=dev)
        if jitter is not None:
            jitter = jitter.float().contiguous()
        t = torch.empty([N, K], dtype=torch.float32, device=dev)
        u4 = torch.empty([N * K, 4], dtype=torch.float32, device=dev)
        valid = torch.empty([N * K], dtype=torch.uint8, device=dev)
        aabb6 = (C.c_float * 6)(*[float(v) for v in self.aabb.detach().cpu().reshape(-1)])
        _lib.call("nsim_distant_shells", _lib.ptr(o), _lib.ptr(d), _lib.ptr(near), _lib.ptr(jitter), N, K, aabb6,
                  self.r_min, self.r_max, _lib.ptr(t), _lib.ptr(u4), _lib.ptr(valid))
        h_appear = ray_tested.get("rays_h_appear", None)
        holder = {}
        sigma, rgb = _DistantFn.apply(self, self.flattened_params, self.den_w, self.den_b, self.rad_w, self.rad_b,
                                      h_appear, u4, d, valid, K, holder)
        alpha = _DensityAlphaFn.apply(sigma, t.reshape(-1), valid, N, K, self.include_inf)
        # (``pack_infos_hit`` next to ``num_per_hit``: the reference's renderer reads it on its distant-only path,
        # app/renderers/single_volume_renderer.py:392-397)
        vb = dict(type="batched", rays_inds_hit=torch.arange(N, device=dev), num_per_hit=K, t=t,
                  pack_infos_hit=torch.stack([torch.arange(N, device=dev) * K, torch.full([N], K, dtype=torch.long, device=dev)], dim=-1),
                  opacity_alpha=alpha.view(N, K), rgb=rgb.view(N, K, 3), sigma=sigma.view(N, K), valid=valid.view(N, K))
        ret = dict(volume_buffer=vb, _bwd_holder=holder)
        if render_per_obj_individual:       # this model alone (single_volume_renderer.py:313-317): all rays are "hit"
            from ..graphics.nerf import ray_alpha_to_vw
            vw = ray_alpha_to_vw(alpha.view(N, K))
            msk = vw.sum(-1)
            dw = vw / (msk[..., None] + 1e-10) if cfg.get("depth_use_normalized_vw", True) else vw
            ret["rendered"] = dict(mask_volume=msk, depth_volume=(dw * t).sum(-1))
            if cfg.get("with_rgb", True):
                ret["rendered"]["rgb_volume"] = (vw[..., None] * rgb.view(N, K, 3)).sum(-2)
        if return_details:
            ret["details"] = dict(u4=u4)
        return ret
