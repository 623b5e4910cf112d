"""NeuS field on the permutohedral-lattice encoding -- the ``PermutoNeuSObj`` of
app/models/single/neus.py:64-76 and, with a per-ray condition concatenated to the position, the
``GenerativePermutoConcat`` family of code_multi/configs/exps/fg_neus=permuto/all_occ.240201.yaml:425-461
(``AD_GenerativePermutoConcatNeuSObj``: latent ``z_ins`` of 4 dims + position -> a 7-D lattice).

Everything but the encoding is ``LoTDNeuSModel``: the decoder kernels of csrc/field.hip work on level-major planes of
features and d features / d x, and csrc/permuto.hip fills those planes (``nsim_permuto_gather``) and turns the backward's
hand-off planes into table gradients (``nsim_permuto_scatter``), second-order term included.  So the sampling pass, the
occupancy grid, ray_query, the losses and the renderers run unchanged on this model; only the four ``_enc_*`` hooks differ.
"""
from typing import Optional

import torch
import torch.nn as nn

from .. import _lib
from ..grid_encodings.lotd import LoTDConfig
from ..grid_encodings.permuto import PermutoConfig
from .neus import LoTDNeuSModel


class _PermutoFieldCfg:
    """What ``LoTDNeuSModel`` reads of an encoding config: the level count, a LoTD meta for the decoder launches (they
    take the number of levels from it -- a stub pyramid of that many 2^3 dense levels) and the AABB normalisation."""

    def __init__(self, pcfg: PermutoConfig):
        self.permuto = pcfg
        self.num_levels, self.out_features, self.n_params = pcfg.num_levels, pcfg.out_features, pcfg.n_params
        self._stub = LoTDConfig([2] * pcfg.num_levels, 2, 4)
        self.meta = self._stub.meta
        self.pmeta = pcfg.meta
        self.lod_res = [2] * pcfg.num_levels
        self.hashmap_size = pcfg.hashmap_size
        self.aabb = None

    def set_aabb(self, aabb):
        """The lattice sees u = (x - centre) / half-extent in [-1,1]^3 (the reference normalises positions with the model's
        AABB space before the encoding): folded into the per-level scale / shift of the spatial inputs."""
        a = torch.as_tensor(aabb, dtype=torch.float64).reshape(2, 3)
        c, h = ((a[0] + a[1]) * 0.5).tolist(), ((a[1] - a[0]) * 0.5).tolist()
        p = self.permuto
        import math
        for l in range(p.num_levels):
            for i in range(3):
                s = p.res[l] / math.sqrt((i + 1) * (i + 2))
                self.pmeta.scale[l][i] = s / h[i]
                self.pmeta.shift[l][i] = float(p.shifts[l, i]) * h[i] - c[i]
        self._stub.set_aabb(aabb)
        self.aabb = a.float()

    def set_active_levels(self, n):
        if n is not None and 0 < int(n) < self.num_levels:
            raise NotImplementedError("hardmask level annealing is a LoTD feature; the permutohedral configs do not use it")


class _PermutoFieldEncoding(nn.Module):
    def __init__(self, pcfg: PermutoConfig, bound: float, seed: int):
        super().__init__()
        self.cfg = _PermutoFieldCfg(pcfg)
        g = torch.Generator().manual_seed(seed)
        p = ((torch.rand(pcfg.n_params, generator=g) * 2 - 1) * bound).half().float()
        self.flattened_params = nn.Parameter(p)
        self.register_buffer("params16", p.half(), persistent=False)
        self._shadow_version = self.flattened_params._version

    def shadow(self) -> torch.Tensor:
        p = self.flattened_params
        if self._shadow_version != p._version or self.params16.device != p.device:
            self.params16 = p.detach().half()
            self._shadow_version = p._version
        return self.params16


class _ZTapFn(torch.autograd.Function):
    """Delivers d L / d z to autograd.  The condition is not an input of ``_FieldFn`` (it rides along ``ridx`` inside the
    kernels); instead the TABLE handed to ``_FieldFn`` passes through this node together with z.  In the backward,
    ``_FieldFn.backward`` runs first (it is downstream), its ``_enc_scatter`` hook leaves d L / d z of its samples in
    ``model._dz_acc``; this node then runs, forwards the table gradient untouched and hands out what has accumulated for
    its z (any split of the sum over several queries of one condition is a valid split of z.grad)."""

    @staticmethod
    def forward(ctx, table, z, model):
        ctx.model, ctx.key, ctx.z_shape = model, id(z), z.shape
        return table.view_as(table)

    @staticmethod
    def backward(ctx, g_table):
        ent = ctx.model._dz_acc.pop(ctx.key, None)
        dz = ent[1] if ent is not None else None
        if dz is not None:
            zs = ctx.z_shape
            dz = dz.sum(0, keepdim=True).reshape(zs) if (len(zs) == 1 or zs[0] == 1) and dz.shape[0] != 1 else dz.reshape(zs)
        return g_table, dz, None


class PermutoNeuSModel(LoTDNeuSModel):
    """``PermutoNeuSObj(surface_cfg{encoding_cfg{permuto_auto_compute_cfg{...}}, decoder_cfg{D, W: 64}}, radiance_cfg, ...)``.
    ``z_dim`` > 0: ``set_condition(z)`` with z [R_or_1, z_dim] makes every ray (or all of them) carry a latent that is
    concatenated to the position (GenerativePermutoConcat); ``z`` rides along ``ridx`` inside the kernels."""
    planes_always = True

    def __init__(self, permuto_auto_compute_cfg: Optional[dict] = None, z_dim: int = 0, param_bound: float = 1e-4,
                 seed: int = 42, device=None, **kw):
        post = {}
        if "surface_cfg" in kw:
            # the reference's ``model_params`` block verbatim (app/resources/asset_bank.py:129-138): the encoding block is
            # ``surface_cfg.encoding_cfg.permuto_auto_compute_cfg``; everything else maps as for the LoTD model
            from . import ref_config
            sc = dict(kw["surface_cfg"])
            enc = dict(sc.get("encoding_cfg") or {})
            permuto_auto_compute_cfg = dict(enc.pop("permuto_auto_compute_cfg"))
            L_ref = int(permuto_auto_compute_cfg.get("n_levels", 16))
            enc["lotd_cfg"] = dict(lod_res=[2] * L_ref, lod_n_feats=2, hashmap_size=16)
            sc["encoding_cfg"] = enc
            lat = kw.pop("latents_cfg", None)
            if lat:      # GenerativePermutoConcat: the instance latent's width (all_occ.240201.yaml:428-431)
                z_dim = sum(int(v.get("dim", 0)) for v in lat.values())
            params = dict(kw, surface_cfg=sc)
            params.setdefault("accel_cfg", None)
            params.setdefault("ray_query_cfg", None)
            kw, post = ref_config.neus_native_kwargs(params, aabb=params.pop("aabb", None) if "aabb" in params else None)
            param_bound = kw.pop("param_bound", param_bound)
        c = dict(permuto_auto_compute_cfg or {})
        c.setdefault("n_levels", 16)
        c.setdefault("log2_hashmap_size", 19)
        pcfg = PermutoConfig(in_dim=3 + int(z_dim), **c)
        kw.pop("lod_res", None)
        kw.pop("log2_hashmap_size", None)
        super().__init__(lod_res=[2] * pcfg.num_levels, log2_hashmap_size=4, param_bound=param_bound, seed=seed, device=None,
                         **kw)
        if "var_ctrl" in post:
            self.set_var_ctrl(**post["var_ctrl"])
        self._reference_post = post
        self.z_dim = int(z_dim)
        self.encoding = _PermutoFieldEncoding(pcfg, param_bound, seed)
        self.encoding.cfg.set_aabb(self.accel.aabb.detach().cpu())
        self.field_meta.lotd = self.encoding.cfg.meta
        self._sdf_fused = False
        self.geo_init_method = "pretrain"
        self._z_rays = None
        self._z_src = None          # the caller's z when it requires grad (learned codes): see _ZTapFn
        self._dz_acc = {}
        if device is not None:
            self.to(device)

    # ---------------------------------------------------------------- condition (GenerativePermutoConcat)
    def set_condition(self, z: Optional[torch.Tensor]):
        """z [R, z_dim] per ray of the next queries, [1, z_dim] / [z_dim] for all of them, None = zeros.
        A z that requires grad (the auto-decoder's learned codes, ``z_ins_all`` of AD_GenerativePermutoConcatNeuSObj)
        receives d L / d z from every with-grad query made under this condition (``nsim_permuto_dz``).  The backward of a
        query uses the condition it was MADE under (the ``enc_state`` its forward hook returned), whatever has been set since."""
        self._z_src = None
        if z is not None:
            assert self.z_dim > 0 and z.shape[-1] == self.z_dim
            if z.requires_grad:
                self._z_src = z
            z = z.detach().float().reshape(-1, self.z_dim).contiguous()
        self._z_rays = z

    def clean_condition(self):
        self.set_condition(None)

    def _per_ray_condition(self) -> bool:
        return self.z_dim > 0 and self._z_rays is not None and int(self._z_rays.shape[0]) != 1

    def _table(self):
        p = self.encoding.flattened_params
        if self._z_src is not None and torch.is_grad_enabled():
            return _ZTapFn.apply(p, self._z_src, self)
        return p

    def _z_for(self, ridx, rays_o, S: int, dev):
        """-> (z [R, z_dim] or None, ridx or a zero index for the point mode with a shared condition)"""
        if self.z_dim == 0 or self._z_rays is None:
            return None, ridx
        z = self._z_rays.to(dev)
        if z.shape[0] == 1:
            R = rays_o.shape[0] if rays_o is not None else 1
            z = z.expand(R, self.z_dim).contiguous()
            if ridx is None:
                ridx = torch.zeros([S], dtype=torch.long, device=dev)
        return z, ridx

    # ---------------------------------------------------------------- encoding hooks (csrc/permuto.hip)
    def _enc_field_fwd(self, grid16, wpack, x, rays_o, rays_d, t, ridx, goff, ha, S, sdf, nablas, rgb, h_pl, J_pl, n_dev,
                       n_add):
        assert goff is None and h_pl is not None
        z, zr = self._z_for(ridx, rays_o, S, sdf.device)
        _lib.call("nsim_permuto_gather", self.encoding.cfg.pmeta, _lib.ptr(grid16), _lib.ptr(x), _lib.ptr(rays_o),
                  _lib.ptr(rays_d), _lib.ptr(t), _lib.ptr(zr), _lib.ptr(z), S, _lib.ptr(n_dev), int(n_add), None,
                  int(J_pl.dtype != torch.float16),      # (with-grad outputs: the flag selects the dh/dx plane type, f16 | f32)
                  _lib.ptr(h_pl), _lib.ptr(J_pl))
        # grid = NULL: the planes are filled -- decoders only
        _lib.call("nsim_field_fwd", self.field_meta, None, _lib.ptr(wpack), _lib.ptr(x), _lib.ptr(rays_o),
                  _lib.ptr(rays_d), _lib.ptr(t), _lib.ptr(ridx), None, _lib.ptr(ha), S, _lib.ptr(sdf),
                  _lib.ptr(nablas), _lib.ptr(rgb), _lib.ptr(h_pl), _lib.ptr(J_pl), _lib.ptr(n_dev), int(n_add))
        # the condition this query was made under travels with the query (``_FieldFn`` keeps it on its ctx and hands it to
        # ``_enc_scatter``): the backward kernels re-read z as they re-read the rays, and the condition may have changed by then
        # (a batched query sets per-pair codes, the next query others).  Rounds 3-4 looked it up in a 16-entry FIFO keyed by the
        # data pointer of a sample array (silent fall-back to the CURRENT condition on a miss)
        return (z, zr, self._z_src) if z is not None else None

    def _enc_gather_feat(self, fm, grid16, x, rays_o, rays_d, t, ridx, goff, S, n_dev, n_add, planes):
        assert goff is None
        z, zr = self._z_for(ridx, rays_o, S, planes.device)
        _lib.call("nsim_permuto_gather", self.encoding.cfg.pmeta, _lib.ptr(grid16), _lib.ptr(x), _lib.ptr(rays_o),
                  _lib.ptr(rays_d), _lib.ptr(t), _lib.ptr(zr), _lib.ptr(z), S, _lib.ptr(n_dev), int(n_add), _lib.ptr(planes),
                  int(fm.precision != 0), None, None)

    def _enc_scatter(self, x, rays_o, rays_d, t, ridx, goff, S, dh_pl, g_pl, gn_total, dgrid, enc_state=None):
        if enc_state is not None:
            z, zr, z_src = enc_state
        else:       # an unconditioned model, or a caller that runs forward and backward under ONE condition and says so by omission
            assert self.z_dim == 0 or self._z_src is None, \
                "a query made under a LEARNED condition must hand its enc_state (z, zr, z_src) to the backward"
            z_src = None
            z, zr = self._z_for(ridx, rays_o, S, dgrid.device)
        _lib.call("nsim_permuto_scatter", self.encoding.cfg.pmeta, _lib.ptr(x), _lib.ptr(rays_o), _lib.ptr(rays_d),
                  _lib.ptr(t), _lib.ptr(zr), _lib.ptr(z), S, _lib.ptr(dh_pl), _lib.ptr(g_pl), _lib.ptr(gn_total),
                  _lib.ptr(dgrid))
        if z is not None and z_src is not None:         # learned condition: d L / d z of these samples
            dz = torch.zeros_like(z)
            _lib.call("nsim_permuto_dz", self.encoding.cfg.pmeta, _lib.ptr(self.encoding.shadow()), _lib.ptr(x), _lib.ptr(rays_o),
                      _lib.ptr(rays_d), _lib.ptr(t), _lib.ptr(zr), _lib.ptr(z), S, _lib.ptr(dh_pl), _lib.ptr(dz))
            k = id(z_src)
            prev = self._dz_acc.get(k)
            prev = prev[1] if prev is not None else None
            if prev is not None and prev.shape != dz.shape:          # a shared [1, z_dim] condition seen through two ray counts
                prev, dz = prev.sum(0, keepdim=True), dz.sum(0, keepdim=True)
            # (the entry holds z_src itself: an id() is only unique among LIVE objects)
            self._dz_acc[k] = (z_src, dz if prev is None else prev + dz)

    def _enc_hess_dx(self, grid16, x, rays_o, rays_d, t, ridx, goff, S, g_pl, gn_total, dx):
        return      # barycentric weights are piecewise LINEAR in x: no second derivative inside a simplex

    # ---------------------------------------------------------------- geometric initialisation = pre-training
    def geometric_init_sphere(self, radius: float = 0.5, num_iters: int = 300, lr: float = 2e-3, num_pts: int = 2 ** 14,
                              seed: int = 0, w_eikonal: float = 0.1, **_):
        """``geo_init_method: pretrain`` (all_occ.240201.yaml:451): there is no dense level to write a sphere into -- the
        SDF is fitted to |u| - radius by ``LoTDNeuSModel.pretrain_sdf_sphere`` (Adam through the model's own kernels)."""
        return self.pretrain_sdf_sphere(radius, num_iters=num_iters, lr=lr, num_pts=num_pts, seed=seed, w_eikonal=w_eikonal)

    def training_initialize(self, config=None, logger=None, log_prefix=None) -> bool:
        """``asset_training_initialize`` -> ``training_initialize`` (app/models/single/neus.py:92-95): the pre-training
        loop of ``geo_init_method: pretrain`` with ``initialize_cfg{num_iters, lr}`` (defaults 300 / 2e-3), then
        ``accel.init(self.query_sdf)``."""
        post = getattr(self, "_reference_post", {})
        cfg = dict(config or {})
        updated = False
        if not bool(self.is_pretrained):
            ext = (self.accel.aabb[1] - self.accel.aabb[0]).cpu()
            r = float(post.get("radius_init", 0.5)) / (float(ext.min()) / 2.0)
            with torch.enable_grad():
                self.geometric_init_sphere(min(r, 0.95), num_iters=int(cfg.get("num_iters", 300)), lr=float(cfg.get("lr", 2e-3)),
                                           num_pts=int(cfg.get("num_pts", 2 ** 14)), w_eikonal=float(cfg.get("w_eikonal", 0.1)))
            updated = True
        if self.accel is not None:
            with torch.no_grad():
                self.accel.init(self.query_sdf, logger=logger)
        return updated

    def geometric_init_fn(self, *a, **k):
        raise NotImplementedError("the permutohedral model is initialised by pre-training (geometric_init_sphere)")
