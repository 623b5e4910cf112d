"""Batched (multi-instance) LoTD-NeuS model -- SURVEY sec. 8 row a20.

Mirror of the API the reference's ``BufferComposeRenderer`` drives on a shared foreground model
(app/renderers/buffer_compose_renderer.py:209-265; app/models/shared/batched_neus.py:295-407):
``set_condition({'ins_ind' | 'ins_id' | 'z_ins'})`` -> ``batched_ray_test(..., compact_batch=True)`` ->
``batched_ray_query(batched_ray_tested=, batched_ray_input=, config=, ...)`` -> ``clean_condition()``, over a batched
occupancy grid (``accel_cfg{type: occ_grid_batched, resolution: [32,32,32]}``,
code_multi/configs/exps/fg_neus=hyper_lotd/no_fg_occ.221218.yaml:369-377).

What is batched here: every instance owns a LoTD table (what the reference's ``lotd_batched_growers`` emit per batch
item) and an occupancy grid; the SDF / radiance decoders are shared.  All tables live in ONE flat parameter
``[num_instances * n_params]`` and all grids in one bitfield, and every kernel on the path takes a per-ray instance
offset (``ray_goff`` / ``ray_word_off`` in include/nsim.h) -- so B instances cost the launches of one.
The hyper-network that GENERATES the tables from a latent code (StyleLoTD ``lotd_grower_cfg``, :322-352) lives in the
absent nr3d_lib; its dense part is restated in grid_encodings/lotd_growers.py: built with ``lotd_grower_cfg`` the model
grows the batch's tables from latent codes (``set_condition({'ins_id' | 'ins_ind' | 'z_ins'})``), without it the tables
are free auto-decoder parameters.
"""
from typing import Dict, List, Optional, Sequence

import torch
import torch.nn as nn

from .. import _lib
from .neus import LoTDNeuSModel, OccGridAccel


class OccGridAccelBatched(OccGridAccel):
    """``occ_grid_batched``: B grids of one resolution over the shared object-space AABB; ``occ_val [B * nvox]``,
    ``occ_bits [B * nvox / 32]`` (instance b at word offset b * nvox / 32)."""

    def __init__(self, aabb: torch.Tensor, num_batches: int, resolution=(32, 32, 32), **kw):
        super().__init__(aabb, resolution=resolution, **kw)
        self.num_batches = int(num_batches)
        self.nvox = self.resolution[0] * self.resolution[1] * self.resolution[2]
        assert self.nvox % 32 == 0, "batched occupancy grids are word-aligned per instance"
        dev = self.occ_val.device
        self.register_buffer("occ_val", torch.zeros(self.num_batches * self.nvox, dtype=torch.float32, device=dev))
        self.register_buffer("occ_bits", torch.full([self.num_batches * self.nvox // 32], -1, dtype=torch.int32,
                                                    device=dev))

    @property
    def words_per_instance(self) -> int:
        return self.nvox // 32

    @property
    def occ_grid(self) -> torch.Tensor:
        r = self.resolution
        return (self.occ_val > self.occ_thre).view(self.num_batches, r[2], r[1], r[0]).permute(0, 3, 2, 1)

    @torch.no_grad()
    def update_from_net(self, query_sdf, num_steps=None, num_pts=None, generator=None):
        """``query_sdf(pts [n,3], ins_ind: int) -> sdf [n]``; every instance is refreshed from its own table."""
        dev = self.occ_val.device
        lo, hi = self.aabb[0], self.aabb[1]
        n = num_pts or self.num_pts
        for b in range(self.num_batches):
            val = self.occ_val[b * self.nvox:(b + 1) * self.nvox]
            for _ in range(num_steps or self.num_steps):
                pts = self.draw_points(n, generator)
                sdf = query_sdf(pts, b).detach().float().contiguous()
                _lib.call("nsim_occ_decay", _lib.ptr(val), self.nvox, self.ema_decay)
                _lib.call("nsim_occ_update", _lib.ptr(val), _lib.ptr(pts), _lib.ptr(sdf), n, self.meta, self.inv_s)
        self.pack_bits()


class BatchedRaysMixin:
    """``batched_ray_test`` / ``batched_ray_query`` over (item, ray) pairs -- shared by the batched LoTD model (per-instance
    tables: table + occupancy offsets per pair) and the batched permutohedral model (one table, a latent per pair:
    fields/batched_permuto_neus.py).  The model provides ``ins_inds_per_batch`` (the condition) and
    ``_pair_extras(ins [R], which [R]) -> dict`` merged into the tested-rays dict of the single-model ``ray_query``."""

    # ------------------------------------------------------------------ batched rays (buffer_compose_renderer.py:222-265)
    def batched_ray_test(self, rays_o: torch.Tensor, rays_d: torch.Tensor, near=None, far=None, compact_batch=True,
                         **extra) -> Dict:
        """rays_o / rays_d [B', N, 3] in the object space of each batch item.  Returns the flat list of (item, ray)
        pairs that hit the AABB: ``rays_inds [R]`` (index into N), ``rays_full_bidx [R]`` (index into B'),
        ``rays_bidx [R]`` (index into the compacted list of items that were hit at all), ``full_bidx_map [B'']``
        (compact -> full), ``num_rays``, ``near``, ``far`` and the filtered inputs."""
        Bq, N = rays_o.shape[0], rays_o.shape[1]
        # RAY-major pair order (ray index ascending, the items of one ray consecutive): the renderer regroups the packs
        # of a ray that crosses several items with ``unique_consecutive`` and relies on exactly this order
        # (buffer_compose_renderer.py:347-368, "[!!!] Requires ridx to be consecutive and monotonically increasing").
        flat = super().ray_test(rays_o.transpose(0, 1).reshape(-1, 3), rays_d.transpose(0, 1).reshape(-1, 3), near=near,
                                far=far)
        pair = flat["rays_inds"]                                     # = ray * Bq + item
        rinds = torch.div(pair, Bq, rounding_mode="floor")
        full_bidx = pair - rinds * Bq
        if compact_batch:
            full_bidx_map, bidx = torch.unique(full_bidx, return_inverse=True)      # sorted: order of items kept
        else:
            full_bidx_map, bidx = torch.arange(Bq, device=pair.device), full_bidx
        ret = dict(num_rays=flat["num_rays"], rays_inds=rinds, rays_bidx=bidx, rays_full_bidx=full_bidx,
                   full_bidx_map=full_bidx_map, rays_o=flat["rays_o"], rays_d=flat["rays_d"], near=flat["near"],
                   far=flat["far"])
        for k, v in extra.items():
            if isinstance(v, torch.Tensor) and v.shape[:2] == (Bq, N):
                vf = v.transpose(0, 1).reshape(N * Bq, *v.shape[2:])
                if vf.requires_grad and vf.dim() == 2 and vf.dtype == torch.float32:
                    from ..losses import embedding_lookup
                    ret[k] = embedding_lookup(vf, pair)
                else:
                    ret[k] = vf[pair]
            else:
                ret[k] = v
        return ret

    def batched_ray_query(self, *, batched_ray_input: dict = None, batched_ray_tested: dict, config,
                          return_buffer: bool = True, return_details: bool = False,
                          render_per_obj_individual: bool = False) -> Dict:
        """``march_occ_multi_upsample[_compressed]`` on the tested (item, ray) pairs; the volume buffer is packed per
        pair and carries ``rays_bidx_hit`` / ``rays_full_bidx_hit`` next to ``rays_inds_hit``."""
        bt = batched_ray_tested
        empty = int(bt["num_rays"]) == 0
        # (no pair was hit: the reference's renderer does not even set a condition then -- ``if num_rays > 0:
        # model.set_condition(...)``, buffer_compose_renderer.py:252-258 -- and still calls the query for its empty buffer)
        assert empty or self.ins_inds_per_batch is not None, "set_condition() first"
        if empty:
            tested = dict(bt)
            cfg0 = dict(config, _render=True) if render_per_obj_individual else config
            ret = super().ray_query(ray_input=None, ray_tested=tested, config=cfg0, return_buffer=return_buffer,
                                    return_details=return_details, render_per_obj_individual=False)
            if render_per_obj_individual and batched_ray_input is not None and batched_ray_input.get("rays_o") is not None:
                Bq, N = batched_ray_input["rays_o"].shape[:2]
                dev = batched_ray_input["rays_o"].device
                keys = ["mask_volume", "depth_volume"] + (["rgb_volume"] if dict(config).get("with_rgb", True) else []) + \
                    (["normals_volume"] if dict(config).get("with_normal", False) else [])
                ret["rendered"] = {k: torch.zeros([Bq, N, *((3,) if k in ("rgb_volume", "normals_volume") else ())],
                                                  dtype=torch.float32, device=dev) for k in keys}
            return ret
        # The reference conditions the model on the COMPACTED batch -- ``set_condition({'ins_id': [ids of the items
        # that were hit at all]})`` after ``batched_ray_test(compact_batch=True)`` (buffer_compose_renderer.py:247-258)
        # -- so the condition is indexed by ``rays_bidx``; a condition given over the full batch by ``rays_full_bidx``
        # (when every item was hit the two coincide).
        n_cond = int(self.ins_inds_per_batch.shape[0])
        if "full_bidx_map" in bt and n_cond == int(bt["full_bidx_map"].shape[0]):
            which = bt["rays_bidx"]
        else:
            which = bt["rays_full_bidx"]
            if batched_ray_input is not None and batched_ray_input.get("rays_o") is not None:
                assert n_cond == int(batched_ray_input["rays_o"].shape[0]), \
                    "set_condition() covers neither the compacted nor the full batch of batched_ray_tested"
        ins = self.ins_inds_per_batch[which] if bt["num_rays"] > 0 else which
        tested = dict(bt)
        tested.update(self._pair_extras(ins, which))
        want_pairs = bool(dict(config).get("_render", False))
        cfg = dict(config, _render=True) if render_per_obj_individual else config
        if dict(cfg).get("_jitter_full") is not None and bt["num_rays"] > 0:
            # perturbation randoms given per RAY of the batch ([N], [N, C]; parity tests): a pair uses its ray's
            cfg = dict(cfg, _jitter=cfg["_jitter_full"][bt["rays_inds"]].contiguous(),
                       _jitter_c=cfg["_jitter_c_full"][bt["rays_inds"]].contiguous())
        ret = super().ray_query(ray_input=None, ray_tested=tested, config=cfg, return_buffer=return_buffer,
                                return_details=return_details, render_per_obj_individual=False)
        if render_per_obj_individual:
            # every batch item alone, as [B', N(, 3)] images over all the rays (buffer_compose_renderer.py:268-275 slices
            # them per object and indexes them with (rays_full_bidx, rays_inds))
            assert batched_ray_input is not None and batched_ray_input.get("rays_o") is not None, \
                "render_per_obj_individual needs batched_ray_input (the [B', N, 3] rays)"
            Bq, N = batched_ray_input["rays_o"].shape[:2]
            pairs = ret.pop("rendered", None)
            dev = bt["rays_inds"].device
            keys = ["mask_volume", "depth_volume"] + (["rgb_volume"] if dict(config).get("with_rgb", True) else []) + \
                (["normals_volume"] if dict(config).get("with_normal", False) else [])
            where = (bt["rays_full_bidx"], bt["rays_inds"])
            full = {}
            for k in keys:
                tail = (3,) if k in ("rgb_volume", "normals_volume") else ()
                z = torch.zeros([Bq, N, *tail], dtype=torch.float32, device=dev)
                full[k] = z.index_put(where, pairs[k]) if pairs is not None and k in pairs else z
            ret["rendered"] = full
            if want_pairs and pairs is not None:
                ret["rendered_pairs"] = pairs
        vb = ret["volume_buffer"]
        if vb["type"] != "empty":
            sel = getattr(self, "_rays_sel", None)      # upsample_on_marched_only: the pairs that produced samples
            vb["rays_bidx_hit"] = bt["rays_bidx"] if sel is None else bt["rays_bidx"][sel]
            vb["rays_full_bidx_hit"] = bt["rays_full_bidx"] if sel is None else bt["rays_full_bidx"][sel]
        return ret

    def ray_test(self, *a, **k):
        raise RuntimeError(f"{type(self).__name__}: use batched_ray_test / batched_ray_query")


class BatchedLoTDNeuSModel(BatchedRaysMixin, LoTDNeuSModel):
    is_ray_query_supported = True
    is_batched_query_supported = True

    def __init__(self, num_instances: int, ins_ids: Optional[Sequence[str]] = None, accel_cfg: dict = None,
                 seed: int = 42, param_bound: float = 1e-4, lotd_grower_cfg: dict = None, latents_cfg: dict = None,
                 own_latents: bool = True, **kw):
        """``lotd_grower_cfg`` (``DenseLoTDGrowerFMM`` parameters: z_dim, lod_res, lod_n_feats, D, W, fmm_rank,
        n_frequencies -- no_fg_occ.221218.yaml:320-337) + ``latents_cfg{z{dim}}`` (:310-312): the per-instance tables are
        GROWN from per-instance latent codes (auto-decoder: one learnable code per instance) instead of being free
        parameters; ``set_condition`` then takes ``ins_id`` / ``ins_ind`` (codes looked up) or ``z_ins`` (codes given).
        The reference's block form ``{target: ...MixedLoTDGrower, param: {grower_configs: [DenseLoTDGrowerFMM, VMSplitLoTDGrowerFMM]}}``
        (:319-352) is accepted as well (``lotd_growers.build_grower``).  ``own_latents=False``: the codes live outside the
        model (the reference's ``AutoDecoderMixin`` keeps them in ``self._latents``) -- conditions then always carry ``z_ins``."""
        accel_cfg = dict(accel_cfg or {})
        self.grower = None
        if lotd_grower_cfg is not None:
            from ..grid_encodings.lotd_growers import build_grower
            gcfg = dict(lotd_grower_cfg)
            lat = dict(latents_cfg or {})
            lat_z = lat.get("z", lat.get("z_ins", {})) or {}
            z_dim = int(lat_z.get("dim", gcfg.get("z_dim", 128)))
            grower = build_grower(gcfg, z_dim=z_dim, seed=seed + 5)
            kw = dict(kw, lod_res=grower.kernel_lod_res, log2_hashmap_size=max(int(kw.get("log2_hashmap_size", 19)),
                                                                               (max(grower.lod_res) ** 3).bit_length()))
        super().__init__(accel_cfg=accel_cfg, seed=seed, param_bound=param_bound, **kw)
        B = self.num_instances = int(num_instances)
        n = self.n_params_per_instance = self.encoding.cfg.n_params
        assert B * n < 2 ** 31, "instance offsets are 32-bit inside the kernels"
        g = torch.Generator().manual_seed(seed + 17)
        self._cond_table = self._cond_table16 = None
        if lotd_grower_cfg is not None:
            self.grower = grower
            assert grower.n_params == n and all(t == "Dense" for t in self.encoding.cfg.lod_types)
            if own_latents:
                self.latents = nn.Parameter(torch.randn(B, grower.z_dim, generator=g) * 0.1)      # ``z_ins_all`` (auto-decoder)
            else:
                self.latents = None
            p = torch.zeros([0])                                           # no free table
        else:
            p = ((torch.rand(B * n, generator=g) * 2 - 1) * param_bound).half().float()
        self.encoding.flattened_params = nn.Parameter(p)
        self.encoding.params16 = p.half()
        self.encoding._shadow_version = -1
        self.accel = OccGridAccelBatched(
            self.accel.aabb, B, resolution=accel_cfg.get("resolution", (32, 32, 32)),
            occ_thre=accel_cfg.get("occ_thre", 0.3), ema_decay=accel_cfg.get("ema_decay", 0.95),
            inv_s=accel_cfg.get("occ_val_fn_cfg", {}).get("inv_s", 256.0), num_steps=accel_cfg.get("num_steps", 4),
            num_pts=accel_cfg.get("num_pts", 2 ** 16), n_steps_between_update=accel_cfg.get("n_steps_between_update", 16),
            n_steps_warmup=accel_cfg.get("n_steps_warmup", 256))
        self._index_maps = {"ins_id": {k: i for i, k in enumerate(ins_ids or [str(i) for i in range(B)])}}
        self.ins_inds_per_batch: Optional[torch.Tensor] = None

    def _param_groups(self, cfg: dict):
        if self.grower is None:
            return super()._param_groups(cfg)
        return ([dict(name="latents.z_ins", params=[self.latents])] if self.latents is not None else []) + [
                dict(name="implicit_surface.encoding.grower", params=list(self.grower.parameters())),
                dict(name="implicit_surface.decoder", params=[self.sdf_w, self.sdf_b]),
                dict(name="radiance_net", params=[self.rad_w, self.rad_b]),
                dict(name="ln_inv_s", params=[self.ln_inv_s], betas=tuple(cfg.get("invs_betas", (0.9, 0.999))))]

    # ------------------------------------------------------------------ conditions (batched_neus.py:380-407)
    def set_condition(self, batched_infos: Dict):
        if self.grower is not None:
            return self._set_condition_grown(batched_infos)
        if "z_ins" in batched_infos:
            raise RuntimeError("set_condition({'z_ins': ...}) needs a model built with lotd_grower_cfg (free per-instance "
                               "tables have no latent)")
        if "ins_id" in batched_infos:
            ids = batched_infos["ins_id"]
            ids = [ids] if isinstance(ids, str) else ids
            inds = torch.tensor([self._index_maps["ins_id"][i] for i in ids], dtype=torch.long, device=self.device)
        elif "ins_ind" in batched_infos:
            inds = torch.as_tensor(batched_infos["ins_ind"], dtype=torch.long, device=self.device).reshape(-1)
        else:
            raise RuntimeError("set_condition needs 'ins_id' or 'ins_ind'")
        self.ins_inds_per_batch = inds

    def _set_condition_grown(self, batched_infos: Dict):
        """``set_condition`` of the latent-conditioned model (app/models/shared/batched_neus.py:380-403): instance
        indices from ``ins_id`` / ``ins_ind`` (optional when ``z_ins`` is given), codes from ``z_ins`` or the
        auto-decoder's table, then the batch's tables are grown ONCE for every query of this condition."""
        inds = None
        if "ins_id" in batched_infos:
            ids = batched_infos["ins_id"]
            ids = [ids] if isinstance(ids, str) else ids
            inds = torch.tensor([self._index_maps["ins_id"][i] for i in ids], dtype=torch.long, device=self.device)
        elif "ins_ind" in batched_infos:
            inds = torch.as_tensor(batched_infos["ins_ind"], dtype=torch.long, device=self.device).reshape(-1)
        if "z_ins" in batched_infos:
            z = batched_infos["z_ins"].to(self.device).float().reshape(-1, self.grower.z_dim)
        else:
            if inds is None or self.latents is None:
                raise RuntimeError("set_condition needs 'ins_id' / 'ins_ind' (and codes owned by the model) when no 'z_ins' "
                                   "is provided")
            z = self.latents[inds]
        if inds is not None and inds.shape[0] != z.shape[0]:
            raise RuntimeError("set_condition: 'z_ins' and the instance list differ in length")
        self.z_ins_per_batch = z
        # without instance indices the batch items use occupancy grid 0..B'-1 (a condition on bare codes has no
        # per-instance state to look up)
        self.ins_inds_per_batch = inds if inds is not None else torch.arange(z.shape[0], device=self.device)
        self._cond_table = self.grower(z).reshape(-1)
        self._cond_table16 = self._cond_table.detach().half()

    def clean_condition(self):
        self.ins_inds_per_batch = None
        self._cond_table = self._cond_table16 = None

    def _table(self):
        if self.grower is None:
            return self.encoding.flattened_params
        assert self._cond_table is not None, "set_condition() first"
        return self._cond_table

    def _table16(self):
        if self.grower is None:
            return self.encoding.shadow()
        assert self._cond_table16 is not None, "set_condition() first"
        return self._cond_table16

    def _offsets(self, ins_inds: torch.Tensor, positions: torch.Tensor = None):
        """-> (table offset, occupancy word offset) per ray.  Free tables: both by instance index.  Grown tables live in
        the order of the CONDITION (``positions`` = index of the ray's item in it), the occupancy grids stay per instance."""
        if self.grower is not None and positions is not None:
            return positions * self.n_params_per_instance, ins_inds * self.accel.words_per_instance
        return ins_inds * self.n_params_per_instance, ins_inds * self.accel.words_per_instance

    def _pair_extras(self, ins: torch.Tensor, which: torch.Tensor) -> Dict:
        goff, woff = self._offsets(ins, which)
        return dict(rays_goff=goff.contiguous(), rays_word_off=woff.contiguous())

    # ------------------------------------------------------------------ per-instance point queries
    @torch.no_grad()
    def query_sdf(self, x: torch.Tensor, ins_ind=None, bidx: torch.Tensor = None) -> torch.Tensor:
        """SDF of points of ONE instance (``ins_ind: int``) or of per-point batch items (``bidx [S]`` indices into the
        current condition)."""
        shape = x.shape[:-1]
        xf = x.detach().float().reshape(-1, 3).contiguous()
        S = xf.shape[0]
        own_condition = False
        if bidx is None:
            if self.grower is not None:                 # one instance: grow its table for this query
                own_condition = self._cond_table is None
                if own_condition:
                    self.set_condition({"ins_ind": torch.tensor([int(ins_ind)], dtype=torch.long, device=xf.device)})
                pos = (self.ins_inds_per_batch == int(ins_ind)).nonzero()[:1, 0]
                goff = pos * self.n_params_per_instance
            else:
                goff = torch.tensor([int(ins_ind) * self.n_params_per_instance], dtype=torch.long, device=xf.device)
            ridx = torch.zeros([S], dtype=torch.long, device=xf.device)
        else:
            goff, _ = self._offsets(self.ins_inds_per_batch, torch.arange(self.ins_inds_per_batch.shape[0], device=xf.device))
            ridx = bidx.reshape(-1).contiguous()
        grid16, wpack = self._shadow()
        out = self._sdf_query(grid16, wpack, xf, None, None, None, ridx, S, xf.device, goff=goff).reshape(shape)
        if own_condition:
            self.clean_condition()
        return out

    def forward_sdf_nablas(self, x: torch.Tensor, bidx: torch.Tensor = None, ins_ind=None, nablas_has_grad=True):
        from .neus import _FieldFn
        shape = x.shape[:-1]
        xf = x.detach().float().reshape(-1, 3).contiguous()
        if bidx is None:
            assert self.grower is None or self._cond_table is not None, "grown tables: set_condition() first, then bidx"
            pos = int(ins_ind) if self.grower is None else int((self.ins_inds_per_batch == int(ins_ind)).nonzero()[0, 0])
            goff = torch.tensor([pos * self.n_params_per_instance], dtype=torch.long, device=xf.device)
            ridx = torch.zeros([xf.shape[0]], dtype=torch.long, device=xf.device)
        else:
            goff, _ = self._offsets(self.ins_inds_per_batch, torch.arange(self.ins_inds_per_batch.shape[0], device=xf.device))
            ridx = bidx.reshape(-1).contiguous()
        sdf, nablas = _FieldFn.apply(self, self._table(), self.sdf_w, self.sdf_b, self.rad_w,
                                     self.rad_b, None, xf, None, None, None, ridx, False, goff)
        if not nablas_has_grad:
            nablas = nablas.detach()
        return dict(sdf=sdf.reshape(shape), nablas=nablas.reshape(*shape, 3))

    def pretrain_generator_sphere(self, radius: float = 0.5, num_iters: int = 200, lr: float = 1e-3, num_pts: int = 2 ** 12,
                                  z_std: float = 0.1, n_codes: int = 4, seed: int = 0, logger=None, w_eikonal: float = 0.05) -> float:
        """Pre-training of a GROWN model (no table to write a sphere into): Adam over the grower and the SDF decoder so that
        the SDF of codes z ~ N(0, z_std^2) -- the neighbourhood the auto-decoder's codes start in (``zero_init_latents``,
        no_fg_occ.221218.yaml:388) -- is |u| - radius, through the model's own forward / backward kernels."""
        assert self.grower is not None
        dev = self.sdf_w.device
        params = list(self.grower.parameters()) + [self.sdf_w, self.sdf_b]
        opt = torch.optim.Adam(params, lr=lr)
        g = torch.Generator(device=dev).manual_seed(seed)
        lo, hi = self.accel.aabb[0].to(dev), self.accel.aabb[1].to(dev)
        saved = (self.ins_inds_per_batch, getattr(self, "z_ins_per_batch", None))
        loss = torch.zeros([])
        with torch.enable_grad():
            for it in range(int(num_iters)):
                z = torch.randn([n_codes, self.grower.z_dim], device=dev, generator=g) * z_std
                self._set_condition_grown({"z_ins": z})
                x = lo + (hi - lo) * torch.rand([num_pts, 3], device=dev, generator=g)
                bidx = torch.randint(0, n_codes, [num_pts], device=dev, generator=g)
                u = (x - (lo + hi) * 0.5) / ((hi - lo) * 0.5)
                out = self.forward_sdf_nablas(x, bidx=bidx)
                loss = (out["sdf"] - (u.norm(dim=-1) - radius) * float(((hi - lo) * 0.5).min())).abs().mean()
                total = loss + w_eikonal * ((out["nablas"].norm(dim=-1) - 1.0) ** 2).mean()
                opt.zero_grad(set_to_none=True)
                total.backward()
                opt.step()
                self._wpack_versions = None
                if logger is not None and it % 100 == 0:
                    logger.info(f"pretrain_generator: it {it} loss {float(loss.detach()):.5f}")
        for q in params:
            q.grad = None
        BatchedLoTDNeuSModel.clean_condition(self)
        self.ins_inds_per_batch = saved[0]
        self.is_pretrained.fill_(True)
        return float(loss.detach())

    @torch.no_grad()
    def geometric_init_sphere(self, radius: float = 0.5, noise_scale: float = 0.25, inside_out: bool = None,
                              level: int = None):
        """The single-instance initialisation, with the sphere level copied into every instance's table."""
        n = self.n_params_per_instance
        full = self.encoding.flattened_params
        self.encoding.flattened_params = nn.Parameter(full.data[:n].clone())
        super().geometric_init_sphere(radius, noise_scale, inside_out, level)
        first = self.encoding.flattened_params.data
        cfg = self.encoding.cfg
        lv = max(l for l, t in enumerate(cfg.lod_types) if t == "Dense") if level is None else int(level)
        lo, hi = cfg.lod_offsets[lv], cfg.lod_offsets[lv] + cfg.lod_sizes[lv] * 2
        for b in range(self.num_instances):
            full.data[b * n + lo: b * n + hi] = first[lo:hi]
        self.encoding.flattened_params = full
        self.encoding.flattened_params.add_(0)
        return self

    @torch.no_grad()
    def geometric_init_instances(self, radii: Sequence[float], noise_scale: float = 0.25):
        """Every instance its own sphere (radius in units of the half bounding size): the pass-through decoder of
        ``geometric_init_sphere`` + per-instance sphere levels -- the synthetic stand-in for per-instance latents."""
        assert len(radii) == self.num_instances
        self.geometric_init_sphere(float(radii[0]), noise_scale=noise_scale)
        cfg = self.encoding.cfg
        lv = max(l for l, t in enumerate(cfg.lod_types) if t == "Dense")
        Rx, Ry, Rz = cfg.lod_res3[lv]
        zz, yy, xx = torch.meshgrid(torch.linspace(-1.0, 1.0, Rz), torch.linspace(-1.0, 1.0, Ry),
                                    torch.linspace(-1.0, 1.0, Rx), indexing="ij")
        rr = torch.sqrt(xx ** 2 + yy ** 2 + zz ** 2).reshape(-1)
        half = float((self.accel.aabb[1] - self.accel.aabb[0]).min()) / 2.0
        n, lo = self.n_params_per_instance, cfg.lod_offsets[lv]
        full = self.encoding.flattened_params.data
        for b, r in enumerate(radii):
            lvl = full[b * n + lo: b * n + lo + cfg.lod_sizes[lv] * 2].view(-1, 2)
            # object-space distance: the unit-cube sphere scaled to the model's box
            lvl[:, 0] = ((rr - float(r) / half) * half * self.sdf_scale).half().float().to(lvl.device)
        self.encoding.flattened_params.add_(0)
        return self

    def init_accel(self, generator=None, **kw):
        self.accel.occ_val.zero_()
        self.accel.update_from_net(lambda pts, b: self.query_sdf(pts, ins_ind=b), generator=generator, **kw)



def num_instances_of(model) -> int:
    return getattr(model, "num_instances", 1)


__all__: List[str] = ["BatchedLoTDNeuSModel", "OccGridAccelBatched"]
