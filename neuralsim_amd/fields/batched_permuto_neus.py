"""Batched GenerativePermutoConcat NeuS model -- SURVEY sec. 8 rows a20 / f4: the shared foreground model of the reference's
current multi-object configs (``model_class: app.models.shared.AD_GenerativePermutoConcatNeuSObj``,
code_multi/configs/exps/fg_neus=permuto/all_occ.240201.yaml:425-506; app/models/shared/batched_neus.py:295-407).

ONE permutohedral table and ONE pair of decoders serve every instance; what distinguishes the instances is a latent code
concatenated to the position (a (3 + z_dim)-dimensional lattice, csrc/permuto.hip) and a per-instance occupancy grid
(``accel_cfg{type: occ_grid_batched}``).  Batched rendering = the (item, ray) pair list of ``BatchedRaysMixin``: every pair
carries its item's code (``z`` rides along the ray index inside the kernels) and its item's occupancy words.  The codes are
LEARNED (auto-decoder): a condition whose ``z`` requires grad receives d L / d z (``nsim_permuto_dz``)."""
from typing import Dict, Optional, Sequence

import torch

from .batched_neus import BatchedRaysMixin, OccGridAccelBatched
from .permuto_neus import PermutoNeuSModel


class BatchedPermutoNeuSModel(BatchedRaysMixin, PermutoNeuSModel):
    is_ray_query_supported = True
    is_batched_query_supported = True

    def __init__(self, num_instances: int, z_dim: int = 4, ins_ids: Optional[Sequence[str]] = None, accel_cfg: dict = None,
                 **kw):
        accel_cfg = dict(accel_cfg or {})
        kw.pop("latents_cfg", None)             # the latent width arrives as z_dim (populate(n_latent_dim=))
        if "surface_cfg" in kw:                 # the reference's block: translated by PermutoNeuSModel, minus the batched accel
            super().__init__(z_dim=int(z_dim), accel_cfg=None, **kw)
        else:
            super().__init__(z_dim=int(z_dim), accel_cfg={k: v for k, v in accel_cfg.items() if k == "resolution"}, **kw)
        assert self.z_dim > 0, "a batched permutohedral model distinguishes its instances by their codes"
        B = self.num_instances = int(num_instances)
        oc = accel_cfg.get("occ_val_fn_cfg") or {}
        self.accel = OccGridAccelBatched(
            self.accel.aabb, B, resolution=accel_cfg.get("resolution", (32, 32, 32)), occ_thre=accel_cfg.get("occ_thre", 0.3),
            ema_decay=accel_cfg.get("ema_decay", 0.95), inv_s=oc.get("inv_s", 256.0),
            num_steps=(accel_cfg.get("init_cfg") or {}).get("num_steps", accel_cfg.get("num_steps", 4)),
            num_pts=(accel_cfg.get("init_cfg") or {}).get("num_pts", accel_cfg.get("num_pts", 2 ** 16)),
            n_steps_between_update=accel_cfg.get("n_steps_between_update", 16), n_steps_warmup=accel_cfg.get("n_steps_warmup", 256))
        self._index_maps = {"ins_id": {k: i for i, k in enumerate(ins_ids or [str(i) for i in range(B)])}}
        self.ins_inds_per_batch: Optional[torch.Tensor] = None
        self.z_ins_per_batch: Optional[torch.Tensor] = None

    # ------------------------------------------------------------------ conditions (batched_neus.py:380-407)
    def set_condition(self, z: torch.Tensor = None, ins_inds_per_batch: torch.Tensor = None):
        """z [B', z_dim]: the codes of the batch items (may require grad); ``ins_inds_per_batch`` [B']: their instance indices
        (occupancy grids), default 0..B'-1."""
        assert z is not None and z.shape[-1] == self.z_dim
        z = z.reshape(-1, self.z_dim)
        dev = self.sdf_w.device
        self.z_ins_per_batch = z.to(dev)
        self.ins_inds_per_batch = (torch.as_tensor(ins_inds_per_batch, dtype=torch.long, device=dev).reshape(-1)
                                   if ins_inds_per_batch is not None else torch.arange(z.shape[0], device=dev))
        assert self.ins_inds_per_batch.shape[0] == z.shape[0], "set_condition: codes and instance list differ in length"
        PermutoNeuSModel.set_condition(self, self.z_ins_per_batch)          # rows = batch items, until a pair list replaces them

    def clean_condition(self):
        self.ins_inds_per_batch = self.z_ins_per_batch = None
        PermutoNeuSModel.set_condition(self, None)

    def _pair_extras(self, ins: torch.Tensor, which: torch.Tensor) -> Dict:
        # the pairs' codes: row r of the kernels' z array is pair r's item (an index op: gradients flow back to the condition)
        PermutoNeuSModel.set_condition(self, self.z_ins_per_batch[which])
        return dict(rays_word_off=(ins * self.accel.words_per_instance).contiguous())

    def batched_ray_query(self, **kw):
        try:
            return super().batched_ray_query(**kw)
        finally:        # back to one row per batch item (point queries index it with bidx); the query's backward keeps the
            if self.z_ins_per_batch is not None:      # per-pair codes it was made under (the enc_state on its ctx)
                PermutoNeuSModel.set_condition(self, self.z_ins_per_batch)

    # ------------------------------------------------------------------ per-instance point queries
    def _point_rows(self, n: int, dev, ins_ind=None, bidx: torch.Tensor = None) -> torch.Tensor:
        if bidx is not None:
            return bidx.reshape(-1).to(dev).long().contiguous()
        if ins_ind is None:         # no item named (pre-training): the points are dealt round-robin to the condition's codes
            assert self._z_rays is not None, "set_condition() first"
            return (torch.arange(n, device=dev) % int(self._z_rays.shape[0])).contiguous()
        assert self.ins_inds_per_batch is not None, "set_condition() first"
        pos = (self.ins_inds_per_batch == int(ins_ind)).nonzero()
        assert pos.numel() > 0, f"instance {ins_ind} is not in the current condition"
        return torch.full([n], int(pos[0, 0]), dtype=torch.long, device=dev)

    @torch.no_grad()
    def query_sdf(self, x: torch.Tensor, ins_ind=None, bidx: torch.Tensor = None) -> torch.Tensor:
        shape = x.shape[:-1]
        xf = x.detach().float().reshape(-1, 3).contiguous()
        ridx = self._point_rows(xf.shape[0], xf.device, ins_ind, bidx)
        grid16, wpack = self._shadow()
        return self._sdf_query(grid16, wpack, xf, None, None, None, ridx, xf.shape[0], xf.device).reshape(shape)

    def forward_sdf_nablas(self, x: torch.Tensor, bidx: torch.Tensor = None, ins_ind=None, nablas_has_grad: bool = True):
        from .neus import _FieldFn
        shape = x.shape[:-1]
        xf = x.detach().float().reshape(-1, 3).contiguous()
        ridx = self._point_rows(xf.shape[0], xf.device, ins_ind, bidx)
        sdf, nablas = _FieldFn.apply(self, self._table(), self.sdf_w, self.sdf_b, self.rad_w, self.rad_b, None, xf, None, None,
                                     None, ridx, False)
        if not nablas_has_grad:
            nablas = nablas.detach()
        return dict(sdf=sdf.reshape(shape), nablas=nablas.reshape(*shape, 3))

    def init_accel(self, generator=None, **kw):
        self.accel.occ_val.zero_()
        self.accel.update_from_net(lambda pts, b: self.query_sdf(pts, ins_ind=b), generator=generator, **kw)

    @torch.no_grad()
    def training_initialize(self, config=None, logger=None, log_prefix=None, skip_accel: bool = False) -> bool:
        """``geo_init_method: pretrain`` (all_occ.240201.yaml:451) with codes drawn around zero: the pre-training loop of the
        single permutohedral model under a [n_codes, z_dim] condition; the all-instance ``accel.init`` is the caller's
        (app/models/shared/batched_neus.py:385-395)."""
        post = getattr(self, "_reference_post", {})
        cfg = dict(config or {})
        updated = False
        if not bool(self.is_pretrained):
            ext = (self.accel.aabb[1] - self.accel.aabb[0]).cpu()
            r = float(post.get("radius_init", 0.5)) / (float(ext.min()) / 2.0)
            saved = (self.z_ins_per_batch, self.ins_inds_per_batch)
            # ``initialize_cfg{num_iters, lr, num_points, batch_size, resample_z}`` (all_occ.240201.yaml:494-499): batch_size codes
            # around zero (the auto-decoder's codes start at zero: ``weight_init: zero``), points dealt round-robin to them
            n_codes = int(cfg.get("batch_size", cfg.get("n_codes", 4)))
            g = torch.Generator(device=self.sdf_w.device).manual_seed(0)
            z = torch.randn([n_codes, self.z_dim], device=self.sdf_w.device, generator=g) * float(cfg.get("z_std", 0.1))
            PermutoNeuSModel.set_condition(self, z)
            with torch.enable_grad():
                self.geometric_init_sphere(min(r, 0.95), num_iters=int(cfg.get("num_iters", 300)), lr=float(cfg.get("lr", 2e-3)),
                                           num_pts=int(cfg.get("num_points", cfg.get("num_pts", 2 ** 14))),
                                           w_eikonal=float(cfg.get("w_eikonal", 0.1)))
            if saved[0] is not None:
                self.set_condition(z=saved[0], ins_inds_per_batch=saved[1])
            else:
                self.clean_condition()
            updated = True
        if not skip_accel and self.accel is not None and self.ins_inds_per_batch is not None:
            self.accel.init(lambda pts, b: self.query_sdf(pts, ins_ind=b), logger=logger)
        return updated
