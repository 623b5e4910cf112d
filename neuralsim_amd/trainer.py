"""One training iteration of the NeuS render step: ray generation -> ray_test -> ray_query -> volume
integration -> losses -> backward -> gradient all-reduce -> Adam (+ the periodic occupancy refresh).

Follows ``Trainer.train_step_pixel`` and the main loop of the reference
(code_single/tools/train.py:544-696, 1444-1502): photometric ``mse`` on the rendered rgb
(app/loss/photometric.py:88-146), eikonal on the render samples and on uniformly sampled points
(app/loss/eikonal.py:216-251; ``num_uniform`` code_single/tools/train.py:602-613), per-frame appearance
embeddings (app/models/scene/image_embeddings.py:23-80).  Data is synthetic (posed pinhole cameras of
SURVEY.md sec. 8d; targets = the analytic image of the synthetic sphere, or random colours).
"""
from typing import Dict, Optional

import torch
import torch.nn as nn

from . import _lib, backward_on_calling_thread, distributed as ndist
import os

from .fields.neus import LoTDNeuSModel, volume_integration, append_extra_points, _flat_sizes
from .graphics.cameras import selected_rays
from .optim import FusedAdam
_PREFETCH_HANDOFF = os.environ.get("NSIM_PREFETCH_HANDOFF", "1") == "1"
from .losses import eikonal_loss, mse_loss, embedding_lookup, mono_depth_loss, mono_normal_loss


class RenderTrainer:
    def __init__(self, model: LoTDNeuSModel, intr, c2w, WH, num_rays: int, lr: float = 1e-2, w_eikonal: float = 0.1,
                 num_uniform: int = 4096, near: float = 0.01, far: Optional[float] = None, n_appear: int = 4,
                 perturb: bool = True, rank: int = 0, world_size: int = 1, seed: int = 42, learn_inv_s: bool = True,
                 distant_model=None, sky_model=None, level_anneal: Optional[dict] = None,
                 target_sphere_radius: Optional[float] = None, pipeline: bool = True,
                 pose_refine: Optional[dict] = None, c2w_true=None, fused_step: Optional[bool] = None,
                 distortion: Optional[torch.Tensor] = None, target_images: Optional[torch.Tensor] = None,
                 mono: Optional[dict] = None, rgb_fn: str = "mse", lidar: Optional[dict] = None):
        """pose_refine: ``dict(lr=1e-4, start_it=500)`` -- per-frame pose corrections (an axis-angle rotation and a
        translation, ``c2w' = [R Exp(w) | T + dT]``) trained through the rays from ``start_it`` on, standing in for the
        reference's ``LearnableParams`` (withmask_withlidar_joint.240219.yaml:338-352; the parametrisation of the
        absent nr3d_lib is not known -- semantics fixed here).  ``c2w_true``: the poses the synthetic targets are
        rendered from when they differ from the (noisy) ``c2w`` the training starts with.
        distortion [V,5]: ``camera_model: opencv`` (k1, k2, p1, p2, k3 per frame; the street configs); None = pinhole.
        target_images [V,H,W,3]: the dataset's images resident in HBM; a batch gathers its pixels from them, as the
        reference's pixel loader does from its preloaded images (dataio/data_loader/pixel_loader.py:323-327).
        mono: the indoor config's monocular supervision (lotd_neus.replica.230814.yaml:236-238, 272-288):
        ``dict(depth=[V,H,W], normals=[V,H,W,3], patch_hw=(64, 64), w_depth=, w_normal=)`` -- the first h*w rays of every
        batch are a contiguous pixel patch of one frame (the reference's ``image_patch`` step; the scale-and-shift
        invariant depth term needs an image region), normals are supervised on every ray.
        rgb_fn: ``mse`` | ``l1`` (``rgb_fn`` of the street configs, withmask_withlidar_joint.240219.yaml:26).
        lidar: the street config's second step of every iteration (``num_rays_lidar 8192``, yaml:8; losses :453-469;
        code_single/tools/train.py:860-960 ``train_step_lidar``): ``dict(rays_o [F,M,3], rays_d [F,M,3], ranges [F,M]
        (0 = no return), num_rays=8192, w_depth=0.02, w_los=0.1, epsilon=1.5, discard_toofar=80)`` -- a batch of lidar
        beams rendered with ``with_rgb=False`` (no radiance network), l1 depth loss on the returns + the line-of-sight
        term of ``neus_unisim`` (squared visibility weights further than epsilon from the return), own backward and
        optimizer step, as in the reference's loop (train.py:1540-1590)."""
        self.model = model
        self.target_images, self.mono, self.rgb_fn = target_images, (dict(mono) if mono else None), rgb_fn
        self.lidar = dict(lidar) if lidar else None
        self._last_aux = None
        self.distortion = distortion
        # fused_step (default on, env NSIM_FUSED_STEP=0 turns it off): the differentiable part of the iteration -- field
        # forward, sdf->alpha, compositing, losses and their whole backward -- is issued as one straight chain of
        # launches without the autograd engine in between (``_train_render_fused``); same kernels, same numbers
        self.fused_step = (os.environ.get("NSIM_FUSED_STEP", "1") == "1") if fused_step is None else bool(fused_step)
        # N > 1: overlap the table-gradient all-reduce with the second half of the scatter (NSIM_OVERLAP_ALLREDUCE=0: off)
        self.overlap_allreduce = os.environ.get("NSIM_OVERLAP_ALLREDUCE", "1") == "1"
        self._step_done = False
        # the with-grad gather + decoders are queued at a capacity BEFORE the size of the kept sample set is read
        # (NSIM_SPEC_FORWARD=0: after it, at the exact size)
        self.spec_forward = os.environ.get("NSIM_SPEC_FORWARD", "1") == "1"
        # the differentiable tail of the fused step (compositing, losses, their backward): four launches (default since round 5), or
        # ONE (NSIM_RENDER_HEAD=1, round 4's nsim_render_head).  Round 4 measured the two alike; with the marched-only sampling of
        # round 5 (55 % of the rays carry empty packs) the four launches are ahead: 1.513 / 1.521 against 1.543 / 1.566 ms per
        # step, alternated in one call (profiles/round5_render_head_ab.txt) -- one wave per ray walking its samples three times
        # is the longer critical path (60 us against 8 + 8 + 7 + 13)
        self.render_head = os.environ.get("NSIM_RENDER_HEAD", "0") == "1"
        # measurement aid (bench.py ``exposed_allreduce_ms``): every rank keeps its local gradients, no collective is issued
        self.skip_allreduce = False
        # synthetic supervision: None = random colours; r = analytic image of a Lambert-free sphere of radius r
        # (colour = 0.5 + 0.5 normal on the sphere, black elsewhere) -- multi-view consistent, keeps the geometry put
        self.target_sphere_radius = target_sphere_radius
        self._ray_cache = None
        # software pipelining of the data side: the next batch (rays, targets, AABB test + its compaction sync) is
        # produced while the host waits for the current batch's sample count -- one blocking wait per step
        self.pipeline = pipeline
        self._prefetched = None
        self._retired = None           # the last consumed batch of an event hand-over (see _prefetch)
        # the prefetch runs on its own HIP stream (NSIM_PREFETCH_STREAM=0: on the caller's): its dozen small launches and
        # its hit-ray compaction sync then neither queue behind the sampling kernels of the current step nor drain them
        self._side = None
        if pipeline and model.device.type == "cuda" and os.environ.get("NSIM_PREFETCH_STREAM", "1") == "1":
            self._side = torch.cuda.Stream(device=model.device)
        # encoding_cfg.anneal_cfg{type: hardmask, start_it, stop_it, start_level} (dtu yaml:104-108)
        self.level_anneal = dict(level_anneal) if level_anneal else None
        self.intr, self.c2w, self.WH = intr, c2w, WH
        self.c2w_true = c2w if c2w_true is None else c2w_true
        self.V = intr.shape[0]
        self.pose_refine = dict(pose_refine) if pose_refine else None
        self.pose_delta, self.pose_optim, self._it = None, None, 0
        if self.pose_refine is not None:
            self.pose_delta = nn.Parameter(torch.zeros([self.V, 6], device=model.device))
            self.pose_optim = torch.optim.Adam([self.pose_delta], lr=float(self.pose_refine.get("lr", 1e-4)))
        self.num_rays = num_rays             # rays per rank per iteration (weak scaling, as the reference's DDP)
        self.w_eikonal, self.num_uniform = w_eikonal, num_uniform
        self.near, self.far, self.perturb = near, far, perturb
        self.rank, self.world_size = rank, world_size
        dev = model.device
        self.gen = torch.Generator(device=dev).manual_seed(seed + 1000 * rank)
        self.gen_shared = torch.Generator(device=dev).manual_seed(seed)     # rank-shared (occupancy refresh)
        g = torch.Generator().manual_seed(seed)
        self.appear = nn.Parameter((torch.randn(self.V, n_appear, generator=g) * 0.1).to(dev))
        self.optim = FusedAdam(model, lr=lr, learn_inv_s=learn_inv_s)
        self.optim.groups.append(dict(p=self.appear, p16=None, betas=(0.9, 0.99), m=torch.zeros_like(self.appear),
                                      v=torch.zeros_like(self.appear)))
        self.distant_model = distant_model
        if distant_model is not None:
            self.optim.add_distant_model(distant_model)
        self.sky_model = sky_model
        if sky_model is not None:
            self.optim.add_sky_model(sky_model)
        from .renderers.single_volume_renderer import SingleVolumeRenderer
        self.renderer = SingleVolumeRenderer(dict(with_rgb=True, with_normal=True, near=near, far=far, perturb=perturb,
                                                  depth_use_normalized_vw=False)).train()
        self.stats: Dict[str, float] = {}
        if world_size > 1 and getattr(model, "accel", None) is not None:
            import torch.distributed as dist

            def _sync_occ(v):        # union of the ranks' render-time occupancy values before every refresh
                if dist.is_initialized() and not self.skip_allreduce:
                    dist.all_reduce(v, op=dist.ReduceOp.MAX)
            model.accel.sync_values = _sync_occ

    def pose_refine_active(self) -> bool:
        return self.pose_refine is not None and self._it >= int(self.pose_refine.get("start_it", 500))

    def current_c2w(self) -> torch.Tensor:
        """The poses rays are generated from: ``c2w`` or, once the refinement is active, [R Exp(w) | T + dT] (Rodrigues,
        broadcast-multiply-sum as the reference insists for pose composition, nodes.py:79-84)."""
        if not self.pose_refine_active():
            return self.c2w
        w, dT = self.pose_delta[:, :3], self.pose_delta[:, 3:]
        th = w.norm(dim=-1, keepdim=True).clamp_min(1e-12)[..., None]              # [V,1,1]
        K = torch.zeros([self.V, 3, 3], device=w.device, dtype=w.dtype)
        K = K.index_put((torch.arange(self.V, device=w.device)[:, None], torch.tensor([[2, 0, 1]], device=w.device),
                         torch.tensor([[1, 2, 0]], device=w.device)), w)              # K[2,1]=wx K[0,2]=wy K[1,0]=wz
        K = K - K.transpose(1, 2)
        K2 = (K[:, :, :, None] * K[:, None, :, :]).sum(-2)                             # K @ K
        E = torch.eye(3, device=w.device) + torch.sin(th) / th * K + (1.0 - torch.cos(th)) / (th * th) * K2
        R = self.c2w[:, :3, :3]
        Rn = (R[:, :, :, None] * E[:, None, :, :]).sum(-2)                             # R @ E
        top = torch.cat([Rn, (self.c2w[:, :3, 3] + dT)[..., None]], dim=-1)
        return torch.cat([top, self.c2w[:, 3:4, :]], dim=1)

    def sample_batch(self):
        dev = self.model.device
        N = self.num_rays
        xy = torch.rand([N, 2], device=dev, generator=self.gen).clamp_(1e-6, 1 - 1e-6)   # cameras.py:247
        fidx = torch.randint(0, self.V, [N], device=dev, generator=self.gen)
        if self.mono is not None and self.mono.get("patch_hw"):
            # image-patch rays (code_single/tools/train.py:738-739: rays_pix [h,w,2]): rows 0 .. h*w-1 of the batch
            ph, pw = self.mono["patch_hw"]
            W_, H_ = int(self.WH[0, 0]), int(self.WH[0, 1])
            r3 = torch.rand([3], device=dev, generator=self.gen)
            f0 = (r3[0] * self.V).long().clamp_(0, self.V - 1)
            x0, y0 = (r3[1] * (W_ - pw)).floor(), (r3[2] * (H_ - ph)).floor()
            yy, xx = torch.meshgrid(torch.arange(ph, device=dev), torch.arange(pw, device=dev), indexing="ij")
            pxy = torch.stack([(xx.reshape(-1) + x0 + 0.5) / W_, (yy.reshape(-1) + y0 + 0.5) / H_], dim=-1)
            xy = torch.cat([pxy, xy[ph * pw:]])
            fidx = torch.cat([f0.expand(ph * pw), fidx[ph * pw:]])
        if self.target_images is not None:
            H_, W_ = self.target_images.shape[1], self.target_images.shape[2]
            ix = (xy[:, 0] * W_).long().clamp_(0, W_ - 1)
            iy = (xy[:, 1] * H_).long().clamp_(0, H_ - 1)
            self._ray_cache = None
            self._last_aux = None
            if self.mono is not None:
                self._last_aux = dict(depth=self.mono["depth"][fidx, iy, ix], normals=self.mono["normals"][fidx, iy, ix])
            return xy, fidx, self.target_images[fidx, iy, ix]
        if self.target_sphere_radius is None:
            gt = torch.rand([N, 3], device=dev, generator=self.gen)
        else:
            with torch.no_grad():           # the targets are pixels of the TRUE cameras
                o, d = selected_rays(xy, fidx, self.intr, self.c2w_true, self.WH, distortion=self.distortion)
            if self.c2w_true is self.c2w and not self.pose_refine_active():
                self._ray_cache = (xy, o, d)
            else:
                self._ray_cache = None
            gt = self.sphere_image(o, d, self.target_sphere_radius)
        return xy, fidx, gt

    @staticmethod
    def sphere_image(o, d, radius: float):
        """Pixel colours of a sphere at the origin seen along unit rays (o, d): 0.5 + 0.5 n at the first hit, else 0."""
        gt = torch.empty_like(o)
        _lib.call("nsim_sphere_image", _lib.ptr(o.contiguous()), _lib.ptr(d.contiguous()), o.shape[0], float(radius),
                  _lib.ptr(gt))
        return gt

    def render(self, xy, fidx, with_normal=True, extra_pts=None, batch: dict = None):
        """rays -> SingleVolumeRenderer (ray_test, ray_query, [distant model + merge], volume integration).
        ``extra_pts`` [M,3]: free points evaluated by the same field launches (``_FieldFn`` extra_x).
        ``batch``: a prefetched batch (``_make_batch``): rays and AABB test are already there, and the prefetch of
        the NEXT batch is queued right before this render's blocking sample-count read."""
        bypass = dict(_extra_pts=extra_pts) if extra_pts is not None else {}
        tested = None
        if batch is not None:
            rays_o, rays_d, tested = batch["rays_o"], batch["rays_d"], dict(batch["tested"])
            if extra_pts is not None and "o_full" in batch and extra_pts is batch.get("x_uni"):
                bypass["_extra_pre"] = (batch["o_full"], batch["d_full"])      # [R + M] ray arrays, built with the batch
            # no prefetch under pose refinement: it would build the next batch's graph on pose parameters that this
            # step's optimizer then updates in place
            if self.pipeline and not self.pose_refine_active():
                bypass["_pre_sync_hook"] = self._prefetch
            if self.perturb:        # the batch's pre-drawn uniforms (rows 0..R-1 for the R hit rays)
                R = tested["num_rays"]
                bypass["_jitter"], bypass["_jitter_c"] = batch["jitter"][:R], batch["jitter_c"][:R]
        elif self._ray_cache is not None and self._ray_cache[0] is xy:
            _, rays_o, rays_d = self._ray_cache
        else:
            rays_o, rays_d = selected_rays(xy, fidx, self.intr, self.current_c2w(), self.WH, distortion=self.distortion)
        h_appear = None
        if tested is None or self.distant_model is not None or self.sky_model is not None:
            h_appear = embedding_lookup(self.appear, fidx)          # per-ray codes for every ray
        if tested is not None:      # current appearance codes of the rays that hit (the test itself was geometry only)
            tested["rays_h_appear"] = embedding_lookup(self.appear, batch["fidx_hit"]) if h_appear is None \
                else embedding_lookup(h_appear, tested["rays_inds"])
        ret = self.renderer.render(self.model, rays=[rays_o, rays_d], rays_h_appear=h_appear, with_normal=with_normal,
                                   return_buffer=True, return_details=True, distant_model=self.distant_model,
                                   sky_model=self.sky_model, bypass_ray_query_cfg=bypass or None, cr_ray_tested=tested)
        return ret

    # ------------------------------------------------------------------ the differentiable part as one launch chain
    def _fused_ok(self) -> bool:
        m = self.model
        plain = type(m) is LoTDNeuSModel or (type(m).__name__ == "PermutoNeuSModel" and getattr(m, "z_dim", 0) == 0)
        return (self.fused_step and plain and not getattr(m, "pos_embed_E", 0) and self.distant_model is None and self.sky_model is None
                and not self.pose_refine_active() and getattr(m, "_ctrl_mix", 0.0) == 0.0 and self.mono is None
                and self.rgb_fn == "mse")

    def _train_render_fused(self, batch: dict) -> Optional[torch.Tensor]:
        """render + loss + backward of one prefetched batch WITHOUT the autograd engine: the launches the autograd path
        issues (``_FieldFn`` / ``_NeusAlphaFn`` / ``_CompositeFn`` / the fused losses and their backwards), back to
        back, gradients assigned to ``.grad``.  Between the sample-count sync and the first large backward kernel the
        GPU has ~0.2 ms of work queued; the autograd path spends ~0.6 ms of host time there (graph nodes, small
        elementwise kernels, engine hops) and the chip idles -- here the same stretch is ~20 plain launches.
        Covers the object-centric training configuration (one NeuS model, photometric mse + eikonal on render samples and
        uniform points, appearance codes); returns None when no ray produced samples (caller falls back)."""
        model = self.model
        tested = batch["tested"]
        R, N = tested["num_rays"], self.num_rays
        if R == 0:
            return None
        dev = model.device
        cfg = dict(model.ray_query_cfg)
        cfg.update(self.renderer.config)
        cfg.update(with_rgb=True, with_normal=True)
        if self.pipeline:
            cfg["_pre_sync_hook"] = self._prefetch
        if self.perturb:
            cfg["_jitter"], cfg["_jitter_c"] = batch["jitter"][:R], batch["jitter_c"][:R]
        qp = dict(cfg.get("query_param", model.ray_query_cfg.get("query_param", {})))
        # the per-RAY arrays of the with-grad query (hit rays + the uniform eikonal points as zero-length rays) do not
        # depend on the sample count: they are queued BEFORE the sampling pass and its blocking size read
        x_uni = batch["x_uni"] if self.num_uniform > 0 else None
        M = int(x_uni.shape[0]) if x_uni is not None else 0
        o_r, d_r = tested["rays_o"].detach().float().contiguous(), tested["rays_d"].detach().float().contiguous()
        A_ = int(self.appear.shape[1])
        if M and "o_full" in batch:
            o, d, rz = batch["o_full"], batch["d_full"], batch["ridx_tail"]
            zc = getattr(self, "_tail_zeros", None)
            if zc is None or zc.shape[0] != M or zc.device != dev:
                zc = self._tail_zeros = torch.zeros([M], dtype=torch.float32, device=dev)
            tz = zc
            # the appearance codes of the hit rays + the zero rows of the appended free points: one launch (rows_gather)
            ha = torch.empty([R + M, A_], dtype=torch.float32, device=dev)
            _lib.call("nsim_rows_gather", _lib.ptr(self.appear.detach()), _lib.ptr(batch["fidx_hit"]), R, A_, self.V, M, _lib.ptr(ha))
        elif M:
            ha = self.appear.detach()[batch["fidx_hit"]]
            e = torch.empty([0], dtype=torch.float32, device=dev)
            o, d, tz, rz, ha = append_extra_points(model, o_r, d_r, e, e.long(), ha, x_uni)
        else:
            ha = self.appear.detach()[batch["fidx_hit"]]
            o, d = o_r, d_r
        # everything that does not depend on the sample count is queued BEFORE the sampling pass and its blocking size
        # read: the shadow / packed weights, and the zero-initialised buffers out of ONE arena (one memset, not nine)
        call, ptr = _lib.call, _lib.ptr
        f32 = dict(dtype=torch.float32, device=dev)
        grid16, wpack = model._shadow()
        fm, NLP = model.field_meta, model.plane_levels
        n_sdf_w, n_sdf_b, n_rad_w, n_rad_b = _flat_sizes(model.sdf_D, model.encoding.cfg.num_levels)
        every_ray = R == N                      # rays_inds sorted & unique: identity
        A = ha.shape[1]
        sizes = [4, 4, n_sdf_w, n_sdf_b, n_rad_w, n_rad_b, (R + M) * A, self.V * A,
                 0 if every_ray else 2 * N, 0 if every_ray else 6 * N]
        offs, tot = [], 0
        for n in sizes:
            offs.append(tot)
            tot += (n + 3) & ~3
        # (the table gradient and the arena are ONE zero fill: the gradient first -- its start stays 16-byte aligned)
        n_grid = model.encoding.flattened_params.numel()
        n_grid_pad = (n_grid + 3) & ~3
        zeros = torch.zeros([n_grid_pad + tot], **f32)
        dgrid, arena = zeros[:n_grid], zeros[n_grid_pad:]
        acc, dln, dsdf_w, dsdf_b, drad_w, drad_b, dha, d_app, sc0, vec0 = [
            arena[o_:o_ + n] for o_, n in zip(offs, sizes)]
        acc, dln = acc[:3], dln[:1]
        dha, d_app = dha.view(R + M, A), d_app.view(self.V, A)
        model._with_tail = None
        cfg["_tail_points"] = M     # the compressed mode's emit kernel appends the free points' samples itself
        spec = {}

        def spec_launch(t_full, ridx_full, pi_k, total_dev, cap_k):
            # queued BEFORE the host knows the size of the kept set: gather + decoders of the with-grad query at the
            # capacity, valid points = total_dev[0] + M read on the device
            Sc = cap_k + M
            PSc = _lib.plane_pitch(Sc)
            bufs = (torch.empty([Sc], **f32), torch.empty([Sc, 3], **f32), torch.empty([Sc, 3], **f32),
                    torch.empty([NLP, PSc, 2], **f32), torch.empty([NLP, PSc, 2, 3], dtype=_lib.jplane_dtype(fm), device=dev))
            # (the encoding's hook: LoTD = nsim_field_fwd with its own gather; permutohedral = nsim_permuto_gather + decoders)
            st = model._enc_field_fwd(grid16, wpack, None, o, d, t_full, ridx_full, None, ha, Sc, bufs[0], bufs[1], bufs[2], bufs[3],
                                      bufs[4], total_dev, M)
            spec.update(bufs=bufs, PS=PSc, enc_state=st)
        if self.spec_forward:
            cfg["_spec_launch"] = spec_launch
        _o, _d, t, pi, ridx, _sdf_ng, _mc, _goff, fis = model._query_samples(tested, cfg, qp)
        S = int(t.shape[0])
        if S == 0:
            return None
        if M and model._with_tail is not None:
            t_a, ridx_a = model._with_tail
        elif M:       # per-SAMPLE arrays: depth 0 on ray R + i for the i-th free point
            t_a, ridx_a = torch.cat([t, tz]), torch.cat([ridx, rz])
        else:
            t_a, ridx_a = t, ridx
        St = S + M
        z4 = torch.zeros([St * 4], **f32)       # d rgb / d sdf: the free points have no colour / alpha consumers
        drgb, dsdf = z4[:St * 3].view(St, 3), z4[St * 3:]
        # ---------------------------------------------------------------- forward
        if spec and getattr(model, "_spec_ok", False):      # already running (or done): views at the exact size
            sdf, nab, rgb = spec["bufs"][0][:St], spec["bufs"][1][:St], spec["bufs"][2][:St]
            h_pl, J_pl, PS = spec["bufs"][3], spec["bufs"][4], spec["PS"]
            enc_state = spec.get("enc_state")
        else:
            sdf, nab, rgb = torch.empty([St], **f32), torch.empty([St, 3], **f32), torch.empty([St, 3], **f32)
            PS = _lib.plane_pitch(St)
            h_pl, J_pl = torch.empty([NLP, PS, 2], **f32), torch.empty([NLP, PS, 2, 3], dtype=_lib.jplane_dtype(fm), device=dev)
            enc_state = model._enc_field_fwd(grid16, wpack, None, o, d, t_a, ridx_a, None, ha, St, sdf, nab, rgb, h_pl, J_pl, None, 0)
        ln_inv_s = model.ln_inv_s.detach()
        alpha, vw, trans = torch.empty([S], **f32), torch.empty([S], **f32), torch.empty([S], **f32)
        nd = int(bool(cfg.get("depth_use_normalized_vw", True)))
        if every_ray:
            sc, vec, out_idx = torch.empty([2, N], **f32), torch.empty([2, N, 3], **f32), None
        else:
            sc, vec, out_idx = sc0.view(2, N), vec0.view(2, N, 3), tested["rays_inds"]
        gt = batch["gt"]
        dalpha = torch.empty([S], **f32)
        if self.render_head:
            # sdf -> alpha -> visibility weights -> images -> photometric + eikonal losses -> their whole backward down to
            # d sdf / d rgb / d nablas / d ln_inv_s: ONE launch (csrc/pack_ops.hip k_render_head; four launches otherwise)
            dnab = torch.empty([St, 3], **f32)
            call("nsim_render_head", ptr(sdf), ptr(ln_inv_s), model.ln_inv_s_factor, fis, ptr(t), ptr(rgb), ptr(nab), ptr(pi), R, nd,
                 ptr(gt), N, S, M, float(self.w_eikonal), ptr(out_idx), ptr(alpha), ptr(vw), ptr(trans), ptr(sc[0]), ptr(sc[1]),
                 ptr(vec[0]), ptr(vec[1]), ptr(acc), ptr(dalpha), ptr(dsdf), ptr(drgb), ptr(dnab), ptr(dln))
        else:
            # sdf -> alpha -> visibility weights -> images: one launch
            call("nsim_neus_composite_fwd", ptr(sdf), ptr(ln_inv_s), model.ln_inv_s_factor, fis, ptr(t), ptr(rgb), ptr(nab),
                 ptr(pi), R, nd, ptr(alpha), ptr(vw), ptr(trans), ptr(sc[0]), ptr(sc[1]), ptr(vec[0]), ptr(vec[1]), ptr(out_idx))
            # the loss head in one launch: acc = (mse, eikonal(render samples), eikonal(uniform points)) and the gradients of
            # loss = mse + w (eik + eik) w.r.t. the image and the nablas (they do not depend on the loss values)
            d_img, dnab = torch.empty([N, 3], **f32), torch.empty([St, 3], **f32)
            call("nsim_train_loss_head", ptr(vec[0]), ptr(gt), N * 3, ptr(nab), S, M, float(self.w_eikonal), ptr(acc),
                 ptr(d_img), ptr(dnab))
            # ---------------------------------------------------------------- backward
            # (drgb / dsdf come zeroed: the free points have no colour / alpha consumers)
            call("nsim_composite_bwd", ptr(alpha), ptr(trans), ptr(vw), ptr(t), ptr(rgb), ptr(nab), ptr(pi), R, nd, ptr(sc[0]),
                 ptr(sc[1]), None, None, ptr(d_img), None, None, ptr(dalpha), ptr(drgb), None, ptr(out_idx))
            call("nsim_neus_alpha_bwd", ptr(sdf), ptr(dalpha), ptr(pi), R, ptr(ln_inv_s), model.ln_inv_s_factor, fis, ptr(dsdf),
                 ptr(dln))
        cst = getattr(self, "_fused_consts", None)
        if cst is None or cst[0] != (dev, float(self.w_eikonal)):
            cst = self._fused_consts = ((dev, float(self.w_eikonal)),
                                        torch.tensor([1.0, self.w_eikonal, self.w_eikonal], **f32))
        w_vec = cst[1]
        gn_total = torch.empty([St, 3], **f32)
        call("nsim_field_bwd_rad", fm, ptr(wpack), ptr(nab), ptr(rgb), None, ptr(o), ptr(d), ptr(t_a), ptr(ridx_a), ptr(ha),
             St, ptr(dnab), ptr(drgb), ptr(gn_total), ptr(drad_w), ptr(drad_b), ptr(dha), None, None)
        dh_pl, g_pl = torch.empty([NLP, St, 2], **f32), torch.empty([NLP, St, 2], **f32)
        call("nsim_field_bwd_sdf", fm, ptr(wpack), ptr(h_pl), ptr(J_pl), St, ptr(dsdf), ptr(gn_total), ptr(dh_pl),
             ptr(g_pl), ptr(dsdf_w), ptr(dsdf_b), None, PS)
        if model.sdf_scale != 1.0:      # d/d(head weights) of head / sdf_scale
            dsdf_w[-64:] /= model.sdf_scale
            dsdf_b[-1:] /= model.sdf_scale
        call("nsim_rows_scatter_add", ptr(dha), ptr(batch["fidx_hit"]), R, A, self.V, ptr(d_app))
        grid_p = model.encoding.flattened_params
        model.sdf_w.grad, model.sdf_b.grad, model.rad_w.grad, model.rad_b.grad = dsdf_w, dsdf_b, drad_w, drad_b
        if model.ln_inv_s.requires_grad:
            model.ln_inv_s.grad = dln
        self.appear.grad = d_app
        scatter_args = (model.encoding.cfg.meta, None, ptr(o), ptr(d), ptr(t_a), ptr(ridx_a), None, St, ptr(dh_pl),
                        ptr(g_pl), ptr(gn_total), ptr(dgrid))
        if type(model) is not LoTDNeuSModel:          # another encoding (permutohedral): its own scatter, in one piece
            model._enc_scatter(None, o, d, t_a, ridx_a, None, St, dh_pl, g_pl, gn_total, dgrid, enc_state=enc_state)
            if self.world_size > 1 and self.overlap_allreduce:
                grid_p.grad = dgrid
                self._dp_reduce_step(dgrid, None)
        elif self.world_size > 1 and self.overlap_allreduce:
            self._dp_reduce_step(dgrid, lambda l0, n: call("nsim_lotd_scatter", *scatter_args, l0, n))
        else:
            call("nsim_lotd_scatter", *scatter_args, 0, 0)
        grid_p.grad = dgrid
        if _lib.TIMER is not None:
            for k in ("nsim_field_fwd", "nsim_field_bwd_sdf", "nsim_lotd_scatter", "nsim_field_bwd_rad"):
                _lib.TIMER.note_units(k, St)
        # R_hit: rays that passed the AABB test; R_live: those whose occupancy march found something (the rays that get
        # coarse / fine samples under ``upsample_on_marched_only``); S_q: SDF-only queries of the sampling pass; S_f: with-grad samples
        lv = getattr(model, "_live", None)
        self.stats = dict(R_hit=R, R_live=int(lv["n"]) if (lv is not None and lv.get("n") is not None) else R, S_f=S,
                          S_q=int(getattr(model, "_last_S_q", 0) or 0))
        return torch.dot(acc, w_vec)

    def _backward_reducer(self):
        """The hook-driven gradient exchange of the autograd path (None: single rank, switched off, the measurement aid
        ``skip_allreduce``, or a configuration that runs the fused launch chain with its own two-half schedule)."""
        if self.world_size <= 1 or not self.overlap_allreduce or self.skip_allreduce or self._fused_ok():
            return None
        if getattr(self, "_reducer", None) is None:
            self._reducer = ndist.BackwardReducer(self.optim.params())
        return self._reducer

    def _dp_reduce_step(self, dgrid: torch.Tensor, scatter=None):
        """Data-parallel reduction + optimizer step with the hash-table gradient leaving in two halves: the all-reduce of
        the coarse half runs while the fine half is still being scattered (``scatter(level_begin, level_count)``), and
        Adam on the first half runs under the second all-reduce.  Collectives complete in issue order, so the small
        bucket (MLP weights, inv_s, appearance codes: ready before the scatter) goes first.  Every rank issues exactly
        this sequence whichever path produced its gradients (``scatter`` None: ``dgrid`` is complete already)."""
        grid_p = self.model.encoding.flattened_params
        if self.skip_allreduce:
            if scatter is not None:
                scatter(0, 0)
            grid_p.grad = dgrid
            self.optim.step(grad_scale=1.0)
            self._step_done = True
            return
        small = [q for q in self.optim.params() if q is not grid_p]
        flat = torch.cat([(q.grad if q.grad is not None else torch.zeros_like(q)).reshape(-1).float() for q in small])
        tok_small = ndist.allreduce_start(flat)
        wire, toks = ndist.wire_dtype_default(), []
        for l0, l1, o0, o1 in self._grid_halves():
            if scatter is not None:
                scatter(l0, l1 - l0)
            toks.append((ndist.allreduce_start(dgrid[o0:o1], wire), o0, o1))
        ndist.allreduce_finish(tok_small)
        off = 0
        for q in small:
            n = q.numel()
            q.grad = flat[off:off + n].view_as(q)
            off += n
        gs = 1.0 / self.world_size
        self.optim.step(grad_scale=gs, skip=(grid_p,))
        for tok, o0, o1 in toks:
            self.optim.step_range(grid_p, o0, o1, ndist.allreduce_finish(tok), grad_scale=gs)
        self._step_done = True

    def _grid_halves(self):
        """Two contiguous level ranges of about equal scatter cost (a hashed level ~1, a dense one ~0.35 -- the
        weights of the gather's level dealing): [(level_begin, level_end, param_begin, param_end)] * 2."""
        cfg = self.model.encoding.cfg
        if not hasattr(cfg, "lod_types"):       # levels of equal size (permutohedral: T entries each)
            L, per = cfg.num_levels, cfg.n_params // cfg.num_levels
            k = max(1, L // 2)
            return [(0, L, 0, cfg.n_params)] if L < 2 else [(0, k, 0, k * per), (k, L, k * per, cfg.n_params)]
        cost = [1.0 if t == "Hash" else 0.35 for t in cfg.lod_types]
        L, tot = cfg.num_levels, sum(cost)
        if L < 2:
            return [(0, L, 0, cfg.n_params)]
        run, k = 0.0, 1
        for i in range(L - 1):
            run += cost[i]
            k = i + 1
            if run >= 0.5 * tot:
                break
        offs = list(cfg.lod_offsets) + [cfg.n_params]
        return [(0, k, 0, offs[k]), (k, L, offs[k], cfg.n_params)]

    def _make_batch(self) -> dict:
        """sample_batch + ray generation + the AABB test (with its hit-ray compaction sync) of one batch, plus -- in
        ONE generator call -- the step's other uniforms: marching jitter [N], coarse-depth jitter [N, C] (the first R
        rows serve the R hit rays) and the uniform eikonal points [M, 3]."""
        xy, fidx, gt = self.sample_batch()
        dev, N, M = self.model.device, self.num_rays, self.num_uniform
        C = int(self.model.ray_query_cfg.get("query_param", {}).get("num_coarse", 64))
        rnd = torch.rand([N * (1 + C) + 3 * M], device=dev, generator=self.gen)
        lo, hi = self.model.accel.aabb[0], self.model.accel.aabb[1]
        extras = dict(jitter=rnd[:N], jitter_c=rnd[N:N * (1 + C)].view(N, C),
                      x_uni=torch.addcmul(lo, rnd[N * (1 + C):].view(M, 3), hi - lo) if M > 0 else None)
        if self._ray_cache is not None and self._ray_cache[0] is xy:
            _, rays_o, rays_d = self._ray_cache
        else:
            rays_o, rays_d = selected_rays(xy, fidx, self.intr, self.current_c2w(), self.WH, distortion=self.distortion)
        if rays_o.requires_grad:        # pose refinement: the compaction of the hit rays stays in the graph
            tested = self.model.ray_test(rays_o, rays_d, near=self.near, far=self.far)
        else:
            with torch.no_grad():
                tested = self.model.ray_test(rays_o, rays_d, near=self.near, far=self.far)
        if M > 0 and not rays_o.requires_grad:
            # per-RAY arrays of the with-grad query (hit rays + the uniform eikonal points as zero-length rays): they do
            # not depend on the sample count, so they are built here, off the step's critical path
            R = tested["num_rays"]
            ez = torch.zeros([M, 3], dtype=torch.float32, device=dev)
            ez[:, 2] = 1.0
            extras.update(o_full=torch.cat([tested["rays_o"], extras["x_uni"]]), d_full=torch.cat([tested["rays_d"], ez]),
                          ridx_tail=torch.arange(R, R + M, device=dev))
        return dict(xy=xy, fidx=fidx, gt=gt, rays_o=rays_o, rays_d=rays_d, tested=tested, aux=self._last_aux,
                    fidx_hit=fidx[tested["rays_inds"]], **extras)

    def _prefetch(self):
        if self._prefetched is not None:
            return
        if self._side is None:
            self._prefetched = self._make_batch()
            return
        main = torch.cuda.current_stream()
        # How the batch's tensors -- allocated on the side stream, consumed on the caller's -- are kept from being re-used too
        # early.  ``record_stream`` per tensor (the general form) makes the caching allocator record one event per block when the
        # batch dies: ~20 event records per step, and every 512 records the runtime's signal pool turns over with the host blocked
        # until the queue drains; every record is a marker packet on the stream, too (together 0.07 ms per step: 1.12 -> 1.05 ms,
        # profiles/round6_bench_instrumentation.txt).  A training step queues every consumer of its batch before it returns -- the
        # launch chain, or the autograd graph and its backward -- so the batch is handed over with ONE event instead: a consumed batch stays referenced
        # (``_retired``, with an event recorded on the caller's stream at the end of its step) until the side stream has been made
        # to wait for that event, and only then returns its blocks to the side stream's pool (NSIM_PREFETCH_HANDOFF=0:
        # record_stream everywhere).
        handoff = _PREFETCH_HANDOFF
        self._release_retired()
        with torch.cuda.stream(self._side):
            b = self._make_batch()
            b["_ready"] = self._side.record_event()
        if handoff:
            b["_handoff"] = True
        else:       # the tensors were allocated on the side stream and are consumed on the caller's
            for v in list(b.values()) + list(b["tested"].values()):
                if isinstance(v, torch.Tensor):
                    v.record_stream(main)
        self._prefetched = b

    def _release_retired(self):
        """Let go of a batch that was handed over by event (see ``_prefetch``): the side stream first waits for the event recorded on
        the caller's stream at the END of the step that consumed it -- behind every consumer of that batch, and long reached by the
        time the next batch is produced -- then the references go."""
        if getattr(self, "_retired", None) is not None and self._side is not None:
            self._side.wait_event(self._retired[1])
        self._retired = None

    def sample_uniform_x(self) -> torch.Tensor:
        lo, hi = self.model.accel.aabb[0], self.model.accel.aabb[1]
        return lo + torch.rand([self.num_uniform, 3], device=self.model.device, generator=self.gen) * (hi - lo)

    def loss(self, ret, gt, uni=None, aux=None):
        """photometric mse (or l1) on all rays + eikonal on the close-range render samples and on uniform points
        (+ the monocular depth / normal terms of the indoor config when ``mono`` supervision is set)."""
        if self.rgb_fn == "l1":
            loss_rgb = (ret["rendered"]["rgb_volume"] - gt).abs().mean()
        else:
            loss_rgb = mse_loss(ret["rendered"]["rgb_volume"], gt)
        if aux is not None and self.mono is not None:
            r_ = ret["rendered"]
            occupied = (r_["mask_volume"].detach() > 0.5).float()          # ``pred_not_occupied`` of the ignore list
            w_n, w_d = float(self.mono.get("w_normal", 0.05)), float(self.mono.get("w_depth", 0.1))
            loss_rgb = loss_rgb + w_n * mono_normal_loss(r_["normals_volume"], aux["normals"], occupied)
            if self.mono.get("patch_hw"):
                n_p = self.mono["patch_hw"][0] * self.mono["patch_hw"][1]
                loss_rgb = loss_rgb + w_d * mono_depth_loss(r_["depth_volume"][:n_p], aux["depth"][:n_p], occupied[:n_p])
        eik = None
        cr_vb = ret["raw_per_obj_model"]["main"]["volume_buffer"]
        if cr_vb["type"] != "empty":
            eik = eikonal_loss(cr_vb["nablas"])
        if self.num_uniform > 0:
            if uni is None:
                uni = ret["raw_per_obj_model"]["main"].get("extra_pts")      # rode on the render launches
            if uni is None:
                uni = self.model.sample_pts_uniform(self.num_uniform, generator=self.gen)
            e2 = eikonal_loss(uni["nablas"])
            eik = e2 if eik is None else eik + e2
        if eik is None:
            eik = torch.zeros([], device=gt.device)
        # (``torch.add(a, b, alpha=w)``: a + w b as ONE launch forward and one backward)
        return torch.add(loss_rgb, eik, alpha=self.w_eikonal), dict(loss_rgb=loss_rgb.detach(), loss_eikonal=eik.detach())

    # ------------------------------------------------------------------ lidar step (street configs)
    def sample_lidar_batch(self):
        L = self.lidar
        F_, M_ = L["ranges"].shape
        n = int(L.get("num_rays", 8192))
        dev = self.model.device
        idx = torch.randint(0, F_ * M_, [n], device=dev, generator=self.gen)
        return L["rays_o"].view(-1, 3)[idx], L["rays_d"].view(-1, 3)[idx], L["ranges"].view(-1)[idx]

    def lidar_losses(self, ret, ranges):
        """-> (loss, parts).  depth: ``l1_loss(depth_pred, ranges, mask, reduction='mean')`` (app/loss/lidar.py:41, 271);
        line of sight ``neus_unisim`` (:174-206): per ray the sum of vw^2 over the samples further than epsilon from the
        return, masked mean over the rays of the buffer."""
        from .graphics import pack_ops as po
        L = self.lidar
        valid = (ranges > 0) & (ranges < float(L.get("discard_toofar", 80.0)))
        m = valid.float()
        depth = ret["rendered"]["depth_volume"]
        parts = dict(depth=float(L.get("w_depth", 0.02)) * ((depth - ranges).abs() * m).mean())
        vb = ret["volume_buffer"]
        if vb["type"] != "empty":
            rih, pi = vb["rays_inds_hit"], vb["pack_infos_hit"]
            gt_ex = torch.repeat_interleave(ranges[rih], pi[:, 1], dim=0)
            far_off = ((vb["t"] - gt_ex).abs() > float(L.get("epsilon", 1.5))).float()
            empty = po.packed_sum(far_off * vb["vw"] ** 2, pi).reshape(-1)
            parts["los"] = float(L.get("w_los", 0.1)) * (empty * m[rih]).mean()
        return sum(parts.values()), parts

    def train_step_lidar(self, it: int) -> torch.Tensor:
        """The iteration's second render + backward + optimizer step (code_single/tools/train.py:860-960, 1540-1590):
        lidar beams rendered with ``with_rgb=False, with_normal=(eikonal loss configured)`` (:899-904), losses = lidar depth
        + line of sight (``lidar_losses``) + the eikonal term in mode 'lidar' -- on the render samples with weight
        ``w_eikonal * on_render_ratio`` (app/loss/eikonal.py:233-250; ``lidar['on_render_ratio']``, default 1) and on
        ``lidar['num_uniform']`` uniform points (:199-208, default 0).  Parameters outside this graph (radiance, appearance,
        sky) have no gradient on ANY rank: they are neither reduced nor stepped (``skip_absent``; FusedAdam keeps per-group
        step counts), as torch's Adam under the reference's DDP leaves them alone."""
        L = self.lidar
        o, d, ranges = self.sample_lidar_batch()
        w_eik = float(L.get("w_eikonal", self.w_eikonal))
        with_normal = w_eik > 0.0
        ret = self.renderer.render(self.model, rays=[o, d], rays_h_appear=None, with_rgb=False, with_normal=with_normal,
                                   return_buffer=True, return_details=with_normal, distant_model=self.distant_model,
                                   sky_model=None, near=float(L.get("near", self.near or 0.0)),
                                   far=float(L.get("far", self.far)) if (L.get("far", self.far) is not None) else None)
        loss, parts = self.lidar_losses(ret, ranges)
        if with_normal:
            cr_vb = ret["raw_per_obj_model"]["main"]["volume_buffer"]
            r_ratio = float(L.get("on_render_ratio", 1.0))
            if cr_vb["type"] != "empty" and r_ratio > 0.0 and "nablas" in cr_vb:
                parts["eikonal_render"] = w_eik * r_ratio * eikonal_loss(cr_vb["nablas"])
                loss = loss + parts["eikonal_render"]
            n_uni = int(L.get("num_uniform", 0))
            if n_uni > 0:
                uni = self.model.sample_pts_uniform(n_uni, generator=self.gen)
                parts["eikonal_uniform"] = w_eik * eikonal_loss(uni["nablas"])
                loss = loss + parts["eikonal_uniform"]
        self.optim.zero_grad()
        if loss.requires_grad:          # no beam hit anything and no distant model: nothing to differentiate on this rank
            with backward_on_calling_thread():
                loss.backward()
        if not self.skip_allreduce:
            ndist.allreduce_grads(self.optim.params(), average=False, skip_absent=True)
        self.optim.step(grad_scale=1.0 if self.skip_allreduce else 1.0 / self.world_size)
        self.stats["lidar_samples"] = int(ret["volume_buffer"]["t"].shape[0]) if ret["volume_buffer"]["type"] != "empty" else 0
        self._lidar_parts = {k: float(v.detach()) for k, v in parts.items()}
        return loss.detach()

    def train_step(self, it: int) -> torch.Tensor:
        loss = self._train_step_pixel(it)
        if self.lidar is not None:
            self._loss_lidar = self.train_step_lidar(it)
        return loss

    def _train_step_pixel(self, it: int) -> torch.Tensor:
        model = self.model
        self._it = int(it)
        refine = self.pose_refine_active()
        if refine and self._prefetched is not None:      # a batch drawn before the refinement started: re-draw with grad
            self._prefetched = None
        # training_before_per_step: occupancy refresh with a rank-shared seed keeps replicas consistent
        acc = model.accel
        if self.level_anneal is not None:
            model.anneal_levels(it, **self.level_anneal)
        # a model built from the reference's model_params drives its own schedules in the hook (level annealing,
        # occupancy refresh); it gets the rank-shared generator so that replicas refresh identically
        self_driven = getattr(model, "_reference_post", None) is not None
        if self_driven:
            model.refresh_generator = self.gen_shared
        model.training_before_per_step(it)          # inv_s control (var_ctrl_cfg); a no-op unless set_var_ctrl() was called
        if not self_driven and it >= acc.n_steps_warmup and it % acc.n_steps_between_update == 0:
            acc.update_from_net(model.query_sdf, generator=self.gen_shared)
        batch, handed = None, False
        if self.pipeline:
            batch, self._prefetched = (self._prefetched or self._make_batch()), None
            ev = batch.pop("_ready", None)
            if ev is not None:
                torch.cuda.current_stream().wait_event(ev)
            handed = batch.pop("_handoff", False)       # handed over by event instead of record_stream (see _prefetch)
            xy, fidx, gt = batch["xy"], batch["fidx"], batch["gt"]
        else:
            xy, fidx, gt = self.sample_batch()
        # the uniform eikonal points (train.py:602-613) ride on the render's field launches as zero-length rays: a
        # separate 4096-point launch chain costs ~0.2 ms of fixed latency (fwd + two backward kernels + scatter)
        if batch is not None:
            x_uni = batch["x_uni"]
        else:
            x_uni = self.sample_uniform_x() if self.num_uniform > 0 else None
        loss = None
        if batch is not None and self._fused_ok():
            self.optim.zero_grad()
            loss = self._train_render_fused(batch)
        if loss is None:
            _lib.arena_begin(model.device)       # zero-filled buffers of the autograd-path step out of one arena (_lib.zeros)
            ret = self.render(xy, fidx, extra_pts=x_uni, batch=batch)
            uni = None
            if x_uni is not None and "extra_pts" not in ret["raw_per_obj_model"]["main"]:     # no ray hit anything
                uni = model.forward_sdf_nablas(x_uni)
            loss, parts = self.loss(ret, gt, uni, aux=batch.get("aux") if batch is not None else self._last_aux)
            self.optim.zero_grad()
            if refine:
                self.pose_optim.zero_grad(set_to_none=True)
            red = self._backward_reducer()
            if red is not None:
                red.begin()
            with backward_on_calling_thread():
                loss.backward()
            _lib.arena_end()
            vb = ret["raw_per_obj_model"]["main"]["volume_buffer"]
            n_hit = int(vb["rays_inds_hit"].shape[0]) if vb["type"] != "empty" else 0
            self.stats = dict(R_hit=int(getattr(model, "_last_R_tested", n_hit)) if n_hit else 0, R_live=n_hit, S_f=int(vb["t"].shape[0]) if vb["type"] != "empty" else 0,
                              S_q=int(getattr(model, "_last_S_q", 0) or 0))
        if not self._step_done and self.world_size > 1 and self.overlap_allreduce and self._fused_ok():
            # this rank fell back to the autograd path (nothing hit): same collective sequence as its peers
            gp = model.encoding.flattened_params
            self._dp_reduce_step(gp.grad if gp.grad is not None else torch.zeros_like(gp))
        if self._step_done:         # the overlapped data-parallel path reduced and stepped already
            self._step_done = False
        elif self._backward_reducer() is not None:
            # autograd-path configurations (distant / sky / lidar models next to the main one): the table gradients left
            # during the backward, one model's exchange under the next model's scatter (ndist.BackwardReducer)
            self._backward_reducer().reduce_and_step(self.optim, 1.0 / self.world_size)
        else:
            # sum over ranks; the 1/world of the mean is folded into the fused Adam pass (no extra sweep over 48 MB)
            if not self.skip_allreduce:
                ndist.allreduce_grads(self.optim.params(), average=False)
            self.optim.step(grad_scale=1.0 if self.skip_allreduce else 1.0 / self.world_size)
        if refine:
            # every rank takes part in the collective, also one whose batch hit nothing (no graph to the poses)
            if self.pose_delta.grad is None:
                self.pose_delta.grad = torch.zeros_like(self.pose_delta)
            if self.world_size > 1 and not self.skip_allreduce:
                ndist.allreduce_grads([self.pose_delta], average=True, wire_dtype=torch.float32)
            self.pose_optim.step()
        if handed:      # every consumer of this batch is queued: it stays referenced until the side stream has been made to wait for them
            self._release_retired()
            self._retired = (batch, torch.cuda.current_stream().record_event())
        return loss.detach()
