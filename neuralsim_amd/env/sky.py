"""``SimpleSky`` -- the directional sky MLP of the street configs.

Mirror of app/models/env/sky.py:16-51 for the configuration the reference ships
(code_single/configs/waymo/streetsurf/withmask_withlidar_joint.240219.yaml:312-322: ``dir_embed_cfg{type: sinusoidal,
n_frequencies: 10}``, ``D: 2``, ``W: 256``, ``n_appear_embedding: 4``).  ``forward(v, h_appear=)`` is what
``SingleVolumeRenderer`` calls (single_volume_renderer.py:449-457) and is autograd-transparent for the weights and
``h_appear``; the arithmetic is csrc/sky.hip.
"""
import math
from typing import Optional

import torch
import torch.nn as nn

from .. import _lib
from ..model_base import ModelMixin


class _SkyFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, model, w, b, v, h_appear):
        N = v.shape[0]
        dev = v.device
        wpack = model._packed()
        need = any(ctx.needs_input_grad)
        Np = model.plane_pitch(N)
        planes = torch.empty([(96 + 512) * Np], dtype=torch.float32, device=dev) if need else None
        rgb = torch.empty([N, 3], dtype=torch.float32, device=dev)
        _lib.call("nsim_sky_fwd", model.meta, _lib.ptr(wpack), _lib.ptr(v), _lib.ptr(h_appear), N, _lib.ptr(rgb),
                  _lib.ptr(planes))
        ctx.model, ctx.N = model, N
        ctx.save_for_backward(planes, rgb, h_appear)     # (rgb is an OUTPUT: a plain ctx attribute would be a reference cycle)
        return rgb

    @staticmethod
    def backward(ctx, g):
        model, N = ctx.model, ctx.N
        planes, rgb, ha = ctx.saved_tensors
        dev = rgb.device
        Np = model.plane_pitch(N)
        scratch = torch.empty([(32 + 512) * Np], dtype=torch.float32, device=dev)
        nw, nb = model.w.numel(), model.b.numel()
        dw, db = torch.zeros([nw + nb], dtype=torch.float32, device=dev).split([nw, nb])
        dha = torch.empty_like(ha) if (ha is not None and ctx.needs_input_grad[4]) else None
        _lib.call("nsim_sky_bwd", model.meta, _lib.ptr(model._packed()), _lib.ptr(rgb), _lib.ptr(g.float().contiguous()),
                  N, _lib.ptr(planes), _lib.ptr(scratch), _lib.ptr(dw), _lib.ptr(db), _lib.ptr(dha))
        return None, dw, db, None, dha


class SimpleSky(ModelMixin, nn.Module):
    def __init__(self, dir_embed_cfg: Optional[dict] = None, D: int = 2, W: int = 256, n_appear_embedding: int = 4,
                 activation: str = "relu", output_activation: str = "sigmoid", precision: str = "fp16", seed: int = 42,
                 device=None, **unused):
        super().__init__()
        cfg = dict(dir_embed_cfg or dict(type="sinusoidal", n_frequencies=10))
        if cfg.get("type", "sinusoidal") != "sinusoidal" or D != 2 or W != 256 or activation != "relu" \
                or output_activation != "sigmoid":
            raise NotImplementedError("SimpleSky: the HIP path covers the shipped configuration "
                                      "(sinusoidal embedding, D=2, W=256, relu, sigmoid)")
        self.n_frequencies = int(cfg.get("n_frequencies", 10))
        self.n_appear = int(n_appear_embedding)
        self.in_dim = 3 + 6 * self.n_frequencies + self.n_appear
        if self.in_dim > 96:
            raise ValueError("SimpleSky: 3 + 6 n_frequencies + n_appear_embedding must be <= 96")
        self.meta = _lib.SkyMeta(self.n_frequencies, self.n_appear, {"fp16": 0, "f32": 1}[precision])
        g = torch.Generator().manual_seed(seed)

        def lin(o, i):      # torch.nn.Linear's default init
            bnd = 1.0 / math.sqrt(i)
            return (torch.rand(o, i, generator=g) * 2 - 1) * bnd, (torch.rand(o, generator=g) * 2 - 1) * bnd
        w1, b1 = lin(W, self.in_dim)
        w2, b2 = lin(W, W)
        w3, b3 = lin(3, W)
        self.w = nn.Parameter(torch.cat([w1.reshape(-1), w2.reshape(-1), w3.reshape(-1)]))
        self.b = nn.Parameter(torch.cat([b1, b2, b3]))
        self._wpack = None
        self._wpack_versions = None
        if device is not None:
            self.to(device)

    @property
    def device(self):
        return self.w.device

    def _param_groups(self, cfg: dict):
        """``training_cfg{lr: skylr, ...}`` (withmask_withlidar_joint.240219.yaml:322-325)."""
        return [dict(name="decoder", params=[self.w, self.b])]

    def _after_optimizer_step(self):
        self._wpack_versions = None

    def _weight_reg_tensors(self):
        return [self.w]

    def set_precision(self, precision: str):
        self.meta.precision = {"fp16": 0, "f32": 1}[precision]
        self._wpack_versions = None

    @staticmethod
    def plane_pitch(N: int) -> int:
        return (N + 127) // 128 * 128

    def _packed(self):
        vers = (self.w._version, self.b._version, self.meta.precision, str(self.w.device))
        if self._wpack is None or self._wpack_versions != vers:
            nbytes = int(_lib.get_lib().nsim_sky_wpack_bytes(self.meta))
            if self._wpack is None or self._wpack.numel() != nbytes or self._wpack.device != self.w.device:
                self._wpack = torch.zeros([nbytes], dtype=torch.uint8, device=self.w.device)
            _lib.call("nsim_sky_pack_weights", self.meta, _lib.ptr(self.w.detach()), _lib.ptr(self.b.detach()),
                      _lib.ptr(self._wpack))
            self._wpack_versions = vers
        return self._wpack

    def forward(self, v: torch.Tensor, *, h_appear: torch.Tensor = None) -> torch.Tensor:
        """v [..., 3] unit view directions, h_appear [..., n_appear] -> rgb [..., 3]."""
        prefix = v.shape[:-1]
        vf = v.detach().float().reshape(-1, 3).contiguous()
        ha = None
        if self.n_appear > 0:
            assert h_appear is not None, "SimpleSky(n_appear_embedding>0) needs h_appear"
            ha = h_appear.float().reshape(-1, self.n_appear).contiguous()
        rgb = _SkyFn.apply(self, self.w, self.b, vf, ha)
        return rgb.reshape(*prefix, 3)
