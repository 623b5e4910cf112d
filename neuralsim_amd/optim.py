"""Fused Adam over the model's flat f32 master parameters (csrc/optim.hip).

Mirrors the reference's optimizer setup ``training_cfg{lr, eps 1e-15, betas [0.9, 0.99], invs_betas [0.9, 0.999]}``
(code_single/configs/object_centric/lotd_neus.dtu.230814.yaml:178-184; step at code_single/tools/train.py:1494-1502).
The LoTD table update also refreshes the fp16 shadow the gather kernels read, in the same pass.
"""
from typing import List, Optional

import torch

from . import _lib


class FusedAdam:
    def __init__(self, model, lr=1e-2, betas=(0.9, 0.99), invs_betas=(0.9, 0.999), eps=1e-15, learn_inv_s=True,
                 lazy_tables: bool = None):
        """``lazy_tables`` (default: env NSIM_LAZY_ADAM=1, else False): the hash TABLES are updated on the entries a step touched
        only -- an entry with a zero gradient keeps its moments and its value (torch.optim.SparseAdam's rule).
        Opt-in and off by default: the reference trains with a dense Adam, which moves an untouched entry by its decaying first
        moment, so the trajectory differs.  Decoder weights and scalars are always dense."""
        import os
        self.model = model
        self.lr, self.eps = lr, eps
        self.lazy_tables = (os.environ.get("NSIM_LAZY_ADAM", "0") == "1") if lazy_tables is None else bool(lazy_tables)
        self.groups = []
        enc = model.encoding
        enc.shadow()
        self.groups.append(dict(p=enc.flattened_params, p16=lambda: enc.params16, betas=betas))
        for p in (model.sdf_w, model.sdf_b, model.rad_w, model.rad_b):
            self.groups.append(dict(p=p, p16=None, betas=betas))
        if learn_inv_s:   # var_ctrl_cfg{ctrl_type: mix_linear} schedules inv_s instead of learning it (dtu yaml:85-91)
            self.groups.append(dict(p=model.ln_inv_s, p16=None, betas=invs_betas))
        for g in self.groups:
            g["m"] = torch.zeros_like(g["p"], dtype=torch.float32)
            g["v"] = torch.zeros_like(g["p"], dtype=torch.float32)
        self._dirty = [model]

    @property
    def t(self) -> int:
        """The largest per-group step count (what a single global counter used to be; logging / tests read it)."""
        return max((g.get("t", 0) for g in self.groups), default=0)

    def add_model(self, model, betas=(0.9, 0.99), invs_betas=(0.9, 0.999), learn_inv_s=True):
        """Parameters of a further NeuS model (a shared batched foreground model next to the background, code_multi)."""
        enc = model.encoding
        enc.shadow()
        new = [dict(p=enc.flattened_params, p16=lambda: enc.params16, betas=betas)]
        new += [dict(p=p, p16=None, betas=betas) for p in (model.sdf_w, model.sdf_b, model.rad_w, model.rad_b)]
        if learn_inv_s:
            new.append(dict(p=model.ln_inv_s, p16=None, betas=invs_betas))
        for g in new:
            g["m"] = torch.zeros_like(g["p"], dtype=torch.float32)
            g["v"] = torch.zeros_like(g["p"], dtype=torch.float32)
        self.groups += new
        self._dirty.append(model)

    def add_distant_model(self, dm, betas=(0.9, 0.99)):
        """Parameters of a ``LoTDNeRFDistantModel`` (``training_cfg{lr: bglr, betas [.9,.99]}``, dtu yaml :242-247)."""
        dm._shadow()
        for p, p16 in ((dm.flattened_params, lambda: dm.params16), (dm.den_w, None), (dm.den_b, None), (dm.rad_w, None),
                       (dm.rad_b, None)):
            self.groups.append(dict(p=p, p16=p16, betas=betas, m=torch.zeros_like(p, dtype=torch.float32),
                                    v=torch.zeros_like(p, dtype=torch.float32)))
        self._dirty.append(dm)

    def add_sky_model(self, sky, betas=(0.9, 0.99)):
        """Parameters of a ``SimpleSky`` (``training_cfg{lr: skylr}``, withmask_withlidar_joint.240219.yaml:322-325)."""
        for p in (sky.w, sky.b):
            self.groups.append(dict(p=p, p16=None, betas=betas, m=torch.zeros_like(p, dtype=torch.float32),
                                    v=torch.zeros_like(p, dtype=torch.float32)))
        self._dirty.append(sky)

    def params(self) -> List[torch.Tensor]:
        return [g["p"] for g in self.groups]

    @torch.no_grad()
    def step(self, lr: Optional[float] = None, grad_scale: float = 1.0, skip=()):
        """``skip``: parameters the caller updates itself in this iteration through ``step_range`` (AFTER this call:
        the step counter of their bias corrections advances here).
        Every group keeps its OWN step count, advanced only when the group is updated -- ``torch.optim.Adam`` keeps
        ``state['step']`` per parameter and skips parameters whose ``.grad`` is None (the reference's second backward +
        optimizer step of an iteration, the lidar step, leaves the radiance / appearance / sky parameters without a
        gradient: code_single/tools/train.py:1540-1590)."""
        lr = self.lr if lr is None else lr
        small = []
        for g in self.groups:
            p = g["p"]
            if any(p is q for q in skip):
                g["t"] = g.get("t", 0) + 1
                continue
            if p.grad is None:
                continue
            t = g["t"] = g.get("t", 0) + 1
            b1, b2 = g["betas"]
            p16 = g["p16"]() if g["p16"] is not None else None
            if p.numel() < (1 << 18) and p16 is None:     # batched below: one launch for all of them
                small.append((g, p.grad.contiguous()))
                continue
            _lib.call("nsim_adam_step", _lib.ptr(p.data), _lib.ptr(p16), _lib.ptr(p.grad.contiguous()), _lib.ptr(g["m"]),
                      _lib.ptr(g["v"]), p.numel(), float(lr), float(b1), float(b2), float(self.eps),
                      1.0 - b1 ** t, 1.0 - b2 ** t, float(grad_scale), 2 if (self.lazy_tables and p16 is not None) else 0)
        for k in range(0, len(small), _lib.ADAM_MULTI_MAX):
            chunk = small[k:k + _lib.ADAM_MULTI_MAX]
            arr = (_lib.AdamTensor * len(chunk))()
            for i, (g, grad) in enumerate(chunk):
                b1, b2 = g["betas"]
                arr[i] = _lib.AdamTensor(g["p"].data_ptr(), None, grad.data_ptr(), g["m"].data_ptr(), g["v"].data_ptr(),
                                         g["p"].numel(), b1, b2, 1.0 - b1 ** g["t"], 1.0 - b2 ** g["t"], 1.0)
            _lib.call("nsim_adam_multi", arr, len(chunk), float(lr), float(self.eps), float(grad_scale), 0)
        for m in self._dirty:
            m._wpack_versions = None           # MLP weights changed in place: re-pack the MFMA fragments lazily

    @torch.no_grad()
    def step_range(self, p: torch.Tensor, lo: int, hi: int, grad: torch.Tensor, lr: Optional[float] = None,
                   grad_scale: float = 1.0):
        """Adam on the flat slice p[lo:hi] with gradient ``grad`` [hi-lo] (f32), using the bias corrections of the
        iteration ``step`` has just opened -- lets a data-parallel caller update one half of the hash table while the
        all-reduce of the other half is still in flight."""
        lr = self.lr if lr is None else lr
        g = next(g for g in self.groups if g["p"] is p)
        b1, b2 = g["betas"]
        t = max(int(g.get("t", 0)), 1)
        p16 = g["p16"]() if g["p16"] is not None else None
        _lib.call("nsim_adam_step", _lib.ptr(p.data.view(-1)[lo:hi]), _lib.ptr(p16.view(-1)[lo:hi] if p16 is not None else None),
                  _lib.ptr(grad.contiguous()), _lib.ptr(g["m"].view(-1)[lo:hi]), _lib.ptr(g["v"].view(-1)[lo:hi]), hi - lo,
                  float(lr), float(b1), float(b2), float(self.eps), 1.0 - b1 ** t, 1.0 - b2 ** t,
                  float(grad_scale), 2 if (self.lazy_tables and p16 is not None) else 0)

    def zero_grad(self):
        for g in self.groups:
            g["p"].grad = None
