"""``nr3d_lib.models.importance`` -- error-map importance sampling of training pixels
(code_single/tools/train.py:104-138, 678-690, 1059-1063; dataio/data_loader/pixel_loader.py:162-170, 282-303;
dataio/data_loader/sampler.py:199-212; config ``training.error_map{error_map_hw, frac_uniform, n_steps_max, ...}``,
lotd_neus.dtu.230814.yaml:316-320).  Implementation absent (nr3d_lib): restated from those call sites --

* ``ErrorMap(n_images, error_map_hw=, n_steps_max=, max_pdf=, min_pdf=, dtype=, device=)``: a coarse [n_images, h, w] running
  error per image; ``step_error_map(i=frame [N], xy=[N,2] in [0,1], val=[N])`` scatters a batch of per-pixel errors into
  it (mean per cell, exponential blend with the stored value); ``get_normalized_error_map(fi)`` for logging;
* ``ImpSampler({name: (ErrorMap, fraction)}, frac_uniform=)``: draws pixels -- ``frac_uniform`` of them uniformly, the rest
  from the 2-D pdfs of the maps (cell by its probability, uniform inside the cell); ``sample_pixel(n, frame)`` for a given
  frame, ``sample_img_pixel(n)`` jointly over (frame, pixel); ``get_pdf_image()`` = per-image probability mass.
"""
from typing import Dict, Tuple

import torch
import torch.nn as nn


class ErrorMap(nn.Module):
    def __init__(self, n_images: int, error_map_hw=(32, 32), n_steps_max: int = None, n_steps_init: int = 0,
                 min_pdf: float = 0.01, max_pdf: float = None, dtype=torch.float32, device=None, **unused):
        super().__init__()
        self.n_images, (self.h, self.w) = int(n_images), (int(error_map_hw[0]), int(error_map_hw[1]))
        self.n_steps_max = n_steps_max
        self.min_pdf, self.max_pdf = float(min_pdf), (float(max_pdf) if max_pdf is not None else None)
        self.register_buffer("error_map", torch.ones([self.n_images, self.h, self.w], dtype=dtype, device=device))
        self.register_buffer("n_steps", torch.full([self.n_images], int(n_steps_init), dtype=torch.long, device=device))

    @torch.no_grad()
    def step_error_map(self, i: torch.Tensor, xy: torch.Tensor, val: torch.Tensor):
        i = i.reshape(-1).long()
        xy = xy.reshape(-1, 2)
        val = val.detach().reshape(-1).to(self.error_map.dtype)
        if i.numel() == 1 and xy.shape[0] > 1:
            i = i.expand(xy.shape[0])
        cx = (xy[:, 0] * self.w).long().clamp_(0, self.w - 1)
        cy = (xy[:, 1] * self.h).long().clamp_(0, self.h - 1)
        flat = (i * self.h + cy) * self.w + cx
        em = self.error_map.view(-1)
        s = torch.zeros_like(em).index_add_(0, flat, val)
        c = torch.zeros_like(em).index_add_(0, flat, torch.ones_like(val))
        hit = c > 0
        em[hit] = 0.5 * em[hit] + 0.5 * (s[hit] / c[hit])          # running blend of the newest mean error per cell
        self.n_steps.index_add_(0, torch.unique(i), torch.ones_like(torch.unique(i)))

    def get_pdf(self, fi=None) -> torch.Tensor:
        """-> [n, h, w] (fi given) or [n_images, h, w] probabilities, each image's cells summing to 1."""
        em = self.error_map if fi is None else self.error_map[torch.as_tensor(fi).reshape(-1)]
        p = em.clamp_min(0) + 1e-12
        p = p / p.sum(dim=(-2, -1), keepdim=True)
        u = 1.0 / (self.h * self.w)
        p = p.clamp_min(self.min_pdf * u)
        if self.max_pdf is not None:
            p = p.clamp_max(max(self.max_pdf, 1.0) * u * self.h * self.w)
        return p / p.sum(dim=(-2, -1), keepdim=True)

    def get_normalized_error_map(self, fi) -> torch.Tensor:
        em = self.error_map[int(fi)]
        return em / em.max().clamp_min(1e-12)

    def get_pdf_image(self) -> torch.Tensor:
        m = self.error_map.sum(dim=(-2, -1))
        return m / m.sum().clamp_min(1e-12)


class ImpSampler(nn.Module):
    def __init__(self, error_maps: Dict[str, Tuple[ErrorMap, float]], frac_uniform: float = 0.5):
        super().__init__()
        self.error_maps = nn.ModuleDict({k: v[0] for k, v in error_maps.items()})
        self.fracs = {k: float(v[1]) for k, v in error_maps.items()}
        self.frac_uniform = float(frac_uniform)
        first = next(iter(self.error_maps.values()))
        self.n_images, self.h, self.w = first.n_images, first.h, first.w

    def _split(self, n: int):
        n_uni = int(round(n * self.frac_uniform))
        rest, tot = n - n_uni, sum(self.fracs.values()) or 1.0
        parts = {k: int(round(rest * f / tot)) for k, f in self.fracs.items()}
        first = next(iter(parts))
        parts[first] += rest - sum(parts.values())
        return n_uni, parts

    @staticmethod
    def _draw_cells(pdf_flat: torch.Tensor, n: int, w: int):
        idx = torch.multinomial(pdf_flat, n, replacement=True)
        jit = torch.rand([n, 2], device=pdf_flat.device)
        return idx, jit

    @torch.no_grad()
    def sample_pixel(self, num_samples: int, fi) -> torch.Tensor:
        """-> xy [num_samples, 2] in (0, 1) of frame ``fi``."""
        dev = next(iter(self.error_maps.values())).error_map.device
        n_uni, parts = self._split(num_samples)
        out = [torch.rand([n_uni, 2], device=dev)]
        for k, n in parts.items():
            if n <= 0:
                continue
            pdf = self.error_maps[k].get_pdf(fi)[0].reshape(-1)
            idx, jit = self._draw_cells(pdf, n, self.w)
            cy, cx = torch.div(idx, self.w, rounding_mode="floor"), idx % self.w
            out.append(torch.stack([(cx + jit[:, 0]) / self.w, (cy + jit[:, 1]) / self.h], dim=-1))
        return torch.cat(out).clamp_(1e-6, 1 - 1e-6)

    @torch.no_grad()
    def sample_img_pixel(self, num_samples: int):
        """-> (frame [num_samples] int64, xy [num_samples, 2]) drawn jointly over all images."""
        dev = next(iter(self.error_maps.values())).error_map.device
        n_uni, parts = self._split(num_samples)
        fis = [torch.randint(0, self.n_images, [n_uni], device=dev)]
        xys = [torch.rand([n_uni, 2], device=dev)]
        for k, n in parts.items():
            if n <= 0:
                continue
            em = self.error_maps[k]
            pdf = em.get_pdf() * em.get_pdf_image()[:, None, None]
            idx, jit = self._draw_cells(pdf.reshape(-1), n, self.w)
            fi = torch.div(idx, self.h * self.w, rounding_mode="floor")
            rem = idx % (self.h * self.w)
            cy, cx = torch.div(rem, self.w, rounding_mode="floor"), rem % self.w
            fis.append(fi)
            xys.append(torch.stack([(cx + jit[:, 0]) / self.w, (cy + jit[:, 1]) / self.h], dim=-1))
        return torch.cat(fis), torch.cat(xys).clamp_(1e-6, 1 - 1e-6)

    def get_pdf_image(self) -> torch.Tensor:
        return torch.stack([em.get_pdf_image() * self.fracs[k] for k, em in self.error_maps.items()]).sum(0)
