"""oracle.sky -- CPU restatement of the directional sky MLP (``SimpleSky``).

TEST INFRASTRUCTURE (see oracle/__init__.py).  The module itself is present in the reference
(app/models/env/sky.py:16-51) but its two building blocks ``get_embedder`` / ``get_blocks`` live in the absent
``nr3d_lib`` -- PARITY UNPINNED for the embedding order.  Fixed here (and mirrored by csrc/sky.hip): the NeRF
positional encoding with the input included, frequency bands 2^0 .. 2^(F-1), per band sin then cos:
``[v, sin(2^0 v), cos(2^0 v), sin(2^1 v), cos(2^1 v), ...]`` (3 + 6F dims); blocks = Linear/ReLU x D, Linear, sigmoid
(``activation='relu', output_activation='sigmoid'``, sky.py:27); config F = 10, D = 2, W = 256, appearance 4
(code_single/configs/waymo/streetsurf/withmask_withlidar_joint.240219.yaml:312-322).
Blend at the call site: ``rgb = rgb_volume + (1 - mask_volume) * sky`` (app/renderers/single_volume_renderer.py:449-457).
"""
import math
from typing import List

import torch
import torch.nn.functional as F


def sinusoidal_embed(v: torch.Tensor, n_frequencies: int = 10) -> torch.Tensor:
    outs = [v]
    for f in range(n_frequencies):
        outs += [torch.sin(v * float(2 ** f)), torch.cos(v * float(2 ** f))]
    return torch.cat(outs, dim=-1)


def make_sky_params(n_frequencies=10, n_appear=4, W=256, seed=11):
    """-> (weights [W1 (W x IN), W2 (W x W), W3 (3 x W)], biases [W, W, 3]); torch.nn.Linear's default init."""
    g = torch.Generator().manual_seed(seed)
    IN = 3 + 6 * n_frequencies + n_appear

    def lin(o, i):
        bnd = 1.0 / math.sqrt(i)
        return (torch.rand(o, i, generator=g) * 2 - 1) * bnd, (torch.rand(o, generator=g) * 2 - 1) * bnd
    w1, b1 = lin(W, IN)
    w2, b2 = lin(W, W)
    w3, b3 = lin(3, W)
    return [w1, w2, w3], [b1, b2, b3]


def sky_forward(v: torch.Tensor, h_appear, ws: List[torch.Tensor], bs: List[torch.Tensor], n_frequencies: int = 10):
    x = sinusoidal_embed(v, n_frequencies)
    if h_appear is not None:
        x = torch.cat([x, h_appear], dim=-1)
    x = F.relu(F.linear(x, ws[0], bs[0]))
    x = F.relu(F.linear(x, ws[1], bs[1]))
    return torch.sigmoid(F.linear(x, ws[2], bs[2]))


def blend_sky(rgb_volume, mask_volume, rgb_sky):
    return rgb_volume + (1.0 - mask_volume[..., None]) * rgb_sky
