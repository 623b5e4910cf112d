"""``nr3d_lib.models.embedders.get_embedder`` (app/models/env/sky.py:11, 33): ``get_embedder(cfg, input_dim) ->
(module, out_dim)`` for ``identity`` / ``sinusoidal`` input encodings (``[x, sin(2^k x), cos(2^k x)]``, oracle/sky.py)."""
import torch
import torch.nn as nn


class _Identity(nn.Module):
    def forward(self, x):
        return x


class _Sinusoidal(nn.Module):
    def __init__(self, n_frequencies: int, include_input: bool = True):
        super().__init__()
        self.n_frequencies, self.include_input = int(n_frequencies), include_input

    def forward(self, x):
        outs = [x] if self.include_input else []
        for f in range(self.n_frequencies):
            outs += [torch.sin(x * float(2 ** f)), torch.cos(x * float(2 ** f))]
        return torch.cat(outs, dim=-1)


def get_embedder(embed_cfg, input_dim: int = 3):
    cfg = dict(embed_cfg or {"type": "identity"}) if not isinstance(embed_cfg, str) else {"type": embed_cfg}
    typ = cfg.get("type", "identity")
    if typ in ("identity", "none", None):
        return _Identity(), input_dim
    if typ in ("sinusoidal", "sinusoidal_legacy"):
        n = int(cfg.get("n_frequencies", 10))
        return _Sinusoidal(n), input_dim * (1 + 2 * n)
    raise NotImplementedError(f"embedder type {typ!r}")
