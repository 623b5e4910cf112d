"""``nr3d_lib.models.embeddings`` -- per-frame learnable codes (app/models/scene/image_embeddings.py:14, 76:
``SeqEmbedding(ts_keyframes, v_keyframes=, dim=, dtype=, device=)``, queried as ``emb(rays_ts, mode='interp')``,
app/renderers/single_volume_renderer.py:173).  Implementation absent (nr3d_lib): restated from those call sites."""
import torch
import torch.nn as nn


class Embedding(nn.Embedding):
    """``Embedding(num_objs, dim=, weight_init=, dtype=, device=)`` -- the auto-decoder's latent table
    (app/models/shared/batched_neus.py:108: ``Embedding(self.num_objs, **self.latents_cfg['z_ins'], dtype=torch.float,
    device=device)`` with ``latents_cfg{z_ins{dim, weight_init: zero | normal | uniform}}``, all_occ.240201.yaml:428-431)."""

    def __init__(self, num_embeddings: int, dim: int = None, weight_init=None, embedding_dim: int = None, dtype=torch.float32,
                 device=None, std: float = None, **unused):
        d = int(dim if dim is not None else embedding_dim)
        super().__init__(int(num_embeddings), d, device=device, dtype=dtype if isinstance(dtype, torch.dtype) else torch.float32)
        wi = weight_init if isinstance(weight_init, (str, type(None))) else dict(weight_init).get("type")
        with torch.no_grad():
            if wi in ("zero", "zeros"):
                self.weight.zero_()
            elif wi in (None, "normal"):
                self.weight.normal_(0.0, float(std) if std is not None else 1.0 / (d ** 0.5))
            elif wi == "uniform":
                self.weight.uniform_(-1.0 / (d ** 0.5), 1.0 / (d ** 0.5))
            else:
                raise NotImplementedError(f"Embedding.weight_init={weight_init!r}: zero | normal | uniform")


class SeqEmbedding(nn.Module):
    """One code per keyframe timestamp; ``forward(ts, mode)``: 'interp' = piecewise-linear in time (exact at keyframes),
    'nearest' = the closest keyframe.  Gradients reach the two neighbouring codes through a plain index + lerp."""

    def __init__(self, ts_keyframes: torch.Tensor, v_keyframes: torch.Tensor = None, dim: int = None, dtype=torch.float32,
                 device=None, learnable: bool = True, **unused):
        super().__init__()
        ts = torch.as_tensor(ts_keyframes, dtype=torch.float32).detach().clone()
        self.register_buffer("ts_keyframes", ts.to(device) if device is not None else ts, persistent=True)
        if v_keyframes is None:
            v_keyframes = torch.zeros(ts.shape[0], int(dim))
        v = torch.as_tensor(v_keyframes).to(dtype=torch.float32)        # f32 master (the kernels read f32 codes)
        self.weight = nn.Parameter(v.to(device) if device is not None else v, requires_grad=learnable)
        self.dim = int(self.weight.shape[-1])

    def forward(self, ts: torch.Tensor, mode: str = "interp") -> torch.Tensor:
        tk = self.ts_keyframes
        T = tk.shape[0]
        shape = ts.shape
        t = ts.reshape(-1).to(tk.dtype)
        if T == 1:
            return self.weight[torch.zeros_like(t, dtype=torch.long)].reshape(*shape, self.dim)
        idx = torch.searchsorted(tk.contiguous(), t.contiguous(), right=True).clamp(1, T - 1)
        t0, t1 = tk[idx - 1], tk[idx]
        w = ((t - t0) / (t1 - t0).clamp_min(1e-12)).clamp(0, 1)
        if mode == "nearest":
            out = self.weight[torch.where(w < 0.5, idx - 1, idx)]
        else:
            out = torch.lerp(self.weight[idx - 1], self.weight[idx], w[:, None])
        return out.reshape(*shape, self.dim)


MultiSeqEmbeddingIndividual = MultiSeqEmbeddingShared = SeqEmbedding
