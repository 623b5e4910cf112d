"""Mirrors ``nr3d_lib.graphics.nerf`` (single_volume_renderer.py:19): alpha -> visibility weights."""
import torch

from .pack_ops import packed_alpha_to_vw, get_pack_infos_from_n

__all__ = ["packed_alpha_to_vw", "ray_alpha_to_vw"]


def ray_alpha_to_vw(alpha: torch.Tensor) -> torch.Tensor:
    """Batched [..., n] variant (single_volume_renderer.py:76-78): every row is one pack."""
    n = alpha.shape[-1]
    flat = alpha.reshape(-1, n)
    pi = get_pack_infos_from_n(torch.full([flat.shape[0]], n, dtype=torch.long, device=alpha.device))
    return packed_alpha_to_vw(flat.reshape(-1), pi).reshape(alpha.shape)
