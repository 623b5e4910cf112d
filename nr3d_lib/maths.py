"""``nr3d_lib.maths`` -- the few closed-form helpers the scene graph and the models import (app/resources/nodes.py:18,
app/models/single/neus.py, app/resources/scenes.py)."""
import torch
import numpy as np


def inverse_transform_matrix(m: torch.Tensor) -> torch.Tensor:
    """Inverse of rigid [..., 4, 4] transforms: [R | t] -> [R^T | -R^T t] (no general matrix inverse; broadcast-multiply-sum
    as the reference demands for pose arithmetic, app/resources/nodes.py:79-84)."""
    R, t = m[..., :3, :3], m[..., :3, 3]
    Rt = R.transpose(-1, -2)
    ti = -(Rt * t.unsqueeze(-2)).sum(-1)
    top = torch.cat([Rt, ti.unsqueeze(-1)], dim=-1)
    bottom = torch.zeros_like(m[..., 3:4, :])
    bottom[..., 0, 3] = 1.0
    return torch.cat([top, bottom], dim=-2)


def inverse_transform_matrix_np(m: np.ndarray) -> np.ndarray:
    return inverse_transform_matrix(torch.from_numpy(np.asarray(m))).numpy()


def geometric_mean(x, dim=None):
    """exp(mean(log x)) of positive numbers (a list / tensor; used for isotropic length scales of AABBs)."""
    t = torch.as_tensor(x, dtype=torch.float32)
    g = torch.exp(torch.log(t).mean() if dim is None else torch.log(t).mean(dim=dim))
    return float(g) if g.dim() == 0 and not isinstance(x, torch.Tensor) else g


def normalized_logistic_density(x: torch.Tensor, inv_s) -> torch.Tensor:
    """4 sigma(s x) (1 - sigma(s x)): the logistic density normalised to 1 at x = 0 (the occupancy value function
    ``occ_val_fn_cfg{type: sdf, inv_s}``: 0.27 at |sdf| = 0.01 for inv_s 256)."""
    sg = torch.sigmoid(x * inv_s)
    return 4.0 * sg * (1.0 - sg)


def normalize(v: torch.Tensor, dim: int = -1, eps: float = 1e-12) -> torch.Tensor:
    return torch.nn.functional.normalize(v, dim=dim, eps=eps)


def get_transform_np(rot=None, trans=None):
    m = np.eye(4, dtype=np.float32)
    if rot is not None:
        m[:3, :3] = rot
    if trans is not None:
        m[:3, 3] = trans
    return m
