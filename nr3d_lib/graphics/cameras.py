"""``nr3d_lib.graphics.cameras`` -- pinhole helpers the reference's Camera node imports (cameras.py:20-24, 125-199, 223-226).
Restated from the call sites; (u, v) are pixel coordinates, d the depth along the optical axis (OpenCV convention)."""
import torch


def pinhole_lift(u: torch.Tensor, v: torch.Tensor, d: torch.Tensor, intr: torch.Tensor) -> torch.Tensor:
    """pixel (u, v) at depth d -> camera-frame point [..., 3] for intrinsics [..., 3, 3] (with skew)."""
    fx, fy, cx, cy, sk = intr[..., 0, 0], intr[..., 1, 1], intr[..., 0, 2], intr[..., 1, 2], intr[..., 0, 1]
    y = (v - cy) / fy
    x = (u - cx - sk * y) / fx
    return torch.stack([x * d, y * d, d], dim=-1)


def pinhole_view_frustum(c2w: torch.Tensor, intr: torch.Tensor, H, W, near=None, far=None) -> torch.Tensor:
    """Inward-facing boundary planes [..., P, 4] = (unit normal, offset): a point x is inside iff n . x + o >= 0 for every
    plane.  P = 4 side planes (+ near, + far when given)."""
    H = torch.as_tensor(H, dtype=c2w.dtype, device=c2w.device)
    W = torch.as_tensor(W, dtype=c2w.dtype, device=c2w.device)
    z0, one = torch.zeros_like(H), torch.ones_like(H)
    corners_uv = [(z0, z0), (W, z0), (W, H), (z0, H)]
    dirs = torch.stack([pinhole_lift(u, v, one, intr) for u, v in corners_uv], dim=-2)          # [..., 4, 3] camera frame
    R, t = c2w[..., :3, :3], c2w[..., :3, 3]
    dirs_w = (R.unsqueeze(-3) * dirs.unsqueeze(-2)).sum(-1)                                      # [..., 4, 3]
    normals = torch.linalg.cross(dirs_w, torch.roll(dirs_w, shifts=-1, dims=-2), dim=-1)       # side planes through the eye
    normals = torch.nn.functional.normalize(normals, dim=-1)
    fwd = R[..., :, 2]
    # orient inwards (towards the optical axis)
    sign = torch.sign((normals * fwd.unsqueeze(-2)).sum(-1, keepdim=True))
    sign = torch.where(sign == 0, torch.ones_like(sign), sign)
    normals = normals * sign
    planes = [torch.cat([normals, -(normals * t.unsqueeze(-2)).sum(-1, keepdim=True)], dim=-1)]
    if near is not None:
        planes.append(torch.cat([fwd, -((fwd * t).sum(-1, keepdim=True) + float(near))], dim=-1).unsqueeze(-2))
    if far is not None:
        planes.append(torch.cat([-fwd, ((fwd * t).sum(-1, keepdim=True) + float(far))], dim=-1).unsqueeze(-2))
    return torch.cat(planes, dim=-2)


def sphere_inside_frustum(sphere_center_radius: torch.Tensor, planes: torch.Tensor, holistic: bool = False) -> torch.Tensor:
    """[..., S, 4] spheres vs [..., P, 4] planes -> [..., S] bool: any part (or, ``holistic``, the whole body) inside."""
    c, r = sphere_center_radius[..., :3], sphere_center_radius[..., 3]
    dist = (planes[..., None, :, :3] * c[..., :, None, :]).sum(-1) + planes[..., None, :, 3]     # [..., S, P]
    margin = -r[..., None] if not holistic else r[..., None]
    return (dist >= margin).all(dim=-1)


def pinhole_get_rays(c2w, intr, H, W, N_rays=-1):
    raise NotImplementedError("use neuralsim_amd.graphics.cameras.pinhole_selected_rays")
