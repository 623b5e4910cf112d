"""Ray generation -- mirrors ``Camera.get_selected_rays`` / ``_get_selected_rays_from_ixy``
(app/resources/observers/cameras.py:281-330) on top of csrc/sampling.hip::k_raygen_pinhole, for the pinhole and the
OpenCV (radial-tangential distortion) and the fisheye camera models (``camera_model: pinhole | opencv | fisheye``,
cameras.py:80-92)."""
import torch

from .. import _lib


class _RaygenFn(torch.autograd.Function):
    """Ray generation with a backward w.r.t. the camera poses (pose refinement: the reference's LearnableParams hand
    refined c2w matrices to ``Camera.get_selected_rays``; pixels and intrinsics are constants)."""

    @staticmethod
    def forward(ctx, c2w, xy, fidx, intr, WH, snap, dist=None, n_iters=0, model="opencv"):
        N = xy.shape[0]
        o = torch.empty([N, 3], dtype=torch.float32, device=xy.device)
        d = torch.empty([N, 3], dtype=torch.float32, device=xy.device)
        c2w = c2w.float().contiguous()
        if dist is None:
            _lib.call("nsim_raygen_pinhole", _lib.ptr(xy), _lib.ptr(fidx), _lib.ptr(intr), _lib.ptr(c2w), _lib.ptr(WH), N,
                      snap, _lib.ptr(o), _lib.ptr(d))
        else:
            _lib.call("nsim_raygen_" + model, _lib.ptr(xy), _lib.ptr(fidx), _lib.ptr(intr), _lib.ptr(dist), int(n_iters),
                      _lib.ptr(c2w), _lib.ptr(WH), N, snap, _lib.ptr(o), _lib.ptr(d))
        ctx.save_for_backward(c2w, xy, fidx, intr, WH, dist)
        ctx.snap, ctx.n_iters, ctx.cam_model = snap, int(n_iters), model
        return o, d

    @staticmethod
    def backward(ctx, g_o, g_d):
        c2w, xy, fidx, intr, WH, dist = ctx.saved_tensors
        d_c2w = torch.zeros_like(c2w)
        g_o = g_o.float().contiguous() if g_o is not None else None
        g_d = g_d.float().contiguous() if g_d is not None else None
        if dist is None:
            _lib.call("nsim_raygen_pinhole_bwd", _lib.ptr(xy), _lib.ptr(fidx), _lib.ptr(intr), _lib.ptr(c2w), _lib.ptr(WH),
                      xy.shape[0], ctx.snap, _lib.ptr(g_o), _lib.ptr(g_d), _lib.ptr(d_c2w))
        else:
            _lib.call("nsim_raygen_" + ctx.cam_model + "_bwd", _lib.ptr(xy), _lib.ptr(fidx), _lib.ptr(intr), _lib.ptr(dist),
                      ctx.n_iters, _lib.ptr(c2w), _lib.ptr(WH), xy.shape[0], ctx.snap, _lib.ptr(g_o), _lib.ptr(g_d),
                      _lib.ptr(d_c2w))
        return d_c2w, None, None, None, None, None, None, None, None


def pinhole_selected_rays(xy: torch.Tensor, fidx: torch.Tensor, intr: torch.Tensor, c2w: torch.Tensor,
                          WH: torch.Tensor, snap_to_pixel_centers: bool = True):
    """xy [N,2] in [0,1], fidx [N] int64, intr [V,3,3], c2w [V,4,4] (OpenCV), WH [V,2] int64 -> rays_o, rays_d [N,3].
    Differentiable w.r.t. ``c2w`` (pose refinement); everything else is a constant of the step."""
    return _RaygenFn.apply(c2w, xy.detach().float().contiguous(), fidx.long().contiguous(),
                           intr.detach().float().contiguous(), WH.long().contiguous(),
                           1 if snap_to_pixel_centers else 0)


def opencv_selected_rays(xy: torch.Tensor, fidx: torch.Tensor, intr: torch.Tensor, distortion: torch.Tensor,
                         c2w: torch.Tensor, WH: torch.Tensor, snap_to_pixel_centers: bool = True, n_iters: int = 5):
    """``pinhole_selected_rays`` for ``camera_model: opencv`` (cameras.py:84-87; Waymo's calibration in the street
    configs): distortion [V,5] = (k1, k2, p1, p2, k3); the lift undistorts with ``n_iters`` rounds of the fixed-point
    iteration of cv::undistortPoints (5 = OpenCV's own count).  Differentiable w.r.t. ``c2w``."""
    dist = distortion.detach().float().contiguous()
    if dist.dim() != 2 or dist.shape[1] != 5 or dist.shape[0] != intr.shape[0]:
        raise ValueError(f"distortion must be [V,5] = (k1, k2, p1, p2, k3) per frame, got {tuple(dist.shape)}")
    return _RaygenFn.apply(c2w, xy.detach().float().contiguous(), fidx.long().contiguous(),
                           intr.detach().float().contiguous(), WH.long().contiguous(),
                           1 if snap_to_pixel_centers else 0, dist, int(n_iters))


def fisheye_selected_rays(xy: torch.Tensor, fidx: torch.Tensor, intr: torch.Tensor, distortion: torch.Tensor,
                          c2w: torch.Tensor, WH: torch.Tensor, snap_to_pixel_centers: bool = True, n_iters: int = 10):
    """``pinhole_selected_rays`` for ``camera_model: fisheye`` (cameras.py:88-92): distortion [V,4] = (k1, k2, k3, k4) of the
    OpenCV fisheye model (app/resources/observers/fisheye.py:36-42); the lift solves theta_d = theta (1 + k1 theta^2 + ...)
    with ``n_iters`` Newton rounds (10 = cv::fisheye::undistortPoints).  Differentiable w.r.t. ``c2w``."""
    dist = distortion.detach().float().contiguous()
    if dist.dim() != 2 or dist.shape[1] != 4 or dist.shape[0] != intr.shape[0]:
        raise ValueError(f"fisheye distortion must be [V,4] = (k1, k2, k3, k4) per frame, got {tuple(dist.shape)}")
    return _RaygenFn.apply(c2w, xy.detach().float().contiguous(), fidx.long().contiguous(),
                           intr.detach().float().contiguous(), WH.long().contiguous(),
                           1 if snap_to_pixel_centers else 0, dist, int(n_iters), "fisheye")


def selected_rays(xy, fidx, intr, c2w, WH, distortion=None, camera_model: str = None, **kw):
    """``pinhole_selected_rays``; with ``distortion`` [V,5] ``opencv_selected_rays``; with ``distortion`` [V,4] (or
    ``camera_model='fisheye'``) ``fisheye_selected_rays`` (``camera_model``, cameras.py:80-92)."""
    if distortion is None:
        return pinhole_selected_rays(xy, fidx, intr, c2w, WH, **kw)
    if camera_model == "fisheye" or (camera_model is None and distortion.shape[-1] == 4):
        return fisheye_selected_rays(xy, fidx, intr, distortion, c2w, WH, **kw)
    return opencv_selected_rays(xy, fidx, intr, distortion, c2w, WH, **kw)


def look_at_cameras(V=100, radius=3.0, H=800, W=800, f=1111.1, seed=42, device=None):
    """Synthetic posed-camera rig of SURVEY.md sec. 8d: V pinhole views (OpenCV convention, +z forward, +y down)
    on a sphere of ``radius`` around the AABB, looking at the origin."""
    import math
    g = torch.Generator().manual_seed(seed)
    c2w = torch.eye(4).repeat(V, 1, 1)
    for i in range(V):
        u = torch.rand(2, generator=g)
        th, ph = float(2 * math.pi * u[0]), float(math.acos(1 - 2 * (0.15 + 0.7 * float(u[1]))))
        eye = radius * torch.tensor([math.sin(ph) * math.cos(th), math.cos(ph), math.sin(ph) * math.sin(th)])
        fwd = -eye / eye.norm()
        right = torch.linalg.cross(fwd, torch.tensor([0.0, -1.0, 0.0]))
        right = right / right.norm()
        down = torch.linalg.cross(fwd, right)
        c2w[i, :3, 0], c2w[i, :3, 1], c2w[i, :3, 2], c2w[i, :3, 3] = right, down, fwd, eye
    intr = torch.tensor([[f, 0, W / 2], [0, f, H / 2], [0, 0, 1.0]]).repeat(V, 1, 1)
    WH = torch.tensor([[W, H]], dtype=torch.long).repeat(V, 1)
    if device is not None:
        intr, c2w, WH = intr.to(device), c2w.to(device), WH.to(device)
    return intr, c2w, WH
