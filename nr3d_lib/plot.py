"""``nr3d_lib.plot`` -- the two colourisers the trainer's validation logging imports (code_single/tools/train.py:39)."""
import numpy as np
import torch


def color_depth(depth, scale=None, cmap="viridis", out="uint8,0,255"):
    """[H,W] depth -> [H,W,3] colours (a simple two-tone ramp; the logging images are not part of the hot path)."""
    d = depth.detach().float().cpu().numpy() if isinstance(depth, torch.Tensor) else np.asarray(depth, dtype=np.float32)
    s = float(scale) if scale is not None else float(max(d.max(), 1e-8))
    x = np.clip(d / s, 0.0, 1.0)
    img = np.stack([x, 1.0 - np.abs(2.0 * x - 1.0), 1.0 - x], axis=-1)
    if out.startswith("uint8"):
        return (img * 255.0).astype(np.uint8)
    return img.astype(np.float32)


def scene_flow_to_rgb(flow, flow_max_radius=None, background="dark"):
    f = flow.detach().float().cpu().numpy() if isinstance(flow, torch.Tensor) else np.asarray(flow, dtype=np.float32)
    r = float(flow_max_radius) if flow_max_radius else float(max(np.abs(f).max(), 1e-8))
    return np.clip(0.5 + 0.5 * f / r, 0.0, 1.0).astype(np.float32)


def _viz_unavailable(*a, **k):
    raise NotImplementedError("interactive visualisation (open3d / vedo) is outside the hot path")


create_camera_frustum_o3d = create_camera_frustum_vedo = vis_camera_mplot = vis_camera_o3d = vis_lidar_vedo = _viz_unavailable
vis_occgrid_voxels_o3d = draw_2dbox_on_im = draw_bool_mask_on_im = draw_patch_on_im = gallery = _viz_unavailable
get_n_ind_colors = get_n_ind_pallete = choose_opposite_color = _viz_unavailable
