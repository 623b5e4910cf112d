"""A dataset for the reference's trainer without files: ``dataset_cfg.target: neuralsim_amd.dataio.SyntheticObjectDataset``.

Implements the ``SceneDataset`` seam of the reference (dataio/scene_dataset.py:13-74; single-object scenario layout as
dataio/dtu/dtu_dataset.py:144-171): ``get_scenario(scene_id)`` describes one object node (class
``Main``), optionally a ``Distant`` node, and one pinhole camera observer with per-frame intrinsics / c2w (OpenCV
convention); ``get_image`` / ``get_image_occupancy_mask`` render the analytic world of ``neuralsim_amd.scenarios`` --
a sphere seen by the posed-camera rig of SURVEY sec. 8d -- so the reference's ``code_single/tools/train.py`` runs end to
end on this repository's kernels with no data on disk."""
from typing import Any, Dict, List

import numpy as np
import torch

from ..scenarios import AnalyticWorld


def _pixel_rays(intr: torch.Tensor, c2w: torch.Tensor, H: int, W: int):
    """World rays through the pixel centres of one pinhole image (host side: this is DATASET synthesis -- the images a
    real dataset would read from disk -- not the render path, whose rays come from csrc/sampling.hip)."""
    j, i = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(W, dtype=torch.float32), indexing="ij")
    x = (i.reshape(-1) + 0.5 - intr[0, 2]) / intr[0, 0]
    y = (j.reshape(-1) + 0.5 - intr[1, 2]) / intr[1, 1]
    dc = torch.stack([x, y, torch.ones_like(x)], dim=-1)
    d = torch.nn.functional.normalize((c2w[:3, :3] * dc.unsqueeze(-2)).sum(-1), dim=-1)
    return c2w[:3, 3].expand_as(d).contiguous(), d


class SyntheticObjectDataset:
    def __init__(self, config: dict = None):
        cfg = dict(config or {})
        self.V = int(cfg.get("n_frames", 24))
        self.H = self.W = int(cfg.get("image_hw", 128))
        self.focal = float(cfg.get("focal_ratio", 1.3889)) * self.W          # 1111.1 / 800
        self.radius = float(cfg.get("camera_radius", 3.0))
        self.sphere_radius = float(cfg.get("sphere_radius", 0.5))
        self.with_distant = bool(cfg.get("with_distant", False))
        self.seed = int(cfg.get("seed", 42))
        self.world = AnalyticWorld(spheres=[([0.0, 0.0, 0.0], self.sphere_radius, [0.8, 0.55, 0.35])])
        from ..graphics.cameras import look_at_cameras
        self.intr, self.c2w, self.WH = look_at_cameras(V=self.V, radius=self.radius, H=self.H, W=self.W, f=self.focal,
                                                       seed=self.seed)
        self._cache: Dict[int, Dict[str, np.ndarray]] = {}

    # ------------------------------------------------------------------ SceneDataset interface
    @property
    def up_vec(self) -> np.ndarray:
        return np.array([0.0, -1.0, 0.0])

    def get_all_available_scenarios(self) -> List[str]:
        return ["synthetic_sphere"]

    def get_scenario(self, scene_id: str, **kwargs) -> Dict[str, Any]:
        V = self.V
        objects = {"main": dict(id="main", class_name="Main")}
        if self.with_distant:
            objects["distant"] = dict(id="distant", class_name="Distant")
        cam = dict(id="camera", class_name="Camera", n_frames=V, camera_model="pinhole",
                   data=dict(hw=np.tile(np.array([[self.H, self.W]], dtype=np.float32), (V, 1)),
                             intr=self.intr.numpy().astype(np.float32), transform=self.c2w.numpy().astype(np.float32),
                             global_frame_inds=np.arange(V)))
        return dict(scene_id=scene_id, metas=dict(n_frames=V, main_class_name="Main"), objects=objects,
                    observers=dict(camera=cam))

    def _frame(self, fi: int):
        fi = int(fi)
        if fi not in self._cache:
            o, d = _pixel_rays(self.intr[fi], self.c2w[fi], self.H, self.W)
            tr = self.world.trace(o, d)
            rgb = torch.where(tr["hit"][:, None], tr["rgb"], torch.zeros_like(tr["rgb"]))      # black background
            self._cache[fi] = dict(rgb=rgb.view(self.H, self.W, 3).numpy().astype(np.float32),
                                   mask=tr["hit"].view(self.H, self.W).numpy())
        return self._cache[fi]

    def get_image_wh(self, scene_id: str, camera_id: str, frame_index) -> np.ndarray:
        return np.array([self.W, self.H])

    def get_image(self, scene_id: str, camera_id: str, frame_index: int) -> np.ndarray:
        return self._frame(frame_index)["rgb"]

    def get_image_occupancy_mask(self, scene_id: str, camera_id: str, frame_index: int, **kw) -> np.ndarray:
        return self._frame(frame_index)["mask"]
