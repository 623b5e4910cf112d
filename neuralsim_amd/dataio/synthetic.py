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


class SyntheticStreetDataset:
    """``dataset_cfg.target: neuralsim_amd.dataio.SyntheticStreetDataset`` -- a street-view scenario in the layout of the
    reference's autonomous-driving datasets (docs/data/autonomous_driving.md:38-100; the scene-graph form
    dataio/autonomous_driving/waymo/waymo_dataset.py:432-735 emits with ``scene_graph_has_ego_car: true``): an ``EgoVehicle``
    observer node driving along +x with its sensors as CHILDREN -- a multi-camera rig (front, front-left / -right, side-left /
    -right; ``camera_model: opencv`` with a distortion vector when ``consider_distortion``) and a roof ``RaysLidar`` -- one
    ``Street`` object (``main_class_name``) and, optionally (``n_vehicles``), moving ``Vehicle`` boxes with per-frame
    ``transform`` / ``scale`` segments.  Images, occupancy masks (not sky) and lidar returns come from tracing the analytic
    world of ``neuralsim_amd.scenarios.street_world`` (+ the vehicles as rounded boxes): no files."""

    CAMERAS = ["camera_FRONT", "camera_FRONT_LEFT", "camera_FRONT_RIGHT", "camera_SIDE_LEFT", "camera_SIDE_RIGHT"]
    YAWS = dict(camera_FRONT=0.0, camera_FRONT_LEFT=50.0, camera_FRONT_RIGHT=-50.0, camera_SIDE_LEFT=100.0, camera_SIDE_RIGHT=-100.0)

    def __init__(self, config: dict = None):
        import math
        from ..scenarios import _pose, street_world
        cfg = dict(config or {})
        self.n_frames = int(cfg.get("n_frames", 8))
        self.H, self.W = int(cfg.get("image_h", 24)), int(cfg.get("image_w", 32))
        self.focal = float(cfg.get("focal_ratio", 0.625)) * self.W
        self.beams = int(cfg.get("lidar_beams", 256))
        self.x_range = tuple(cfg.get("x_range", (-40.0, 40.0)))
        self.n_vehicles = int(cfg.get("n_vehicles", 0))
        self.distortion = [float(v) for v in cfg.get("distortion", [0.02, -0.05, 0.0005, -0.0005, 0.0])]
        self.seed = int(cfg.get("seed", 42))
        self.world = street_world()
        F_ = self.n_frames
        # ego poses (vehicle-to-world, waymo convention: +x forward, +y left, +z up)
        self.v2w = np.tile(np.eye(4, dtype=np.float32)[None], (F_, 1, 1))
        for e in range(F_):
            self.v2w[e, 0, 3] = self.x_range[0] + (self.x_range[1] - self.x_range[0]) * (e + 0.5) / F_
            self.v2w[e, 1, 3] = 0.3 * math.sin(1.3 * e)
        # camera-to-vehicle (OpenCV camera axes), slight downward pitch, mounted 1.2 m ahead of / 0.2 m above the origin
        self.c2v = {}
        for cid in self.CAMERAS:
            a = math.radians(self.YAWS[cid])
            self.c2v[cid] = _pose([1.2 * math.cos(a), 1.2 * math.sin(a), 0.2], [math.cos(a), math.sin(a), -0.08]).numpy().astype(np.float32)
        self.l2v = np.eye(4, dtype=np.float32)
        self.l2v[2, 3] = 0.5
        self.intr = np.array([[self.focal, 0, self.W / 2], [0, self.focal, self.H / 2], [0, 0, 1.0]], dtype=np.float32)
        # vehicles: boxes 4.6 x 1.9 x 1.6 m on the road, moving along x
        self.vehicles = []
        g = torch.Generator().manual_seed(self.seed)
        for k in range(self.n_vehicles):
            u = torch.rand(4, generator=g)
            y = (-1.0 if k % 2 else 1.0) * (3.5 + 1.5 * float(u[0]))
            x0, v = self.x_range[0] + float(u[1]) * (self.x_range[1] - self.x_range[0]), 2.0 + 4.0 * float(u[2])
            tr = np.tile(np.eye(4, dtype=np.float32)[None], (F_, 1, 1))
            for e in range(F_):
                tr[e, :3, 3] = [x0 + v * (e - F_ / 2) * 0.5, y, -2.0 + 0.8]
            self.vehicles.append(dict(id=f"veh{k}", transform=tr, scale=np.tile(np.array([[4.6, 1.9, 1.6]], dtype=np.float32), (F_, 1)),
                                      color=[0.2 + 0.7 * float(u[3]), 0.3, 0.8 - 0.6 * float(u[3])]))
        self._cache: Dict[Any, Dict[str, np.ndarray]] = {}

    # ------------------------------------------------------------------ SceneDataset interface
    @property
    def up_vec(self) -> np.ndarray:
        return np.array([0.0, 0.0, 1.0])

    @property
    def forward_vec(self) -> np.ndarray:
        return np.array([1.0, 0.0, 0.0])

    @property
    def right_vec(self) -> np.ndarray:
        return np.array([0.0, -1.0, 0.0])

    main_class_name = "Street"

    def get_all_available_scenarios(self) -> List[str]:
        return ["synthetic_street"]

    def get_scenario(self, scene_id: str, *, observer_cfgs: dict = None, object_cfgs: dict = None, load_class_names=None,
                     no_objects: bool = False, consider_distortion: bool = True, scene_graph_has_ego_car: bool = True,
                     align_orientation: bool = True, start=None, stop=None, aabb_extend: float = 60.0, **unused) -> Dict[str, Any]:
        F_ = self.n_frames
        assert start in (None, 0) and stop in (None, F_), "the synthetic street is generated at its full length"
        ts = np.linspace(-0.95, 0.95, F_).astype(np.float32)
        fi = np.arange(F_)
        cams = list((observer_cfgs or {}).get("Camera", {}).get("list", self.CAMERAS)) if observer_cfgs else list(self.CAMERAS)
        lidars = list((observer_cfgs or {}).get("RaysLidar", {}).get("list", ["lidar_TOP"])) if observer_cfgs else ["lidar_TOP"]
        objects: Dict[str, Any] = {}
        street = dict(id="street", class_name="Street")
        if align_orientation:
            street.update(n_frames=F_, data=dict(transform=np.tile(np.eye(4, dtype=np.float32)[None], (F_, 1, 1))))
        objects["street"] = street
        if not no_objects and (load_class_names is None or "Vehicle" in load_class_names):
            for v in self.vehicles:
                objects[v["id"]] = dict(id=v["id"], class_name="Vehicle", segments=[dict(
                    start_frame=0, n_frames=F_, data=dict(transform=v["transform"], scale=v["scale"], global_frame_inds=fi,
                                                          global_timestamps=ts))])
        observers: Dict[str, Any] = {}
        hw = np.tile(np.array([[self.H, self.W]], dtype=np.float32), (F_, 1))
        intr = np.tile(self.intr[None], (F_, 1, 1))
        dist = np.tile(np.array([self.distortion], dtype=np.float32), (F_, 1))

        def cam_node(cid, transform):
            return dict(id=cid, class_name="Camera", n_frames=F_, camera_model="opencv" if consider_distortion else "pinhole",
                        data=dict(hw=hw, intr=intr, distortion=dist, transform=transform, global_frame_inds=fi, global_timestamps=ts))

        def lidar_node(lid, transform):
            return dict(id=lid, class_name="RaysLidar", n_frames=F_, data=dict(transform=transform, global_frame_inds=fi,
                                                                                 global_timestamps=ts))
        if scene_graph_has_ego_car:
            ego = dict(id="ego_car", class_name="EgoVehicle", n_frames=F_, children=dict(),
                       data=dict(transform=self.v2w.copy(), global_frame_inds=fi, global_timestamps=ts))
            for cid in cams:
                ego["children"][cid] = cam_node(cid, np.tile(self.c2v[cid][None], (F_, 1, 1)))
            for lid in lidars:
                ego["children"][lid] = lidar_node(lid, np.tile(self.l2v[None], (F_, 1, 1)))
            observers["ego_car"] = ego
        else:
            for cid in cams:
                observers[cid] = cam_node(cid, self.v2w @ self.c2v[cid][None])
            for lid in lidars:
                observers[lid] = lidar_node(lid, self.v2w @ self.l2v[None])
        track = self.v2w[:, :3, 3]
        metas = dict(n_frames=F_, main_class_name="Street", frame_timestamps=ts, up_vec="+z", align_orientation=align_orientation,
                     average_rot_z=0.0, average_rot_mat=np.eye(3),
                     aabb=np.stack([track.min(0) - aabb_extend, track.max(0) + aabb_extend], axis=0))
        return dict(scene_id=scene_id, metas=metas, objects=objects, observers=observers)

    def get_scenario_background_only(self, scene_id: str, **kwargs):
        return self.get_scenario(scene_id, no_objects=True, **kwargs)

    # ------------------------------------------------------------------ sensor data (traced)
    def _c2w(self, cid: str, fi: int) -> torch.Tensor:
        return torch.from_numpy(self.v2w[int(fi)] @ self.c2v[cid])

    def _trace(self, o, d, fi: int):
        """The analytic street + the vehicle boxes of frame fi (axis-aligned rounded boxes: slab test)."""
        tr = self.world.trace(o, d)
        t, hit, rgb = tr["t"].clone(), tr["hit"].clone(), tr["rgb"].clone()
        for v in self.vehicles:
            c = torch.from_numpy(v["transform"][int(fi)][:3, 3])
            h = torch.from_numpy(v["scale"][int(fi)]) * 0.5
            inv = 1.0 / torch.where(d.abs() < 1e-9, torch.full_like(d, 1e-9), d)
            t0, t1 = (c - h - o) * inv, (c + h - o) * inv
            tn, tf = torch.minimum(t0, t1).amax(-1), torch.maximum(t0, t1).amin(-1)
            hb = (tn < tf) & (tn > 0) & (~hit | (tn < t))
            t = torch.where(hb, tn, t)
            rgb = torch.where(hb[:, None], torch.tensor(v["color"]).expand_as(rgb), rgb)
            hit = hit | hb
        return t, hit, rgb

    def _frame(self, cid: str, fi: int):
        key = (cid, int(fi))
        if key not in self._cache:
            o, d = _pixel_rays(torch.from_numpy(self.intr), self._c2w(cid, fi), self.H, self.W)      # (pinhole pixels: the
            t, hit, rgb = self._trace(o, d, fi)                                                       # distortion is mild)
            sky = torch.tensor([0.55, 0.7, 0.9])
            img = torch.where(hit[:, None], rgb, sky.expand_as(rgb))
            self._cache[key] = dict(rgb=img.view(self.H, self.W, 3).numpy().astype(np.float32),
                                    mask=hit.view(self.H, self.W).numpy())
        return self._cache[key]

    def get_image_wh(self, scene_id: str, camera_id: str, frame_index) -> np.ndarray:
        return np.array([self.W, self.H])

    def get_image(self, scene_id: str, camera_id: str, frame_index: int) -> np.ndarray:
        return self._frame(camera_id, frame_index)["rgb"]

    def get_image_occupancy_mask(self, scene_id: str, camera_id: str, frame_index: int, **kw) -> np.ndarray:
        return self._frame(camera_id, frame_index)["mask"]

    def get_image_semantic_mask_by_type(self, scene_id: str, camera_id: str, sem_type: str, frame_index: int, **kw) -> np.ndarray:
        """``sem_type`` in dynamic | human | road (dataio/autonomous_driving/waymo/waymo_dataset.py:274-295): the vehicles are the
        dynamic pixels, the ground plane is the road, nobody walks here."""
        if sem_type == "human":
            return np.zeros([self.H, self.W], dtype=bool)
        o, d = _pixel_rays(torch.from_numpy(self.intr), self._c2w(camera_id, frame_index), self.H, self.W)
        tr = self.world.trace(o, d)
        t_all, hit_all, _ = self._trace(o, d, frame_index)
        if sem_type == "dynamic":
            m = hit_all & (~tr["hit"] | (t_all < tr["t"] - 1e-4))
        elif sem_type == "road":
            z = o[:, 2] + t_all * d[:, 2]
            m = hit_all & (z < -2.0 + 1e-2)
        else:
            raise RuntimeError(f"Invalid sem_type={sem_type}")
        return m.view(self.H, self.W).numpy()

    def get_lidar(self, scene_id: str, lidar_id: str, frame_index: int) -> Dict[str, np.ndarray]:
        """Beams in the LIDAR's frame (the scene graph applies ego x lidar-to-vehicle): azimuth uniform, elevation in
        [-22, +2.5] degrees; ``ranges`` = 0 where nothing is hit within 200 m."""
        import math
        g = torch.Generator().manual_seed(self.seed + 977 * int(frame_index))
        az = torch.rand(self.beams, generator=g) * (2 * math.pi)
        el = torch.deg2rad(-22.0 + 24.5 * torch.rand(self.beams, generator=g))
        d_l = torch.stack([torch.cos(el) * torch.cos(az), torch.cos(el) * torch.sin(az), torch.sin(el)], dim=-1)
        l2w = torch.from_numpy(self.v2w[int(frame_index)] @ self.l2v)
        o_w = l2w[:3, 3].expand(self.beams, 3).contiguous()
        d_w = (l2w[:3, :3] * d_l.unsqueeze(-2)).sum(-1)
        t, hit, _ = self._trace(o_w, d_w, frame_index)
        rng = torch.where(hit & (t < 200.0), t, torch.zeros_like(t))
        return dict(rays_o=np.zeros([self.beams, 3], dtype=np.float32), rays_d=d_l.numpy().astype(np.float32),
                    ranges=rng.numpy().astype(np.float32))
