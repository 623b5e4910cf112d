"""Test-only harness that runs the REFERENCE's own renderer sources on top of this repository's models.

The reference's ``app/renderers/single_volume_renderer.py`` and ``buffer_compose_renderer.py`` are Python that exists
in ``/root/reference``; everything they call below ``ray_test`` / ``ray_query`` / ``nr3d_lib.graphics.*`` is the absent
nr3d_lib, i.e. exactly the surface this repository provides.  This module loads those two files *unchanged, from where
they lie* (nothing is copied), gives them

  * the repository's ``nr3d_lib`` shim for the hot-path imports (``nr3d_lib.graphics.{nerf, pack_ops}``,
    ``nr3d_lib.models.utils``, ``nr3d_lib.profile``, ``nr3d_lib.config``), and
  * duck-typed stand-ins for the scene-graph classes they import but that are out of scope here
    (``app.resources.{Scene, AssetBank, SceneNode}``, the observer classes, ``render_parallel``),

and lets the tests drive ``SingleVolumeRenderer.ray_query`` of the reference against ``FakeScene`` objects that hold
this repository's models.  It is used (i) to show the drop-in claim -- the reference's renderer code runs on the
mirror of the nr3d_lib operator surface -- and (ii) to pin the renderer mirror
(``neuralsim_amd.renderers.SingleVolumeRenderer``) and the oracle's integration to outputs of the reference's own
glue (``tests/golden/make_renderer_fixture.py`` -> ``tests/golden/renderer_fixture.pt``).

``/root/reference`` exists in the authoring container only: every user of this module skips when it is absent.
"""
import contextlib
import importlib.util
import sys
import types
from pathlib import Path

import torch

REF_ROOT = Path("/root/reference")
_FILES = {
    "app.renderers.utils": "app/renderers/utils.py",
    "app.renderers.single_volume_renderer": "app/renderers/single_volume_renderer.py",
}


def reference_available() -> bool:
    return all((REF_ROOT / f).exists() for f in _FILES.values())


def _stub_module(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    return m


class _Named:
    """Stand-in for a scene-graph class the renderer only uses in annotations / isinstance checks."""
    def __init__(self, id="obs0"):
        self.id = id


@contextlib.contextmanager
def reference_renderer_modules():
    """-> dict of the reference's renderer modules, imported from /root/reference with the stand-ins installed.
    ``sys.modules`` is restored on exit (the ``app`` package of the reference never stays importable)."""
    assert reference_available()
    root = str(Path(__file__).resolve().parent.parent)
    if root not in sys.path:
        sys.path.insert(0, root)
    import nr3d_lib.config  # noqa: F401  (the shim's harness-free subset)
    import nr3d_lib.profile  # noqa: F401
    saved = {k: v for k, v in sys.modules.items() if k == "app" or k.startswith("app.")}
    for k in saved:
        del sys.modules[k]
    classes = {n: type(n, (_Named,), {}) for n in
               ("Scene", "AssetBank", "SceneNode", "Camera", "MultiCamBundle", "Lidar", "RaysLidar", "MultiRaysLidarBundle")}
    app = _stub_module("app")
    app.__path__ = []
    rend = _stub_module("app.renderers")
    rend.__path__ = []
    sys.modules.update({
        "app": app, "app.renderers": rend,
        "app.resources": _stub_module("app.resources", **{k: classes[k] for k in ("Scene", "AssetBank", "SceneNode")}),
        "app.resources.observers": _stub_module("app.resources.observers", **{
            k: classes[k] for k in ("Camera", "MultiCamBundle", "Lidar", "RaysLidar", "MultiRaysLidarBundle")}),
        "app.renderers.render_parallel": _stub_module(
            "app.renderers.render_parallel", render_parallel=None, render_parallel_with_replicas=None,
            EvalParallelWrapper=None),
    })
    try:
        mods = {}
        for name, rel in _FILES.items():
            spec = importlib.util.spec_from_file_location(name, str(REF_ROOT / rel))
            mod = importlib.util.module_from_spec(spec)
            sys.modules[name] = mod
            spec.loader.exec_module(mod)
            mods[name] = mod
        mods["classes"] = classes
        if (REF_ROOT / "app/resources/scenes.py").exists():
            with reference_scene_class() as Scene:       # the reference's own world -> object ray conversion
                mods["convert_rays_in_node"] = Scene.convert_rays_in_node
                mods["convert_rays_in_nodes_list"] = Scene.convert_rays_in_nodes_list
            sys.modules["app"], sys.modules["app.resources"] = app, sys.modules.get("app.resources") or app
        yield mods
    finally:
        for k in [k for k in sys.modules if k == "app" or k.startswith("app.")]:
            del sys.modules[k]
        sys.modules.update(saved)


@contextlib.contextmanager
def reference_compose_renderer_modules():
    """-> the reference's ``app/renderers/buffer_compose_renderer.py`` (multi-object renderer of code_multi), loaded
    unchanged, plus the grouping / ray-conversion statics of its ``Scene``.  Stand-ins: ``torch_scatter`` and
    ``matplotlib`` (imported at module level, used by the segmentation z-buffer / debug plots only),
    ``nr3d_lib.utils.IDListedDict``, ``app.models.asset_base`` (the ``AssetAssignment`` enum, same members)."""
    import enum
    assert (REF_ROOT / "app/renderers/buffer_compose_renderer.py").exists()
    with reference_renderer_modules() as mods:          # installs app / app.renderers / utils / resources stubs
        with reference_scene_class() as Scene:
            pass
        names = ["torch_scatter", "matplotlib", "matplotlib.pyplot", "nr3d_lib.utils", "app.models", "app.models.asset_base",
                 "app.renderers.buffer_compose_renderer"]
        saved = {k: sys.modules.get(k) for k in names}

        class AssetAssignment(enum.Enum):
            OBJECT = 0
            SCENE = 1
            MULTI_OBJ_ONE_SCENE = 2
            MULTI_OBJ_MULTI_SCENE = 3
            MULTI_OBJ = 3
            MULTI_SCENE = 4
            MISC = 5
        import nr3d_lib.config  # noqa: F401
        scenes_mod = sys.modules.get("app.resources.scenes")
        res = sys.modules["app.resources"]
        res.namedtuple_ind_id_obj = Scene.group_drawables_by_class_name.__globals__["namedtuple_ind_id_obj"]
        mpl = _stub_module("matplotlib")
        mpl.__path__ = []
        am = _stub_module("app.models")
        am.__path__ = []
        sys.modules.update({
            "torch_scatter": _stub_module("torch_scatter", scatter_min=_scatter_min),
            "matplotlib": mpl, "matplotlib.pyplot": _stub_module("matplotlib.pyplot"),
            "nr3d_lib.utils": _stub_module("nr3d_lib.utils", IDListedDict=dict),
            "app.models": am,
            "app.models.asset_base": _stub_module("app.models.asset_base", AssetAssignment=AssetAssignment,
                                                  AssetModelMixin=object),
        })
        try:
            spec = importlib.util.spec_from_file_location("app.renderers.buffer_compose_renderer",
                                                          str(REF_ROOT / "app/renderers/buffer_compose_renderer.py"))
            mod = importlib.util.module_from_spec(spec)
            sys.modules[spec.name] = mod
            spec.loader.exec_module(mod)
            mods = dict(mods, compose=mod, Scene=Scene, AssetAssignment=AssetAssignment)
            del scenes_mod
            yield mods
        finally:
            for k, v in saved.items():
                if v is None:
                    sys.modules.pop(k, None)
                else:
                    sys.modules[k] = v


def _scatter_min(src, index, dim=0):
    """torch_scatter.scatter_min for 1-D inputs (the segmentation z-buffer of buffer_compose_renderer.py:296-301)."""
    n = int(index.max()) + 1
    out = torch.full([n], float("inf"), dtype=src.dtype, device=src.device).scatter_reduce(0, index, src, reduce="amin")
    is_min = src == out[index]
    pos = torch.arange(src.numel(), device=src.device)
    arg = torch.full([n], src.numel(), dtype=torch.long, device=src.device).scatter_reduce(
        0, index[is_min], pos[is_min], reduce="amin")
    return out, arg


class FakeComposeScene:
    """What ``BufferComposeRenderer.ray_query`` touches of the Scene (buffer_compose_renderer.py:133-168, 198):
    drawables in a fixed order, grouping by class name and the batched ray conversion (both the REFERENCE's static
    methods), class / instance index maps, image embeddings."""
    def __init__(self, device, Scene, image_embeddings=None):
        self.device, self._Scene, self.image_embeddings = device, Scene, image_embeddings
        self.nodes = []
        self.drawable_groups_by_class_name = {}
        self.convert_rays_in_nodes_list = Scene.convert_rays_in_nodes_list
        self.group_drawables_by_class_name = Scene.group_drawables_by_class_name

    def add(self, node):
        self.nodes.append(node)
        self.drawable_groups_by_class_name.setdefault(node.class_name, []).append(node)
        node.full_unique_id = node.id
        node.i_valid_flags = torch.tensor(True, device=self.device)
        return node

    def get_drawables(self):
        return [n for n in self.nodes if n.class_name != "Sky"]

    def get_drawable_groups_by_class_name(self, class_name):
        return self.drawable_groups_by_class_name.get(class_name, [])

    def get_drawable_class_ind_map(self):
        return {c: i for i, c in enumerate(self.drawable_groups_by_class_name)}

    def get_drawable_instance_ind_map(self):
        return {n.id: i for i, n in enumerate(self.nodes)}


class FakeObserver(_Named):
    def filter_drawable_groups(self, drawables):
        return drawables


@contextlib.contextmanager
def reference_camera_class():
    """-> the reference's ``Camera`` class (app/resources/observers/cameras.py, loaded unchanged).  Its ray methods are
    called UNBOUND on a ``FakeCamera``: the snapping / clamping / normalisation / origin logic is the reference's,
    ``intr.lift`` and ``world_transform.rotate`` (nr3d_lib.models.attributes, absent) are the stand-ins below."""
    assert (REF_ROOT / "app/resources/observers/cameras.py").exists()
    names = ["nr3d_lib.utils", "nr3d_lib.models.attributes", "nr3d_lib.graphics.cameras", "app", "app.resources",
             "app.resources.nodes", "app.resources.observers", "app.resources.observers.cameras"]
    saved = {k: sys.modules.get(k) for k in names}
    attrs = _stub_module("nr3d_lib.models.attributes")
    attrs.__all__ = []
    app, res, obs = _stub_module("app"), _stub_module("app.resources"), _stub_module("app.resources.observers")
    app.__path__, res.__path__, obs.__path__ = [], [], []
    sys.modules.update({
        "nr3d_lib.utils": _stub_module("nr3d_lib.utils", is_scalar=lambda x: not hasattr(x, "__len__")),
        "nr3d_lib.models.attributes": attrs,
        "nr3d_lib.graphics.cameras": _stub_module("nr3d_lib.graphics.cameras", pinhole_lift=None, sphere_inside_frustum=None),
        "app": app, "app.resources": res, "app.resources.observers": obs,
        "app.resources.nodes": _stub_module("app.resources.nodes", SceneNode=type("SceneNode", (), {})),
    })
    try:
        spec = importlib.util.spec_from_file_location("app.resources.observers.cameras",
                                                      str(REF_ROOT / "app/resources/observers/cameras.py"))
        mod = importlib.util.module_from_spec(spec)
        sys.modules[spec.name] = mod
        spec.loader.exec_module(mod)
        yield mod.Camera
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v


@contextlib.contextmanager
def reference_scene_class():
    """-> the reference's ``Scene`` class (app/resources/scenes.py, loaded unchanged) for its static ray conversions
    ``convert_rays_in_node`` / ``convert_rays_in_nodes_list`` (:629-708)."""
    assert (REF_ROOT / "app/resources/scenes.py").exists()
    names = ["nr3d_lib.utils", "nr3d_lib.models.attributes", "nr3d_lib.models.accelerations.occgrid_accel", "app",
             "app.resources", "app.resources.observers", "app.resources.scenes"]
    saved = {k: sys.modules.get(k) for k in names}
    attrs = _stub_module("nr3d_lib.models.attributes")
    attrs.__all__ = []
    app = _stub_module("app")
    app.__path__ = []
    res = _stub_module("app.resources", SceneNode=type("SceneNode", (), {}))
    res.__path__ = []
    obs = _stub_module("app.resources.observers", OBSERVER_CLASS_NAMES=[], OBSERVER_TYPE=object,
                       Camera=type("Camera", (), {}), Lidar=type("Lidar", (), {}), RaysLidar=type("RaysLidar", (), {}))
    import nr3d_lib.models.accelerations as acc
    sys.modules.update({
        "nr3d_lib.utils": _stub_module("nr3d_lib.utils", IDListedDict=dict, get_shape=None, import_str=None,
                                       check_to_torch=None),
        "nr3d_lib.models.attributes": attrs,
        "nr3d_lib.models.accelerations.occgrid_accel": _stub_module("nr3d_lib.models.accelerations.occgrid_accel",
                                                                    OccGridAccel=acc.OccGridAccel),
        "app": app, "app.resources": res, "app.resources.observers": obs,
    })
    try:
        spec = importlib.util.spec_from_file_location("app.resources.scenes", str(REF_ROOT / "app/resources/scenes.py"))
        mod = importlib.util.module_from_spec(spec)
        sys.modules[spec.name] = mod
        spec.loader.exec_module(mod)
        yield mod.Scene
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v


@contextlib.contextmanager
def reference_lidar_loss_module():
    """-> the reference's app/loss/lidar.py (LineOfSightLoss, DepthLoss), loaded unchanged.  ``packed_sum`` & co come
    from the shim; the harness pieces it imports (logger, annealers, the elementwise recon losses of nr3d_lib) get
    minimal stand-ins."""
    assert (REF_ROOT / "app/loss/lidar.py").exists()
    names = ["nr3d_lib.logger", "nr3d_lib.models.annealers", "nr3d_lib.models.loss", "nr3d_lib.models.loss.recon", "app",
             "app.resources", "app.loss", "app.loss.lidar"]
    saved = {k: sys.modules.get(k) for k in names}

    def masked_mean(x, mask):
        return x.mean() if mask is None else (x * mask).sum() / mask.sum().clamp_min(1)
    recon = _stub_module(
        "nr3d_lib.models.loss.recon",
        l1_loss=lambda p, g, mask=None, reduction="mean": masked_mean((p - g).abs(), mask),
        l2_loss=lambda p, g, mask=None, reduction="mean": masked_mean((p - g) ** 2, mask),
        relative_l2_loss=lambda p, g, mask=None, reduction="mean": masked_mean((p - g) ** 2 / (g ** 2 + 1e-2), mask),
        huber_loss=lambda p, g, mask=None, reduction="mean", alpha=1.0: masked_mean(
            torch.nn.functional.huber_loss(p, g, reduction="none", delta=alpha), mask))
    app, lossp = _stub_module("app"), _stub_module("app.loss")
    app.__path__, lossp.__path__ = [], []
    sys.modules.update({
        "nr3d_lib.logger": _stub_module("nr3d_lib.logger", Logger=object),
        "nr3d_lib.models.annealers": _stub_module("nr3d_lib.models.annealers", get_annealer=None, get_anneal_val=None),
        "nr3d_lib.models.loss": _stub_module("nr3d_lib.models.loss"), "nr3d_lib.models.loss.recon": recon,
        "app": app, "app.loss": lossp, "app.resources": _stub_module("app.resources", Scene=object),
    })
    try:
        spec = importlib.util.spec_from_file_location("app.loss.lidar", str(REF_ROOT / "app/loss/lidar.py"))
        mod = importlib.util.module_from_spec(spec)
        sys.modules[spec.name] = mod
        spec.loader.exec_module(mod)
        yield mod
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v


@contextlib.contextmanager
def reference_eikonal_loss_module():
    """-> the reference's app/loss/eikonal.py (EikonalLoss), loaded unchanged.  ``packed_sum`` comes from the shim
    (``nr3d_lib.graphics.pack_ops.pack_ops``); ``safe_mse_loss`` (absent nr3d_lib) is a stand-in that is only reached
    with ``safe_mse=True`` -- the tests construct the loss with ``safe_mse=False`` (plain ``F.mse_loss``)."""
    assert (REF_ROOT / "app/loss/eikonal.py").exists()
    names = ["nr3d_lib.logger", "nr3d_lib.models.annealers", "nr3d_lib.models.loss", "nr3d_lib.models.loss.safe",
             "nr3d_lib.utils", "app", "app.resources", "app.loss", "app.loss.eikonal"]
    saved = {k: sys.modules.get(k) for k in names}
    app, lossp = _stub_module("app"), _stub_module("app.loss")
    app.__path__, lossp.__path__ = [], []
    sys.modules.update({
        "nr3d_lib.logger": _stub_module("nr3d_lib.logger", Logger=object),
        "nr3d_lib.models.annealers": _stub_module("nr3d_lib.models.annealers", get_annealer=None, get_anneal_val=None),
        "nr3d_lib.models.loss": _stub_module("nr3d_lib.models.loss"),
        "nr3d_lib.models.loss.safe": _stub_module("nr3d_lib.models.loss.safe", safe_mse_loss=None),
        "nr3d_lib.utils": _stub_module("nr3d_lib.utils", tensor_statistics=None),
        "app": app, "app.loss": lossp, "app.resources": _stub_module("app.resources", Scene=object, SceneNode=object),
    })
    try:
        spec = importlib.util.spec_from_file_location("app.loss.eikonal", str(REF_ROOT / "app/loss/eikonal.py"))
        mod = importlib.util.module_from_spec(spec)
        sys.modules[spec.name] = mod
        spec.loader.exec_module(mod)
        yield mod
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v


@contextlib.contextmanager
def reference_loss_module(name: str):
    """-> the reference's ``app/loss/<name>.py`` loaded unchanged, for the loss modules that only need the shim's pack
    ops plus harness stand-ins (``get_annealer`` is only reached with an ``anneal`` config, which the tests leave None)."""
    path = REF_ROOT / "app" / "loss" / f"{name}.py"
    assert path.exists()
    names = ["nr3d_lib.logger", "nr3d_lib.models.annealers", "app", "app.resources", "app.loss", f"app.loss.{name}"]
    saved = {k: sys.modules.get(k) for k in names}
    app, lossp = _stub_module("app"), _stub_module("app.loss")
    app.__path__, lossp.__path__ = [], []
    sys.modules.update({
        "nr3d_lib.logger": _stub_module("nr3d_lib.logger", Logger=object),
        "nr3d_lib.models.annealers": _stub_module("nr3d_lib.models.annealers", get_annealer=None, get_anneal_val=None),
        "app": app, "app.loss": lossp, "app.resources": _stub_module("app.resources", Scene=object, SceneNode=object),
    })
    try:
        spec = importlib.util.spec_from_file_location(f"app.loss.{name}", str(path))
        mod = importlib.util.module_from_spec(spec)
        sys.modules[spec.name] = mod
        spec.loader.exec_module(mod)
        yield mod
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v


class FakePinhole:
    """Stand-in for nr3d_lib's PinholeCameraMatKHW attribute: mat [...,3,3], W/H (scalars or [...]); ``lift`` is the
    textbook pinhole back-projection ((u - cx) / fx * d, (v - cy) / fy * d, d)."""
    def __init__(self, mat, WH):
        self.mat, self._wh = mat, WH
        self.W, self.H = int(WH.reshape(-1, 2)[0, 0]), int(WH.reshape(-1, 2)[0, 1])

    def __getitem__(self, i):
        return FakePinhole(self.mat[i], self._wh[i])

    def wh(self):
        return self._wh

    def lift(self, u, v, d):
        m = self.mat
        fx, fy, cx, cy = m[..., 0, 0], m[..., 1, 1], m[..., 0, 2], m[..., 1, 2]
        return torch.stack([(u - cx) / fx * d, (v - cy) / fy * d, d], dim=-1)


class FakeOpenCV(FakePinhole):
    """Stand-in for nr3d_lib's OpenCVCameraMatHW (``camera_model: opencv``, cameras.py:84-87): the pinhole lift of the
    UNDISTORTED coordinates; the undistortion is the oracle's restatement (cv::undistortPoints iteration)."""
    def __init__(self, mat, WH, distortion, n_iters=5):
        super().__init__(mat, WH)
        self.distortion, self.n_iters = distortion, n_iters

    def __getitem__(self, i):
        return FakeOpenCV(self.mat[i], self._wh[i], self.distortion[i], self.n_iters)

    def lift(self, u, v, d):
        from oracle import render as orr
        m = self.mat
        fx, fy, cx, cy = m[..., 0, 0], m[..., 1, 1], m[..., 0, 2], m[..., 1, 2]
        x, y = orr.opencv_undistort((u - cx) / fx, (v - cy) / fy, self.distortion, self.n_iters)
        return torch.stack([x * d, y * d, d], dim=-1)


class FakeFisheye(FakePinhole):
    """Stand-in for nr3d_lib's FisheyeCameraMatHW (``camera_model: fisheye``, cameras.py:88-92): the lift inverts the OpenCV
    fisheye polynomial the reference applies in app/resources/observers/fisheye.py:36-42 (the oracle's restatement, 10 Newton
    rounds); the direction it returns is scaled to depth d along z like the pinhole lift (the Camera code normalises it)."""
    def __init__(self, mat, WH, distortion, n_iters=10):
        super().__init__(mat, WH)
        self.distortion, self.n_iters = distortion, n_iters

    def __getitem__(self, i):
        return FakeFisheye(self.mat[i], self._wh[i], self.distortion[i], self.n_iters)

    def lift(self, u, v, d):
        from oracle import render as orr
        m = self.mat
        fx, fy, cx, cy = m[..., 0, 0], m[..., 1, 1], m[..., 0, 2], m[..., 1, 2]
        x, y, z = orr.fisheye_lift((u - cx) / fx, (v - cy) / fy, self.distortion, self.n_iters)
        return torch.stack([x * d, y * d, z * d], dim=-1)


class FakePose:
    """Stand-in for nr3d_lib's TransformMat4x4: ``rotate`` is broadcast-multiply-sum (cameras.py:355-359 forbids mm)."""
    def __init__(self, mat):
        self.mat = mat

    def __getitem__(self, i):
        return FakePose(self.mat[i])

    def rotate(self, v):
        R = self.mat[..., :3, :3]
        if R.dim() == 2:
            return (R * v.unsqueeze(-2)).sum(-1)
        return (R.view(*R.shape[:-2], *[1] * (v.dim() - R.dim() + 1), 3, 3) * v.unsqueeze(-2)).sum(-1)

    def translation(self):
        return self.mat[..., :3, 3]


class FakeCamera:
    def __init__(self, intr, c2w, WH, i_prefix=(), distortion=None):
        if distortion is None:
            self.intr = FakePinhole(intr, WH)
        else:
            self.intr = FakeFisheye(intr, WH, distortion) if distortion.shape[-1] == 4 else FakeOpenCV(intr, WH, distortion)
        self.world_transform = FakePose(c2w)
        self.i_prefix, self.device, self.dtype = tuple(i_prefix), intr.device, intr.dtype


class FakeTransform:
    """world_transform of a scene node: rotation [3,3], translation [3], scale (scalar)."""
    def __init__(self, R=None, t=None, s=1.0, device=None):
        self.R = torch.eye(3, device=device) if R is None else R
        self.t = torch.zeros(3, device=device) if t is None else t
        self.s = s

    def rotation(self):
        return self.R

    def translation(self):
        return self.t

    def rotate(self, v):
        return (self.R * v.unsqueeze(-2)).sum(-1)

    def vec_3(self):          # node.scale.vec_3()
        return torch.full([3], float(self.s), device=self.R.device) if not torch.is_tensor(self.s) else self.s


class FakeNode:
    def __init__(self, model, class_name, id, world_transform=None):
        self.model, self.class_name, self.id = model, class_name, id
        self.world_transform = world_transform or FakeTransform()
        self.scale = self.world_transform
        if not hasattr(model, "id"):
            model.id = f"{class_name}#model"


class FakeScene:
    """The part of ``app.resources.Scene`` that ``SingleVolumeRenderer.ray_query`` touches
    (single_volume_renderer.py:157-263): drawable groups by class name, the device, the image embeddings and
    ``convert_rays_in_node`` (scenes.py:686-708 -- world -> object by the node's rotation / translation / scale)."""
    def __init__(self, device, main_class_name="Main", image_embeddings=None, convert_rays_in_node=None):
        """``convert_rays_in_node``: the reference's own static method (``reference_scene_class``) when the caller
        loaded it; the restatement below otherwise."""
        self.device = device
        if convert_rays_in_node is not None:
            self.convert_rays_in_node = convert_rays_in_node
        self.main_class_name = main_class_name
        self.image_embeddings = image_embeddings
        self.drawable_groups_by_class_name = {}

    def add(self, node: FakeNode):
        self.drawable_groups_by_class_name.setdefault(node.class_name, []).append(node)
        return node

    def get_drawable_groups_by_class_name(self, class_name):
        return self.drawable_groups_by_class_name.get(class_name, [])

    def convert_rays_in_node(self, rays_o, rays_d, node):
        tf = node.world_transform
        Rt = tf.R.t()
        o = (Rt * (rays_o - tf.t).unsqueeze(-2)).sum(-1) / tf.s
        d = (Rt * rays_d.unsqueeze(-2)).sum(-1) / tf.s
        return o, d


class FixedEmbeddings:
    """scene.image_embeddings[observer.id](rays_ts, mode='interp') with a per-ray table handed in by the test."""
    def __init__(self, h):
        self.h = h

    def __getitem__(self, key):
        return lambda rays_ts, mode="interp": self.h


def make_reference_renderer(mods, common: dict, train: dict = None, val: dict = None, training=True):
    from nr3d_lib.config import ConfigDict
    cls = mods["app.renderers.single_volume_renderer"].SingleVolumeRenderer
    r = cls(ConfigDict(common=ConfigDict(common), train=ConfigDict(train or {}), val=ConfigDict(val or {})))
    r.populate(None)
    r.train(training)
    return r


@contextlib.contextmanager
def reference_model_wrapper_modules():
    """-> the reference's asset wrappers and the two model-side losses, loaded UNCHANGED from /root/reference:
    ``app/models/asset_base.py`` (AssetMixin / AssetModelMixin / AssetAssignment), ``app/models/single/neus.py``
    (LoTDNeuSObj, LoTDNeuSStreet), ``app/models/single/nerf.py`` (LoTDNeRFDistant), ``app/loss/clearance.py``,
    ``app/loss/weight_reg.py``.  Everything they import from nr3d_lib resolves in this repository's shim
    (``nr3d_lib.{logger, config, models.{model_base, spatial, autodecoder, annealers, loss.recon, fields.{neus, nerf, sdf},
    fields_distant.nerf}}``); stand-ins only for the scene graph (``app.resources.{Scene, SceneNode}``, the observers) and
    for three harness names ``nerf.py`` imports for its MLP-NeRF classes (``check_to_torch``, ``get_embedder``,
    ``TransformMat4x4``), which the hot path never touches."""
    assert (REF_ROOT / "app/models/single/neus.py").exists()
    root = str(Path(__file__).resolve().parent.parent)
    if root not in sys.path:
        sys.path.insert(0, root)
    names = ["nr3d_lib.utils", "nr3d_lib.models.embedders", "nr3d_lib.models.attributes"]
    saved = {k: sys.modules.get(k) for k in names}
    saved_app = {k: v for k, v in sys.modules.items() if k == "app" or k.startswith("app.")}
    for k in saved_app:
        del sys.modules[k]
    classes = {n: type(n, (_Named,), {}) for n in ("Scene", "SceneNode", "Camera")}
    pk = {}
    for n in ("app", "app.models", "app.models.single", "app.models.shared", "app.loss"):
        pk[n] = _stub_module(n)
        pk[n].__path__ = []
    sys.modules.update(pk)
    sys.modules.update({
        "app.resources": _stub_module("app.resources", Scene=classes["Scene"], SceneNode=classes["SceneNode"]),
        "app.resources.observers": _stub_module("app.resources.observers", Camera=classes["Camera"]),
        "nr3d_lib.utils": _stub_module("nr3d_lib.utils", check_to_torch=lambda x, dtype=None, device=None, **k: torch.as_tensor(x, dtype=dtype, device=device),
                                       check_per_batch_tensors=__import__("nr3d_lib.utils", fromlist=["x"]).check_per_batch_tensors),
        "nr3d_lib.models.embedders": _stub_module("nr3d_lib.models.embedders", get_embedder=None),
        "nr3d_lib.models.attributes": _stub_module("nr3d_lib.models.attributes", TransformMat4x4=None),
    })
    try:
        mods = {}
        for name, rel in (("app.models.asset_base", "app/models/asset_base.py"),
                          ("app.models.single.neus", "app/models/single/neus.py"),
                          ("app.models.single.nerf", "app/models/single/nerf.py"),
                          ("app.models.shared.batched_neus", "app/models/shared/batched_neus.py"),
                          ("app.loss.clearance", "app/loss/clearance.py"),
                          ("app.loss.weight_reg", "app/loss/weight_reg.py")):
            spec = importlib.util.spec_from_file_location(name, str(REF_ROOT / rel))
            mod = importlib.util.module_from_spec(spec)
            sys.modules[name] = mod
            spec.loader.exec_module(mod)
            mods[name] = mod
        # ``model_class: app.models.single.LoTDNeuSObj`` (the package star-imports its modules, app/models/single/__init__.py)
        for m in ("app.models.single.neus", "app.models.single.nerf"):
            for n in mods[m].__all__:
                if hasattr(mods[m], n):
                    setattr(pk["app.models.single"], n, getattr(mods[m], n))
        # ``model_class: app.models.shared.AD_GenerativePermutoConcatNeuSObj`` (app/models/shared/__init__.py star-imports)
        for n in mods["app.models.shared.batched_neus"].__all__:
            if hasattr(mods["app.models.shared.batched_neus"], n):
                setattr(pk["app.models.shared"], n, getattr(mods["app.models.shared.batched_neus"], n))
        mods["classes"] = classes
        yield mods
    finally:
        for k in [k for k in sys.modules if k == "app" or k.startswith("app.")]:
            del sys.modules[k]
        sys.modules.update(saved_app)
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v


@contextlib.contextmanager
def reference_mono_loss_module():
    """-> the reference's app/loss/mono.py (MonoDepthLoss / MonoSDFDepthLoss / MonoNormalLoss: the monocular-cue losses of
    BASELINE configs[2], lotd_neus.replica.230814.yaml:272-293), loaded unchanged.  Shim: ``nr3d_lib.{logger, config,
    models.loss.{recon, utils}, models.annealers, graphics.pack_ops}``; stand-ins: the scene graph, ``kornia`` (mask
    erosion: unused with ``mask_erode 0``) and ``torchmetrics`` (Pearson variant of the depth loss: a torch restatement)."""
    assert (REF_ROOT / "app/loss/mono.py").exists()
    with reference_renderer_modules():          # app / app.renderers.utils (rotate_volume_buffer_nablas) / app.resources
        names = ["kornia", "torchmetrics", "torchmetrics.functional", "torchmetrics.functional.regression", "app.loss",
                 "app.loss.mono"]
        saved = {k: sys.modules.get(k) for k in names}

        def pearson_corrcoef(a, b):
            a, b = a - a.mean(), b - b.mean()
            return (a * b).sum() / (a.norm() * b.norm()).clamp_min(1e-12)
        tm, tmf = _stub_module("torchmetrics", PearsonCorrCoef=object), _stub_module("torchmetrics.functional")
        tm.__path__, tmf.__path__ = [], []
        lossp = _stub_module("app.loss")
        lossp.__path__ = []
        sys.modules.update({
            "kornia": _stub_module("kornia", morphology=None), "torchmetrics": tm, "torchmetrics.functional": tmf,
            "torchmetrics.functional.regression": _stub_module("torchmetrics.functional.regression",
                                                               pearson_corrcoef=pearson_corrcoef),
            "app.loss": lossp,
        })
        try:
            spec = importlib.util.spec_from_file_location("app.loss.mono", str(REF_ROOT / "app/loss/mono.py"))
            mod = importlib.util.module_from_spec(spec)
            sys.modules[spec.name] = mod
            spec.loader.exec_module(mod)
            yield mod
        finally:
            for k, v in saved.items():
                if v is None:
                    sys.modules.pop(k, None)
                else:
                    sys.modules[k] = v
