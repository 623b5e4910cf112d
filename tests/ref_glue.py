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
        yield mods
    finally:
        for k in [k for k in sys.modules if k == "app" or k.startswith("app.")]:
            del sys.modules[k]
        sys.modules.update(saved)


class FakeTransform:
    """world_transform of a scene node: rotation [3,3], translation [3], scale (scalar)."""
    def __init__(self, R=None, t=None, s=1.0, device=None):
        self.R = torch.eye(3, device=device) if R is None else R
        self.t = torch.zeros(3, device=device) if t is None else t
        self.s = s

    def rotation(self):
        return self.R

    def rotate(self, v):
        return (self.R * v.unsqueeze(-2)).sum(-1)


class FakeNode:
    def __init__(self, model, class_name, id, world_transform=None):
        self.model, self.class_name, self.id = model, class_name, id
        self.world_transform = world_transform or FakeTransform()
        if not hasattr(model, "id"):
            model.id = f"{class_name}#model"


class FakeScene:
    """The part of ``app.resources.Scene`` that ``SingleVolumeRenderer.ray_query`` touches
    (single_volume_renderer.py:157-263): drawable groups by class name, the device, the image embeddings and
    ``convert_rays_in_node`` (scenes.py:686-708 -- world -> object by the node's rotation / translation / scale)."""
    def __init__(self, device, main_class_name="Main", image_embeddings=None):
        self.device = device
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
