"""``nr3d_lib.models.attributes`` -- the typed per-frame attribute tensors of the reference's scene graph
(app/resources/nodes.py:20-60, 283-300, 402-470; app/resources/observers/cameras.py:52-99; usage census: ``mat_4x4``,
``rotation``, ``translation``, ``forward``, ``rotate``, ``vec_3``, ``value``, ``subattr``, ``new``, ``concat``, ``interp1d``,
``intr.{H, W, wh, mat_3x3, lift, proj, set_downscale, get_view_frustum}``).

The implementation is absent (nr3d_lib is an un-vendored submodule); this is the subset the ``code_single`` training path
touches, restated from those call sites: an ``Attr`` is an ``nn.Module`` around ONE tensor ``[*prefix, *shape]`` (prefix =
frames / batch), indexable on the prefix (``attr[fi]`` -> an Attr of the same type: ``SceneNode._slice_at``), with a
class-level default value; ``AttrNested`` groups named Attrs (``frame_data``); ``ObjectWithAttr`` registers every Attr
assigned to it (``named_attrs``) so a node can be frozen at a frame and reset.  Pose composition uses
broadcast-multiply-sum, never mm / bmm / einsum (the reference insists: nodes.py:79-84, cameras.py:355-359).
"""
from typing import Dict, Iterable, List, Tuple

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

__all__ = ["Attr", "AttrNested", "ObjectWithAttr", "Valid", "Scalar", "Scale", "Vector_3", "Vector_4", "make_vector",
           "Translation", "TransformMat4x4", "TransformMat3x4", "TransformRT", "RotationMat3x3", "CameraMatrix3x3",
           "RotationQuaternion", "RotationQuaternionRefinedAdd", "TranslationRefinedAdd", "ScalarRefinedAdd",
           "CameraBase", "PinholeCameraMatHW", "OpenCVCameraMatHW", "FisheyeCameraMatHW", "OrthoCameraIntrinsics",
           "check_to_torch"]


def check_to_torch(x, dtype=None, device=None):
    from nr3d_lib.utils import check_to_torch as c
    return c(x, dtype=dtype, device=device)


class Attr(nn.Module):
    """One typed tensor ``[*prefix, *cls.shape]``."""
    shape: Tuple[int, ...] = ()
    dtype = torch.float32

    @classmethod
    def default_value(cls) -> torch.Tensor:
        return torch.zeros(cls.shape, dtype=cls.dtype)

    def __init__(self, tensor=None, *, learnable: bool = False, dtype=None, device=None):
        super().__init__()
        dt = dtype if dtype is not None else type(self).dtype
        if tensor is None:
            t = self.default_value().to(dtype=dt, device=device)
        else:
            t = check_to_torch(tensor, dtype=dt, device=device)
        self.learnable = bool(learnable)
        if learnable:
            self.tensor = nn.Parameter(t)
        else:
            object.__setattr__(self, "_t", None)
            self.register_buffer("tensor", t, persistent=True)

    # ---- shape helpers
    @property
    def prefix(self) -> Tuple[int, ...]:
        n = len(type(self).shape)
        return tuple(self.tensor.shape[:self.tensor.dim() - n])

    @property
    def device(self):
        return self.tensor.device

    def __len__(self):
        return self.prefix[0] if len(self.prefix) else 0

    def _like(self, t: torch.Tensor):
        o = type(self).__new__(type(self))
        nn.Module.__init__(o)
        o.learnable = False
        o.register_buffer("tensor", t, persistent=True)
        for k, v in self.__dict__.items():       # per-instance extras of subclasses (none are tensors)
            if k not in o.__dict__ and not k.startswith("_"):
                o.__dict__[k] = v
        return o

    def __getitem__(self, i):
        return self._like(self.tensor[i])

    def __setitem__(self, i, v):
        with torch.no_grad():
            self.tensor[i] = v.tensor if isinstance(v, Attr) else check_to_torch(v, dtype=self.tensor.dtype, device=self.tensor.device)

    def new(self, prefix: Iterable[int]):
        """An Attr of this type filled with the default value, with the given prefix (nodes.py:428, 456)."""
        d = self.default_value().to(self.tensor)
        return self._like(d.expand(*tuple(prefix), *d.shape).clone())

    @classmethod
    def concat(cls, attrs: List["Attr"], dim: int = 0):
        return attrs[0]._like(torch.cat([a.tensor for a in attrs], dim=dim))

    @classmethod
    def stack(cls, attrs: List["Attr"], dim: int = 0):
        return attrs[0]._like(torch.stack([a.tensor for a in attrs], dim=dim))

    def value(self) -> torch.Tensor:
        return self.tensor

    def detach(self):
        return self._like(self.tensor.detach())

    def clone(self):
        return self._like(self.tensor.clone())

    def tile(self, prefix):
        t = self.tensor
        return self._like(t.expand(*tuple(prefix), *t.shape[len(self.prefix):]).contiguous())

    def take_along_dim(self, idx: torch.Tensor, dim: int = 0):
        """``torch.take_along_dim`` on the prefix: ``idx`` has the rank of the prefix (``stacked.take_along_dim(li.unsqueeze(0),
        dim=0)[0]`` picks, for every frozen frame, the sensor ``li`` names -- observers/lidars.py:158, cameras.py:494)."""
        t = self.tensor
        tail = t.shape[idx.dim():]
        ie = idx.reshape(*idx.shape, *([1] * len(tail))).expand(*idx.shape, *tail)
        return self._like(torch.take_along_dim(t, ie, dim=dim))

    def interp1d(self, ts_keyframes: torch.Tensor, ts: torch.Tensor):
        """Piecewise-linear interpolation of the per-keyframe values at timestamps ``ts`` (nodes.py:513-518); values
        outside the keyframe range clamp to the ends."""
        T = ts_keyframes.shape[0]
        if T == 1:
            return self._like(self.tensor[torch.zeros_like(ts, dtype=torch.long)])
        idx = torch.searchsorted(ts_keyframes.contiguous(), ts.contiguous(), right=True).clamp(1, T - 1)
        t0, t1 = ts_keyframes[idx - 1], ts_keyframes[idx]
        w = ((ts - t0) / (t1 - t0).clamp_min(1e-12)).clamp(0, 1)
        a, b = self.tensor[idx - 1], self.tensor[idx]
        w = w.reshape(*w.shape, *([1] * (a.dim() - w.dim())))
        if not torch.is_floating_point(a):
            return self._like(torch.where(w < 0.5, a, b))
        return self._like(a + (b - a) * w)

    def extra_repr(self) -> str:
        return f"prefix={list(self.prefix)}"


class Valid(Attr):
    dtype = torch.bool

    @classmethod
    def default_value(cls):
        return torch.ones([], dtype=torch.bool)


class Scalar(Attr):
    def __init__(self, tensor=None, **kw):
        if tensor is not None and "dtype" not in kw:
            t = check_to_torch(tensor)
            kw["dtype"] = t.dtype if not t.dtype == torch.float64 else torch.float32
        super().__init__(tensor, **kw)


def make_vector(n: int):
    """A vector-valued Attr type of length n (``make_vector(distortion.shape[-1])``, cameras.py:88)."""
    return type(f"Vector_{n}", (Attr,), dict(shape=(n,), vec=lambda self: self.tensor))


Vector_3, Vector_4 = make_vector(3), make_vector(4)


class Scale(Attr):
    shape = (3,)

    @classmethod
    def default_value(cls):
        return torch.ones(3)

    def __init__(self, tensor=None, **kw):
        if tensor is not None:
            t = check_to_torch(tensor, dtype=torch.float32)
            if t.dim() == 0 or t.shape[-1] != 3:          # isotropic scales
                t = t.unsqueeze(-1).expand(*t.shape, 3).contiguous()
            tensor = t
        super().__init__(tensor, **kw)

    def vec_3(self) -> torch.Tensor:
        return self.tensor

    def value(self) -> torch.Tensor:
        return self.tensor


class Translation(Attr):
    shape = (3,)

    def vec_3(self):
        return self.tensor

    def translation(self):
        return self.tensor

    def forward(self, x, inv: bool = False):
        return x - self.tensor if inv else x + self.tensor


class RotationMat3x3(Attr):
    shape = (3, 3)

    @classmethod
    def default_value(cls):
        return torch.eye(3)

    def mat_3x3(self):
        return self.tensor

    def rotate(self, x, inv: bool = False):
        R = self.tensor.transpose(-1, -2) if inv else self.tensor
        return (R * x.unsqueeze(-2)).sum(-1)


class TransformMat4x4(Attr):
    """Rigid node-to-parent / node-to-world transform (nodes.py:50-52)."""
    shape = (4, 4)

    @classmethod
    def default_value(cls):
        return torch.eye(4)

    def mat_4x4(self) -> torch.Tensor:
        return self.tensor

    def mat_3x4(self) -> torch.Tensor:
        return self.tensor[..., :3, :]

    def rotation(self) -> torch.Tensor:
        return self.tensor[..., :3, :3]

    def translation(self) -> torch.Tensor:
        return self.tensor[..., :3, 3]

    @staticmethod
    def _over_points(M: torch.Tensor, x: torch.Tensor, tail: int) -> torch.Tensor:
        """x [*prefix, *points, 3] against an attribute of prefix [*prefix]: singleton dims for the point dims (the frustum
        corners of every frame, cameras.py:166-175: pts [F, 8, 3] through a world transform of prefix [F])."""
        extra = (x.dim() - 1) - (M.dim() - tail)
        if extra > 0 and M.dim() > tail:
            M = M.reshape(*M.shape[:M.dim() - tail], *([1] * extra), *M.shape[M.dim() - tail:])
        return M

    def rotate(self, x: torch.Tensor, inv: bool = False) -> torch.Tensor:
        R = self.rotation()
        if inv:
            R = R.transpose(-1, -2)
        return (self._over_points(R, x, 2) * x.unsqueeze(-2)).sum(-1)            # broadcast-multiply-sum (cameras.py:355-359)

    def forward(self, x: torch.Tensor, inv: bool = False) -> torch.Tensor:
        t = self._over_points(self.translation(), x, 1)
        if inv:
            return self.rotate(x - t, inv=True)
        return self.rotate(x) + t


class TransformMat3x4(TransformMat4x4):
    shape = (3, 4)

    @classmethod
    def default_value(cls):
        return torch.eye(4)[:3]

    def mat_4x4(self):
        t = self.tensor
        bottom = t.new_zeros(*t.shape[:-2], 1, 4)
        bottom[..., 0, 3] = 1.0
        return torch.cat([t, bottom], dim=-2)

    def mat_3x4(self):
        return self.tensor


class CameraMatrix3x3(Attr):
    shape = (3, 3)

    @classmethod
    def default_value(cls):
        return torch.eye(3)

    def mat_3x3(self):
        return self.tensor


class AttrNested(nn.Module):
    """Named group of Attrs sharing a prefix (``SceneNode.frame_data``; ``.subattr`` is the name -> Attr mapping)."""

    def __init__(self, allow_new_attr: bool = False, device=None, **attrs):
        super().__init__()
        self.allow_new_attr = allow_new_attr
        self.subattr = _SubAttr()
        for k, v in attrs.items():
            self.subattr[k] = v.to(device) if device is not None else v

    @property
    def prefix(self):
        for v in self.subattr.values():
            return v.prefix
        return ()

    def __len__(self):
        return self.prefix[0] if len(self.prefix) else 0

    def __getattr__(self, k):
        try:
            return super().__getattr__(k)
        except AttributeError:
            sub = self.__dict__.get("_modules", {}).get("subattr")
            if sub is not None and k in sub:
                return sub[k]
            raise

    def __getitem__(self, i):
        return AttrNested(allow_new_attr=self.allow_new_attr, **{k: v[i] for k, v in self.subattr.items()})

    def __setitem__(self, i, other: "AttrNested"):
        for k, v in other.subattr.items():
            self.subattr[k][i] = v

    def new(self, prefix):
        return AttrNested(allow_new_attr=self.allow_new_attr, **{k: v.new(prefix) for k, v in self.subattr.items()})

    def interp1d(self, ts_keyframes, ts):
        return AttrNested(allow_new_attr=self.allow_new_attr,
                          **{k: v.interp1d(ts_keyframes, ts) for k, v in self.subattr.items()})


class _SubAttr(nn.ModuleDict):
    def __getattr__(self, k):
        try:
            return super().__getattr__(k)
        except AttributeError:
            mods = self.__dict__.get("_modules", {})
            if k in mods:
                return mods[k]
            raise


class ObjectWithAttr:
    """A plain object (NOT an ``nn.Module``: models keep references to scene nodes -- ``LoTDNeRFDistant.cr_obj``,
    app/models/single/nerf.py:165 -- which must not become sub-modules of the model) that keeps a registry of the Attrs
    assigned to it (``named_attrs``), so that a scene node can be frozen at a frame (``setattr(node, k,
    frame_data[k][i])``) and reset to its defaults (nodes.py:112-120, 470-482), with the ``to`` / ``_apply`` surface
    the nodes extend (nodes.py:543-560)."""

    def __init__(self, device=None, dtype=torch.float):
        object.__setattr__(self, "_attrs", {})
        self.device, self.dtype = device, dtype

    def __setattr__(self, name, value):
        if isinstance(value, (Attr, AttrNested)) and name != "frame_data":
            self.__dict__.setdefault("_attrs", {})[name] = value
            self.__dict__.pop(name, None)
            return
        attrs = self.__dict__.get("_attrs")
        if attrs is not None and name in attrs:
            del attrs[name]
        object.__setattr__(self, name, value)

    def __getattr__(self, name):
        attrs = self.__dict__.get("_attrs")
        if attrs is not None and name in attrs:
            return attrs[name]
        raise AttributeError(f"{type(self).__name__!r} object has no attribute {name!r}")

    def named_attrs(self):
        return list(self.__dict__.get("_attrs", {}).items())

    def _reset(self):
        for k, v in self.named_attrs():
            if not isinstance(v, AttrNested):
                self._attrs[k] = type(v)(device=self.device)

    def _apply(self, fn):
        for v in self.__dict__.get("_attrs", {}).values():
            v._apply(fn)
        return self

    def to(self, *args, **kwargs):
        for v in self.__dict__.get("_attrs", {}).values():
            v.to(*args, **kwargs)
        dev = kwargs.get("device", next((a for a in args if isinstance(a, (str, torch.device))), None))
        if dev is not None:
            object.__setattr__(self, "device", torch.device(dev) if isinstance(dev, str) else dev)
        return self

    def cuda(self, device=None):
        return self.to(torch.device("cuda", device or 0))

    def cpu(self):
        return self.to(torch.device("cpu"))

    def float(self):
        return self._apply(lambda t: t.float() if t.is_floating_point() else t)

    def double(self):
        return self._apply(lambda t: t.double() if t.is_floating_point() else t)

    def half(self):
        return self._apply(lambda t: t.half() if t.is_floating_point() else t)

    def parameters(self):
        for v in self.__dict__.get("_attrs", {}).values():
            yield from v.parameters()


# ------------------------------------------------------------------------------------------------ camera intrinsics
class CameraBase(AttrNested):
    """Camera intrinsics: a 3x3 matrix + image height / width per frame (cameras.py:52-99)."""
    model = "base"

    def __init__(self, mat: CameraMatrix3x3 = None, H=None, W=None, distortion: Attr = None, device=None, **unused):
        attrs = {}
        if mat is not None:
            attrs["mat"] = mat
            attrs["hw"] = make_vector(2)(torch.stack([check_to_torch(H, dtype=torch.float32),
                                                      check_to_torch(W, dtype=torch.float32)], dim=-1))
        elif type(self).__dict__.get("_defaults", True):      # an un-populated camera (``CameraBase(device=)``): identity
            attrs["mat"] = CameraMatrix3x3()
            attrs["hw"] = make_vector(2)(torch.ones(2))
        if distortion is not None:
            attrs["distortion"] = distortion
        super().__init__(allow_new_attr=True, device=device, **attrs)
        self.downscale = 1.0

    def __getitem__(self, i):
        o = type(self)(device=None)
        for k, v in self.subattr.items():
            o.subattr[k] = v[i]
        o.downscale = self.downscale
        return o

    def new(self, prefix):
        o = type(self)(device=None)
        for k, v in self.subattr.items():
            o.subattr[k] = v.new(prefix)
        o.downscale = self.downscale
        return o

    def interp1d(self, ts_keyframes, ts):
        o = type(self)(device=None)
        for k, v in self.subattr.items():
            o.subattr[k] = v.interp1d(ts_keyframes, ts)
        o.downscale = self.downscale
        return o

    @classmethod
    def stack(cls, cams: List["CameraBase"], dim: int = 0):
        """``type(cams[0].intr).stack(intrs)`` (app/resources/observers/cameras.py:479: MultiCamBundle): a new leading
        prefix dimension over the cameras."""
        o = cls(device=None)
        for k in cams[0].subattr.keys():
            o.subattr[k] = type(cams[0].subattr[k]).stack([c.subattr[k] for c in cams], dim=dim)
        o.downscale = cams[0].downscale
        return o

    def take_along_dim(self, idx: torch.Tensor, dim: int = 0):
        o = type(self)(device=None)
        for k, v in self.subattr.items():
            o.subattr[k] = v.take_along_dim(idx, dim=dim)
        o.downscale = self.downscale
        return o

    def set_downscale(self, downscale):
        """``image_downscale`` = (w, h) ratio old / new of the images actually trained on (a scalar, or the 2-vector of
        dataio/data_loader/base_loader.py:355-384); intrinsics and H / W follow it."""
        d = torch.as_tensor(downscale, dtype=torch.float32).reshape(-1)
        self.downscale = (float(d[0]), float(d[-1]))          # (w, h)

    def _ds(self):
        d = self.downscale
        return (float(d), float(d)) if not isinstance(d, tuple) else d

    @property
    def H(self) -> torch.Tensor:
        return torch.round(self.subattr["hw"].tensor[..., 0] / self._ds()[1]).long()

    @property
    def W(self) -> torch.Tensor:
        return torch.round(self.subattr["hw"].tensor[..., 1] / self._ds()[0]).long()

    def wh(self) -> torch.Tensor:
        return torch.stack([self.W, self.H], dim=-1)

    def unscaled_wh(self) -> torch.Tensor:
        hw = self.subattr["hw"].tensor
        return torch.stack([hw[..., 1], hw[..., 0]], dim=-1).long()

    def mat_3x3(self) -> torch.Tensor:
        m = self.subattr["mat"].tensor
        dw, dh = self._ds()
        if dw != 1.0 or dh != 1.0:
            s = m.new_tensor([1.0 / dw, 1.0 / dh, 1.0])
            m = m * s[:, None]
        return m

    def mat_4x4(self) -> torch.Tensor:
        m = self.mat_3x3()
        out = torch.eye(4, dtype=m.dtype, device=m.device).expand(*m.shape[:-2], 4, 4).clone()
        out[..., :3, :3] = m
        return out

    def focal(self) -> torch.Tensor:
        m = self.mat_3x3()
        return torch.stack([m[..., 0, 0], m[..., 1, 1]], dim=-1)

    def lift(self, u, v, d) -> torch.Tensor:
        from nr3d_lib.graphics.cameras import pinhole_lift
        m = self.mat_3x3()
        if m.dim() > 2:
            m = m.reshape(*m.shape[:-2], *([1] * (u.dim() - (m.dim() - 2))), 3, 3)
        return pinhole_lift(u, v, d, m)

    def proj(self, xyz: torch.Tensor):
        """camera-frame points [..., 3] -> (u, v, depth)."""
        m = self.mat_3x3()
        if m.dim() > 2:
            m = m.reshape(*m.shape[:-2], *([1] * (xyz.dim() - 1 - (m.dim() - 2))), 3, 3)
        x, y, z = xyz[..., 0], xyz[..., 1], xyz[..., 2]
        zs = torch.where(z.abs() < 1e-9, torch.full_like(z, 1e-9), z)
        u = m[..., 0, 0] * x / zs + m[..., 0, 1] * y / zs + m[..., 0, 2]
        v = m[..., 1, 1] * y / zs + m[..., 1, 2]
        return u, v, z

    def get_view_frustum(self, c2w: torch.Tensor, near=None, far=None) -> torch.Tensor:
        from nr3d_lib.graphics.cameras import pinhole_view_frustum
        W, H = self.W.to(c2w.dtype), self.H.to(c2w.dtype)
        return pinhole_view_frustum(c2w, self.mat_3x3(), H, W, near=near, far=far)


class PinholeCameraMatHW(CameraBase):
    model = "pinhole"


class OpenCVCameraMatHW(CameraBase):
    model = "opencv"

    N_ITERS = 5          # cv::undistortPoints' default, and the HIP ray generator's (csrc/sampling.hip raygen_lift)

    def _dist(self, like: torch.Tensor) -> torch.Tensor:
        dd = self.subattr["distortion"].tensor
        if dd.shape[-1] < 5:
            dd = torch.cat([dd, dd.new_zeros(*dd.shape[:-1], 5 - dd.shape[-1])], dim=-1)
        if dd.dim() > 1:
            dd = dd.reshape(*dd.shape[:-1], *([1] * (like.dim() - (dd.dim() - 1))), dd.shape[-1])
        return dd

    def lift(self, u, v, d):
        """pixel (u, v) at depth d -> camera-frame point: the pinhole coordinates are the DISTORTED ones, the undistorted
        (x, y) come from ``N_ITERS`` rounds of the fixed-point iteration of cv::undistortPoints on (k1, k2, p1, p2, k3) --
        the arithmetic of the HIP ray generator, operation for operation (``cameras.py:84-87, 281-310`` call this for
        ``camera_model: opencv``)."""
        m = self.mat_3x3()
        if m.dim() > 2:
            m = m.reshape(*m.shape[:-2], *([1] * (u.dim() - (m.dim() - 2))), 3, 3)
        x0, y0 = (u - m[..., 0, 2]) / m[..., 0, 0], (v - m[..., 1, 2]) / m[..., 1, 1]
        dd = self._dist(u)
        k1, k2, p1, p2, k3 = dd[..., 0], dd[..., 1], dd[..., 2], dd[..., 3], dd[..., 4]
        x, y = x0, y0
        for _ in range(self.N_ITERS):
            r2 = x * x + y * y
            icd = 1.0 / (1.0 + ((k3 * r2 + k2) * r2 + k1) * r2)
            dx = 2.0 * p1 * x * y + p2 * (r2 + 2.0 * x * x)
            dy = p1 * (r2 + 2.0 * y * y) + 2.0 * p2 * x * y
            x, y = (x0 - dx) * icd, (y0 - dy) * icd
        return torch.stack([x * d, y * d, d], dim=-1)

    def proj(self, xyz: torch.Tensor):
        """camera-frame points -> (u, v, depth) through the forward distortion model."""
        m = self.mat_3x3()
        if m.dim() > 2:
            m = m.reshape(*m.shape[:-2], *([1] * (xyz.dim() - 1 - (m.dim() - 2))), 3, 3)
        z = xyz[..., 2]
        zs = torch.where(z.abs() < 1e-9, torch.full_like(z, 1e-9), z)
        x, y = xyz[..., 0] / zs, xyz[..., 1] / zs
        dd = self._dist(x)
        k1, k2, p1, p2, k3 = dd[..., 0], dd[..., 1], dd[..., 2], dd[..., 3], dd[..., 4]
        r2 = x * x + y * y
        cd = 1.0 + ((k3 * r2 + k2) * r2 + k1) * r2
        xd = x * cd + 2.0 * p1 * x * y + p2 * (r2 + 2.0 * x * x)
        yd = y * cd + p1 * (r2 + 2.0 * y * y) + 2.0 * p2 * x * y
        return m[..., 0, 0] * xd + m[..., 0, 1] * yd + m[..., 0, 2], m[..., 1, 1] * yd + m[..., 1, 2], z


class FisheyeCameraMatHW(OpenCVCameraMatHW):
    """``camera_model: fisheye`` (cameras.py:88-92): the OpenCV fisheye (Kannala-Brandt equidistant) model the reference
    applies in app/resources/observers/fisheye.py:31-42, theta_d = theta (1 + k1 theta^2 + k2 theta^4 + k3 theta^6 + k4
    theta^8) with distortion = (k1, k2, k3, k4).  (The class lives in the absent nr3d_lib: semantics fixed here, the
    arithmetic of the HIP ray generator -- csrc/sampling.hip raygen_lift_fisheye.)"""
    model = "fisheye"

    N_ITERS = 10         # cv::fisheye::undistortPoints' count, and the HIP ray generator's default

    def _dist(self, like: torch.Tensor) -> torch.Tensor:
        dd = self.subattr["distortion"].tensor
        if dd.shape[-1] < 4:
            dd = torch.cat([dd, dd.new_zeros(*dd.shape[:-1], 4 - dd.shape[-1])], dim=-1)
        if dd.dim() > 1:
            dd = dd.reshape(*dd.shape[:-1], *([1] * (like.dim() - (dd.dim() - 1))), dd.shape[-1])
        return dd

    def lift(self, u, v, d):
        """pixel (u, v) -> d x the UNIT direction of its ray in the camera frame, (sin theta x_d / theta_d, sin theta y_d /
        theta_d, cos theta) (a lens beyond 90 degrees has no point at z = d; the reference normalises the lifted directions,
        cameras.py:355-359)."""
        m = self.mat_3x3()
        if m.dim() > 2:
            m = m.reshape(*m.shape[:-2], *([1] * (u.dim() - (m.dim() - 2))), 3, 3)
        xd, yd = (u - m[..., 0, 2]) / m[..., 0, 0], (v - m[..., 1, 2]) / m[..., 1, 1]
        dd = self._dist(u)
        k1, k2, k3, k4 = dd[..., 0], dd[..., 1], dd[..., 2], dd[..., 3]
        td = torch.sqrt(xd * xd + yd * yd)
        th = td
        for _ in range(self.N_ITERS):
            t2 = th * th
            f = th * (1.0 + (((k4 * t2 + k3) * t2 + k2) * t2 + k1) * t2) - td
            fp = 1.0 + (((9.0 * k4 * t2 + 7.0 * k3) * t2 + 5.0 * k2) * t2 + 3.0 * k1) * t2
            th = th - f / fp
        sc = torch.where(td > 1e-8, torch.sin(th) / td.clamp_min(1e-12), torch.ones_like(td))
        return torch.stack([xd * sc * d, yd * sc * d, torch.cos(th) * d], dim=-1)

    def proj(self, xyz: torch.Tensor):
        """camera-frame points -> (u, v, depth): theta = atan2(|xy|, z) through the forward polynomial."""
        m = self.mat_3x3()
        if m.dim() > 2:
            m = m.reshape(*m.shape[:-2], *([1] * (xyz.dim() - 1 - (m.dim() - 2))), 3, 3)
        x, y, z = xyz[..., 0], xyz[..., 1], xyz[..., 2]
        r = torch.sqrt(x * x + y * y)
        th = torch.atan2(r, z)
        dd = self._dist(x)
        k1, k2, k3, k4 = dd[..., 0], dd[..., 1], dd[..., 2], dd[..., 3]
        t2 = th * th
        thd = th * (1.0 + (((k4 * t2 + k3) * t2 + k2) * t2 + k1) * t2)
        sc = torch.where(r > 1e-12, thd / r.clamp_min(1e-12), torch.ones_like(r))
        return m[..., 0, 0] * (x * sc) + m[..., 0, 2], m[..., 1, 1] * (y * sc) + m[..., 1, 2], z


class OrthoCameraIntrinsics(CameraBase):
    model = "ortho"


def quat_to_mat(q: torch.Tensor) -> torch.Tensor:
    """unit quaternion (w, x, y, z) [..., 4] -> rotation matrix [..., 3, 3]"""
    w, x, y, z = q.unbind(-1)
    return torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y),
                        2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x),
                        2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)], dim=-1).reshape(*q.shape[:-1], 3, 3)


def mat_to_quat(R: torch.Tensor) -> torch.Tensor:
    """rotation matrix [..., 3, 3] -> unit quaternion (w, x, y, z), w >= 0 (the branch-free 'largest component' form)."""
    m00, m11, m22 = R[..., 0, 0], R[..., 1, 1], R[..., 2, 2]
    q_abs = torch.sqrt(torch.clamp(torch.stack([1 + m00 + m11 + m22, 1 + m00 - m11 - m22, 1 - m00 + m11 - m22,
                                                1 - m00 - m11 + m22], dim=-1), min=0.0))
    cand = torch.stack([
        torch.stack([q_abs[..., 0] ** 2, R[..., 2, 1] - R[..., 1, 2], R[..., 0, 2] - R[..., 2, 0], R[..., 1, 0] - R[..., 0, 1]], -1),
        torch.stack([R[..., 2, 1] - R[..., 1, 2], q_abs[..., 1] ** 2, R[..., 1, 0] + R[..., 0, 1], R[..., 0, 2] + R[..., 2, 0]], -1),
        torch.stack([R[..., 0, 2] - R[..., 2, 0], R[..., 1, 0] + R[..., 0, 1], q_abs[..., 2] ** 2, R[..., 2, 1] + R[..., 1, 2]], -1),
        torch.stack([R[..., 1, 0] - R[..., 0, 1], R[..., 2, 0] + R[..., 0, 2], R[..., 2, 1] + R[..., 1, 2], q_abs[..., 3] ** 2], -1),
    ], dim=-2) / (2.0 * q_abs[..., None].clamp_min(0.1))
    best = q_abs.argmax(dim=-1)
    q = torch.gather(cand, -2, best[..., None, None].expand(*best.shape, 1, 4)).squeeze(-2)
    q = torch.where(q[..., :1] < 0, -q, q)
    return F.normalize(q, dim=-1)


class RotationQuaternion(Attr):
    """Unit quaternion (w, x, y, z) (app/models/scene/learnable_params.py:101: ``RotationQuaternion.from_mat_3x3(...)``)."""
    shape = (4,)

    @classmethod
    def default_value(cls):
        return torch.tensor([1.0, 0.0, 0.0, 0.0])

    @classmethod
    def from_mat_3x3(cls, R: torch.Tensor, **kw):
        return cls(mat_to_quat(R.detach()), **kw)

    def quat(self) -> torch.Tensor:
        return self.tensor

    def mat_3x3(self) -> torch.Tensor:
        return quat_to_mat(F.normalize(self.tensor, dim=-1))


class _RefinedAdd(nn.Module):
    """``<X>RefinedAdd(attr0=<X>, delta=<Vector>(zeros, learnable=True))`` (learnable_params.py:100-109): the node keeps its
    dataset value ``attr0`` and learns an additive correction; the sum behaves as an <X>.  The two parts are reachable as
    ``.subattr.attr0`` / ``.subattr.delta`` (learnable_params.py:318, 337).  Slicing on the prefix returns a plain <X>
    holding the sliced sum (gradients flow to ``delta``)."""

    def __init__(self, attr0: Attr = None, delta: Attr = None):
        super().__init__()
        self.subattr = _SubAttr()
        self.subattr["attr0"], self.subattr["delta"] = attr0, delta

    @property
    def attr0(self):
        return self.subattr["attr0"]

    @property
    def delta(self):
        return self.subattr["delta"]

    @property
    def tensor(self) -> torch.Tensor:
        return self.attr0.tensor.detach() + self.delta.tensor

    @property
    def prefix(self):
        return self.attr0.prefix

    @property
    def device(self):
        return self.attr0.tensor.device

    def __len__(self):
        return len(self.attr0)

    def _plain(self, t):
        return self.attr0._like(t)

    def __getitem__(self, i):
        return self._plain(self.tensor[i])

    def interp1d(self, ts_keyframes, ts):
        return self._plain(self.tensor).interp1d(ts_keyframes, ts)

    def value(self):
        return self.tensor


class RotationQuaternionRefinedAdd(_RefinedAdd):
    def quat(self):
        return F.normalize(self.tensor, dim=-1)

    def mat_3x3(self):
        return quat_to_mat(self.quat())


class TranslationRefinedAdd(_RefinedAdd):
    def vec_3(self):
        return self.tensor

    def translation(self):
        return self.tensor

    def forward(self, x, inv: bool = False):
        return x - self.tensor if inv else x + self.tensor


class ScalarRefinedAdd(_RefinedAdd):
    pass


class TransformRT(AttrNested):
    """Rotation + translation as two Attrs (``learnable_params`` builds refined poses this way)."""

    def __init__(self, rot: Attr = None, trans: Attr = None, device=None, **kw):
        super().__init__(allow_new_attr=True, device=device, rot=rot if rot is not None else RotationMat3x3(),
                         trans=trans if trans is not None else Translation())

    def _sub(self, rot, trans):
        o = TransformRT.__new__(TransformRT)
        AttrNested.__init__(o, allow_new_attr=True, rot=rot, trans=trans)
        return o

    def __getitem__(self, i):
        return self._sub(self.subattr["rot"][i], self.subattr["trans"][i])

    def new(self, prefix):
        return TransformMat4x4().new(prefix)

    def interp1d(self, ts_keyframes, ts):
        return self._sub(self.subattr["rot"].interp1d(ts_keyframes, ts), self.subattr["trans"].interp1d(ts_keyframes, ts))

    @property
    def prefix(self):
        return self.subattr["trans"].prefix

    def rotation(self):
        return self.subattr["rot"].mat_3x3()

    def translation(self):
        return self.subattr["trans"].vec_3()

    def mat_4x4(self):
        R, t = self.rotation(), self.translation()
        top = torch.cat([R, t.unsqueeze(-1)], dim=-1)
        bottom = top.new_zeros(*top.shape[:-2], 1, 4)
        bottom[..., 0, 3] = 1.0
        return torch.cat([top, bottom], dim=-2)

    def mat_3x4(self):
        return torch.cat([self.rotation(), self.translation().unsqueeze(-1)], dim=-1)

    def rotate(self, x, inv=False):
        R = self.rotation().transpose(-1, -2) if inv else self.rotation()
        return (TransformMat4x4._over_points(R, x, 2) * x.unsqueeze(-2)).sum(-1)

    def forward(self, x, inv=False):
        t = TransformMat4x4._over_points(self.translation(), x, 1)
        if inv:
            return self.rotate(x - t, inv=True)
        return self.rotate(x) + t
