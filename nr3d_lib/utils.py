"""``nr3d_lib.utils`` -- the container / conversion helpers the reference's scene graph, data loaders and trainer import
(census: SURVEY.md Appendix B; call sites cited per function).  Implementations are absent (nr3d_lib is an un-vendored
submodule): each helper is restated from its name and from how the reference calls it."""
import importlib
import numbers
import os
from typing import Any, Dict, Iterable, List

import numpy as np
import torch


class IDListedDict:
    """An ordered collection of objects addressable by ``obj.id`` AND by position (app/resources/scenes.py:132-140,
    548-571: ``IDListedDict([root])``, ``d[node.id] = node``, ``for o in d``, ``d[0]``, ``d['cam0']``, ``d.keys()``)."""

    def __init__(self, items: Iterable = ()):
        self._d: Dict[Any, Any] = {}
        for o in items:
            self._d[o.id] = o

    def __class_getitem__(cls, item):      # ``IDListedDict[Scene]`` in annotations
        return cls

    def __getitem__(self, k):
        if isinstance(k, (int, np.integer)) and k not in self._d:
            return list(self._d.values())[k]
        if isinstance(k, slice):
            return IDListedDict(list(self._d.values())[k])
        if isinstance(k, (list, tuple)):        # ``scene.observers[lidar_id]`` with a list of ids (train.py:877-885)
            return [self[i] for i in k]
        return self._d[k]

    def __setitem__(self, k, v):
        self._d[k] = v

    def __delitem__(self, k):
        del self._d[k]

    def __contains__(self, k):
        return k in self._d

    def __iter__(self):
        return iter(list(self._d.values()))

    def __len__(self):
        return len(self._d)

    def __repr__(self):
        return f"IDListedDict({list(self._d.keys())})"

    def append(self, o):
        self._d[o.id] = o

    def keys(self):
        return self._d.keys()

    def values(self):
        return self._d.values()

    def items(self):
        return self._d.items()

    def get(self, k, default=None):
        return self._d.get(k, default)

    def pop(self, k, *a):
        return self._d.pop(k, *a)

    def to_list(self) -> List:
        return list(self._d.values())


def import_str(path: str):
    """'pkg.mod.Name' -> the object (``import_str(cfg.model_class)``, app/resources/asset_bank.py:129-138)."""
    mod, _, name = path.rpartition(".")
    return getattr(importlib.import_module(mod), name)


def is_scalar(x) -> bool:
    if isinstance(x, numbers.Number):
        return True
    if isinstance(x, (np.ndarray, torch.Tensor)):
        return x.ndim == 0
    return False


def torch_dtype(dtype):
    """'half' | 'float' | 'double' | torch.dtype -> torch.dtype (``dtype: half`` of the model configs)."""
    if isinstance(dtype, torch.dtype):
        return dtype
    return {"half": torch.float16, "float16": torch.float16, "float": torch.float32, "float32": torch.float32,
            "double": torch.float64, "float64": torch.float64, "bfloat16": torch.bfloat16}[str(dtype)]


def check_to_torch(x, ref: torch.Tensor = None, dtype=None, device=None) -> torch.Tensor:
    """numpy / list / scalar / tensor -> tensor on (device, dtype) (defaults from ``ref``); None stays None."""
    if x is None:
        return None
    if ref is not None:
        dtype = dtype if dtype is not None else ref.dtype
        device = device if device is not None else ref.device
    if isinstance(x, torch.Tensor):
        return x.to(dtype=dtype if dtype is not None else x.dtype, device=device if device is not None else x.device)
    if isinstance(x, np.ndarray):
        return torch.from_numpy(np.ascontiguousarray(x)).to(dtype=dtype, device=device)
    return torch.tensor(x, dtype=dtype, device=device)


def check_per_batch_tensors(*tensors) -> int:
    """``batch_size = check_per_batch_tensors(ins_inds_per_batch, z_ins_per_batch)`` (app/models/shared/batched_neus.py:154):
    the common leading size of the per-batch-item tensors that are given (None entries are skipped)."""
    sizes = {int(t.shape[0]) for t in tensors if t is not None}
    assert len(sizes) <= 1, f"per-batch tensors differ in their leading size: {sorted(sizes)}"
    return sizes.pop() if sizes else 0


def get_shape(x) -> List[int]:
    """Shape of a tensor / array / nested list; [] for scalars and None."""
    if x is None or isinstance(x, numbers.Number):
        return []
    if isinstance(x, (torch.Tensor, np.ndarray)):
        return list(x.shape)
    if isinstance(x, (list, tuple)):
        return [len(x)] + (get_shape(x[0]) if len(x) else [])
    return []


def cond_mkdir(path: str):
    os.makedirs(path, exist_ok=True)


def nested_dict_items(d: dict, prefix=()):
    """Depth-first ``(*key_path, leaf)`` tuples of a nested dict (``for *k, v in nested_dict_items(ret)``,
    code_single/tools/train.py:341)."""
    for k, v in d.items():
        if isinstance(v, dict):
            yield from nested_dict_items(v, prefix + (k,))
        else:
            yield (*prefix, k, v)


def _index_nested(d, i):
    if isinstance(d, dict):
        return {k: _index_nested(v, i) for k, v in d.items()}
    if isinstance(d, (torch.Tensor, np.ndarray, list, tuple)):
        return d[i]
    return d


def _batch_len(d):
    if isinstance(d, dict):
        for v in d.values():
            n = _batch_len(v)
            if n is not None:
                return n
        return None
    if isinstance(d, (torch.Tensor, np.ndarray)):
        return d.shape[0] if d.ndim > 0 else None
    if isinstance(d, (list, tuple)):
        return len(d)
    return None


def zip_dict(d: dict):
    """Un-collate: yields the nested dict of the i-th batch element for every i (code_single/tools/train.py:1150)."""
    for i in range(_batch_len(d) or 0):
        yield _index_nested(d, i)


def zip_two_nested_dict(a: dict, b: dict):
    """Un-collate two collated nested dicts in lockstep: yields (a_i, b_i) (``sample, ground_truth = next(
    zip_two_nested_dict(sample, ground_truth))  # bs=1``, code_single/tools/train.py:444)."""
    n = _batch_len(a) or _batch_len(b) or 0
    for i in range(n):
        yield _index_nested(a, i), _index_nested(b, i)


def collate_nested_dict(batch: List[dict], stack: bool = True) -> dict:
    """list of (nested) dicts -> (nested) dict of stacked tensors / lists (the data loaders' collate function)."""
    first = batch[0]
    out = {}
    for k, v in first.items():
        vals = [b[k] for b in batch]
        if isinstance(v, dict):
            out[k] = collate_nested_dict(vals, stack=stack)
        elif isinstance(v, torch.Tensor) and stack:
            out[k] = torch.stack(vals, dim=0)
        elif isinstance(v, np.ndarray) and stack:
            out[k] = torch.from_numpy(np.stack(vals, axis=0))
        elif isinstance(v, numbers.Number) and stack:
            out[k] = torch.tensor(vals)
        else:
            out[k] = vals
    return out


def collate_tuple_of_nested_dict(batch: List[tuple], stack: bool = True) -> tuple:
    """list of tuples of nested dicts (``(sample, ground_truth)``) -> tuple of collated nested dicts."""
    return tuple(collate_nested_dict([b[i] for b in batch], stack=stack) for i in range(len(batch[0])))


def pad_images_to_same_size(imgs: List, value=0, batched: bool = False, padding: str = "top_left"):
    """Pad [H,W(,C)] images (tensors or arrays) with ``value`` to the largest H and W; content stays top-left."""
    H = max(i.shape[1 if batched else 0] for i in imgs)
    W = max(i.shape[2 if batched else 1] for i in imgs)
    out = []
    for im in imgs:
        h, w = im.shape[1 if batched else 0], im.shape[2 if batched else 1]
        if isinstance(im, torch.Tensor):
            shape = list(im.shape)
            shape[1 if batched else 0], shape[2 if batched else 1] = H, W
            p = torch.full(shape, value, dtype=im.dtype, device=im.device)
            if batched:
                p[:, :h, :w] = im
            else:
                p[:h, :w] = im
        else:
            shape = list(im.shape)
            shape[1 if batched else 0], shape[2 if batched else 1] = H, W
            p = np.full(shape, value, dtype=im.dtype)
            if batched:
                p[:, :h, :w] = im
            else:
                p[:h, :w] = im
        out.append(p)
    return out


def img_to_torch_and_downscale(img, hw=None, dtype=torch.float32, device=None, downscale: float = 1.0, **unused):
    """[H,W(,C)] array / tensor -> tensor, optionally resized by 1 / downscale (area averaging)."""
    t = check_to_torch(img, dtype=dtype, device=device)
    if downscale != 1.0:
        chw = t.permute(2, 0, 1)[None] if t.dim() == 3 else t[None, None]
        H, W = t.shape[0], t.shape[1]
        nh, nw = max(1, int(round(H / downscale))), max(1, int(round(W / downscale)))
        chw = torch.nn.functional.interpolate(chw.float(), size=(nh, nw), mode="area")
        t = (chw[0].permute(1, 2, 0) if t.dim() == 3 else chw[0, 0]).to(dtype)
    return t


def image_downscale(img, downscale: float = 1.0, **kw):
    return img_to_torch_and_downscale(img, downscale=downscale, **kw)


def get_image_size(img) -> tuple:
    return tuple(img.shape[:2])


def tensor_statistics(t: torch.Tensor, prefix: str = "", metrics=("mean", "std", "min", "max", "absmax")) -> dict:
    t = t.detach().float()
    if t.numel() == 0:
        return {}
    vals = dict(mean=t.mean(), std=t.std() if t.numel() > 1 else t.new_zeros(()), min=t.min(), max=t.max(),
                absmax=t.abs().max(), norm=t.norm())
    pre = f"{prefix}." if prefix else ""
    return {pre + k: float(vals[k]) for k in metrics if k in vals}


def backup_project(backup_dir: str, source_dir: str, subdirs_to_copy: List[str], filetypes_to_copy: List[str]):
    """Copy the project's sources next to an experiment (code_single/tools/train.py:1233-1237)."""
    import shutil
    for sub in subdirs_to_copy:
        src = os.path.join(source_dir, sub)
        if not os.path.isdir(src):
            continue
        for root, _dirs, files in os.walk(src):
            for f in files:
                if os.path.splitext(f)[1] in filetypes_to_copy:
                    dst = os.path.join(backup_dir, os.path.relpath(os.path.join(root, f), source_dir))
                    os.makedirs(os.path.dirname(dst), exist_ok=True)
                    shutil.copyfile(os.path.join(root, f), dst)


def wait_for_pid(pid: int, poll_s: float = 5.0):
    import time
    while os.path.exists(f"/proc/{pid}"):
        time.sleep(poll_s)


def is_file_being_written(path: str, wait_s: float = 0.5) -> bool:
    import time
    s0 = os.path.getsize(path)
    time.sleep(wait_s)
    return os.path.getsize(path) != s0


def glob_imgs(path: str) -> List[str]:
    import glob
    out = []
    for ext in ("*.png", "*.jpg", "*.JPEG", "*.JPG", "*.jpeg"):
        out.extend(glob.glob(os.path.join(path, ext)))
    return sorted(out)


def load_rgb(path: str, downscale: float = 1.0):
    raise NotImplementedError("image files: the synthetic datasets of this repository render their images analytically")


load_mask = cpu_resize = crop_image = load_rgb
