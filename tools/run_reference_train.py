#!/usr/bin/env python
"""Run the REFERENCE's trainer -- ``/root/reference/code_single/tools/train.py``, source unchanged, executed with runpy --
on this repository: its ``nr3d_lib`` imports resolve to the shim package of this repository (HIP kernels underneath),
its dataset is ``neuralsim_amd.dataio.SyntheticObjectDataset`` (``--dataset_cfg.target=...`` override: no files).

    python tools/run_reference_train.py [--script code_multi/tools/train.py] --config <reference yaml> [--a.b.c=value ...]

On a machine without a HIP device (the authoring container) pass ``--emulate``: the kernels run on the test-only host
emulator (tests/emu) and every ``cuda`` device the trainer asks for is mapped to the CPU -- the trainer hard-codes
``torch.device('cuda', local_rank)`` (train.py:1204).  Nothing of this is needed on a GPU box.
"""
import os
import runpy
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
REF = Path(os.environ.get("NSIM_REFERENCE_ROOT", "/root/reference"))


def _install_third_party_stubs():
    """Import-time stand-ins for packages the trainer's modules import at the top but the hot path never calls (none is
    installed here): a meta-path finder that serves an inert stub for ``<pkg>`` and any ``<pkg>.sub.module``."""
    import importlib.abc
    import importlib.machinery
    import importlib.util
    import types
    tops = ("imageio", "skimage", "cv2", "open3d", "vedo", "kornia", "lpips", "icecream", "matplotlib", "mediapy",
            "trimesh", "plyfile", "pytorch_msssim", "torchmetrics", "torchvision", "tensorboardX", "ffmpeg", "PIL",
            "pynvml", "tensorboard", "seaborn", "pandas_unused")
    missing = [t for t in tops if importlib.util.find_spec(t) is None]

    class _Stub(types.ModuleType):
        __path__ = []

        def __getattr__(self, k):
            if k.startswith("__"):
                raise AttributeError(k)
            return _inert

    class _Inert:
        def __call__(self, *a, **k):
            return self

        def __getattr__(self, k):
            if k.startswith("__"):
                raise AttributeError(k)
            return self

        def __mro_entries__(self, bases):
            return (object,)
    _inert = _Inert()

    class _Finder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
        def find_spec(self, fullname, path=None, target=None):
            if fullname.split(".")[0] in missing:
                return importlib.machinery.ModuleSpec(fullname, self, is_package=True)
            return None

        def create_module(self, spec):
            return _Stub(spec.name)

        def exec_module(self, module):
            pass
    sys.meta_path.append(_Finder())
    if "torch_scatter" not in sys.modules:
        if importlib.util.find_spec("torch_scatter") is None:
            import torch
            ts = types.ModuleType("torch_scatter")

            def scatter_min(src, index, dim=0, out=None, dim_size=None):
                """(min per index, argmin) along ``dim`` -- what app/renderers/buffer_compose_renderer.py:25 imports."""
                n = int(dim_size if dim_size is not None else (out.shape[dim] if out is not None else int(index.max()) + 1))
                base = out if out is not None else torch.full([n], float("inf"), dtype=src.dtype, device=src.device)
                res = base.scatter_reduce(dim, index, src, reduce="amin", include_self=True)
                hit = src == res.gather(dim, index)
                arg = torch.full([n], src.shape[dim], dtype=torch.long, device=src.device)
                pos = torch.arange(src.shape[dim], device=src.device)
                arg = arg.scatter_reduce(dim, index[hit], pos[hit], reduce="amin", include_self=True)
                if out is not None:
                    out.copy_(res)
                return res, arg
            ts.scatter_min = scatter_min
            sys.modules["torch_scatter"] = ts


def _emulate_cuda_on_cpu():
    """No HIP device: kernels on the host emulator, ``cuda`` devices mapped to the CPU (authoring container only)."""
    import ctypes
    import torch
    sys.path.insert(0, str(ROOT / "tests" / "emu"))
    import build_emu
    from neuralsim_amd import _lib
    lib = _lib.bind(ctypes.CDLL(str(build_emu.build())))
    _lib.get_lib = lambda: lib
    _lib.stream_handle = lambda: 0
    _lib.require_device = lambda t, name="tensor": None
    real = torch.device

    class _Meta(type):
        def __instancecheck__(cls, inst):
            return isinstance(inst, real)

        def __call__(cls, *a, **k):
            if a and isinstance(a[0], str) and a[0].startswith("cuda"):
                return real("cpu")
            return real(*a, **k)

    class device(metaclass=_Meta):
        pass
    torch.device = device
    noop = lambda *a, **k: None       # noqa: E731
    for fn in ("synchronize", "set_device", "empty_cache", "manual_seed", "manual_seed_all", "reset_peak_memory_stats"):
        setattr(torch.cuda, fn, noop)
    torch.cuda.current_device = lambda: 0
    torch.cuda.max_memory_allocated = lambda *a, **k: 0
    torch.cuda.memory_allocated = lambda *a, **k: 0
    _to = torch.Tensor.to

    def to(self, *a, **k):
        a = tuple(real("cpu") if (isinstance(x, str) and x.startswith("cuda")) else x for x in a)
        if isinstance(k.get("device"), str) and k["device"].startswith("cuda"):
            k["device"] = real("cpu")
        return _to(self, *a, **k)
    torch.Tensor.to = to
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.nn.Module.cuda = lambda self, *a, **k: self
    # ``--ddp`` (train.py:1401-1406: DDP(trainer, device_ids=..., output_device=local_rank, find_unused_parameters=True)):
    # the same wrapper over gloo; a CPU module takes no device ids
    import torch.nn.parallel as tnp
    ddp_init = tnp.DistributedDataParallel.__init__

    def ddp_init_cpu(self, module, device_ids=None, output_device=None, **k):
        return ddp_init(self, module, device_ids=None, output_device=None, **k)
    tnp.DistributedDataParallel.__init__ = ddp_init_cpu


def _remember_ddp_modules():
    """``--dump-replica-state DIR``: after the trainer returns, every rank writes the state dict (parameters AND buffers) of the
    module the trainer wrapped in DistributedDataParallel to ``DIR/rank<r>.pt`` -- the replica-consistency check of
    tests/test_reference_train.py::test_reference_trainer_ddp_two_ranks (the trainer itself only checkpoints on rank 0)."""
    import torch.nn.parallel as tnp
    seen = []
    init = tnp.DistributedDataParallel.__init__

    def remembering_init(self, *a, **k):
        init(self, *a, **k)
        seen.append(self)
    tnp.DistributedDataParallel.__init__ = remembering_init
    return seen


def main(argv):
    emulate = "--emulate" in argv
    argv = [a for a in argv if a != "--emulate"]
    dump_dir = None
    if "--dump-replica-state" in argv:
        i = argv.index("--dump-replica-state")
        dump_dir = argv[i + 1]
        argv = argv[:i] + argv[i + 2:]
    sys.path.insert(0, str(ROOT))
    if str(REF) not in sys.path:
        sys.path.append(str(REF))
    _install_third_party_stubs()
    if emulate:
        _emulate_cuda_on_cpu()
    if os.environ.get("NSIM_AUTOGRAD_MT", "0") != "1":
        # this process IS the trainer (one GPU per process): its ``loss.backward()`` runs on the calling thread -- the hand-off
        # to autograd's device thread costs 0.28 ms per step on the Python-side backward functions (neuralsim_amd/__init__.py)
        import torch
        torch.autograd.set_multithreading_enabled(False)
    rel = "code_single/tools/train.py"
    if "--script" in argv:                 # another entry point of the reference (code_multi/tools/train.py), also unchanged
        i = argv.index("--script")
        rel = argv[i + 1]
        argv = argv[:i] + argv[i + 2:]
    script = REF / rel
    assert script.exists(), f"{script} not found (NSIM_REFERENCE_ROOT)"
    sys.argv = [str(script)] + argv
    wrapped = _remember_ddp_modules() if dump_dir else None
    cwd = os.getcwd()
    os.chdir(str(REF))                 # the trainer backs up "./app" etc. relative to the project root
    try:
        runpy.run_path(str(script), run_name="__main__")
    finally:
        os.chdir(cwd)
    if dump_dir:
        import torch
        assert len(wrapped) == 1, f"expected the trainer to build one DistributedDataParallel wrapper, saw {len(wrapped)}"
        ddp = wrapped[0]
        # (named_parameters / named_buffers: the AssetBank's own state_dict refuses a prefix, asset_bank.py:246)
        state = {"param:" + k: v.detach().cpu().clone() for k, v in ddp.module.named_parameters()}
        state.update({"buffer:" + k: v.detach().cpu().clone() for k, v in ddp.module.named_buffers()})
        rank = int(os.environ.get("RANK", "0"))
        os.makedirs(dump_dir, exist_ok=True)
        # what every rank would render with at its NEXT forward: DistributedDataParallel(broadcast_buffers=True, the default
        # the trainer leaves in place) re-broadcasts rank 0's buffers before each forward (collective: every rank calls it)
        if ddp.will_sync_module_buffers():
            ddp._sync_buffers()
        state.update({"synced_buffer:" + k: v.detach().cpu().clone() for k, v in ddp.module.named_buffers()})
        torch.save(dict(state=state, broadcast_buffers=bool(ddp.broadcast_buffers), find_unused_parameters=bool(ddp.find_unused_parameters), world=int(os.environ.get("WORLD_SIZE", "1")),
                        ddp_params=sum(p.numel() for p in ddp.module.parameters() if p.requires_grad)),
                   os.path.join(dump_dir, f"rank{rank}.pt"))


if __name__ == "__main__":
    main(sys.argv[1:])
