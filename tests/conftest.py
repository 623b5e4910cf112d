"""Test harness.

Two backends run the SAME test bodies through the SAME host code (neuralsim_amd/*):
  * ``hip``  -- the product: libnsim_hip.so on a real MI355X (marked ``gpu``);
  * ``emu``  -- the identical kernel sources compiled for the host against the test-only SIMT emulator
                (tests/emu/), injected by monkeypatching the three loader hooks of ``neuralsim_amd._lib``.
                This exists so kernel logic can be checked against the oracle on a CPU-only machine; the
                product itself has no such path (``_lib.get_lib`` only ever loads libnsim_hip.so).
The checker is always ``oracle/`` (pure PyTorch on CPU).
"""
import ctypes
import os
import sys
from pathlib import Path

import pytest
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

BACKENDS = ["emu", pytest.param("hip", marks=pytest.mark.gpu)]


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: long-running emulator test")


@pytest.fixture(scope="session")
def emu_lib():
    sys.path.insert(0, str(ROOT / "tests" / "emu"))
    import build_emu
    from neuralsim_amd import _lib
    path = build_emu.build()
    return _lib.bind(ctypes.CDLL(str(path)))


@pytest.fixture
def backend(request, monkeypatch):
    """-> torch.device the tensors must live on for the selected backend."""
    from neuralsim_amd import _lib
    kind = request.param
    if kind == "emu":
        lib = request.getfixturevalue("emu_lib")
        monkeypatch.setattr(_lib, "get_lib", lambda: lib)
        monkeypatch.setattr(_lib, "stream_handle", lambda: 0)
        monkeypatch.setattr(_lib, "require_device", lambda t, name="tensor": None)
        return torch.device("cpu")
    assert torch.cuda.is_available(), "gpu test selected but no HIP device is visible"
    _lib.get_lib()
    return torch.device("cuda", 0)


@pytest.fixture(autouse=True)
def poisoned_empty(monkeypatch, request):
    """Every test runs with ``torch.empty`` / ``empty_like`` handing out NaN-filled float buffers: a kernel that reads
    a buffer (or a plane level) nothing wrote turns its outputs into NaN instead of passing on whatever the allocator
    happened to hold.  (Found this way in round 3: the distant backward read feature planes past the pyramid's levels
    after the gather had become level-major -- 0-weighted garbage, NaN on an unlucky allocation.)  NSIM_TEST_NO_POISON=1: off."""
    # (the full-size GPU tests allocate GBs of buffers per test: filling them all would double the suite's run time -- the
    # same code paths are poisoned at emulator size)
    if os.environ.get("NSIM_TEST_NO_POISON") == "1" or request.module.__name__.startswith("test_fullsize"):
        yield
        return
    real, real_like = torch.empty, torch.empty_like

    def _fill(t):
        if t.is_floating_point() and t.numel():
            t.fill_(float("nan"))
        return t
    monkeypatch.setattr(torch, "empty", lambda *a, **k: _fill(real(*a, **k)))
    monkeypatch.setattr(torch, "empty_like", lambda *a, **k: _fill(real_like(*a, **k)))
    yield


def _needs_reference(metafunc) -> bool:
    """tests that execute the reference's own sources from /root/reference (``needs_reference`` skipif markers)"""
    for m in metafunc.definition.iter_markers("skipif"):
        if "/root/reference" in str(m.kwargs.get("reason", "")):
            return True
    return False


def pytest_generate_tests(metafunc):
    if "backend" in metafunc.fixturenames:
        # A test that runs the reference's sources can only run where /root/reference exists -- the authoring container, which
        # has no GPU -- and the GPU box has no reference: its ``hip`` variant could never execute anywhere (22 permanent skips
        # of the round-4 GPU run).  Such tests get the emulator backend only; what they pin is replayed on the GPU box from
        # frozen outputs of the reference (tests/test_reference_frozen.py, the renderer / compose / pack fixtures).
        backends = ["emu"] if _needs_reference(metafunc) else BACKENDS
        metafunc.parametrize("backend", backends, indirect=True)


# Collection order = SURVEY.md section 8's rows, innermost first: the unit parity tests of every kernel family, then the full-size
# parity tests, then the replays of the reference's own frozen outputs, then renderer / scene composition, and the multi-step
# trainer and multi-process tests last.  The driver runs ``pytest -m gpu -x``: with the alphabetical default one failing
# trainer test (round 5: test_permuto.py, collected before test_sampling.py .. test_sky.py) hid 69 parity tests behind it.
_FILE_ORDER = [
    ("test_abi",),
    ("test_sampling", "test_sampling_fuzz", "test_field", "test_pack_ops", "test_pack_ops_fuzz", "test_sky", "test_distant",
     "test_permuto", "test_optim", "test_losses", "test_memory"),
    ("test_ray_query", "test_fullsize_parity", "test_fullsize_properties", "test_fullsize_configs"),
    ("test_reference_frozen", "test_reference_glue", "test_reference_models", "test_reference_configs", "test_shim"),
    ("test_renderer", "test_compose", "test_batched", "test_convergence"),
    ("test_trainer", "test_reference_train", "test_distributed"),
]
_FILE_RANK = {name: (rank, i) for rank, names in enumerate(_FILE_ORDER) for i, name in enumerate(names)}
_LAST = (len(_FILE_ORDER) - 1, 0)
# multi-step trainer tests that live in a kernel family's file run with the trainers (NOT the one-step oracle comparisons)
_TRAINER_WORDS = ("training_steps", "train_steps", "pretrain", "_trains", "fused_step_equals_autograd", "soak", "converge")


def pytest_collection_modifyitems(session, config, items):
    def key(item):
        mod = item.module.__name__.rsplit(".", 1)[-1]
        rank = _FILE_RANK.get(mod, (len(_FILE_ORDER), 0))
        if rank < _LAST and any(w in item.name.split("[")[0] for w in _TRAINER_WORDS):
            rank = _LAST
        return rank
    items.sort(key=key)                  # stable: the order inside a file (and of its parametrisations) is kept


def sync(device):
    if device.type == "cuda":
        torch.cuda.synchronize()
