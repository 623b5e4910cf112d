"""Generate tests/golden/renderer_fixture.pt: outputs of the REFERENCE's own renderer source
(/root/reference/app/renderers/single_volume_renderer.py, loaded unchanged by tests/ref_glue.py) driving this
repository's models on the seeded scenarios of tests/renderer_scenario.py.

Runs in the authoring container only (needs /root/reference; no GPU there, so the kernels run on the test-only
emulator build of the same .hip sources, f32 MFMA mode).  The GPU-box test
``tests/test_reference_glue.py::test_mirror_matches_reference_fixture`` replays the renderer MIRROR on the device
against these tensors.   python tests/golden/make_renderer_fixture.py
"""
import ctypes
import sys
from pathlib import Path

import torch

HERE = Path(__file__).resolve().parent
ROOT = HERE.parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "tests"), str(ROOT / "tests" / "emu")]

import build_emu  # noqa: E402
import ref_glue  # noqa: E402
from neuralsim_amd import _lib  # noqa: E402
from renderer_scenario import SCENARIOS, build_scenario, run_reference  # noqa: E402


def main():
    assert ref_glue.reference_available(), "needs /root/reference"
    lib = _lib.bind(ctypes.CDLL(str(build_emu.build())))
    _lib.get_lib = lambda: lib                      # the emulator backend, as tests/conftest.py installs it
    _lib.stream_handle = lambda: 0
    _lib.require_device = lambda t, name="tensor": None
    out = {}
    with ref_glue.reference_renderer_modules() as mods:
        for name in SCENARIOS:
            sc = build_scenario(name, torch.device("cpu"))
            out[name] = run_reference(mods, sc, backward=True)
            r = out[name]
            # large gradients (hash tables, the sky's 256-wide layers) are frozen as every 8th entry + their norm
            r["grads"] = {k: (g if g.numel() <= 8192 else dict(stride=8, sample=g[::8].clone(), norm=float(g.norm())))
                          for k, g in r["grads"].items()}
            print(name, {k: tuple(v.shape) for k, v in r["rendered"].items()}, "grads:", len(r["grads"]))
    import compose_scenario as cs
    with ref_glue.reference_compose_renderer_modules() as mods:      # code_multi: the reference's BufferComposeRenderer
        out["compose"] = cs.run_reference(mods, cs.build(torch.device("cpu")))
    for r in out.values():
        r["grads"] = {k: (g if (isinstance(g, dict) or g.numel() <= 8192) else
                          dict(stride=8, sample=g[::8].clone(), norm=float(g.norm()))) for k, g in r["grads"].items()}

    def own(v):         # torch.save writes whole storages: detach views from the buffers they were sliced out of
        if torch.is_tensor(v):
            return v.clone().contiguous()
        return {k: own(x) for k, x in v.items()} if isinstance(v, dict) else v
    torch.save(own(out), HERE / "renderer_fixture.pt")
    print("wrote", HERE / "renderer_fixture.pt", (HERE / "renderer_fixture.pt").stat().st_size, "bytes")


if __name__ == "__main__":
    main()
