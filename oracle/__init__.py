"""oracle/ -- TEST INFRASTRUCTURE, not product code.

A pure-PyTorch (CPU, f32/f64, autograd-native) restatement of the NeuS / StreetSurf
volume-render hot path of PJLab-ADG/neuralsim (reference @ /root/reference, VERSION 0.6.0).

PARITY UNPINNED (except pack-op bookkeeping): the arithmetic of this path lives in the
un-vendored third-party submodule ``nr3d_lib`` (https://github.com/pjlab-ADG/nr3d_lib,
``/root/reference/.gitmodules:1-3``; directory empty, no pinned commit recoverable, CUDA-only,
no network).  The only reference-side fixture that pins results at this boundary is
``test_multi_buffer_collect_and_merge`` (``app/renderers/buffer_compose_renderer.py:972-1049``),
reproduced in ``tests/golden/pack_fixture.json``.  Everything else restates the published
algorithms (NeuS arXiv 2106.10689 sec. 3, Instant-NGP hash encoding, StreetSurf arXiv
2306.04988) anchored on the reference's own call sites and configs; each function cites the
reference file:line it follows.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this package -- as the checker / the timed CPU baseline, never as the thing shipped.
"""
