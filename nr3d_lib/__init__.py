"""``nr3d_lib`` import-path shim: the subset of the (un-vendored, CUDA-only) nr3d_lib surface that sits on the
NeuS / StreetSurf render hot path, re-exported from the gfx950 implementation in ``neuralsim_amd`` so that the
reference's ``from nr3d_lib... import ...`` lines on that path resolve unchanged.

Only hot-path symbols are provided.  The harness parts of nr3d_lib (config, attributes, logger, checkpoint,
dataset helpers, GUI) are out of scope of this repository and are NOT shimmed.
"""
