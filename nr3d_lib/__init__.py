"""``nr3d_lib`` import-path shim: the subset of the (un-vendored, CUDA-only) nr3d_lib surface the reference's NeuS /
StreetSurf training path imports, so that its ``from nr3d_lib... import ...`` lines resolve unchanged (SURVEY.md sec. 8b,
INTEGRATION.md).  Two kinds of modules live here:

* hot-path names -- ``graphics.{nerf, pack_ops}``, ``models.{fields, fields_distant, fields_conditional, accelerations,
  grid_encodings, spatial, model_base, autodecoder}`` -- re-exported from (or thin classes over) the gfx950 implementation in
  ``neuralsim_amd``;
* the harness the reference's trainers need around them (since round 3) -- ``config`` (YAML + ``${...}`` interpolation +
  ``BaseConfig``), ``models.attributes`` (typed per-frame tensors of the scene graph, camera models, pose-refinement types),
  ``utils``, ``checkpoint``, ``logger``, ``fmt``, ``plot``, ``maths``, ``graphics.{cameras, utils}``, ``models.{importance,
  embeddings, embedders, blocks, loss, annealers, utils}`` -- restated from the reference's call sites (the implementation is
  absent), plain PyTorch on the host side.  With it ``code_single/tools/train.py`` (object and street configs) and
  ``code_multi/tools/train.py`` (multi-object config) run UNCHANGED: tests/test_reference_train.py.

Names the reference imports for paths outside SURVEY sec. 8 (dynamic / time-conditioned fields, forest blocks, GUI) import
as placeholders that raise on construction.
"""

import neuralsim_amd  # noqa: F401,E402
