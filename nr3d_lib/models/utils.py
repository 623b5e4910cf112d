"""``nr3d_lib.models.utils.batchify_query`` (reference call site: app/renderers/single_volume_renderer.py:565)."""
from neuralsim_amd.renderers.single_volume_renderer import batchify_query  # noqa: F401
