"""``nr3d_lib.graphics.nerf`` (reference import: app/renderers/single_volume_renderer.py:19)."""
from neuralsim_amd.graphics.nerf import packed_alpha_to_vw, ray_alpha_to_vw  # noqa: F401
