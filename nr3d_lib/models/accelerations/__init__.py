"""``nr3d_lib.models.accelerations`` (reference imports: code_single/tools/render.py:213-220)."""
from neuralsim_amd.fields.neus import OccGridAccel  # noqa: F401
OccGridEma = OccGridAccel
from neuralsim_amd.fields.batched_neus import OccGridAccelBatched  # noqa: F401,E402  (``occ_grid_batched``)
