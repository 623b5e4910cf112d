"""``nr3d_lib.models.accelerations`` (reference imports: code_single/tools/render.py:213-220)."""
from neuralsim_amd.fields.neus import OccGridAccel  # noqa: F401
OccGridEma = OccGridAccel
