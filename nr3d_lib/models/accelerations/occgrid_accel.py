"""``nr3d_lib.models.accelerations.occgrid_accel`` (app/models/single/neus.py: ``from ...occgrid_accel import OccGridAccel``)."""
from neuralsim_amd.fields.neus import OccGridAccel  # noqa: F401
