"""``nr3d_lib.models.fields.neus`` (reference import: app/models/single/neus.py:24)."""
from neuralsim_amd.fields.neus import LoTDNeuSModel  # noqa: F401
