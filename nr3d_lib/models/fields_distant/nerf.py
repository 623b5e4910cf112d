"""``nr3d_lib.models.fields_distant.nerf`` (reference import: app/models/single/nerf.py:27)."""
from neuralsim_amd.fields.nerf_distant import LoTDNeRFDistantModel  # noqa: F401
from ..fields.neus import _NotOnTheHotPath


class NeRFDistantModel(_NotOnTheHotPath):
    pass


class PermutoNeRFDistantModel(_NotOnTheHotPath):
    pass
