"""``nr3d_lib.models.fields.nerf`` (reference import: app/models/single/nerf.py:26): close-range NeRF models are not on
the NeuS / StreetSurf hot path -- importable names only."""
from .neus import _NotOnTheHotPath


class LoTDNeRFModel(_NotOnTheHotPath):
    pass


class NeRFModel(_NotOnTheHotPath):
    pass
