"""``nr3d_lib.models.fields.neus`` (reference import: app/models/single/neus.py:24)."""
from neuralsim_amd.fields.neus import LoTDNeuSModel  # noqa: F401
from neuralsim_amd.fields.permuto_neus import PermutoNeuSModel  # noqa: F401


class _NotOnTheHotPath:
    """Names the reference imports next to the hash-grid models (``MlpPENeuSModel``): importable so that
    ``app/models/single/neus.py`` loads unchanged, not constructible -- a positional-encoding MLP field is outside
    this repository (its SDF network is a plain dense MLP: no encoding kernel to speak of)."""
    def __init__(self, *args, **kwargs):
        raise NotImplementedError(f"{type(self).__name__}: the LoTD and permutohedral NeuS models are built for gfx950")


class MlpPENeuSModel(_NotOnTheHotPath):
    pass
