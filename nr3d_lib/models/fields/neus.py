"""``nr3d_lib.models.fields.neus`` (reference import: app/models/single/neus.py:24)."""
from neuralsim_amd.fields.neus import LoTDNeuSModel  # noqa: F401


class _NotOnTheHotPath:
    """Names the reference imports next to LoTDNeuSModel (``MlpPENeuSModel``, ``PermutoNeuSModel``): importable so that
    ``app/models/single/neus.py`` loads unchanged, not constructible -- other encodings are outside this repository."""
    def __init__(self, *args, **kwargs):
        raise NotImplementedError(f"{type(self).__name__}: only the LoTD (hash-grid) NeuS model is built for gfx950")


class MlpPENeuSModel(_NotOnTheHotPath):
    pass


class PermutoNeuSModel(_NotOnTheHotPath):
    pass
