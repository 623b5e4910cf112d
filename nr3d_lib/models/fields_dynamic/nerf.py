"""Names only (see fields_dynamic/neus.py)."""


class EmerNeRFModel:
    def __init__(self, *a, **k):
        raise NotImplementedError("dynamic (time-conditioned) fields are outside the hot path of BASELINE.json")


EmerNeRFOnlyDynamicModel = EmerNeRFModel
