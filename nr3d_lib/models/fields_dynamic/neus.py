"""Names only: ``app/models/single/__init__.py`` star-imports the dynamic models whenever a static one is imported
(SURVEY Appendix B); no BASELINE config instantiates them."""


class DynamicPermutoConcatNeuSModel:
    def __init__(self, *a, **k):
        raise NotImplementedError("dynamic (time-conditioned) fields are outside the hot path of BASELINE.json")
