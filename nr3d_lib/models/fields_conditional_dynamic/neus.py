"""``nr3d_lib.models.fields_conditional_dynamic.neus`` (reference import: app/models/shared/batched_dynamic_neus.py): the
time-dependent conditional model of the Pedestrian class (``AD_Dynamic_GenerativePermutoConcatNeuSObj_Decomp``,
all_occ.240201.yaml:507).  Dynamic (4-D, time-conditioned) fields are outside SURVEY sec. 8's hot path: the name imports so that
``app.models.shared`` loads unchanged; constructing it raises."""


class DynamicGenerativePermutoConcatNeuSModel:
    def __init__(self, *a, **k):
        raise NotImplementedError("DynamicGenerativePermutoConcatNeuSModel: time-conditioned fields are out of scope (SURVEY sec. 8)")
