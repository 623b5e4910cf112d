"""``nr3d_lib.models.fields_conditional.sdf`` (reference import census, SURVEY appendix B: ``StyleLoTDSDF``): the SDF
network of the StyleLoTD model is part of ``StyleLoTDNeuSModel`` here (grower + fused decoder kernels), not a module of its
own; the name is importable, constructing it is not supported."""


class StyleLoTDSDF:
    def __init__(self, *a, **k):
        raise NotImplementedError("StyleLoTDSDF: use StyleLoTDNeuSModel (the SDF network is fused into its kernels)")
