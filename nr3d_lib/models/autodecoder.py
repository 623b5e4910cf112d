"""``nr3d_lib.models.autodecoder.AutoDecoderMixin`` (reference import: app/models/asset_base.py:17): the latent-per-instance
machinery of the shared (code_multi) models.  Only the name is needed to import ``app.models.asset_base``; the
conditional generators themselves live in the absent nr3d_lib and are out of scope."""


class AutoDecoderMixin:
    def autodecoder_populate(self, *args, **kwargs):
        raise NotImplementedError("AutoDecoderMixin: latent-conditioned models are outside this repository's scope")
