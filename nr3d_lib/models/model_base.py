"""``nr3d_lib.models.model_base.ModelMixin`` (reference import: app/models/asset_base.py:16)."""
from neuralsim_amd.model_base import ModelMixin  # noqa: F401
