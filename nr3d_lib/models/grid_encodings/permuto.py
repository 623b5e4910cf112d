"""``nr3d_lib.models.grid_encodings.permuto`` (reference imports: docs/exps/exp_permuto_3d_modulated.py:29,
docs/exps/permuto_enc_video.py:17)."""
from neuralsim_amd.grid_encodings.permuto import PermutoConfig, PermutoEncoding  # noqa: F401
