"""``nr3d_lib.models.grid_encodings.lotd`` (reference import: code_single/tools/inspect_rendering.py:50)."""
from neuralsim_amd.grid_encodings.lotd import LoTDConfig, LoTDEncoding, gen_ngp_res  # noqa: F401
