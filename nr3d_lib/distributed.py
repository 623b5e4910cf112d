"""``nr3d_lib.distributed`` (reference import: code_single/tools/train.py:33-45, used :1195)."""
from neuralsim_amd.distributed import get_rank, get_world_size, init_env, is_master  # noqa: F401
