"""``nr3d_lib.graphics.pack_ops.pack_ops`` -- the submodule spelling used by app/loss/eikonal.py:22."""
from neuralsim_amd.graphics.pack_ops import *  # noqa: F401,F403
from neuralsim_amd.graphics.pack_ops import __all__  # noqa: F401
