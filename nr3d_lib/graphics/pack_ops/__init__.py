"""``nr3d_lib.graphics.pack_ops`` (reference imports: app/renderers/single_volume_renderer.py:20,
app/renderers/buffer_compose_renderer.py:33, app/renderers/utils.py:15, app/loss/lidar.py:17)."""
from neuralsim_amd.graphics.pack_ops import *  # noqa: F401,F403
from neuralsim_amd.graphics.pack_ops import __all__  # noqa: F401
