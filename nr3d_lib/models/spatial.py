"""``nr3d_lib.models.spatial.AABBSpace`` (reference imports: app/models/asset_base.py:15, app/resources/nodes.py).
The block / forest spaces of the large-scene models are outside the hot path."""
from neuralsim_amd.spatial import AABBSpace  # noqa: F401
