"""``nr3d_lib.models.accelerations`` (reference imports: code_single/tools/render.py:213-220)."""
from neuralsim_amd.fields.neus import OccGridAccel  # noqa: F401
OccGridEma = OccGridAccel
from neuralsim_amd.fields.batched_neus import OccGridAccelBatched  # noqa: F401,E402  (``occ_grid_batched``)


OccGridEmaBatched = OccGridAccelBatched_Ema = OccGridAccelBatched


class _NotBuilt:
    """Dynamic (time-dependent) occupancy grids: names the reference's class-membership tests mention
    (``accel_cls in accel_types_dynamic``, app/models/single/dynamic_neus.py); never returned by ``get_accel_class`` here."""


class OccGridAccelDynamic(_NotBuilt):
    pass


class OccGridAccelBatchedDynamic_Ema(_NotBuilt):
    pass


# tuples of CLASSES: the reference tests ``get_accel_class(cfg.type) in accel_types_batched`` (batched_neus.py:111, 355)
accel_types_single = (OccGridAccel,)
accel_types_batched = (OccGridAccelBatched,)
accel_types_dynamic = (OccGridAccelDynamic,)
accel_types_batched_dynamic = (OccGridAccelBatchedDynamic_Ema,)
_BATCHED_NAMES = ("occ_grid_batched", "occ_grid_batched_ema")


def get_accel_class(type: str):
    """``accel_cfg.type`` -> class (app/models/single/neus.py: ``get_accel_class(accel_cfg.type)``)."""
    if type in ("occ_grid", "occ_grid_ema"):
        return OccGridAccel
    if type in _BATCHED_NAMES:
        return OccGridAccelBatched
    raise NotImplementedError(f"accel type {type!r}")


def get_accel(type: str = "occ_grid", **kw):
    return get_accel_class(type)(**kw)
