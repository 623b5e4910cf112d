"""``nr3d_lib.models.accelerations`` (reference imports: code_single/tools/render.py:213-220)."""
from neuralsim_amd.fields.neus import OccGridAccel  # noqa: F401
OccGridEma = OccGridAccel
from neuralsim_amd.fields.batched_neus import OccGridAccelBatched  # noqa: F401,E402  (``occ_grid_batched``)


accel_types_single = ("occ_grid", "occ_grid_ema", None)
accel_types_batched = ("occ_grid_batched", "occ_grid_batched_ema")
accel_types_dynamic = ("occ_grid_dynamic",)
accel_types_batched_dynamic = ("occ_grid_batched_dynamic",)
OccGridEmaBatched = OccGridAccelBatched_Ema = OccGridAccelBatched


def get_accel_class(type: str):
    """``accel_cfg.type`` -> class (app/models/single/neus.py: ``get_accel_class(accel_cfg.type)``)."""
    if type in ("occ_grid", "occ_grid_ema"):
        return OccGridAccel
    if type in accel_types_batched:
        return OccGridAccelBatched
    raise NotImplementedError(f"accel type {type!r}")


def get_accel(type: str = "occ_grid", **kw):
    return get_accel_class(type)(**kw)
