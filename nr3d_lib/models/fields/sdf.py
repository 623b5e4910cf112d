"""``nr3d_lib.models.fields.sdf.{pretrain_sdf_capsule, pretrain_sdf_road_surface}`` as the street model's
``asset_training_initialize`` calls them (app/models/single/neus.py:26, 222-227;
``initialize_cfg{target_shape: road_surface, floor_dim: z, floor_up_sign: 1, ego_height: 2.0, lr, num_iters, ...}``,
withmask_withlidar_joint.240219.yaml:232-241).

The reference fits the SDF network to the target shape with ``num_iters`` optimisation steps; the implementation lives
in the absent nr3d_lib.  By default the target is written deterministically into the finest dense level of the table
(``LoTDNeuSModel.geometric_init_fn``): same starting geometry, no optimisation loop.  ``geo_init_impl: pretrain`` in the
``initialize_cfg`` block (or NSIM_GEO_INIT=pretrain) runs the reference's procedure instead: ``num_iters`` Adam steps of
``lr`` on ``num_points`` random points per step through the model's own kernels (``LoTDNeuSModel.pretrain_sdf_fn``)."""
import os

import torch


def _init(implicit_surface, fn_metres, cfg, logger):
    # ``sdf_scale``: "the real-world length represented by one unit of SDF" (withmask_withlidar_joint.240219.yaml:24; the
    # occupancy shell ``inv_s 256 => +- 0.01 sdf`` = +- 0.25 m and ``clearance_sdf 0.02 = 0.5 m`` count in those units): the
    # network's SDF is distance / sdf_scale, so that is what the targets (distances in object units) are fitted as
    scale = float(getattr(implicit_surface, "sdf_scale", 1.0))
    fn = (lambda x: fn_metres(x) / scale) if scale != 1.0 else fn_metres
    if cfg.get("geo_init_impl", os.environ.get("NSIM_GEO_INIT", "write")) == "pretrain":
        implicit_surface.pretrain_sdf_fn(lambda x: fn(x.detach().cpu()), num_iters=int(cfg.get("num_iters", 500)),
                                         lr=float(cfg.get("lr", 2e-3)), num_pts=int(cfg.get("num_points", cfg.get("num_pts", 2 ** 14))),
                                         logger=logger, w_eikonal=float(cfg.get("w_eikonal", 0.0)))
    else:
        implicit_surface.geometric_init_fn(fn, level=implicit_surface._geo_init_level())

_DIM = dict(x=0, y=1, z=2)


def pretrain_sdf_road_surface(implicit_surface, tracks_in_obj: torch.Tensor, floor_dim: str = "z", floor_up_sign: int = 1,
                              ego_height: float = 0.0, logger=None, log_prefix: str = None, **unused):
    """Signed height above a road surface that follows the ego trajectory: at every vertex the track point that is
    nearest in the two ground-plane axes gives the local floor level ``track - up * ego_height``."""
    d = _DIM[floor_dim]
    others = [i for i in range(3) if i != d]
    tr = tracks_in_obj.detach().reshape(-1, 3).float().cpu()

    def fn(x):
        out = torch.empty(x.shape[0])
        for lo in range(0, x.shape[0], 65536):          # chunked nearest-track search
            xc = x[lo:lo + 65536]
            k = torch.cdist(xc[:, others], tr[:, others]).argmin(dim=1)
            floor = tr[k, d] - float(floor_up_sign) * float(ego_height)
            out[lo:lo + 65536] = float(floor_up_sign) * (xc[:, d] - floor)
        return out
    _init(implicit_surface, fn, unused, logger)


def pretrain_sdf_capsule(implicit_surface, tracks_in_obj: torch.Tensor, surface_distance: float = 0.2, logger=None,
                         log_prefix: str = None, **unused):
    """Free space inside a tube of radius ``surface_distance`` around the ego trajectory, solid outside:
    sdf = surface_distance - distance to the nearest track point (positive = free, as seen from the cameras on the
    track)."""
    tr = tracks_in_obj.detach().reshape(-1, 3).float().cpu()

    def fn(x):
        out = torch.empty(x.shape[0])
        for lo in range(0, x.shape[0], 65536):
            out[lo:lo + 65536] = float(surface_distance) - torch.cdist(x[lo:lo + 65536], tr).min(dim=1).values
        return out
    _init(implicit_surface, fn, unused, logger)
