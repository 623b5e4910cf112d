"""``nr3d_lib.models.annealers.get_anneal_val`` as the reference's losses call it (``get_anneal_val(it=it, **anneal_cfg)``
with ``anneal_cfg{type: linear, start_it, stop_it, start_val, stop_val, update_every}``,
lotd_neus.dtu.230814.yaml:337-343; app/loss/clearance.py:76, app/loss/weight_reg.py:60).  Implementation absent
(nr3d_lib): the schedules are restated from the config keys -- value moves from ``start_val`` to ``stop_val`` between
``start_it`` and ``stop_it`` (held outside), re-evaluated every ``update_every`` iterations."""
import math


def get_anneal_val(type: str = "linear", it: int = 0, start_it: int = 0, stop_it: int = 1, start_val: float = 0.0,
                   stop_val: float = 1.0, update_every: int = 1, **unused) -> float:
    it = int(it)
    if update_every and update_every > 1:
        it = (it // int(update_every)) * int(update_every)
    p = min(max((it - start_it) / max(stop_it - start_it, 1), 0.0), 1.0)
    if type == "linear":
        f = p
    elif type in ("cosine", "cos"):
        f = 0.5 * (1.0 - math.cos(math.pi * p))
    elif type in ("log", "logspace", "exponential"):
        if start_val > 0 and stop_val > 0:
            return float(math.exp(math.log(start_val) + p * (math.log(stop_val) - math.log(start_val))))
        f = p
    elif type in ("milestones", "hardmask", "step"):
        f = 1.0 if p >= 1.0 else 0.0
    else:
        raise NotImplementedError(f"anneal type {type!r}")
    return float(start_val + f * (stop_val - start_val))


def get_annealer(**cfg):
    return lambda it: get_anneal_val(it=it, **cfg)


def get_anneal_val_milestones(it: int, milestones, vals):
    """``vals[k]`` for the k-th interval of ``milestones`` (len(vals) == len(milestones) + 1): milestones mark the ends of
    the intervals (dataio/data_loader/patch_sampler.py:214-228)."""
    k = sum(1 for m in milestones if int(it) >= int(m))
    return vals[min(k, len(vals) - 1)]
