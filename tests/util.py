"""Shared helpers for the parity tests (oracle <-> product weight exchange, synthetic cameras)."""
import math
import os

import torch

from oracle import field as ofield


def leaf(t, dev=None, dtype=None):
    t = t.detach().clone()
    if dtype is not None:
        t = t.to(dtype)
    if dev is not None:
        t = t.to(dev)
    return t.requires_grad_(True)


SMALL_RES = [4, 6, 8, 11, 14, 18, 23, 29, 36, 44, 53, 63, 74, 86, 99, 113]


def make_params(sdf_D=2, small=True, sphere=True, grid_bound=None, seed=42, ln_inv_s=0.3, noise_scale=0.25):
    """Oracle FieldParams; ``small`` uses a 16-level pyramid with a 2^12 hash table so that emulator runs stay fast
    while still exercising 8 dense + 8 hashed levels."""
    if small:
        p = ofield.make_field_params(lod_res=SMALL_RES, log2_hashmap_size=12, sdf_D=sdf_D, seed=seed,
                                     sphere_init=sphere, grid_bound=grid_bound if grid_bound is not None else 1e-4,
                                     ln_inv_s=ln_inv_s, noise_scale=noise_scale)
    else:
        p = ofield.make_field_params(sdf_D=sdf_D, seed=seed, sphere_init=sphere,
                                     grid_bound=grid_bound if grid_bound is not None else 1e-4, ln_inv_s=ln_inv_s,
                                     noise_scale=noise_scale)
    p.grid = p.grid.float()   # fp16-representable values held in f32 (exact f32 grads in the oracle)
    return p


def model_from_params(p, device, precision="f32", log2_hashmap_size=None):
    """Product model carrying exactly the oracle's weights."""
    from neuralsim_amd.fields.neus import LoTDNeuSModel
    l2 = int(math.log2(p.spec.hashmap_size))
    m = LoTDNeuSModel(lod_res=p.spec.lod_res, log2_hashmap_size=l2, sdf_D=len(p.sdf_w) - 1, precision=precision,
                      softplus_beta=-1.0 if getattr(p, "sdf_activation", "softplus") == "relu" else 100.0,
                      ln_inv_s_init=float(p.ln_inv_s), ln_inv_s_factor=p.ln_inv_s_factor,
                      pos_embed_frequencies=getattr(p, "pos_embed_n", None), sdf_scale=float(getattr(p, "sdf_scale", 1.0)),
                      aabb=torch.as_tensor(p.spec.aabb, dtype=torch.float32) if getattr(p.spec, "aabb", None) is not None else None)
    with torch.no_grad():
        m.encoding.flattened_params.copy_(p.grid.detach().float())
        m.sdf_w.copy_(torch.cat([w.detach().reshape(-1) for w in p.sdf_w]))
        m.sdf_b.copy_(torch.cat([b.detach().reshape(-1) for b in p.sdf_b]))
        m.rad_w.copy_(torch.cat([w.detach().reshape(-1) for w in p.rad_w]))
        m.rad_b.copy_(torch.cat([b.detach().reshape(-1) for b in p.rad_b]))
    return m.to(device)


def oracle_flat_grads(p):
    """Gradients of the oracle params flattened into the product's flat layouts."""
    def g(t):
        return t.grad if t.grad is not None else torch.zeros_like(t)
    return dict(grid=g(p.grid), sdf_w=torch.cat([g(w).reshape(-1) for w in p.sdf_w]),
                sdf_b=torch.cat([g(b).reshape(-1) for b in p.sdf_b]),
                rad_w=torch.cat([g(w).reshape(-1) for w in p.rad_w]),
                rad_b=torch.cat([g(b).reshape(-1) for b in p.rad_b]), ln_inv_s=g(p.ln_inv_s).reshape(-1))


def look_at_cameras(V=4, radius=3.0, H=800, W=800, f=1111.1, seed=0):
    """V pinhole cameras (OpenCV convention: +z forward, +y down) on a sphere looking at the origin
    (SURVEY.md sec. 8d synthetic inputs)."""
    g = torch.Generator().manual_seed(seed)
    c2w = torch.eye(4).repeat(V, 1, 1)
    for i in range(V):
        u = torch.rand(2, generator=g)
        th, ph = float(2 * math.pi * u[0]), float(math.acos(1 - 2 * (0.15 + 0.7 * u[1])))
        eye = radius * torch.tensor([math.sin(ph) * math.cos(th), math.cos(ph), math.sin(ph) * math.sin(th)])
        fwd = -eye / eye.norm()
        up = torch.tensor([0.0, -1.0, 0.0])
        right = torch.linalg.cross(fwd, up)
        right = right / right.norm()
        down = torch.linalg.cross(fwd, right)
        c2w[i, :3, 0], c2w[i, :3, 1], c2w[i, :3, 2], c2w[i, :3, 3] = right, down, fwd, eye
    intr = torch.tensor([[f, 0, W / 2], [0, f, H / 2], [0, 0, 1.0]]).repeat(V, 1, 1)
    WH = torch.tensor([[W, H]], dtype=torch.long).repeat(V, 1)
    return intr, c2w, WH


def rel_l2(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30))


# ----------------------------------------------------------------- product -> oracle weight exchange (whole models)
def oracle_of_neus(m, table="master"):
    """oracle.field.FieldParams carrying the weights of a product LoTDNeuSModel (any pyramid / AABB / sdf_scale).
    ``table``: "master" = the f32 optimizer copy rounded to fp16 (what the kernels' shadow holds), as leaf tensors."""
    cfg = m.encoding.cfg
    p = ofield.params_from_flat(cfg.lod_res, int(math.log2(cfg.hashmap_size)), m.encoding.flattened_params, m.sdf_w,
                                m.sdf_b, m.rad_w, m.rad_b, m.ln_inv_s, sdf_D=m.sdf_D, ln_inv_s_factor=m.ln_inv_s_factor,
                                sdf_scale=m.sdf_scale, aabb=m.accel.aabb)
    n_act = int(m.field_meta.lotd.n_active_levels)
    if n_act:
        p.spec.n_active = n_act
    return p


def oracle_of_distant(dm):
    """oracle.distant.DistantParams carrying the weights of a product LoTDNeRFDistantModel."""
    from oracle import distant as od
    c = dm._ctor
    auto = dict(c.get("lotd_auto_compute_cfg") or {})
    ext = (dm.aabb[1] - dm.aabb[0]).detach().cpu().tolist()
    aspect = ext if (dm.lotd_use_cuboid and max(ext) / min(ext) > 1.0 + 1e-6) else None
    spec = od.make_ngp4d_spec(auto.get("target_num_params", 8 * 2 ** 20), auto.get("min_res_xyz", 8), auto.get("min_res_w", 4),
                              2, auto.get("log2_hashmap_size", 19), auto.get("per_level_scale", 1.382), aspect=aspect)
    assert spec.n_params == dm.cfg.n_params and spec.types == dm.cfg.types
    Fd = 2 * spec.num_levels
    K1 = Fd + (20 if dm.use_view_dirs else 4)
    dw, db = dm.den_w.detach().cpu().float(), dm.den_b.detach().cpu().float()
    rw, rb = dm.rad_w.detach().cpu().float(), dm.rad_b.detach().cpu().float()
    return od.DistantParams(spec, dm.flattened_params.detach().cpu().half().float(),
                            [dw[:64 * Fd].view(64, Fd).clone(), dw[64 * Fd:].view(1, 64).clone()],
                            [db[:64].clone(), db[64:].clone()],
                            [rw[:64 * K1].view(64, K1).clone(), rw[64 * K1:64 * K1 + 4096].view(64, 64).clone(),
                             rw[64 * K1 + 4096:].view(3, 64).clone()],
                            [rb[:64].clone(), rb[64:128].clone(), rb[128:].clone()])


def distant_flat_grads(pd):
    def g(t):
        return t.grad if t.grad is not None else torch.zeros_like(t)
    return dict(dv_grid=g(pd.grid), dv_den_w=torch.cat([g(w).reshape(-1) for w in pd.den_w]),
                dv_den_b=torch.cat([g(b).reshape(-1) for b in pd.den_b]),
                dv_rad_w=torch.cat([g(w).reshape(-1) for w in pd.rad_w]),
                dv_rad_b=torch.cat([g(b).reshape(-1) for b in pd.rad_b]))


def oracle_of_sky(sm):
    """(ws, bs) of oracle.sky carrying the weights of a product SimpleSky."""
    IN, W = sm.in_dim, 256
    w, b = sm.w.detach().cpu().float(), sm.b.detach().cpu().float()
    ws = [w[:W * IN].view(W, IN).clone(), w[W * IN:W * IN + W * W].view(W, W).clone(), w[W * IN + W * W:].view(3, W).clone()]
    bs = [b[:W].clone(), b[W:2 * W].clone(), b[2 * W:].clone()]
    return ws, bs


def steps_on_a_fixed_objective(tr, its):
    """``tr.train_step(it)`` for every ``it`` of ``its`` with the trainer's random streams rewound before each step, on a
    trainer whose ``sample_batch`` was pinned to one batch: every step then sees the SAME pixels, the SAME marching / depth
    jitter and the SAME uniform eikonal points, so the returned losses are values of one fixed function of the parameters
    along the optimiser's trajectory.  "The loss went down" on that sequence is a statement about the step (gradient sign,
    Adam, shadow refresh), not about which jitter a step happened to draw -- a trend over freshly jittered 24-ray batches is
    noise (0.0154, 0.0140, 0.0163, 0.0139 ... on the driver's box)."""
    state = (tr.gen.get_state(), tr.gen_shared.get_state(), torch.get_rng_state(),
             torch.cuda.get_rng_state() if torch.cuda.is_available() else None)
    losses = []
    for it in its:
        tr._prefetched = None                       # (a batch drawn ahead of time carries the previous draw's jitter)
        tr.gen.set_state(state[0])
        tr.gen_shared.set_state(state[1])
        torch.set_rng_state(state[2])
        if state[3] is not None:
            torch.cuda.set_rng_state(state[3])
        losses.append(float(tr.train_step(it)))
    tr._prefetched = None
    if os.environ.get('NSIM_PRINT_LOSSES'):
        print('losses', ['%.5f' % l for l in losses])
    return losses
