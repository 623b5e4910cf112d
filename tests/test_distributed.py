"""N>1 path on CPU: world_size-2 gloo processes exercise neuralsim_amd.distributed (ray sharding + the one gradient
all-reduce per step) and check that averaging per-shard gradients reproduces the full-batch gradient of the render
loss (computed with the oracle -- the HIP kernels are not involved, the collective logic is)."""
import os
import socket
import sys
from pathlib import Path

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = Path(__file__).resolve().parent.parent


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, str(ROOT))
    sys.path.insert(0, str(ROOT / "tests"))
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    torch.set_num_threads(2)
    from neuralsim_amd import distributed as nd
    from oracle import render as orr
    from util import look_at_cameras, make_params
    r, lr, w = nd.init_env(backend="gloo", device_type="cpu")
    assert (r, w) == (rank, world) and nd.get_world_size() == world and nd.is_master() == (rank == 0)

    # replicas start different on purpose; broadcast_module makes them identical (DDP construction semantics)
    lin = torch.nn.Linear(4, 3)
    with torch.no_grad():
        lin.weight.add_(rank)
    lin.register_buffer("occ", torch.full((5,), float(rank)))
    nd.broadcast_module(lin)
    assert float(lin.weight.sum()) == float(lin.weight.sum()) and torch.equal(lin.occ, torch.zeros(5))

    p = make_params(sdf_D=1, small=True, sphere=True, seed=3, ln_inv_s=0.45, grid_bound=2e-2, noise_scale=1.0)
    p.requires_grad_(True)
    g = torch.Generator().manual_seed(11)            # identical global batch on every rank
    intr, c2w, WH = look_at_cameras(V=3, seed=3)
    N = 24
    xy = torch.rand(N, 2, generator=g) * 0.5 + 0.25
    fidx = torch.randint(0, 3, (N,), generator=g)
    gt = torch.rand(N, 3, generator=g)
    o, d = orr.pinhole_rays(xy, fidx, intr, c2w, WH)
    aabb = torch.tensor([[-1.0, -1, -1], [1.0, 1, 1]])
    occ = torch.ones(16 ** 3, dtype=torch.bool)
    kw = dict(near=0.01, num_coarse=8, num_fine=(4,), upsample_inv_s_factors=(1,), step_size=0.1, max_steps=64)

    def loss_on(lo, hi):
        ret = orr.ray_query(p, o[lo:hi], d[lo:hi], None, occ, aabb[0], aabb[1], [16, 16, 16], **kw)
        rgb = torch.zeros(hi - lo, 3).index_put((ret["rays_inds"],), ret["rendered"]["rgb_volume"])
        return ((rgb - gt[lo:hi]) ** 2).mean()      # per-shard mean; shards are equal-sized

    lo, hi = nd.shard_range(N, rank, world)
    assert hi - lo == N // world
    loss_on(lo, hi).backward()
    params = [t for t in p.tensors() if t.requires_grad]
    params[-1].grad = None                           # a parameter without a local gradient must still take part
    local = [t.grad.clone() if t.grad is not None else None for t in params]
    results = {}
    for wire in (torch.float32, torch.bfloat16):     # exact wire and the 2-byte default for the big tensors
        for t, g0 in zip(params, local):
            t.grad = g0.clone() if g0 is not None else None
        nd.allreduce_grads(params, average=True, small_numel=1000, wire_dtype=wire)
        results[wire] = [t.grad.clone() for t in params]
    for t in params:
        t.grad = None
    loss_on(0, N).backward()
    for t, ga, gb in zip(params[:-1], results[torch.float32][:-1], results[torch.bfloat16][:-1]):
        ref = t.grad if t.grad is not None else torch.zeros_like(t)
        assert torch.allclose(ga, ref, atol=1e-6, rtol=1e-4), (rank, float((ga - ref).abs().max()))
        assert float((gb - ref).norm()) <= 8e-3 * float(ref.norm()) + 1e-9, (rank, "bf16 wire")
    # ``skip_absent`` (the lidar step: ADVICE r4): a parameter without a gradient on EVERY rank stays without one (the optimizer
    # then skips it, as torch.optim.Adam does); one that has a gradient on SOME rank is reduced on all, zeros from the others
    a_, b_, c_ = (torch.nn.Parameter(torch.zeros(5)) for _ in range(3))
    a_.grad = torch.full((5,), float(rank + 1))
    b_.grad = torch.ones(5) if rank == 0 else None
    c_.grad = None
    nd.allreduce_grads([a_, b_, c_], average=False, skip_absent=True)
    assert torch.equal(a_.grad, torch.full((5,), float(sum(range(1, world + 1))))) and torch.equal(b_.grad, torch.ones(5))
    assert c_.grad is None
    c_.grad = None
    nd.allreduce_grads([a_, c_], average=False)               # default: zero-filled, takes part
    assert c_.grad is not None and float(c_.grad.abs().max()) == 0.0
    dist.barrier()
    (Path(out_dir) / f"ok{rank}").write_text("ok")
    dist.destroy_process_group()


def test_two_rank_gloo_allreduce(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    assert all((tmp_path / f"ok{r}").exists() for r in range(world))


def test_shard_range_covers_batch():
    from neuralsim_amd.distributed import shard_range
    for n, w in ((131072, 8), (65536, 4), (10, 3), (5, 8)):
        spans = [shard_range(n, r, w) for r in range(w)]
        assert spans[0][0] == 0 and spans[-1][1] == n
        assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))


def _bench_worker(rank, world, port, out_dir, overlap="1", wire="f32", algo=None, steps=2, model_kind="lotd"):
    """bench.timed_run under two gloo ranks, with the kernel emulator standing in for the GPU (control flow of the
    N>1 path: sharded rays, gradient all-reduce inside train_step, barrier + max-over-ranks timing, rank-0 JSON)."""
    import ctypes
    sys.path.insert(0, str(ROOT))
    sys.path.insert(0, str(ROOT / "tests"))
    sys.path.insert(0, str(ROOT / "tests" / "emu"))
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port), NSIM_OVERLAP_ALLREDUCE=overlap, NSIM_ALLREDUCE_DTYPE=wire)   # f32 = exact wire:
    if algo is not None:
        os.environ["NSIM_ALLREDUCE_ALGO"] = algo
    # the two schedules must then agree to rounding (with the 2-byte wire, a + b of nearly cancelling rank gradients
    # may change sign, which Adam turns into a full step)
    torch.set_num_threads(2 if world <= 2 else 1)
    torch.manual_seed(0)
    import build_emu
    from neuralsim_amd import _lib, distributed as nd
    lib = _lib.bind(ctypes.CDLL(str(build_emu.build())))
    _lib.get_lib = lambda: lib
    _lib.stream_handle = lambda: 0
    _lib.require_device = lambda t, name="tensor": None
    import bench
    from test_trainer import _tiny
    from neuralsim_amd.graphics.cameras import look_at_cameras
    from neuralsim_amd.trainer import RenderTrainer
    nd.init_env(backend="gloo", device_type="cpu")
    dev = torch.device("cpu")
    if model_kind == "permuto":       # the permutohedral model through the same N > 1 chain (row f4)
        from neuralsim_amd.fields.permuto_neus import PermutoNeuSModel
        qp = dict(nablas_has_grad=True, num_coarse=8, num_fine=[4, 4], upsample_inv_s=64.0, upsample_inv_s_factors=[1, 4],
                  upsample_use_estimate_alpha=True, march_cfg=dict(step_size=0.05, max_steps=128))
        m = PermutoNeuSModel(permuto_auto_compute_cfg=dict(type="multi_res", n_levels=8, n_feats=2, log2_hashmap_size=10,
                                                           coarsest_res=2.0, finest_res=24.0), sdf_D=2, precision="fp16",
                             ln_inv_s_init=0.3, seed=42 + rank,
                             accel_cfg=dict(resolution=(16, 16, 16), update_from_net_cfg=dict(num_steps=1, num_pts=2048),
                                            update_from_samples_cfg={}, n_steps_between_update=4, n_steps_warmup=2),
                             ray_query_cfg=dict(query_mode="march_occ_multi_upsample", query_param=qp)).to(dev)
        m.geometric_init_sphere(0.5, num_iters=60, num_pts=1024, lr=5e-3)
        m.accel.init(m.query_sdf, num_steps=1, num_pts=2048)
    else:
        m = _tiny(dev, seed=42 + rank)                    # replicas differ until broadcast
    nd.broadcast_module(m)
    intr, c2w, WH = look_at_cameras(V=4, seed=1, device=dev)
    tr = RenderTrainer(m, intr, c2w, WH, num_rays=16, lr=1e-3, num_uniform=16, rank=rank, world_size=world)
    assert tr._fused_ok() and tr.overlap_allreduce == (overlap == "1")
    if overlap == "1" and model_kind == "lotd":      # two contiguous level ranges covering the pyramid and its parameters
        h = tr._grid_halves()
        cfg = m.encoding.cfg
        assert h[0][0] == 0 and h[0][1] == h[1][0] and h[1][1] == cfg.num_levels
        assert h[0][2] == 0 and h[0][3] == h[1][2] == cfg.lod_offsets[h[0][1]] and h[1][3] == cfg.n_params
    p0 = m.encoding.flattened_params.detach().clone()
    out, it_next = bench.timed_run(tr, steps=steps, warmup=1, rank=rank, world=world, dev=dev, rays_per_gpu=16)
    assert it_next == 257 + steps
    info = bench.distributed_info(world, dev)            # what the JSON line records about the N > 1 run (every rank calls it)
    assert info["ranks_seen"] == world and len(info["devices"]) == world and info["backend"] == "gloo"
    assert info["allreduce"] in ("ring", "direct") and info["wire_dtype"] in ("bfloat16", "float32", "float16")
    # replicas must still agree after the all-reduced updates
    w = m.sdf_w.detach().clone()
    ws = [torch.zeros_like(w) for _ in range(world)]
    dist.all_gather(ws, w)
    assert all(torch.equal(ws[0], x) for x in ws[1:])
    if rank == 0:
        torch.save(dict(grid=m.encoding.flattened_params.detach().clone(), sdf_w=w, rad_w=m.rad_w.detach().clone(),
                        appear=tr.appear.detach().clone(), grid0=p0),
                   str(Path(out_dir) / f"params_overlap{overlap}{wire}{algo or ''}{'' if model_kind == 'lotd' else model_kind}.pt"))
        assert out["n_gpus"] == world and out["steps"] == steps and out["value"] > 0 and out["scaling"] == "weak"
        # the driver's contract: every key of the one JSON line (``roofline`` is None without HIP events)
        for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                  "vs_baseline", "dtype", "data", "config", "roofline"):
            assert k in out, k
        assert out["unit"] == "rays/s" and out["higher_is_better"] is True and out["vs_baseline"] is None
        assert out["metric"].startswith("training rays/sec") and "workload" in out["config"] and out["data"] == "synthetic"
        assert abs(out["value"] - 16 * world * steps / (out["ms_per_step"] * steps * 1e-3)) / out["value"] < 1e-2
    else:
        assert out is None
    if overlap == "1" and wire == "f32":      # the N > 1 tail of bench.main(): same steps without the collectives
        rec = out if rank == 0 else {}
        it2 = bench.measure_exposed_allreduce(tr, rec, 2, it_next, rank, dev)
        assert it2 == it_next + 4 and tr.skip_allreduce is False
        if rank == 0:
            assert "exposed_allreduce_ms" in rec and rec["ms_per_step_without_allreduce"] > 0
    dist.barrier()
    (Path(out_dir) / f"bench_ok{rank}").write_text("ok")
    dist.destroy_process_group()


def test_bench_control_flow_two_ranks(tmp_path):
    """... with the table-gradient all-reduce overlapped in two halves (default) and as one collective after the
    backward: both leave the replicas in sync, and both arrive at the same parameters."""
    world = 2
    for overlap, wire in (("1", "f32"), ("0", "f32"), ("1", "bf16")):        # bf16 = the production wire format
        mp.spawn(_bench_worker, args=(world, _free_port(), str(tmp_path), overlap, wire), nprocs=world, join=True)
        assert all((tmp_path / f"bench_ok{r}").exists() for r in range(world))
        for r in range(world):
            (tmp_path / f"bench_ok{r}").unlink()
    a, b = (torch.load(str(tmp_path / f"params_overlap{o}f32.pt")) for o in ("1", "0"))
    for k in a:
        assert torch.allclose(a[k], b[k], rtol=1e-5, atol=1e-7), (k, float((a[k] - b[k]).abs().max()))


def test_two_ranks_train_the_permuto_model(tmp_path):
    """the permutohedral model in the N > 1 chain: overlapped halves (equal-sized lattice levels) and the single
    collective leave the replicas in sync and arrive at the same parameters"""
    world = 2
    for overlap in ("1", "0"):
        mp.spawn(_bench_worker, args=(world, _free_port(), str(tmp_path), overlap, "f32", None, 2, "permuto"), nprocs=world,
                 join=True)
        assert all((tmp_path / f"bench_ok{r}").exists() for r in range(world))
        for r in range(world):
            (tmp_path / f"bench_ok{r}").unlink()
    a, b = (torch.load(str(tmp_path / f"params_overlap{o}f32permuto.pt")) for o in ("1", "0"))
    for k in a:
        assert torch.allclose(a[k], b[k], rtol=1e-5, atol=1e-7), (k, float((a[k] - b[k]).abs().max()))


def test_eight_ranks_two_byte_wire_against_the_exact_reference(tmp_path):
    """world_size 8 (the node size the driver scales to; gloo + emulator, tiny model): the production schedule -- table
    gradient on a bf16 wire, ``direct`` all-reduce (one rounding per contribution, f32 accumulation), overlapped in two
    halves -- against the exact reference (f32 wire, one collective after the backward) on the PARAMETERS after several
    optimizer steps; the backend's own bf16 all-reduce (``ring``: accumulates in bf16) for comparison."""
    world, steps = 8, 3
    runs = (("0", "f32", None), ("1", "bf16", "direct"), ("1", "bf16", "ring"))
    for overlap, wire, algo in runs:
        mp.spawn(_bench_worker, args=(world, _free_port(), str(tmp_path), overlap, wire, algo, steps), nprocs=world, join=True)
        assert all((tmp_path / f"bench_ok{r}").exists() for r in range(world))
        for r in range(world):
            (tmp_path / f"bench_ok{r}").unlink()
    ref, direct, ring = (torch.load(str(tmp_path / f"params_overlap{o}{w}{a or ''}.pt")) for o, w, a in runs)
    lr = 1e-3
    err = {}
    for name, run in (("direct", direct), ("ring", ring)):
        for k in ("grid", "sdf_w", "rad_w", "appear"):
            assert float((run[k] - ref[k]).abs().max()) <= 2 * lr * steps + 1e-7, (name, k)     # Adam: |step| <= lr
        moved = ref["grid"] - ref["grid0"]
        err[name] = float(((run["grid"] - run["grid0"]) - moved).norm() / moved.norm())
    assert float(moved.abs().max()) > 0.5 * lr
    # a gradient rounded to 8 bits of mantissa moves an Adam update (g / sqrt(v): scale-free) only where contributions of
    # different ranks nearly cancel; the direct schedule rounds each contribution once
    assert err["direct"] < 0.1, err
    assert err["direct"] <= err["ring"] * 1.25 + 1e-3, err


def _gpu_dp_worker(rank, world, port, out_dir, overlap):
    """Two data-parallel ranks sharing ONE GPU (gloo moves the CUDA tensors; RCCL refuses two ranks per device): the real
    HIP kernels, real streams and the asynchronous collectives of the overlapped schedule."""
    sys.path.insert(0, str(ROOT))
    sys.path.insert(0, str(ROOT / "tests"))
    os.environ.update(RANK=str(rank), LOCAL_RANK="0", WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port), NSIM_OVERLAP_ALLREDUCE=overlap, NSIM_ALLREDUCE_DTYPE="f32")
    torch.manual_seed(0)
    torch.cuda.set_device(0)
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    from neuralsim_amd import distributed as nd
    from test_trainer import _tiny
    from neuralsim_amd.graphics.cameras import look_at_cameras
    from neuralsim_amd.trainer import RenderTrainer
    dev = torch.device("cuda", 0)
    m = _tiny(dev, seed=42)
    nd.broadcast_module(m)
    intr, c2w, WH = look_at_cameras(V=4, seed=1, device=dev)
    tr = RenderTrainer(m, intr, c2w, WH, num_rays=256, lr=1e-3, num_uniform=64, rank=rank, world_size=world,
                       target_sphere_radius=0.5)
    assert tr._fused_ok() and tr.overlap_allreduce == (overlap == "1")
    # 34 iterations = eight occupancy refreshes (every 4 from iteration 2): render-time values are collected per rank
    # (update_from_samples_cfg) and united by the MAX all-reduce before each refresh
    assert m.accel.update_from_samples_cfg is not None and m.accel.sync_values is not None
    losses = [float(tr.train_step(it)) for it in range(34)]
    assert all(l == l for l in losses)
    m.accel.sync_values(m.accel.occ_val)              # the values collected since the last refresh are per rank: unite them
    for name, t in (("grid", m.encoding.flattened_params), ("sdf_w", m.sdf_w), ("rad_w", m.rad_w), ("appear", tr.appear),
                    ("occ_bits", m.accel.occ_bits), ("occ_val", m.accel.occ_val)):
        g = t.detach().clone()
        gs = [torch.zeros_like(g) for _ in range(world)]
        dist.all_gather(gs, g)
        assert torch.equal(gs[0], gs[1]), name        # replicas bit-identical: parameters AND occupancy
    g = m.encoding.flattened_params.detach().clone()
    if rank == 0:
        torch.save(dict(grid=g.cpu(), sdf_w=m.sdf_w.detach().cpu(), losses=torch.tensor(losses)),
                   str(Path(out_dir) / f"gpu_overlap{overlap}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
def test_two_rank_overlapped_schedule_on_gpu(tmp_path):
    world = 2
    for overlap in ("1", "0"):
        mp.spawn(_gpu_dp_worker, args=(world, _free_port(), str(tmp_path), overlap), nprocs=world, join=True)
    a, b = (torch.load(str(tmp_path / f"gpu_overlap{o}.pt")) for o in ("1", "0"))
    # the two schedules see the same batches for the first iterations; float atomics commute only approximately, Adam
    # amplifies that, and after the first refresh the occupancy (hence the sample sets) may differ: compare the start
    assert torch.allclose(a["losses"][:4], b["losses"][:4], rtol=1e-4, atol=1e-6)
    assert float(a["losses"][-1]) < float(a["losses"][0]) and float(b["losses"][-1]) < float(b["losses"][0])


def _gpu_street_worker(rank, world, port, out_dir):
    """The street trainer's hook-driven exchange (``BackwardReducer``) with the real HIP kernels and streams: two ranks sharing ONE
    GPU over gloo, the small tables counted as 'big' so that they leave through the hooks."""
    sys.path.insert(0, str(ROOT))
    sys.path.insert(0, str(ROOT / "tests"))
    os.environ.update(RANK=str(rank), LOCAL_RANK="0", WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port), NSIM_OVERLAP_ALLREDUCE="1", NSIM_ALLREDUCE_DTYPE="f32")
    torch.manual_seed(0)
    torch.cuda.set_device(0)
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    from neuralsim_amd import distributed as nd, scenarios as sc
    dev = torch.device("cuda", 0)
    tr = sc.build_street_trainer(dev, rank=rank, world_size=world, small=True, rays_per_gpu=64, lidar_rays=64, num_uniform=32, seed=42)
    assert not tr._fused_ok()
    tr._reducer = red = nd.BackwardReducer(tr.optim.params(), small_numel=1 << 12)
    logs = []
    for it in range(4):
        loss = tr.train_step(it)
        assert float(loss) == float(loss)
        logs.append(list(red.log))
    assert all(w == "finish" for _, w in logs[0]) and sum(w == "backward" for _, w in logs[-1]) >= 2, logs
    for name, t in (("grid", tr.model.encoding.flattened_params), ("sdf_w", tr.model.sdf_w), ("rad_w", tr.model.rad_w),
                    ("distant", tr.distant_model.flattened_params), ("sky", tr.sky_model.w), ("appear", tr.appear)):
        g = t.detach().clone()
        gs = [torch.zeros_like(g) for _ in range(world)]
        dist.all_gather(gs, g)
        assert torch.equal(gs[0], gs[1]), name
    dist.barrier()
    (Path(out_dir) / f"gpu_street_ok{rank}").write_text("ok")
    dist.destroy_process_group()


@pytest.mark.gpu
def test_street_exchange_during_the_backward_on_gpu(tmp_path):
    mp.spawn(_gpu_street_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    assert all((tmp_path / f"gpu_street_ok{r}").exists() for r in range(2))


def _street_worker(rank, world, port, out_dir, steps, overlap="1", tag=None):
    """The street trainer (configs[3] shape, small: NeuS street + distant + sky, pixel step AND lidar step per iteration) under
    ``world`` gloo ranks on the kernel emulator: the autograd-path schedule -- ``allreduce_grads`` after each backward, the
    lidar step with ``skip_absent`` (parameters outside its graph are neither reduced nor stepped on any rank)."""
    sys.path.insert(0, str(ROOT))
    sys.path.insert(0, str(ROOT / "tests"))
    sys.path.insert(0, str(ROOT / "tests" / "emu"))
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    os.environ["NSIM_OVERLAP_ALLREDUCE"] = overlap
    if tag is not None:
        os.environ["NSIM_ALLREDUCE_DTYPE"] = "f32"       # exact sums: the two schedules must agree to the bit
    torch.set_num_threads(1)
    torch.manual_seed(0)
    import ctypes
    import build_emu
    from neuralsim_amd import _lib, distributed as nd, scenarios as sc
    lib = _lib.bind(ctypes.CDLL(str(build_emu.build())))
    _lib.get_lib = lambda: lib
    _lib.stream_handle = lambda: 0
    _lib.require_device = lambda t, name="tensor": None
    nd.init_env(backend="gloo", device_type="cpu")
    dev = torch.device("cpu")
    tr = sc.build_street_trainer(dev, rank=rank, world_size=world, small=True, rays_per_gpu=32, lidar_rays=32, num_uniform=16, seed=42)
    assert not tr._fused_ok() and tr.distant_model is not None and tr.sky_model is not None
    released = []
    if overlap == "1":
        # the small tables of this test count as "big" parameters: they leave through the hook-driven exchange
        tr._reducer = red = nd.BackwardReducer(tr.optim.params(), small_numel=1 << 12)
        names = {id(tr.model.encoding.flattened_params): "street_table", id(tr.distant_model.flattened_params): "distant_table"}
        _lib.CALL_COUNT = 0
        red.on_release = lambda k: released.append((names.get(id(red.big[k]), "other"), _lib.CALL_COUNT))
        fin0 = red.finish_small

        def finish_small():
            released.append(("backward_done", _lib.CALL_COUNT))
            fin0()
        red.finish_small = finish_small
    per_step = []
    for it in range(steps):
        del released[:]
        loss = tr.train_step(it)
        assert float(loss) == float(loss)
        per_step.append(list(released))
    if overlap == "1":
        assert {"street_table", "distant_table"} <= {n for n, _ in per_step[0]}, per_step[0]
        # first armed step: the completion order is observed, nothing leaves before the backward is over
        i0 = [n for n, _ in per_step[0]].index("backward_done")
        assert i0 == 0, per_step[0]
        # every later step: the table of the model whose backward finishes first leaves while kernels of the other models'
        # backward are still being launched (ABI calls between its release and the end of the backward)
        for ev in per_step[1:]:
            seq = [n for n, _ in ev]
            done = dict(ev)["backward_done"]
            early = [(n, c) for n, c in ev[:seq.index("backward_done")]]
            assert early and early[0][1] < done, ev
            # the distant model's table is on its way before the street model's three backward launches + scatter are issued
            assert dict(ev)["distant_table"] + 3 <= dict(ev)["street_table"] <= done, ev
        # ... and every rank released in the same order
        order = [n for n, _ in per_step[-1] if n != "backward_done"]
        orders = [None] * world
        dist.all_gather_object(orders, order)
        assert all(o == orders[0] for o in orders), orders
    grp = {id(g["p"]): g for g in tr.optim.groups}
    assert grp[id(tr.model.sdf_w)]["t"] == 2 * steps and grp[id(tr.model.rad_w)]["t"] == steps       # lidar: no radiance update
    for name, t in (("grid", tr.model.encoding.flattened_params), ("sdf_w", tr.model.sdf_w), ("rad_w", tr.model.rad_w),
                    ("distant", tr.distant_model.flattened_params), ("sky", tr.sky_model.w), ("appear", tr.appear)):
        g = t.detach().clone()
        gs = [torch.zeros_like(g) for _ in range(world)]
        dist.all_gather(gs, g)
        assert all(torch.equal(gs[0], x) for x in gs[1:]), name
    dist.barrier()
    if tag is not None and rank == 0:
        torch.save({n: t.detach().clone() for n, t in (("grid", tr.model.encoding.flattened_params), ("sdf_w", tr.model.sdf_w),
                                                         ("distant", tr.distant_model.flattened_params), ("sky", tr.sky_model.w),
                                                         ("appear", tr.appear))}, str(Path(out_dir) / f"street_{tag}.pt"))
        (Path(out_dir) / f"street_{tag}_events.txt").write_text(repr(per_step))
    (Path(out_dir) / f"street_ok{rank}").write_text("ok")
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 8])
def test_street_trainer_replicas_stay_in_sync(tmp_path, world):
    """configs[3]'s trainer (pixel + lidar step) at world sizes 2 and 8 (the node size): every rank issues the same
    collective sequence whatever its batches hit, and the replicas are bit-identical after the all-reduced updates."""
    mp.spawn(_street_worker, args=(world, _free_port(), str(tmp_path), 2), nprocs=world, join=True)
    assert all((tmp_path / f"street_ok{r}").exists() for r in range(world))


def test_street_exchange_overlaps_the_backward(tmp_path):
    """The autograd-path exchange of configs[3] (``ndist.BackwardReducer``): from the second step on a table's all-reduce is
    issued during the backward, before the kernels of the model that finishes last; all ranks release in one order; and the
    overlapped schedule gives the bit-identical parameters of the plain one (``allreduce_grads`` after the backward) with an
    exact wire."""
    world = 2
    for overlap, tag in (("1", "overlap"), ("0", "plain")):
        mp.spawn(_street_worker, args=(world, _free_port(), str(tmp_path), 3, overlap, tag), nprocs=world, join=True)
    a, b = torch.load(tmp_path / "street_overlap.pt"), torch.load(tmp_path / "street_plain.pt")
    for k in a:
        assert torch.equal(a[k], b[k]), k
    ev = (tmp_path / "street_overlap_events.txt").read_text()
    assert "distant_table" in ev and "street_table" in ev


def _multi_worker(rank, world, port, out_dir, overlap, tag):
    """The multi-object trainer (configs[4] shape, small: street + two posed vehicle instances of one shared model + distant +
    sky through ``BufferComposeRenderer``) under ``world`` gloo ranks on the kernel emulator, exact wire."""
    sys.path.insert(0, str(ROOT))
    sys.path.insert(0, str(ROOT / "tests"))
    sys.path.insert(0, str(ROOT / "tests" / "emu"))
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                      NSIM_OVERLAP_ALLREDUCE=overlap, NSIM_ALLREDUCE_DTYPE="f32")
    torch.set_num_threads(1)
    torch.manual_seed(0)
    import ctypes
    import build_emu
    from neuralsim_amd import _lib, distributed as nd, scenarios as sc
    lib = _lib.bind(ctypes.CDLL(str(build_emu.build())))
    _lib.get_lib = lambda: lib
    _lib.stream_handle = lambda: 0
    _lib.require_device = lambda t, name="tensor": None
    nd.init_env(backend="gloo", device_type="cpu")
    tr = sc.build_multi_trainer(torch.device("cpu"), rank=rank, world_size=world, small=True, rays_per_gpu=32, seed=42, B=2)
    if overlap == "1":
        tr._reducer = nd.BackwardReducer(tr.optim.params(), small_numel=1 << 12)
    logs = []
    for it in range(3):
        loss = tr.train_step(it)
        assert float(loss) == float(loss)
        if overlap == "1":
            logs.append(list(tr._reducer.log))
    state = {f"p{i}": p.detach().clone() for i, p in enumerate(tr.optim.params())}
    for n, t in state.items():
        gs = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(gs, t)
        assert all(torch.equal(gs[0], x) for x in gs[1:]), n
    if overlap == "1":
        # street table, vehicle table (the shared model's grower), distant table, sky, decoder blocks >= 4096 entries
        assert len(tr._reducer.big) >= 4
        assert all(w == "finish" for _, w in logs[0]) and sum(w == "backward" for _, w in logs[-1]) >= 3, logs
    if rank == 0:
        torch.save(state, str(Path(out_dir) / f"multi_{tag}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_multi_object_trainer_two_ranks_overlapped_exchange(tmp_path):
    """configs[4]'s trainer at world size 2: replicas bit-identical after three steps, the hook-driven exchange releases the
    models' tables during the backward, and its result equals the plain schedule's to the bit (exact wire)."""
    world = 2
    for overlap, tag in (("1", "overlap"), ("0", "plain")):
        mp.spawn(_multi_worker, args=(world, _free_port(), str(tmp_path), overlap, tag), nprocs=world, join=True)
    a, b = torch.load(tmp_path / "multi_overlap.pt"), torch.load(tmp_path / "multi_plain.pt")
    assert a.keys() == b.keys()
    for k in a:
        assert torch.equal(a[k], b[k]), k


def _reducer_worker(rank, world, port, out_dir):
    """``BackwardReducer`` alone: three 'tables' whose gradients arrive in DIFFERENT orders on the two ranks, one of them not at
    all on rank 1 -- every rank must issue the same collective sequence (no deadlock, no mismatched sizes) and every sum be right."""
    sys.path.insert(0, str(ROOT))
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                      NSIM_ALLREDUCE_DTYPE="f32")
    torch.set_num_threads(1)
    from neuralsim_amd import distributed as nd
    nd.init_env(backend="gloo", device_type="cpu")
    g = torch.Generator().manual_seed(5)
    A, B, Cc = (torch.nn.Parameter(torch.randn(n, generator=g)) for n in (5000, 7001, 4099))      # different sizes: a mismatch would fail
    small = torch.nn.Parameter(torch.randn(17, generator=g))
    # ... and two parameters NO rank has a gradient for (a sky / distant model outside every rank's batch): they must come out
    # with ``.grad is None`` -- un-stepped, as at world size 1 -- while B (absent on rank 1 only) gets rank 0's sum
    D_big, d_small = torch.nn.Parameter(torch.randn(4500, generator=g)), torch.nn.Parameter(torch.randn(9, generator=g))
    red = nd.BackwardReducer([A, B, Cc, small, D_big, d_small], small_numel=4096)
    assert len(red.big) == 4

    def loss_fn(step):
        w = float(rank + 1 + step)
        if rank == 0:       # completion order of the engine: the LAST used parameter's gradient is ready first
            return (A * w).sum() * 1.0 + (B * 2 * w).sum() + (Cc * 3 * w).sum() + (small * w).sum()
        # rank 1: another expression order, and B not in the graph at all (its rays missed that model)
        return (Cc * 3 * w).sum() + (small * w).sum() + (A * w).sum()
    for step in range(3):
        for p in (A, B, Cc, small, D_big, d_small):
            p.grad = None
        red.begin()
        loss_fn(step).backward()
        red.finish_small()
        stepped = [p for p in red.finish_big()]
        assert D_big.grad is None and d_small.grad is None and not any(p is D_big for p in stepped)
        assert red.absent == {id(D_big), id(d_small)}, (rank, step, [(i, id(p) in red.absent) for i, p in enumerate(red.params)])
        assert sum(any(p is q for p in stepped) for q in (A, B, Cc)) == 3
        w0, w1 = 1.0 + step, 2.0 + step
        assert torch.allclose(A.grad, torch.full_like(A, w0 + w1))
        assert torch.allclose(B.grad, torch.full_like(B, 2 * w0))                 # rank 1 contributed zeros
        assert torch.allclose(Cc.grad, torch.full_like(Cc, 3 * (w0 + w1)))
        assert torch.allclose(small.grad, torch.full_like(small, w0 + w1))
        order = [id(p) for p in red.big]
        orders = [None] * world
        dist.all_gather_object(orders, [[id(A), id(B), id(Cc), id(D_big)].index(i) for i in order])
        assert orders[0] == orders[1], orders                                     # one release order on every rank
        if step >= 1:
            assert [k for k, _ in red.log] == [0, 1, 2, 3]                          # released in that order ...
            if rank == 0:
                assert sum(w == "backward" for _, w in red.log) >= 1                # ... and during the backward where possible
    dist.barrier()
    (Path(out_dir) / f"reducer_ok{rank}").write_text("ok")
    dist.destroy_process_group()


def test_backward_reducer_keeps_one_collective_order(tmp_path):
    mp.spawn(_reducer_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    assert all((tmp_path / f"reducer_ok{r}").exists() for r in range(2))


def test_bench_cli_gpus_2_becomes_two_ranks():
    """``python bench.py --gpus 2`` WITHOUT a launcher environment must become two ranks by itself (it
    re-executes under ``python -m torch.distributed.run --nproc-per-node 2``) and the line must say n_gpus 2 == ranks seen 2.
    Driven through the command line, not through ``timed_run``; ``--emulator`` swaps RCCL + libnsim_hip.so for gloo + the
    host emulator of tests/emu (this machine has no GPU) and a tiny model -- every other line of bench.main's N > 1 flow
    (launcher, rank-count checks, barrier + max-over-ranks timing, exposed all-reduce leg, rank-0 JSON) is the product's."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--emulator"],
                       capture_output=True, text=True, timeout=900, env=env, cwd=str(ROOT))
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout                                  # ONE JSON line, from rank 0
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["distributed"]["ranks_seen"] == 2 and out["distributed"]["backend"] == "gloo"
    assert len(out["distributed"]["devices"]) == 2 and out["steps"] == 2 and out["warmup"] == 1 and out["emulator"] is True
    assert "exposed_allreduce_ms" in out and out["scaling"] == "weak" and out["value"] > 0
    # a launcher environment that disagrees with --gpus is refused instead of printing a line for another rank count
    r2 = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "1", "--emulator"], capture_output=True, text=True,
                        timeout=300, env=dict(env, WORLD_SIZE="2", RANK="0"), cwd=str(ROOT))
    assert r2.returncode != 0 and "WORLD_SIZE=2" in (r2.stdout + r2.stderr)
    # without the emulator flag (the product) a node with fewer than N devices is refused too: no silent 1-rank run
    if not torch.cuda.is_available():
        r3 = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2"], capture_output=True, text=True,
                            timeout=300, env=env, cwd=str(ROOT))
        assert r3.returncode != 0 and "HIP device" in r3.stderr
