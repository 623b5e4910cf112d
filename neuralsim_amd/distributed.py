"""Process-per-GPU data parallelism over RCCL/xGMI for the render step.

Mirrors what the reference gets from ``nr3d_lib.distributed.init_env`` + ``DistributedDataParallel``
(code_single/tools/train.py:1195,1401-1406): every rank holds a full replica (24-100 MB: trivial next to 288 GB
of HBM), draws / receives its own shard of the ray batch, and the only data-path collective per step is ONE
sum-all-reduce of the gradients.  The fused launch chain of the headline step needs no bucketing-by-autograd-hook: its
backward is a fixed kernel sequence and the table gradient leaves in two halves around the scatter
(``RenderTrainer._dp_reduce_step``).  The autograd-path steps (street / indoor / multi-object: several models per
backward) use ``BackwardReducer`` below: one table's exchange under the next model's backward kernels.
"""
import os
from typing import List, Optional, Sequence

import torch
import torch.distributed as dist


def init_env(backend: Optional[str] = None, device_type: str = "cuda"):
    """Read RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* (torch.distributed.run) and join the process group.
    backend "nccl" IS RCCL on ROCm; "gloo" is used by the CPU tests."""
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            # NSIM_DIST_BACKEND=gloo: two ranks sharing ONE GPU (RCCL refuses duplicate devices) -- test / measurement aid
            backend = os.environ.get("NSIM_DIST_BACKEND") or ("nccl" if device_type == "cuda" else "gloo")
        if device_type == "cuda":
            torch.cuda.set_device(local_rank)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local_rank, world


def get_rank():
    return dist.get_rank() if dist.is_initialized() else 0


def get_world_size():
    return dist.get_world_size() if dist.is_initialized() else 1


def is_master():
    return get_rank() == 0


def shard_range(n_total: int, rank: int, world: int):
    """Contiguous shard [lo, hi) of a global ray batch (render_parallel.py:248-252 scatters rays the same way)."""
    per = (n_total + world - 1) // world
    lo = min(rank * per, n_total)
    return lo, min(lo + per, n_total)


def allreduce_grads(params: Sequence[torch.Tensor], average: bool = True, small_numel: int = 1 << 20,
                    wire_dtype: Optional[torch.dtype] = None, skip_absent: bool = False):
    """Sum (or average) ``p.grad`` over ranks: big tensors on their own (asynchronously), everything else through
    one flat f32 bucket.  Parameters whose grad is None on this rank (e.g. no ray hit) contribute zeros.

    ``wire_dtype`` (default: env NSIM_ALLREDUCE_DTYPE = bf16 | f32, bf16 if unset): the big tensors (the 12.2 M-entry
    hash-table gradient, 48.8 MB in f32) travel in this type.  The reference trains fp16 parameters under DDP, i.e. it
    all-reduces 2-byte gradients as well (code_single/tools/train.py:1401-1412); bf16 keeps the f32 exponent range, so
    no loss scale is needed.  xGMI links are per-link bound: halving the bytes halves the exposed time of the one
    collective of the step.  A 2-byte wire uses the ``direct`` schedule (``_DirectToken``: one rounding per
    contribution, f32 accumulation) instead of the backend's ring all-reduce, which accumulates in the wire type.

    ``skip_absent``: parameters whose grad is None on EVERY rank stay without a gradient (one extra MAX-all-reduce of a
    presence vector decides it identically on all ranks) -- the optimizer then skips them, as ``torch.optim.Adam`` skips a
    None grad; zero-filling them instead would give them a momentum-only update that a single-GPU run does not make
    (the lidar step: no radiance / appearance / sky gradients on any rank)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return
    params = list(params)
    if skip_absent and params:
        ref = next((p for p in params if p.grad is not None), params[0])
        present = torch.tensor([0.0 if p.grad is None else 1.0 for p in params], dtype=torch.float32, device=ref.device)
        dist.all_reduce(present, op=dist.ReduceOp.MAX)
        keep = present.cpu().tolist()
        params = [p for p, k in zip(params, keep) if k > 0.5]
        if not params:
            return
    if wire_dtype is None:
        wire_dtype = {"bf16": torch.bfloat16, "f32": torch.float32, "fp16": torch.float16}[
            os.environ.get("NSIM_ALLREDUCE_DTYPE", "bf16")]
    world = dist.get_world_size()
    big, small = [], []
    for p in params:
        if p.grad is None:
            p.grad = torch.zeros_like(p, dtype=torch.float32)
        (big if p.grad.numel() >= small_numel else small).append(p)
    handles = []
    for p in big:
        if not p.grad.is_contiguous():
            p.grad = p.grad.contiguous()
        handles.append(allreduce_start(p.grad, wire_dtype))
    if small:
        flat = torch.cat([p.grad.reshape(-1) for p in small])
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
        off = 0
        for p in small:
            n = p.grad.numel()
            p.grad.copy_(flat[off:off + n].view_as(p.grad))
            off += n
    for tok in handles:
        allreduce_finish(tok)
    if average:
        for p in params:
            p.grad.div_(world)


def wire_dtype_default() -> torch.dtype:
    return {"bf16": torch.bfloat16, "f32": torch.float32, "fp16": torch.float16}[
        os.environ.get("NSIM_ALLREDUCE_DTYPE", "bf16")]


def _algo(wire_dtype: torch.dtype) -> str:
    """``direct`` (default for a 2-byte wire) | ``ring`` (the backend's own all-reduce; default for an f32 wire) --
    env NSIM_ALLREDUCE_ALGO overrides."""
    a = os.environ.get("NSIM_ALLREDUCE_ALGO")
    return a if a in ("direct", "ring") else ("ring" if wire_dtype == torch.float32 else "direct")


class _DirectToken:
    """Sum-all-reduce as all-to-all + local f32 reduction + all-gather.

    Why not the backend's all-reduce for the 2-byte wire: a ring all-reduce ACCUMULATES in the wire type -- a bf16 sum
    over 8 ranks rounds after every hop (seven roundings of a growing partial sum).  Here every rank's contribution is
    rounded to the wire type ONCE, the W contributions of a shard are summed in f32 on the rank that owns the shard, and
    the total is rounded once more for the way back: the error does not grow with the world size.  Bytes per rank are
    those of a ring (2 (W-1)/W of the buffer), and on xGMI -- a full mesh of point-to-point links, 7 per GPU -- every
    rank talks to every peer over its own link in both phases instead of pushing W-1 hops around one ring."""

    def __init__(self, t: torch.Tensor, wire_dtype: torch.dtype):
        W = dist.get_world_size()
        n = t.numel()
        chunk = (n + W - 1) // W
        flat = t.reshape(-1)
        send = torch.zeros([W * chunk], dtype=wire_dtype, device=t.device) if W * chunk != n else None
        if send is None:
            send = flat.to(wire_dtype).contiguous()
        else:
            send[:n] = flat
        self.t, self.n, self.W, self.chunk, self.wire = t, n, W, chunk, wire_dtype
        self.recv = torch.empty_like(send)
        self.send = send
        self.h = dist.all_to_all_single(self.recv, send, async_op=True)

    def finish(self) -> torch.Tensor:
        self.h.wait()
        part = self.recv.view(self.W, self.chunk).float().sum(0).to(self.wire)
        out = torch.empty([self.W * self.chunk], dtype=self.wire, device=self.t.device)
        dist.all_gather_into_tensor(out, part)
        self.t.reshape(-1).copy_(out[:self.n])
        return self.t


_DIRECT_OK = {}


def _direct_ok(device) -> bool:
    """One-time self-check of the direct exchange on this backend / device (every rank reaches it at the same collective):
    a 1000-element f32 sum (padding path included) through ``_DirectToken`` against the backend's own all-reduce.  A
    mismatch or an exception switches this process group to ``ring`` for good, with a warning -- the direct form has run
    on gloo (world sizes 2 and 8) and on RCCL only at world size 1 before the first multi-GPU run of a round."""
    key = (dist.get_backend(), str(device))
    if key in _DIRECT_OK:
        return _DIRECT_OK[key]
    if os.environ.get("NSIM_ALLREDUCE_ALGO") == "direct":      # forced: no check
        _DIRECT_OK[key] = True
        return True
    ok = True
    # (1) everything a single rank can fail at on its own -- allocation, dtype support, a missing all_to_all_single --
    # is tried WITHOUT a collective and agreed on (MIN) before any rank enters the exchange: a rank that raised alone
    # inside the exchange would leave its peers waiting in all_to_all.  (2) the exchange itself runs the same
    # code on the same shapes on every rank; what is left to diverge there is the transport, and that is the watchdog's
    # (NCCL_ASYNC_ERROR_HANDLING / the process group's timeout) to report.
    try:
        g = torch.Generator().manual_seed(1234 + dist.get_rank())
        x = torch.randn(1000, generator=g).to(device)
        W_ = dist.get_world_size()
        _probe = torch.zeros([W_ * ((1000 + W_ - 1) // W_)], dtype=torch.bfloat16, device=device)
        _probe[:1000] = x
        pre = callable(getattr(dist, "all_to_all_single", None)) and callable(getattr(dist, "all_gather_into_tensor", None))
    except Exception as e:      # noqa: BLE001
        print(f"[neuralsim_amd.distributed] direct all-reduce precondition raised {type(e).__name__}: {e}", flush=True)
        pre = False
    flag0 = torch.tensor([1.0 if pre else 0.0], device=device)
    dist.all_reduce(flag0, op=dist.ReduceOp.MIN)
    if flag0.item() < 0.5:
        print("[neuralsim_amd.distributed] direct all-reduce unavailable on some rank: using the backend's ring all-reduce",
              flush=True)
        _DIRECT_OK[key] = False
        return False
    try:
        ref = x.clone()
        dist.all_reduce(ref)
        y = x.clone()
        _DirectToken(y, torch.float32).finish()
        ok = bool(torch.allclose(y, ref, rtol=1e-5, atol=1e-5))
    except Exception as e:      # noqa: BLE001
        print(f"[neuralsim_amd.distributed] direct all-reduce self-check raised {type(e).__name__}: {e}", flush=True)
        ok = False
    try:
        flag = torch.tensor([1.0 if ok else 0.0], device=device)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        ok = bool(flag.item() > 0.5)
    except Exception:      # noqa: BLE001
        ok = False
    if not ok:
        print("[neuralsim_amd.distributed] direct 2-byte all-reduce failed its self-check on this backend: using the "
              "backend's ring all-reduce (NSIM_ALLREDUCE_ALGO=ring)", flush=True)
    _DIRECT_OK[key] = ok
    return ok


def allreduce_start(t: torch.Tensor, wire_dtype: Optional[torch.dtype] = None):
    """Begin an asynchronous sum-all-reduce of ``t`` (travelling as ``wire_dtype``); collectives complete in issue
    order, so a caller interleaves them with the kernels that produce the next tensor.  -> token for
    ``allreduce_finish``."""
    wire_dtype = torch.float32 if wire_dtype is None else wire_dtype
    if _algo(wire_dtype) == "direct" and t.is_contiguous() and _direct_ok(t.device):
        return _DirectToken(t, wire_dtype)
    buf = t if (wire_dtype == t.dtype and t.is_contiguous()) else t.to(wire_dtype).contiguous()
    return t, buf, dist.all_reduce(buf, op=dist.ReduceOp.SUM, async_op=True)


def allreduce_finish(token):
    """Wait for ``allreduce_start`` and write the sum back into the original tensor."""
    if isinstance(token, _DirectToken):
        return token.finish()
    t, buf, h = token
    h.wait()
    if buf is not t:
        t.copy_(buf)
    return t


class BackwardReducer:
    """The gradient exchange of an AUTOGRAD-path training step, overlapped with its backward (what DDP's bucketed
    all-reduce does for the reference: ``DDP(trainer, find_unused_parameters=True)``, code_single/tools/train.py:1401-1406).

    The steps of BASELINE configs[2..4] (street, indoor, multi-object) run several models through the autograd engine; the
    engine finishes their table gradients one after the other (a model queried LATER in the forward gets its gradient
    EARLIER).  A post-accumulate hook on every big parameter (>= ``small_numel`` entries: the hash tables) starts that
    parameter's all-reduce the moment its gradient exists, so that it travels while the other models' backward kernels --
    scatters of several ms -- still run; the small parameters go through one flat bucket after the backward, as in
    ``allreduce_grads``.

    Collectives are matched across ranks by ISSUE ORDER, and ranks may finish their gradients in different orders (or not at
    all: a rank whose rays missed a model).  The big parameters therefore leave in ONE fixed order on every rank: parameter k
    is released only once parameters 0..k-1 have been; whatever the backward did not release (in that order) is released by
    ``finish``, absent gradients as zeros.  The order is the one rank 0 observed in the first armed step (broadcast then):
    the first step runs without overlap, every later step overlaps whatever finishes in that order.

    Use: ``r.begin(); loss.backward(); r.finish_small(); optim.step(skip=r.big); for p in r.finish_big(): optim.step_range(p)``
    (``reduce_and_step`` does exactly that)."""

    def __init__(self, params: Sequence[torch.Tensor], small_numel: int = 1 << 20, wire_dtype: Optional[torch.dtype] = None):
        self.params = list(params)
        self.big = [p for p in self.params if p.numel() >= small_numel]
        self.wire = wire_dtype
        self._armed = False
        self._order_known = False
        self._fired: List[int] = []
        self._ready = [False] * len(self.big)
        self._toks = [None] * len(self.big)
        self._next = 0
        self.log: List[tuple] = []       # (big index, "backward" | "finish") per released parameter of the last step
        self._absent_here = set()        # ids of big parameters this rank had no gradient for (zeros went on the wire)
        self.absent = set()              # ids of the parameters NO rank had a gradient for in this step (after finish_small)
        self.on_release = None           # test hook: called with the big index right before its collective is issued
        self._pos = {id(p): i for i, p in enumerate(self.big)}       # position of a parameter in the CURRENT release order
        for p in self.big:
            p.register_post_accumulate_grad_hook(lambda p_: self._on_grad(self._pos[id(p_)]))

    def begin(self):
        self._armed = True
        self._fired = []
        self._ready = [False] * len(self.big)
        self._toks = [None] * len(self.big)
        self._next = 0
        self.log = []
        self._absent_here = set()
        self.absent = set()

    def _start(self, k: int, where: str):
        p = self.big[k]
        if p.grad is None:
            self._absent_here.add(id(p))          # zeros on the wire (the issue order is fixed); see finish_small
            p.grad = torch.zeros_like(p, dtype=torch.float32)
        elif not p.grad.is_contiguous():
            p.grad = p.grad.contiguous()
        if self.on_release is not None:
            self.on_release(k)
        wire = self.wire if self.wire is not None else wire_dtype_default()
        self._toks[k] = allreduce_start(p.grad, wire)
        self.log.append((k, where))

    def _on_grad(self, i: int):
        if not self._armed:
            return
        self._fired.append(i)
        self._ready[i] = True
        if not self._order_known:        # first armed step: observe only
            return
        while self._next < len(self.big) and self._ready[self._next]:
            self._start(self._next, "backward")
            self._next += 1

    def finish_small(self):
        """After the backward: release what is left of the big parameters (fixed order), then reduce the small ones through
        one flat f32 bucket (blocking: a few hundred KB) and write the sums back into their ``.grad``."""
        self._armed = False
        for k in range(self._next, len(self.big)):
            self._start(k, "finish")
        self._next = len(self.big)
        small = [p for p in self.params if not any(p is q for q in self.big)]
        # Presence agreement rides on the small bucket (ONE collective, at the same place of every rank's sequence): a flag per
        # parameter, 1 where this rank has a gradient.  A parameter NO rank has a gradient for -- a sky or distant model absent
        # from every rank's batch, the radiance heads in the lidar step -- keeps ``.grad = None`` and is not stepped, exactly as
        # at world size 1 (torch Adam and the reference's DDP(find_unused_parameters=True) skip it).  Rounds 4-5 zero-filled
        # it on every rank: a momentum-only update and an advanced step count, i.e. world size > 1 drifted from world size 1.
        if self.params:
            here = [0.0 if (p.grad is None or id(p) in self._absent_here) else 1.0 for p in self.params]
            dev = self.params[0].device
            flags = torch.tensor(here, dtype=torch.float32, device=dev)
            for p in small:
                if p.grad is None:
                    p.grad = torch.zeros_like(p, dtype=torch.float32)
            flat = torch.cat([p.grad.reshape(-1).float() for p in small] + [flags])
            dist.all_reduce(flat, op=dist.ReduceOp.SUM)
            seen = flat[flat.numel() - len(self.params):].tolist()
            self.absent = {id(p) for p, c in zip(self.params, seen) if c == 0.0}
            off = 0
            for p in small:
                n = p.grad.numel()
                if id(p) in self.absent:
                    p.grad = None
                else:
                    p.grad.copy_(flat[off:off + n].view_as(p.grad))
                off += n
        if not self._order_known:
            # adopt rank 0's completion order (parameters it never saw a gradient for go last, in index order)
            order = self._fired + [i for i in range(len(self.big)) if i not in self._fired]
            t = torch.tensor(order, dtype=torch.int64, device=self.big[0].device if self.big else "cpu")
            if self.big:
                dist.broadcast(t, src=0)
            order = [int(v) for v in t.tolist()]
            self.big = [self.big[i] for i in order]
            self._toks = [self._toks[i] for i in order]
            self._pos = {id(p): i for i, p in enumerate(self.big)}
            self._order_known = True

    def finish_big(self):
        """Yields every big parameter once its sum has arrived in ``.grad`` (issue order = arrival order)."""
        for k, p in enumerate(self.big):
            if self._toks[k] is not None:
                allreduce_finish(self._toks[k])
                self._toks[k] = None
            if id(p) in self.absent:     # zeros from every rank: no gradient anywhere, no step (see finish_small)
                p.grad = None
                continue
            yield p

    def drop_absent_big(self):
        """Wait out the (all-zero) exchange of the big parameters no rank had a gradient for and restore ``.grad = None``."""
        for k, p in enumerate(self.big):
            if id(p) in self.absent and self._toks[k] is not None:
                allreduce_finish(self._toks[k])
                self._toks[k] = None
                p.grad = None

    def reduce_and_step(self, optim, grad_scale: float):
        self.finish_small()
        self.drop_absent_big()
        optim.step(grad_scale=grad_scale, skip=tuple(p for p in self.big if id(p) not in self.absent))
        for p in self.finish_big():
            optim.step_range(p, 0, p.numel(), p.grad.reshape(-1), grad_scale=grad_scale)


def broadcast_module(module: torch.nn.Module, src: int = 0):
    """Make replicas bit-identical (parameters AND buffers such as the occupancy grid) -- what DDP does at
    construction / every forward for buffers (code_single/tools/train.py:1401-1406)."""
    if module is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return
    for t in list(module.parameters()) + list(module.buffers()):
        dist.broadcast(t.data, src=src)
