"""ctypes binding of libnsim_hip.so (include/nsim.h) -- the ONLY compute backend of this package.

There is no CPU fallback: if the gfx950 library is missing or an op is given a non-CUDA tensor the call
raises.  (The pure-PyTorch restatement under ``oracle/`` is test infrastructure and is never imported here.)
"""
import ctypes as C
import os
from pathlib import Path

import torch

NSIM_MAX_LEVELS = 32
_HERE = Path(__file__).resolve().parent
LIB_PATH = _HERE / "csrc" / "libnsim_hip.so"


class LotdMeta(C.Structure):
    _fields_ = [("num_levels", C.c_int32), ("n_feats", C.c_int32), ("n_active_levels", C.c_int32),
                ("res", (C.c_int32 * 3) * NSIM_MAX_LEVELS),
                ("type", C.c_int32 * NSIM_MAX_LEVELS), ("size", C.c_uint32 * NSIM_MAX_LEVELS),
                ("offset", C.c_int64 * NSIM_MAX_LEVELS), ("x_scale", C.c_float * 3), ("x_shift", C.c_float * 3)]


class OccMeta(C.Structure):
    _fields_ = [("aabb_min", C.c_float * 3), ("aabb_max", C.c_float * 3), ("scale", C.c_float * 3),
                ("res", C.c_int32 * 3)]


class Lotd4Meta(C.Structure):
    _fields_ = [("num_levels", C.c_int32), ("res_xyz", C.c_int32 * 16), ("res_w", C.c_int32 * 16),
                ("type", C.c_int32 * 16), ("size", C.c_uint32 * 16), ("offset", C.c_int64 * 16),
                ("res_y", C.c_int32 * 16), ("res_z", C.c_int32 * 16)]


class PermutoMeta(C.Structure):
    _fields_ = [("in_dim", C.c_int32), ("num_levels", C.c_int32), ("n_feats", C.c_int32), ("hashmap_size", C.c_uint32),
                ("scale", (C.c_float * 8) * NSIM_MAX_LEVELS), ("shift", (C.c_float * 8) * NSIM_MAX_LEVELS)]


class DistantMeta(C.Structure):
    _fields_ = [("lotd", Lotd4Meta), ("precision", C.c_int32)]


class SkyMeta(C.Structure):
    _fields_ = [("n_frequencies", C.c_int32), ("n_appear", C.c_int32), ("precision", C.c_int32)]


class AdamTensor(C.Structure):
    _fields_ = [("p", C.c_void_p), ("p16", C.c_void_p), ("grad", C.c_void_p), ("m", C.c_void_p), ("v", C.c_void_p),
                ("n", C.c_int64), ("beta1", C.c_float), ("beta2", C.c_float), ("bias1", C.c_float), ("bias2", C.c_float),
                ("lr_scale", C.c_float)]


ADAM_MULTI_MAX = 12


def plane_pitch(S: int) -> int:
    """NSIM_PLANE_PITCH (include/nsim.h): pitch of the h / dh-dx planes handed from nsim_field_fwd to the backward."""
    return (int(S) + 31) & ~31


def jplane_dtype(fm):
    """torch dtype of the dh/dx planes (J_planes) the kernels selected by ``fm`` write / read: f16 in the fp16 field mode, else f32
    (nsim_jplane_elem_bytes, include/nsim.h)."""
    import torch
    return torch.float16 if int(get_lib().nsim_jplane_elem_bytes(fm)) == 2 else torch.float32


class FieldMeta(C.Structure):
    _fields_ = [("lotd", LotdMeta), ("sdf_D", C.c_int32), ("precision", C.c_int32), ("softplus_beta", C.c_float),
                ("embed_E", C.c_int32)]


class ComposeSrc(C.Structure):      # include/nsim.h NsimComposeSrc
    _fields_ = [("t", C.c_void_p), ("rays_inds", C.c_void_p), ("pack_infos", C.c_void_p), ("P", C.c_int64), ("dst", C.c_void_p)]


_P = C.c_void_p
_I64 = C.c_int64
_I = C.c_int
_F = C.c_float

# name -> argtypes (every function additionally takes the trailing ``void* stream`` unless listed in _NOSTREAM)
SIGNATURES = {
    "nsim_pack_infos_from_n": [_P, _I64, _P, _P, _I64],
    "nsim_pack_infos_from_n_notify": [_P, _I64, _P, _P, _I64, _P, _I64],
    "nsim_live_rank": [_P, _I64, _I, _I, _I, _I, _I, _P, _P, _P, _P, _I64, _P, _P, _I64, _P, _I64],
    "nsim_packed_sum": [_P, _I, _P, _I64, _P],
    "nsim_packed_binary": [_P, _I, _P, _I, _P, _I64, _I, _P],
    "nsim_packed_cmp": [_P, _P, _P, _I64, _I, _P],
    "nsim_packed_matmul3": [_P, _P, _P, _I64, _I, _P],
    "nsim_packed_sort": [_P, _P, _I64, _P, _P],
    "nsim_interleave_linstep": [_P, _P, _I64, _I64, _P],
    "nsim_merge_two_packs": [_P, _P, _P, _I64, _P, _P, _P, _I64, _P, _I64, _P, _P],
    "nsim_alpha_to_vw_fwd": [_P, _P, _I64, _P, _P],
    "nsim_alpha_to_vw_bwd": [_P, _P, _P, _P, _P, _I64, _P],
    "nsim_composite_fwd": [_P, _P, _P, _P, _P, _I64, _I, _P, _P, _P, _P, _P, _P, _P],
    "nsim_neus_composite_fwd": [_P, _P, _F, _F, _P, _P, _P, _P, _I64, _I, _P, _P, _P, _P, _P, _P, _P, _P],
    "nsim_render_head": [_P, _P, _F, _F, _P, _P, _P, _P, _I64, _I, _P, _I64, _I64, _I64, _F, _P] + [_P] * 13,
    "nsim_composite_bwd": [_P, _P, _P, _P, _P, _P, _P, _I64, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P],
    "nsim_neus_alpha_fwd": [_P, _P, _I64, _P, _F, _F, _P],
    "nsim_neus_alpha_bwd": [_P, _P, _P, _I64, _P, _F, _F, _P, _P],
    "nsim_raygen_pinhole": [_P, _P, _P, _P, _P, _I64, _I, _P, _P],
    "nsim_raygen_pinhole_bwd": [_P, _P, _P, _P, _P, _I64, _I, _P, _P, _P],
    "nsim_raygen_opencv": [_P, _P, _P, _P, _I, _P, _P, _I64, _I, _P, _P],
    "nsim_raygen_opencv_bwd": [_P, _P, _P, _P, _I, _P, _P, _I64, _I, _P, _P, _P],
    "nsim_raygen_fisheye": [_P, _P, _P, _P, _I, _P, _P, _I64, _I, _P, _P],
    "nsim_raygen_fisheye_bwd": [_P, _P, _P, _P, _I, _P, _P, _I64, _I, _P, _P, _P],
    "nsim_aabb_ray_test": [_P, _P, _I64, C.POINTER(OccMeta), _F, _F, _P, _P, _P],
    "nsim_occ_decay": [_P, _I64, _F],
    "nsim_occ_update": [_P, _P, _P, _I64, C.POINTER(OccMeta), _F],
    "nsim_occ_pack_bits": [_P, _I64, _F, _P],
    "nsim_occ_collect": [_P, _P, _P, _I64, _P, _I64, C.POINTER(OccMeta), _F],
    "nsim_march_count": [_P, _P, _P, _P, _P, _I64, _P, _P, C.POINTER(OccMeta), _F, _I, _P],
    "nsim_march_emit": [_P, _P, _P, _P, _P, _I64, _P, _P, C.POINTER(OccMeta), _F, _I, _P, _P],
    "nsim_coarse_depths": [_P, _P, _P, _I64, _I, _P, _P],
    "nsim_upsample_stage": [_P, _P, _P, _I64, _F, _I, _I, _P, _P, _P, _P, _P, _P],
    "nsim_merge_sorted": [_P, _P, _P, _P, _P, _I64, _I, _P, _P, _P, _P, _P, _P, _P, _P],
    "nsim_merge_upsample": [_P, _P, _P, _P, _P, _I64, _I, _P, _P, _P, _P, _F, _I, _I, _P, _P, _P, _P, _P, _P],
    "nsim_compress_count": [_P, _P, _I64, _P, _F, _F, _F, _P],
    "nsim_compress_emit": [_P, _P, _P, _I64, _P, _F, _F, _F, _P, _P, _P, _I64],
    "nsim_lotd_fwd": [_P, _P, C.POINTER(LotdMeta), _I64, _P, _P],
    "nsim_lotd_bwd": [_P, _P, _P, C.POINTER(LotdMeta), _I64, _P],
    "nsim_field_pack_weights": [C.POINTER(FieldMeta), _P, _P, _P, _P, _P],
    "nsim_field_sdf": [C.POINTER(FieldMeta), _P, _P, _P, _P, _P, _P, _P, _P, _I64, _P, _I64, _P, _P, _P, C.POINTER(OccMeta), _F],
    "nsim_lotd_gather_lm": [C.POINTER(FieldMeta), _P, _P, _P, _P, _P, _P, _P, _I64, _P, _I64, _P],
    "nsim_field_fwd": [C.POINTER(FieldMeta), _P, _P, _P, _P, _P, _P, _P, _P, _P, _I64, _P, _P, _P, _P, _P, _P, _I64],
    "nsim_compose_collect_sort": [_P, _I, _P, _I64, _P],
    "nsim_wide_sdf": [C.POINTER(FieldMeta), _I, _P, _P, _P, _P, _P, _P, _P, _I64, _P, _I64, _P, _P],
    "nsim_wide_bwd_sdf": [C.POINTER(FieldMeta), _P, _P, _P, _P, _P, _P, _I64, _P, _P, _I64, _P, _P, _P, _P, _P, _P],
    "nsim_set_grad_scratch": [_P, _I64],
    "nsim_permuto_fwd": [C.POINTER(PermutoMeta), _P, _P, _I64, _P, _P],
    "nsim_permuto_bwd": [C.POINTER(PermutoMeta), _P, _I64, _P, _P],
    "nsim_permuto_gather": [C.POINTER(PermutoMeta), _P, _P, _P, _P, _P, _P, _P, _I64, _P, _I64, _P, _I, _P, _P],
    "nsim_permuto_scatter": [C.POINTER(PermutoMeta), _P, _P, _P, _P, _P, _P, _I64, _P, _P, _P, _P],
    "nsim_permuto_dz": [C.POINTER(PermutoMeta), _P, _P, _P, _P, _P, _P, _P, _I64, _P, _P],
    "nsim_field_bwd_rad": [C.POINTER(FieldMeta), _P, _P, _P, _P, _P, _P, _P, _P, _P, _I64, _P, _P, _P, _P, _P, _P, _P, _P],
    "nsim_field_bwd_sdf": [C.POINTER(FieldMeta), _P, _P, _P, _I64, _P, _P, _P, _P, _P, _P, _P, _I64],
    "nsim_lotd_hess_dx": [C.POINTER(LotdMeta), _P, _P, _P, _P, _P, _P, _P, _I64, _P, _P, _P],
    "nsim_ray_grad_reduce": [_P, _P, _P, _P, _I64, _P, _P],
    "nsim_lotd_scatter": [C.POINTER(LotdMeta), _P, _P, _P, _P, _P, _P, _I64, _P, _P, _P, _P, _I, _I],
    "nsim_distant_pack_weights": [C.POINTER(DistantMeta), _P, _P, _P, _P, _P],
    "nsim_distant_shells": [_P, _P, _P, _P, _I64, _I, C.POINTER(C.c_float * 6), _F, _F, _P, _P, _P],
    "nsim_density_alpha_fwd": [_P, _P, _P, _I64, _I, _I, _P],
    "nsim_density_alpha_bwd": [_P, _P, _P, _P, _I64, _I, _I, _P],
    "nsim_distant_fwd": [C.POINTER(DistantMeta), _P, _P, _P, _P, _P, _I64, _I, _P, _P, _P],
    "nsim_distant_bwd": [C.POINTER(DistantMeta), _P, _P, _P, _P, _P, _P, _P, _I64, _I, _P, _P, _P, _P, _P, _P, _P, _P],
    "nsim_lotd4_scatter": [C.POINTER(Lotd4Meta), _P, _P, _I64, _P, _P],
    "nsim_sky_pack_weights": [C.POINTER(SkyMeta), _P, _P, _P],
    "nsim_sky_fwd": [C.POINTER(SkyMeta), _P, _P, _P, _I64, _P, _P],
    "nsim_sky_bwd": [C.POINTER(SkyMeta), _P, _P, _P, _I64, _P, _P, _P, _P, _P],
    "nsim_eikonal_loss_fwd": [_P, _I64, _P],
    "nsim_eikonal_loss_bwd": [_P, _I64, _P, _P],
    "nsim_mse_loss_fwd": [_P, _P, _I64, _P],
    "nsim_train_loss_head": [_P, _P, _I64, _P, _I64, _I64, _F, _P, _P, _P],
    "nsim_mse_loss_bwd": [_P, _P, _I64, _P, _P],
    "nsim_rows_scatter_add": [_P, _P, _I64, _I, _I64, _P],
    "nsim_rows_gather": [_P, _P, _I64, _I, _I64, _I64, _P],
    "nsim_field_pack_weights2": [C.POINTER(FieldMeta), _P, C.POINTER(FieldMeta), _P, _P, _P, _P, _P],
    "nsim_gather_rays": [_P, _P, _P, _P, _P, _I64, _P, _P, _P, _P],
    "nsim_sphere_image": [_P, _P, _I64, _F, _P],
    "nsim_adam_step": [_P, _P, _P, _P, _P, _I64, _F, _F, _F, _F, _F, _F, _F, _I],
    "nsim_adam_multi": [C.POINTER(AdamTensor), _I, _F, _F, _F, _I],
    "nsim_selftest_mfma": [_P, _P, _P, _I],
}
NOSTREAM = {
    "nsim_strerror": ([_I], C.c_char_p),
    "nsim_version": ([], _I),
    "nsim_field_wpack_bytes": ([C.POINTER(FieldMeta)], _I64),
    "nsim_jplane_elem_bytes": ([C.POINTER(FieldMeta)], _I),
    "nsim_distant_wpack_bytes": ([C.POINTER(DistantMeta)], _I64),
    "nsim_sky_wpack_bytes": ([C.POINTER(SkyMeta)], _I64),
    "nsim_sky_plane_pitch": ([_I64], _I64),
}


def bind(cdll):
    """Attach argtypes/restype for every symbol of include/nsim.h (raises AttributeError if one is missing)."""
    for name, args in SIGNATURES.items():
        fn = getattr(cdll, name)
        fn.argtypes = list(args) + [_P]
        fn.restype = _I
    for name, (args, res) in NOSTREAM.items():
        fn = getattr(cdll, name)
        fn.argtypes = list(args)
        fn.restype = res
    return cdll


_LIB = None


def get_lib():
    """The gfx950 library; raises loudly when it has not been built (``python -m neuralsim_amd.csrc.build``)."""
    global _LIB
    if _LIB is None:
        if not LIB_PATH.exists():
            raise RuntimeError(
                f"neuralsim_amd: {LIB_PATH} is missing. This package has no CPU fallback -- build the HIP "
                f"extension with `python -m neuralsim_amd.csrc.build` (needs hipcc, --offload-arch=gfx950).")
        _LIB = bind(C.CDLL(str(LIB_PATH)))
    return _LIB


def stream_handle() -> int:
    """Current HIP stream of the current device (ops run on the caller's stream)."""
    return torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice())


def require_device(t: torch.Tensor, name: str = "tensor"):
    if not t.is_cuda:
        raise RuntimeError(f"neuralsim_amd: {name} must live on a HIP device (got {t.device}); there is no CPU path")


def ptr(t, dtype=None, name="tensor"):
    """Marks a tensor argument of a C call (None -> NULL).  The tensor itself travels to ``call`` -- which turns it into
    its device pointer -- so the argument tuple keeps every temporary (``g.contiguous()`` ...) alive for the duration of
    the launch; a bare ``data_ptr()`` of a temporary would let the caching allocator hand the block to the next one."""
    if t is not None and dtype is not None and t.dtype != dtype:
        raise TypeError(f"neuralsim_amd: {name} must be {dtype}, got {t.dtype}")
    return t


class HostNotify:
    """Host-mapped (pinned) words a kernel stores a size to, read by the host WITHOUT a stream synchronisation or a copy
    (``nsim_pack_infos_from_n_notify``): slot s = (value, seq).  ``arm(s)`` -> (address of the slot, the sequence number
    the kernel must store); ``wait(s, seq)`` spins until the slot carries seq and returns the value, or None after
    ``timeout_s`` (the caller then falls back to a synchronising read of the device copy)."""

    def __init__(self, slots: int = 2):
        buf = torch.zeros([slots, 2], dtype=torch.long)
        self.buf = buf.pin_memory() if torch.cuda.is_available() else buf
        self.view = self.buf.numpy()
        self.seq = 0

    def __deepcopy__(self, memo):         # a copy of a model gets words of its own
        return HostNotify(self.buf.shape[0])

    def __reduce__(self):
        return (HostNotify, (self.buf.shape[0],))

    def arm(self, slot: int):
        self.seq += 1
        return self.buf.data_ptr() + 16 * slot, self.seq

    def wait(self, slot: int, seq: int, timeout_s: float = 10.0):
        v = self.view
        n = 0
        t0 = None
        while int(v[slot, 1]) != seq:
            n += 1
            if n & 1023 == 0:
                import time
                now = time.perf_counter()
                t0 = now if t0 is None else t0
                if now - t0 > timeout_s:
                    return None
        return int(v[slot, 0])


class KernelTimer:
    """Optional per-entry-point HIP-event timing on the launch stream (used by bench.py for the roofline)."""

    def __init__(self, only=None):
        self.events = {}
        self.units = {}
        self.only = set(only) if only is not None else None

    def note_units(self, name, n):
        self.units[name] = self.units.get(name, 0) + int(n)

    def summary(self):
        torch.cuda.synchronize()
        out = {}
        for name, evs in self.events.items():
            ms = [a.elapsed_time(b) for a, b in evs]
            out[name] = dict(calls=len(ms), total_ms=sum(ms), avg_ms=sum(ms) / max(1, len(ms)),
                             units=self.units.get(name, 0))
        return out


TIMER = None   # set to a KernelTimer() to record events around the calls named in TIMER.only (all if None)


_Tensor = torch.Tensor


def _marshal(args):
    out = []
    ap = out.append
    for a in args:
        if isinstance(a, _Tensor):
            if not a.is_cuda:                 # (one attribute read on the hot path; the refusal itself is require_device's)
                require_device(a)
            if not a.is_contiguous():
                raise ValueError("neuralsim_amd: tensor arguments must be contiguous")
            ap(a.data_ptr())
        else:
            ap(a)
    return out


_GRAD_SCRATCH = {}
_GRAD_SCRATCH_USERS = ("nsim_field_bwd_rad", "nsim_field_bwd_sdf")
GRAD_SCRATCH_FLOATS = 16 * 8448       # 16 replicas of the largest decoder gradient (64 x 64 + 64 x 64 + 64 + 129 floats)


def ensure_grad_scratch(device):
    """Register (once per device and stream) the zeroed scratch the backward launches spread their weight-gradient
    flush over (include/nsim.h: nsim_set_grad_scratch); the tensor is kept alive here."""
    key = (str(device), stream_handle())
    if key not in _GRAD_SCRATCH:
        buf = torch.zeros([GRAD_SCRATCH_FLOATS], dtype=torch.float32, device=device)
        _GRAD_SCRATCH[key] = buf
        call("nsim_set_grad_scratch", buf, buf.numel())
    return _GRAD_SCRATCH[key]


CALL_COUNT = None        # bench.py sets it to 0 over the timed steps: number of C-ABI calls issued
# ------------------------------------------------------------------------------------------------ zero arena
# The autograd-path step (``ray_test`` + ``ray_query`` + autograd: what a reference renderer drives) zero-fills ~19 buffers per
# iteration -- gradient accumulators, images that hit rays are scattered into, loss scalars --, each its own fill launch
# (58 us of GPU time and ~19 allocator + launch round trips on the host per step, profiles/round6_step_kernels_api.txt).  A
# trainer that owns the step brackets it with ``arena_begin`` / ``arena_end``: ONE zeroed f32 buffer sized from the previous
# step's demand, ``zeros()`` hands out 16-byte aligned views of it.  Outside a bracket (the reference's own trainer on the shim,
# the tests' direct calls) and for anything that does not fit, ``zeros()`` IS ``torch.zeros``.
class _Arena:
    __slots__ = ("buf", "off", "cap", "demand", "device", "stream")


_ARENA = None
_ARENA_DEMAND = {}


def arena_begin(device):
    global _ARENA
    # OFF by default -- a measured null (profiles/round6_api_path_arena_ab.txt, MI355X, alternated in one call): API path 1.84 /
    # 1.90 ms per step without, 1.92 / 1.97 with; street 12.2 vs 12.7, multi-object 15.5 vs 15.9.  The fills it removes are
    # not on the critical path (the stretch between the forward and the large backward kernels is paced by the host's
    # autograd nodes, not by the GPU), and one large memset at the start of the step is in the way of the sampling pass.
    if os.environ.get("NSIM_ZERO_ARENA", "0") != "1":
        _ARENA = None
        return
    a = _Arena()
    a.device, a.off, a.demand = device, 0, 0
    want = _ARENA_DEMAND.get(str(device), 0)
    a.cap = (int(want * 1.125) + 1024) if want else 0
    a.buf = torch.zeros([a.cap], dtype=torch.float32, device=device) if a.cap else None
    a.stream = stream_handle() if device.type == "cuda" else 0
    _ARENA = a


def arena_end():
    global _ARENA
    a = _ARENA
    if a is not None:
        _ARENA_DEMAND[str(a.device)] = a.demand
    _ARENA = None


def zeros(shape, dtype=torch.float32, device=None):
    """``torch.zeros(shape, dtype=torch.float32, device=device)``, as a view of the step's arena when one is open (same device,
    same stream -- the prefetch of the next batch runs on its own stream and must not touch a buffer the main stream zeroes)."""
    a = _ARENA
    if a is None or dtype != torch.float32 or device != a.device:
        return torch.zeros(shape, dtype=dtype, device=device)
    n = 1
    for d_ in shape:
        n *= int(d_)
    pad = (n + 3) & ~3
    a.demand += pad
    if a.off + pad > a.cap or (a.device.type == "cuda" and stream_handle() != a.stream):
        return torch.zeros(shape, dtype=dtype, device=device)
    v = a.buf[a.off:a.off + n].view(shape)
    a.off += pad
    return v


HOST_WAIT = None         # bench.py sets it to 0.0: seconds the host spent blocked on the step's one size read (fields/neus.py _compress)


def call(name: str, *args):
    """Invoke a C-ABI entry point on the current stream and raise on a non-zero return code.  Tensor arguments are
    passed as their device pointers (checked: device-resident, contiguous)."""
    global CALL_COUNT
    if CALL_COUNT is not None:
        CALL_COUNT += 1
    lib = get_lib()
    if name in _GRAD_SCRATCH_USERS:
        for a_ in args:
            if isinstance(a_, _Tensor):
                ensure_grad_scratch(a_.device)
                break
    cargs = _marshal(args)
    if TIMER is not None and (TIMER.only is None or name in TIMER.only):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        rc = getattr(lib, name)(*cargs, stream_handle())
        e1.record()
        TIMER.events.setdefault(name, []).append((e0, e1))
    else:
        rc = getattr(lib, name)(*cargs, stream_handle())
    if rc != 0:
        msg = lib.nsim_strerror(rc)
        raise RuntimeError(f"{name} failed with code {rc}: {msg.decode() if msg else '?'}")
