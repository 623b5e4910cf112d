// nsim_prims.h -- the hardware primitives of the kernels, gfx950 (MI355X / CDNA4) implementation: cross-lane moves on
// DPP / ds_bpermute, MFMA, raw transcendental pipes, buffer loads, LDS-DMA, system-scope stores.  nsim_common.h includes
// it as <nsim_prims.h>; this directory is the only place the product build looks.  (The test-only host emulator puts its
// own file of the same name earlier on the include path of ITS build, tests/emu/; nothing in this tree refers to it.)
#pragma once
#include <hip/hip_runtime.h>

#define NSIM_DYN_SMEM(name) extern __shared__ __attribute__((aligned(16))) char name[]

// ------------------------------------------------------------------ cross-lane
__device__ __forceinline__ int nsim_lane() { return (int)(threadIdx.x & 63); }

template <class T>
__device__ __forceinline__ T wave_shfl(T v, int src) {
  return __shfl(v, src, 64);
}

__device__ __forceinline__ unsigned long long wave_ballot(int pred) { return __ballot(pred); }

// broadcast the value of lane I of every aligned group of 4 lanes (DPP quad_perm)
template <int I, class T>
__device__ __forceinline__ T quad_bcast(T v) {
  static_assert(sizeof(T) == 4, "quad_bcast: 32-bit types only");
  int iv;
  __builtin_memcpy(&iv, &v, 4);
  iv = __builtin_amdgcn_update_dpp(0, iv, I | (I << 2) | (I << 4) | (I << 6), 0xf, 0xf, true);
  T r;
  __builtin_memcpy(&r, &iv, 4);
  return r;
}

// float scans / reductions stay in the VALU (the generic shuffle forms of nsim_common.h serve every other type).
// Scans: DPP row_shr 1/2/4/8 scans every row of 16, row_bcast15 / row_bcast31 carry the row totals forward (lanes without
// a source keep the identity) -- six VALU steps, no LDS crossbar.
#define NSIM_DPP_OLD_F32(oldv, x, ctrl, rmask)                                                              \
  __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, (float)(oldv)),             \
                                                        __builtin_bit_cast(int, (x)), (ctrl), (rmask), 0xf, false))
#define NSIM_DPP_F32(x, ctrl) \
  __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, (x)), (ctrl), 0xf, 0xf, true))

__device__ __forceinline__ float nsim_readlane_f32(float v, int l) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l));
}

__device__ __forceinline__ float wave_incl_sum(float v) {
  v += NSIM_DPP_OLD_F32(0.f, v, 0x111, 0xf);
  v += NSIM_DPP_OLD_F32(0.f, v, 0x112, 0xf);
  v += NSIM_DPP_OLD_F32(0.f, v, 0x114, 0xf);
  v += NSIM_DPP_OLD_F32(0.f, v, 0x118, 0xf);
  v += NSIM_DPP_OLD_F32(0.f, v, 0x142, 0xa);   // row_bcast15 into rows 1 and 3
  v += NSIM_DPP_OLD_F32(0.f, v, 0x143, 0xc);   // row_bcast31 into rows 2 and 3
  return v;
}

// the same six DPP steps for a 32-bit integer (ray / sample counts below 2^31; 64-bit sums take the shuffle form)
#define NSIM_DPP_OLD_I32(oldv, x, ctrl, rmask) __builtin_amdgcn_update_dpp((int)(oldv), (int)(x), (ctrl), (rmask), 0xf, false)
__device__ __forceinline__ int wave_incl_sum(int v) {
  v += NSIM_DPP_OLD_I32(0, v, 0x111, 0xf);
  v += NSIM_DPP_OLD_I32(0, v, 0x112, 0xf);
  v += NSIM_DPP_OLD_I32(0, v, 0x114, 0xf);
  v += NSIM_DPP_OLD_I32(0, v, 0x118, 0xf);
  v += NSIM_DPP_OLD_I32(0, v, 0x142, 0xa);   // row_bcast15 into rows 1 and 3
  v += NSIM_DPP_OLD_I32(0, v, 0x143, 0xc);   // row_bcast31 into rows 2 and 3
  return v;
}

// ... and for a 64-bit integer: the two words travel separately, the carry of the low word is added locally (the shuffle
// form costs two ds_bpermute per step on the LDS crossbar; the single-workgroup scans of pack_ops.hip sit on the critical path of
// the sampling pass)
__device__ __forceinline__ int64_t wave_incl_sum(int64_t v) {
  uint32_t lo = (uint32_t)(uint64_t)v, hi = (uint32_t)((uint64_t)v >> 32);
#define NSIM_SCAN64_STEP(ctrl, rmask)                                  \
  {                                                                    \
    const uint32_t sl = (uint32_t)NSIM_DPP_OLD_I32(0, lo, ctrl, rmask); \
    const uint32_t sh = (uint32_t)NSIM_DPP_OLD_I32(0, hi, ctrl, rmask); \
    const uint32_t nl = lo + sl;                                       \
    hi = hi + sh + (nl < lo ? 1u : 0u);                                \
    lo = nl;                                                           \
  }
  NSIM_SCAN64_STEP(0x111, 0xf)
  NSIM_SCAN64_STEP(0x112, 0xf)
  NSIM_SCAN64_STEP(0x114, 0xf)
  NSIM_SCAN64_STEP(0x118, 0xf)
  NSIM_SCAN64_STEP(0x142, 0xa)
  NSIM_SCAN64_STEP(0x143, 0xc)
#undef NSIM_SCAN64_STEP
  return (int64_t)(((uint64_t)hi << 32) | (uint64_t)lo);
}

__device__ __forceinline__ float wave_incl_prod(float v) {
  v *= NSIM_DPP_OLD_F32(1.f, v, 0x111, 0xf);
  v *= NSIM_DPP_OLD_F32(1.f, v, 0x112, 0xf);
  v *= NSIM_DPP_OLD_F32(1.f, v, 0x114, 0xf);
  v *= NSIM_DPP_OLD_F32(1.f, v, 0x118, 0xf);
  v *= NSIM_DPP_OLD_F32(1.f, v, 0x142, 0xa);
  v *= NSIM_DPP_OLD_F32(1.f, v, 0x143, 0xc);
  return v;
}

// Reductions: four DPP steps reduce every row of 16 lanes (quad_perm xor 1 / xor 2, row_half_mirror, row_mirror), four
// v_readlane + scalar-side combine finish across rows -- ~10 issue slots, against six dependent ds_bpermute round trips
// (~100 cycles each) for the shuffle tree.
__device__ __forceinline__ float wave_sum(float v) {
  v += NSIM_DPP_F32(v, 0xB1);
  v += NSIM_DPP_F32(v, 0x4E);
  v += NSIM_DPP_F32(v, 0x141);
  v += NSIM_DPP_F32(v, 0x140);
  return (nsim_readlane_f32(v, 0) + nsim_readlane_f32(v, 16)) + (nsim_readlane_f32(v, 32) + nsim_readlane_f32(v, 48));
}

__device__ __forceinline__ float wave_max(float v) {
  v = fmaxf(v, NSIM_DPP_F32(v, 0xB1));
  v = fmaxf(v, NSIM_DPP_F32(v, 0x4E));
  v = fmaxf(v, NSIM_DPP_F32(v, 0x141));
  v = fmaxf(v, NSIM_DPP_F32(v, 0x140));
  return fmaxf(fmaxf(nsim_readlane_f32(v, 0), nsim_readlane_f32(v, 16)),
               fmaxf(nsim_readlane_f32(v, 32), nsim_readlane_f32(v, 48)));
}

// ------------------------------------------------------------------ ordering inside a wave
// LDS hand-off between lanes of ONE wave (writes by some lanes, reads by others)
__device__ __forceinline__ void wave_sync_lds() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// the same for values handed over through memory (LDS or global) by plain stores
__device__ __forceinline__ void nsim_wave_fence() { __threadfence_block(); }
// a zero the compiler cannot see through (keeps loop-invariant address arithmetic from being hoisted into registers)
__device__ __forceinline__ int nsim_opaque_zero() {
  int v = 0;
  asm volatile("" : "+v"(v));
  return v;
}

// ------------------------------------------------------------------ transcendental pipes
// v_exp_f32 / v_log_f32 based fast forms, and the raw base-2 pipes (the ranges used never reach their denormal corners)
__device__ __forceinline__ float nsim_fast_exp(float x) { return __expf(x); }
__device__ __forceinline__ float nsim_fast_log(float x) { return __logf(x); }
__device__ __forceinline__ float nsim_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
__device__ __forceinline__ float nsim_log2(float x) { return __builtin_amdgcn_logf(x); }
// v_sin_f32 / v_cos_f32 (input in revolutions, range reduction in the pipe): |x| of a few tens, abs. error ~2e-6 -- the
// libm forms carry a Payne-Hanek slow path that costs every lane 84 bytes of scratch
__device__ __forceinline__ float nsim_sin(float x) { return __builtin_amdgcn_sinf(x * 0.15915494309189535f); }
__device__ __forceinline__ float nsim_cos(float x) { return __builtin_amdgcn_cosf(x * 0.15915494309189535f); }

// ------------------------------------------------------------------ direct global -> LDS copies
// global_load_lds_dwordx4: every lane names its own 16-byte global source, the destination is the wave-uniform LDS base +
// 16 * lane; completion is tracked by vmcnt.
__device__ __forceinline__ void nsim_glds16(const void* gsrc, char* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}
__device__ __forceinline__ void nsim_wait_vm0() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
__device__ __forceinline__ void nsim_wait_lgkm0() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// ------------------------------------------------------------------ MFMA
// v_mfma_f32_32x32x16_f16: A lane l -> row (l&31), B lane l -> col (l&31), 8 K-slots per lane indexed by (l>>5, e);
// C/D: col = l&31, row = (r&3) + 8*(r>>2) + 4*(l>>5).  (CDNA4 guide, "Fragment layout".)
__device__ __forceinline__ f32x16 mfma_32x32x16_f16(f16x8 a, f16x8 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}
// v_mfma_f32_32x32x16_bf16: same fragment layout and rate as the f16 form, operands with the f32 exponent range.
__device__ __forceinline__ f32x16 mfma_32x32x16_bf16(bf16x8 a, bf16x8 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
// v_mfma_f32_32x32x2_f32: exact-f32 matrix FMA, one K-slot per lane (k = l>>5).
__device__ __forceinline__ f32x16 mfma_32x32x2_f32(float a, float b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}
// two f32 -> two bf16 in ONE v_cvt_pk_bf16_f32 (the halves then leave through ds_write_b16 / ds_write_b16_d16_hi): the LDS staging of
// the weight-gradient operands converts 16 values per m-tile, one conversion instruction each when written as scalar casts
__device__ __forceinline__ void nsim_cvt2_bf16(float a, float b, bf16& lo, bf16& hi) {
  typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  const f32x2 v = {a, b};
  const bf16x2 r = __builtin_convertvector(v, bf16x2);
  lo = r[0];
  hi = r[1];
}

// sum of the eight bf16 of a fragment into s: v_dot2c_f32_bf16 against (1, 1) adds two per issue
__device__ __forceinline__ float nsim_bf16x8_sum(bf16x8 v, float s) {
  typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
  const bf16x2 one = {(bf16)1.0f, (bf16)1.0f};
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const bf16x2 pr = {v[2 * e], v[2 * e + 1]};
    s = __builtin_amdgcn_fdot2_f32_bf16(pr, one, s, false);
  }
  return s;
}

// ------------------------------------------------------------------ the LoTD table behind a buffer resource
// ONE buffer resource (4 SGPRs) + a 32-bit per-lane byte offset: half the address registers and no 64-bit address
// arithmetic per corner compared with flat loads (the table is 24.4 MB, far below 4 GiB).
struct GridRef {
  __amdgpu_buffer_rsrc_t r;
};
__device__ __forceinline__ GridRef grid_ref(const f16* grid) {
  GridRef g;
  g.r = __builtin_amdgcn_make_buffer_rsrc(const_cast<f16*>(grid), 0, 0xffffffff, 0x00020000);
  return g;
}
// the 32-bit word (two fp16 features) at fp16-element offset elem_off
__device__ __forceinline__ uint32_t grid_load_u32(const GridRef& g, uint32_t elem_off) {
  return __builtin_amdgcn_raw_buffer_load_b32(g.r, (int)(elem_off * 2u), 0, 0);
}

// ------------------------------------------------------------------ a word the HOST reads while the stream keeps running
// The decoder kernels are 9-33 KB of straight-line code and run for ~0.1 ms; between two launches of one of them hundreds
// of MB stream through L2, so every launch starts with its code in HBM and the first pass through it is a chain of
// instruction-cache misses to memory.  One 64-byte line per thread, read as DATA from the kernel's own program counter
// on, puts the next ``bytes`` of code into this XCD's L2 in one round trip; the instruction fetches behind it then miss
// to L2, not to HBM.  (``bytes`` is at most the distance to the end of the code object's text: the callers are not the
// last functions of field.hip's ~1 MB text.)
__device__ __forceinline__ void nsim_prefetch_own_code(int bytes, char* smem) {
  if (bytes <= 0) return;
  const char* pc = reinterpret_cast<const char*>(__builtin_amdgcn_s_getpc());
  uint32_t acc = 0u;
  for (int i = (int)threadIdx.x * 64; i < bytes; i += (int)blockDim.x * 64)
    acc ^= *reinterpret_cast<const volatile uint32_t*>(pc + i);
  if (acc == 0x9e3779b9u && bytes < 0) reinterpret_cast<volatile uint32_t*>(smem)[0] = acc;      // (keeps the loads)
}

// (host-mapped pinned memory): system-scope store
__device__ __forceinline__ void nsim_store_system(int64_t* p, int64_t v, bool release) {
  if (release) __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  else __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
