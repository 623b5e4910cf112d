// nsim_common.h -- shared device helpers for the gfx950 (MI355X / CDNA4) kernels.
//
// All kernels are written for 64-lane wavefronts.  Cross-lane primitives and the
// MFMA instructions are reached only through the thin wrappers below, so that the
// test-only host emulator (tests/emu/hip_emu.h, -DNSIM_HOST_EMU) can stand in for
// the hardware when the kernel *logic* is checked on a CPU-only machine.  The
// product build never defines NSIM_HOST_EMU.
#pragma once
#include <stdint.h>
#include <string.h>
#include <math.h>

#ifdef NSIM_HOST_EMU
#include "hip_emu.h"
#define NSIM_DYN_SMEM(name) char* name = emu::st().dyn_smem
#else
#include <hip/hip_runtime.h>
#define NSIM_DYN_SMEM(name) extern __shared__ __attribute__((aligned(16))) char name[]
#endif

#include "../../include/nsim.h"

#define NSIM_WAVE 64

typedef _Float16 f16;
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// ------------------------------------------------------------------ cross-lane
__device__ __forceinline__ int nsim_lane() {
#ifdef NSIM_HOST_EMU
  return emu::lane_id();
#else
  return (int)(threadIdx.x & 63);
#endif
}

template <class T>
__device__ __forceinline__ T wave_shfl(T v, int src) {
#ifdef NSIM_HOST_EMU
  return emu::shfl(v, src);
#else
  return __shfl(v, src, 64);
#endif
}

template <class T>
__device__ __forceinline__ T wave_shfl_xor(T v, int mask) {
  return wave_shfl(v, nsim_lane() ^ mask);
}

__device__ __forceinline__ unsigned long long wave_ballot(int pred) {
#ifdef NSIM_HOST_EMU
  return emu::ballot(pred);
#else
  return __ballot(pred);
#endif
}

// broadcast the value of lane I of every aligned group of 4 lanes (DPP quad_perm on the device)
template <int I, class T>
__device__ __forceinline__ T quad_bcast(T v) {
#ifdef NSIM_HOST_EMU
  return emu::shfl(v, (emu::lane_id() & ~3) + I);
#else
  static_assert(sizeof(T) == 4, "quad_bcast: 32-bit types only");
  int iv;
  __builtin_memcpy(&iv, &v, 4);
  iv = __builtin_amdgcn_update_dpp(0, iv, I | (I << 2) | (I << 4) | (I << 6), 0xf, 0xf, true);
  T r;
  __builtin_memcpy(&r, &iv, 4);
  return r;
#endif
}

// inclusive scans across the 64 lanes.  Device, float: DPP row_shr 1/2/4/8 scans every row of 16, row_bcast15 /
// row_bcast31 carry the row totals forward (lanes without a source keep the identity) -- six VALU steps, no LDS
// crossbar.  Other types / the host emulator: Hillis-Steele over shuffles.
#ifndef NSIM_HOST_EMU
#define NSIM_DPP_OLD_F32(oldv, x, ctrl, rmask)                                                              \
  __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, (float)(oldv)),             \
                                                        __builtin_bit_cast(int, (x)), (ctrl), (rmask), 0xf, false))
#endif

template <class T>
__device__ __forceinline__ T wave_incl_sum(T v) {
#ifndef NSIM_HOST_EMU
  if constexpr (__is_same(T, float)) {
    v += NSIM_DPP_OLD_F32(0.f, v, 0x111, 0xf);
    v += NSIM_DPP_OLD_F32(0.f, v, 0x112, 0xf);
    v += NSIM_DPP_OLD_F32(0.f, v, 0x114, 0xf);
    v += NSIM_DPP_OLD_F32(0.f, v, 0x118, 0xf);
    v += NSIM_DPP_OLD_F32(0.f, v, 0x142, 0xa);   // row_bcast15 into rows 1 and 3
    v += NSIM_DPP_OLD_F32(0.f, v, 0x143, 0xc);   // row_bcast31 into rows 2 and 3
    return v;
  }
#endif
  const int lane = nsim_lane();
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    T u = wave_shfl(v, lane - o);
    if (lane >= o) v += u;
  }
  return v;
}

template <class T>
__device__ __forceinline__ T wave_incl_prod(T v) {
#ifndef NSIM_HOST_EMU
  if constexpr (__is_same(T, float)) {
    v *= NSIM_DPP_OLD_F32(1.f, v, 0x111, 0xf);
    v *= NSIM_DPP_OLD_F32(1.f, v, 0x112, 0xf);
    v *= NSIM_DPP_OLD_F32(1.f, v, 0x114, 0xf);
    v *= NSIM_DPP_OLD_F32(1.f, v, 0x118, 0xf);
    v *= NSIM_DPP_OLD_F32(1.f, v, 0x142, 0xa);
    v *= NSIM_DPP_OLD_F32(1.f, v, 0x143, 0xc);
    return v;
  }
#endif
  const int lane = nsim_lane();
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    T u = wave_shfl(v, lane - o);
    if (lane >= o) v *= u;
  }
  return v;
}

// Wave-wide reductions.  On the device the float versions stay in the VALU: four DPP steps reduce every row of 16
// lanes (quad_perm xor 1 / xor 2, row_half_mirror, row_mirror), four v_readlane + scalar-side combine finish across
// rows -- ~10 issue slots, against six dependent ds_bpermute round trips (~100 cycles each) for the shuffle tree.
#ifndef NSIM_HOST_EMU
#define NSIM_DPP_F32(x, ctrl) \
  __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, (x)), (ctrl), 0xf, 0xf, true))
__device__ __forceinline__ float nsim_readlane_f32(float v, int l) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l));
}
#endif

template <class T>
__device__ __forceinline__ T wave_sum(T v) {
#ifndef NSIM_HOST_EMU
  if constexpr (__is_same(T, float)) {
    v += NSIM_DPP_F32(v, 0xB1);
    v += NSIM_DPP_F32(v, 0x4E);
    v += NSIM_DPP_F32(v, 0x141);
    v += NSIM_DPP_F32(v, 0x140);
    return (nsim_readlane_f32(v, 0) + nsim_readlane_f32(v, 16)) + (nsim_readlane_f32(v, 32) + nsim_readlane_f32(v, 48));
  }
#endif
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += wave_shfl_xor(v, o);
  return v;
}

template <class T>
__device__ __forceinline__ T wave_max(T v) {
#ifndef NSIM_HOST_EMU
  if constexpr (__is_same(T, float)) {
    v = fmaxf(v, NSIM_DPP_F32(v, 0xB1));
    v = fmaxf(v, NSIM_DPP_F32(v, 0x4E));
    v = fmaxf(v, NSIM_DPP_F32(v, 0x141));
    v = fmaxf(v, NSIM_DPP_F32(v, 0x140));
    return fmaxf(fmaxf(nsim_readlane_f32(v, 0), nsim_readlane_f32(v, 16)),
                 fmaxf(nsim_readlane_f32(v, 32), nsim_readlane_f32(v, 48)));
  }
#endif
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    T u = wave_shfl_xor(v, o);
    v = u > v ? u : v;
  }
  return v;
}

// native exp / log (v_exp_f32 / v_log_f32 based on the device; libm under the host emulator)
__device__ __forceinline__ float nsim_fast_exp(float x) {
#ifdef NSIM_HOST_EMU
  return expf(x);
#else
  return __expf(x);
#endif
}
__device__ __forceinline__ float nsim_fast_log(float x) {
#ifdef NSIM_HOST_EMU
  return logf(x);
#else
  return __logf(x);
#endif
}

// raw base-2 pipes: v_exp_f32 / v_log_f32 (the ranges used here never reach their denormal corner cases)
__device__ __forceinline__ float nsim_exp2(float x) {
#ifdef NSIM_HOST_EMU
  return exp2f(x);
#else
  return __builtin_amdgcn_exp2f(x);
#endif
}
__device__ __forceinline__ float nsim_log2(float x) {
#ifdef NSIM_HOST_EMU
  return log2f(x);
#else
  return __builtin_amdgcn_logf(x);
#endif
}

// ------------------------------------------------------------------------ direct global -> LDS copies (gfx950)
// global_load_lds_dwordx4: every lane names its own 16-byte global source, the destination is the wave-uniform LDS base +
// 16 * lane (1 KB per instruction), no VGPR is written.  The copy is asynchronous on the VM counter: wait vmcnt(0) before
// reading the image, and wait lgkmcnt(0) after the last ds_read of an image before overwriting it.
__device__ __forceinline__ void nsim_glds16(const void* gsrc, char* lds_wave_base) {
#ifdef NSIM_HOST_EMU
  memcpy(lds_wave_base + 16 * nsim_lane(), gsrc, 16);
#else
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
#endif
}
__device__ __forceinline__ void nsim_wait_vm0() {
#ifdef NSIM_HOST_EMU
  emu::wave_barrier();      // the emulator runs the lanes one after another: all of them have issued their copies
#else
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
}
__device__ __forceinline__ void nsim_wait_lgkm0() {
#ifdef NSIM_HOST_EMU
  emu::wave_barrier();
#else
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
}

// ------------------------------------------------------------------------ MFMA
// v_mfma_f32_32x32x16_f16: A lane l -> row (l&31), B lane l -> col (l&31),
// 8 K-slots per lane indexed by (l>>5, e); C/D: col = l&31,
// row = (r&3) + 8*(r>>2) + 4*(l>>5).  (CDNA4 guide, "Fragment layout".)
__device__ __forceinline__ f32x16 mfma_32x32x16_f16(f16x8 a, f16x8 b, f32x16 c) {
#ifdef NSIM_HOST_EMU
  f16 aa[8], bb[8];
  float cc[16], dd[16];
  for (int e = 0; e < 8; ++e) { aa[e] = a[e]; bb[e] = b[e]; }
  for (int r = 0; r < 16; ++r) cc[r] = c[r];
  emu::mfma32<f16, 8>(aa, bb, cc, dd);
  f32x16 d;
  for (int r = 0; r < 16; ++r) d[r] = dd[r];
  return d;
#else
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
#endif
}

// v_mfma_f32_32x32x16_bf16: same fragment layout and rate as the f16 form, operands with the f32 exponent range.
__device__ __forceinline__ f32x16 mfma_32x32x16_bf16(bf16x8 a, bf16x8 b, f32x16 c) {
#ifdef NSIM_HOST_EMU
  bf16 aa[8], bb[8];
  float cc[16], dd[16];
  for (int e = 0; e < 8; ++e) { aa[e] = a[e]; bb[e] = b[e]; }
  for (int r = 0; r < 16; ++r) cc[r] = c[r];
  emu::mfma32<bf16, 8>(aa, bb, cc, dd);
  f32x16 d;
  for (int r = 0; r < 16; ++r) d[r] = dd[r];
  return d;
#else
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
#endif
}

// v_mfma_f32_32x32x2_f32: exact-f32 matrix FMA, one K-slot per lane (k = l>>5).
__device__ __forceinline__ f32x16 mfma_32x32x2_f32(float a, float b, f32x16 c) {
#ifdef NSIM_HOST_EMU
  float cc[16], dd[16];
  for (int r = 0; r < 16; ++r) cc[r] = c[r];
  emu::mfma32<float, 1>(&a, &b, cc, dd);
  f32x16 d;
  for (int r = 0; r < 16; ++r) d[r] = dd[r];
  return d;
#else
  return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
#endif
}

// row of accumulator register r for lane-half hi within a 32-row MFMA tile
__device__ __forceinline__ constexpr int mfma_row(int r, int hi) {
  return (r & 3) + 8 * (r >> 2) + 4 * hi;
}

// a word the HOST reads while the stream keeps running (host-mapped pinned memory): system-scope store
__device__ __forceinline__ void nsim_store_system(int64_t* p, int64_t v, bool release) {
#ifdef NSIM_HOST_EMU
  __atomic_store_n(p, v, release ? __ATOMIC_RELEASE : __ATOMIC_RELAXED);
#else
  if (release) __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  else __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
#endif
}

// ------------------------------------------------------------------- launching
#define NSIM_CHECK_LAUNCH()                          \
  do {                                               \
    hipError_t e__ = hipGetLastError();              \
    if (e__ != hipSuccess) return 1000 + (int)e__;   \
  } while (0)

static inline unsigned nsim_blocks(int64_t n, int per_block, int64_t cap = 1 << 20) {
  int64_t b = (n + per_block - 1) / per_block;
  if (b < 1) b = 1;
  if (b > cap) b = cap;
  return (unsigned)b;
}
