// nsim_common.h -- shared device helpers for the gfx950 (MI355X / CDNA4) kernels.
//
// All kernels are written for 64-lane wavefronts.  Everything that names a gfx950 instruction or builtin lives in
// nsim_prims.h; this file holds the helpers written on top of those primitives.
#pragma once
// vertex order of every table gather: 1 = slots by vertex-coordinate parity (round 6, lotd_dev.h), 0 = by corner offset
#ifndef NSIM_GATHER_PARITY
#define NSIM_GATHER_PARITY 1
#endif
#include <stdint.h>
#include <string.h>
#include <math.h>

#include "../../include/nsim.h"

#define NSIM_WAVE 64

typedef _Float16 f16;
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// dh/dx planes of the with-grad query (nsim_field_fwd -> nsim_field_bwd_sdf; include/nsim.h): f16 in the fp16 field mode
// (192 instead of 384 bytes per point: the dominant plane traffic of the with-grad gather and of both decoders that read them),
// f32 in the f32 validation mode.  NSIM_J16=0 keeps f32 planes in both modes (A/B aid).
#ifndef NSIM_J16
#define NSIM_J16 1
#endif
template <int PREC>
struct JPlane {
  using T = float;
};
#if NSIM_J16
template <>
struct JPlane<0> {
  using T = f16;
};
#endif
#define NSIM_J_ELEM_BYTES(precision) ((NSIM_J16 && (precision) == 0) ? 2 : 4)

// The hardware primitives: nsim_lane, wave_shfl, wave_ballot, quad_bcast, the float scans / reductions on DPP, the
// in-wave ordering points, the transcendental pipes, LDS-DMA, MFMA, the buffer-resource table loads, the system-scope
// store -- neuralsim_amd/csrc/nsim_prims.h (gfx950).
#include <nsim_prims.h>

// ------------------------------------------------------------------ cross-lane, generic forms
template <class T>
__device__ __forceinline__ T wave_shfl_xor(T v, int mask) {
  return wave_shfl(v, nsim_lane() ^ mask);
}

// inclusive scans across the 64 lanes: Hillis-Steele over shuffles (float has a DPP form in nsim_prims.h, chosen by
// overload resolution)
template <class T>
__device__ __forceinline__ T wave_incl_sum(T v) {
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
  const int lane = nsim_lane();
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    T u = wave_shfl(v, lane - o);
    if (lane >= o) v *= u;
  }
  return v;
}

// wave-wide reductions: butterfly over shuffles (float: DPP form in nsim_prims.h)
template <class T>
__device__ __forceinline__ T wave_sum(T v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += wave_shfl_xor(v, o);
  return v;
}

template <class T>
__device__ __forceinline__ T wave_max(T v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    T u = wave_shfl_xor(v, o);
    v = u > v ? u : v;
  }
  return v;
}

// row of accumulator register r for lane-half hi within a 32-row MFMA tile
__device__ __forceinline__ constexpr int mfma_row(int r, int hi) {
  return (r & 3) + 8 * (r >> 2) + 4 * hi;
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
