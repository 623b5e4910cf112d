// nsim_prims.h (TEST INFRASTRUCTURE) -- the emulator's implementation of the hardware primitives the kernels reach
// through <nsim_prims.h>.  tests/emu/build_emu.py puts this directory FIRST on the include path of the emulator build, so
// this file stands in for neuralsim_amd/csrc/nsim_prims.h there; the product build never sees it.
// Float scans / reductions are NOT declared here: the generic shuffle forms of nsim_common.h serve every type.
#pragma once
#include "hip_emu.h"

#define NSIM_DYN_SMEM(name) char* name = emu::st().dyn_smem

inline int nsim_lane() { return emu::lane_id(); }

template <class T>
inline T wave_shfl(T v, int src) {
  return emu::shfl(v, src);
}

inline unsigned long long wave_ballot(int pred) { return emu::ballot(pred); }

template <int I, class T>
inline T quad_bcast(T v) {
  return emu::shfl(v, (emu::lane_id() & ~3) + I);
}

// the emulator runs the lanes of a wave one after another: every ordering point of the hardware is a wave barrier
inline void wave_sync_lds() { emu::wave_barrier(); }
inline void nsim_wave_fence() { emu::wave_barrier(); }
inline int nsim_opaque_zero() { return 0; }

inline float nsim_fast_exp(float x) { return expf(x); }
inline float nsim_fast_log(float x) { return logf(x); }
inline float nsim_exp2(float x) { return exp2f(x); }
inline float nsim_log2(float x) { return log2f(x); }
inline float nsim_sin(float x) { return sinf(x); }
inline float nsim_cos(float x) { return cosf(x); }

inline void nsim_glds16(const void* gsrc, char* lds_wave_base) { memcpy(lds_wave_base + 16 * nsim_lane(), gsrc, 16); }
inline void nsim_wait_vm0() { emu::wave_barrier(); }       // all lanes have issued their copies
inline void nsim_wait_lgkm0() { emu::wave_barrier(); }

inline f32x16 mfma_32x32x16_f16(f16x8 a, f16x8 b, f32x16 c) {
  f16 aa[8], bb[8];
  float cc[16], dd[16];
  for (int e = 0; e < 8; ++e) { aa[e] = a[e]; bb[e] = b[e]; }
  for (int r = 0; r < 16; ++r) cc[r] = c[r];
  emu::mfma32<f16, 8>(aa, bb, cc, dd);
  f32x16 d;
  for (int r = 0; r < 16; ++r) d[r] = dd[r];
  return d;
}
inline f32x16 mfma_32x32x16_bf16(bf16x8 a, bf16x8 b, f32x16 c) {
  bf16 aa[8], bb[8];
  float cc[16], dd[16];
  for (int e = 0; e < 8; ++e) { aa[e] = a[e]; bb[e] = b[e]; }
  for (int r = 0; r < 16; ++r) cc[r] = c[r];
  emu::mfma32<bf16, 8>(aa, bb, cc, dd);
  f32x16 d;
  for (int r = 0; r < 16; ++r) d[r] = dd[r];
  return d;
}
inline f32x16 mfma_32x32x2_f32(float a, float b, f32x16 c) {
  float cc[16], dd[16];
  for (int r = 0; r < 16; ++r) cc[r] = c[r];
  emu::mfma32<float, 1>(&a, &b, cc, dd);
  f32x16 d;
  for (int r = 0; r < 16; ++r) d[r] = dd[r];
  return d;
}
inline void nsim_cvt2_bf16(float a, float b, bf16& lo, bf16& hi) {
  lo = (bf16)a;
  hi = (bf16)b;
}
inline float nsim_bf16x8_sum(bf16x8 v, float s) {
  for (int e = 0; e < 8; ++e) s += (float)v[e];
  return s;
}

struct GridRef {
  const f16* p;
};
inline GridRef grid_ref(const f16* grid) {
  GridRef g;
  g.p = grid;
  return g;
}
inline uint32_t grid_load_u32(const GridRef& g, uint32_t elem_off) {
  return *reinterpret_cast<const uint32_t*>(g.p + elem_off);
}

inline void nsim_store_system(int64_t* p, int64_t v, bool release) {
  __atomic_store_n(p, v, release ? __ATOMIC_RELEASE : __ATOMIC_RELAXED);
}

// (product: pulls the kernel's own code into L2 -- nothing to emulate)
inline void nsim_prefetch_own_code(int, char*) {}
