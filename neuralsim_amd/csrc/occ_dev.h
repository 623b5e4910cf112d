// occ_dev.h -- occupancy-grid device helpers shared by the sampling kernels (sampling.hip) and the no-grad decoder
// (field.hip), which folds the SDFs it has just computed into the value grid (``update_from_samples_cfg``).
#pragma once
#include "nsim_common.h"

struct OccDev {
  float mn[3], mx[3], sc[3];
  int res[3];
};
static inline OccDev occ_dev(const NsimOccMeta* m) {
  OccDev o;
  for (int i = 0; i < 3; ++i) {
    o.mn[i] = m->aabb_min[i];
    o.mx[i] = m->aabb_max[i];
    o.sc[i] = m->scale[i];
    o.res[i] = m->res[i];
  }
  return o;
}

__device__ __forceinline__ bool occ_voxel(const OccDev& m, float px, float py, float pz, int64_t& flat) {
  const float gx = floorf((px - m.mn[0]) * m.sc[0]);
  const float gy = floorf((py - m.mn[1]) * m.sc[1]);
  const float gz = floorf((pz - m.mn[2]) * m.sc[2]);
  const bool inside = gx >= 0.f && gy >= 0.f && gz >= 0.f && gx < (float)m.res[0] && gy < (float)m.res[1] &&
                      gz < (float)m.res[2];
  flat = inside ? ((int64_t)gx + (int64_t)m.res[0] * ((int64_t)gy + (int64_t)m.res[1] * (int64_t)gz)) : 0;
  return inside;
}

// val[voxel(p)] = max(val[voxel(p)], 4 sig(s sdf)(1 - sig(s sdf))) for the lanes with ``ok`` (all 64 lanes call).
// Consecutive samples of a ray share voxels (64^3 grid: ~6 marching steps per voxel) and most values do not exceed what
// the grid already holds: same-address atomics are separate requests that serialise in L2, so (1) equal voxels of
// neighbouring lanes are max-reduced inside the wave and only the last lane of a run goes on, (2) it skips the atomic
// when a plain read already shows a value >= its own (the grid only grows between refreshes).  f(sdf) >= 0: the integer
// atomicMax on the bit pattern is exact.
__device__ __forceinline__ void occ_collect_wave(float* __restrict__ val, const OccDev& m, bool ok, float px, float py,
                                                 float pz, float sdf, float inv_s) {
  const int lane = nsim_lane();
  int64_t flat = -1 - lane;
  float v = 0.f;
  if (ok) {
    int64_t f;
    ok = occ_voxel(m, px, py, pz, f);
    if (ok) {
      flat = f;
      const float s = 1.0f / (1.0f + expf(-sdf * inv_s));
      v = 4.0f * s * (1.0f - s);
    }
  }
  const int64_t pk = wave_shfl(flat, lane - 1);
  const unsigned long long heads = wave_ballot(lane == 0 || pk != flat);
  const unsigned long long below = heads & ((2ull << lane) - 1ull);
  const int run_start = 63 - __builtin_clzll(below);
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const float o = wave_shfl(v, lane - d);
    if (lane - d >= run_start) v = fmaxf(v, o);
  }
  const bool last = lane == 63 || ((heads >> (lane + 1)) & 1ull);
  if (!ok || !last) return;
  if (val[flat] >= v) return;
  int iv;
  memcpy(&iv, &v, 4);
  atomicMax((int*)val + flat, iv);
}
