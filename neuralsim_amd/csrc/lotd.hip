// lotd.hip -- standalone LoTD encoding kernels (forward, forward + dy/dx, backward to the grid including
// the dy/dx path) for gfx950.  These are the API-level equivalents of
// nr3d_lib.models.grid_encodings.lotd's forward / forward_dydx / backward entry points
// (code_single/tools/inspect_rendering.py:468-474; docs/exps/exp_permuto_3d_modulated.py:63-76); the
// training hot path uses the fused kernels of field.hip instead, which share lotd_dev.h.
//
// One thread per (point, level), level fastest: the 16 lanes of a point write one coalesced 128-B
// feature row; a level's table (<= 2 MiB) stays L2-resident.  Gather cost: 8 corners x 4 B per level
// = 512 B per point.
#include "lotd_dev.h"

__global__ void __launch_bounds__(256) k_lotd_fwd(const float* __restrict__ x, const f16* __restrict__ grid,
                                                   LotdDev m, int64_t S, float* __restrict__ out,
                                                   float* __restrict__ dydx) {
  const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int L = m.num_levels;
  if (tid >= S * L) return;
  const int64_t s = tid / L;
  const int l = (int)(tid % L);
  const float xx[3] = {x[3 * s], x[3 * s + 1], x[3 * s + 2]};
  const LotdRes R = m.res[l];
  const LotdCell c = lotd_cell(xx, R, m);
  float f0 = 0.f, f1 = 0.f;
  float j0[3] = {0.f, 0.f, 0.f}, j1[3] = {0.f, 0.f, 0.f};
  const int pm = lotd_slot_mask(c);      // the library's one vertex order (lotd_dev.h)
  if (l < m.n_active)
#pragma unroll
  for (int slot = 0; slot < 8; ++slot) {
    const int corner = slot ^ pm;
    float w, dw[3];
    lotd_corner_w(c, corner, w, dw);
    const uint32_t idx = lotd_index(c.c0[0] + (corner & 1), c.c0[1] + ((corner >> 1) & 1),
                                    c.c0[2] + ((corner >> 2) & 1), R, m.type[l], m.size[l]);
    float g0, g1;
    lotd_load2(grid, m.offset[l], idx, g0, g1);
    f0 = f0 + w * g0;
    f1 = f1 + w * g1;
    if (dydx) {
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        j0[a] = j0[a] + dw[a] * g0;
        j1[a] = j1[a] + dw[a] * g1;
      }
    }
  }
  const int64_t o = s * (2 * L) + 2 * l;
  out[o] = f0;
  out[o + 1] = f1;
  if (dydx) {
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      dydx[o * 3 + a] = j0[a] * c.dscale[a];
      dydx[(o + 1) * 3 + a] = j1[a] * c.dscale[a];
    }
  }
}

__global__ void __launch_bounds__(256) k_lotd_bwd(const float* __restrict__ x, const float* __restrict__ dout,
                                                   const float* __restrict__ ddydx, LotdDev m, int64_t S,
                                                   float* __restrict__ dgrid) {
  const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int L = m.num_levels;
  if (tid >= S * L) return;
  const int64_t s = tid / L;
  const int l = (int)(tid % L);
  if (l >= m.n_active) return;   // masked level: no gradient
  const float xx[3] = {x[3 * s], x[3 * s + 1], x[3 * s + 2]};
  const LotdRes R = m.res[l];
  const LotdCell c = lotd_cell(xx, R, m);
  const int64_t o = s * (2 * L) + 2 * l;
  const float d0 = dout ? dout[o] : 0.f, d1 = dout ? dout[o + 1] : 0.f;
  float q0[3] = {0.f, 0.f, 0.f}, q1[3] = {0.f, 0.f, 0.f};
  if (ddydx) {
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      q0[a] = ddydx[o * 3 + a] * c.dscale[a];
      q1[a] = ddydx[(o + 1) * 3 + a] * c.dscale[a];
    }
  }
#pragma unroll
  for (int corner = 0; corner < 8; ++corner) {
    float w, dw[3];
    lotd_corner_w(c, corner, w, dw);
    const uint32_t idx = lotd_index(c.c0[0] + (corner & 1), c.c0[1] + ((corner >> 1) & 1),
                                    c.c0[2] + ((corner >> 2) & 1), R, m.type[l], m.size[l]);
    const float v0 = w * d0 + (dw[0] * q0[0] + dw[1] * q0[1] + dw[2] * q0[2]);
    const float v1 = w * d1 + (dw[0] * q1[0] + dw[1] * q1[1] + dw[2] * q1[2]);
    float* dst = dgrid + m.offset[l] + 2 * (int64_t)idx;
    if (v0 != 0.f) atomicAdd(dst, v0);
    if (v1 != 0.f) atomicAdd(dst + 1, v1);
  }
}

extern "C" {

int nsim_lotd_fwd(const float* x, const void* grid_f16, const NsimLotdMeta* meta, int64_t S, float* out,
                  float* dydx, void* stream) {
  const int rc = lotd_meta_check(meta);
  if (rc) return rc;
  if (S <= 0) return 0;
  hipLaunchKernelGGL(k_lotd_fwd, dim3(nsim_blocks(S * meta->num_levels, 256)), dim3(256), 0, (hipStream_t)stream, x,
                     (const f16*)grid_f16, lotd_dev(meta), S, out, dydx);
  NSIM_CHECK_LAUNCH();
  return 0;
}

int nsim_lotd_bwd(const float* x, const float* dL_dout, const float* dL_ddydx, const NsimLotdMeta* meta, int64_t S,
                  float* dgrid, void* stream) {
  const int rc = lotd_meta_check(meta);
  if (rc) return rc;
  if (S <= 0) return 0;
  hipLaunchKernelGGL(k_lotd_bwd, dim3(nsim_blocks(S * meta->num_levels, 256)), dim3(256), 0, (hipStream_t)stream, x,
                     dL_dout, dL_ddydx, lotd_dev(meta), S, dgrid);
  NSIM_CHECK_LAUNCH();
  return 0;
}

}  // extern "C"
