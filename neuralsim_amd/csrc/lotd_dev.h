// lotd_dev.h -- device-side LoTD (multi-resolution Dense/Hash grid) addressing shared by lotd.hip and
// field.hip.  Conventions are those of oracle/lotd.py (the executable spec standing in for the absent
// nr3d_lib.models.grid_encodings.lotd; config: lotd_neus.dtu.230814.yaml:96-111):
//   u = x/2 + 0.5 ; pos_a = u_a * (R_a-1) ; c0 = clamp(floor(pos), 0, R-2) ; w = pos - c0   (R_a: vertices on axis a;
//   cubic levels have R_x = R_y = R_z, ``lotd_use_cuboid`` levels follow the aspect of the AABB)
//   Dense index = cx + Rx*(cy + Ry*cz) ; Hash index = (cx ^ cy*2654435761 ^ cz*805459861) mod T (uint32)
//   levels l >= n_active are masked (``anneal_cfg{type: hardmask}``): feature 0, no gradient, no memory access.
//   params: flat fp16, level l at [offset_l, offset_l + size_l*2), feature index fastest.
#pragma once
#include "nsim_common.h"

struct LotdRes {
  int r[3];
  __host__ __device__ int max() const { return r[0] > r[1] ? (r[0] > r[2] ? r[0] : r[2]) : (r[1] > r[2] ? r[1] : r[2]); }
};

struct LotdDev {
  int num_levels;
  int n_active;                        // levels >= n_active are masked (hardmask annealing)
  LotdRes res[NSIM_MAX_LEVELS];
  int type[NSIM_MAX_LEVELS];
  uint32_t size[NSIM_MAX_LEVELS];
  int64_t offset[NSIM_MAX_LEVELS];
  float xs[3], xb[3];                  // u_a = x_a * xs[a] + xb[a]  (AABB -> [0,1] per axis; (0.5, 0.5) for [-1,1]^3)
};

static inline LotdDev lotd_dev(const NsimLotdMeta* m) {
  LotdDev d;
  d.num_levels = m->num_levels;
  d.n_active = (m->n_active_levels > 0 && m->n_active_levels < m->num_levels) ? m->n_active_levels : m->num_levels;
  for (int l = 0; l < NSIM_MAX_LEVELS; ++l) {
    for (int a = 0; a < 3; ++a) d.res[l].r[a] = l < m->num_levels ? m->res[l][a] : 2;
    d.type[l] = l < m->num_levels ? m->type[l] : 0;
    d.size[l] = l < m->num_levels ? m->size[l] : 8;
    d.offset[l] = l < m->num_levels ? m->offset[l] : 0;
  }
  const bool unit = m->x_scale[0] == 0.f && m->x_scale[1] == 0.f && m->x_scale[2] == 0.f;
  for (int a = 0; a < 3; ++a) {
    d.xs[a] = unit ? 0.5f : m->x_scale[a];
    d.xb[a] = unit ? 0.5f : m->x_shift[a];
  }
  return d;
}

static inline int lotd_meta_check(const NsimLotdMeta* m) {
  if (!m) return 10;
  if (m->n_feats != 2) return 11;
  if (m->num_levels < 1 || m->num_levels > NSIM_MAX_LEVELS) return 12;
  for (int l = 0; l < m->num_levels; ++l) {
    if (m->res[l][0] < 2 || m->res[l][1] < 2 || m->res[l][2] < 2) return 13;
    if (m->type[l] == NSIM_LOTD_DENSE) {
      if ((uint64_t)m->res[l][0] * m->res[l][1] * m->res[l][2] != (uint64_t)m->size[l]) return 14;
    } else if (m->type[l] == NSIM_LOTD_HASH) {
      if (m->size[l] == 0 || (m->size[l] & (m->size[l] - 1)) != 0) return 17;  // hash tables: power of two
    } else {
      return 15;
    }
    if (m->offset[l] & 1) return 16;
  }
  return 0;
}

struct LotdCell {
  int c0[3];
  float w[3];
  float dscale[3];  // d pos_a / d x_a = x_scale_a * (R_a - 1)
};

__device__ __forceinline__ LotdCell lotd_cell(const float x[3], const LotdRes& R, const LotdDev& L) {
  LotdCell c;
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const float rm1 = (float)(R.r[a] - 1);
    const float u = x[a] * L.xs[a] + L.xb[a];
    const float pos = u * rm1;
    float f = floorf(pos);
    f = fminf(fmaxf(f, 0.f), (float)(R.r[a] - 2));
    c.c0[a] = (int)f;
    c.w[a] = pos - f;
    c.dscale[a] = L.xs[a] * rm1;
  }
  return c;
}

__device__ __forceinline__ uint32_t lotd_index(int cx, int cy, int cz, const LotdRes& R, int type, uint32_t T) {
  if (type == NSIM_LOTD_DENSE)
    return (uint32_t)cx + (uint32_t)R.r[0] * ((uint32_t)cy + (uint32_t)R.r[1] * (uint32_t)cz);
  const uint32_t h = (uint32_t)cx ^ ((uint32_t)cy * 2654435761u) ^ ((uint32_t)cz * 805459861u);
  return h & (T - 1u);  // T is a power of two (checked on the host)
}

// Vertex enumeration of a gather (round 6): slot k = 0..7 reads the vertex whose COORDINATES have the parities of k's bits, i.e.
// corner k ^ (c0 & 1 per axis).  The eight vertices of a cell have the eight parity combinations once each, so a vertex two
// neighbouring cells share is the same slot in both: the 64 consecutive samples of one load instruction name fewer distinct
// vertices, and the level-major gathers run 9-10 % faster (MI355X: 0.0673 -> 0.0610 ms per sampling launch, the street
// forward 1.009 -> 0.962 ms; profiles/round6_gather_parity_ab.txt).  Only the ORDER of the eight-term sums changes; every
// gather of the library uses the same order (the fused and the level-major forms stay bit-identical).
// -DNSIM_GATHER_PARITY=0: slots by corner offset, as rounds 1-5.
// (the switch itself lives in nsim_common.h: the 4-D pyramid of nerf_field.hip follows it too)
__device__ __forceinline__ int lotd_slot_mask(const LotdCell& c) {
  return NSIM_GATHER_PARITY ? ((c.c0[0] & 1) | ((c.c0[1] & 1) << 1) | ((c.c0[2] & 1) << 2)) : 0;
}

// The same enumeration with the parity folded into the OPERANDS instead of the corner index: per axis the coordinate, the weight
// and the derivative sign of the vertex with parity bit 0 / 1 (two selects per axis), so the eight slots index them with
// compile-time bits -- the eight runtime corner indices of ``corner = slot ^ mask`` cost six selects each, and kept the compiler from
// sharing the per-axis products and hash terms between slots.  lotd_slot_w(S, k, ..) returns the values of
// lotd_corner_w(c, k ^ lotd_slot_mask(c), ..) bit for bit: (+-1 a) b == +-(a b) exactly.
struct LotdSlots {
  int v[3][2];      // coordinate of the vertex with parity bit k on axis a
  float w[3][2];    // its interpolation weight along the axis
  float s[3];       // d w[a][0] / d pos_a (= -d w[a][1] / d pos_a): +-1
};
__device__ __forceinline__ LotdSlots lotd_slots(const LotdCell& c) {
  LotdSlots S;
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const int p = NSIM_GATHER_PARITY ? (c.c0[a] & 1) : 0;
    S.v[a][0] = c.c0[a] + p;
    S.v[a][1] = c.c0[a] + (p ^ 1);
    const float w1 = c.w[a], w0 = 1.0f - c.w[a];
    S.w[a][0] = p ? w1 : w0;
    S.w[a][1] = p ? w0 : w1;
    S.s[a] = p ? 1.0f : -1.0f;
  }
  return S;
}
__device__ __forceinline__ void lotd_slot_w(const LotdSlots& S, int k, float& w, float dw[3]) {
  const int kx = k & 1, ky = (k >> 1) & 1, kz = (k >> 2) & 1;
  const float wx = S.w[0][kx], wy = S.w[1][ky], wz = S.w[2][kz];
  const float wxy = wx * wy;
  w = wxy * wz;
  dw[0] = (kx ? -S.s[0] : S.s[0]) * (wy * wz);
  dw[1] = (ky ? -S.s[1] : S.s[1]) * (wx * wz);
  dw[2] = (kz ? -S.s[2] : S.s[2]) * wxy;
}
__device__ __forceinline__ uint32_t lotd_slot_index(const LotdSlots& S, int k, const LotdRes& R, int type, uint32_t T) {
  return lotd_index(S.v[0][k & 1], S.v[1][(k >> 1) & 1], S.v[2][(k >> 2) & 1], R, type, T);
}

// trilinear weight of corner (dx,dy,dz) and its derivative w.r.t. the three cell coordinates
__device__ __forceinline__ void lotd_corner_w(const LotdCell& c, int corner, float& w, float dw[3]) {
  const int dx = corner & 1, dy = (corner >> 1) & 1, dz = (corner >> 2) & 1;
  const float wx = dx ? c.w[0] : 1.0f - c.w[0];
  const float wy = dy ? c.w[1] : 1.0f - c.w[1];
  const float wz = dz ? c.w[2] : 1.0f - c.w[2];
  w = wx * wy * wz;
  dw[0] = (dx ? 1.0f : -1.0f) * wy * wz;
  dw[1] = (dy ? 1.0f : -1.0f) * wx * wz;
  dw[2] = (dz ? 1.0f : -1.0f) * wx * wy;
}

__device__ __forceinline__ void lotd_load2(const f16* grid, int64_t off, uint32_t idx, float& f0, float& f1) {
  const uint32_t raw = *reinterpret_cast<const uint32_t*>(grid + off + 2 * (int64_t)idx);
  union {
    uint32_t u;
    f16 h[2];
  } cv;
  cv.u = raw;
  f0 = (float)cv.h[0];
  f1 = (float)cv.h[1];
}

// (GridRef / grid_ref / grid_load_u32: the table behind one buffer resource, nsim_prims.h)

// elem_off = level offset + 2 * vertex index (in fp16 elements)
__device__ __forceinline__ void lotd_load2(const GridRef& g, uint32_t elem_off, float& f0, float& f1) {
  union {
    uint32_t u;
    f16 h[2];
  } cv;
  cv.u = grid_load_u32(g, elem_off);
  f0 = (float)cv.h[0];
  f1 = (float)cv.h[1];
}
