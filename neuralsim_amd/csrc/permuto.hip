// permuto.hip -- multi-resolution permutohedral-lattice hash encoding for gfx950: the encoding behind the
// reference's PermutoNeuSObj / GenerativePermutoConcat models (nr3d_lib.models.grid_encodings.permuto.PermutoEncoding;
// call sites: app/models/single/neus.py:64-76, docs/exps/exp_permuto_3d_modulated.py:52-60,
// code_multi/configs/exps/fg_neus=permuto/all_occ.240201.yaml:438-446).  The implementation lives in the absent nr3d_lib;
// this follows the published algorithm (Adams et al. 2010; Rosu & Behnke 2023), restated in oracle/permuto.py.
//
// One thread per (point, level), blockIdx.y = level: a level's table (T x f16x2, 2 MB at T = 2^19) is what the chip reads at
// a time.  A point touches d + 1 vertices per level (4 in 3-D, against the 8 corners of a trilinear level): d + 1 random
// 4-byte loads forward, 2 (d + 1) f32 atomics backward.
//
// Two front ends share the lattice code:
//  * standalone encoding (nsim_permuto_fwd / _bwd): x [S,d] -> features [S, L F] (+ d features / d x over all d inputs),
//    point-major, for callers that decode elsewhere;
//  * the NeuS field (nsim_permuto_gather / _scatter): positions from rays (o + t d) or points, optional per-ray condition
//    z [R, d-3] (GenerativePermutoConcat: the latent is concatenated to the position), LEVEL-MAJOR planes in the layout of
//    field.hip's level-major LoTD gather -- feature planes [NL][S] (f16x2 | f32x2) for the no-grad decoder, h [NL][PS][2] and
//    dh/dx [NL][PS][2][3] (spatial derivative only) for the with-grad decoders -- so every decoder kernel of field.hip runs
//    unchanged on a permutohedral model; the scatter takes the same dL/dh and g = dsdf/dh planes and the total dL/dnablas
//    (second-order term: the weights are piecewise linear in x, d w_v / d x is constant inside a simplex).
#include "nsim_common.h"

#define PERMUTO_MAX_DIM 8
#define PERMUTO_SDF_H_SCALE 1024.0f      // = field.hip SDF_H_SCALE (fp16 feature planes are pre-scaled)

struct PermutoDev {
  int in_dim, num_levels;
  uint32_t T;
  float scale[NSIM_MAX_LEVELS][PERMUTO_MAX_DIM];
  float shift[NSIM_MAX_LEVELS][PERMUTO_MAX_DIM];
};

static PermutoDev permuto_dev(const NsimPermutoMeta* m) {
  PermutoDev d;
  d.in_dim = m->in_dim;
  d.num_levels = m->num_levels;
  d.T = m->hashmap_size;
  for (int l = 0; l < NSIM_MAX_LEVELS; ++l)
    for (int i = 0; i < PERMUTO_MAX_DIM; ++i) {
      d.scale[l][i] = m->scale[l][i];
      d.shift[l][i] = m->shift[l][i];
    }
  return d;
}

static int permuto_meta_check(const NsimPermutoMeta* m) {
  if (!m) return 40;
  if (m->in_dim < 2 || m->in_dim > PERMUTO_MAX_DIM) return 41;
  if (m->num_levels < 1 || m->num_levels > NSIM_MAX_LEVELS) return 42;
  if (m->n_feats != 2) return 43;
  if (m->hashmap_size == 0 || (m->hashmap_size & (m->hashmap_size - 1)) != 0) return 44;
  return 0;
}

__device__ __forceinline__ double diff_of(double e, int rem0) { return e - (double)rem0; }

// The enclosing simplex of one point on one level.
template <int D>
struct Simplex {
  int rem0[D + 1];
  int rank[D + 1];
  float bary[D + 1];
};

// elevated E_0 = sum_j cf_j, E_i = sum_{j > i} cf_j - i cf_i (1-based j; cf_j = (x_{j-1} + shift) * scale), nearest
// remainder-0 point, ranks, barycentric weights.  WD: also dB[r][c] = d bary[r] / d x_{C0 + c} for NS inputs from C0 on
// (C0 = 0: the spatial inputs; C0 = 3: the condition's, for d L / d z).
template <int D, int NS, bool WD, int C0 = 0>
__device__ __forceinline__ void permuto_simplex(const float (&x)[D], const float* scale, const float* shift, Simplex<D>& sp,
                                                float (&dB)[D + 1][NS]) {
  // The elevation runs in f64 (full-rate VALU on CDNA): with the per-level random shifts (up to 10) the finest levels work
  // at |E| ~ 3e4, where an f32 ulp is 2e-3 lattice units -- in f32 the weights carry ~1e-3 of rounding noise, the SDF along
  // a ray becomes a (slightly) noisy function, and the up-sampler (inv_s up to 1024) amplifies that into different sample
  // sets for any two implementations that do not round identically (measured at the BASELINE size: 56 of 2038 rays).
  double cf[D], E[D + 1];
#pragma unroll
  for (int i = 0; i < D; ++i) cf[i] = ((double)x[i] + (double)shift[i]) * (double)scale[i];
  double sm = 0.0;
#pragma unroll
  for (int i = D; i >= 1; --i) {
    E[i] = sm - (double)i * cf[i - 1];
    sm = sm + cf[i - 1];
  }
  E[0] = sm;
  int ssum = 0;
  double diff[D + 1];
#pragma unroll
  for (int i = 0; i <= D; ++i) {
    const double v = E[i] * (1.0 / (double)(D + 1));
    const double up = ceil(v) * (double)(D + 1), down = floor(v) * (double)(D + 1);
    const double r = (up - E[i] < E[i] - down) ? up : down;
    sp.rem0[i] = (int)r;
    diff[i] = E[i] - r;
    sp.rank[i] = 0;
  }
#pragma unroll
  for (int i = 0; i <= D; ++i) ssum += sp.rem0[i];
  ssum /= D + 1;      // exact: every rem0 is a multiple of D + 1
#pragma unroll
  for (int i = 0; i < D; ++i)
#pragma unroll
    for (int j = i + 1; j <= D; ++j) {
      if (diff[i] < diff[j]) sp.rank[i]++;
      else sp.rank[j]++;
    }
  if (ssum > 0) {
#pragma unroll
    for (int i = 0; i <= D; ++i) {
      if (sp.rank[i] >= D + 1 - ssum) {
        sp.rem0[i] -= D + 1;
        sp.rank[i] += ssum - (D + 1);
      } else {
        sp.rank[i] += ssum;
      }
    }
  } else if (ssum < 0) {
#pragma unroll
    for (int i = 0; i <= D; ++i) {
      if (sp.rank[i] < -ssum) {
        sp.rem0[i] += D + 1;
        sp.rank[i] += (D + 1) + ssum;
      } else {
        sp.rank[i] += ssum;
      }
    }
  }
  float b[D + 2];
#pragma unroll
  for (int k = 0; k < D + 2; ++k) b[k] = 0.f;
  if constexpr (WD) {
#pragma unroll
    for (int r = 0; r <= D; ++r)
#pragma unroll
      for (int c = 0; c < NS; ++c) dB[r][c] = 0.f;
  }
#pragma unroll
  for (int i = 0; i <= D; ++i) {
    const float delta = (float)(diff_of(E[i], sp.rem0[i]) * (1.0 / (double)(D + 1)));
    const int kp = D - sp.rank[i], km = D + 1 - sp.rank[i];
#pragma unroll
    for (int k = 0; k < D + 2; ++k) {      // (select form: no dynamically indexed registers)
      b[k] = b[k] + (k == kp ? delta : 0.f) - (k == km ? delta : 0.f);
    }
    if constexpr (WD) {
      const int rm = km == D + 1 ? 0 : km;                   // the wrap-around slot folds into vertex 0
#pragma unroll
      for (int c = 0; c < NS; ++c) {
        // d E_i / d x_c  (c 0-based; j = c + 1):  E_0: scale_c;  E_i: [j > i] scale_c - [i == j] i scale_c
        const int jj = C0 + c + 1;
        const float sc_ = scale[C0 + c];
        const float dE = (i == 0 ? sc_ : ((jj > i ? sc_ : 0.f) - (i == jj ? (float)i * sc_ : 0.f))) *
                         (1.0f / (float)(D + 1));
#pragma unroll
        for (int r = 0; r <= D; ++r) dB[r][c] = dB[r][c] + (r == kp ? dE : 0.f) - (r == rm ? dE : 0.f);
      }
    }
  }
  sp.bary[0] = b[0] + 1.0f + b[D + 1];
#pragma unroll
  for (int r = 1; r <= D; ++r) sp.bary[r] = b[r];
}

template <int D>
__device__ __forceinline__ uint32_t permuto_vertex(const Simplex<D>& sp, int remainder, uint32_t T) {
  uint32_t k = 0u;
#pragma unroll
  for (int i = 0; i < D; ++i) {
    int key = sp.rem0[i] + remainder;
    if (sp.rank[i] > D - remainder) key -= D + 1;
    k = (k + (uint32_t)key) * 2531011u;
  }
  return k & (T - 1u);
}

__device__ __forceinline__ void permuto_load2(const f16* grid, int64_t base, uint32_t idx, float& g0, float& g1) {
  const uint32_t raw = *reinterpret_cast<const uint32_t*>(grid + base + 2 * (int64_t)idx);
  union {
    uint32_t u;
    f16 h[2];
  } cv;
  cv.u = raw;
  g0 = (float)cv.h[0];
  g1 = (float)cv.h[1];
}

struct PermutoArgs {
  PermutoDev pm;
  const f16* grid;
  const float *x, *rays_o, *rays_d, *t, *z;      // x [S,D] (standalone) / [S,3] (field), or rays + t + ridx; z [R, D-3] or NULL
  const int64_t* ridx;
  int64_t S, PS;
  const int64_t* S_dev;
  int64_t S_add;
  // outputs / inputs of the four modes
  float *out, *dydx;                               // standalone forward
  void* feat_pl;                                   // field, no-grad: [NL][S] (f16x2 scaled | f32x2)
  int feat_f32;
  float* h_pl;                                     // field, with-grad
  void* J_pl;                                      // ... dh/dx planes: f16 (feat_f32 == 0 and NSIM_J16) | f32
  const float *dL_dout;                            // standalone backward [S, L F]
  const float *dh_pl, *g_pl, *gn;                  // field backward
  float* dgrid;
};

// input point of sample s: D coordinates (field mode: 3 spatial from x or the ray, D - 3 from the ray's condition)
template <int D, bool FIELD>
__device__ __forceinline__ void permuto_point(const PermutoArgs& a, int64_t s, float (&x)[D]) {
  if constexpr (!FIELD) {
#pragma unroll
    for (int i = 0; i < D; ++i) x[i] = a.x[(int64_t)D * s + i];
  } else {
    int64_t ray = 0;
    if (a.x) {
#pragma unroll
      for (int i = 0; i < 3; ++i) x[i] = a.x[3 * s + i];
      if (a.ridx) ray = a.ridx[s];
    } else {
      ray = a.ridx[s];
      const float tt = a.t[s];
#pragma unroll
      for (int i = 0; i < 3; ++i) x[i] = a.rays_o[3 * ray + i] + tt * a.rays_d[3 * ray + i];
    }
#pragma unroll
    for (int i = 3; i < D; ++i) x[i] = a.z ? a.z[(int64_t)(D - 3) * ray + (i - 3)] : 0.f;
  }
}

// MODE 0: standalone forward; 1: field feature planes (no-grad); 2: field h + dh/dx planes
template <int D, int MODE>
__global__ void __launch_bounds__(256) k_permuto_fwd(PermutoArgs a) {
  const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int l = blockIdx.y;
  int64_t Sv = a.S;
  if (a.S_dev) {
    const int64_t sd = a.S_dev[0] + a.S_add;
    Sv = sd <= Sv ? sd : 0;
  }
  if (s >= Sv) return;
  if (MODE != 0 && l >= a.pm.num_levels) {
    // plane levels past the pyramid (pyramids of fewer than 16 levels): the 16-level decoder kernels read all 16 planes
    // (zero weight columns) -- keep them finite, as field.hip's level-major gather does
    if constexpr (MODE == 1) {
      const int64_t e = (int64_t)l * a.PS + s;      // feature planes [NL][P]
      if (a.feat_f32) {
        reinterpret_cast<float*>(a.feat_pl)[2 * e] = 0.f;
        reinterpret_cast<float*>(a.feat_pl)[2 * e + 1] = 0.f;
      } else {
        reinterpret_cast<uint32_t*>(a.feat_pl)[e] = 0u;
      }
    } else if constexpr (MODE == 2) {
      const int64_t ep = (int64_t)l * a.PS + s;
      a.h_pl[ep * 2] = a.h_pl[ep * 2 + 1] = 0.f;
#pragma unroll
      for (int c = 0; c < 6; ++c) {
        if (NSIM_J16 && !a.feat_f32) reinterpret_cast<f16*>(a.J_pl)[ep * 6 + c] = (f16)0.f;
        else reinterpret_cast<float*>(a.J_pl)[ep * 6 + c] = 0.f;
      }
    }
    return;
  }
  constexpr int NS = MODE == 0 ? D : 3;
  float x[D];
  permuto_point<D, MODE != 0>(a, s, x);
  Simplex<D> sp;
  float dB[D + 1][NS];
  const bool want_d = MODE == 2 || (MODE == 0 && a.dydx != nullptr);      // (wave-uniform)
  if constexpr (MODE == 1) permuto_simplex<D, NS, false>(x, a.pm.scale[l], a.pm.shift[l], sp, dB);
  else if constexpr (MODE == 2) permuto_simplex<D, NS, true>(x, a.pm.scale[l], a.pm.shift[l], sp, dB);
  else if (want_d) permuto_simplex<D, NS, true>(x, a.pm.scale[l], a.pm.shift[l], sp, dB);
  else permuto_simplex<D, NS, false>(x, a.pm.scale[l], a.pm.shift[l], sp, dB);
  const int64_t base = (int64_t)l * a.pm.T * 2;
  float f0 = 0.f, f1 = 0.f, j0[NS], j1[NS];
#pragma unroll
  for (int c = 0; c < NS; ++c) j0[c] = j1[c] = 0.f;
#pragma unroll
  for (int r = 0; r <= D; ++r) {
    float g0, g1;
    permuto_load2(a.grid, base, permuto_vertex<D>(sp, r, a.pm.T), g0, g1);
    f0 = f0 + sp.bary[r] * g0;
    f1 = f1 + sp.bary[r] * g1;
    if (want_d) {
#pragma unroll
      for (int c = 0; c < NS; ++c) {
        j0[c] = j0[c] + dB[r][c] * g0;
        j1[c] = j1[c] + dB[r][c] * g1;
      }
    }
  }
  if constexpr (MODE == 0) {
    const int64_t o = s * (2 * a.pm.num_levels) + 2 * l;
    a.out[o] = f0;
    a.out[o + 1] = f1;
    if (a.dydx) {
#pragma unroll
      for (int c = 0; c < D; ++c) {
        a.dydx[o * D + c] = j0[c];
        a.dydx[(o + 1) * D + c] = j1[c];
      }
    }
  } else if constexpr (MODE == 1) {
    const int64_t e = (int64_t)l * a.PS + s;      // feature planes [NL][P]
    if (a.feat_f32) {
      reinterpret_cast<float*>(a.feat_pl)[2 * e] = f0;
      reinterpret_cast<float*>(a.feat_pl)[2 * e + 1] = f1;
    } else {
      union {
        uint32_t u;
        f16 h[2];
      } cv;
      cv.h[0] = (f16)fminf(fmaxf(f0 * PERMUTO_SDF_H_SCALE, -65504.0f), 65504.0f);
      cv.h[1] = (f16)fminf(fmaxf(f1 * PERMUTO_SDF_H_SCALE, -65504.0f), 65504.0f);
      reinterpret_cast<uint32_t*>(a.feat_pl)[e] = cv.u;
    }
  } else {
    const int64_t ep = (int64_t)l * a.PS + s;
    float* hp = a.h_pl + ep * 2;
    hp[0] = f0;
    hp[1] = f1;
    if (NSIM_J16 && !a.feat_f32) {      // the decoders of an fp16 field read f16 dh/dx planes (nsim_jplane_elem_bytes)
      f16* jp = reinterpret_cast<f16*>(a.J_pl) + ep * 6;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        jp[c] = (f16)fminf(fmaxf(j0[c], -65504.0f), 65504.0f);        // (saturate: a fine level's scale x feature jump can pass the f16 range)
        jp[3 + c] = (f16)fminf(fmaxf(j1[c], -65504.0f), 65504.0f);
      }
    } else {
      float* jp = reinterpret_cast<float*>(a.J_pl) + ep * 6;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        jp[c] = j0[c];
        jp[3 + c] = j1[c];
      }
    }
  }
}

// FIELD: dgrid[v][f] += w_v dL/dh[f] + g[f] (d w_v / d x . gn);  standalone: dgrid[v][f] += w_v dL/dout[f]
// Consecutive lanes are consecutive samples (of a ray, in the field's use): neighbours that fall into the same simplex hit
// the same d + 1 vertices, and same-address atomics of one instruction are separate, serialising requests (field.hip
// "grid scatter") -- runs of equal vertex indices are summed inside the wave first (ballot run heads + segmented shuffle
// scan, as k_lotd_scatter / k_lotd4_scatter do): 2.65 -> 0.51 ms per 0.31 M points of a 16-level pyramid on MI355X
// (the LoTD scatter on the same points: 0.50 ms).
template <int D, bool FIELD>
__global__ void __launch_bounds__(256) k_permuto_bwd(PermutoArgs a) {
  const int lane = nsim_lane();
  const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int l = blockIdx.y;
  const bool valid = s < a.S;            // (no early return: every lane takes part in the shuffles)
  float x[D];
#pragma unroll
  for (int i = 0; i < D; ++i) x[i] = 0.f;
  if (valid) permuto_point<D, FIELD>(a, s, x);
  Simplex<D> sp;
  float dB[D + 1][3];
  const bool second = FIELD && a.gn != nullptr;      // (wave-uniform)
  if (second) permuto_simplex<D, 3, true>(x, a.pm.scale[l], a.pm.shift[l], sp, dB);
  else permuto_simplex<D, 3, false>(x, a.pm.scale[l], a.pm.shift[l], sp, dB);
  float d0 = 0.f, d1 = 0.f, g0 = 0.f, g1 = 0.f, gn[3] = {0.f, 0.f, 0.f};
  if (valid) {
    if constexpr (FIELD) {
      const float* dp = a.dh_pl + ((int64_t)l * a.S + s) * 2;
      d0 = dp[0];
      d1 = dp[1];
      if (second) {
        const float* gp = a.g_pl + ((int64_t)l * a.S + s) * 2;
        g0 = gp[0];
        g1 = gp[1];
#pragma unroll
        for (int c = 0; c < 3; ++c) gn[c] = a.gn[3 * s + c];
      }
    } else {
      const int64_t o = s * (2 * a.pm.num_levels) + 2 * l;
      d0 = a.dL_dout[o];
      d1 = a.dL_dout[o + 1];
    }
  }
  float* base = a.dgrid + (int64_t)l * a.pm.T * 2;
#pragma unroll
  for (int r = 0; r <= D; ++r) {
    const uint32_t idx = permuto_vertex<D>(sp, r, a.pm.T);
    float v0 = sp.bary[r] * d0, v1 = sp.bary[r] * d1;
    if (second) {
      const float dwg = dB[r][0] * gn[0] + dB[r][1] * gn[1] + dB[r][2] * gn[2];
      v0 = v0 + g0 * dwg;
      v1 = v1 + g1 * dwg;
    }
    // runs of lanes with the same vertex: the run total ends up on the run's last lane
    const uint32_t key = valid ? idx : 0xffffffffu - (uint32_t)lane;
    const uint32_t pk = wave_shfl(key, lane - 1);
    const unsigned long long heads = wave_ballot(lane == 0 || pk != key);
    const unsigned long long below = heads & ((2ull << lane) - 1ull);
    const int run_start = 63 - __builtin_clzll(below);
#pragma unroll
    for (int dd = 1; dd < 64; dd <<= 1) {
      const float o0 = wave_shfl(v0, lane - dd), o1 = wave_shfl(v1, lane - dd);
      if (lane - dd >= run_start) {
        v0 = v0 + o0;
        v1 = v1 + o1;
      }
    }
    const bool emit = valid && (lane == 63 || ((heads >> (lane + 1)) & 1ull));
    if (emit) {
      if (v0 != 0.f) atomicAdd(base + 2 * (int64_t)idx, v0);
      if (v1 != 0.f) atomicAdd(base + 2 * (int64_t)idx + 1, v1);
    }
  }
}

// d L / d z of the conditioned field (GenerativePermutoConcat: z is LEARNED -- the auto-decoder's per-instance codes,
// app/models/shared/batched_neus.py:295-407): dz[ray][c] += sum_levels sum_f dL/dh[l][s][f] . d h[l][f] / d z_c, with
// d h / d z_c = sum_r (d bary_r / d z_c) table[v_r].  The second-order term has no z part: d h / d x (spatial) is a sum of
// table values times CONSTANTS inside a simplex, so the normals do not move with z.  Consecutive lanes are consecutive
// samples of a ray: the run total per (ray, c) is formed in the wave, one atomic per run.
template <int D>
__global__ void __launch_bounds__(256) k_permuto_dz(PermutoArgs a, float* __restrict__ dz) {
  constexpr int NZ = D - 3;
  const int lane = nsim_lane();
  const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int l = blockIdx.y;
  const bool valid = s < a.S;
  float x[D];
#pragma unroll
  for (int i = 0; i < D; ++i) x[i] = 0.f;
  if (valid) permuto_point<D, true>(a, s, x);
  Simplex<D> sp;
  float dB[D + 1][NZ];
  permuto_simplex<D, NZ, true, 3>(x, a.pm.scale[l], a.pm.shift[l], sp, dB);
  float d0 = 0.f, d1 = 0.f;
  int64_t ray = -1 - lane;
  if (valid) {
    const float* dp = a.dh_pl + ((int64_t)l * a.S + s) * 2;
    d0 = dp[0];
    d1 = dp[1];
    ray = a.ridx ? a.ridx[s] : 0;
  }
  float acc[NZ];
#pragma unroll
  for (int c = 0; c < NZ; ++c) acc[c] = 0.f;
  const int64_t base = (int64_t)l * a.pm.T * 2;
#pragma unroll
  for (int r = 0; r <= D; ++r) {
    float g0 = 0.f, g1 = 0.f;
    if (valid) permuto_load2(a.grid, base, permuto_vertex<D>(sp, r, a.pm.T), g0, g1);
    const float w = d0 * g0 + d1 * g1;
#pragma unroll
    for (int c = 0; c < NZ; ++c) acc[c] = acc[c] + dB[r][c] * w;
  }
  const int64_t pk = wave_shfl(ray, lane - 1);
  const unsigned long long heads = wave_ballot(lane == 0 || pk != ray);
  const unsigned long long below = heads & ((2ull << lane) - 1ull);
  const int run_start = 63 - __builtin_clzll(below);
#pragma unroll
  for (int c = 0; c < NZ; ++c) {
    float v = acc[c];
#pragma unroll
    for (int dd = 1; dd < 64; dd <<= 1) {
      const float o = wave_shfl(v, lane - dd);
      if (lane - dd >= run_start) v = v + o;
    }
    const bool emit = valid && (lane == 63 || ((heads >> (lane + 1)) & 1ull));
    if (emit && v != 0.f) atomicAdd(dz + (int64_t)NZ * ray + c, v);
  }
}

// ================================================================================== C ABI
template <int MODE>
static int permuto_launch_fwd(const NsimPermutoMeta* meta, const PermutoArgs& a, hipStream_t stream) {
  // field modes with fewer than 16 levels: rows 0..15 of the plane arrays are all written (zeros past the pyramid)
  // (more than 16 levels: zeros up to the next multiple of 8, the sampling decoder's K-step)
  const int rows = MODE == 0 ? meta->num_levels : (meta->num_levels < 16 ? 16 : ((meta->num_levels + 7) & ~7));
  const dim3 grid((unsigned)nsim_blocks(a.S, 256), (unsigned)rows), block(256);
  switch (meta->in_dim) {
    case 2: hipLaunchKernelGGL((k_permuto_fwd<2, 0>), grid, block, 0, stream, a); break;      // (standalone only: the field front end needs in_dim >= 3)
    case 3: hipLaunchKernelGGL((k_permuto_fwd<3, MODE>), grid, block, 0, stream, a); break;
    case 4: hipLaunchKernelGGL((k_permuto_fwd<4, MODE>), grid, block, 0, stream, a); break;
    case 5: hipLaunchKernelGGL((k_permuto_fwd<5, MODE>), grid, block, 0, stream, a); break;
    case 6: hipLaunchKernelGGL((k_permuto_fwd<6, MODE>), grid, block, 0, stream, a); break;
    case 7: hipLaunchKernelGGL((k_permuto_fwd<7, MODE>), grid, block, 0, stream, a); break;
    case 8: hipLaunchKernelGGL((k_permuto_fwd<8, MODE>), grid, block, 0, stream, a); break;
    default: return 41;
  }
  NSIM_CHECK_LAUNCH();
  return 0;
}

template <bool FIELD>
static int permuto_launch_bwd(const NsimPermutoMeta* meta, const PermutoArgs& a, hipStream_t stream) {
  const dim3 grid((unsigned)nsim_blocks(a.S, 256), (unsigned)meta->num_levels), block(256);
  switch (meta->in_dim) {
    case 2: hipLaunchKernelGGL((k_permuto_bwd<2, false>), grid, block, 0, stream, a); break;
    case 3: hipLaunchKernelGGL((k_permuto_bwd<3, FIELD>), grid, block, 0, stream, a); break;
    case 4: hipLaunchKernelGGL((k_permuto_bwd<4, FIELD>), grid, block, 0, stream, a); break;
    case 5: hipLaunchKernelGGL((k_permuto_bwd<5, FIELD>), grid, block, 0, stream, a); break;
    case 6: hipLaunchKernelGGL((k_permuto_bwd<6, FIELD>), grid, block, 0, stream, a); break;
    case 7: hipLaunchKernelGGL((k_permuto_bwd<7, FIELD>), grid, block, 0, stream, a); break;
    case 8: hipLaunchKernelGGL((k_permuto_bwd<8, FIELD>), grid, block, 0, stream, a); break;
    default: return 41;
  }
  NSIM_CHECK_LAUNCH();
  return 0;
}

extern "C" {

int nsim_permuto_fwd(const NsimPermutoMeta* meta, const void* grid_f16, const float* x, int64_t S, float* out, float* dydx,
                     void* stream) {
  const int rc = permuto_meta_check(meta);
  if (rc) return rc;
  if (S <= 0) return 0;
  if (!grid_f16 || !x || !out) return 4;
  PermutoArgs a = PermutoArgs();
  a.pm = permuto_dev(meta);
  a.grid = (const f16*)grid_f16;
  a.x = x; a.S = S; a.out = out; a.dydx = dydx;
  return permuto_launch_fwd<0>(meta, a, (hipStream_t)stream);
}

int nsim_permuto_bwd(const NsimPermutoMeta* meta, const float* x, int64_t S, const float* dL_dout, float* dgrid,
                     void* stream) {
  const int rc = permuto_meta_check(meta);
  if (rc) return rc;
  if (S <= 0) return 0;
  if (!x || !dL_dout || !dgrid) return 4;
  PermutoArgs a = PermutoArgs();
  a.pm = permuto_dev(meta);
  a.x = x; a.S = S; a.dL_dout = dL_dout; a.dgrid = dgrid;
  return permuto_launch_bwd<false>(meta, a, (hipStream_t)stream);
}

int nsim_permuto_gather(const NsimPermutoMeta* meta, const void* grid_f16, const float* x, const float* rays_o,
                        const float* rays_d, const float* t, const int64_t* ridx, const float* z, int64_t S,
                        const int64_t* n_dev, int64_t n_add, void* feat_planes, int feat_f32, float* h_planes,
                        void* J_planes, void* stream) {
  const int rc = permuto_meta_check(meta);
  if (rc) return rc;
  if (meta->in_dim < 3) return 41;
  if (S <= 0) return 0;
  if (!grid_f16) return 4;
  if (!x && !(rays_o && rays_d && t && ridx)) return 24;
  if (z && !ridx) return 29;
  if ((h_planes != nullptr) != (J_planes != nullptr)) return 28;
  if ((feat_planes != nullptr) == (h_planes != nullptr)) return 28;      // exactly one kind of output
  PermutoArgs a = PermutoArgs();
  a.pm = permuto_dev(meta);
  a.grid = (const f16*)grid_f16;
  a.x = x; a.rays_o = rays_o; a.rays_d = rays_d; a.t = t; a.ridx = ridx; a.z = z;
  a.S = S; a.PS = NSIM_PLANE_PITCH(S);
  a.S_dev = n_dev; a.S_add = n_add;
  a.feat_pl = feat_planes; a.feat_f32 = feat_f32;
  a.h_pl = h_planes; a.J_pl = J_planes;
  if (feat_planes) return permuto_launch_fwd<1>(meta, a, (hipStream_t)stream);
  return permuto_launch_fwd<2>(meta, a, (hipStream_t)stream);
}

int nsim_permuto_scatter(const NsimPermutoMeta* meta, const float* x, const float* rays_o, const float* rays_d,
                         const float* t, const int64_t* ridx, const float* z, int64_t S, const float* dh_planes,
                         const float* g_planes, const float* gn, float* dgrid, void* stream) {
  const int rc = permuto_meta_check(meta);
  if (rc) return rc;
  if (meta->in_dim < 3) return 41;
  if (S <= 0) return 0;
  if (!x && !(rays_o && rays_d && t && ridx)) return 24;
  if (z && !ridx) return 29;
  if (!dh_planes || !dgrid) return 28;
  if (gn && !g_planes) return 28;
  PermutoArgs a = PermutoArgs();
  a.pm = permuto_dev(meta);
  a.x = x; a.rays_o = rays_o; a.rays_d = rays_d; a.t = t; a.ridx = ridx; a.z = z;
  a.S = S;
  a.dh_pl = dh_planes; a.g_pl = g_planes; a.gn = gn; a.dgrid = dgrid;
  return permuto_launch_bwd<true>(meta, a, (hipStream_t)stream);
}

int nsim_permuto_dz(const NsimPermutoMeta* meta, const void* grid_f16, const float* x, const float* rays_o,
                    const float* rays_d, const float* t, const int64_t* ridx, const float* z, int64_t S,
                    const float* dh_planes, float* dz, void* stream) {
  const int rc = permuto_meta_check(meta);
  if (rc) return rc;
  if (meta->in_dim < 4) return 41;            // no condition inputs
  if (S <= 0) return 0;
  if (!grid_f16 || !dh_planes || !dz) return 28;
  if (!x && !(rays_o && rays_d && t && ridx)) return 24;
  if (!z || !ridx) return 29;
  PermutoArgs a = PermutoArgs();
  a.pm = permuto_dev(meta);
  a.grid = (const f16*)grid_f16;
  a.x = x; a.rays_o = rays_o; a.rays_d = rays_d; a.t = t; a.ridx = ridx; a.z = z;
  a.S = S;
  a.dh_pl = dh_planes;
  const dim3 grid((unsigned)nsim_blocks(S, 256), (unsigned)meta->num_levels), block(256);
  hipStream_t st = (hipStream_t)stream;
  switch (meta->in_dim) {
    case 4: hipLaunchKernelGGL((k_permuto_dz<4>), grid, block, 0, st, a, dz); break;
    case 5: hipLaunchKernelGGL((k_permuto_dz<5>), grid, block, 0, st, a, dz); break;
    case 6: hipLaunchKernelGGL((k_permuto_dz<6>), grid, block, 0, st, a, dz); break;
    case 7: hipLaunchKernelGGL((k_permuto_dz<7>), grid, block, 0, st, a, dz); break;
    case 8: hipLaunchKernelGGL((k_permuto_dz<8>), grid, block, 0, st, a, dz); break;
    default: return 41;
  }
  NSIM_CHECK_LAUNCH();
  return 0;
}

}  // extern "C"
