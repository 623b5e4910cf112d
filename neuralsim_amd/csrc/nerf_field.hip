// nerf_field.hip -- the NeRF++ distant-view model (``LoTDNeRFDistant``) on gfx950: inverse-radius cuboid-shell
// sampling, 4-D LoTD (quadrilinear, 16 corners per level) gather / scatter, density MLP (F -> 64 -> 1, softplus
// output) and radiance MLP ([features, SH-4 view dir, appearance-4] -> 64 -> 64 -> 3) forward + backward on the
// matrix cores, and sigma -> alpha.
//
// Replaces the native half of nr3d_lib.models.fields_distant.nerf.LoTDNeRFDistantModel as the reference drives it
// (app/models/single/nerf.py:145-196; call site app/renderers/single_volume_renderer.py:281-309; config
// code_single/configs/object_centric/lotd_neus.dtu.230814.yaml:186-247).  Executable spec: oracle/distant.py.
// Same register-resident transposed-MFMA scheme as field.hip (mfma_mlp.h); no second-order terms are needed here
// (the distant model has ``use_nablas: false``).
#ifndef NSIM_SCATTER_SCAN_EXIT
#define NSIM_SCATTER_SCAN_EXIT 1
#endif
#include "mfma_mlp.h"
#include <stdlib.h>

#define D4_MAX_LEVELS 16

struct Lotd4Dev {
  int num_levels;
  int res_xyz[D4_MAX_LEVELS], res_y[D4_MAX_LEVELS], res_z[D4_MAX_LEVELS], res_w[D4_MAX_LEVELS], type[D4_MAX_LEVELS];
  uint32_t size[D4_MAX_LEVELS];
  int64_t offset[D4_MAX_LEVELS];
};

static inline Lotd4Dev lotd4_dev(const NsimLotd4Meta* m) {
  Lotd4Dev d;
  d.num_levels = m->num_levels;
  for (int l = 0; l < D4_MAX_LEVELS; ++l) {
    const bool ok = l < m->num_levels;
    d.res_xyz[l] = ok ? m->res_xyz[l] : 2;
    d.res_y[l] = ok ? (m->res_y[l] > 0 ? m->res_y[l] : m->res_xyz[l]) : 2;      // 0: cubic level
    d.res_z[l] = ok ? (m->res_z[l] > 0 ? m->res_z[l] : m->res_xyz[l]) : 2;
    d.res_w[l] = ok ? m->res_w[l] : 2;
    d.type[l] = ok ? m->type[l] : 0;
    d.size[l] = ok ? m->size[l] : 16;
    d.offset[l] = ok ? m->offset[l] : 0;
  }
  return d;
}

static int lotd4_meta_check(const NsimLotd4Meta* m) {
  if (!m) return 10;
  if (m->num_levels < 1 || m->num_levels > D4_MAX_LEVELS) return 12;
  for (int l = 0; l < m->num_levels; ++l) {
    if (m->res_xyz[l] < 2 || m->res_w[l] < 2) return 13;
    if (m->res_y[l] == 1 || m->res_z[l] == 1 || m->res_y[l] < 0 || m->res_z[l] < 0) return 13;
    const uint64_t ry = m->res_y[l] > 0 ? m->res_y[l] : m->res_xyz[l], rz = m->res_z[l] > 0 ? m->res_z[l] : m->res_xyz[l];
    if (m->type[l] == NSIM_LOTD_DENSE) {
      if ((uint64_t)m->res_xyz[l] * ry * rz * m->res_w[l] != (uint64_t)m->size[l]) return 14;
    } else if (m->type[l] == NSIM_LOTD_HASH) {
      if (m->size[l] == 0 || (m->size[l] & (m->size[l] - 1)) != 0) return 17;
    } else {
      return 15;
    }
    if (m->offset[l] & 1) return 16;
  }
  return 0;
}

struct Cell4 {
  int c0[4];
  float w[4];
};

__device__ __forceinline__ Cell4 lotd4_cell(const float u[4], int Rx, int Ry, int Rz, int Rw) {
  Cell4 c;
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    const int R = a == 0 ? Rx : (a == 1 ? Ry : (a == 2 ? Rz : Rw));
    const float pos = u[a] * (float)(R - 1);
    float f = floorf(pos);
    f = fminf(fmaxf(f, 0.f), (float)(R - 2));
    c.c0[a] = (int)f;
    c.w[a] = pos - f;
  }
  return c;
}

__device__ __forceinline__ uint32_t lotd4_index(int cx, int cy, int cz, int cw, int Rx, int Ry, int Rz, int type,
                                                uint32_t T) {
  if (type == NSIM_LOTD_DENSE)
    return (uint32_t)cx + (uint32_t)Rx * ((uint32_t)cy + (uint32_t)Ry * ((uint32_t)cz + (uint32_t)Rz * (uint32_t)cw));
  const uint32_t h = (uint32_t)cx ^ ((uint32_t)cy * 2654435761u) ^ ((uint32_t)cz * 805459861u) ^
                     ((uint32_t)cw * 3674653429u);
  return h & (T - 1u);
}

// vertex order of the 4-D gathers: slot k reads the vertex with the coordinate parities of k's bits (lotd_dev.h, NSIM_GATHER_PARITY)
__device__ __forceinline__ int lotd4_slot_mask(const Cell4& c) {
  return NSIM_GATHER_PARITY ? ((c.c0[0] & 1) | ((c.c0[1] & 1) << 1) | ((c.c0[2] & 1) << 2) | ((c.c0[3] & 1) << 3)) : 0;
}

__device__ __forceinline__ float lotd4_weight(const Cell4& c, int corner) {
  float w = 1.0f;
#pragma unroll
  for (int a = 0; a < 4; ++a) w = w * (((corner >> a) & 1) ? c.w[a] : 1.0f - c.w[a]);
  return w;
}

__device__ __forceinline__ void lotd4_load2(const f16* grid, int64_t off, uint32_t idx, float& f0, float& f1) {
  const uint32_t raw = *reinterpret_cast<const uint32_t*>(grid + off + 2 * (int64_t)idx);
  union {
    uint32_t u;
    f16 h[2];
  } cv;
  cv.u = raw;
  f0 = (float)cv.h[0];
  f1 = (float)cv.h[1];
}

// ----------------------------------------------------------------------------------------- sampling
// One thread per (ray, shell): 1/r uniform in [1/r_max, 1/r_min]; the sample sits where the ray leaves the AABB
// scaled by r about its centre.
__global__ void __launch_bounds__(256) k_distant_shells(const float* __restrict__ rays_o, const float* __restrict__ rays_d,
                                                         const float* __restrict__ near, const float* __restrict__ jitter,
                                                         int64_t N, int K, float cx, float cy, float cz, float hx, float hy, float hz, float r_min,
                                                         float r_max, float* __restrict__ t_out,
                                                         float* __restrict__ u4_out, uint8_t* __restrict__ valid_out) {
  const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= N * K) return;
  const int64_t ray = s / K;
  const int k = (int)(s % K);
  const float u = jitter ? jitter[s] : 0.5f;
  const float inv_r = 1.0f / r_min + (((float)k + u) / (float)K) * (1.0f / r_max - 1.0f / r_min);
  const float r = 1.0f / inv_r;
  const float c[3] = {cx, cy, cz}, hf[3] = {hx, hy, hz};
  float tmin = -INFINITY, tmax = INFINITY, o[3], d[3];
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    o[a] = rays_o[3 * ray + a] - c[a];
    d[a] = rays_d[3 * ray + a];
    float ds = d[a];
    if (fabsf(ds) < 1e-12f) ds = (ds < 0.f) ? -1e-12f : 1e-12f;
    const float inv = 1.0f / ds;
    const float hr = hf[a] * r;
    const float t1 = (-hr - o[a]) * inv, t2 = (hr - o[a]) * inv;
    tmin = fmaxf(tmin, fminf(t1, t2));
    tmax = fminf(tmax, fmaxf(t1, t2));
  }
  const bool valid = (tmax > tmin) && (tmax > near[ray]);
  t_out[s] = tmax;
  valid_out[s] = valid ? 1 : 0;
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const float x = rays_o[3 * ray + a] + tmax * d[a];
    float un = (x - c[a]) / hf[a] * inv_r;
    un = un * 0.5f + 0.5f;
    u4_out[4 * s + a] = fminf(fmaxf(un, 0.f), 1.f);
  }
  u4_out[4 * s + 3] = fminf(fmaxf(inv_r, 0.f), 1.f);
}

// alpha_k = 1 - exp(-sigma_k * delta_k), delta = t_{k+1} - t_k; the last shell reaches to infinity (1e10,
// ``include_inf_distance: true``, object-centric configs) or repeats the previous interval (``false``: the street
// config, whose sky model takes the remaining transmittance); 0 on invalid shells
__device__ __forceinline__ float shell_delta(const float* t, const uint8_t* valid, int64_t s, int k, int K, int include_inf) {
  if (k + 1 < K) return t[s + 1] - t[s];
  if (include_inf) return 1e10f;
  return (k > 0 && valid[s - 1]) ? t[s] - t[s - 1] : 0.f;
}

__global__ void __launch_bounds__(256) k_density_alpha_fwd(const float* __restrict__ sigma, const float* __restrict__ t,
                                                            const uint8_t* __restrict__ valid, int64_t N, int K,
                                                            int include_inf, float* __restrict__ alpha) {
  const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= N * K) return;
  const int k = (int)(s % K);
  const float delta = shell_delta(t, valid, s, k, K, include_inf);
  alpha[s] = valid[s] ? 1.0f - expf(-sigma[s] * delta) : 0.f;
}

__global__ void __launch_bounds__(256) k_density_alpha_bwd(const float* __restrict__ sigma, const float* __restrict__ t,
                                                            const uint8_t* __restrict__ valid,
                                                            const float* __restrict__ dalpha, int64_t N, int K,
                                                            int include_inf, float* __restrict__ dsigma) {
  const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= N * K) return;
  const int k = (int)(s % K);
  const float delta = shell_delta(t, valid, s, k, K, include_inf);
  dsigma[s] = valid[s] ? dalpha[s] * delta * expf(-sigma[s] * delta) : 0.f;
}

// ----------------------------------------------------------------------------------------- weight pack
enum { N_D1 = 0, N_D1T, N_Q1, N_Q2, N_Q3, N_Q3T, N_Q2T, N_Q1T, N_MCOUNT };
enum { NV_DB1 = 0, NV_DWH, NV_RB1, NV_RB2, NV_RB3, NV_SCAL, NV_COUNT };
static const int kNUo[N_MCOUNT] = {64, 32, 64, 64, 32, 64, 64, 64};
static const int kNUi[N_MCOUNT] = {32, 64, 64, 64, 64, 32, 64, 64};

struct NerfLayout {
  int64_t mat[N_MCOUNT], vec[NV_COUNT], total;
  int elt;
};
static inline NerfLayout nerf_layout(int precision) {
  NerfLayout L;
  L.elt = precision == 0 ? 2 : 4;
  int64_t off = 0;
  for (int m = 0; m < N_MCOUNT; ++m) {
    L.mat[m] = off;
    off += (int64_t)kNUo[m] * kNUi[m] * L.elt;
  }
  for (int v = 0; v < NV_COUNT; ++v) {
    L.vec[v] = off;
    off += 64 * 4;
  }
  L.total = off;
  return L;
}

// radiance input slot (0..63) -> column of the [64 x (F+20)] first radiance layer, or -1
__host__ __device__ inline int q1_col(int slot, int F) {
  if (slot < 32) return slot < F ? slot : -1;
  if (slot < 48) return F + (slot - 32);
  if (slot < 52) return F + 16 + (slot - 48);
  return -1;
}

__device__ __forceinline__ float nerf_src(int mat, int row, int col, int F, const float* den_w, const float* rad_w) {
  const int K1 = F + 20;
  const int q2 = 64 * K1, q3 = q2 + 4096;
  switch (mat) {
    case N_D1: return col < F ? den_w[row * F + col] : 0.f;
    case N_D1T: return row < F ? den_w[col * F + row] : 0.f;
    case N_Q1: { const int c = q1_col(col, F); return c >= 0 ? rad_w[row * K1 + c] : 0.f; }
    case N_Q1T: { const int c = q1_col(row, F); return c >= 0 ? rad_w[col * K1 + c] : 0.f; }
    case N_Q2: return rad_w[q2 + row * 64 + col];
    case N_Q2T: return rad_w[q2 + col * 64 + row];
    case N_Q3: return row < 3 ? rad_w[q3 + row * 64 + col] : 0.f;
    case N_Q3T: return col < 3 ? rad_w[q3 + col * 64 + row] : 0.f;
  }
  return 0.f;
}

struct NerfDims {
  int uo[N_MCOUNT], ui[N_MCOUNT];
};

__global__ void __launch_bounds__(256) k_nerf_pack(NerfLayout L, NerfDims dims, int F, const float* __restrict__ den_w,
                                                    const float* __restrict__ den_b, const float* __restrict__ rad_w,
                                                    const float* __restrict__ rad_b, char* __restrict__ wpack) {
  const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int64_t base = 0;
  for (int m = 0; m < N_MCOUNT; ++m) {
    const int Uo = dims.uo[m], Ui = dims.ui[m];
    const int64_t cnt = (int64_t)Uo * Ui;
    if (tid >= base && tid < base + cnt) {
      const int64_t k = tid - base;
      if (L.elt == 2) {
        const int e = (int)(k & 7), lane = (int)((k >> 3) & 63), fs = (int)(k >> 9);
        const int nS = Ui / 16, mo = fs / nS, s = fs % nS;
        const int row = 32 * mo + (lane & 31), col = 16 * s + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
        ((f16*)(wpack + L.mat[m]))[k] = (f16)nerf_src(m, row, col, F, den_w, rad_w);
      } else {
        const int lane = (int)(k & 63), fr = (int)(k >> 6), r = fr & 15, fm = fr >> 4;
        const int nMi = Ui / 32, mo = fm / nMi, mi = fm % nMi;
        ((float*)(wpack + L.mat[m]))[k] = nerf_src(m, 32 * mo + (lane & 31), unit_of(mi, r, lane >> 5), F, den_w, rad_w);
      }
      return;
    }
    base += cnt;
  }
  const int64_t vtid = tid - base;
  if (vtid >= 0 && vtid < (int64_t)NV_COUNT * 64) {
    const int v = (int)(vtid >> 6), k = (int)(vtid & 63);
    const int hi = k >> 5, m = (k >> 4) & 1, r = k & 15, u = unit_of(m, r, hi);
    float val = 0.f;
    switch (v) {
      case NV_DB1: val = den_b[u]; break;
      case NV_DWH: val = den_w[64 * F + u]; break;
      case NV_RB1: val = rad_b[u]; break;
      case NV_RB2: val = rad_b[64 + u]; break;
      case NV_RB3: val = u < 3 ? rad_b[128 + u] : 0.f; break;
      case NV_SCAL: val = (k == 0) ? den_b[64] : 0.f; break;
    }
    ((float*)(wpack + L.vec[v]))[k] = val;
  }
}

// ----------------------------------------------------------------------------------------- MLP kernels
struct NerfArgs {
  Lotd4Dev lotd;
  NerfLayout lay;
  int F;                                  // real feature count = 2 * num_levels (<= 32)
  const f16* grid;
  const char* wpack;
  const float* u4;                        // [S,4] in [0,1]
  const float* rays_d;                    // [N,3]
  const float* h_appear;                  // [N,4] or NULL
  const uint8_t* valid;                   // [S]
  int64_t S;
  int K;                                  // shells per ray (ray = s / K)
  float *sigma, *rgb;                     // forward outputs
  float* h_pl;                            // [16][S][2] saved features
  int h_from_planes;                      // forward: the features were gathered level-major (k_lotd4_gather_lm)
  const float *sigma_fwd, *rgb_fwd;       // saved forward outputs
  const float *dsigma, *drgb;             // upstream
  float* dh_pl;                           // [16][S][2] hand-off to the scatter
  float *dden_w, *dden_b, *drad_w, *drad_b, *dh_appear;
};

#define NERF_WAVES 4

// Stage the pack into LDS (fp16 mode): everything, or -- VEC_ONLY -- just the per-lane vectors (the matrices are then
// read from L2, which frees LDS for private weight-gradient accumulators).  Returns the base the VECTORS are addressed
// from: nvec() uses offsets relative to lay.vec[0].
template <int PREC, bool VEC_ONLY>
__device__ __forceinline__ const char* nerf_stage_weights(char* smem, const NerfArgs& a, int& used, const char*& WM) {
  if constexpr (PREC == 0) {
    const int64_t first = VEC_ONLY ? a.lay.vec[0] : 0;
    const int n16 = (int)((a.lay.total - first + 15) >> 4);
    const f16x8* src = reinterpret_cast<const f16x8*>(a.wpack + first);
    f16x8* dst = reinterpret_cast<f16x8*>(smem);
    for (int i = threadIdx.x; i < n16; i += blockDim.x) dst[i] = src[i];
    used = n16 << 4;
    __syncthreads();
    WM = VEC_ONLY ? a.wpack : smem;
    return VEC_ONLY ? smem : smem + a.lay.vec[0];
  } else {
    used = 0;
    WM = a.wpack;
    return a.wpack + a.lay.vec[0];
  }
}

__device__ __forceinline__ float nvec(const char* WV, const NerfLayout& L, int v, int hi, int k) {
  return reinterpret_cast<const float*>(WV + (L.vec[v] - L.vec[0]))[hi * 32 + k];
}

// radiance input, second M-tile (slots 32..63): SH-4 of the view direction, appearance code, zero padding
__device__ __forceinline__ void nerf_rin_tail(float (&rin)[32], const float vd[3], const float ha[4], int hi) {
  float sh[16];
  {
    const float x = vd[0], y = vd[1], z = vd[2];
    const float xy = x * y, xz = x * z, yz = y * z, x2 = x * x, y2 = y * y, z2 = z * z;
    sh[0] = 0.28209479177387814f;
    sh[1] = -0.48860251190291987f * y;
    sh[2] = 0.48860251190291987f * z;
    sh[3] = -0.48860251190291987f * x;
    sh[4] = 1.0925484305920792f * xy;
    sh[5] = -1.0925484305920792f * yz;
    sh[6] = 0.94617469575755997f * z2 - 0.31539156525251999f;
    sh[7] = -1.0925484305920792f * xz;
    sh[8] = 0.54627421529603959f * x2 - 0.54627421529603959f * y2;
    sh[9] = 0.59004358992664352f * y * (-3.0f * x2 + y2);
    sh[10] = 2.8906114426405538f * xy * z;
    sh[11] = 0.45704579946446572f * y * (1.0f - 5.0f * z2);
    sh[12] = 0.3731763325901154f * z * (5.0f * z2 - 3.0f);
    sh[13] = 0.45704579946446572f * x * (1.0f - 5.0f * z2);
    sh[14] = 1.4453057213202769f * z * (x2 - y2);
    sh[15] = 0.59004358992664352f * x * (-x2 + 3.0f * y2);
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int slot = unit_of(0, r, hi);  // position inside the second M-tile
    float v = 0.f;
    if (slot < 16) v = sh[slot];
    else if (slot < 20) v = ha[slot - 16];
    rin[16 + r] = v;
  }
}

struct NerfPoint {
  int64_t s, ray;
  bool valid;
  float vd[3], ha[4];
};
__device__ __forceinline__ NerfPoint nerf_point(const NerfArgs& a, int64_t tile, int j) {
  NerfPoint p;
  p.s = tile * 32 + j;
  p.valid = p.s < a.S;
  p.ray = p.valid ? p.s / a.K : 0;
  p.vd[0] = p.vd[1] = 0.f;
  p.vd[2] = 1.f;
  p.ha[0] = p.ha[1] = p.ha[2] = p.ha[3] = 0.f;
  if (p.valid) {
#pragma unroll
    for (int c = 0; c < 3; ++c) p.vd[c] = a.rays_d[3 * p.ray + c];
    if (a.h_appear) {
#pragma unroll
      for (int c = 0; c < 4; ++c) p.ha[c] = a.h_appear[4 * p.ray + c];
    }
  }
  return p;
}

__device__ __forceinline__ float softplus1(float x) { return x > 20.f ? x : log1pf(expf(x)); }

// fp16 backward: three waves per workgroup, each with a PRIVATE copy of the weight-gradient accumulators in LDS
// (plain read-add-write -- LDS float atomics cost ~800 cycles per instruction, mfma_mlp.h:dw_flush), matrices from L2.
#define NERF_WAVES_PRIV 3
template <int PREC, int BWD>
__global__ void __launch_bounds__(64 * ((PREC == 0 && BWD) ? NERF_WAVES_PRIV : NERF_WAVES)) k_nerf(NerfArgs a) {
  NSIM_DYN_SMEM(smem);
  constexpr bool PRIV = (PREC == 0 && BWD);
  constexpr int NW = PRIV ? NERF_WAVES_PRIV : NERF_WAVES;
  const int lane = nsim_lane(), j = lane & 31, hi = lane >> 5;
  const int wave = (int)(threadIdx.x >> 6);
  const NerfLayout& L = a.lay;
  int wbytes;
  const char* WM;
  const char* W = nerf_stage_weights<PREC, PRIV>(smem, a, wbytes, WM);
  // LDS accumulators (backward): D1 [64x32], DWH [64], DB1 [64], bd [4], Q1 [64x64], Q2 [64x64], Q3 [3x64], RB1, RB2, RB3
  constexpr int A_D1 = 0, A_DWH = 2048, A_DB1 = 2112, A_BD = 2176, A_Q1 = 2180, A_Q2 = A_Q1 + 4096, A_Q3 = A_Q2 + 4096,
                A_RB1 = A_Q3 + 192, A_RB2 = A_RB1 + 64, A_RB3 = A_RB2 + 64, A_TOTAL = A_RB3 + 4;
  float* accum = nullptr;
  char* stA = nullptr;
  char* stB = nullptr;
  if constexpr (BWD) {
    constexpr int ACC_BYTES = (A_TOTAL * 4 + 15) & ~15;
    accum = reinterpret_cast<float*>(smem + wbytes + (PRIV ? wave * ACC_BYTES : 0));
    char* stbase = smem + wbytes + (PRIV ? NW : 1) * ACC_BYTES + wave * stage_bytes_per_wave<PREC>();
    stA = stbase;
    stB = stbase + stage_bytes_per_wave<PREC>() / 2;
    if constexpr (PRIV) {
      for (int i = lane; i < A_TOTAL; i += 64) accum[i] = 0.f;
    } else {
      for (int i = threadIdx.x; i < A_TOTAL; i += blockDim.x) accum[i] = 0.f;
    }
    __syncthreads();
  }
  const float b_d = reinterpret_cast<const float*>(W + (L.vec[NV_SCAL] - L.vec[0]))[0];
  const int64_t ntiles = (a.S + 31) / 32;
  const int64_t wstride = (int64_t)gridDim.x * NW;
  for (int64_t tile = (int64_t)blockIdx.x * NW + wave; tile < ntiles; tile += wstride) {
    const NerfPoint p = nerf_point(a, tile, j);
    const int64_t s = p.s;
    if constexpr (BWD) {
      // a tile without a single live shell (behind an opaque foreground: the renderer's transmittance mask, folded into
      // ``valid`` by the host) contributes nothing -- its planes are not read, its dh rows not written (the scatter
      // skips the same shells)
      if (wave_ballot(p.valid && a.valid[s] != 0) == 0ull) continue;
    }
    // -------------------------------------------------------------- features: gather (forward) or planes (backward)
    float rin[32];  // [0,16): own features (first M-tile of the radiance input), [16,32): SH / appearance slots
    if (!BWD && a.h_from_planes) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
#pragma unroll
        for (int b = 0; b < 2; ++b) {
          const int l = 4 * q + 2 * hi + b;
          float f0 = 0.f, f1 = 0.f;
          if (p.valid && l < a.lotd.num_levels) {
            const float* hp = a.h_pl + ((int64_t)l * a.S + s) * 2;
            f0 = hp[0];
            f1 = hp[1];
          }
          rin[4 * q + 2 * b] = f0;
          rin[4 * q + 2 * b + 1] = f1;
        }
      }
    } else if constexpr (!BWD) {
      float u[4] = {0.f, 0.f, 0.f, 0.f};
      if (p.valid) {
#pragma unroll
        for (int c = 0; c < 4; ++c) u[c] = a.u4[4 * s + c];
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
#pragma unroll
        for (int b = 0; b < 2; ++b) {
          const int l = 4 * q + 2 * hi + b;   // level slot; slots >= num_levels are padding
          float f0 = 0.f, f1 = 0.f;
          if (l < a.lotd.num_levels) {
            const int Rx = a.lotd.res_xyz[l], Ry = a.lotd.res_y[l], Rz = a.lotd.res_z[l], Rw = a.lotd.res_w[l];
            const Cell4 c = lotd4_cell(u, Rx, Ry, Rz, Rw);
            const int pm4 = lotd4_slot_mask(c);
#pragma unroll
            for (int slot = 0; slot < 16; ++slot) {
              const int corner = slot ^ pm4;
              const float w = lotd4_weight(c, corner);
              const uint32_t idx = lotd4_index(c.c0[0] + (corner & 1), c.c0[1] + ((corner >> 1) & 1),
                                               c.c0[2] + ((corner >> 2) & 1), c.c0[3] + ((corner >> 3) & 1), Rx, Ry, Rz,
                                               a.lotd.type[l], a.lotd.size[l]);
              float g0, g1;
              lotd4_load2(a.grid, a.lotd.offset[l], idx, g0, g1);
              f0 = f0 + w * g0;
              f1 = f1 + w * g1;
            }
          }
          rin[4 * q + 2 * b] = f0;
          rin[4 * q + 2 * b + 1] = f1;
          if (a.h_pl && p.valid) {
            float* hp = a.h_pl + ((int64_t)l * a.S + s) * 2;
            hp[0] = f0;
            hp[1] = f1;
          }
        }
      }
    } else {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
#pragma unroll
        for (int b = 0; b < 2; ++b) {
          const int l = 4 * q + 2 * hi + b;
          float f0 = 0.f, f1 = 0.f;
          if (p.valid && l < a.lotd.num_levels) {      // (the level-major gather writes the pyramid's own levels only)
            const float* hp = a.h_pl + ((int64_t)l * a.S + s) * 2;
            f0 = hp[0];
            f1 = hp[1];
          }
          rin[4 * q + 2 * b] = f0;
          rin[4 * q + 2 * b + 1] = f1;
        }
      }
    }
    float h[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) h[r] = rin[r];
    nerf_rin_tail(rin, p.vd, p.ha, hi);
    // -------------------------------------------------------------- density + radiance forward
    float a1[32];
    dense<PREC, 2, 1>(a1, WM + L.mat[N_D1], h, true);
#pragma unroll
    for (int k = 0; k < 32; ++k) a1[k] = fmaxf(a1[k] + nvec(W, L, NV_DB1, hi, k), 0.f);
    float r1[32], r2[32];
    dense<PREC, 2, 2>(r1, WM + L.mat[N_Q1], rin, true);
#pragma unroll
    for (int k = 0; k < 32; ++k) r1[k] = fmaxf(r1[k] + nvec(W, L, NV_RB1, hi, k), 0.f);
    dense<PREC, 2, 2>(r2, WM + L.mat[N_Q2], r1, false);
#pragma unroll
    for (int k = 0; k < 32; ++k) r2[k] = fmaxf(r2[k] + nvec(W, L, NV_RB2, hi, k), 0.f);
    if constexpr (!BWD) {
      float raw = 0.f;
#pragma unroll
      for (int k = 0; k < 32; ++k) raw = raw + nvec(W, L, NV_DWH, hi, k) * a1[k];
      raw = raw + wave_shfl_xor(raw, 32) + b_d;
      float o3[16];
      dense<PREC, 1, 2>(o3, WM + L.mat[N_Q3], r2, false);
      if (p.valid && hi == 0) {
        a.sigma[s] = softplus1(raw);
#pragma unroll
        for (int c = 0; c < 3; ++c) a.rgb[3 * s + c] = 1.0f / (1.0f + nsim_fast_exp(-(o3[c] + nvec(W, L, NV_RB3, hi, c))));
      }
      continue;
    }
    if constexpr (BWD) {
      float gs = 0.f, sg = 0.f, gr[3] = {0.f, 0.f, 0.f}, rgbv[3] = {0.f, 0.f, 0.f};
      if (p.valid && a.valid[s]) {
        gs = a.dsigma ? a.dsigma[s] : 0.f;
        sg = a.sigma_fwd[s];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          gr[c] = a.drgb ? a.drgb[3 * s + c] : 0.f;
          rgbv[c] = a.rgb_fwd[3 * s + c];
        }
      }
      // ---- radiance branch
      float dout[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) dout[r] = 0.f;
      if (hi == 0) {
#pragma unroll
        for (int c = 0; c < 3; ++c) dout[c] = gr[c] * rgbv[c] * (1.0f - rgbv[c]);
      }
      dw_product<PREC, 1, 2, PRIV>(stA, stB, dout, r2, accum + A_Q3, 64, 3, 64, accum + A_RB3);
      float dr2[32];
      dense<PREC, 2, 1>(dr2, WM + L.mat[N_Q3T], dout, true);
#pragma unroll
      for (int k = 0; k < 32; ++k) dr2[k] = r2[k] > 0.f ? dr2[k] : 0.f;
      dw_product<PREC, 2, 2, PRIV>(stA, stB, dr2, r1, accum + A_Q2, 64, 64, 64, accum + A_RB2);
      float dr1[32];
      dense<PREC, 2, 2>(dr1, WM + L.mat[N_Q2T], dr2, true);
#pragma unroll
      for (int k = 0; k < 32; ++k) dr1[k] = r1[k] > 0.f ? dr1[k] : 0.f;
      dw_product<PREC, 2, 2, PRIV>(stA, stB, dr1, rin, accum + A_Q1, 64, 64, 64, accum + A_RB1);
      float din[32];
      dense<PREC, 2, 2>(din, WM + L.mat[N_Q1T], dr1, true);
      // appearance slots 48..51 = second M-tile local 16..19: (hi0: r8..11 -> 16..19)
      // (summed over the shells of a ray inside the wave first: one atomic per ray and channel, not one per sample)
      if (a.dh_appear) {
        float c0 = din[16 + 8], c1 = din[16 + 9], c2 = din[16 + 10], c3 = din[16 + 11];
        const bool v = p.valid && hi == 0;
        const bool last = halfwave_run_sum2(p.ray, v, c0, c1);
        halfwave_run_sum2(p.ray, v, c2, c3);
        if (last) {
          float* dst = a.dh_appear + 4 * p.ray;
          atomicAdd(dst, c0);
          atomicAdd(dst + 1, c1);
          atomicAdd(dst + 2, c2);
          atomicAdd(dst + 3, c3);
        }
      }
      // ---- density branch: sigma = softplus(raw) -> d raw = d sigma * (1 - exp(-sigma))
      const float draw = gs * (1.0f - nsim_fast_exp(-sg));
      float da[32], whv[32];
#pragma unroll
      for (int k = 0; k < 32; ++k) {
        whv[k] = draw * a1[k];
        da[k] = a1[k] > 0.f ? draw * nvec(W, L, NV_DWH, hi, k) : 0.f;
      }
      rowsum_acc<PREC, 2, PRIV>(stA, whv, accum + A_DWH, 64);
      {
        float v = (hi == 0) ? draw : 0.f;
        v = wave_sum(v);
        if (lane == 0 && v != 0.f) {
          if constexpr (PRIV) accum[A_BD] = accum[A_BD] + v;
          else atomicAdd(&accum[A_BD], v);
        }
      }
      dw_product<PREC, 2, 1, PRIV>(stA, stB, da, h, accum + A_D1, 32, 64, 32, accum + A_DB1);
      float dh[16];
      dense<PREC, 1, 2>(dh, WM + L.mat[N_D1T], da, true);
      if (p.valid && a.dh_pl) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
#pragma unroll
          for (int b = 0; b < 2; ++b) {
            const int l = 4 * q + 2 * hi + b;
            if (l >= a.lotd.num_levels) continue;      // (the scatter reads the pyramid's own levels only)
            float* dp = a.dh_pl + ((int64_t)l * a.S + s) * 2;
            dp[0] = dh[4 * q + 2 * b] + din[4 * q + 2 * b];
            dp[1] = dh[4 * q + 2 * b + 1] + din[4 * q + 2 * b + 1];
          }
        }
      }
    }
  }
  if constexpr (BWD) {
    __syncthreads();
    const int F = a.F, K1 = a.F + 20;
    for (int i = threadIdx.x; i < A_TOTAL; i += blockDim.x) {
      float v;
      if constexpr (PRIV) {
        constexpr int ACC_FLOATS = ((A_TOTAL * 4 + 15) & ~15) / 4;
        const float* a0 = reinterpret_cast<const float*>(smem + wbytes);
        v = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) v += a0[w * ACC_FLOATS + i];
      } else {
        v = accum[i];
      }
      if (v == 0.f) continue;
      float* dst = nullptr;
      if (i < A_DWH) {
        const int row = i >> 5, col = i & 31;
        if (col < F) dst = a.dden_w + row * F + col;
      } else if (i < A_DB1) dst = a.dden_w + 64 * F + (i - A_DWH);
      else if (i < A_BD) dst = a.dden_b + (i - A_DB1);
      else if (i < A_Q1) dst = (i == A_BD) ? a.dden_b + 64 : nullptr;
      else if (i < A_Q2) {
        const int row = (i - A_Q1) >> 6, c = q1_col((i - A_Q1) & 63, F);
        if (c >= 0) dst = a.drad_w + row * K1 + c;
      } else if (i < A_Q3) dst = a.drad_w + 64 * K1 + (i - A_Q2);
      else if (i < A_RB1) dst = a.drad_w + 64 * K1 + 4096 + (i - A_Q3);
      else if (i < A_RB2) dst = a.drad_b + (i - A_RB1);
      else if (i < A_RB3) dst = a.drad_b + 64 + (i - A_RB2);
      else dst = (i - A_RB3 < 3) ? a.drad_b + 128 + (i - A_RB3) : nullptr;
      if (dst) atomicAdd(dst, v);
    }
  }
}

// ----------------------------------------------------------------------------------------- 4-D level-major gather
// The forward's table reads as their own launch, in the form of field.hip's k_lotd_gather_lm: every wave owns
// G4_PTS x 64 shell points and walks the levels dealt to ITS XCD (block b runs on XCD b % 8 -- used for speed only), so
// at any moment an XCD streams one or two tables (<= 2 MB each at T = 2^19) through its 4 MB L2 instead of all 16-32 MB
// of the pyramid at once, with G4_PTS x 16 independent 4-byte loads in flight per lane.  The 64 lanes are consecutive
// shells of one ray: on the coarse levels they share cells (same lines: coalesced by the texture unit), the fine hashed
// levels cost one line per x-pair.  Output: the f32 planes h [16][S][2] the decoders (forward AND backward) read.
#ifndef G4_PTS
#define G4_PTS 2
#endif
struct Gather4Args {
  Lotd4Dev lotd;
  const f16* grid;
  const float* u4;
  int64_t S;
  float* h_pl;
  signed char n[8], lv[8][D4_MAX_LEVELS], half[8][D4_MAX_LEVELS];
};

__global__ void __launch_bounds__(64) k_lotd4_gather_lm(Gather4Args a) {
  const int lane = nsim_lane();
  const int xcd = (int)(blockIdx.x & 7u);
  const int64_t s0 = (int64_t)(blockIdx.x >> 3) * (64 * G4_PTS) + lane;
  float u[G4_PTS][4];
#pragma unroll
  for (int q = 0; q < G4_PTS; ++q) {
    const int64_t s = s0 + 64 * q;
#pragma unroll
    for (int c = 0; c < 4; ++c) u[q][c] = s < a.S ? a.u4[4 * s + c] : 0.f;
  }
  const bool second_half = (blockIdx.x >> 3) >= ((gridDim.x >> 3) + 1) / 2;
  const int nl = a.n[xcd];
#pragma unroll 1
  for (int k = 0; k < nl; ++k) {
    const int l = a.lv[xcd][k];
    const int hf = a.half[xcd][k];
    if ((hf == 1 && second_half) || (hf == 2 && !second_half)) continue;
    const int Rx = a.lotd.res_xyz[l], Ry = a.lotd.res_y[l], Rz = a.lotd.res_z[l], Rw = a.lotd.res_w[l];
    const int type = a.lotd.type[l];
    const uint32_t T = a.lotd.size[l];
    const int64_t off = a.lotd.offset[l];
    float f0[G4_PTS], f1[G4_PTS];
#pragma unroll
    for (int q = 0; q < G4_PTS; ++q) {
      const Cell4 c = lotd4_cell(u[q], Rx, Ry, Rz, Rw);
      f0[q] = f1[q] = 0.f;
      const int pm4 = lotd4_slot_mask(c);
#pragma unroll
      for (int slot = 0; slot < 16; ++slot) {
        const int corner = slot ^ pm4;
        const float w = lotd4_weight(c, corner);
        const uint32_t idx = lotd4_index(c.c0[0] + (corner & 1), c.c0[1] + ((corner >> 1) & 1), c.c0[2] + ((corner >> 2) & 1),
                                         c.c0[3] + ((corner >> 3) & 1), Rx, Ry, Rz, type, T);
        float g0, g1;
        lotd4_load2(a.grid, off, idx, g0, g1);
        f0[q] = f0[q] + w * g0;
        f1[q] = f1[q] + w * g1;
      }
    }
#pragma unroll
    for (int q = 0; q < G4_PTS; ++q) {
      const int64_t s = s0 + 64 * q;
      if (s < a.S) {
        float* hp = a.h_pl + ((int64_t)l * a.S + s) * 2;
        hp[0] = f0[q];
        hp[1] = f1[q];
      }
    }
  }
}

// hashed levels cost 1, dense ones ~0.35 (their lines mostly hit L1): most expensive first onto the least loaded XCD,
// split into two halves of the point range when a whole level would overload it (field.hip deal_levels)
static void deal_levels4(const NsimLotd4Meta* m, Gather4Args& a) {
  const int NL = m->num_levels;
  float cost[D4_MAX_LEVELS], load[8] = {0, 0, 0, 0, 0, 0, 0, 0}, total = 0.f;
  bool used[D4_MAX_LEVELS] = {false};
  for (int l = 0; l < NL; ++l) {
    cost[l] = m->type[l] == NSIM_LOTD_HASH ? 1.0f : 0.35f;
    total += cost[l];
  }
  const float limit = total / 8.0f * 1.08f;
  for (int xc = 0; xc < 8; ++xc) a.n[xc] = 0;
  auto least = [&]() {
    int tx = 0;
    for (int xc = 1; xc < 8; ++xc)
      if (load[xc] < load[tx]) tx = xc;
    return tx;
  };
  auto put = [&](int xc, int l, int hf, float c) {
    a.lv[xc][(int)a.n[xc]] = (signed char)l;
    a.half[xc][(int)a.n[xc]++] = (signed char)hf;
    load[xc] += c;
  };
  for (int it = 0; it < NL; ++it) {
    int best = -1;
    for (int l = 0; l < NL; ++l)
      if (!used[l] && (best < 0 || cost[l] > cost[best] || (cost[l] == cost[best] && m->size[l] > m->size[best]))) best = l;
    used[best] = true;
    const int x0 = least();
    if (load[x0] + cost[best] <= limit || load[x0] == 0.f) {
      put(x0, best, 0, cost[best]);
    } else {
      put(x0, best, 1, 0.5f * cost[best]);
      put(least(), best, 2, 0.5f * cost[best]);
    }
  }
}

// ----------------------------------------------------------------------------------------- 4-D scatter
// dgrid[level][vertex][f] += w_c * dh[f]; level-major, one lane per sample, x-adjacent corner pairs issued
// quad-transposed so {x0.f0,x0.f1,x1.f0,x1.f1} leave as one request (see field.hip / tools/atomic_bench2.hip).
struct Scatter4Args {
  Lotd4Dev lotd;
  const float* u4;
  const uint8_t* valid;
  int64_t S;
  const float* dh_pl;
  float* dgrid;
  int dedup_max_rw;
  int parity_slots;
};

// CONSEC (default since round 6, NSIM_SCATTER_GROUP=0: off): issue I of the quad transposition carries the 16 consecutive shells
// 16 I + q instead of the shells 4q + I; with the parity slots (NSIM_SCATTER_PARITY) the street step's 4-D scatter went
// 2.15 -> 1.99 (parity slots) -> 1.74 ms (+ consecutive issue), see k_lotd_scatter in field.hip
template <bool CONSEC>
__global__ void __launch_bounds__(256) k_lotd4_scatter(Scatter4Args a) {
  const int lane = nsim_lane();
  const int l = blockIdx.y;
  const int Rx = a.lotd.res_xyz[l], Ry = a.lotd.res_y[l], Rz = a.lotd.res_z[l], Rw = a.lotd.res_w[l];
  // K = 64 shells per ray: a wave holds consecutive shells of (mostly) ONE ray.  Far shells converge to a fixed
  // (x/r) direction and step through few 1/r cells on the coarser levels, so runs of lanes hit the same vertex:
  // collapse them (run heads by ballot, segmented shuffle scan) before the atomics -- the kernel is bound by the
  // atomic request rate, the extra shuffles are free.
  const bool dedup = Rw <= a.dedup_max_rw;
  const int rq = lane & 3;
  float* base = a.dgrid + a.lotd.offset[l];
  const int64_t nchunks = (a.S + 63) / 64;
  const int64_t wstride = (int64_t)gridDim.x * 4;
  for (int64_t chunk = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); chunk < nchunks; chunk += wstride) {
    const int64_t s = chunk * 64 + lane;
    const bool valid = s < a.S && a.valid[s];
    float u[4] = {0.f, 0.f, 0.f, 0.f}, dh0 = 0.f, dh1 = 0.f;
    if (valid) {
#pragma unroll
      for (int c = 0; c < 4; ++c) u[c] = a.u4[4 * s + c];
      const float* dp = a.dh_pl + ((int64_t)l * a.S + s) * 2;
      dh0 = dp[0];
      dh1 = dp[1];
    }
    const Cell4 c = lotd4_cell(u, Rx, Ry, Rz, Rw);
    const int pmask = a.parity_slots ? ((c.c0[0] & 1) | ((c.c0[1] & 1) << 1) | ((c.c0[2] & 1) << 2) | ((c.c0[3] & 1) << 3)) : 0;
#pragma unroll
    for (int yzw = 0; yzw < 8; ++yzw) {
      uint32_t idx[2];
      float v0[2], v1[2];
      int emit[2];
#pragma unroll
      for (int dx = 0; dx < 2; ++dx) {
        // parity slots (see k_lotd_scatter, field.hip): slot (dx, yzw) holds the vertex with those coordinate parities, so the
        // vertices two consecutive shells' cells share sit in the same slot and the run merge below folds them
        const int corner = (dx | (yzw << 1)) ^ pmask;
        const float w = lotd4_weight(c, corner);
        idx[dx] = lotd4_index(c.c0[0] + (corner & 1), c.c0[1] + ((corner >> 1) & 1), c.c0[2] + ((corner >> 2) & 1), c.c0[3] + (corner >> 3), Rx,
                              Ry, Rz, a.lotd.type[l], a.lotd.size[l]);
        v0[dx] = w * dh0;
        v1[dx] = w * dh1;
        emit[dx] = valid;
        if (dedup) {
          const uint32_t key = valid ? idx[dx] : 0xffffffffu;
          const uint32_t pk = wave_shfl(key, lane - 1);
          const unsigned long long heads = wave_ballot(lane == 0 || pk != key);
          const unsigned long long below = heads & ((2ull << lane) - 1ull);
          const int run_start = 63 - __builtin_clzll(below);
#pragma unroll
          for (int d = 1; d < 64; d <<= 1) {
            // (wave-uniform early exit: no run of this slot reaches d lanes back -- at the fine levels runs are 1-3 lanes long
            // and two of the six rounds do all the work; -DNSIM_SCATTER_SCAN_EXIT=0 keeps all six)
            if (NSIM_SCATTER_SCAN_EXIT && !wave_ballot(lane - d >= run_start)) break;
            const float o0 = wave_shfl(v0[dx], lane - d), o1 = wave_shfl(v1[dx], lane - d);
            if (lane - d >= run_start) {
              v0[dx] += o0;
              v1[dx] += o1;
            }
          }
          emit[dx] = valid && (lane == 63 || ((heads >> (lane + 1)) & 1ull));   // last lane of the run
        }
      }
#define NSIM_QUAD4(I)                                                                                      \
  {                                                                                                        \
    const uint32_t i0 = quad_bcast<I>(idx[0]), i1 = quad_bcast<I>(idx[1]);                                 \
    const float a0 = quad_bcast<I>(v0[0]), a1 = quad_bcast<I>(v1[0]);                                      \
    const float b0 = quad_bcast<I>(v0[1]), b1 = quad_bcast<I>(v1[1]);                                      \
    const int e0 = quad_bcast<I>(emit[0]), e1 = quad_bcast<I>(emit[1]);                                    \
    const uint32_t ii = rq < 2 ? i0 : i1;                                                                  \
    const float vv = rq == 0 ? a0 : (rq == 1 ? a1 : (rq == 2 ? b0 : b1));                                  \
    const int ee = rq < 2 ? e0 : e1;                                                                       \
    if (ee) atomicAdd(base + 2 * (int64_t)ii + (rq & 1), vv);                                              \
  }
      if constexpr (CONSEC) {
        const uint32_t k0 = emit[0] ? idx[0] : 0xffffffffu, k1 = emit[1] ? idx[1] : 0xffffffffu;
#pragma unroll
        for (int I = 0; I < 4; ++I) {
          const int src = 16 * I + (lane >> 2);
          const uint32_t i0 = wave_shfl(k0, src), i1 = wave_shfl(k1, src);
          const float a0 = wave_shfl(v0[0], src), a1 = wave_shfl(v1[0], src);
          const float b0 = wave_shfl(v0[1], src), b1 = wave_shfl(v1[1], src);
          const uint32_t ii = rq < 2 ? i0 : i1;
          const float vv = rq == 0 ? a0 : (rq == 1 ? a1 : (rq == 2 ? b0 : b1));
          if (ii != 0xffffffffu) atomicAdd(base + 2 * (int64_t)ii + (rq & 1), vv);
        }
      } else {
        NSIM_QUAD4(0)
        NSIM_QUAD4(1)
        NSIM_QUAD4(2)
        NSIM_QUAD4(3)
      }
#undef NSIM_QUAD4
    }
  }
}

// ================================================================================== C ABI
static int nerf_meta_check(const NsimDistantMeta* m) {
  if (!m) return 20;
  const int rc = lotd4_meta_check(&m->lotd);
  if (rc) return rc;
  if (m->precision != 0 && m->precision != 1) return 23;
  return 0;
}

static NerfArgs nerf_args(const NsimDistantMeta* meta) {
  NerfArgs a = NerfArgs();
  a.lotd = lotd4_dev(&meta->lotd);
  a.lay = nerf_layout(meta->precision);
  a.F = 2 * meta->lotd.num_levels;
  return a;
}

static unsigned nerf_grid(int64_t S, int64_t cap) {
  const int64_t tiles = (S + 31) / 32;
  int64_t b = (tiles + NERF_WAVES - 1) / NERF_WAVES;
  b = b > cap ? cap : (b < 1 ? 1 : b);
  return (unsigned)b;
}

extern "C" {

int64_t nsim_distant_wpack_bytes(const NsimDistantMeta* meta) {
  if (nerf_meta_check(meta)) return -1;
  return nerf_layout(meta->precision).total;
}

int nsim_distant_pack_weights(const NsimDistantMeta* meta, const float* den_w, const float* den_b, const float* rad_w,
                              const float* rad_b, void* wpack, void* stream) {
  const int rc = nerf_meta_check(meta);
  if (rc) return rc;
  const NerfLayout L = nerf_layout(meta->precision);
  NerfDims dims;
  int64_t total = 0;
  for (int m = 0; m < N_MCOUNT; ++m) {
    dims.uo[m] = kNUo[m];
    dims.ui[m] = kNUi[m];
    total += (int64_t)kNUo[m] * kNUi[m];
  }
  total += (int64_t)NV_COUNT * 64;
  hipLaunchKernelGGL(k_nerf_pack, dim3(nsim_blocks(total, 256)), dim3(256), 0, (hipStream_t)stream, L, dims,
                     2 * meta->lotd.num_levels, den_w, den_b, rad_w, rad_b, (char*)wpack);
  NSIM_CHECK_LAUNCH();
  return 0;
}

int nsim_distant_shells(const float* rays_o, const float* rays_d, const float* near, const float* jitter, int64_t N,
                        int K, const float* aabb /* host [6]: min, max */, float r_min, float r_max, float* t,
                        float* u4, uint8_t* valid, void* stream) {
  if (N <= 0 || K <= 0) return 0;
  if (!aabb || !(r_min > 0.f) || !(r_max > r_min)) return 5;
  hipLaunchKernelGGL(k_distant_shells, dim3(nsim_blocks(N * K, 256)), dim3(256), 0, (hipStream_t)stream, rays_o, rays_d,
                     near, jitter, N, K, (aabb[0] + aabb[3]) * 0.5f, (aabb[1] + aabb[4]) * 0.5f, (aabb[2] + aabb[5]) * 0.5f,
                     (aabb[3] - aabb[0]) * 0.5f, (aabb[4] - aabb[1]) * 0.5f, (aabb[5] - aabb[2]) * 0.5f, r_min, r_max, t, u4,
                     valid);
  NSIM_CHECK_LAUNCH();
  return 0;
}

int nsim_density_alpha_fwd(const float* sigma, const float* t, const uint8_t* valid, int64_t N, int K,
                           int include_inf_distance, float* alpha, void* stream) {
  if (N <= 0 || K <= 0) return 0;
  hipLaunchKernelGGL(k_density_alpha_fwd, dim3(nsim_blocks(N * K, 256)), dim3(256), 0, (hipStream_t)stream, sigma, t,
                     valid, N, K, include_inf_distance, alpha);
  NSIM_CHECK_LAUNCH();
  return 0;
}

int nsim_density_alpha_bwd(const float* sigma, const float* t, const uint8_t* valid, const float* dalpha, int64_t N,
                           int K, int include_inf_distance, float* dsigma, void* stream) {
  if (N <= 0 || K <= 0) return 0;
  hipLaunchKernelGGL(k_density_alpha_bwd, dim3(nsim_blocks(N * K, 256)), dim3(256), 0, (hipStream_t)stream, sigma, t,
                     valid, dalpha, N, K, include_inf_distance, dsigma);
  NSIM_CHECK_LAUNCH();
  return 0;
}

int nsim_distant_fwd(const NsimDistantMeta* meta, const void* grid_f16, const void* wpack, const float* u4,
                     const float* rays_d, const float* h_appear, int64_t S, int K, float* sigma, float* rgb,
                     float* h_planes, void* stream) {
  const int rc = nerf_meta_check(meta);
  if (rc) return rc;
  if (S <= 0) return 0;
  if (!u4 || !rays_d || !sigma || !rgb || K <= 0) return 4;
  NerfArgs a = nerf_args(meta);
  a.grid = (const f16*)grid_f16;
  a.wpack = (const char*)wpack;
  a.u4 = u4; a.rays_d = rays_d; a.h_appear = h_appear;
  a.S = S; a.K = K;
  a.sigma = sigma; a.rgb = rgb; a.h_pl = h_planes;
  static const bool fused_gather = getenv("NSIM_DISTANT_FUSED_GATHER") && atoi(getenv("NSIM_DISTANT_FUSED_GATHER")) == 1;
  if (h_planes && !fused_gather) {        // training: the table reads as their own level-major launch, decoders on the planes
    Gather4Args g;
    g.lotd = a.lotd;
    g.grid = a.grid;
    g.u4 = u4;
    g.S = S;
    g.h_pl = h_planes;
    deal_levels4(&meta->lotd, g);
    const dim3 gg((unsigned)(8 * nsim_blocks(S, 64 * G4_PTS)));
    hipLaunchKernelGGL(k_lotd4_gather_lm, gg, dim3(64), 0, (hipStream_t)stream, g);
    NSIM_CHECK_LAUNCH();
    a.h_from_planes = 1;
  }
  const size_t shmem = meta->precision == 0 ? (size_t)((a.lay.total + 15) & ~15) : 0;
  // persistent workgroups, one per CU (decoder part of nsim_distant_fwd per 0.52 M shells: 0.125 ms at 1024, 0.113 at 256)
  static const int fwd_grid = getenv("NSIM_NERF_FWD_GRID") ? atoi(getenv("NSIM_NERF_FWD_GRID")) : 256;
  const dim3 grid(nerf_grid(S, fwd_grid)), block(64 * NERF_WAVES);
  if (meta->precision == 0) hipLaunchKernelGGL((k_nerf<0, 0>), grid, block, shmem, (hipStream_t)stream, a);
  else hipLaunchKernelGGL((k_nerf<1, 0>), grid, block, shmem, (hipStream_t)stream, a);
  NSIM_CHECK_LAUNCH();
  return 0;
}

int nsim_distant_bwd(const NsimDistantMeta* meta, const void* wpack, const float* h_planes, const float* sigma_fwd,
                     const float* rgb_fwd, const float* rays_d, const float* h_appear, const uint8_t* valid, int64_t S,
                     int K, const float* dsigma, const float* drgb, float* dh_planes, float* dden_w, float* dden_b,
                     float* drad_w, float* drad_b, float* dh_appear, void* stream) {
  const int rc = nerf_meta_check(meta);
  if (rc) return rc;
  if (S <= 0) return 0;
  if (!h_planes || !sigma_fwd || !rgb_fwd || !rays_d || !valid) return 28;
  if (!dden_w || !dden_b || !drad_w || !drad_b) return 26;
  NerfArgs a = nerf_args(meta);
  a.wpack = (const char*)wpack;
  a.rays_d = rays_d; a.h_appear = h_appear; a.valid = valid;
  a.S = S; a.K = K;
  a.h_pl = const_cast<float*>(h_planes);
  a.sigma_fwd = sigma_fwd; a.rgb_fwd = rgb_fwd;
  a.dsigma = dsigma; a.drgb = drgb;
  a.dh_pl = dh_planes;
  a.dden_w = dden_w; a.dden_b = dden_b; a.drad_w = drad_w; a.drad_b = drad_b; a.dh_appear = dh_appear;
  const size_t st = meta->precision == 0 ? stage_bytes_per_wave<0>() : stage_bytes_per_wave<1>();
  const size_t acc = (10700 * 4 + 15) & ~15;
  const bool priv = meta->precision == 0;
  const int nw = priv ? NERF_WAVES_PRIV : NERF_WAVES;
  const size_t vec_bytes = (size_t)((a.lay.total - a.lay.vec[0] + 15) & ~15);
  const size_t shmem = priv ? vec_bytes + nw * acc + nw * st : acc + nw * st;
  const int64_t tiles = (S + 31) / 32;
  int64_t nb = (tiles + nw - 1) / nw;
  nb = nb > 256 ? 256 : (nb < 1 ? 1 : nb);
  const dim3 grid((unsigned)nb), block(64 * nw);
  if (meta->precision == 0) hipLaunchKernelGGL((k_nerf<0, 1>), grid, block, shmem, (hipStream_t)stream, a);
  else hipLaunchKernelGGL((k_nerf<1, 1>), grid, block, shmem, (hipStream_t)stream, a);
  NSIM_CHECK_LAUNCH();
  return 0;
}

int nsim_lotd4_scatter(const NsimLotd4Meta* meta, const float* u4, const uint8_t* valid, int64_t S,
                       const float* dh_planes, float* dgrid, void* stream) {
  const int rc = lotd4_meta_check(meta);
  if (rc) return rc;
  if (S <= 0) return 0;
  if (!u4 || !valid || !dh_planes || !dgrid) return 28;
  Scatter4Args sa;
  sa.lotd = lotd4_dev(meta);
  sa.u4 = u4; sa.valid = valid; sa.S = S; sa.dh_pl = dh_planes; sa.dgrid = dgrid;
  sa.dedup_max_rw = 1 << 30;      // every level: 3.62 -> 1.73 ms per 0.52 M shell points (levels with Rw <= 16 / 64 only: 2.07 / 1.76)
  if (const char* e = getenv("NSIM_DEDUP4_MAX_RW")) sa.dedup_max_rw = atoi(e);
  const char* ep = getenv("NSIM_SCATTER_PARITY");
  sa.parity_slots = !(ep && atoi(ep) == 0);
  const dim3 grid(nsim_blocks((S + 63) / 64, 4, 4096), meta->num_levels);
  const char* eg = getenv("NSIM_SCATTER_GROUP");
  if (!(eg && atoi(eg) == 0))
    hipLaunchKernelGGL(k_lotd4_scatter<true>, grid, dim3(256), 0, (hipStream_t)stream, sa);
  else
    hipLaunchKernelGGL(k_lotd4_scatter<false>, grid, dim3(256), 0, (hipStream_t)stream, sa);
  NSIM_CHECK_LAUNCH();
  return 0;
}

}  // extern "C"
