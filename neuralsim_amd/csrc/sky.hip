// sky.hip -- the directional sky MLP of the street configs on the gfx950 matrix cores.
//
// Replaces ``SimpleSky.forward`` (app/models/env/sky.py:16-51) as configured by
// code_single/configs/waymo/streetsurf/withmask_withlidar_joint.240219.yaml:312-322:
//   sinusoidal embedding of the view direction (n_frequencies F = 10, input included: 3 + 6F dims)
//   ++ appearance embedding (A = 4)  ->  256 -> 256 -> 3, ReLU, sigmoid output (D = 2, W = 256).
// Call site: app/renderers/single_volume_renderer.py:449-457 (one query per RAY, blended with 1 - mask_volume).
//
// One wave = 32 rays in the activation-register convention of mfma_mlp.h (layers computed transposed, a lane's
// accumulators are its B fragment of the next layer); the 256-wide layers are 8 M-tiles.  Weights are read as
// pre-packed A fragments straight from L2 (the 256x256 matrix does not fit LDS next to its transpose, and with one
// query per ray there is no reuse to win).  The backward keeps the per-ray part (deltas) in one kernel that writes
// unit-major planes [unit][ray] and contracts the weight gradients over rays in a second, GEMM-shaped kernel (tile x
// ray-chunk grid, f32 atomics of whole 32x32 tiles) -- so no per-ray atomics on a 64 K-entry matrix.
#include <math.h>
#include <string.h>

#include "../../include/nsim.h"
#include "mfma_mlp.h"

#define SKY_W 256
#define SKY_IN 96            // padded input units (3 M-tiles)
#define SKY_MW (SKY_W / 32)
#define SKY_MI (SKY_IN / 32)
#define SKY_CHUNK 128        // rays per weight-gradient work item; plane pitch is a multiple of this

enum { SM_W1 = 0, SM_W2, SM_W3, SM_W3T, SM_W2T, SM_W1T, SM_COUNT };
enum { SV_B1 = 0, SV_B2, SV_B3, SV_COUNT };

struct SkyLayout {
  int elt;                   // 2: f16 A fragments, 4: f32
  int64_t mat[SM_COUNT];
  int64_t vec[SV_COUNT];
  int64_t total;
};

static inline void sky_dims(int m, int& uo, int& ui) {
  switch (m) {
    case SM_W1: uo = SKY_W; ui = SKY_IN; break;
    case SM_W2: uo = SKY_W; ui = SKY_W; break;
    case SM_W3: uo = 32; ui = SKY_W; break;
    case SM_W3T: uo = SKY_W; ui = 32; break;
    case SM_W2T: uo = SKY_W; ui = SKY_W; break;
    default: uo = SKY_IN; ui = SKY_W; break;   // SM_W1T
  }
}

static inline SkyLayout sky_layout(int precision) {
  SkyLayout L;
  L.elt = precision == 0 ? 2 : 4;
  int64_t off = 0;
  for (int m = 0; m < SM_COUNT; ++m) {
    int uo, ui;
    sky_dims(m, uo, ui);
    L.mat[m] = off;
    off += (int64_t)uo * ui * L.elt;
    off = (off + 255) & ~(int64_t)255;
  }
  const int vlen[SV_COUNT] = {2 * SKY_MW * 16, 2 * SKY_MW * 16, 2 * 16};
  for (int v = 0; v < SV_COUNT; ++v) {
    L.vec[v] = off;
    off += (int64_t)vlen[v] * 4;
    off = (off + 255) & ~(int64_t)255;
  }
  L.total = off;
  return L;
}

struct SkyDims {
  int uo[SM_COUNT], ui[SM_COUNT];
};

// flat weights in the reference's layer order: w = [W1 (256 x IN), W2 (256 x 256), W3 (3 x 256)], b = [256, 256, 3]
__device__ __forceinline__ float sky_src(int mat, int row, int col, int IN, const float* w) {
  const int64_t o1 = 0, o2 = (int64_t)SKY_W * IN, o3 = o2 + (int64_t)SKY_W * SKY_W;
  switch (mat) {
    case SM_W1: return col < IN ? w[o1 + (int64_t)row * IN + col] : 0.f;
    case SM_W2: return w[o2 + (int64_t)row * SKY_W + col];
    case SM_W3: return row < 3 ? w[o3 + (int64_t)row * SKY_W + col] : 0.f;
    case SM_W3T: return col < 3 ? w[o3 + (int64_t)col * SKY_W + row] : 0.f;
    case SM_W2T: return w[o2 + (int64_t)col * SKY_W + row];
    default: return row < IN ? w[o1 + (int64_t)col * IN + row] : 0.f;
  }
}

__global__ void __launch_bounds__(256) k_sky_pack(SkyLayout L, SkyDims dims, int IN, const float* __restrict__ w,
                                                   const float* __restrict__ b, char* __restrict__ wpack) {
  const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int64_t base = 0;
  for (int m = 0; m < SM_COUNT; ++m) {
    const int Uo = dims.uo[m], Ui = dims.ui[m];
    const int64_t cnt = (int64_t)Uo * Ui;
    if (tid >= base && tid < base + cnt) {
      const int64_t k = tid - base;
      int row, col;
      if (L.elt == 2) {   // f16x8 per lane per K-step: [(mo * nS + s) * 64 + lane][e]
        const int e = (int)(k & 7), lane = (int)((k >> 3) & 63);
        const int fs = (int)(k >> 9);
        const int nS = Ui / 16;
        const int mo = fs / nS, s = fs % nS;
        row = 32 * mo + (lane & 31);
        col = 16 * s + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
        ((f16*)(wpack + L.mat[m]))[k] = (f16)sky_src(m, row, col, IN, w);
      } else {            // one float per lane per (mi, r): [((mo * nMi + mi) * 16 + r) * 64 + lane]
        const int lane = (int)(k & 63);
        const int fr = (int)(k >> 6);
        const int r = fr & 15, fm = fr >> 4;
        const int nMi = Ui / 32;
        const int mo = fm / nMi, mi = fm % nMi;
        row = 32 * mo + (lane & 31);
        col = unit_of(mi, r, lane >> 5);
        ((float*)(wpack + L.mat[m]))[k] = sky_src(m, row, col, IN, w);
      }
      return;
    }
    base += cnt;
  }
  // biases in per-lane order [hi][m * 16 + r]
  int64_t vt = tid - base;
  if (vt >= 0 && vt < 2 * 2 * SKY_MW * 16) {
    const int v = (int)(vt / (2 * SKY_MW * 16)), k = (int)(vt % (2 * SKY_MW * 16));
    const int hi = k / (SKY_MW * 16), q = k % (SKY_MW * 16);
    ((float*)(wpack + L.vec[v]))[k] = b[v * SKY_W + unit_of(q >> 4, q & 15, hi)];
    return;
  }
  vt -= 2 * 2 * SKY_MW * 16;
  if (vt >= 0 && vt < 32) {
    const int hi = (int)(vt >> 4), r = (int)(vt & 15);
    const int u = unit_of(0, r, hi);
    ((float*)(wpack + L.vec[SV_B3]))[vt] = u < 3 ? b[2 * SKY_W + u] : 0.f;
  }
}

struct SkyArgs {
  SkyLayout L;
  const char* wpack;
  int F, A, IN;
  int64_t N, Np;
  const float* v;         // [N,3] unit view directions
  const float* ha;        // [N,A] or NULL
  float* rgb;             // fwd: out [N,3]; bwd: the saved forward output
  const float* drgb;      // [N,3]
  float* emb_pl;          // [SKY_IN][Np]
  float* a1_pl;           // [SKY_W][Np]
  float* a2_pl;           // [SKY_W][Np]
  float* d3_pl;           // [32][Np]
  float* d2_pl;           // [SKY_W][Np]
  float* d1_pl;           // [SKY_W][Np]
  float* dha;             // [N,A] or NULL
  float* dw;              // flat, as w
  float* db;              // flat, as b
};

// input unit u of the network: [v (3), {sin(2^f v), cos(2^f v)}_{f < F} (6F), h_appear (A), 0 ...]
__device__ __forceinline__ float sky_input(int u, float vx, float vy, float vz, const float* ha, int F, int A) {
  if (u < 3) return u == 0 ? vx : (u == 1 ? vy : vz);
  int k = u - 3;
  if (k < 6 * F) {
    const int f = k / 6, c = k % 6, ax = c % 3;
    const float x = (ax == 0 ? vx : (ax == 1 ? vy : vz)) * (float)(1 << f);
    return c < 3 ? sinf(x) : cosf(x);
  }
  k -= 6 * F;
  return (ha && k < A) ? ha[k] : 0.f;
}

// Plane element of (unit U(m,r,hi), ray pt) = row base (wave-uniform: unit U(m,r,0)) + a 32-bit per-lane offset
// loff = 4 hi Np + pt: keeps the 128 addresses of a 256-unit activation out of the vector registers.
__device__ __forceinline__ const float* plane_row(const float* pl, int64_t Np, int m, int r) {
  return pl + (int64_t)(32 * m + (r & 3) + 8 * (r >> 2)) * Np;
}
template <int NM>
__device__ __forceinline__ void plane_store(float* pl, int64_t Np, int loff, const float (&v)[NM * 16]) {
#pragma unroll
  for (int m = 0; m < NM; ++m)
#pragma unroll
    for (int r = 0; r < 16; ++r) const_cast<float*>(plane_row(pl, Np, m, r))[loff] = v[m * 16 + r];
}

template <int PREC>
__global__ void __launch_bounds__(64) k_sky_fwd(SkyArgs a) {
  const int lane = nsim_lane(), j = lane & 31, hi = lane >> 5;
  const int64_t pt = (int64_t)blockIdx.x * 32 + j;      // < Np by construction of the grid
  const int loff = (int)(4 * hi * a.Np + pt);
  const bool ok = pt < a.N;
  float vx = 0.f, vy = 0.f, vz = 0.f;
  if (ok) {
    vx = a.v[3 * pt];
    vy = a.v[3 * pt + 1];
    vz = a.v[3 * pt + 2];
  }
  const float* ha = (ok && a.ha) ? a.ha + pt * a.A : nullptr;
  float in[SKY_MI * 16];
#pragma unroll
  for (int m = 0; m < SKY_MI; ++m)
#pragma unroll
    for (int r = 0; r < 16; ++r) in[m * 16 + r] = ok ? sky_input(unit_of(m, r, hi), vx, vy, vz, ha, a.F, a.A) : 0.f;
  if (a.emb_pl) plane_store<SKY_MI>(a.emb_pl, a.Np, loff, in);

  const float* b1 = reinterpret_cast<const float*>(a.wpack + a.L.vec[SV_B1]) + hi * SKY_MW * 16;
  const float* b2 = reinterpret_cast<const float*>(a.wpack + a.L.vec[SV_B2]) + hi * SKY_MW * 16;
  const float* b3 = reinterpret_cast<const float*>(a.wpack + a.L.vec[SV_B3]) + hi * 16;
  float h1[SKY_MW * 16];
  dense<PREC, SKY_MW, SKY_MI>(h1, a.wpack + a.L.mat[SM_W1], in, false);
#pragma unroll
  for (int q = 0; q < SKY_MW * 16; ++q) h1[q] = fmaxf(h1[q] + b1[q], 0.f);
  if (a.a1_pl) plane_store<SKY_MW>(a.a1_pl, a.Np, loff, h1);
  float h2[SKY_MW * 16];
  dense<PREC, SKY_MW, SKY_MW>(h2, a.wpack + a.L.mat[SM_W2], h1, false);
#pragma unroll
  for (int q = 0; q < SKY_MW * 16; ++q) h2[q] = fmaxf(h2[q] + b2[q], 0.f);
  if (a.a2_pl) plane_store<SKY_MW>(a.a2_pl, a.Np, loff, h2);
  float o[16];
  dense<PREC, 1, SKY_MW>(o, a.wpack + a.L.mat[SM_W3], h2, false);
  if (ok && hi == 0) {
#pragma unroll
    for (int r = 0; r < 3; ++r) a.rgb[3 * pt + r] = 1.0f / (1.0f + expf(-(o[r] + b3[r])));
  }
}

// out = relu'(act) .* (W^T-packed matrix . in): a 256-unit layer of the backward, by halves
template <int PREC, int NI>
__device__ __forceinline__ void sky_bwd_layer(float (&out)[SKY_MW * 16], const char* wmat, const float (&in)[NI * 16],
                                              const float* act_pl, int64_t Np, int loff) {
  constexpr int64_t half_bytes = PREC == 0 ? (int64_t)4 * 2 * NI * 64 * 16 : (int64_t)4 * NI * 16 * 64 * 4;
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    float part[4 * 16];
    dense<PREC, 4, NI>(part, wmat + h * half_bytes, in, true);
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const bool on = plane_row(act_pl, Np, 4 * h + m, r)[loff] > 0.f;
        out[(4 * h + m) * 16 + r] = on ? part[m * 16 + r] : 0.f;
      }
  }
}

template <int PREC>
__global__ void __launch_bounds__(64) k_sky_bwd(SkyArgs a) {
  const int lane = nsim_lane(), j = lane & 31, hi = lane >> 5;
  const int64_t pt = (int64_t)blockIdx.x * 32 + j;
  const int loff = (int)(4 * hi * a.Np + pt);
  const bool ok = pt < a.N;
  float d3[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) d3[r] = 0.f;
  if (ok && hi == 0) {
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      const float y = a.rgb[3 * pt + r];
      d3[r] = a.drgb[3 * pt + r] * y * (1.0f - y);
    }
  }
  plane_store<1>(a.d3_pl, a.Np, loff, d3);
  // 256-wide layers in two halves of 4 M-tiles: bounds the live set (inputs + half the outputs + the ReLU masks)
  float d2[SKY_MW * 16];
  sky_bwd_layer<PREC, 1>(d2, a.wpack + a.L.mat[SM_W3T], d3, a.a2_pl, a.Np, loff);
  plane_store<SKY_MW>(a.d2_pl, a.Np, loff, d2);
  float d1[SKY_MW * 16];
  sky_bwd_layer<PREC, SKY_MW>(d1, a.wpack + a.L.mat[SM_W2T], d2, a.a1_pl, a.Np, loff);
  plane_store<SKY_MW>(a.d1_pl, a.Np, loff, d1);
  if (a.dha && a.A > 0) {
    float din[SKY_MI * 16];
    dense<PREC, SKY_MI, SKY_MW>(din, a.wpack + a.L.mat[SM_W1T], d1, true);
    if (ok) {
#pragma unroll
      for (int m = 0; m < SKY_MI; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int k = unit_of(m, r, hi) - 3 - 6 * a.F;
          if (k >= 0 && k < a.A) a.dha[pt * a.A + k] = din[m * 16 + r];
        }
    }
  }
}

// dW[rows x cols] (leading dimension ld) += A[rows..][rays] . B[cols..][rays]^T over one chunk of SKY_CHUNK rays;
// db[rows] += row sums of A (done by the no == 0 tiles).  A, B are unit-major planes of pitch Np.
template <int PREC>
__device__ __forceinline__ void sky_dw_tile(const float* __restrict__ A, const float* __restrict__ B, int64_t Np,
                                            int64_t p0, int mo, int no, int rows, int cols, int ld, float* dW,
                                            float* db) {
  const int lane = nsim_lane(), i = lane & 31, hi = lane >> 5;
  const float* ap = A + (int64_t)(32 * mo + i) * Np + p0;
  const float* bp = B + (int64_t)(32 * no + i) * Np + p0;
  f32x16 acc = zero16();
  float rowsum = 0.f;
  float unscale = 1.0f;
  if constexpr (PREC == 0) {
    constexpr int NS = SKY_CHUNK / 16;
    float av[NS * 8], bv[NS * 8];
#pragma unroll
    for (int s = 0; s < NS; ++s)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        av[s * 8 + e] = ap[16 * s + 8 * hi + e];
        bv[s * 8 + e] = bp[16 * s + 8 * hi + e];
      }
    const float sa = dyn_scale<0, NS * 8>(av), sb = dyn_scale<0, NS * 8>(bv);
    unscale = 1.0f / (sa * sb);
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      f16x8 a8, b8;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        a8[e] = (f16)(av[s * 8 + e] * sa);
        b8[e] = (f16)(bv[s * 8 + e] * sb);
        rowsum += av[s * 8 + e];
      }
      acc = mfma_32x32x16_f16(a8, b8, acc);
    }
  } else {
#pragma unroll 8
    for (int q = 0; q < SKY_CHUNK / 2; ++q) {
      const float x = ap[2 * q + hi], y = bp[2 * q + hi];
      rowsum += x;
      acc = mfma_32x32x2_f32(x, y, acc);
    }
  }
  dw_flush(dW, ld, rows, cols, mo, no, acc, unscale);
  if (db && no == 0) {
    rowsum += wave_shfl_xor(rowsum, 32);
    if (hi == 0 && 32 * mo + i < rows && rowsum != 0.f) atomicAdd(&db[32 * mo + i], rowsum);
  }
}

// grid: x = weight tile (24 of dW1, 64 of dW2, 8 of dW3), y = ray chunk
template <int PREC>
__global__ void __launch_bounds__(64) k_sky_dw(SkyArgs a) {
  const int tile = blockIdx.x;
  const int64_t p0 = (int64_t)blockIdx.y * SKY_CHUNK;
  const int64_t o2 = (int64_t)SKY_W * a.IN, o3 = o2 + (int64_t)SKY_W * SKY_W;
  if (tile < SKY_MW * SKY_MI) {
    sky_dw_tile<PREC>(a.d1_pl, a.emb_pl, a.Np, p0, tile / SKY_MI, tile % SKY_MI, SKY_W, a.IN, a.IN, a.dw, a.db);
  } else if (tile < SKY_MW * SKY_MI + SKY_MW * SKY_MW) {
    const int t = tile - SKY_MW * SKY_MI;
    sky_dw_tile<PREC>(a.d2_pl, a.a1_pl, a.Np, p0, t / SKY_MW, t % SKY_MW, SKY_W, SKY_W, SKY_W, a.dw + o2, a.db + SKY_W);
  } else {
    const int t = tile - SKY_MW * SKY_MI - SKY_MW * SKY_MW;
    sky_dw_tile<PREC>(a.d3_pl, a.a2_pl, a.Np, p0, 0, t, 3, SKY_W, SKY_W, a.dw + o3, a.db + 2 * SKY_W);
  }
}

static inline int sky_check(const NsimSkyMeta* m) {
  if (!m) return 30;
  if (m->n_frequencies < 0 || m->n_appear < 0 || 3 + 6 * m->n_frequencies + m->n_appear > SKY_IN) return 31;
  if (m->precision != 0 && m->precision != 1) return 23;
  return 0;
}

static inline int64_t sky_pitch(int64_t N) { return (N + SKY_CHUNK - 1) / SKY_CHUNK * SKY_CHUNK; }

static inline SkyArgs sky_args(const NsimSkyMeta* m, const void* wpack, int64_t N) {
  SkyArgs a = SkyArgs();
  a.L = sky_layout(m->precision);
  a.wpack = (const char*)wpack;
  a.F = m->n_frequencies;
  a.A = m->n_appear;
  a.IN = 3 + 6 * a.F + a.A;
  a.N = N;
  a.Np = sky_pitch(N);
  return a;
}

extern "C" {

int64_t nsim_sky_wpack_bytes(const NsimSkyMeta* meta) {
  if (sky_check(meta)) return -1;
  return sky_layout(meta->precision).total;
}

int64_t nsim_sky_plane_pitch(int64_t N) { return N < 0 ? -1 : sky_pitch(N); }

int nsim_sky_pack_weights(const NsimSkyMeta* meta, const float* w, const float* b, void* wpack, void* stream) {
  if (int rc = sky_check(meta)) return rc;
  if (!w || !b || !wpack) return 4;
  const SkyLayout L = sky_layout(meta->precision);
  SkyDims dims;
  int64_t cnt = 0;
  for (int m = 0; m < SM_COUNT; ++m) {
    sky_dims(m, dims.uo[m], dims.ui[m]);
    cnt += (int64_t)dims.uo[m] * dims.ui[m];
  }
  cnt += 2 * 2 * SKY_MW * 16 + 32;
  hipLaunchKernelGGL(k_sky_pack, dim3(nsim_blocks(cnt, 256)), dim3(256), 0, (hipStream_t)stream, L, dims,
                     3 + 6 * meta->n_frequencies + meta->n_appear, w, b, (char*)wpack);
  NSIM_CHECK_LAUNCH();
  return 0;
}

int nsim_sky_fwd(const NsimSkyMeta* meta, const void* wpack, const float* v, const float* h_appear, int64_t N,
                 float* rgb, float* planes_fwd, void* stream) {
  if (int rc = sky_check(meta)) return rc;
  if (N < 0 || N > (1 << 24)) return 2;
  if (N == 0) return 0;
  if (!wpack || !v || !rgb) return 4;
  if (meta->n_appear > 0 && !h_appear) return 32;
  SkyArgs a = sky_args(meta, wpack, N);
  a.v = v;
  a.ha = meta->n_appear > 0 ? h_appear : nullptr;
  a.rgb = rgb;
  if (planes_fwd) {
    a.emb_pl = planes_fwd;
    a.a1_pl = planes_fwd + (int64_t)SKY_IN * a.Np;
    a.a2_pl = a.a1_pl + (int64_t)SKY_W * a.Np;
  }
  const dim3 grid((unsigned)(a.Np / 32));
  if (meta->precision == 0) hipLaunchKernelGGL(k_sky_fwd<0>, grid, dim3(64), 0, (hipStream_t)stream, a);
  else hipLaunchKernelGGL(k_sky_fwd<1>, grid, dim3(64), 0, (hipStream_t)stream, a);
  NSIM_CHECK_LAUNCH();
  return 0;
}

int nsim_sky_bwd(const NsimSkyMeta* meta, const void* wpack, const float* rgb_fwd, const float* drgb, int64_t N,
                 const float* planes_fwd, float* planes_bwd, float* dw, float* db, float* dh_appear, void* stream) {
  if (int rc = sky_check(meta)) return rc;
  if (N < 0 || N > (1 << 24)) return 2;
  if (N == 0) return 0;
  if (!wpack || !rgb_fwd || !drgb || !planes_fwd || !planes_bwd) return 4;
  if (!dw || !db) return 26;
  SkyArgs a = sky_args(meta, wpack, N);
  a.rgb = const_cast<float*>(rgb_fwd);
  a.drgb = drgb;
  a.emb_pl = const_cast<float*>(planes_fwd);
  a.a1_pl = a.emb_pl + (int64_t)SKY_IN * a.Np;
  a.a2_pl = a.a1_pl + (int64_t)SKY_W * a.Np;
  a.d3_pl = planes_bwd;
  a.d2_pl = planes_bwd + (int64_t)32 * a.Np;
  a.d1_pl = a.d2_pl + (int64_t)SKY_W * a.Np;
  a.dha = meta->n_appear > 0 ? dh_appear : nullptr;
  a.dw = dw;
  a.db = db;
  const dim3 grid((unsigned)(a.Np / 32));
  const dim3 gdw(SKY_MW * SKY_MI + SKY_MW * SKY_MW + SKY_MW, (unsigned)(a.Np / SKY_CHUNK));
  if (meta->precision == 0) {
    hipLaunchKernelGGL(k_sky_bwd<0>, grid, dim3(64), 0, (hipStream_t)stream, a);
    hipLaunchKernelGGL(k_sky_dw<0>, gdw, dim3(64), 0, (hipStream_t)stream, a);
  } else {
    hipLaunchKernelGGL(k_sky_bwd<1>, grid, dim3(64), 0, (hipStream_t)stream, a);
    hipLaunchKernelGGL(k_sky_dw<1>, gdw, dim3(64), 0, (hipStream_t)stream, a);
  }
  NSIM_CHECK_LAUNCH();
  return 0;
}

}  // extern "C"
