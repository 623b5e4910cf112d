// wide_field.hip -- the NO-GRAD query of the SDF decoder with an embedded-position input block, f32 on the VALU, gfx950.
//
// ``surface_cfg.extra_pos_embed_cfg{type: sinusoidal_legacy, n_frequencies: 6}`` of the StyleLoTD Vehicle block
// (code_multi/configs/exps/fg_neus=hyper_lotd/no_fg_occ.221218.yaml:319-321): the decoder input is
// [grown features (32) | x, sin(2^k x), cos(2^k x), k = 0..5 (39)] = 71 values.  The with-grad forward and the backward of such
// a model run on the matrix cores (csrc/field.hip: k_field / k_field_bwd_j with NE = 2 embedded-position chunks); this file
// is the sampling pass's and the occupancy refresh's SDF query, which places samples and therefore runs in f32 whatever the
// model's precision (the role the split-precision k_field_sdf has for models without the block):
//
//   * a wave works on 4 points at a time, lane j = hidden unit j (width 64); the weights live in LDS with ODD row pitches
//     (W1: 129, W2: 65 floats), so "lane j walks row j" is bank-conflict free and every weight read serves all points of
//     the group; a layer input is staged once per wave and read back as 16-byte LDS broadcasts;
//   * f32 throughout (explicit FMAs), 8 waves per workgroup (two per SIMD hide the LDS latency of the weight walks);
//   * features come from the f32 level-major planes of nsim_lotd_gather_lm, the embedded position is generated in the kernel.
//
// Conventions fixed here (nr3d_lib is absent): the embedding sees the AABB-normalised position x_n = 2 u - 1 in [-1, 1]
// (u = the pyramid's unit coordinate), order [x_n (3) | for k: sin(2^k x_n) (3), cos(2^k x_n) (3)], no factor pi.
#include "nsim_common.h"

#define WIDE_WAVES 8      // waves per workgroup
#define WIDE_P 4          // points per wave and pass over the weights
#define WIDE_P1 129       // LDS pitch of a W1 row (<= 128 inputs), odd
#define WIDE_P2 65        // LDS pitch of a 64-wide row, odd
#define WIDE_LOG2E 1.4426950408889634f
#define WIDE_LN2 0.6931471805599453f

struct WideArgs {
  NsimLotdMeta lotd;
  int sdf_D, n_freq, F1, FIN;
  float beta;                       // < 0: relu
  const float *sdf_w, *sdf_b;
  const float *x, *rays_o, *rays_d, *t;
  const int64_t* ridx;
  int64_t S, PS;
  const int64_t* S_dev;
  int64_t S_add;
  const float* h_pl;                // f32 planes [NLP][PS][2]
  float* sdf;
};

__device__ __forceinline__ float wide_act(float z, float beta) {
  if (beta < 0.f) return fmaxf(z, 0.f);      // relu
  const float tt = nsim_exp2(-fabsf(z) * (beta * WIDE_LOG2E));      // exp(-beta |z|)
  return fmaxf(z, 0.f) + nsim_log2(1.0f + tt) * (WIDE_LN2 / beta);
}

__device__ __forceinline__ int64_t wide_valid_count(const WideArgs& a) {
  if (!a.S_dev) return a.S;
  const int64_t n = a.S_dev[0] + a.S_add;
  return n > a.S ? 0 : n;
}

// position of point s
__device__ __forceinline__ void wide_point(const WideArgs& a, int64_t s, float (&xx)[3]) {
  if (a.x) {
    xx[0] = a.x[3 * s]; xx[1] = a.x[3 * s + 1]; xx[2] = a.x[3 * s + 2];
  } else {
    const int64_t ray = a.ridx[s];
    const float tt = a.t[s];
#pragma unroll
    for (int c = 0; c < 3; ++c) xx[c] = a.rays_o[3 * ray + c] + tt * a.rays_d[3 * ray + c];
  }
}

// embedded-position input i (F1 <= i < FIN) of the first layer; zero for the feature block and past the end (branch-free: selects)
__device__ __forceinline__ float wide_embed(const WideArgs& a, const float (&xx)[3], int i) {
  const int m = i - a.F1;
  const bool on = m >= 0 && i < a.FIN;
  const int mm = on ? m : 0;
  const int c = mm < 3 ? mm : (mm - 3) % 3;
  // (selects, not indexed private arrays: a lane-dependent index would put xx into scratch)
  const bool unit_cube = a.lotd.x_scale[0] == 0.f && a.lotd.x_scale[1] == 0.f && a.lotd.x_scale[2] == 0.f;
  const float xc = c == 0 ? xx[0] : (c == 1 ? xx[1] : xx[2]);
  const float sc = unit_cube ? 0.5f : (c == 0 ? a.lotd.x_scale[0] : (c == 1 ? a.lotd.x_scale[1] : a.lotd.x_scale[2]));
  const float sh = unit_cube ? 0.5f : (c == 0 ? a.lotd.x_shift[0] : (c == 1 ? a.lotd.x_shift[1] : a.lotd.x_shift[2]));
  const float xn = 2.0f * (xc * sc + sh) - 1.0f;
  const int k = mm < 3 ? 0 : (mm - 3) / 6;
  const bool is_cos = mm >= 3 && ((mm - 3) % 6) >= 3;
  const float ang = (float)(1 << k) * xn;
  float val = is_cos ? nsim_cos(ang) : nsim_sin(ang);
  if (mm < 3) val = xn;
  return on ? val : 0.f;
}

struct WideLds {
  float *W1, *W2, *b1, *b2, *wh;
  float bh;
};

// stage the SDF decoder's weights (flat f32: W1 [64 x FIN], (W2 [64 x 64]), wh [64]; b1 [64], (b2 [64]), bh) into LDS;
// W1 rows are zero-padded to 128
__device__ __forceinline__ WideLds wide_stage(float* base, const WideArgs& a, float*& next) {
  WideLds L;
  L.W1 = base;
  L.W2 = L.W1 + 64 * WIDE_P1;
  L.b1 = L.W2 + 64 * WIDE_P2;
  L.b2 = L.b1 + 64;
  L.wh = L.b2 + 64;
  next = L.wh + 64;
  const int FIN = a.FIN;
  for (int i = threadIdx.x; i < 64 * 128; i += blockDim.x) {
    const int r = i >> 7, c = i & 127;
    L.W1[r * WIDE_P1 + c] = c < FIN ? a.sdf_w[r * FIN + c] : 0.f;
  }
  const float* w2 = a.sdf_w + 64 * FIN;
  if (a.sdf_D == 2)
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) L.W2[(i >> 6) * WIDE_P2 + (i & 63)] = w2[i];
  const float* wh = w2 + (a.sdf_D == 2 ? 4096 : 0);
  for (int i = threadIdx.x; i < 64; i += blockDim.x) {
    L.b1[i] = a.sdf_b[i];
    L.b2[i] = a.sdf_D == 2 ? a.sdf_b[64 + i] : 0.f;
    L.wh[i] = wh[i];
  }
  L.bh = a.sdf_b[a.sdf_D == 2 ? 128 : 64];
  return L;
}

// per-wave, per-point staging vectors (floats; every vector 16-byte aligned)
#define WS_IN 0          // [128] layer input
#define WS_A 128         // [64] a1
#define WS_PT 192

struct __attribute__((aligned(16))) float4w {
  float x, y, z, w;
};
__device__ __forceinline__ float4w wide_ld4(const float* p) { return *reinterpret_cast<const float4w*>(p); }
// acc + w . v with fused multiply-adds (this translation unit is built with -ffp-contract=off: the fusion is spelled out)
__device__ __forceinline__ float wide_dot4(float acc, float w0, float w1, float w2, float w3, const float4w& v) {
  return fmaf(w3, v.w, fmaf(w2, v.z, fmaf(w1, v.y, fmaf(w0, v.x, acc))));
}

// WIDE_P points of a wave: lane j = hidden unit j; rows of W1, then rows of W2, read once for all of them.
// FINP: the first-layer width rounded up (compile-time trip count of the W1 walk)
template <int FINP>
__device__ __forceinline__ void wide_points(const WideArgs& a, const WideLds& L, float* ws, const int64_t (&sp)[WIDE_P], int64_t Sv,
                                            int lane) {
  constexpr int P = WIDE_P;
  bool ok[P];
  float xx[P][3];
  // ---- inputs.  Every global load of the group is issued before the first one is waited for (branch-free addresses: a lane
  // outside the feature block, or a point past the end, reads element 0 of its array and drops the value) -- one memory
  // latency per group instead of one per load.
  const int na = a.lotd.n_active_levels;
  const int lv_on = (na > 0 && na < a.lotd.num_levels) ? na : a.lotd.num_levels;
  const bool f0 = lane < a.F1 && (lane >> 1) < lv_on, f1 = lane + 64 < a.F1 && ((lane + 64) >> 1) < lv_on;
  float h0[P], h1[P];
#pragma unroll
  for (int p = 0; p < P; ++p) {
    ok[p] = sp[p] < Sv;
    const int64_t s = ok[p] ? sp[p] : 0;
    const int64_t o0 = f0 ? ((int64_t)(lane >> 1) * a.PS + s) * 2 + (lane & 1) : 0;
    const int64_t o1 = f1 ? ((int64_t)((lane + 64) >> 1) * a.PS + s) * 2 + (lane & 1) : 0;
    h0[p] = a.h_pl[o0];
    h1[p] = a.h_pl[o1];
    wide_point(a, s, xx[p]);
  }
#pragma unroll
  for (int p = 0; p < P; ++p) {
    const float e0 = wide_embed(a, xx[p], lane), e1 = wide_embed(a, xx[p], lane + 64);
    const float m = ok[p] ? 1.f : 0.f;
    float* w = ws + p * WS_PT;
    w[WS_IN + lane] = (f0 ? h0[p] : e0) * m;
    w[WS_IN + 64 + lane] = (f1 ? h1[p] : e1) * m;
  }
  wave_sync_lds();
  // ---- rows of W1
  float z1[P];
#pragma unroll
  for (int p = 0; p < P; ++p) z1[p] = L.b1[lane];
  {
    const float* w1 = L.W1 + lane * WIDE_P1;
#pragma unroll 4
    for (int i = 0; i < FINP; i += 4) {
      const float w0 = w1[i], wa = w1[i + 1], wb = w1[i + 2], wc = w1[i + 3];
#pragma unroll
      for (int p = 0; p < P; ++p) z1[p] = wide_dot4(z1[p], w0, wa, wb, wc, wide_ld4(ws + p * WS_PT + WS_IN + i));
    }
  }
  float act[P];
#pragma unroll
  for (int p = 0; p < P; ++p) act[p] = wide_act(z1[p], a.beta);
  if (a.sdf_D == 2) {
    // ---- rows of W2
#pragma unroll
    for (int p = 0; p < P; ++p) ws[p * WS_PT + WS_A + lane] = act[p];
    wave_sync_lds();
    float z2[P];
#pragma unroll
    for (int p = 0; p < P; ++p) z2[p] = L.b2[lane];
    const float* w2 = L.W2 + lane * WIDE_P2;
#pragma unroll 4
    for (int k = 0; k < 64; k += 4) {
      const float w0 = w2[k], wa = w2[k + 1], wb = w2[k + 2], wc = w2[k + 3];
#pragma unroll
      for (int p = 0; p < P; ++p) z2[p] = wide_dot4(z2[p], w0, wa, wb, wc, wide_ld4(ws + p * WS_PT + WS_A + k));
    }
#pragma unroll
    for (int p = 0; p < P; ++p) act[p] = wide_act(z2[p], a.beta);
  }
  const float wh = L.wh[lane];
#pragma unroll
  for (int p = 0; p < P; ++p) {
    const float sdf = wave_sum(wh * act[p]) + L.bh;
    if (ok[p] && lane == 0) a.sdf[sp[p]] = sdf;
  }
  wave_sync_lds();
}

template <int FINP>
__global__ void __launch_bounds__(64 * WIDE_WAVES) k_wide_sdf(WideArgs a) {
  NSIM_DYN_SMEM(smem);
  float* nx;
  const WideLds L = wide_stage(reinterpret_cast<float*>(smem), a, nx);
  __syncthreads();
  const int lane = nsim_lane(), wave = (int)(threadIdx.x >> 6);
  float* ws = nx + wave * (WIDE_P * WS_PT);
  const int64_t Sv = wide_valid_count(a);
  const int64_t stride = (int64_t)gridDim.x * WIDE_WAVES * WIDE_P;
  for (int64_t s0 = ((int64_t)blockIdx.x * WIDE_WAVES + wave) * WIDE_P; s0 < Sv; s0 += stride) {
    int64_t sp[WIDE_P];
#pragma unroll
    for (int p = 0; p < WIDE_P; ++p) sp[p] = s0 + p;
    wide_points<FINP>(a, L, ws, sp, Sv, lane);
  }
}

// ------------------------------------------------------------------------------------------------ C ABI
static int wide_args(const NsimFieldMeta* meta, int n_freq, WideArgs& a) {
  if (!meta) return 2;
  if (meta->lotd.n_feats != 2 || meta->lotd.num_levels < 1 || meta->lotd.num_levels > NSIM_MAX_LEVELS) return 3;
  if (meta->sdf_D != 1 && meta->sdf_D != 2) return 22;
  if (n_freq < 0 || n_freq > 10) return 36;      // 2^9 rad = 81 revolutions: inside v_sin_f32's +-256-revolution domain
  if (meta->embed_E != 0 && meta->embed_E != 3 + 6 * n_freq) return 36;
  memset(&a, 0, sizeof(a));
  a.lotd = meta->lotd;
  a.sdf_D = meta->sdf_D;
  a.n_freq = n_freq;
  a.F1 = 2 * meta->lotd.num_levels;
  a.FIN = a.F1 + 3 + 6 * n_freq;
  if (a.FIN > 128) return 36;
  a.beta = meta->softplus_beta > 0.f ? meta->softplus_beta : -1.f;
  return 0;
}

static unsigned wide_grid(int64_t S) {
  int64_t b = (S + (int64_t)WIDE_WAVES * WIDE_P * 8 - 1) / ((int64_t)WIDE_WAVES * WIDE_P * 8);      // >= 8 groups per wave amortise the weight staging
  if (b < 1) b = 1;
  if (b > 256) b = 256;                                                                             // one workgroup per CU (LDS)
  return (unsigned)b;
}

extern "C" {

int nsim_wide_sdf(const NsimFieldMeta* meta, int32_t n_freq, const float* sdf_w, const float* sdf_b, const float* x,
                  const float* rays_o, const float* rays_d, const float* t, const int64_t* ridx, int64_t S,
                  const int64_t* n_dev, int64_t n_add, const float* feat_planes, float* sdf, void* stream) {
  WideArgs a;
  const int rc = wide_args(meta, n_freq, a);
  if (rc) return rc;
  if (S <= 0) return 0;
  if (!sdf_w || !sdf_b || !feat_planes || !sdf) return 2;
  if (!x && !(rays_o && rays_d && t && ridx)) return 24;
  a.sdf_w = sdf_w; a.sdf_b = sdf_b;
  a.x = x; a.rays_o = rays_o; a.rays_d = rays_d; a.t = t; a.ridx = ridx;
  a.S = S; a.PS = NSIM_PLANE_PITCH(S);
  a.S_dev = n_dev; a.S_add = n_add;
  a.h_pl = feat_planes;
  a.sdf = sdf;
  const dim3 grid(wide_grid(S)), block(64 * WIDE_WAVES);
  const size_t sh = (64 * WIDE_P1 + 64 * WIDE_P2 + 192 + (size_t)WIDE_WAVES * WIDE_P * WS_PT) * sizeof(float);
  // FINP: the first-layer width rounded up to one of the instantiated row lengths
  if (a.FIN <= 56) hipLaunchKernelGGL((k_wide_sdf<56>), grid, block, sh, (hipStream_t)stream, a);
  else if (a.FIN <= 72) hipLaunchKernelGGL((k_wide_sdf<72>), grid, block, sh, (hipStream_t)stream, a);
  else if (a.FIN <= 104) hipLaunchKernelGGL((k_wide_sdf<104>), grid, block, sh, (hipStream_t)stream, a);
  else hipLaunchKernelGGL((k_wide_sdf<128>), grid, block, sh, (hipStream_t)stream, a);
  NSIM_CHECK_LAUNCH();
  return 0;
}

}  // extern "C"
