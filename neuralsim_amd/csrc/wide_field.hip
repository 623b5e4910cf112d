// wide_field.hip -- the SDF decoder with an embedded-position input block ("wide" first layer), gfx950.
//
// ``surface_cfg.extra_pos_embed_cfg{type: sinusoidal_legacy, n_frequencies: 6}`` of the StyleLoTD Vehicle block
// (code_multi/configs/exps/fg_neus=hyper_lotd/no_fg_occ.221218.yaml:319-321): the decoder input is
// [grown features (32) | x, sin(2^k x), cos(2^k x), k = 0..5 (39)] = 71 values.  The MFMA decoders of field.hip contract
// over at most 64 inputs (two 16-level chunks of the level-major planes), and their joint backward is at its register
// limit; this file is the decoder for first layers up to 128 wide, written for the shape these models
// have in a step -- tens of thousands of samples of a few posed instances next to the street model's millions:
//
//   * a wave works on 4 (forward) / 2 (backward) points at a time, lane j = hidden unit j (width 64); the weights live in
//     LDS with ODD row pitches (W1: 129, W2: 65 floats), so "lane j walks row j" and "lane k walks column k" are both
//     bank-conflict free, and every weight read serves all points of the group; a layer input is staged once per wave and
//     read back as 16-byte LDS broadcasts;
//   * f32 throughout on the VALU (explicit FMAs); measured on MI355X (profiles/round5_wide_decoder_bench.txt): forward +
//     radiance 2.9 ns, backward 7.3 ns per point -- 0.19 + 0.48 ms per 65 k points, 4x the MFMA kernels' time for the same
//     model without the embedding, next to a street step of 12 ms;
//   * features h and dh/dx come from the level-major planes of nsim_field_fwd's gather (h_planes / J_planes), the
//     embedded position and ITS x-derivative are generated in the kernel; everything downstream -- radiance backward,
//     table scatter, occupancy, sampling -- is the common path;
//   * the backward mirrors k_field_bwd_j term by term (second-order path through the normals included); the weight-
//     gradient rows of a lane accumulate in registers over the wave's whole loop and leave once per workgroup.
//
// Conventions fixed here (nr3d_lib is absent): the embedding sees the AABB-normalised position x_n = 2 u - 1 in [-1, 1]
// (u = the pyramid's unit coordinate), order [x_n (3) | for k: sin(2^k x_n) (3), cos(2^k x_n) (3)], no factor pi.
#include "nsim_common.h"

// waves per workgroup: forward kernels 8 (two per SIMD hide the LDS latency of the weight walks; their compact staging fits);
// backward NSIM_WIDE_BWD_WAVES (4: the weight-gradient rows of a lane live in registers, ~500 of them -> one wave per SIMD;
// 8: the dW1 row goes through LDS atomics and the kernel fits 256 registers -> two waves per SIMD)
#ifndef NSIM_WIDE_BWD_WAVES
#define NSIM_WIDE_BWD_WAVES 4
#endif
#define WIDE_WAVES_OF(MODE) ((MODE) == 2 ? NSIM_WIDE_BWD_WAVES : 8)
#define WIDE_REGROW_MAX (NSIM_WIDE_BWD_WAVES == 4 ? 72 : 0)      // widest first layer whose dW1 row is held in registers
#define WIDE_P1 129       // LDS pitch of a W1 row (<= 128 inputs), odd
#define WIDE_P2 65        // LDS pitch of a 64-wide row, odd
#define WIDE_PR 27        // pitch of the radiance first layer (26 inputs), odd
#define WIDE_LOG2E 1.4426950408889634f
#define WIDE_LN2 0.6931471805599453f

struct WideArgs {
  NsimLotdMeta lotd;
  int sdf_D, n_freq, F1, FIN;
  float beta;                       // < 0: relu
  const float *sdf_w, *sdf_b, *rad_w, *rad_b;
  const float *x, *rays_o, *rays_d, *t, *h_appear;
  const int64_t* ridx;
  int64_t S, PS;
  const int64_t* S_dev;
  int64_t S_add;
  const float* h_pl;                // f32 planes [NLP][PS][2]
  const void* J_pl;                 // dh/dx planes [NLP][PS][2][3], f16 (j16) | f32
  int j16;
  float *sdf, *nablas, *rgb;
  // backward
  const float *dsdf, *dnablas;
  float *dh_pl, *g_pl;              // [NLP][S][2] hand-off to the scatter
  float *dsdf_w, *dsdf_b;
};

// activation value / first / second derivative of z
struct Act3 {
  float a, s, c;
};
__device__ __forceinline__ Act3 wide_act(float z, float beta) {
  Act3 r;
  if (beta < 0.f) {      // relu
    r.a = fmaxf(z, 0.f);
    r.s = z > 0.f ? 1.f : 0.f;
    r.c = 0.f;
    return r;
  }
  const float tt = nsim_exp2(-fabsf(z) * (beta * WIDE_LOG2E));      // exp(-beta |z|)
  r.a = fmaxf(z, 0.f) + nsim_log2(1.0f + tt) * (WIDE_LN2 / beta);
  r.s = z >= 0.f ? 1.0f / (1.0f + tt) : tt / (1.0f + tt);
  r.c = beta * r.s * (1.0f - r.s);
  return r;
}

__device__ __forceinline__ int64_t wide_valid_count(const WideArgs& a) {
  if (!a.S_dev) return a.S;
  const int64_t n = a.S_dev[0] + a.S_add;
  return n > a.S ? 0 : n;
}

// position of point s, its view direction and ray
__device__ __forceinline__ void wide_point(const WideArgs& a, int64_t s, float (&xx)[3], int64_t& ray) {
  ray = 0;
  if (a.x) {
    xx[0] = a.x[3 * s]; xx[1] = a.x[3 * s + 1]; xx[2] = a.x[3 * s + 2];
    if (a.ridx) ray = a.ridx[s];
  } else {
    ray = a.ridx[s];
    const float tt = a.t[s];
#pragma unroll
    for (int c = 0; c < 3; ++c) xx[c] = a.rays_o[3 * ray + c] + tt * a.rays_d[3 * ray + c];
  }
}

// embedded-position input i (F1 <= i < FIN) of the first layer and its derivative w.r.t. the three position axes; zero for the
// feature block and past the end (branch-free: selects)
__device__ __forceinline__ void wide_embed(const WideArgs& a, const float (&xx)[3], int i, float& v, float (&dv)[3]) {
  const int m = i - a.F1;
  const bool on = m >= 0 && i < a.FIN;
  const int mm = on ? m : 0;
  const int c = mm < 3 ? mm : (mm - 3) % 3;
  // (selects, not indexed private arrays: a lane-dependent index would put xx / dv into scratch)
  const bool unit_cube = a.lotd.x_scale[0] == 0.f && a.lotd.x_scale[1] == 0.f && a.lotd.x_scale[2] == 0.f;
  const float xc = c == 0 ? xx[0] : (c == 1 ? xx[1] : xx[2]);
  const float sc = unit_cube ? 0.5f : (c == 0 ? a.lotd.x_scale[0] : (c == 1 ? a.lotd.x_scale[1] : a.lotd.x_scale[2]));
  const float sh = unit_cube ? 0.5f : (c == 0 ? a.lotd.x_shift[0] : (c == 1 ? a.lotd.x_shift[1] : a.lotd.x_shift[2]));
  const float xn = 2.0f * (xc * sc + sh) - 1.0f, dxn = 2.0f * sc;
  const int k = mm < 3 ? 0 : (mm - 3) / 6;
  const bool is_cos = mm >= 3 && ((mm - 3) % 6) >= 3;
  const float fr = (float)(1 << k), ang = fr * xn;
  const float sn = nsim_sin(ang), cs = nsim_cos(ang);
  float val = is_cos ? cs : sn, d = (is_cos ? -sn : cs) * fr * dxn;
  if (mm < 3) {
    val = xn;
    d = dxn;
  }
  v = on ? val : 0.f;
  d = on ? d : 0.f;
  dv[0] = c == 0 ? d : 0.f;
  dv[1] = c == 1 ? d : 0.f;
  dv[2] = c == 2 ? d : 0.f;
}

// ---------------------------------------------------------------------------------------------------------------
// Work of a wave: P points at a time (forward kernels 4, backward 2: its register rows) (the weights read from LDS are used for all of them).  Lane j = hidden unit j.
// Four passes over the weights -- two quantities per pass share every weight read:
//   A  rows of W1:     z1 = W1 in + b1        | dL/d d1 = W1 gh          (gh = dh/dx . gn: the input tangent along dL/dnablas)
//   B  rows of W2:     z2 = W2 a1 + b2        | dL/d d2 = W2 eh1         (eh1 = dL/d e1 = dL/d d1 . s1)
//   C  columns of W2:  e1 = W2^T d2           | dL/d a1 = W2^T dz2
//   D  columns of W1:  g = W1^T d1            | dL/d in = W1^T dz1
// (the forward kernels run the left halves only).  Layer inputs are staged per wave and read back as 16-byte LDS
// broadcasts; the weight-gradient rows dW1[j][.], dW2[j][.] of lane j accumulate in REGISTERS over the wave's whole loop
// (compile-time indices: FINP = the first-layer width rounded up, a template parameter) and meet the other waves' in LDS once.
#define WIDE_P_OF(MODE) ((MODE) == 2 ? 2 : 4)

struct WideLds {
  float *W1, *W2, *b1, *b2, *wh;
  float bh;
};

// stage the SDF decoder's weights (flat f32: W1 [64 x FIN], (W2 [64 x 64]), wh [64]; b1 [64], (b2 [64]), bh) into LDS;
// W1 rows are zero-padded to FINP
__device__ __forceinline__ WideLds wide_stage(float* base, const WideArgs& a, float*& next) {
  WideLds L;
  L.W1 = base;
  L.W2 = L.W1 + 64 * WIDE_P1;
  L.b1 = L.W2 + 64 * WIDE_P2;
  L.b2 = L.b1 + 64;
  L.wh = L.b2 + 64;
  next = L.wh + 64;
  const int FIN = a.FIN;
  for (int i = threadIdx.x; i < 64 * 128; i += blockDim.x) {      // (all 128 columns: the column passes read past FINP)
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

// per-wave, per-point staging vectors (floats; every vector 16-byte aligned).  Backward: all of them; forward: the compact set
#define WS_IN 0                              // [128] layer input
#define WS_GH 128                            // [128] its tangent along gn              (backward only)
#define WS_A (MODE == 2 ? 256 : 128)         // [64] a1
#define WS_E (MODE == 2 ? 320 : 192)         // [64] eh1 | radiance hidden layer
#define WS_B (MODE == 2 ? 384 : 256)         // [64] d2
#define WS_F 448                             // [64] dz2                                (backward only)
#define WS_C (MODE == 2 ? 512 : 320)         // [64] d1
#define WS_G 576                             // [64] dz1                                (backward only)
#define WS_R 384                             // [32] radiance input                     (forward only)
#define WS_PT_OF(MODE) ((MODE) == 2 ? 640 : 416)
#define WS_PT WS_PT_OF(MODE)

struct __attribute__((aligned(16))) float4w {
  float x, y, z, w;
};
__device__ __forceinline__ float4w wide_ld4(const float* p) { return *reinterpret_cast<const float4w*>(p); }
// acc + w . v with fused multiply-adds (this translation unit is built with -ffp-contract=off: the fusion is spelled out)
__device__ __forceinline__ float wide_dot4(float acc, float w0, float w1, float w2, float w3, const float4w& v) {
  return fmaf(w3, v.w, fmaf(w2, v.z, fmaf(w1, v.y, fmaf(w0, v.x, acc))));
}

__device__ __forceinline__ void wide_sh4(const float (&d)[3], float (&o)[16]) {
  const float x = d[0], y = d[1], z = d[2];
  const float xy = x * y, xz = x * z, yz = y * z, x2 = x * x, y2 = y * y, z2 = z * z;
  o[0] = 0.28209479177387814f;
  o[1] = -0.48860251190291987f * y;
  o[2] = 0.48860251190291987f * z;
  o[3] = -0.48860251190291987f * x;
  o[4] = 1.0925484305920792f * xy;
  o[5] = -1.0925484305920792f * yz;
  o[6] = 0.94617469575755997f * z2 - 0.31539156525251999f;
  o[7] = -1.0925484305920792f * xz;
  o[8] = 0.54627421529603959f * x2 - 0.54627421529603959f * y2;
  o[9] = 0.59004358992664352f * y * (-3.0f * x2 + y2);
  o[10] = 2.8906114426405538f * xy * z;
  o[11] = 0.45704579946446572f * y * (1.0f - 5.0f * z2);
  o[12] = 0.3731763325901154f * z * (5.0f * z2 - 3.0f);
  o[13] = 0.45704579946446572f * x * (1.0f - 5.0f * z2);
  o[14] = 1.4453057213202769f * z * (x2 - y2);
  o[15] = 0.59004358992664352f * x * (-x2 + 3.0f * y2);
}

// MODE 0: sdf | 1: sdf, nablas (+ rgb) | 2: backward of the SDF branch
template <int MODE, int FINP>
__device__ __forceinline__ void wide_points(const WideArgs& a, const WideLds& L, float* ws, const float* R1, const float* R2,
                                            const float* R3, const float* rb, const int64_t (&sp)[WIDE_P_OF(MODE)], int64_t Sv, int lane,
                                            float (&dW1r)[(MODE == 2 && FINP <= WIDE_REGROW_MAX) ? FINP : 1], float (&dW2r)[MODE == 2 ? 64 : 1],
                                            float (&dvec)[4], float* dW1lds) {
  constexpr int P = WIDE_P_OF(MODE);
  const bool two = lane + 64 < FINP;
  bool ok[P];
  float xx[P][3];
  int64_t ray[P];
  float J0[P][3], J1[P][3], gs[P], gn[P][3];
  // ---- inputs.  Every global load of the group is issued before the first one is waited for (branch-free addresses: a lane
  // outside the feature block, or a point past the end, reads element 0 of its array and drops the value) -- one memory
  // latency per group instead of one per load (the first version of this kernel: 8 x ~1 us per group).
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
#pragma unroll
    for (int c = 0; c < 3; ++c) J0[p][c] = J1[p][c] = 0.f;
    if (a.J_pl) {      // (the no-grad query has feature planes only)
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        if (a.j16) {
          J0[p][c] = (float)reinterpret_cast<const f16*>(a.J_pl)[3 * o0 + c];
          J1[p][c] = (float)reinterpret_cast<const f16*>(a.J_pl)[3 * o1 + c];
        } else {
          J0[p][c] = reinterpret_cast<const float*>(a.J_pl)[3 * o0 + c];
          J1[p][c] = reinterpret_cast<const float*>(a.J_pl)[3 * o1 + c];
        }
      }
    }
    gs[p] = 0.f;
    gn[p][0] = gn[p][1] = gn[p][2] = 0.f;
    if constexpr (MODE == 2) {
      if (a.dsdf) gs[p] = a.dsdf[s];
      if (a.dnablas) {
#pragma unroll
        for (int c = 0; c < 3; ++c) gn[p][c] = a.dnablas[3 * s + c];
      }
    }
    wide_point(a, s, xx[p], ray[p]);
  }
#pragma unroll
  for (int p = 0; p < P; ++p) {
    float e0, e1v, de0[3], de1[3];
    wide_embed(a, xx[p], lane, e0, de0);
    wide_embed(a, xx[p], lane + 64, e1v, de1);
    const float m = ok[p] ? 1.f : 0.f;
    const float in0 = (f0 ? h0[p] : e0) * m, in1 = (f1 ? h1[p] : e1v) * m;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      J0[p][c] = (f0 ? J0[p][c] : de0[c]) * m;
      J1[p][c] = (f1 ? J1[p][c] : de1[c]) * m;
    }
    if constexpr (MODE == 2) {
      gs[p] *= m;
#pragma unroll
      for (int c = 0; c < 3; ++c) gn[p][c] *= m;
    }
    float* w = ws + p * WS_PT;
    w[WS_IN + lane] = in0;
    w[WS_IN + 64 + lane] = in1;
    if constexpr (MODE == 2) {
      w[WS_GH + lane] = J0[p][0] * gn[p][0] + J0[p][1] * gn[p][1] + J0[p][2] * gn[p][2];
      w[WS_GH + 64 + lane] = J1[p][0] * gn[p][0] + J1[p][1] * gn[p][1] + J1[p][2] * gn[p][2];
    }
  }
  wave_sync_lds();
  // ---- pass A: rows of W1
  float z1[P], dh1[P];
#pragma unroll
  for (int p = 0; p < P; ++p) {
    z1[p] = L.b1[lane];
    dh1[p] = 0.f;
  }
  {
    const float* w1 = L.W1 + lane * WIDE_P1;
#pragma unroll 4
    for (int i = 0; i < FINP; i += 4) {
      const float w0 = w1[i], wa = w1[i + 1], wb = w1[i + 2], wc = w1[i + 3];
#pragma unroll
      for (int p = 0; p < P; ++p) {
        const float4w v = wide_ld4(ws + p * WS_PT + WS_IN + i);
        z1[p] = wide_dot4(z1[p], w0, wa, wb, wc, v);
        if constexpr (MODE == 2) {
          const float4w g = wide_ld4(ws + p * WS_PT + WS_GH + i);
          dh1[p] = wide_dot4(dh1[p], w0, wa, wb, wc, g);
        }
      }
    }
  }
  Act3 u1[P], u2[P];
  float eh1[P];
#pragma unroll
  for (int p = 0; p < P; ++p) {
    u1[p] = wide_act(z1[p], a.beta);
    eh1[p] = dh1[p] * u1[p].s;
    ws[p * WS_PT + WS_A + lane] = u1[p].a;
    if constexpr (MODE == 2) ws[p * WS_PT + WS_E + lane] = eh1[p];
  }
  wave_sync_lds();
  // ---- pass B: rows of W2
  float sdf[P], d2[P], dz2[P], e1[P], da1[P], whv[P], dz1[P];
#pragma unroll
  for (int p = 0; p < P; ++p) e1[p] = da1[p] = 0.f;
  const float wh = L.wh[lane];
  if (a.sdf_D == 2) {
    float z2[P], dh2[P];
#pragma unroll
    for (int p = 0; p < P; ++p) {
      z2[p] = L.b2[lane];
      dh2[p] = 0.f;
    }
    const float* w2 = L.W2 + lane * WIDE_P2;
#pragma unroll 4
    for (int k = 0; k < 64; k += 4) {
      const float w0 = w2[k], wa = w2[k + 1], wb = w2[k + 2], wc = w2[k + 3];
#pragma unroll
      for (int p = 0; p < P; ++p) {
        const float4w v = wide_ld4(ws + p * WS_PT + WS_A + k);
        z2[p] = wide_dot4(z2[p], w0, wa, wb, wc, v);
        if constexpr (MODE == 2) {
          const float4w g = wide_ld4(ws + p * WS_PT + WS_E + k);
          dh2[p] = wide_dot4(dh2[p], w0, wa, wb, wc, g);
        }
      }
    }
#pragma unroll
    for (int p = 0; p < P; ++p) {
      u2[p] = wide_act(z2[p], a.beta);
      sdf[p] = wave_sum(wh * u2[p].a) + L.bh;
      d2[p] = wh * u2[p].s;
      whv[p] = dh2[p] * u2[p].s + gs[p] * u2[p].a;
      dz2[p] = gs[p] * wh * u2[p].s + dh2[p] * wh * u2[p].c;
      if constexpr (MODE >= 1) ws[p * WS_PT + WS_B + lane] = d2[p];
      if constexpr (MODE == 2) ws[p * WS_PT + WS_F + lane] = dz2[p];
    }
    if constexpr (MODE >= 1) {
      wave_sync_lds();
      // ---- pass C: columns of W2
#pragma unroll 4
      for (int jj = 0; jj < 64; jj += 4) {
        const float w0 = L.W2[jj * WIDE_P2 + lane], wa = L.W2[(jj + 1) * WIDE_P2 + lane], wb = L.W2[(jj + 2) * WIDE_P2 + lane],
                    wc = L.W2[(jj + 3) * WIDE_P2 + lane];
#pragma unroll
        for (int p = 0; p < P; ++p) {
          const float4w v = wide_ld4(ws + p * WS_PT + WS_B + jj);
          e1[p] = wide_dot4(e1[p], w0, wa, wb, wc, v);
          if constexpr (MODE == 2) {
            const float4w g = wide_ld4(ws + p * WS_PT + WS_F + jj);
            da1[p] = wide_dot4(da1[p], w0, wa, wb, wc, g);
          }
        }
      }
    }
#pragma unroll
    for (int p = 0; p < P; ++p) dz1[p] = dh1[p] * e1[p] * u1[p].c + da1[p] * u1[p].s;
  } else {
#pragma unroll
    for (int p = 0; p < P; ++p) {
      u2[p] = u1[p];
      sdf[p] = wave_sum(wh * u1[p].a) + L.bh;
      d2[p] = dz2[p] = da1[p] = 0.f;
      e1[p] = wh;
      whv[p] = dh1[p] * u1[p].s + gs[p] * u1[p].a;
      dz1[p] = gs[p] * wh * u1[p].s + dh1[p] * wh * u1[p].c;
    }
  }
  if constexpr (MODE == 0) {
#pragma unroll
    for (int p = 0; p < P; ++p)
      if (ok[p] && lane == 0) a.sdf[sp[p]] = sdf[p];
    wave_sync_lds();
    return;
  }
  float d1[P];
#pragma unroll
  for (int p = 0; p < P; ++p) {
    d1[p] = u1[p].s * e1[p];
    ws[p * WS_PT + WS_C + lane] = d1[p];
    if constexpr (MODE == 2) ws[p * WS_PT + WS_G + lane] = dz1[p];
  }
  wave_sync_lds();
  // ---- pass D: columns of W1
  float g0[P], g1[P], dh0[P], dhh[P];
#pragma unroll
  for (int p = 0; p < P; ++p) g0[p] = g1[p] = dh0[p] = dhh[p] = 0.f;
#pragma unroll 4
  for (int jj = 0; jj < 64; jj += 4) {
    float wl[4], wu[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      wl[q] = L.W1[(jj + q) * WIDE_P1 + lane];
      wu[q] = two ? L.W1[(jj + q) * WIDE_P1 + 64 + lane] : 0.f;
    }
#pragma unroll
    for (int p = 0; p < P; ++p) {
      const float4w v = wide_ld4(ws + p * WS_PT + WS_C + jj);
      g0[p] = wide_dot4(g0[p], wl[0], wl[1], wl[2], wl[3], v);
      g1[p] = wide_dot4(g1[p], wu[0], wu[1], wu[2], wu[3], v);
      if constexpr (MODE == 2) {
        const float4w g = wide_ld4(ws + p * WS_PT + WS_G + jj);
        dh0[p] = wide_dot4(dh0[p], wl[0], wl[1], wl[2], wl[3], g);
        dhh[p] = wide_dot4(dhh[p], wu[0], wu[1], wu[2], wu[3], g);
      }
    }
  }
  if constexpr (MODE == 1) {
#pragma unroll
    for (int p = 0; p < P; ++p) {
      const float nab0 = wave_sum(g0[p] * J0[p][0] + g1[p] * J1[p][0]);
      const float nab1 = wave_sum(g0[p] * J0[p][1] + g1[p] * J1[p][1]);
      const float nab2 = wave_sum(g0[p] * J0[p][2] + g1[p] * J1[p][2]);
      const int64_t s = sp[p];
      if (ok[p] && lane == 0) {
        a.sdf[s] = sdf[p];
        a.nablas[3 * s] = nab0;
        a.nablas[3 * s + 1] = nab1;
        a.nablas[3 * s + 2] = nab2;
      }
      if (a.rgb) {      // radiance: input [x (3) | SH4(view dir) (16) | nablas (3) | appearance (4)], relu 2 x 64, sigmoid
        float* w = ws + p * WS_PT;
        if (lane == 0) {
          float vd[3] = {0.f, 0.f, 1.f};
          if (ok[p]) {
            vd[0] = a.rays_d[3 * ray[p]];
            vd[1] = a.rays_d[3 * ray[p] + 1];
            vd[2] = a.rays_d[3 * ray[p] + 2];
          }
          float sh[16];
          wide_sh4(vd, sh);
          w[WS_R + 0] = xx[p][0];
          w[WS_R + 1] = xx[p][1];
          w[WS_R + 2] = xx[p][2];
#pragma unroll
          for (int k = 0; k < 16; ++k) w[WS_R + 3 + k] = sh[k];
          w[WS_R + 19] = nab0;
          w[WS_R + 20] = nab1;
          w[WS_R + 21] = nab2;
#pragma unroll
          for (int c = 0; c < 4; ++c) w[WS_R + 22 + c] = (ok[p] && a.h_appear) ? a.h_appear[4 * ray[p] + c] : 0.f;
        }
      }
    }
    if (a.rgb) {
      wave_sync_lds();
      float r1[P], r2[P];
#pragma unroll
      for (int p = 0; p < P; ++p) r1[p] = rb[lane];
      for (int i = 0; i < 26; ++i) {
        const float wv = R1[lane * WIDE_PR + i];
#pragma unroll
        for (int p = 0; p < P; ++p) r1[p] += wv * ws[p * WS_PT + WS_R + i];
      }
#pragma unroll
      for (int p = 0; p < P; ++p) ws[p * WS_PT + WS_E + lane] = fmaxf(r1[p], 0.f);
      wave_sync_lds();
#pragma unroll
      for (int p = 0; p < P; ++p) r2[p] = rb[64 + lane];
#pragma unroll 4
      for (int k = 0; k < 64; k += 4) {
        const float w0 = R2[lane * WIDE_P2 + k], wa = R2[lane * WIDE_P2 + k + 1], wb = R2[lane * WIDE_P2 + k + 2],
                    wc = R2[lane * WIDE_P2 + k + 3];
#pragma unroll
        for (int p = 0; p < P; ++p) {
          const float4w v = wide_ld4(ws + p * WS_PT + WS_E + k);
          r2[p] = wide_dot4(r2[p], w0, wa, wb, wc, v);
        }
      }
#pragma unroll
      for (int p = 0; p < P; ++p) {
        const float r = fmaxf(r2[p], 0.f);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          const float o = wave_sum(R3[64 * c + lane] * r) + rb[128 + c];
          if (ok[p] && lane == 0) a.rgb[3 * sp[p] + c] = 1.0f / (1.0f + expf(-o));
        }
      }
    }
    wave_sync_lds();
    return;
  }
  if constexpr (MODE == 2) {
    // ---- hand-off planes of the scatter: g = d sdf / d h, dL/dh
#pragma unroll
    for (int p = 0; p < P; ++p) {
      if (!ok[p]) continue;
      const int64_t s = sp[p];
      if (lane < a.F1) {
        const int64_t o = ((int64_t)(lane >> 1) * a.S + s) * 2 + (lane & 1);
        if (a.g_pl) a.g_pl[o] = g0[p];
        if (a.dh_pl) a.dh_pl[o] = dh0[p];
      }
      if (lane + 64 < a.F1) {
        const int64_t o = ((int64_t)((lane + 64) >> 1) * a.S + s) * 2 + (lane & 1);
        if (a.g_pl) a.g_pl[o] = g1[p];
        if (a.dh_pl) a.dh_pl[o] = dhh[p];
      }
    }
    // ---- weight gradients of lane j's rows (invalid points carry zeros: gs = gn = 0 and zero inputs give dz = d1 . 0 ...
    // except through the biases -- mask them)
#pragma unroll
    for (int p = 0; p < P; ++p) {
      const float m = ok[p] ? 1.f : 0.f;
      const float dd1 = d1[p] * m, dd2 = d2[p] * m, zz1 = dz1[p] * m, zz2 = dz2[p] * m;
      const float* w = ws + p * WS_PT;
      if constexpr (FINP <= WIDE_REGROW_MAX) {
#pragma unroll
        for (int i = 0; i < FINP; i += 4) {
          const float4w g = wide_ld4(w + WS_GH + i), v = wide_ld4(w + WS_IN + i);
          dW1r[i] = fmaf(dd1, g.x, fmaf(zz1, v.x, dW1r[i]));
          dW1r[i + 1] = fmaf(dd1, g.y, fmaf(zz1, v.y, dW1r[i + 1]));
          dW1r[i + 2] = fmaf(dd1, g.z, fmaf(zz1, v.z, dW1r[i + 2]));
          dW1r[i + 3] = fmaf(dd1, g.w, fmaf(zz1, v.w, dW1r[i + 3]));
        }
      } else {      // wider first layers: the row goes through LDS (the register row would spill)
        float* dw1 = dW1lds + lane * WIDE_P1;
        for (int i = 0; i < FINP; i += 4) {
          const float4w g = wide_ld4(w + WS_GH + i), v = wide_ld4(w + WS_IN + i);
          atomicAdd(&dw1[i], dd1 * g.x + zz1 * v.x);
          atomicAdd(&dw1[i + 1], dd1 * g.y + zz1 * v.y);
          atomicAdd(&dw1[i + 2], dd1 * g.z + zz1 * v.z);
          atomicAdd(&dw1[i + 3], dd1 * g.w + zz1 * v.w);
        }
      }
      if (a.sdf_D == 2) {
#pragma unroll
        for (int k = 0; k < 64; k += 4) {
          const float4w e = wide_ld4(w + WS_E + k), v = wide_ld4(w + WS_A + k);
          dW2r[k] = fmaf(dd2, e.x, fmaf(zz2, v.x, dW2r[k]));
          dW2r[k + 1] = fmaf(dd2, e.y, fmaf(zz2, v.y, dW2r[k + 1]));
          dW2r[k + 2] = fmaf(dd2, e.z, fmaf(zz2, v.z, dW2r[k + 2]));
          dW2r[k + 3] = fmaf(dd2, e.w, fmaf(zz2, v.w, dW2r[k + 3]));
        }
      }
      dvec[0] += zz1;               // d b1[j]
      dvec[1] += zz2;               // d b2[j]
      dvec[2] += whv[p] * m;        // d wh[j]
      dvec[3] += gs[p] * m;         // d bh (every lane carries the same sum)
    }
    wave_sync_lds();
  }
}

template <int MODE, int FINP>
__global__ void __launch_bounds__(64 * WIDE_WAVES_OF(MODE)) k_wide(WideArgs a) {
  NSIM_DYN_SMEM(smem);
  float* nx;
  const WideLds L = wide_stage(reinterpret_cast<float*>(smem), a, nx);
  float *R1 = nullptr, *R2 = nullptr, *R3 = nullptr, *rb = nullptr;
  if constexpr (MODE == 1) {      // radiance network (rad_w: [Wr1 (64 x 26), Wr2 (64 x 64), Wr3 (3 x 64)], rad_b [64, 64, 3])
    R1 = nx;
    R2 = R1 + 64 * WIDE_PR;
    R3 = R2 + 64 * WIDE_P2;
    rb = R3 + 192;
    nx = rb + 132;
    if (a.rgb) {
      for (int i = threadIdx.x; i < 64 * 26; i += blockDim.x) R1[(i / 26) * WIDE_PR + (i % 26)] = a.rad_w[i];
      for (int i = threadIdx.x; i < 4096; i += blockDim.x) R2[(i >> 6) * WIDE_P2 + (i & 63)] = a.rad_w[64 * 26 + i];
      for (int i = threadIdx.x; i < 192; i += blockDim.x) R3[i] = a.rad_w[64 * 26 + 4096 + i];
      for (int i = threadIdx.x; i < 131; i += blockDim.x) rb[i] = a.rad_b[i];
    }
  }
  float *dW1 = nullptr, *dW2 = nullptr, *dv = nullptr;
  if constexpr (MODE == 2) {
    dW1 = nx;
    dW2 = dW1 + 64 * WIDE_P1;
    dv = dW2 + 64 * WIDE_P2;      // [4][64]: d b1, d b2, d wh, d bh (lane 0)
    nx = dv + 256;
    for (int i = threadIdx.x; i < (int)(nx - dW1); i += blockDim.x) dW1[i] = 0.f;
  }
  __syncthreads();
  const int lane = nsim_lane(), wave = (int)(threadIdx.x >> 6);
  constexpr int P = WIDE_P_OF(MODE);
  float* ws = nx + wave * (P * WS_PT);
  constexpr int NR1 = (MODE == 2 && FINP <= WIDE_REGROW_MAX) ? FINP : 1;
  float dW1r[NR1], dW2r[MODE == 2 ? 64 : 1], dvec[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < NR1; ++i) dW1r[i] = 0.f;
#pragma unroll
  for (int i = 0; i < (MODE == 2 ? 64 : 1); ++i) dW2r[i] = 0.f;
  const int64_t Sv = wide_valid_count(a);
  constexpr int NWV = WIDE_WAVES_OF(MODE);
  const int64_t stride = (int64_t)gridDim.x * NWV * P;
  for (int64_t s0 = ((int64_t)blockIdx.x * NWV + wave) * P; s0 < Sv; s0 += stride) {
    int64_t sp[P];
#pragma unroll
    for (int p = 0; p < P; ++p) sp[p] = s0 + p;
    wide_points<MODE, FINP>(a, L, ws, R1, R2, R3, rb, sp, Sv, lane, dW1r, dW2r, dvec, dW1);
  }
  if constexpr (MODE == 2) {
    // the waves' register rows meet in LDS, then one flush per workgroup:
    // dsdf_w = [dW1 (64 x FIN), (dW2 (64 x 64)), d wh (64)], dsdf_b = [d b1, (d b2), d bh]
    if constexpr (FINP <= WIDE_REGROW_MAX) {
#pragma unroll
      for (int i = 0; i < FINP; ++i)
        if (dW1r[i] != 0.f) atomicAdd(&dW1[lane * WIDE_P1 + i], dW1r[i]);
    }
    if (a.sdf_D == 2) {
#pragma unroll
      for (int k = 0; k < 64; ++k)
        if (dW2r[k] != 0.f) atomicAdd(&dW2[lane * WIDE_P2 + k], dW2r[k]);
    }
#pragma unroll
    for (int q = 0; q < 3; ++q)
      if (dvec[q] != 0.f) atomicAdd(&dv[64 * q + lane], dvec[q]);
    if (lane == 0 && dvec[3] != 0.f) atomicAdd(&dv[192], dvec[3]);
    __syncthreads();
    const int FIN = a.FIN;
    float* gw2 = a.dsdf_w + 64 * FIN;
    float* gwh = gw2 + (a.sdf_D == 2 ? 4096 : 0);
    for (int i = threadIdx.x; i < 64 * FIN; i += blockDim.x) {
      const float v = dW1[(i / FIN) * WIDE_P1 + (i % FIN)];
      if (v != 0.f) atomicAdd(&a.dsdf_w[i], v);
    }
    if (a.sdf_D == 2)
      for (int i = threadIdx.x; i < 4096; i += blockDim.x) {
        const float v = dW2[(i >> 6) * WIDE_P2 + (i & 63)];
        if (v != 0.f) atomicAdd(&gw2[i], v);
      }
    for (int i = threadIdx.x; i < 64; i += blockDim.x) {
      if (dv[i] != 0.f) atomicAdd(&a.dsdf_b[i], dv[i]);
      if (a.sdf_D == 2 && dv[64 + i] != 0.f) atomicAdd(&a.dsdf_b[64 + i], dv[64 + i]);
      if (dv[128 + i] != 0.f) atomicAdd(&gwh[i], dv[128 + i]);
    }
    if (threadIdx.x == 0 && dv[192] != 0.f) atomicAdd(&a.dsdf_b[a.sdf_D == 2 ? 128 : 64], dv[192]);
  }
}


// ------------------------------------------------------------------------------------------------ C ABI
static int wide_args(const NsimFieldMeta* meta, int n_freq, WideArgs& a) {
  if (!meta) return 2;
  if (meta->lotd.n_feats != 2 || meta->lotd.num_levels < 1 || meta->lotd.num_levels > NSIM_MAX_LEVELS) return 3;
  if (meta->sdf_D != 1 && meta->sdf_D != 2) return 22;
  if (n_freq < 0 || n_freq > 10) return 36;      // 2^9 rad = 81 revolutions: inside v_sin_f32's +-256-revolution domain
  memset(&a, 0, sizeof(a));
  a.lotd = meta->lotd;
  a.sdf_D = meta->sdf_D;
  a.n_freq = n_freq;
  a.F1 = 2 * meta->lotd.num_levels;
  a.FIN = a.F1 + 3 + 6 * n_freq;
  if (a.FIN > 128) return 36;
  a.beta = meta->softplus_beta > 0.f ? meta->softplus_beta : -1.f;
  a.j16 = NSIM_J_ELEM_BYTES(meta->precision) == 2;
  return 0;
}

static unsigned wide_grid(int64_t S, int waves, int P) {
  int64_t b = (S + (int64_t)waves * P * 8 - 1) / ((int64_t)waves * P * 8);      // >= 8 groups per wave amortise the weight staging
  if (b < 1) b = 1;
  if (b > 256) b = 256;                                                           // one workgroup per CU (LDS)
  return (unsigned)b;
}

static size_t wide_lds_floats(int mode) {
  size_t n = 64 * WIDE_P1 + 64 * WIDE_P2 + 192;
  if (mode == 1) n += 64 * WIDE_PR + 64 * WIDE_P2 + 192 + 132;
  if (mode == 2) n += 64 * WIDE_P1 + 64 * WIDE_P2 + 256;
  return n + (size_t)WIDE_WAVES_OF(mode) * WIDE_P_OF(mode) * WS_PT_OF(mode);
}

// FINP: the first-layer width rounded up to one of the instantiated row lengths
template <int MODE>
static void wide_launch(const WideArgs& a, hipStream_t stream) {
  const dim3 grid(wide_grid(a.S, WIDE_WAVES_OF(MODE), WIDE_P_OF(MODE))), block(64 * WIDE_WAVES_OF(MODE));
  const size_t sh = wide_lds_floats(MODE) * sizeof(float);
  if (a.FIN <= 56) hipLaunchKernelGGL((k_wide<MODE, 56>), grid, block, sh, stream, a);
  else if (a.FIN <= 72) hipLaunchKernelGGL((k_wide<MODE, 72>), grid, block, sh, stream, a);
  else if (a.FIN <= 104) hipLaunchKernelGGL((k_wide<MODE, 104>), grid, block, sh, stream, a);
  else hipLaunchKernelGGL((k_wide<MODE, 128>), grid, block, sh, stream, a);
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
  a.J_pl = nullptr;      // feature planes only: no x-derivative is read
  wide_launch<0>(a, (hipStream_t)stream);
  NSIM_CHECK_LAUNCH();
  return 0;
}

int nsim_wide_fwd(const NsimFieldMeta* meta, int32_t n_freq, const float* sdf_w, const float* sdf_b, const float* rad_w,
                  const float* rad_b, const float* x, const float* rays_o, const float* rays_d, const float* t,
                  const int64_t* ridx, const float* h_appear, int64_t S, const float* h_planes, const void* J_planes,
                  float* sdf, float* nablas, float* rgb, const int64_t* n_dev, int64_t n_add, void* stream) {
  WideArgs a;
  const int rc = wide_args(meta, n_freq, a);
  if (rc) return rc;
  if (S <= 0) return 0;
  if (!sdf_w || !sdf_b || !h_planes || !J_planes || !sdf || !nablas) return 2;
  if (!x && !(rays_o && rays_d && t && ridx)) return 24;
  if (rgb && !(rays_d && ridx && rad_w && rad_b)) return 25;
  a.sdf_w = sdf_w; a.sdf_b = sdf_b; a.rad_w = rad_w; a.rad_b = rad_b;
  a.x = x; a.rays_o = rays_o; a.rays_d = rays_d; a.t = t; a.ridx = ridx; a.h_appear = h_appear;
  a.S = S; a.PS = NSIM_PLANE_PITCH(S);
  a.S_dev = n_dev; a.S_add = n_add;
  a.h_pl = h_planes; a.J_pl = J_planes;
  a.sdf = sdf; a.nablas = nablas; a.rgb = rgb;
  wide_launch<1>(a, (hipStream_t)stream);
  NSIM_CHECK_LAUNCH();
  return 0;
}

int nsim_wide_bwd_sdf(const NsimFieldMeta* meta, int32_t n_freq, const float* sdf_w, const float* sdf_b, const float* x,
                      const float* rays_o, const float* rays_d, const float* t, const int64_t* ridx, int64_t S,
                      const float* h_planes, const void* J_planes, int64_t plane_pitch, const float* dsdf,
                      const float* dnablas, float* dh_planes, float* g_planes, float* dsdf_w, float* dsdf_b, void* stream) {
  WideArgs a;
  const int rc = wide_args(meta, n_freq, a);
  if (rc) return rc;
  if (S <= 0) return 0;
  if (!sdf_w || !sdf_b || !h_planes || !J_planes || !dsdf_w || !dsdf_b) return 2;
  if (!x && !(rays_o && rays_d && t && ridx)) return 24;
  a.sdf_w = sdf_w; a.sdf_b = sdf_b;
  a.x = x; a.rays_o = rays_o; a.rays_d = rays_d; a.t = t; a.ridx = ridx;
  a.S = S; a.PS = plane_pitch > 0 ? plane_pitch : NSIM_PLANE_PITCH(S);
  a.h_pl = h_planes; a.J_pl = J_planes;
  a.dsdf = dsdf; a.dnablas = dnablas;
  a.dh_pl = dh_planes; a.g_pl = g_planes;
  a.dsdf_w = dsdf_w; a.dsdf_b = dsdf_b;
  wide_launch<2>(a, (hipStream_t)stream);
  NSIM_CHECK_LAUNCH();
  return 0;
}

}  // extern "C"
