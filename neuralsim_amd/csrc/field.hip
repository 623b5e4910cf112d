// field.hip -- the fused NeuS field kernels for gfx950: LoTD gather -> SDF decoder MLP (+ analytic
// normals) -> radiance MLP, forward and backward (including the second-order terms of the normals), with
// every dense contraction on the matrix cores.
//
// Replaces the native half of nr3d_lib's LoTDNeuSModel.forward_sdf / forward_sdf_nablas / radiance query
// and their autograd backward (call sites: app/renderers/single_volume_renderer.py:244-246,
// code_single/tools/inspect_rendering.py:120-128; normals mechanism
// docs/exps/exp_permuto_3d_modulated.py:63-76; network shapes
// code_single/configs/object_centric/lotd_neus.dtu.230814.yaml:92-139; second-order requirement
// app/loss/eikonal.py:216-251).  The executable spec is oracle/field.py.
//
// MI355X design (not a translation of any CUDA kernel):
//  * one 64-lane wave owns a tile of 32 points; lane (j = l&31, hi = l>>5) owns point j and HALF of the
//    feature / hidden units -- exactly the C/D fragment of v_mfma_f32_32x32x*: register r of M-tile m on
//    lane-half hi holds unit U(m,r,hi) = 32m + (r&3) + 8(r>>2) + 4hi;
//  * every layer is computed TRANSPOSED (Out^T[units x points] = W . In^T): weights are the A operand,
//    activations the B operand.  Because the K slots of A and B are indexed identically by (l>>5, e), a
//    lane's own accumulator registers ARE its B fragment for the next layer -- the whole
//    gather -> MLP -> dMLP/dh -> normals -> radiance chain runs in registers, no LDS round trips;
//  * the lane-half that owns feature U(0,r,hi) also gathers it: each lane gathers 8 of the 16 levels of
//    its point (64 corner loads of one half2 each) and keeps d feature / d x (48 VGPRs) for the normals;
//  * weight gradients contract over points, so they (and only they) need a transpose: activations are
//    staged [unit][point] in LDS (pitch 40 halfs => conflict-free ds_read_b128), multiplied on the MFMA
//    and accumulated per workgroup in LDS f32 atomics, then flushed once per block with global atomics;
//  * fp16 MFMA operands are re-scaled per tile by an exact power of two (dyn_scale) so that the tiny
//    upstream gradients of a volume-render loss do not underflow -- no global loss scaler is required;
//  * PREC=1 selects v_mfma_f32_32x32x2_f32 (exact f32 at the vector rate) for tight parity tests.
#ifndef NSIM_SCATTER_SCAN_EXIT
#define NSIM_SCATTER_SCAN_EXIT 1
#endif
#include "lotd_dev.h"
#include "mfma_mlp.h"
#include "occ_dev.h"
#include <stdlib.h>

// ----------------------------------------------------------------------------------------- layout
enum { M_W1 = 0, M_W2, M_W2T, M_W1T, M_R1, M_R2, M_R3, M_R3T, M_R2T, M_R1T, M_COUNT };
enum { V_B1 = 0, V_B2, V_WH, V_RB1, V_RB2, V_RB3, V_SCAL, V_COUNT };

struct FieldLayout {
  int64_t mat[M_COUNT];  // byte offsets
  int64_t vec[V_COUNT];
  int64_t total;
  int elt;               // bytes per matrix element (2 | 4)
  int split;             // 1: every matrix is TWO f16 fragment sets, W = hi + lo / SPLIT_LO_SCALE (precision 2)
};

static const int kMatUo[M_COUNT] = {64, 64, 64, 32, 64, 64, 32, 64, 64, 32};
static const int kMatUi[M_COUNT] = {32, 64, 64, 64, 32, 64, 64, 32, 64, 64};

// nc = number of 32-input chunks of the decoder's first layer: 16-level feature chunks (1: <= 16 levels, 2: 17..32 levels)
// + ne embedded-position chunks (0, or 2 with NsimFieldMeta.embed_E > 0).  Only the first layer's matrices grow: W1 is
// [64 x 32 nc], W1T [32 nc x 64].
static inline int mat_uo(int m, int nc) { return m == M_W1T ? 32 * nc : kMatUo[m]; }
static inline int mat_ui(int m, int nc) { return m == M_W1 ? 32 * nc : kMatUi[m]; }
static inline int field_nc(int num_levels) { return num_levels > 16 ? 2 : 1; }
static inline int field_ne(const NsimFieldMeta* meta) { return meta->embed_E > 0 ? 2 : 0; }
static inline int field_ni(const NsimFieldMeta* meta) { return field_nc(meta->lotd.num_levels) + field_ne(meta); }

static inline FieldLayout field_layout(int precision, int nc = 1) {
  FieldLayout L;
  L.elt = precision == 0 ? 2 : 4;
  L.split = precision == 2 ? 1 : 0;
  int64_t off = 0;
  for (int m = 0; m < M_COUNT; ++m) {
    L.mat[m] = off;
    off += (int64_t)mat_uo(m, nc) * mat_ui(m, nc) * L.elt;
  }
  for (int v = 0; v < V_COUNT; ++v) {
    L.vec[v] = off;
    off += 64 * 4;
  }
  L.total = off;
  return L;
}

// flat f32 master-weight layouts (shared with the gradient buffers)
struct SrcOff {
  int w1, w2, wh, b1, b2, bh;   // sdf_w / sdf_b
  int r1, r2, r3, rb1, rb2, rb3;
  int n_sdf_w, n_sdf_b, n_rad_w, n_rad_b;
};
// F1 = 2 * num_levels: width of the decoder input (<= 32 nc; the columns up to 32 nc are zero padding in the pack)
__host__ __device__ inline SrcOff src_off(int D, int F1 = 32) {
  SrcOff o;
  o.w1 = 0;
  o.w2 = 64 * F1;
  o.wh = (D == 2) ? 64 * F1 + 4096 : 64 * F1;
  o.n_sdf_w = o.wh + 64;
  o.b1 = 0;
  o.b2 = 64;
  o.bh = (D == 2) ? 128 : 64;
  o.n_sdf_b = o.bh + 1;
  o.r1 = 0;
  o.r2 = 64 * 26;
  o.r3 = 64 * 26 + 4096;
  o.n_rad_w = o.r3 + 192;
  o.rb1 = 0;
  o.rb2 = 64;
  o.rb3 = 128;
  o.n_rad_b = 131;
  return o;
}


// ------------------------------------------------------------------------------------ weight packing
// precision 2 ("split"): an f32 value travels through the f16 matrix cores as v = hi + lo / SPLIT_LO_SCALE with
// hi = f16(v), lo = f16((v - hi) * SPLIT_LO_SCALE) -- 22 significant bits; a product W . x is three MFMAs (hi.hi into the
// main accumulator, hi.lo + lo.hi into a correction accumulator that is folded in with 1 / SPLIT_LO_SCALE), the
// dropped lo.lo term is 2^-22 relative.  The scale keeps the residuals out of the f16 subnormals.
#define SPLIT_LO_SCALE 2048.0f
// (saturating: a feature beyond the f16 range -- a street table holds heights in metres, x SDF_H_SCALE -- becomes +-65504, a
// point that far from every surface, instead of inf and then NaN through the products)
__device__ __forceinline__ float f16_sat(float v) { return fminf(fmaxf(v, -65504.0f), 65504.0f); }
__device__ __forceinline__ void split_f16(float v, f16& hi, f16& lo) {
  v = f16_sat(v);
  hi = (f16)v;
  lo = (f16)((v - (float)hi) * SPLIT_LO_SCALE);
}

// first-layer input i of the packed matrices -> column of W1 (row length FIN = F1 + E), -1 = zero padding: the feature
// chunks hold columns [0, F1), the embedded-position chunks (from input EB = 32 x feature chunks on) columns [F1, F1 + E)
__host__ __device__ inline int w1_col(int i, int F1, int EB, int E) {
  if (i < EB || E == 0) return i < F1 ? i : -1;
  return (i - EB) < E ? F1 + (i - EB) : -1;
}

// F1 = 2 num_levels feature inputs, E embedded-position inputs behind them (0: none), EB = 32 x feature chunks
__device__ __forceinline__ float pack_src(int mat, int row, int col, int D, int F1, int E, int EB, const float* sdf_w,
                                          const float* rad_w) {
  const int FIN = F1 + E;
  const SrcOff o = src_off(D, FIN);
  switch (mat) {
    case M_W1: { const int c = w1_col(col, F1, EB, E); return c >= 0 ? sdf_w[o.w1 + row * FIN + c] : 0.f; }
    case M_W2: return D == 2 ? sdf_w[o.w2 + row * 64 + col] : 0.f;
    case M_W2T: return D == 2 ? sdf_w[o.w2 + col * 64 + row] : 0.f;
    case M_W1T: { const int c = w1_col(row, F1, EB, E); return c >= 0 ? sdf_w[o.w1 + col * FIN + c] : 0.f; }
    case M_R1: return col < 26 ? rad_w[o.r1 + row * 26 + col] : 0.f;
    case M_R2: return rad_w[o.r2 + row * 64 + col];
    case M_R3: return row < 3 ? rad_w[o.r3 + row * 64 + col] : 0.f;
    case M_R3T: return col < 3 ? rad_w[o.r3 + col * 64 + row] : 0.f;
    case M_R2T: return rad_w[o.r2 + col * 64 + row];
    case M_R1T: return row < 26 ? rad_w[o.r1 + col * 26 + row] : 0.f;
  }
  return 0.f;
}

struct PackDims {
  int uo[M_COUNT], ui[M_COUNT];
  int E, EB;      // embedded-position inputs of the first layer (0: none) and the input index they start at
};

__device__ __forceinline__ void field_pack_elem(const FieldLayout& L, const PackDims& dims, int D, int F1,
                                                const float* __restrict__ sdf_w, const float* __restrict__ sdf_b,
                                                const float* __restrict__ rad_w, const float* __restrict__ rad_b,
                                                char* __restrict__ wpack, int64_t tid) {
  // matrices: one thread per element
  int64_t base = 0;
  for (int m = 0; m < M_COUNT; ++m) {
    const int Uo = dims.uo[m], Ui = dims.ui[m];
    const int64_t cnt = (int64_t)Uo * Ui;
    if (tid >= base && tid < base + cnt) {
      const int64_t k = tid - base;
      int row, col;
      if (L.elt == 2 || L.split) {
        const int e = (int)(k & 7), lane = (int)((k >> 3) & 63);
        const int fs = (int)(k >> 9);
        const int nS = Ui / 16;
        const int mo = fs / nS, s = fs % nS;
        row = 32 * mo + (lane & 31);
        col = 16 * s + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
        const float w = pack_src(m, row, col, D, F1, dims.E, dims.EB, sdf_w, rad_w);
        const f16 whi = (f16)w;
        ((f16*)(wpack + L.mat[m]))[k] = whi;
        if (L.split) ((f16*)(wpack + L.mat[m]))[cnt + k] = (f16)((w - (float)whi) * SPLIT_LO_SCALE);
      } else {
        const int lane = (int)(k & 63);
        const int fr = (int)(k >> 6);
        const int r = fr & 15, fm = fr >> 4;
        const int nMi = Ui / 32;
        const int mo = fm / nMi, mi = fm % nMi;
        row = 32 * mo + (lane & 31);
        col = unit_of(mi, r, lane >> 5);
        ((float*)(wpack + L.mat[m]))[k] = pack_src(m, row, col, D, F1, dims.E, dims.EB, sdf_w, rad_w);
      }
      return;
    }
    base += cnt;
  }
  // vectors: per-lane order [hi][m*16 + r]
  const int64_t vtid = tid - base;
  if (vtid >= 0 && vtid < (int64_t)V_COUNT * 64) {
    const int v = (int)(vtid >> 6), k = (int)(vtid & 63);
    const int hi = k >> 5, m = (k >> 4) & 1, r = k & 15;
    const int u = unit_of(m, r, hi);
    const SrcOff o = src_off(D, F1 + dims.E);
    float val = 0.f;
    switch (v) {
      case V_B1: val = sdf_b[o.b1 + u]; break;
      case V_B2: val = D == 2 ? sdf_b[o.b2 + u] : 0.f; break;
      case V_WH: val = sdf_w[o.wh + u]; break;
      case V_RB1: val = rad_b[o.rb1 + u]; break;
      case V_RB2: val = rad_b[o.rb2 + u]; break;
      case V_RB3: val = (u < 3) ? rad_b[o.rb3 + u] : 0.f; break;
      case V_SCAL: val = (k == 0) ? sdf_b[o.bh] : 0.f; break;
    }
    ((float*)(wpack + L.vec[v]))[k] = val;
  }
}

__global__ void __launch_bounds__(256) k_field_pack(FieldLayout L, PackDims dims, int D, int F1, const float* __restrict__ sdf_w,
                                                     const float* __restrict__ sdf_b, const float* __restrict__ rad_w,
                                                     const float* __restrict__ rad_b, char* __restrict__ wpack) {
  field_pack_elem(L, dims, D, F1, sdf_w, sdf_b, rad_w, rad_b, wpack, (int64_t)blockIdx.x * blockDim.x + threadIdx.x);
}

// two packs (field precision + sampling precision) of the same weights: blockIdx.y selects the pack
struct PackTwo {
  FieldLayout L[2];
  PackDims dims[2];
  char* out[2];
};
__global__ void __launch_bounds__(256) k_field_pack2(PackTwo p, int D, int F1, const float* __restrict__ sdf_w,
                                                      const float* __restrict__ sdf_b, const float* __restrict__ rad_w,
                                                      const float* __restrict__ rad_b) {
  const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (blockIdx.y == 0) field_pack_elem(p.L[0], p.dims[0], D, F1, sdf_w, sdf_b, rad_w, rad_b, p.out[0], tid);
  else field_pack_elem(p.L[1], p.dims[1], D, F1, sdf_w, sdf_b, rad_w, rad_b, p.out[1], tid);
}

// ------------------------------------------------------------------------------------ activations
// softplus(z; beta) = max(z, 0) + log(1 + exp(-beta |z|)) / beta on the raw base-2 pipes: two quarter-rate
// transcendentals + five full-rate ops per value (the libm-style __expf / __logf forms and torch's explicit threshold
// select cost twelve).  torch's ``threshold`` (beta z > 20 -> z) is implied: there the log term is < 2.1e-9 / beta,
// below half an ulp of z >= 0.2, so the sum rounds to z.  This is THE VALU cost of the decoder kernels (64 per point
// and pass against 12 MFMAs per 32 points).
#define NSIM_LOG2E 1.4426950408889634f
#define NSIM_LN2 0.6931471805599453f
// relu (``decoder_cfg.activation: relu``, no_fg_occ.221218.yaml:354-357; meta.softplus_beta < 0) runs through the SAME
// formulas with beta = 1e30 (field_args): the log term is scaled by 1 / beta -> max(z, 0); sigma = 1 - exp2(-1e30 a) is
// exactly [a > 0]; the curvature terms of the backward, written as beta s (1 - s), are 1e30 x 0.  No branch, no second
// code path (a wave-uniform branch here cost the 17..32-level backward 17 % in registers held across both arms).
__device__ __forceinline__ float softplus_exact(float z, float beta, float inv_beta) {
  const float t = nsim_exp2(-fabsf(z) * (beta * NSIM_LOG2E));
  return fmaxf(z, 0.f) + nsim_log2(1.0f + t) * (inv_beta * NSIM_LN2);
}
#ifdef NSIM_PROBE_CHEAP_ACT
// MEASUREMENT AID, never the product build (tools/act_probe.sh): the with-grad decoder kernels with a transcendental-free
// stand-in of comparable full-rate cost -- the time they lose is the most ANY polynomial activation could win.  The
// sampling pass (k_field_sdf) keeps the exact form, so the sample sets of a step do not change.
__device__ __forceinline__ float softplus_b(float z, float beta, float inv_beta) {
  const float q = fmaxf(0.f, 1.0f - 0.25f * beta * fabsf(z));
  return fmaxf(z, 0.f) + q * q * (inv_beta * NSIM_LN2);
}
__device__ __forceinline__ float sig_from_softplus(float a, float beta) { return fminf(1.0f, 0.72f * a * beta); }
#else
__device__ __forceinline__ float softplus_b(float z, float beta, float inv_beta) { return softplus_exact(z, beta, inv_beta); }
// sigma(beta z) recovered from a = softplus(z):  1 - exp(-beta a)   (abs. error <= 6e-8)
__device__ __forceinline__ float sig_from_softplus(float a, float beta) { return 1.0f - nsim_exp2(-a * (beta * NSIM_LOG2E)); }
#endif

__device__ __forceinline__ void sh4_eval(const float d[3], float (&o)[16]) {
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

// out[c] = sum_k g[k] * d sh4_k / d d_c   (pose refinement: gradient w.r.t. the view direction)
__device__ __forceinline__ void sh4_grad(const float d[3], const float (&g)[16], float (&out)[3]) {
  const float x = d[0], y = d[1], z = d[2];
  const float x2 = x * x, y2 = y * y, z2 = z * z;
  const float a1 = 0.48860251190291987f, b = 1.0925484305920792f, c1 = 0.94617469575755997f;
  const float e = 0.54627421529603959f, f = 0.59004358992664352f, gg = 2.8906114426405538f;
  const float h = 0.45704579946446572f, i3 = 0.3731763325901154f, jj = 1.4453057213202769f;
  out[0] = -a1 * g[3] + b * y * g[4] - b * z * g[7] + 2.0f * e * x * g[8] - 6.0f * f * x * y * g[9] + gg * y * z * g[10] +
           h * (1.0f - 5.0f * z2) * g[13] + 2.0f * jj * z * x * g[14] + f * (-3.0f * x2 + 3.0f * y2) * g[15];
  out[1] = -a1 * g[1] + b * x * g[4] - b * z * g[5] - 2.0f * e * y * g[8] + f * (-3.0f * x2 + 3.0f * y2) * g[9] +
           gg * x * z * g[10] + h * (1.0f - 5.0f * z2) * g[11] - 2.0f * jj * z * y * g[14] + 6.0f * f * x * y * g[15];
  out[2] = a1 * g[2] - b * y * g[5] + 2.0f * c1 * z * g[6] - b * x * g[7] + gg * x * y * g[10] - 10.0f * h * y * z * g[11] +
           i3 * (15.0f * z2 - 3.0f) * g[12] - 10.0f * h * x * z * g[13] + jj * (x2 - y2) * g[14];
}

// Plane arrays hold 16 NC levels.  NC == 2 (17..32 levels): the gather writes the pyramid's own levels only and every
// plane access of the decoders is guarded (a 19-level street pyramid moves 19 / 32 of the bytes).  NC == 1: the gather
// writes zeros into the planes past num_levels and the guards fold away (measured: run-time guards cost the full 16-level
// headline kernels 9-10 %).  ``nlv`` = a.lotd.num_levels in scope.
#define LV_OK(l) (NC == 1 || (l) < nlv)

// ------------------------------------------------------------------------------------------ kernels
struct FieldArgs {
  LotdDev lotd;
  FieldLayout lay;
  float beta;
  const f16* grid;
  const char* wpack;
  const float *x, *rays_o, *rays_d, *t;
  const int64_t* ridx;
  const int64_t* ray_goff;                         // batched model: table offset (in scalars, even) of ray r's instance
  const float* h_appear;
  int64_t S;
  const int64_t* S_dev;                            // no-grad query with speculatively sized buffers: the number of valid
  int64_t S_add;                                   // points is min(S, *S_dev + S_add); S stays the plane pitch
  float *sdf, *nablas, *rgb;                       // forward outputs
  const float *nablas_fwd, *rgb_fwd;               // saved forward outputs (radiance backward)
  const float *dsdf, *dnablas, *drgb;              // upstream gradients
  float* dnab_total;                               // [S,3] scratch: dnablas + d(radiance)/d nablas
  float *dx, *dv;                                  // pose refinement: dL/d(sample position) / dL/d(view dir) [S,3], or NULL
  void* feat_pl;                                   // no-grad SDF query: level-major feature planes [16 nc][S] x (f16x2 | f32x2)
  signed char glm_n[8], glm_lv[8][32];             // levels gathered by the blocks of XCD x (blockIdx % 8) ...
  signed char glm_half[8][32];                     // ... for all points (0), the first (1) or the second (2) half of them
  float* h_pl;                                     // level-major planes [16 nc][PS][2] saved by the forward
  void* J_pl;                                      // ... and [16 nc][PS][2][3] of JPlane<precision>::T (f16 | f32)
  int64_t PS;                                      // ... their pitch: S rounded up to 32 (NSIM_PLANE_PITCH)
  float *dh_pl, *g_pl;                             // backward -> scatter hand-off planes [16 nc][S][2]
  float* occ_val;                                  // no-grad query of a training step: fold f(sdf) into this value grid
  OccDev occ;                                      // ... (update_from_samples_cfg; needs positions in ``x``)
  float occ_inv_s;
  int ablate;                                      // profiling aid (NSIM_ABLATE): 1 no scatter, 4 no dW products, 8 no dh_appear atomics
  int code_pf;                                     // bytes of this kernel's own code pulled into L2 by the prologue (stage_weights)
  int rep_mask;                                    // weight-gradient replicas (nsim_set_grad_scratch): workgroup b flushes into
  int64_t rep_stride;                              // copy (b & rep_mask) at + rep_stride floats; 0 / 0 = the caller's buffers
  float *dgrid, *dsdf_w, *dsdf_b, *drad_w, *drad_b, *dh_appear;
  int has_rgb;
  int embed_E;                                     // embedded-position inputs of the SDF decoder's first layer (k_field / k_field_bwd_j NE = 2)
};

// LDS accumulator layouts (floats)
struct AccOff {
  int w1, w2, wh, b1, b2, bh, total;
};
__host__ __device__ inline AccOff acc_off(int nc = 1) {
  AccOff a;
  int o = 0;
  a.w1 = o; o += 64 * 32 * nc;
  a.w2 = o; o += 64 * 64;
  a.wh = o; o += 64;
  a.b1 = o; o += 64;
  a.b2 = o; o += 64;
  a.bh = o; o += 4;
  a.total = o;
  return a;
}
struct RadAccOff {
  int r1, r2, r3, rb1, rb2, rb3, total;
};
__host__ __device__ inline RadAccOff rad_acc_off() {
  RadAccOff a;
  int o = 0;
  a.r1 = o; o += 64 * 26;
  a.r2 = o; o += 64 * 64;
  a.r3 = o; o += 3 * 64;
  a.rb1 = o; o += 64;
  a.rb2 = o; o += 64;
  a.rb3 = o; o += 4;
  a.total = o;
  return a;
}


#define FIELD_WAVES 4
#ifndef NSIM_BWD_DBUF
#define NSIM_BWD_DBUF 0         // k_field_bwd_j, fp16: 1 = two alternating staging sets for the weight-gradient products (4 instead of 8 barriers
                                // per group).  Measured null on MI355X (nsim_field_bwd_sdf 0.1208 vs 0.1210 ms, street 0.6419 vs 0.6418;
                                // gpurun_out/r6_s2_call5) at +36 registers: the barriers are not where the kernel's time goes
#endif
#ifndef NSIM_BWD_JDIRECT
#define NSIM_BWD_JDIRECT 1      // k_field_bwd_j, <= 16 levels, fp16 mode: features through the LDS image, dh/dx as packed pairs straight into registers
#endif
#ifndef NSIM_FWD_JPACKED
#define NSIM_FWD_JPACKED 1      // ... and held there as the packed pairs of the planes until the normals are formed (0: converted at the load)
#endif
#ifndef NSIM_FWD_JDIRECT
#define NSIM_FWD_JDIRECT 1      // k_field MODE 3, <= 16 levels: features through the LDS image, dh/dx straight into registers
#endif

// Development aid (-DNSIM_KTIME, never in the product build): s_memtime stamps of wave 0 of the first 64 workgroups at
// phase boundaries of their SECOND group iteration, read back by tools/ktime.py through nsim_debug_ktime.
#ifdef NSIM_KTIME
__device__ long long g_ktime[3][64 * 24];
__device__ long long g_ktime1[3][64 * 24];      // the FIRST group iteration of the same workgroups (cold pass)
__device__ long long g_kiter[3][64 * 32];       // loop-top stamp of EVERY group iteration (first 32) of the same workgroups
#define KT(K, i)                                                                                          \
  if (blockIdx.x < 64 && wave == 0 && lane == 0) {                                                        \
    if ((i) == 0 && grp >= 0 && (grp - (int64_t)blockIdx.x) / (int64_t)gridDim.x < 32)                    \
      g_kiter[K][blockIdx.x * 32 + (grp - (int64_t)blockIdx.x) / (int64_t)gridDim.x] =                   \
          (long long)__builtin_amdgcn_s_memtime();                                                        \
    if (grp == (int64_t)blockIdx.x + gridDim.x || (i) >= 20)                                              \
      g_ktime[K][blockIdx.x * 24 + (i)] = (long long)__builtin_amdgcn_s_memtime();                        \
    else if (grp == (int64_t)blockIdx.x)                                                                  \
      g_ktime1[K][blockIdx.x * 24 + (i)] = (long long)__builtin_amdgcn_s_memtime();                       \
  }
extern "C" int nsim_debug_ktime(long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_ktime), sizeof(g_ktime));
}
extern "C" int nsim_debug_kiter(long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_kiter), sizeof(g_kiter));
}
extern "C" int nsim_debug_ktime1(long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_ktime1), sizeof(g_ktime1));
}
#else
#define KT(K, i)
#endif

// Copy the MFMA fragments a kernel needs into LDS (fp16 mode) and return a layout whose offsets are relative to the
// LDS copy: matrices [first, first+count) are contiguous in the pack, the per-lane vectors follow.  The f32
// validation mode reads fragments from global/L2.
template <int PREC>
__device__ __forceinline__ const char* stage_weights(char* smem, const FieldArgs& a, int first, int count,
                                                     FieldLayout& Lk, int& lds_used) {
  Lk = a.lay;
  nsim_prefetch_own_code(a.code_pf, smem);
  if constexpr (PREC != 1) {
    const int64_t m0 = a.lay.mat[first];
    const int64_t m1 = (first + count < M_COUNT) ? a.lay.mat[first + count] : a.lay.vec[0];
    const int64_t mbytes = m1 - m0, vbytes = a.lay.total - a.lay.vec[0];
    const f16x8* src = reinterpret_cast<const f16x8*>(a.wpack + m0);
    f16x8* dst = reinterpret_cast<f16x8*>(smem);
    for (int i = threadIdx.x; i < (int)(mbytes >> 4); i += blockDim.x) dst[i] = src[i];
    const f16x8* vsrc = reinterpret_cast<const f16x8*>(a.wpack + a.lay.vec[0]);
    f16x8* vdst = reinterpret_cast<f16x8*>(smem + mbytes);
    for (int i = threadIdx.x; i < (int)(vbytes >> 4); i += blockDim.x) vdst[i] = vsrc[i];
    for (int m = 0; m < M_COUNT; ++m) Lk.mat[m] = a.lay.mat[m] - m0;
    for (int v = 0; v < V_COUNT; ++v) Lk.vec[v] = mbytes + (a.lay.vec[v] - a.lay.vec[0]);
    lds_used = (int)((mbytes + vbytes + 15) & ~15);
    __syncthreads();
    return smem;
  } else {
    lds_used = 0;
    return a.wpack;
  }
}

__device__ __forceinline__ float vecf(const char* W, const FieldLayout& L, int v, int hi, int k) {
  return reinterpret_cast<const float*>(W + L.vec[v])[hi * 32 + k];
}

struct alignas(8) nsim_f2 { float x, y; };
// the six dh/dx values of one (level, point) -- d f0 / dx, d f1 / dx -- from a plane or its LDS image: three aligned pair loads
// (f16: three 32-bit loads + unpack instead of six 16-bit ones)
template <class T>
struct alignas(2 * sizeof(T)) JPair;
template <class T>
__device__ __forceinline__ void jload6(const T* p, float (&a)[3], float (&b)[3]);
// two consecutive dh/dx plane elements as one aligned load (8 bytes f32 | 4 bytes f16)
template <class T>
struct alignas(2 * sizeof(T)) JPair {
  T x, y;
};
template <class T>
__device__ __forceinline__ void jstore6(T* p, const float (&a)[3], const float (&b)[3]) {
  JPair<T>* q = reinterpret_cast<JPair<T>*>(p);
  JPair<T> v0, v1, v2;
  if constexpr (sizeof(T) == 2) {   // f16 planes: dscale (R - 1) * (feature difference) of a fine level can pass 65504 -> saturate, never inf
    v0.x = (T)f16_sat(a[0]); v0.y = (T)f16_sat(a[1]); v1.x = (T)f16_sat(a[2]);
    v1.y = (T)f16_sat(b[0]); v2.x = (T)f16_sat(b[1]); v2.y = (T)f16_sat(b[2]);
  } else {
    v0.x = (T)a[0]; v0.y = (T)a[1]; v1.x = (T)a[2];
    v1.y = (T)b[0]; v2.x = (T)b[1]; v2.y = (T)b[2];
  }
  q[0] = v0; q[1] = v1; q[2] = v2;
}
template <class T>
__device__ __forceinline__ void jload6(const T* p, float (&a)[3], float (&b)[3]) {
  const JPair<T>* q = reinterpret_cast<const JPair<T>*>(p);
  const JPair<T> v0 = q[0], v1 = q[1], v2 = q[2];
  a[0] = (float)v0.x; a[1] = (float)v0.y; a[2] = (float)v1.x;
  b[0] = (float)v1.y; b[1] = (float)v2.x; b[2] = (float)v2.y;
}

struct TilePoint {
  float xx[3], vd[3];
  int64_t s, ray;
  uint32_t goff;      // instance offset into the table (batched model), 0 otherwise
  bool valid;
};

__device__ __forceinline__ TilePoint load_point(const FieldArgs& a, int64_t tile, int j, bool need_dir) {
  TilePoint p;
  p.s = tile * 32 + j;
  p.valid = p.s < a.S;
  p.ray = 0;
  p.goff = 0u;
  p.xx[0] = p.xx[1] = p.xx[2] = 0.f;
  p.vd[0] = p.vd[1] = 0.f;
  p.vd[2] = 1.f;
  if (p.valid) {
    if (a.x) {
      p.xx[0] = a.x[3 * p.s]; p.xx[1] = a.x[3 * p.s + 1]; p.xx[2] = a.x[3 * p.s + 2];
      if (a.ridx) p.ray = a.ridx[p.s];
    } else if (a.ridx) {      // (neither given: a launch that works on the saved planes only, e.g. the SDF-branch backward)
      p.ray = a.ridx[p.s];
      const float tt = a.t[p.s];
#pragma unroll
      for (int c = 0; c < 3; ++c) p.xx[c] = a.rays_o[3 * p.ray + c] + tt * a.rays_d[3 * p.ray + c];
    }
    if (need_dir && a.rays_d) {
#pragma unroll
      for (int c = 0; c < 3; ++c) p.vd[c] = a.rays_d[3 * p.ray + c];
    }
    if (a.ray_goff) p.goff = (uint32_t)a.ray_goff[p.ray];
  }
  return p;
}

// radiance input in activation-register order: slots 0-2 x, 3-18 SH4(view dir), 19-21 nablas, 22-25 appearance
__device__ __forceinline__ void make_rin(float (&rin)[16], const TilePoint& p, const float nab[3], const float* h_appear,
                                         int hi) {
  float sh[16];
  sh4_eval(p.vd, sh);
  float ha[4] = {0.f, 0.f, 0.f, 0.f};
  if (p.valid && h_appear) {
#pragma unroll
    for (int c = 0; c < 4; ++c) ha[c] = h_appear[4 * p.ray + c];
  }
  // register r holds slot U(0,r,0) on the hi == 0 lanes and U(0,r,1) = U(0,r,0) + 4 on the others: both candidates are
  // compile-time selections, one v_cndmask per register (a slot computed from the lane would cost a compare chain each)
  auto slot_val = [&](int slot) -> float {
    if (slot < 3) return p.xx[slot];
    if (slot < 19) return sh[slot - 3];
    if (slot < 22) return nab[slot - 19];
    if (slot < 26) return ha[slot - 22];
    return 0.f;
  };
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const float v0 = slot_val(unit_of(0, r, 0)), v1 = slot_val(unit_of(0, r, 1));
    rin[r] = hi ? v1 : v0;
  }
}

// ---------------------------------------------------------------------------------- embedded-position input block
// ``surface_cfg.extra_pos_embed_cfg{type: sinusoidal_legacy, n_frequencies: N}`` (no_fg_occ.221218.yaml:319-321): the SDF
// decoder's first layer also reads [x_n (3) | sin(2^k x_n) (3), cos(2^k x_n) (3), k = 0..N-1] (E = 3 + 6 N <= 63 values), x_n =
// the AABB-normalised position in [-1, 1].  On the matrix cores the block is two more 32-input chunks (NE = 2) behind the NC
// feature chunks, in activation-register order: register r of a lane holds input I(r, hi) = unit_of(r >> 4, r & 15, hi) of the
// block -- both candidates (hi = 0 / 1: I and I + 4) are compile-time, one select each.  Conventions fixed here (nr3d_lib is
// absent): no factor pi, x_n = 2 u - 1 with u the pyramid's unit coordinate; the arithmetic is the one of csrc/wide_field.hip's
// f32 decoder (the no-grad query of the same model), term for term.
struct EmbedSlot {
  int c, k;            // axis, octave
  bool lin, is_cos;    // the x_n entries | a cosine entry
};
__device__ __forceinline__ EmbedSlot embed_slot(int idx) {
  EmbedSlot e;
  e.lin = idx < 3;
  const int m = e.lin ? 0 : idx - 3;
  e.c = e.lin ? idx : m % 3;
  e.k = m / 6;
  e.is_cos = !e.lin && (m % 6) >= 3;
  return e;
}
// v[r]: this lane's 32 inputs of the block (zero past E and for an invalid point); d[r]: d v[r] / d x along the slot's OWN
// axis (the other two components are zero)
__device__ __forceinline__ void embed_eval(const FieldArgs& a, const float (&xx)[3], bool valid, int hi, float (&v)[32],
                                           float (&d)[32]) {
  float xn[3], dxn[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    xn[c] = 2.0f * (xx[c] * a.lotd.xs[c] + a.lotd.xb[c]) - 1.0f;
    dxn[c] = 2.0f * a.lotd.xs[c];
  }
  const float m = valid ? 1.f : 0.f;
#pragma unroll
  for (int r = 0; r < 32; ++r) {
    const int i0 = unit_of(r >> 4, r & 15, 0), i1 = i0 + 4;
    const EmbedSlot e0 = embed_slot(i0), e1 = embed_slot(i1);
    const float xc = hi ? xn[e1.c] : xn[e0.c], dx = hi ? dxn[e1.c] : dxn[e0.c];
    const float fr = hi ? (float)(1 << e1.k) : (float)(1 << e0.k);
    const bool lin = hi ? e1.lin : e0.lin, is_cos = hi ? e1.is_cos : e0.is_cos;
    const float ang = fr * xc;
    const float sn = nsim_sin(ang), cs = nsim_cos(ang);
    float val = is_cos ? cs : sn, dd = (is_cos ? -sn : cs) * fr * dx;
    if (lin) {
      val = xc;
      dd = dx;
    }
    const bool on = (hi ? i1 : i0) < a.embed_E;
    v[r] = on ? val * m : 0.f;
    d[r] = on ? dd * m : 0.f;
  }
}
// acc[c] += sum over the lane's slots on axis c of g[r] d[r]   (the block's share of the normals: nablas = J^T g)
__device__ __forceinline__ void embed_nablas(const float* g, const float (&d)[32], int hi, float (&acc)[3]) {
#pragma unroll
  for (int r = 0; r < 32; ++r) {
    const int i0 = unit_of(r >> 4, r & 15, 0);
    const int c0 = embed_slot(i0).c, c1 = embed_slot(i0 + 4).c;
    const float pr = g[r] * d[r];
    acc[c0] = acc[c0] + ((hi && c0 != c1) ? 0.f : pr);
    if (c0 != c1) acc[c1] = acc[c1] + (hi ? pr : 0.f);
  }
}
// gh[r] = d[r] gn[axis of slot r]   (the block's share of the input tangent along dL/dnablas)
__device__ __forceinline__ void embed_tangent(float* gh, const float (&d)[32], const float (&gn)[3], int hi) {
#pragma unroll
  for (int r = 0; r < 32; ++r) {
    const int i0 = unit_of(r >> 4, r & 15, 0);
    const int c0 = embed_slot(i0).c, c1 = embed_slot(i0 + 4).c;
    gh[r] = d[r] * (hi ? gn[c1] : gn[c0]);
  }
}

template <int PREC>
__device__ __forceinline__ void radiance_hidden(float (&r1)[32], float (&r2)[32], const float (&rin)[16], const char* W,
                                                const FieldLayout& L, int hi, const char* WM = nullptr,
                                                const FieldLayout* LMp = nullptr) {
  // W / L: per-lane vectors; WM / LM: matrix fragments when they live elsewhere (L2 instead of LDS)
  const char* Wm = WM ? WM : W;
  const FieldLayout& LM = LMp ? *LMp : L;
  dense<PREC, 2, 1>(r1, Wm + LM.mat[M_R1], rin, false);
#pragma unroll
  for (int k = 0; k < 32; ++k) r1[k] = fmaxf(r1[k] + vecf(W, L, V_RB1, hi, k), 0.f);
  dense<PREC, 2, 2>(r2, Wm + LM.mat[M_R2], r1, false);
#pragma unroll
  for (int k = 0; k < 32; ++k) r2[k] = fmaxf(r2[k] + vecf(W, L, V_RB2, hi, k), 0.f);
}

// MODE 3: MODE 1 on features / dh-dx planes already gathered level-major by k_lotd_gather_lm<., true> (training).
// MODE 0: sdf only; MODE 1: sdf + nablas (+ rgb); MODE 2: backward of the SDF branch (gradient w.r.t. grid and
// decoder weights given dL/dsdf and the TOTAL dL/dnablas, which already includes the radiance net's share).
// NC: 16-level feature chunks of the decoder input (2 for pyramids of 17..32 levels -- the first layer contracts over
// 64 features; only on the level-major planes, MODE 2 / 3).
// GL2 (MODE 3, NC == 2): the plane image of a tile (1 KB per level of the pyramid) is prefetched into LDS as in the
// 16-level kernel -- possible while weights + 4 images fit the 160 KB (pyramids of up to 23 levels: the street's 19).
// NE (MODE 3): 32-input chunks of the embedded-position block behind the feature chunks (0 | 2: NsimFieldMeta.embed_E > 0);
// such a forward reads the planes directly (no LDS image) and stages the forward matrices only.
template <int PREC, int SDF_D, int MODE, int NC = 1, bool GL2 = false, int NE = 0>
__global__ void __launch_bounds__(64 * FIELD_WAVES) k_field(FieldArgs a) {
  using JT = typename JPlane<PREC>::T;      // element type of the dh/dx planes
  NSIM_DYN_SMEM(smem);
  const int lane = nsim_lane(), j = lane & 31, hi = lane >> 5;
  const int wave = (int)(threadIdx.x >> 6);
  const float beta = a.beta, inv_beta = 1.0f / a.beta;
  FieldLayout L;
  int wbytes = 0;
  // fp16 backward: every wave owns a private copy of the weight-gradient accumulators in LDS (plain read-add-write;
  // LDS float atomics cost ~800 cycles per instruction) and reads the weight fragments from L2 instead of LDS to
  // make room for the four copies; the f32 validation mode keeps one shared accumulator with atomics.
  constexpr bool PRIV = (MODE == 2 && PREC == 0);
  constexpr bool FWD = (MODE == 1 || MODE == 3);          // forward with normals (+ radiance)
  constexpr bool FROM_PLANES = (MODE == 2 || MODE == 3);  // h / dh-dx come from the level-major planes
  static_assert(NC == 1 || FROM_PLANES, "more than 16 levels: level-major planes only");
  static_assert(NE == 0 || (MODE == 3 && !GL2), "the embedded-position block: forward on the planes (backward: k_field_bwd_j)");
  constexpr int NI = NC + NE;      // 32-input chunks of the first layer
  // NC == 2: a private accumulator copy is 34 KB -> three waves per workgroup fit the 160 KB of LDS
  constexpr int NW = (PRIV && NC == 2) ? 3 : FIELD_WAVES;
  // W / L: per-lane vectors (always LDS in fp16 mode); WM / LM: matrix fragments (LDS, or L2 when PRIV)
  // MODE 0 / 2 touch only the SDF decoder (W1, W2, W2T, W1T); MODE 1 also the radiance matrices
  const char* W = stage_weights<PREC>(smem, a, 0, PRIV ? 0 : (FWD ? (NE ? M_R3 + 1 : M_COUNT) : 4), L, wbytes);
  const char* WM = PRIV ? a.wpack : W;
  const FieldLayout LM = PRIV ? a.lay : L;

  float* accum = nullptr;
  char* stA = nullptr;
  char* stB = nullptr;
  const AccOff AO = acc_off(NC);
  constexpr int ACC_BYTES = (((6400 + 2048 * (NC - 1)) * 4 + 15) & ~15);      // >= AO.total floats, one copy
  if constexpr (MODE == 2) {
    const int ncopies = PRIV ? NW : 1;
    accum = reinterpret_cast<float*>(smem + wbytes + (PRIV ? wave * ACC_BYTES : 0));
    char* stbase = smem + wbytes + ncopies * ACC_BYTES + wave * stage_bytes_per_wave<PREC>();
    stA = stbase;
    stB = stbase + stage_bytes_per_wave<PREC>() / 2;
    if constexpr (PRIV) {
      for (int i = lane; i < AO.total; i += 64) accum[i] = 0.f;
    } else {
      for (int i = threadIdx.x; i < AO.total; i += blockDim.x) accum[i] = 0.f;
    }
    __syncthreads();
  }
  const float b_out = reinterpret_cast<const float*>(W + L.vec[V_SCAL])[0];
  const GridRef gref = grid_ref(a.grid);

  if (MODE == 3 && a.S_dev) {     // speculatively sized launch: the point count comes from device memory, a.S is the
    const int64_t sd = a.S_dev[0] + a.S_add;      // capacity (and a.PS the pitch); overflow: touch nothing
    a.S = sd <= a.S ? sd : 0;
  }
  const int64_t ntiles = (a.S + 31) / 32;
  const int64_t wstride = (int64_t)gridDim.x * NW;
  // MODE 3, <= 16 levels: the planes of the NEXT tile (16 KB: per level 256 B of h and 768 B of dh/dx, each one aligned
  // piece thanks to the 32-point pitch) are copied global -> LDS by 16 global_load_lds_dwordx4 while this tile computes;
  // half of a tile used to be the wait for these reads (all workgroups burst together at one wave per SIMD).
  static_assert(!GL2 || (MODE == 3 && NC == 2), "GL2 is the 17..32-level forward");
  constexpr bool GLDS = (MODE == 3 && (NC == 1 || GL2) && NE == 0);
  const int nlv = a.lotd.num_levels;      // plane levels past it are neither written by the gather nor read here
  // JDIR (<= 16 levels, NSIM_FWD_JDIRECT): the LDS image holds the FEATURES only (4 KB per wave, 4 copies of 4 levels each);
  // dh/dx -- consumed once, at the very end of the tile -- is loaded straight into registers after the image has been
  // read, so its latency hides behind the decoder.  LDS per workgroup 124 KB -> 76 KB: TWO workgroups per CU, two waves per
  // SIMD (the registers, 244 of 512, already allowed it).
  constexpr bool JDIR = GLDS && NC == 1 && NSIM_FWD_JDIRECT;
  char* pf = GLDS ? smem + wbytes + wave * (NC == 1 ? (JDIR ? 4096 : 16384) : 1024 * nlv) : nullptr;
  auto prefetch_planes = [&](int64_t tile_n) {
    const int64_t s0 = tile_n * 32;
    if constexpr (JDIR) {
#pragma unroll
      for (int i = 0; i < 4; ++i)      // lanes 16 k .. 16 k + 15: the 256 B of level 4 i + k
        nsim_glds16(a.h_pl + ((int64_t)(4 * i + (lane >> 4)) * a.PS + s0) * 2 + 4 * (lane & 15), pf + 1024 * i);
      return;
    }
#pragma unroll
    for (int l = 0; l < 16 * NC; ++l) {
      if (!LV_OK(l)) continue;
      // 16 bytes per lane: lanes 0..15 the 256 B of features, the next 48 (f32) | 24 (f16) lanes the tile's dh/dx
      constexpr int JL = 16 / (int)sizeof(JT);      // dh/dx elements per lane
      const void* src = lane < 16 ? (const void*)(a.h_pl + ((int64_t)l * a.PS + s0) * 2 + 4 * lane)
                                  : (const void*)(reinterpret_cast<const JT*>(a.J_pl) + ((int64_t)l * a.PS + s0) * 6 + JL * (lane - 16));
      if (lane < 16 + 192 / JL) nsim_glds16(src, pf + 1024 * l);
    }
  };
  if constexpr (GLDS) {
    const int64_t t0 = (int64_t)blockIdx.x * NW + wave;
    if (t0 < ntiles) prefetch_planes(t0);
  }
  for (int64_t tile = (int64_t)blockIdx.x * NW + wave; tile < ntiles; tile += wstride) {
    const int64_t grp = tile / NW;      // (stamps of -DNSIM_KTIME builds: the wave-0 tile sequence of this workgroup)
    (void)grp;
    if constexpr (MODE == 2) { KT(1, 0); }
    if constexpr (MODE == 3) { KT(2, 0); }
    // JDIR: the point's position / ray / direction (needed by the radiance head only) are requested AFTER the wait for the plane
    // image below -- in front of it that s_waitcnt vmcnt(0) exposed their (dependent: ridx -> rays) latency at the top of every tile
    constexpr bool JDIR_ = (MODE == 3 && NC == 1 && NE == 0 && NSIM_FWD_JDIRECT);
    TilePoint p;
    if constexpr (!JDIR_) p = load_point(a, tile, j, FWD);
    const int64_t s = tile * 32 + j;
    const bool valid = s < a.S;
    // ---------------------------------------------------------------- gather (8 of 16 levels per lane)
    // The backward does NOT gather again: the forward saved h and dh/dx as level-major planes
    // ([level][sample][..], coalesced across the 32 samples of a tile) -- 512 B per sample of sequential HBM
    // traffic instead of a second latency-bound random gather.
    float h[16 * NI];                               // [features (16 NC) | embedded position (16 NE)]
    float J[((NC == 1 && NE == 0) || GL2) ? 16 * NC : 1][3];     // NC == 2 without the LDS image (and NE: registers) re-reads dh/dx where it is consumed
    float ed[NE ? 32 : 1];                          // NE: x-derivative of the embedded-position inputs (own axis)
    constexpr bool JPK = JDIR && NSIM_FWD_JPACKED;
    JPair<JT> Jq[JPK ? 8 : 1][3];                   // JPK: this lane's (level, point) entries as stored
    if constexpr (JDIR) {
      nsim_wait_vm0();                          // this tile's image has landed
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
          const int l = 4 * q + 2 * hi + b, r0 = 4 * q + 2 * b;
          const float* hp = reinterpret_cast<const float*>(pf + 256 * l) + 2 * j;
          h[r0] = valid ? hp[0] : 0.f;
          h[r0 + 1] = valid ? hp[1] : 0.f;
        }
      nsim_wait_lgkm0();                        // every lane has read the image: the next copy may overwrite it
      p = load_point(a, tile, j, FWD);
      // dh/dx of a point past the end: the last point's (finite values; nothing of such a lane is stored)
      const int64_t sc = valid ? s : a.S - 1;
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
          const int l = 4 * q + 2 * hi + b, r0 = 4 * q + 2 * b;
          if constexpr (JPK) {      // held as stored -- three packed pairs -- until the normals are formed after the decoder: 24 registers in f16
            const JPair<JT>* jp = reinterpret_cast<const JPair<JT>*>(reinterpret_cast<const JT*>(a.J_pl) + ((int64_t)l * a.PS + sc) * 6);
            Jq[2 * q + b][0] = jp[0];
            Jq[2 * q + b][1] = jp[1];
            Jq[2 * q + b][2] = jp[2];
          } else {
            jload6(reinterpret_cast<const JT*>(a.J_pl) + ((int64_t)l * a.PS + sc) * 6, J[r0], J[r0 + 1]);
          }
        }
      if (tile + wstride < ntiles) prefetch_planes(tile + wstride);
    } else if constexpr (GLDS) {
      nsim_wait_vm0();                          // this tile's image has landed
#pragma unroll
      for (int m = 0; m < NC; ++m)
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
          const int l = 16 * m + 4 * q + 2 * hi + b, r0 = 16 * m + 4 * q + 2 * b;
          const float* hp = reinterpret_cast<const float*>(pf + 1024 * l) + 2 * j;
          const JT* jp = reinterpret_cast<const JT*>(pf + 1024 * l + 256) + 6 * j;
          const bool lv = valid && LV_OK(l);
          h[r0] = lv ? hp[0] : 0.f;
          h[r0 + 1] = lv ? hp[1] : 0.f;
          float ja[3], jb[3];
          jload6(jp, ja, jb);
#pragma unroll
          for (int c3 = 0; c3 < 3; ++c3) {
            J[r0][c3] = lv ? ja[c3] : 0.f;
            J[r0 + 1][c3] = lv ? jb[c3] : 0.f;
          }
        }
      nsim_wait_lgkm0();                        // every lane has read the image: the next copy may overwrite it
      if (tile + wstride < ntiles) prefetch_planes(tile + wstride);
    } else if constexpr (FROM_PLANES) {
      // all feature reads first, then dh/dx: the first layer needs h only, so the 384 B / point of dh/dx are still in
      // flight (vmcnt is in order) while the decoder starts
#pragma unroll
      for (int m = 0; m < NC; ++m)
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
          for (int b = 0; b < 2; ++b) {
            const int l = 16 * m + 4 * q + 2 * hi + b;
            const int r0 = 16 * m + 4 * q + 2 * b;
            h[r0] = h[r0 + 1] = 0.f;
            if (valid && LV_OK(l)) {
              const float* hp = a.h_pl + ((int64_t)l * a.PS + s) * 2;
              h[r0] = hp[0];
              h[r0 + 1] = hp[1];
            }
          }
      if constexpr (NC == 1 && NE == 0) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
          for (int b = 0; b < 2; ++b) {
            const int l = 4 * q + 2 * hi + b;
            const int r0 = 4 * q + 2 * b;
#pragma unroll
            for (int c3 = 0; c3 < 3; ++c3) J[r0][c3] = J[r0 + 1][c3] = 0.f;
            if (valid && LV_OK(l)) {
              jload6(reinterpret_cast<const JT*>(a.J_pl) + ((int64_t)l * a.PS + s) * 6, J[r0], J[r0 + 1]);
            }
          }
      }
    } else if constexpr (NC == 1) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
#pragma unroll
        for (int b = 0; b < 2; ++b) {
          const int l = 4 * q + 2 * hi + b;
          const LotdRes R = a.lotd.res[l];
          const LotdCell c = lotd_cell(p.xx, R, a.lotd);
          float f0 = 0.f, f1 = 0.f, j0[3] = {0.f, 0.f, 0.f}, j1[3] = {0.f, 0.f, 0.f};
          const int pm = lotd_slot_mask(c);
          if (l < a.lotd.n_active)      // hardmask annealing: masked levels are never read
#pragma unroll
          for (int slot = 0; slot < 8; ++slot) {
            const int corner = slot ^ pm;
            float w, dw[3];
            lotd_corner_w(c, corner, w, dw);
            const uint32_t idx = lotd_index(c.c0[0] + (corner & 1), c.c0[1] + ((corner >> 1) & 1),
                                            c.c0[2] + ((corner >> 2) & 1), R, a.lotd.type[l], a.lotd.size[l]);
            float g0, g1;
            lotd_load2(gref, (uint32_t)a.lotd.offset[l] + p.goff + 2u * idx, g0, g1);
            f0 = f0 + w * g0;
            f1 = f1 + w * g1;
            if (MODE >= 1) {
#pragma unroll
              for (int c3 = 0; c3 < 3; ++c3) {
                j0[c3] = j0[c3] + dw[c3] * g0;
                j1[c3] = j1[c3] + dw[c3] * g1;
              }
            }
          }
          const int r0 = 4 * q + 2 * b;
          h[r0] = f0;
          h[r0 + 1] = f1;
          if (MODE >= 1) {
#pragma unroll
            for (int c3 = 0; c3 < 3; ++c3) {
              J[r0][c3] = j0[c3] * c.dscale[c3];
              J[r0 + 1][c3] = j1[c3] * c.dscale[c3];
            }
            if (a.h_pl && valid) {
              float* hp = a.h_pl + ((int64_t)l * a.PS + s) * 2;
              JT* jp = reinterpret_cast<JT*>(a.J_pl) + ((int64_t)l * a.PS + s) * 6;
              hp[0] = f0;
              hp[1] = f1;
              jstore6(jp, J[r0], J[r0 + 1]);
            }
          }
        }
      }
    }
    if constexpr (NE > 0) embed_eval(a, p.xx, valid, hi, reinterpret_cast<float(&)[32]>(h[16 * NC]), ed);
    if constexpr (MODE == 2) { KT(1, 1); }
    if constexpr (MODE == 3) { KT(2, 1); }
    // ---------------------------------------------------------------- SDF decoder forward
    float a1[32];
    dense<PREC, 2, NI>(a1, WM + LM.mat[M_W1], h, true);
#pragma unroll
    for (int k = 0; k < 32; ++k) a1[k] = softplus_b(a1[k] + vecf(W, L, V_B1, hi, k), beta, inv_beta);
    float a2[32];  // last hidden activation (== a1 when SDF_D == 1)
    if constexpr (SDF_D == 2) {
      dense<PREC, 2, 2>(a2, WM + LM.mat[M_W2], a1, false);
#pragma unroll
      for (int k = 0; k < 32; ++k) a2[k] = softplus_b(a2[k] + vecf(W, L, V_B2, hi, k), beta, inv_beta);
    } else {
#pragma unroll
      for (int k = 0; k < 32; ++k) a2[k] = a1[k];
    }
    float sdf = 0.f;
#pragma unroll
    for (int k = 0; k < 32; ++k) sdf = sdf + vecf(W, L, V_WH, hi, k) * a2[k];
    sdf = sdf + wave_shfl_xor(sdf, 32);
    sdf = sdf + b_out;
    if constexpr (MODE == 0) {
      if (valid && hi == 0) a.sdf[s] = sdf;
      continue;
    }
    if constexpr (MODE == 3) { KT(2, 2); }
    // ---------------------------------------------------------------- d sdf / d h  (the "g chain")
    float e1[32];  // d sdf / d a1  (only SDF_D == 2)
    float d1[32];  // d sdf / d z1
    if constexpr (SDF_D == 2) {
      float d2[32];
#pragma unroll
      for (int k = 0; k < 32; ++k) d2[k] = sig_from_softplus(a2[k], beta) * vecf(W, L, V_WH, hi, k);
      dense<PREC, 2, 2>(e1, WM + LM.mat[M_W2T], d2, false);
#pragma unroll
      for (int k = 0; k < 32; ++k) d1[k] = sig_from_softplus(a1[k], beta) * e1[k];
    } else {
#pragma unroll
      for (int k = 0; k < 32; ++k) {
        e1[k] = vecf(W, L, V_WH, hi, k);
        d1[k] = sig_from_softplus(a1[k], beta) * e1[k];
      }
    }
    float g[16 * NI];
    dense<PREC, NI, 2>(g, WM + LM.mat[M_W1T], d1, false);
    if constexpr (FWD) {
      if constexpr (MODE == 3) { KT(2, 3); }
      float nab[3];
      if constexpr (JPK) {
        float acc[3] = {0.f, 0.f, 0.f};
#pragma unroll
        for (int e = 0; e < 8; ++e) {      // entry e = 2 q + b: features r0 = 4 q + 2 b (d f0 / dx) and r0 + 1 (d f1 / dx)
          const int r0 = 4 * (e >> 1) + 2 * (e & 1);
          acc[0] = acc[0] + g[r0] * (float)Jq[e][0].x + g[r0 + 1] * (float)Jq[e][1].y;
          acc[1] = acc[1] + g[r0] * (float)Jq[e][0].y + g[r0 + 1] * (float)Jq[e][2].x;
          acc[2] = acc[2] + g[r0] * (float)Jq[e][1].x + g[r0 + 1] * (float)Jq[e][2].y;
        }
#pragma unroll
        for (int c3 = 0; c3 < 3; ++c3) nab[c3] = acc[c3] + wave_shfl_xor(acc[c3], 32);
      } else if constexpr ((NC == 1 && NE == 0) || GL2) {
#pragma unroll
        for (int c3 = 0; c3 < 3; ++c3) {
          float acc = 0.f;
#pragma unroll
          for (int f = 0; f < 16 * NC; ++f) acc = acc + g[f] * J[f][c3];
          nab[c3] = acc + wave_shfl_xor(acc, 32);
        }
      } else {
        float acc[3] = {0.f, 0.f, 0.f};
        if (valid) {
#pragma unroll
          for (int m = 0; m < NC; ++m)
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
              for (int b = 0; b < 2; ++b) {
                const int l = 16 * m + 4 * q + 2 * hi + b;
                const int r0 = 16 * m + 4 * q + 2 * b;
                if (!LV_OK(l)) continue;
                float ja[3], jb[3];
                jload6(reinterpret_cast<const JT*>(a.J_pl) + ((int64_t)l * a.PS + s) * 6, ja, jb);
#pragma unroll
                for (int c3 = 0; c3 < 3; ++c3) acc[c3] = acc[c3] + g[r0] * ja[c3] + g[r0 + 1] * jb[c3];
              }
        }
#pragma unroll
        for (int c3 = 0; c3 < 3; ++c3) nab[c3] = acc[c3] + wave_shfl_xor(acc[c3], 32);
      }
      if constexpr (NE > 0) {      // the normals' share through the embedded position's own x-derivative
        float ea[3] = {0.f, 0.f, 0.f};
        embed_nablas(&g[16 * NC], ed, hi, ea);
#pragma unroll
        for (int c3 = 0; c3 < 3; ++c3) nab[c3] = nab[c3] + (ea[c3] + wave_shfl_xor(ea[c3], 32));
      }
      if constexpr (MODE == 3) { KT(2, 4); }
      float rgbv[3] = {0.f, 0.f, 0.f};
      if (a.has_rgb) {
        float rin[16], r1[32], r2[32];
        make_rin(rin, p, nab, a.h_appear, hi);
        radiance_hidden<PREC>(r1, r2, rin, W, L, hi);
        float o3[16];
        dense<PREC, 1, 2>(o3, WM + LM.mat[M_R3], r2, false);
#pragma unroll
        for (int c = 0; c < 3; ++c) {  // rows 0..2 live on the hi == 0 half in registers 0..2
          const float v = o3[c] + vecf(W, L, V_RB3, hi, c);
          rgbv[c] = 1.0f / (1.0f + nsim_fast_exp(-v));
        }
      }
      if constexpr (MODE == 3) { KT(2, 5); }
      if (valid && hi == 0) {
        a.sdf[s] = sdf;
#pragma unroll
        for (int c = 0; c < 3; ++c) a.nablas[3 * s + c] = nab[c];
        if (a.has_rgb && a.rgb) {
#pragma unroll
          for (int c = 0; c < 3; ++c) a.rgb[3 * s + c] = rgbv[c];
        }
      }
      if constexpr (MODE == 3) { KT(2, 6); KT(2, 17); }
      continue;
    }
    // ======================================================================================= backward
    if constexpr (MODE == 2) {
      KT(1, 2);
      float gs = 0.f, gn[3] = {0.f, 0.f, 0.f};
      if (valid) {
        if (a.dsdf) gs = a.dsdf[s];
        if (a.dnablas) {
#pragma unroll
          for (int c = 0; c < 3; ++c) gn[c] = a.dnablas[3 * s + c];
        }
      }
      // ------------------------------------------------------------ second-order path through the normals
      float gh[16 * NC];  // dL / dg
      if constexpr (NC == 1) {
#pragma unroll
        for (int f = 0; f < 16; ++f) gh[f] = J[f][0] * gn[0] + J[f][1] * gn[1] + J[f][2] * gn[2];
      } else {
#pragma unroll
        for (int m = 0; m < NC; ++m)
#pragma unroll
          for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int b = 0; b < 2; ++b) {
              const int l = 16 * m + 4 * q + 2 * hi + b;
              const int r0 = 16 * m + 4 * q + 2 * b;
              gh[r0] = gh[r0 + 1] = 0.f;
              if (valid && LV_OK(l)) {
                float ja[3], jb[3];
                jload6(reinterpret_cast<const JT*>(a.J_pl) + ((int64_t)l * a.PS + s) * 6, ja, jb);
                gh[r0] = ja[0] * gn[0] + ja[1] * gn[1] + ja[2] * gn[2];
                gh[r0 + 1] = jb[0] * gn[0] + jb[1] * gn[1] + jb[2] * gn[2];
              }
            }
      }
      KT(1, 3);
      float dh1[32];  // dL / d d1  = W1 . gh
      dense<PREC, 2, NC>(dh1, WM + LM.mat[M_W1], gh, true);
      KT(1, 4);
      const bool do_dw = !(a.ablate & 4);
      if (do_dw) dw_product<PREC, 2, NC, PRIV>(stA, stB, d1, gh, accum + AO.w1, 32 * NC, 64, 32 * NC, nullptr);
      KT(1, 5);
      float dz1[32];
      float whv[32];  // vector-shaped gradient of the SDF head weights
      if constexpr (SDF_D == 2) {
        float eh1[32];  // dL / d e1
#pragma unroll
        for (int k = 0; k < 32; ++k) {
          const float s1 = sig_from_softplus(a1[k], beta);
          dz1[k] = dh1[k] * e1[k] * (beta * s1 * (1.0f - s1));
          eh1[k] = dh1[k] * s1;
        }
        float d2[32];
#pragma unroll
        for (int k = 0; k < 32; ++k) d2[k] = sig_from_softplus(a2[k], beta) * vecf(W, L, V_WH, hi, k);
        KT(1, 6);
        if (do_dw) dw_product<PREC, 2, 2, PRIV>(stA, stB, d2, eh1, accum + AO.w2, 64, 64, 64, nullptr);
        KT(1, 7);
        float dh2[32];  // dL / d d2 = W2 . eh1
        dense<PREC, 2, 2>(dh2, WM + LM.mat[M_W2], eh1, true);
        float dz2[32];
#pragma unroll
        for (int k = 0; k < 32; ++k) {
          const float s2 = sig_from_softplus(a2[k], beta);
          const float wh = vecf(W, L, V_WH, hi, k);
          whv[k] = dh2[k] * s2 + gs * a2[k];
          dz2[k] = gs * wh * s2 + dh2[k] * wh * (beta * s2 * (1.0f - s2));
        }
        KT(1, 8);
        if (do_dw) dw_product<PREC, 2, 2, PRIV>(stA, stB, dz2, a1, accum + AO.w2, 64, 64, 64, accum + AO.b2);
        KT(1, 9);
        float da1[32];
        dense<PREC, 2, 2>(da1, WM + LM.mat[M_W2T], dz2, true);
        KT(1, 10);
#pragma unroll
        for (int k = 0; k < 32; ++k) dz1[k] = dz1[k] + da1[k] * sig_from_softplus(a1[k], beta);
      } else {
#pragma unroll
        for (int k = 0; k < 32; ++k) {
          const float s1 = sig_from_softplus(a1[k], beta);
          const float wh = vecf(W, L, V_WH, hi, k);
          whv[k] = dh1[k] * s1 + gs * a1[k];
          dz1[k] = gs * wh * s1 + dh1[k] * wh * (beta * s1 * (1.0f - s1));
        }
      }
      KT(1, 11);
      if (do_dw) rowsum_acc<PREC, 2, PRIV>(stA, whv, accum + AO.wh, 64);
      KT(1, 12);
      {
        float v = (hi == 0) ? gs : 0.f;
        v = wave_sum(v);
        if (lane == 0 && v != 0.f) {
          if constexpr (PRIV) accum[AO.bh] = accum[AO.bh] + v;
          else atomicAdd(&accum[AO.bh], v);
        }
      }
      if (do_dw) dw_product<PREC, 2, NC, PRIV>(stA, stB, dz1, h, accum + AO.w1, 32 * NC, 64, 32 * NC, accum + AO.b1);
      KT(1, 13);
      float dh[16 * NC];
      dense<PREC, NC, 2>(dh, WM + LM.mat[M_W1T], dz1, true);
      KT(1, 14);
      if (a.dx) {   // pose refinement: dL/dx += (dh/dx)^T dL/dh  (dh/dx re-read from the planes: this path is rare)
        float acc[3] = {0.f, 0.f, 0.f};
        if (valid) {
#pragma unroll
          for (int m = 0; m < NC; ++m)
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
              for (int b = 0; b < 2; ++b) {
                const int l = 16 * m + 4 * q + 2 * hi + b;
                const int r0 = 16 * m + 4 * q + 2 * b;
                if (!LV_OK(l)) continue;
                float ja[3], jb[3];
                jload6(reinterpret_cast<const JT*>(a.J_pl) + ((int64_t)l * a.PS + s) * 6, ja, jb);
#pragma unroll
                for (int c3 = 0; c3 < 3; ++c3) acc[c3] = acc[c3] + dh[r0] * ja[c3] + dh[r0 + 1] * jb[c3];
              }
        }
#pragma unroll
        for (int c3 = 0; c3 < 3; ++c3) acc[c3] = acc[c3] + wave_shfl_xor(acc[c3], 32);
        if (valid && hi == 0) {
#pragma unroll
          for (int c3 = 0; c3 < 3; ++c3) a.dx[3 * s + c3] = a.dx[3 * s + c3] + acc[c3];
        }
      }
      // ------------------------------------------------------------ hand-off to the scatter kernel
      // dL/dh and g = d sdf/d h as level-major planes + the total dL/dnablas per sample; k_lotd_scatter turns
      // them into grid gradients at full occupancy (it is bound by the atomic unit, not by this kernel's MFMA chain)
      if (valid && a.dh_pl) {
#pragma unroll
        for (int m = 0; m < NC; ++m)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
#pragma unroll
          for (int b = 0; b < 2; ++b) {
            const int l = 16 * m + 4 * q + 2 * hi + b;
            const int r0 = 16 * m + 4 * q + 2 * b;
            if (!LV_OK(l)) continue;             // the scatter reads the planes of real levels only
            float* dp = a.dh_pl + ((int64_t)l * a.S + s) * 2;
            float* gp = a.g_pl + ((int64_t)l * a.S + s) * 2;
            dp[0] = dh[r0];
            dp[1] = dh[r0 + 1];
            gp[0] = g[r0];
            gp[1] = g[r0 + 1];
          }
        }
      }
      KT(1, 15);
      KT(1, 16);
      KT(1, 17);
    }
  }

  if constexpr (MODE == 2) {
    __syncthreads();
    const int F1 = 2 * a.lotd.num_levels;
    const SrcOff so = src_off(SDF_D, F1);
    for (int i = threadIdx.x; i < AO.total; i += blockDim.x) {
      float v;
      if constexpr (PRIV) {      // sum the four private copies
        const float* a0 = reinterpret_cast<const float*>(smem + wbytes);
        v = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) v += a0[w * (ACC_BYTES / 4) + i];
      } else {
        v = accum[i];
      }
      if (v == 0.f) continue;
      float* dst = nullptr;
      if (i < AO.w2) {           // accumulator rows are 32 NC wide, the parameter rows F1 (<= 32 NC) wide
        const int row = (i - AO.w1) / (32 * NC), col = (i - AO.w1) % (32 * NC);
        dst = col < F1 ? a.dsdf_w + so.w1 + row * F1 + col : nullptr;
      } else if (i < AO.wh) dst = (SDF_D == 2) ? a.dsdf_w + so.w2 + (i - AO.w2) : nullptr;
      else if (i < AO.b1) dst = a.dsdf_w + so.wh + (i - AO.wh);
      else if (i < AO.b2) dst = a.dsdf_b + so.b1 + (i - AO.b1);
      else if (i < AO.bh) dst = (SDF_D == 2) ? a.dsdf_b + so.b2 + (i - AO.b2) : nullptr;
      else dst = (i == AO.bh) ? a.dsdf_b + so.bh : nullptr;
      if (dst) atomicAdd(dst, v);
    }
  }
}

// Backward of the SDF branch (the work of k_field<., ., 2>) with workgroup-joint weight gradients (mfma_mlp.h): per group of
// 4 x 32 points the waves stage the operands of the four weight-gradient products side by side; wave w owns the dW2 tile
// (w >> 1, w & 1) and the dW1 tile w & 1 over the point half w >> 1, both in MFMA accumulator registers for the whole
// launch; bias / head-weight sums are one register each.  LDS: W1, W2, W2T, W1T + vectors (26 KB) + staging 192 rows
// (51 KB) = 77 KB -> two workgroups per CU.  J (dh/dx, 48 registers in k_field) is consumed straight from its loads
// into dL/dg, h is re-read for the last product, g leaves for the scatter as soon as it exists: the live set fits
// 256 registers = two waves per SIMD (k_field<0,2,2>: 468 registers, 145 KB LDS -> one wave per SIMD).
// NC = 2 (17..32 levels, the street pyramids): the first layer contracts over two 16-level chunks, dW1 is 64 x 64 -> wave w
// owns its tile (w >> 1, w & 1) over all 128 points like dW2; h / dL/dg / g / dL/dh are 32 wide; dh/dx is consumed
// straight from its loads (never held: 96 registers), the next group's features are prefetched into registers (the LDS
// image of a 32-level tile would be 32 KB per wave).  Round 3: k_field<0,1,2,2> 1.76 ms -> this kernel on the street step.
// NE = 2 (NsimFieldMeta.embed_E > 0): the embedded-position block as two more 32-input chunks of the first layer -- its values
// and x-derivative are regenerated from the sample positions, its columns of dW1 are more accumulator tiles (column tiles are
// dealt in PAIRS: wave w owns tile (w >> 1, 2 p + (w & 1)) of pair p over all 128 points; an odd last column tile is split by point
// halves as for NC = 1), no hand-off planes and no dL/dx for it.
template <int PREC, int SDF_D, int NC = 1, int NE = 0>
__global__ void __launch_bounds__(64 * JOINT_WAVES) k_field_bwd_j(FieldArgs a) {
  constexpr int NI = NC + NE;                  // 32-input chunks of the first layer
  constexpr bool SP = !(NC == 2 && SDF_D == 1 && NE == 0);      // paired bf16 staging (mfma_mlp.h jstage): measured per instantiation
  constexpr int NPAIR = NI / 2, NODD = NI & 1; // dW1 column tiles: pairs + an odd last one
  using JT = typename JPlane<PREC>::T;      // element type of the dh/dx planes
  NSIM_DYN_SMEM(smem);
  const int lane = nsim_lane(), j = lane & 31, hi = lane >> 5;
  const int wave = (int)(threadIdx.x >> 6);
  const float beta = a.beta, inv_beta = 1.0f / a.beta;
  FieldLayout L;
  int wbytes = 0;
  { const int64_t grp = -1; KT(1, 23); }
  const char* W = stage_weights<PREC>(smem, a, 0, 4, L, wbytes);
  // Staging set = A (64 rows) + B (32 NI, at least 64) + C (64).  DB (fp16 mode without the embedded block: LDS allows it): TWO sets
  // used alternately by the group's weight-gradient products, so only the barrier between a product's writers and its readers
  // remains -- the set a product writes was last read two products ago, and every wave passed the barrier in between after
  // its reads (8 -> 4 barriers per group of 128 points; at one wave per SIMD a barrier costs the waves' drift).
  constexpr bool DB = NSIM_BWD_DBUF && PREC == 0 && NE == 0 && (NC == 2 || NSIM_BWD_JDIRECT);
  constexpr int SET_ROWS = 128 + (NI > 2 ? 32 * NI : 64);
  char* const st0 = smem + wbytes;
  auto set_base = [&](int k) -> char* { return st0 + (DB ? (k & 1) * SET_ROWS * jstage_row_bytes<PREC>() : 0); };
  char* stA = st0;
  char* stB = stA + 64 * jstage_row_bytes<PREC>();
  char* stC = stB + (NI > 2 ? 32 * NI : 64) * jstage_row_bytes<PREC>();
  auto use_set = [&](int k) {
    stA = set_base(k);
    stB = stA + 64 * jstage_row_bytes<PREC>();
    stC = stB + (NI > 2 ? 32 * NI : 64) * jstage_row_bytes<PREC>();
  };
  f32x16 accW1[NPAIR + NODD], accW2 = zero16();
#pragma unroll
  for (int q = 0; q < NPAIR + NODD; ++q) accW1[q] = zero16();
  // dW1 += A (x) B over the staged group, this wave's tiles
  auto dw1_tiles = [&](const void* sa, const void* sb) {
#pragma unroll
    for (int q = 0; q < NPAIR; ++q) accW1[q] = jdw_tile<PREC>(sa, wave >> 1, sb, 2 * q + (wave & 1), accW1[q]);
    if constexpr (NODD) {
      if (wave < 2) accW1[NPAIR] = jdw_tile<PREC, 0, JOINT_PTS / 32>(sa, wave & 1, sb, NI - 1, accW1[NPAIR]);
      else accW1[NPAIR] = jdw_tile<PREC, JOINT_PTS / 32, JOINT_PTS / 16>(sa, wave & 1, sb, NI - 1, accW1[NPAIR]);
    }
  };
  float bs1 = 0.f, bs2 = 0.f, bsh = 0.f, bh = 0.f;      // this wave's share of d b1[lane], d b2[lane], d wh[lane], d b_head
  const bool do_dw = !(a.ablate & 4);

  const int64_t ntiles = (a.S + 31) / 32;
  const int64_t ngroups = (ntiles + JOINT_WAVES - 1) / JOINT_WAVES;
  { const int64_t grp = -1; KT(1, 20); }
  // Software pipeline over the plane reads (512 B per point, the only bulk HBM traffic of this kernel): the features of
  // the NEXT group are requested while this group computes, and dh/dx -- needed only after the recomputed forward -- is
  // requested before it, so that the bursts of all workgroups no longer alternate with their compute phases.
  const int nlv = a.lotd.num_levels;      // plane levels past it are neither written by the gather nor read here
  auto load_h_at = [&](float (&h)[16 * NC], int64_t sp) {
    const bool v = sp < a.S;
#pragma unroll
    for (int m = 0; m < NC; ++m)
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        const int l = 16 * m + 4 * q + 2 * hi + b, r0 = 16 * m + 4 * q + 2 * b;
        h[r0] = h[r0 + 1] = 0.f;
        if (v && LV_OK(l)) {
          const float* hp = a.h_pl + ((int64_t)l * a.PS + sp) * 2;
          h[r0] = hp[0];
          h[r0 + 1] = hp[1];
        }
      }
  };
  // fp16 mode: the whole 16 KB plane image of a wave's NEXT tile (h and dh/dx) is copied global -> LDS while the group
  // computes (as in k_field MODE 3); f32 validation mode (its f32 staging leaves no LDS for it) prefetches h into registers
  constexpr bool GLDS = (PREC == 0 && NC == 1 && NE == 0);
  char* pf = GLDS ? st0 + (DB ? 2 : 1) * SET_ROWS * jstage_row_bytes<PREC>() + wave * (NSIM_BWD_JDIRECT ? 4096 : 16384) : nullptr;
  // BJD: the image holds the FEATURES only (4 copies of 4 levels each, 256 B per level); dh/dx -- consumed once, for dL/dg = J . gn,
  // AFTER the recomputed forward -- is loaded as packed pairs straight from the planes at the top of a group: 24 registers
  // (f16) instead of 48 converted floats live across the forward, no LDS round trip, the loads fly under the forward
  constexpr bool BJD = GLDS && NSIM_BWD_JDIRECT;
  auto prefetch_planes = [&](int64_t tile_n) {
    const int64_t s0 = tile_n * 32;
    if constexpr (BJD) {
#pragma unroll
      for (int i = 0; i < 4; ++i)      // lanes 16 k .. 16 k + 15: the 256 B of level 4 i + k
        nsim_glds16(a.h_pl + ((int64_t)(4 * i + (lane >> 4)) * a.PS + s0) * 2 + 4 * (lane & 15), pf + 1024 * i);
      return;
    }
#pragma unroll
    for (int l = 0; l < 16; ++l) {
      // 16 bytes per lane: lanes 0..15 the 256 B of features, the next 48 (f32) | 24 (f16) lanes the tile's dh/dx
      constexpr int JL = 16 / (int)sizeof(JT);      // dh/dx elements per lane
      const void* src = lane < 16 ? (const void*)(a.h_pl + ((int64_t)l * a.PS + s0) * 2 + 4 * lane)
                                  : (const void*)(reinterpret_cast<const JT*>(a.J_pl) + ((int64_t)l * a.PS + s0) * 6 + JL * (lane - 16));
      if (lane < 16 + 192 / JL) nsim_glds16(src, pf + 1024 * l);
    }
  };
  float hn[16 * NC];
  if constexpr (GLDS) {
    const int64_t t0 = (int64_t)blockIdx.x * JOINT_WAVES + wave;
    if (t0 < ntiles) prefetch_planes(t0);
  } else {
    load_h_at(hn, ((int64_t)blockIdx.x * JOINT_WAVES + wave) * 32 + j);
  }
  for (int64_t grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
    KT(1, 0);
    const int64_t s = (grp * JOINT_WAVES + wave) * 32 + j;      // past the end: an invalid point contributes zeros
    const bool valid = s < a.S;
    // the per-lane bias / head vectors are re-read from LDS where they are used: an address the compiler cannot prove
    // loop-invariant keeps it from hoisting 96 of them into registers for the whole launch
    const int vo = nsim_opaque_zero();
    const char* Wv = W + vo;
    float gs = 0.f, gn[3] = {0.f, 0.f, 0.f};
    auto load_upstream = [&]() {
      if (valid) {
        if (a.dsdf) gs = a.dsdf[s];
        if (a.dnablas) {
#pragma unroll
          for (int c = 0; c < 3; ++c) gn[c] = a.dnablas[3 * s + c];
        }
      }
    };
    // BJD: the upstream gradients are requested AFTER the wait for the plane image below -- issued in front of it, that
    // s_waitcnt vmcnt(0) exposed their full memory latency at the top of every group (s_memtime: 4.4 k of 27 k ticks per group)
    if constexpr (!(PREC == 0 && NC == 1 && NE == 0 && NSIM_BWD_JDIRECT)) load_upstream();
    // ---- dL/dg = J . gn (second-order path through the normals) and the features, from the level-major planes
    float h[16 * NI];                   // [features | embedded position]
    float Jr[(NC == 1 && !BJD) ? 16 : 1][3];      // dh/dx of this group (NC == 2 / BJD: consumed straight from its loads, below)
    JPair<JT> Jq[BJD ? 8 : 1][3];                 // BJD: the (level, point) entries of this lane as stored (three pairs each)
    float gh[16 * NI];                  // dL / dg = J . gn
    if constexpr (NE > 0) {             // the block's values and its tangent along dL/dnablas, from the sample position
      const TilePoint tp = load_point(a, grp * JOINT_WAVES + wave, j, false);
      float ed[32];
      embed_eval(a, tp.xx, valid, hi, reinterpret_cast<float(&)[32]>(h[16 * NC]), ed);
      embed_tangent(&gh[16 * NC], ed, gn, hi);
    }
    if constexpr (BJD) {
      nsim_wait_vm0();                          // this tile's feature image has landed
      const int64_t sc = valid ? s : a.S - 1;   // (a point past the end reads the last point's finite values; nothing of it is stored)
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
          const int l = 4 * q + 2 * hi + b, r0 = 4 * q + 2 * b;
          const float* hp = reinterpret_cast<const float*>(pf + 256 * l) + 2 * j;
          h[r0] = valid ? hp[0] : 0.f;
          h[r0 + 1] = valid ? hp[1] : 0.f;
        }
      nsim_wait_lgkm0();                        // every lane has read the image: the next copy may overwrite it
      load_upstream();
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
          const int l = 4 * q + 2 * hi + b;
          const JPair<JT>* jp = reinterpret_cast<const JPair<JT>*>(reinterpret_cast<const JT*>(a.J_pl) + ((int64_t)l * a.PS + sc) * 6);
          Jq[2 * q + b][0] = jp[0];
          Jq[2 * q + b][1] = jp[1];
          Jq[2 * q + b][2] = jp[2];
        }
      const int64_t tn = (grp + gridDim.x) * JOINT_WAVES + wave;
      if (tn < ntiles) prefetch_planes(tn);
    } else if constexpr (GLDS) {
      nsim_wait_vm0();                          // this tile's image has landed
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
          const int l = 4 * q + 2 * hi + b, r0 = 4 * q + 2 * b;
          const float* hp = reinterpret_cast<const float*>(pf + 1024 * l) + 2 * j;
          const JT* jp = reinterpret_cast<const JT*>(pf + 1024 * l + 256) + 6 * j;
          const bool lv = valid && LV_OK(l);
          h[r0] = lv ? hp[0] : 0.f;
          h[r0 + 1] = lv ? hp[1] : 0.f;
          float ja[3], jb[3];
          jload6(jp, ja, jb);
#pragma unroll
          for (int c3 = 0; c3 < 3; ++c3) {
            Jr[r0][c3] = lv ? ja[c3] : 0.f;
            Jr[r0 + 1][c3] = lv ? jb[c3] : 0.f;
          }
        }
      nsim_wait_lgkm0();                        // every lane has read the image: the next copy may overwrite it
      const int64_t tn = (grp + gridDim.x) * JOINT_WAVES + wave;
      if (tn < ntiles) prefetch_planes(tn);
    } else if constexpr (NC == 2) {
#pragma unroll
      for (int r = 0; r < 16 * NC; ++r) h[r] = hn[r];
#pragma unroll
      for (int m = 0; m < NC; ++m)
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
          for (int b = 0; b < 2; ++b) {
            const int l = 16 * m + 4 * q + 2 * hi + b, r0 = 16 * m + 4 * q + 2 * b;
            gh[r0] = gh[r0 + 1] = 0.f;
            if (valid && LV_OK(l)) {
              float ja[3], jb[3];
              jload6(reinterpret_cast<const JT*>(a.J_pl) + ((int64_t)l * a.PS + s) * 6, ja, jb);
              gh[r0] = ja[0] * gn[0] + ja[1] * gn[1] + ja[2] * gn[2];
              gh[r0 + 1] = jb[0] * gn[0] + jb[1] * gn[1] + jb[2] * gn[2];
            }
          }
      load_h_at(hn, ((grp + gridDim.x) * JOINT_WAVES + wave) * 32 + j);      // next group's features (zeros past the end)
    } else {
#pragma unroll
      for (int r = 0; r < 16; ++r) h[r] = hn[r];
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
          const int l = 4 * q + 2 * hi + b, r0 = 4 * q + 2 * b;
#pragma unroll
          for (int c3 = 0; c3 < 3; ++c3) Jr[r0][c3] = Jr[r0 + 1][c3] = 0.f;
          if (valid && LV_OK(l)) {
            jload6(reinterpret_cast<const JT*>(a.J_pl) + ((int64_t)l * a.PS + s) * 6, Jr[r0], Jr[r0 + 1]);
          }
        }
      load_h_at(hn, ((grp + gridDim.x) * JOINT_WAVES + wave) * 32 + j);      // next group's features (zeros past the end)
    }
    KT(1, 1);
    // ---- decoder forward (recomputed) and d sdf / d h
    float a1[32];
    dense<PREC, 2, NI>(a1, W + L.mat[M_W1], h, true);
#pragma unroll
    for (int k = 0; k < 32; ++k) a1[k] = softplus_b(a1[k] + vecf(Wv, L, V_B1, hi, k), beta, inv_beta);
    float a2[32];
    float d1[32];      // d sdf / d z1 = sigma(beta z1) . e1, e1 = d sdf / d a1 (not kept: e1 sigma1 == d1 wherever it is needed)
    if constexpr (SDF_D == 2) {
      dense<PREC, 2, 2>(a2, W + L.mat[M_W2], a1, false);
#pragma unroll
      for (int k = 0; k < 32; ++k) a2[k] = softplus_b(a2[k] + vecf(Wv, L, V_B2, hi, k), beta, inv_beta);
      float d2[32];
#pragma unroll
      for (int k = 0; k < 32; ++k) d2[k] = sig_from_softplus(a2[k], beta) * vecf(Wv, L, V_WH, hi, k);
      dense<PREC, 2, 2>(d1, W + L.mat[M_W2T], d2, false);
#pragma unroll
      for (int k = 0; k < 32; ++k) d1[k] = sig_from_softplus(a1[k], beta) * d1[k];
    } else {
#pragma unroll
      for (int k = 0; k < 32; ++k) {
        a2[k] = a1[k];
        d1[k] = sig_from_softplus(a1[k], beta) * vecf(Wv, L, V_WH, hi, k);
      }
    }
    {
      float g[16 * NC];
      dense<PREC, NC, 2>(g, W + L.mat[M_W1T], d1, false);
      if (valid && a.g_pl) {      // hand-off to the scatter kernel: g = d sdf / d h
#pragma unroll
        for (int m = 0; m < NC; ++m)
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
          for (int b = 0; b < 2; ++b) {
            const int l = 16 * m + 4 * q + 2 * hi + b, r0 = 16 * m + 4 * q + 2 * b;
            if (!LV_OK(l)) continue;
            float* gp = a.g_pl + ((int64_t)l * a.S + s) * 2;
            gp[0] = g[r0];
            gp[1] = g[r0 + 1];
          }
      }
    }
    // ======================================================================================= backward
    if constexpr (BJD) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {      // entry e = 2 q + b holds features r0 = 4 q + 2 b (d f0 / dx) and r0 + 1 (d f1 / dx)
        const int r0 = 4 * (e >> 1) + 2 * (e & 1);
        const float m = valid ? 1.f : 0.f;
        gh[r0] = m * ((float)Jq[e][0].x * gn[0] + (float)Jq[e][0].y * gn[1] + (float)Jq[e][1].x * gn[2]);
        gh[r0 + 1] = m * ((float)Jq[e][1].y * gn[0] + (float)Jq[e][2].x * gn[1] + (float)Jq[e][2].y * gn[2]);
      }
    } else if constexpr (NC == 1) {
#pragma unroll
      for (int f = 0; f < 16; ++f) gh[f] = Jr[f][0] * gn[0] + Jr[f][1] * gn[1] + Jr[f][2] * gn[2];
    }
    // ---- dW1 += d1 (x) gh
    KT(1, 2);
    use_set(0);
    if constexpr (!DB) __syncthreads();
    KT(1, 3);                              // the previous group's readers of the staging areas are done
    jstage<PREC, 2, SP>(stA, d1, wave);
    jstage<PREC, NI, SP>(stB, gh, wave);
    __syncthreads();
    if (do_dw) dw1_tiles(stA, stB);
    KT(1, 4);
    float dh1[32];  // dL / d d1 = W1 . gh
    dense<PREC, 2, NI>(dh1, W + L.mat[M_W1], gh, true);
    KT(1, 5);
    float dz1[32], whv[32];
    if constexpr (SDF_D == 2) {
      float eh1[32];  // dL / d e1
#pragma unroll
      for (int k = 0; k < 32; ++k) {
        const float s1 = sig_from_softplus(a1[k], beta);
        dz1[k] = dh1[k] * d1[k] * (beta * (1.0f - s1));      // dh1 e1 beta s1 (1 - s1), e1 s1 = d1
        eh1[k] = dh1[k] * s1;
      }
      float d2[32];
#pragma unroll
      for (int k = 0; k < 32; ++k) d2[k] = sig_from_softplus(a2[k], beta) * vecf(Wv, L, V_WH, hi, k);
      // ---- dW2 += d2 (x) eh1
      KT(1, 6);
      use_set(1);
      if constexpr (!DB) __syncthreads();
      KT(1, 7);
      jstage<PREC, 2, SP>(stA, d2, wave);
      jstage<PREC, 2, SP>(stB, eh1, wave);
      __syncthreads();
      if (do_dw) accW2 = jdw_tile<PREC>(stA, wave >> 1, stB, wave & 1, accW2);
      KT(1, 8);
      float dh2[32];  // dL / d d2 = W2 . eh1
      dense<PREC, 2, 2>(dh2, W + L.mat[M_W2], eh1, true);
      float dz2[32];
#pragma unroll
      for (int k = 0; k < 32; ++k) {
        const float s2 = sig_from_softplus(a2[k], beta);
        const float wh = vecf(Wv, L, V_WH, hi, k);
        whv[k] = dh2[k] * s2 + gs * a2[k];
        dz2[k] = gs * wh * s2 + dh2[k] * wh * (beta * s2 * (1.0f - s2));
      }
      // ---- dW2 += dz2 (x) a1, d b2 += rowsum(dz2), d wh += rowsum(whv)
      KT(1, 9);
      use_set(0);
      if constexpr (!DB) __syncthreads();
      KT(1, 10);
      jstage<PREC, 2, SP>(stA, dz2, wave);
      jstage<PREC, 2, SP>(stB, a1, wave);
      jstage<PREC, 2, SP>(stC, whv, wave);
      __syncthreads();
      if (do_dw) {
        accW2 = jdw_tile<PREC>(stA, wave >> 1, stB, wave & 1, accW2);
        bs2 += jrow_sum<PREC>(stA, 64, wave);
        bsh += jrow_sum<PREC>(stC, 64, wave);
      }
      KT(1, 11);
      float da1[32];
      dense<PREC, 2, 2>(da1, W + L.mat[M_W2T], dz2, true);
#pragma unroll
      for (int k = 0; k < 32; ++k) dz1[k] = dz1[k] + da1[k] * sig_from_softplus(a1[k], beta);
    } else {
#pragma unroll
      for (int k = 0; k < 32; ++k) {
        const float s1 = sig_from_softplus(a1[k], beta);
        const float wh = vecf(Wv, L, V_WH, hi, k);
        whv[k] = dh1[k] * s1 + gs * a1[k];
        dz1[k] = gs * wh * s1 + dh1[k] * wh * (beta * s1 * (1.0f - s1));
      }
    }
    {
      float v = (hi == 0) ? gs : 0.f;
      bh += wave_sum(v);
    }
    // ---- dW1 += dz1 (x) h, d b1 += rowsum(dz1)  (+ d wh when there is one hidden layer)
    KT(1, 12);
    use_set(1);
    if constexpr (!DB) __syncthreads();
    KT(1, 13);
    jstage<PREC, 2, SP>(stA, dz1, wave);
    jstage<PREC, NI, SP>(stB, h, wave);
    if constexpr (SDF_D == 1) jstage<PREC, 2, SP>(stC, whv, wave);
    __syncthreads();
    if (do_dw) {
      dw1_tiles(stA, stB);
      bs1 += jrow_sum<PREC>(stA, 64, wave);
      if constexpr (SDF_D == 1) bsh += jrow_sum<PREC>(stC, 64, wave);
    }
    KT(1, 14);
    float dh[16 * NC];
    dense<PREC, NC, 2>(dh, W + L.mat[M_W1T], dz1, true);
    KT(1, 15);
    if (a.dx) {   // pose refinement: dL/dx += (dh/dx)^T dL/dh  (dh/dx re-read from the planes: this path is rare)
      float acc[3] = {0.f, 0.f, 0.f};
      if (valid) {
#pragma unroll
        for (int m = 0; m < NC; ++m)
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
          for (int b = 0; b < 2; ++b) {
            const int l = 16 * m + 4 * q + 2 * hi + b, r0 = 16 * m + 4 * q + 2 * b;
            if (!LV_OK(l)) continue;
            float ja[3], jb[3];
            jload6(reinterpret_cast<const JT*>(a.J_pl) + ((int64_t)l * a.PS + s) * 6, ja, jb);
#pragma unroll
            for (int c3 = 0; c3 < 3; ++c3) acc[c3] = acc[c3] + dh[r0] * ja[c3] + dh[r0 + 1] * jb[c3];
          }
      }
#pragma unroll
      for (int c3 = 0; c3 < 3; ++c3) acc[c3] = acc[c3] + wave_shfl_xor(acc[c3], 32);
      if (valid && hi == 0) {
#pragma unroll
        for (int c3 = 0; c3 < 3; ++c3) a.dx[3 * s + c3] = a.dx[3 * s + c3] + acc[c3];
      }
    }
    if (valid && a.dh_pl) {      // hand-off to the scatter kernel: dL/dh
#pragma unroll
      for (int m = 0; m < NC; ++m)
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
          const int l = 16 * m + 4 * q + 2 * hi + b, r0 = 16 * m + 4 * q + 2 * b;
          if (!LV_OK(l)) continue;
          float* dp = a.dh_pl + ((int64_t)l * a.S + s) * 2;
          dp[0] = dh[r0];
          dp[1] = dh[r0 + 1];
        }
    }
    KT(1, 16);
    KT(1, 17);
  }
  { const int64_t grp = -1; KT(1, 21); }
  // ---- one flush per wave
  const int F1 = 2 * a.lotd.num_levels, FIN = F1 + (NE ? a.embed_E : 0);      // W1 rows: [features (F1) | embedded position]
  const SrcOff so = src_off(SDF_D, FIN);
  const int64_t ro = (int64_t)(blockIdx.x & (unsigned)a.rep_mask) * a.rep_stride;      // this workgroup's replica
  // column tile ``no`` of the accumulators -> columns of W1: feature tiles [0, NC) hold columns 32 no + i < F1, the block's
  // tiles columns F1 + 32 (no - NC) + i < FIN
  auto flush_w1 = [&](int mo, int no, const f32x16& acc) {
    if (no < NC) jflush_tile(a.dsdf_w + ro + so.w1, FIN, 64, F1, mo, no, acc);
    else jflush_tile(a.dsdf_w + ro + so.w1 + (F1 - 32 * NC), FIN, 64, 32 * NC + a.embed_E, mo, no, acc);
  };
#pragma unroll
  for (int q = 0; q < NPAIR; ++q) flush_w1(wave >> 1, 2 * q + (wave & 1), accW1[q]);
  if constexpr (NODD) flush_w1(wave & 1, NI - 1, accW1[NPAIR]);
  if constexpr (SDF_D == 2) {
    jflush_tile(a.dsdf_w + ro + so.w2, 64, 64, 64, wave >> 1, wave & 1, accW2);
    if (bs2 != 0.f) atomicAdd(&a.dsdf_b[ro + so.b2 + lane], bs2);
  }
  if (bs1 != 0.f) atomicAdd(&a.dsdf_b[ro + so.b1 + lane], bs1);
  if (bsh != 0.f) atomicAdd(&a.dsdf_w[ro + so.wh + lane], bsh);
  if (lane == 0 && bh != 0.f) atomicAdd(&a.dsdf_b[ro + so.bh], bh);
  { const int64_t grp = -1; KT(1, 22); }
}

// No-grad SDF query (all sampling / occupancy-refresh traffic goes through here).  Lean variant of the forward:
// the gather is split in two phases of 4 levels per lane whose 8 features feed one MFMA K-step each, so only 8
// features (and 32 corner loads) are live at a time -> ~half the registers of the fused forward -> more waves per
// SIMD to hide the gather latency.  Features are scaled by a fixed exact power of two before the f16 conversion.
#define SDF_H_SCALE 1024.0f
#ifndef NSIM_SDF_MIN_WAVES
#define NSIM_SDF_MIN_WAVES 2
#endif
#ifndef NSIM_SDF_NBUF
#define NSIM_SDF_NBUF 1      // LDS plane images per wave of the sampling decoder (<= 16 levels); 2 = double-buffered (measured: 0.0387 vs 0.0372 ms)
#endif
// Level-major gather of the no-grad SDF query (sampling pass, occupancy refresh): every wave owns GLM_PTS x 64 points
// and walks the 16 levels in the same order as every other wave of the launch, so at any moment the chip reads ONE
// level's table (<= 2 MB for T = 2^19), which stays resident in each XCD's 4 MB L2 -- the point-major fused kernel
// spreads its accesses over all 24 MB at once and fetched 3.6x the algorithmic bytes from beyond L2.  GLM_PTS x 8
// independent 4-byte loads are in flight per lane.  Output: planes [16][S] of (f16x2 * SDF_H_SCALE | f32x2).
// The levels are dealt to the 8 XCDs (block b runs on XCD b % 8 -- an observed placement used for speed only, the
// result does not depend on it): an XCD pulls only ITS levels' tables (~3 MB) through its L2 instead of all 24 MB.
// Measured ceiling (tools/gather_bench.hip): a random 4-byte gather retires 267 G lines/s from an L2-resident 2 MB
// table (= the 34 TB/s aggregate L2->L1 rate at 128 B per miss) but only 65-120 G/s from an 8-32 MB one.
// WJ = true (with-grad forward of the training step): writes the f32 planes h [16][S][2] and dh/dx [16][S][2][3] that
// the decoder kernels (k_field MODE 3 forward, MODE 2 backward) read -- the same split, so the 6.7 k-instruction
// forward no longer carries the gather's address registers (it spilled 440 B per lane).
#ifndef GLM_PTS
#define GLM_PTS 2      // points per lane of the no-grad forms (round 6: 4 -> 2, nsim_lotd_gather_lm 0.0608 -> 0.0593 ms per launch over two
#endif                 // alternated runs each, gpurun_out/r6_s2_call15; 1 point and / or the operand tables: the same within noise)
#ifndef GLM_PTS_WJ
#define GLM_PTS_WJ 1      // points per lane of the with-grad form (eight accumulators per point: f, dh/dx): one -- 57 registers, eight
                          // waves per SIMD -- with the slot tables below: nsim_field_fwd 0.1200 -> 0.1131 ms on the bench step (4 points:
                          // 96 registers, five waves; 2 points 0.1161, 2 + tables 0.1154, 3 + tables 0.1192; gpurun_out/r6_s2_call13)
#endif
#ifndef NSIM_GATHER_WJ_WAVES
#define NSIM_GATHER_WJ_WAVES 1   // WJ (with-grad planes): waves per SIMD the register allocation must leave room for (A/B knob of the slot tables)
#endif
#ifndef NSIM_GATHER_SLOTS_WJ
#define NSIM_GATHER_SLOTS_WJ 1   // ... for the with-grad form only: that form is VALU-heavy enough (dh/dx) for the tables to pay once its
                                 // register count no longer costs waves
#endif
#ifndef NSIM_GATHER_SLOTS
#define NSIM_GATHER_SLOTS 0      // 1: the parity enumeration through per-axis operand tables (lotd_slots, ~30 % fewer VALU operations per level) instead
                                 // of runtime corner indices.  Measured null on MI355X (gpurun_out/r6_s2_call7: nsim_lotd_gather_lm 0.0603 -> 0.0599 ms,
                                 // the with-grad gather inside nsim_field_fwd +1.5 % at 99 registers / four waves or 96 + 20 B scratch): the
                                 // gathers wait on L2 requests, not on the VALU
#endif
template <int PREC, bool WJ>
__global__ void __launch_bounds__(64, WJ ? NSIM_GATHER_WJ_WAVES : 1) k_lotd_gather_lm(FieldArgs a) {
  using JT = typename JPlane<PREC>::T;      // element type of the dh/dx planes (WJ)
  // points per lane.  (Round 5: 1 / 2 points per lane for launches of <= 98 k / 196 k points -- four times the waves for the
  // small up-sampling draws -- measured nothing: 0.0643-0.0658 against 0.0651-0.0666 ms per launch, profiles/round5_gather_ab.txt)
  constexpr int NP = WJ ? GLM_PTS_WJ : GLM_PTS;
  const int lane = nsim_lane();
  const int xcd = (int)(blockIdx.x & 7u);
  const int64_t s0 = (int64_t)(blockIdx.x >> 3) * (64 * NP) + lane;
  int64_t Sv = a.S;                                 // valid points (<= capacity a.S)
  if (a.S_dev) {
    // more points than the buffers were sized for: the packs are no longer contiguous (nsim_pack_infos_from_n
    // emptied some), the caller will redo the pass -- touch nothing
    const int64_t sd = a.S_dev[0] + a.S_add;
    Sv = sd <= Sv ? sd : 0;
  }
  // the grid covers the CAPACITY a.S; a workgroup whose points all lie beyond the valid count has nothing to read or write
  // (round 5: with device-side counts -- speculative sizes, upsample_on_marched_only -- that is up to half of the grid)
  if (s0 - lane >= Sv) return;
  float xx[NP][3];
  uint32_t goff[NP];
#pragma unroll
  for (int q = 0; q < NP; ++q) {
    const int64_t s = s0 + 64 * q;
    xx[q][0] = xx[q][1] = xx[q][2] = 0.f;
    goff[q] = 0u;
    if (s < Sv) {
      if (a.ray_goff) goff[q] = (uint32_t)a.ray_goff[a.ridx[s]];
      if (a.x) {
        xx[q][0] = a.x[3 * s]; xx[q][1] = a.x[3 * s + 1]; xx[q][2] = a.x[3 * s + 2];
      } else {
        const int64_t ray = a.ridx[s];
        const float tt = a.t[s];
#pragma unroll
        for (int c = 0; c < 3; ++c) xx[q][c] = a.rays_o[3 * ray + c] + tt * a.rays_d[3 * ray + c];
      }
    }
  }
  const GridRef gref = grid_ref(a.grid);
  const int nl = a.glm_n[xcd];
  // (a level dealt to two XCDs is split between the two halves of the VALID point range)
  const int64_t nblk_valid = (Sv + 64 * NP - 1) / (64 * NP);
  const bool second_half = (int64_t)(blockIdx.x >> 3) >= (nblk_valid + 1) / 2;
#pragma unroll 1
  for (int k = 0; k < nl; ++k) {
    const int l = a.glm_lv[xcd][k];
    const int half = a.glm_half[xcd][k];
    if ((half == 1 && second_half) || (half == 2 && !second_half)) continue;
    const LotdRes R = a.lotd.res[l];
    const int type = a.lotd.type[l];
    const uint32_t T = a.lotd.size[l], off = (uint32_t)a.lotd.offset[l];
    float f0[NP], f1[NP];
    float j0[WJ ? NP : 1][3], j1[WJ ? NP : 1][3];
#pragma unroll
    for (int q = 0; q < NP; ++q) {
      const LotdCell c = lotd_cell(xx[q], R, a.lotd);
      f0[q] = f1[q] = 0.f;
      if constexpr (WJ) {
#pragma unroll
        for (int c3 = 0; c3 < 3; ++c3) j0[q][c3] = j1[q][c3] = 0.f;
      }
      // slots by vertex parity (lotd_dev.h): through per-axis operand tables (SLOTS) or runtime corner indices
      constexpr bool SLOTS = NSIM_GATHER_SLOTS || (WJ && NSIM_GATHER_SLOTS_WJ);
      const LotdSlots SL = lotd_slots(c);
      const int pm = lotd_slot_mask(c);
      if (l < a.lotd.n_active)
#pragma unroll
      for (int slot = 0; slot < 8; ++slot) {
        float w, dw[3];
        uint32_t idx;
        if constexpr (SLOTS) {
          lotd_slot_w(SL, slot, w, dw);
          idx = lotd_slot_index(SL, slot, R, type, T);
        } else {
          const int corner = slot ^ pm;
          lotd_corner_w(c, corner, w, dw);
          idx = lotd_index(c.c0[0] + (corner & 1), c.c0[1] + ((corner >> 1) & 1), c.c0[2] + ((corner >> 2) & 1), R, type, T);
        }
        float g0, g1;
        lotd_load2(gref, off + goff[q] + 2u * idx, g0, g1);
        f0[q] = f0[q] + w * g0;
        f1[q] = f1[q] + w * g1;
        if constexpr (WJ) {
#pragma unroll
          for (int c3 = 0; c3 < 3; ++c3) {
            j0[q][c3] = j0[q][c3] + dw[c3] * g0;
            j1[q][c3] = j1[q][c3] + dw[c3] * g1;
          }
        }
      }
      if constexpr (WJ) {
#pragma unroll
        for (int c3 = 0; c3 < 3; ++c3) {
          j0[q][c3] = j0[q][c3] * c.dscale[c3];
          j1[q][c3] = j1[q][c3] * c.dscale[c3];
        }
      }
    }
#pragma unroll
    for (int q = 0; q < NP; ++q) {
      const int64_t s = s0 + 64 * q;
      if (s < Sv) {
        const int64_t e = (int64_t)l * a.PS + s;      // feature planes [NL][P], P = NSIM_PLANE_PITCH(S)
        if constexpr (WJ) {
          const int64_t ep = (int64_t)l * a.PS + s;
          float* hp = a.h_pl + ep * 2;
          JT* jp = reinterpret_cast<JT*>(a.J_pl) + ep * 6;
          hp[0] = f0[q];
          hp[1] = f1[q];
#ifdef NSIM_PROBE_GATHER_NOJ      // timing probe (wrong results): the dh/dx stores dropped but for an impossible case
          if (f0[q] == 123.456f)
#endif
          jstore6(jp, j0[q], j1[q]);
        } else if constexpr (PREC == 0) {
          union {
            uint32_t u;
            f16 h[2];
          } cv;
          cv.h[0] = (f16)f16_sat(f0[q] * SDF_H_SCALE);
          cv.h[1] = (f16)f16_sat(f1[q] * SDF_H_SCALE);
          reinterpret_cast<uint32_t*>(a.feat_pl)[e] = cv.u;
        } else {
          reinterpret_cast<float*>(a.feat_pl)[2 * e] = f0[q];
          reinterpret_cast<float*>(a.feat_pl)[2 * e + 1] = f1[q];
        }
      }
    }
  }
}

template <int PREC, bool WJ>
static void launch_gather_lm(const FieldArgs& a, int64_t S, hipStream_t stream) {
  const dim3 gg((unsigned)(8 * nsim_blocks(S, 64 * (WJ ? GLM_PTS_WJ : GLM_PTS))));
  hipLaunchKernelGGL((k_lotd_gather_lm<PREC, WJ>), gg, dim3(64), 0, stream, a);
}

// GL (PLANES only): the tile's plane image -- per level 32 points x (f16x2 | f32x2) = 128 B | 256 B, one aligned piece
// thanks to the 32-point pitch -- is copied global -> LDS (global_load_lds_dwordx4: 8 | 4 levels per instruction) one tile
// AHEAD, double-buffered for <= 16 levels (2 x 4 KB per wave), single-buffered above (8 KB): without it a tile began with
// eight dependent-on-nothing but unhidden plane reads per K-step at two waves per SIMD (round 4; the with-grad decoders got
// the same treatment in round 2).  NSIM_SDF_GLDS=0 launches the GL = false form.
template <int PREC, int SDF_D, bool PLANES, int NC = 1, bool GL = false>
__global__ void __launch_bounds__(64 * FIELD_WAVES, NSIM_SDF_MIN_WAVES) k_field_sdf(FieldArgs a) {
  static_assert(!GL || PLANES, "the LDS image is an image of the level-major planes");
  NSIM_DYN_SMEM(smem);
  const int lane = nsim_lane(), j = lane & 31, hi = lane >> 5;
  const int wave = (int)(threadIdx.x >> 6);
  const float beta = a.beta, inv_beta = 1.0f / a.beta;
  FieldLayout L;
  int wbytes;
  const char* W = stage_weights<PREC>(smem, a, 0, 2, L, wbytes);   // W1, W2 only
  const float b_out = reinterpret_cast<const float*>(W + L.vec[V_SCAL])[0];
  const GridRef gref = grid_ref(a.grid);
  int64_t Sv = a.S;
  if (a.S_dev) {
    const int64_t sd = a.S_dev[0] + a.S_add;
    Sv = sd <= Sv ? sd : 0;       // overflow of a speculative capacity: see k_lotd_gather_lm
  }
  const int64_t ntiles = (Sv + 31) / 32;
  const int64_t wstride = (int64_t)gridDim.x * FIELD_WAVES;
  // ---- LDS image of the planes (GL)
  constexpr int LV_BYTES = PREC == 0 ? 128 : 256;               // one level of one tile
  constexpr int LV_PER_COPY = 1024 / LV_BYTES;                  // levels per global_load_lds_dwordx4 (64 lanes x 16 B)
  constexpr int IMG_BYTES = 16 * NC * LV_BYTES;
  constexpr int NBUF = NC == 1 ? NSIM_SDF_NBUF : 1;
  const int lv_used = NC == 1 ? 16 : ((a.lotd.num_levels + 7) & ~7);       // (wave-uniform) levels the K-steps read
  char* img = GL ? smem + wbytes + wave * (NBUF * IMG_BYTES) : nullptr;
  auto prefetch_planes = [&](int64_t tile_n, int buf) {
    const int64_t s0 = tile_n * 32;
    const int sub = lane / (64 / LV_PER_COPY), part = lane % (64 / LV_PER_COPY);
#pragma unroll
    for (int k = 0; k < 16 * NC / LV_PER_COPY; ++k) {
      if (NC == 2 && k * LV_PER_COPY >= lv_used) continue;
      const int l = k * LV_PER_COPY + sub;
      const char* src = reinterpret_cast<const char*>(a.feat_pl) + ((int64_t)l * a.PS + s0) * (LV_BYTES / 32) + 16 * part;
      nsim_glds16(src, img + buf * IMG_BYTES + 1024 * k);
    }
  };
  int buf = 0;
  if constexpr (GL) {
    const int64_t t0 = (int64_t)blockIdx.x * FIELD_WAVES + wave;
    if (t0 < ntiles) prefetch_planes(t0, 0);
  }
  for (int64_t tile = (int64_t)blockIdx.x * FIELD_WAVES + wave; tile < ntiles; tile += wstride) {
    TilePoint p;
    if constexpr (PLANES) {
      p.s = tile * 32 + j;
      p.valid = p.s < Sv;
    } else {
      p = load_point(a, tile, j, false);
      p.valid = p.valid && p.s < Sv;            // (a device-side point count below the capacity: round 5)
    }
    const char* cur = nullptr;
    if constexpr (GL) {
      nsim_wait_vm0();                          // this tile's image has landed
      cur = img + buf * IMG_BYTES;
      if constexpr (NBUF == 2) {                // the next tile's image goes to the other buffer while this one is read
        if (tile + wstride < ntiles) prefetch_planes(tile + wstride, buf ^ 1);
        buf ^= 1;
      }
    }
    f32x16 acc[2] = {zero16(), zero16()};
    f32x16 accc[PREC == 2 ? 2 : 1];            // split mode: the hi.lo + lo.hi correction, in units of 1 / SPLIT_LO_SCALE
    if constexpr (PREC == 2) accc[0] = accc[1] = zero16();
#pragma unroll 1
    for (int rb = 0; rb < 2 * NC; ++rb) {      // one K-step of 8 features per lane-half: levels 16 (rb/2) + 8 (rb%2) + ..
      const int lb = 16 * (rb >> 1) + 8 * (rb & 1);
      // 17..32 levels: K-steps entirely past the pyramid are skipped; inside the last partial one the planes hold zeros
      // (written by the gather) -- per-lane guards on these loads cost 40 % of the kernel (0.094 -> 0.134 ms per 1.39 M points)
      if (NC == 2 && lb >= ((a.lotd.num_levels + 7) & ~7)) continue;
      float f8[8];
      f16x8 bvp;
      if constexpr (GL) {
        // from the LDS image; lanes past the valid points read whatever the pitch padding holds: zeroed (their columns of
        // the products are discarded, but nothing non-finite is fed to the matrix cores)
#pragma unroll
        for (int qq = 0; qq < 2; ++qq)
#pragma unroll
          for (int b = 0; b < 2; ++b) {
            const int l = lb + 4 * qq + 2 * hi + b;
            if constexpr (PREC == 0) {
              union {
                uint32_t u;
                f16 h[2];
              } cv;
              cv.u = reinterpret_cast<const uint32_t*>(cur + LV_BYTES * l)[j];
              if (!p.valid) cv.u = 0u;
              bvp[4 * qq + 2 * b] = cv.h[0];
              bvp[4 * qq + 2 * b + 1] = cv.h[1];
            } else {
              const float* ip = reinterpret_cast<const float*>(cur + LV_BYTES * l) + 2 * j;
              f8[4 * qq + 2 * b] = p.valid ? ip[0] : 0.f;
              f8[4 * qq + 2 * b + 1] = p.valid ? ip[1] : 0.f;
            }
          }
      } else if constexpr (PLANES) {
        // features were gathered level-major by k_lotd_gather_lm (fp16 mode: already scaled by SDF_H_SCALE)
#pragma unroll
        for (int qq = 0; qq < 2; ++qq)
#pragma unroll
          for (int b = 0; b < 2; ++b) {
            const int l = lb + 4 * qq + 2 * hi + b;
            const int64_t e = (int64_t)l * a.PS + (p.valid ? p.s : 0);
            if constexpr (PREC == 0) {
              union {
                uint32_t u;
                f16 h[2];
              } cv;
              cv.u = reinterpret_cast<const uint32_t*>(a.feat_pl)[e];
              bvp[4 * qq + 2 * b] = cv.h[0];
              bvp[4 * qq + 2 * b + 1] = cv.h[1];
            } else {
              f8[4 * qq + 2 * b] = reinterpret_cast<const float*>(a.feat_pl)[2 * e];
              f8[4 * qq + 2 * b + 1] = reinterpret_cast<const float*>(a.feat_pl)[2 * e + 1];
            }
          }
      } else {
#pragma unroll
      for (int qq = 0; qq < 2; ++qq) {
#pragma unroll
        for (int b = 0; b < 2; ++b) {
          const int l = lb + 4 * qq + 2 * hi + b;
          const LotdRes R = a.lotd.res[l];
          const LotdCell c = lotd_cell(p.xx, R, a.lotd);
          float f0 = 0.f, f1 = 0.f;
          const int pm = lotd_slot_mask(c);
          if (l < a.lotd.n_active)
#pragma unroll
          for (int slot = 0; slot < 8; ++slot) {
            const int corner = slot ^ pm;
            float w, dw[3];
            lotd_corner_w(c, corner, w, dw);
            const uint32_t idx = lotd_index(c.c0[0] + (corner & 1), c.c0[1] + ((corner >> 1) & 1),
                                            c.c0[2] + ((corner >> 2) & 1), R, a.lotd.type[l], a.lotd.size[l]);
            float g0, g1;
            lotd_load2(gref, (uint32_t)a.lotd.offset[l] + p.goff + 2u * idx, g0, g1);
            f0 = f0 + w * g0;
            f1 = f1 + w * g1;
          }
          f8[4 * qq + 2 * b] = f0;
          f8[4 * qq + 2 * b + 1] = f1;
        }
      }
      }
      if constexpr (PREC == 0) {
        f16x8 bv;
        if constexpr (PLANES) {
          bv = bvp;
        } else {
#pragma unroll
          for (int e = 0; e < 8; ++e) bv[e] = (f16)f16_sat(f8[e] * SDF_H_SCALE);
        }
        const f16x8* A = reinterpret_cast<const f16x8*>(W + L.mat[M_W1]);
#pragma unroll
        for (int mo = 0; mo < 2; ++mo) acc[mo] = mfma_32x32x16_f16(A[(mo * 2 * NC + rb) * 64 + lane], bv, acc[mo]);
      } else if constexpr (PREC == 2) {
        f16x8 bh, bl;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          f16 h_, l_;
          split_f16(f8[e] * SDF_H_SCALE, h_, l_);
          bh[e] = h_;
          bl[e] = l_;
        }
        const f16x8* Ah = reinterpret_cast<const f16x8*>(W + L.mat[M_W1]);
        const f16x8* Al = Ah + 64 * 4 * NC;           // lo fragments follow the [64 x 32 NC] hi ones
#pragma unroll
        for (int mo = 0; mo < 2; ++mo) {
          const int fi = (mo * 2 * NC + rb) * 64 + lane;
          acc[mo] = mfma_32x32x16_f16(Ah[fi], bh, acc[mo]);
          accc[mo] = mfma_32x32x16_f16(Ah[fi], bl, accc[mo]);
          accc[mo] = mfma_32x32x16_f16(Al[fi], bh, accc[mo]);
        }
      } else {
        const float* A = reinterpret_cast<const float*>(W + L.mat[M_W1]);
#pragma unroll
        for (int e = 0; e < 8; ++e)
#pragma unroll
          for (int mo = 0; mo < 2; ++mo)
            acc[mo] = mfma_32x32x2_f32(A[(mo * 16 * NC + 8 * rb + e) * 64 + lane], f8[e], acc[mo]);
      }
    }
    if constexpr (GL && NBUF == 1) {            // one buffer: every lane has read the image, the next copy may overwrite it
      nsim_wait_lgkm0();
      if (tile + wstride < ntiles) prefetch_planes(tile + wstride, 0);
    }
    const float inv_h = PREC != 1 ? 1.0f / SDF_H_SCALE : 1.0f;
    float sdf = 0.f;
    if constexpr (PREC == 2) {
#pragma unroll
      for (int mo = 0; mo < 2; ++mo)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mo][r] = acc[mo][r] + accc[mo][r] * (1.0f / SPLIT_LO_SCALE);
    }
    if constexpr (PREC == 2 && SDF_D == 2) {
      // second layer on split operands: activations and weights as hi + lo, three MFMAs per K-step
      f16x8 bqh[4], bql[4];
#pragma unroll
      for (int mo = 0; mo < 2; ++mo)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          f16 h_, l_;
          split_f16(softplus_exact(acc[mo][r] * inv_h + vecf(W, L, V_B1, hi, mo * 16 + r), beta, inv_beta), h_, l_);
          bqh[2 * mo + (r >> 3)][r & 7] = h_;
          bql[2 * mo + (r >> 3)][r & 7] = l_;
        }
      const f16x8* A2h = reinterpret_cast<const f16x8*>(W + L.mat[M_W2]);
      const f16x8* A2l = A2h + 64 * 8;                 // [64 x 64] hi fragments: 8 per lane
#pragma unroll
      for (int mo = 0; mo < 2; ++mo) {
        f32x16 acc2 = zero16(), acc2c = zero16();
#pragma unroll
        for (int st = 0; st < 4; ++st) {
          const int fi = (mo * 4 + st) * 64 + lane;
          acc2 = mfma_32x32x16_f16(A2h[fi], bqh[st], acc2);
          acc2c = mfma_32x32x16_f16(A2h[fi], bql[st], acc2c);
          acc2c = mfma_32x32x16_f16(A2l[fi], bqh[st], acc2c);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r)
          sdf = sdf + vecf(W, L, V_WH, hi, mo * 16 + r) *
                          softplus_exact(acc2[r] + acc2c[r] * (1.0f / SPLIT_LO_SCALE) + vecf(W, L, V_B2, hi, mo * 16 + r), beta,
                                     inv_beta);
      }
    } else if constexpr (PREC == 2) {
#pragma unroll
      for (int mo = 0; mo < 2; ++mo)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          sdf = sdf + vecf(W, L, V_WH, hi, mo * 16 + r) *
                          softplus_exact(acc[mo][r] * inv_h + vecf(W, L, V_B1, hi, mo * 16 + r), beta, inv_beta);
    } else if constexpr (PREC == 0 && SDF_D == 2) {
      // register-lean second layer: layer-1 activations are packed to f16 B fragments at once (16 VGPRs), each
      // output M-tile is consumed by the head dot-product as soon as its four MFMAs retire
      f16x8 bq[4];
#pragma unroll
      for (int mo = 0; mo < 2; ++mo)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          bq[2 * mo + (r >> 3)][r & 7] =
              (f16)softplus_exact(acc[mo][r] * inv_h + vecf(W, L, V_B1, hi, mo * 16 + r), beta, inv_beta);
      const f16x8* A2 = reinterpret_cast<const f16x8*>(W + L.mat[M_W2]);
#pragma unroll
      for (int mo = 0; mo < 2; ++mo) {
        f32x16 acc2 = zero16();
#pragma unroll
        for (int st = 0; st < 4; ++st) acc2 = mfma_32x32x16_f16(A2[(mo * 4 + st) * 64 + lane], bq[st], acc2);
#pragma unroll
        for (int r = 0; r < 16; ++r)
          sdf = sdf + vecf(W, L, V_WH, hi, mo * 16 + r) *
                          softplus_exact(acc2[r] + vecf(W, L, V_B2, hi, mo * 16 + r), beta, inv_beta);
      }
    } else {
      float a1[32];
#pragma unroll
      for (int mo = 0; mo < 2; ++mo)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          a1[mo * 16 + r] = softplus_exact(acc[mo][r] * inv_h + vecf(W, L, V_B1, hi, mo * 16 + r), beta, inv_beta);
      if constexpr (SDF_D == 2) {
        float a2[32];
        dense<PREC, 2, 2>(a2, W + L.mat[M_W2], a1, false);
#pragma unroll
        for (int k = 0; k < 32; ++k)
          sdf = sdf + vecf(W, L, V_WH, hi, k) * softplus_exact(a2[k] + vecf(W, L, V_B2, hi, k), beta, inv_beta);
      } else {
#pragma unroll
        for (int k = 0; k < 32; ++k) sdf = sdf + vecf(W, L, V_WH, hi, k) * a1[k];
      }
    }
    sdf = sdf + wave_shfl_xor(sdf, 32);
    sdf = sdf + b_out;
    if (p.valid && hi == 0) a.sdf[p.s] = sdf;
    if (a.occ_val) {      // the 32 points of the tile are the hi == 0 lanes (neighbouring samples of a ray)
      const bool ok = p.valid && hi == 0;
      occ_collect_wave(a.occ_val, a.occ, ok, ok ? a.x[3 * p.s] : 0.f, ok ? a.x[3 * p.s + 1] : 0.f, ok ? a.x[3 * p.s + 2] : 0.f,
                       sdf, a.occ_inv_s);
    }
  }
}

// Backward of the radiance branch: given dL/drgb and the saved forward nablas / rgb -> gradients of the radiance weights,
// of the appearance codes, and dnab_total[s] = dL/dnablas[s] (upstream) + d(radiance path)/d nablas[s].  No grid access:
// per point 12 B (x) + 12 B (nablas) + 12 B (rgb) + 12 B (drgb) in, 12 B out.
// Workgroup-joint weight gradients (mfma_mlp.h: "workgroup-joint weight-gradient products"): the four waves of a workgroup stage their 32 points side by side (K = 128 per product), every wave owns
// two of the eight output tiles -- dR2 tile (wave >> 1, wave & 1) and either a dR3 tile (waves 0, 1) or a dR1 tile
// (waves 2, 3) -- in MFMA accumulator registers for the whole launch, the bias sums live in one register of the wave
// that idles during the product.  LDS: the six weight matrices (34.5 KB) + one staging area (34.8 KB) = 69 KB -> two
// workgroups per CU, two waves per SIMD (round 1's per-wave LDS accumulators: 141 KB, one).
// Measured (MI355X, 274 k points): 0.208 ms in round 1 -> 0.162 ms with this structure -> 0.104 ms once the per-sample
// appearance-code atomics were summed per ray inside the wave (they had been 80 % of a group's time: s_memtime stamps,
// tools/ktime.py); the round-1 structure with the same atomics fix measured 0.100 ms -- the atomics were the bound.
template <int PREC>
__global__ void __launch_bounds__(64 * JOINT_WAVES, PREC == 0 ? 2 : 1) k_rad_bwd_j(FieldArgs a) {
  NSIM_DYN_SMEM(smem);
  const int lane = nsim_lane(), j = lane & 31, hi = lane >> 5;
  const int wave = (int)(threadIdx.x >> 6);
  FieldLayout L;
  int wbytes;
  { const int64_t grp = -1; KT(0, 23); }
  const char* W = stage_weights<PREC>(smem, a, M_R1, 6, L, wbytes);
  char* stA = smem + wbytes;
  char* stB = stA + 64 * jstage_row_bytes<PREC>();
  // fp16 mode (round 5): the network input of the group is staged ONCE, right after the first barrier of the iteration, into a
  // third area of 32 rows (8.7 KB: 78 KB in all, still two workgroups per CU) -- it is the B operand of the LAST weight-gradient
  // product (dR1 += dr1 (x) rin), and holding its 16 registers across the whole backward chain was what pushed the kernel
  // 13 dwords past its 256 registers (52 B of scratch since round 3).  f32 validation mode: one workgroup per CU, staged late.
  constexpr bool RIN_EARLY = PREC == 0;
  char* stC = stB + 64 * jstage_row_bytes<PREC>();
  f32x16 accR2 = zero16(), accX = zero16();
  float bs1 = 0.f, bs2 = 0.f, bs3 = 0.f;      // this wave's share of d rb1[lane], d rb2[lane], d rb3[lane < 3]

  const int64_t ntiles = (a.S + 31) / 32;
  const int64_t ngroups = (ntiles + JOINT_WAVES - 1) / JOINT_WAVES;
  { const int64_t grp = -1; KT(0, 20); }
  for (int64_t grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
    KT(0, 0);
    const int64_t tile = grp * JOINT_WAVES + wave;          // may lie past the end: all its points are invalid (zeros)
    const TilePoint p = load_point(a, tile, j, true);
    const int64_t s = p.s;
    float nab[3] = {0.f, 0.f, 0.f}, rgbv[3] = {0.f, 0.f, 0.f}, gr[3] = {0.f, 0.f, 0.f}, gn[3] = {0.f, 0.f, 0.f};
    if (p.valid) {
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        nab[c] = a.nablas_fwd[3 * s + c];
        rgbv[c] = a.rgb_fwd[3 * s + c];
        gr[c] = a.drgb[3 * s + c];
        if (a.dnablas) gn[c] = a.dnablas[3 * s + c];
      }
    }
    float rin[16], r1[32], r2[32];
    make_rin(rin, p, nab, a.h_appear, hi);
    KT(0, 1);
    radiance_hidden<PREC>(r1, r2, rin, W, L, hi);
    KT(0, 2);
    float dout[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) dout[r] = 0.f;
    if (hi == 0) {
#pragma unroll
      for (int c = 0; c < 3; ++c) dout[c] = gr[c] * rgbv[c] * (1.0f - rgbv[c]);
    }
    // ONE exact power-of-two scale per tile for the whole backward chain (fp16 operands): the chain only multiplies by
    // weight matrices (|W| <~ 1, width 64) and relu masks, so every later value stays within ~2^-13 .. 2^6 of the
    // scaled input -- well inside fp16's range; it is carried scaled (``*s``) and unscaled where it leaves the chain
    const float sc = dyn_scale<PREC, 16>(dout), inv_sc = 1.0f / sc;
    float douts[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) douts[r] = dout[r] * sc;
    // ---- dR3 += dout (x) r2, d rb3 += rowsum(dout)
    KT(0, 3);
    __syncthreads();                              // the previous group's readers of the staging area are done
    KT(0, 4);
    jstage<PREC, 1>(stA, dout, wave);
    jstage<PREC, 2>(stB, r2, wave);
    if constexpr (RIN_EARLY) jstage<PREC, 1>(stC, rin, wave);      // (its readers of the previous group passed the barrier above)
    KT(0, 5);
    __syncthreads();
    KT(0, 6);
    if (wave < 2) accX = jdw_tile<PREC>(stA, 0, stB, wave, accX);
    bs3 += jrow_sum<PREC>(stA, 3, wave);
    KT(0, 7);
    float dr2s[32];
    dense<PREC, 2, 1>(dr2s, W + L.mat[M_R3T], douts, false);
#pragma unroll
    for (int k = 0; k < 32; ++k) dr2s[k] = r2[k] > 0.f ? dr2s[k] : 0.f;
    // ---- dR2 += dr2 (x) r1, d rb2 += rowsum(dr2)      (dr2 = dr2s / sc, formed where it is staged)
    KT(0, 8);
    __syncthreads();
    KT(0, 9);
    jstage_scaled<PREC, 2>(stA, dr2s, inv_sc, wave);
    jstage<PREC, 2>(stB, r1, wave);
    KT(0, 10);
    __syncthreads();
    KT(0, 11);
    accR2 = jdw_tile<PREC>(stA, wave >> 1, stB, wave & 1, accR2);
    bs2 += jrow_sum<PREC>(stA, 64, wave);
    KT(0, 12);
    float dr1s[32];
    dense<PREC, 2, 2>(dr1s, W + L.mat[M_R2T], dr2s, false);
#pragma unroll
    for (int k = 0; k < 32; ++k) dr1s[k] = r1[k] > 0.f ? dr1s[k] : 0.f;
    // ---- dR1 += dr1 (x) rin, d rb1 += rowsum(dr1)
    KT(0, 13);
    __syncthreads();
    KT(0, 14);
    jstage_scaled<PREC, 2>(stA, dr1s, inv_sc, wave);
    if constexpr (!RIN_EARLY) jstage<PREC, 1>(stB, rin, wave);
    __syncthreads();
    KT(0, 15);
    if (wave >= 2) accX = jdw_tile<PREC>(stA, wave - 2, RIN_EARLY ? stC : stB, 0, accX);
    bs1 += jrow_sum<PREC>(stA, 64, wave);
    KT(0, 16);
    float din[16];
    dense<PREC, 1, 2>(din, W + L.mat[M_R1T], dr1s, false);
#pragma unroll
    for (int r = 0; r < 16; ++r) din[r] = din[r] * inv_sc;
    // slots 19 (hi0,r11) 20,21 (hi1,r8,r9): gradient w.r.t. the normals fed to the radiance net
    const float v0 = hi == 0 ? din[11] : 0.f, v1 = hi == 1 ? din[8] : 0.f, v2 = hi == 1 ? din[9] : 0.f;
    gn[0] += v0 + wave_shfl_xor(v0, 32);
    gn[1] += v1 + wave_shfl_xor(v1, 32);
    gn[2] += v2 + wave_shfl_xor(v2, 32);
    if (p.valid && hi == 0) {
#pragma unroll
      for (int c = 0; c < 3; ++c) a.dnab_total[3 * s + c] = gn[c];
    }
    if (a.dx) {   // pose refinement: the radiance net's position input (slots 0-2: hi 0, r 0..2) ...
      if (p.valid && hi == 0) {
#pragma unroll
        for (int c = 0; c < 3; ++c) a.dx[3 * s + c] = din[c];
      }
    }
    if (a.dv) {   // ... and its view direction through SH4 (slots 3-18)
      float gsh[16];
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        const int slot = k + 3, ohi = (slot >> 2) & 1, r = (slot & 3) + 4 * (slot >> 3);
        const float v = (hi == ohi) ? din[r] : 0.f;
        gsh[k] = v + wave_shfl_xor(v, 32);
      }
      float gv[3];
      sh4_grad(p.vd, gsh, gv);
      if (p.valid && hi == 0) {
#pragma unroll
        for (int c = 0; c < 3; ++c) a.dv[3 * s + c] = gv[c];
      }
    }
    // slots 22,23 (hi1,r10,r11) 24,25 (hi0,r12,r13): appearance embedding gradient
    // (summed over the samples of a ray inside the wave: one atomic per ray, channel and wave, not one per sample)
    if (a.dh_appear && !(a.ablate & 8)) {
      float c0 = hi == 1 ? din[10] : din[12], c1 = hi == 1 ? din[11] : din[13];
      if (halfwave_run_sum2(p.ray, p.valid, c0, c1)) {
        // (an offset the compiler cannot hoist: the lane-constant half of this address was one of the values it kept in -- and
        // spilled from -- registers across the whole loop)
        float* dst = a.dh_appear + 4 * p.ray + ((lane + nsim_opaque_zero()) >> 5 == 1 ? 0 : 2);
        atomicAdd(dst, c0);
        atomicAdd(dst + 1, c1);
      }
    }
    KT(0, 17);
  }
  { const int64_t grp = -1; KT(0, 21); }
  // ---- one flush per wave
  const SrcOff so = src_off(1);
  const int64_t ro = (int64_t)(blockIdx.x & (unsigned)a.rep_mask) * a.rep_stride;      // this workgroup's replica
  jflush_tile(a.drad_w + ro + so.r2, 64, 64, 64, wave >> 1, wave & 1, accR2);
  if (wave < 2) jflush_tile(a.drad_w + ro + so.r3, 64, 3, 64, 0, wave, accX);
  else jflush_tile(a.drad_w + ro + so.r1, 26, 64, 26, wave - 2, 0, accX);
  if (bs1 != 0.f) atomicAdd(&a.drad_b[ro + so.rb1 + lane], bs1);
  if (bs2 != 0.f) atomicAdd(&a.drad_b[ro + so.rb2 + lane], bs2);
  if (lane < 3 && bs3 != 0.f) atomicAdd(&a.drad_b[ro + so.rb3 + lane], bs3);
  { const int64_t grp = -1; KT(0, 22); }
}

// ------------------------------------------------------------------------------------ grid scatter
// dgrid[level][vertex][f] += w_c * dh[f] + g[f] * dscale * (dw_c . gn)      (first + second order terms)
//
// Measured on MI355X (tools/atomic_bench*.hip): f32 atomics retire at ~21 G *64-byte-line requests*/s chip-wide,
// independent of scope, and lanes of ONE instruction that hit adjacent dwords share a request (x2/x4/x16 lane-ops
// per request).  So: one block row per LEVEL (the level's table stays L2-resident), one lane per sample with
// consecutive lanes = consecutive samples of a ray; (1) on coarse levels whole runs of lanes hit the SAME vertex and
// are collapsed by a segmented shuffle reduction; (2) the remaining adds are issued quad-transposed: the 4 lanes of
// a quad serve one sample at a time and write the 4 dwords {x0.f0, x0.f1, x1.f0, x1.f1} of an x-adjacent corner
// pair -- one 16-byte request on dense levels (and on hashed levels whenever x0 is even) instead of four.
struct ScatterArgs {
  LotdDev lotd;
  const float *x, *rays_o, *rays_d, *t;
  const int64_t* ridx;
  int64_t S;
  const float *dh_pl, *g_pl, *gn;
  const int64_t* ray_goff;
  float* dgrid;
  int dedup_max_res;
  int level_begin;      // this launch covers levels [level_begin, level_begin + gridDim.y)
  int parity_slots;     // the eight vertices of a cell are enumerated by the PARITY of their coordinates (round 6, default on)
};

// CONSEC (default since round 6; NSIM_SCATTER_GROUP=0: the strided form of rounds 1-5): which 16 samples of the wave's 64 share
// one atomic instruction.  The atomic unit retires one request per distinct 64-byte sector per wave instruction at ~21 G/s, and
// lanes of one instruction on the SAME address are separate requests (profiles/round4_atomic_line_bench.txt,
// profiles/round6_atomic_conflict_bench.txt) -- so a launch costs the number of such requests, which tools/scatter_sector_model.py
// counts on a dumped step's sample set (25.1 modelled, 24.9 measured per point for the rounds 1-5 kernel).  Issue I of the quad
// transposition used to take the samples 4q + I (a DPP quad broadcast); it now takes the 16 CONSECUTIVE samples 16 I + q (six
// ds_bpermute per issue): neighbours on a ray, whose vertex rows share sectors.  On its own that measured nothing (neighbouring
// cells share VERTICES, i.e. addresses: 0.376 ms either way); with the parity slots below, which fold the shared vertices
// first, it is worth another 8 %.  MI355X, bench step / street step (profiles/round6_scatter_requests.json):
//     slots by corner offset, strided issue (rounds 1-5)   0.376 ms   3.32 ms    25.1 / 43.5 requests per point (model)
//     parity slots, strided issue                           0.309      2.84       20.4 / 37.5
//     parity slots, consecutive issue (default)             0.286      2.47       17.8 / 32.4
#ifndef NSIM_SCATTER_MIN_WAVES
#define NSIM_SCATTER_MIN_WAVES 5      // waves per SIMD the register allocation leaves room for: 98 -> 96 registers, five waves instead of
                                      // four (MI355X, bench step: 0.2865 -> 0.2811 ms; 6 -- 80 registers + 64 B scratch -- 0.328 ms)
#endif
template <bool CONSEC>
__global__ void __launch_bounds__(256, NSIM_SCATTER_MIN_WAVES) k_lotd_scatter(ScatterArgs a) {
  const int lane = nsim_lane();
  const int l = a.level_begin + (int)blockIdx.y;
  const LotdRes R = a.lotd.res[l];
  if (l >= a.lotd.n_active) return;     // masked level: no gradient
  const bool dedup = R.max() <= a.dedup_max_res;
  const int rq = lane & 3;
  float* base = a.dgrid + a.lotd.offset[l];
  const int64_t nchunks = (a.S + 63) / 64;
  const int64_t wstride = (int64_t)gridDim.x * 4;
  for (int64_t chunk = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); chunk < nchunks; chunk += wstride) {
    const int64_t s = chunk * 64 + lane;
    const bool valid = s < a.S;
    float xx[3] = {0.f, 0.f, 0.f}, gn[3] = {0.f, 0.f, 0.f}, dh0 = 0.f, dh1 = 0.f, g0 = 0.f, g1 = 0.f;
    uint32_t voff = 0u;      // batched model: vertex offset of the sample's instance (also separates the dedup keys)
    if (valid) {
      if (a.ray_goff) voff = (uint32_t)(a.ray_goff[a.ridx[s]] >> 1);
      if (a.x) {
        xx[0] = a.x[3 * s]; xx[1] = a.x[3 * s + 1]; xx[2] = a.x[3 * s + 2];
      } else {
        const int64_t ray = a.ridx[s];
        const float tt = a.t[s];
#pragma unroll
        for (int c = 0; c < 3; ++c) xx[c] = a.rays_o[3 * ray + c] + tt * a.rays_d[3 * ray + c];
      }
      const float* dp = a.dh_pl + ((int64_t)l * a.S + s) * 2;
      const float* gp = a.g_pl + ((int64_t)l * a.S + s) * 2;
      dh0 = dp[0]; dh1 = dp[1];
      g0 = gp[0]; g1 = gp[1];
      if (a.gn) {
#pragma unroll
        for (int c = 0; c < 3; ++c) gn[c] = a.gn[3 * s + c];
      }
    }
    const LotdCell c = lotd_cell(xx, R, a.lotd);
    const float q0[3] = {g0 * gn[0] * c.dscale[0], g0 * gn[1] * c.dscale[1], g0 * gn[2] * c.dscale[2]};
    const float q1[3] = {g1 * gn[0] * c.dscale[0], g1 * gn[1] * c.dscale[1], g1 * gn[2] * c.dscale[2]};
    const int px0 = a.parity_slots ? (c.c0[0] & 1) : 0, py0 = a.parity_slots ? (c.c0[1] & 1) : 0, pz0 = a.parity_slots ? (c.c0[2] & 1) : 0;
#pragma unroll
    for (int yz = 0; yz < 4; ++yz) {
      uint32_t idx[2];
      float v0[2], v1[2];
      bool emit[2];
#pragma unroll
      for (int dx = 0; dx < 2; ++dx) {
        // Parity slots: slot (dx, yz) holds the vertex whose coordinates have the parities (dx, yz & 1, yz >> 1) -- the eight
        // vertices of a cell have the eight parity combinations exactly once -- so a vertex SHARED by neighbouring cells sits in
        // the same slot in both, and the run merge below (equal vertex index on consecutive lanes) folds the four vertices of the
        // face two consecutive cells on a ray share, chains of cells included, with no further logic.  Atomic requests per point
        // by the request model (tools/scatter_sector_model.py): 25.1 -> 20.4 on the object step, 43.5 -> 37.5 on the street step.
        const int corner = (dx ^ px0) | (((yz & 1) ^ py0) << 1) | (((yz >> 1) ^ pz0) << 2);
        float w, dw[3];
        lotd_corner_w(c, corner, w, dw);
        idx[dx] = voff + lotd_index(c.c0[0] + (corner & 1), c.c0[1] + ((corner >> 1) & 1), c.c0[2] + (corner >> 2), R, a.lotd.type[l],
                                    a.lotd.size[l]);
        v0[dx] = w * dh0 + (dw[0] * q0[0] + dw[1] * q0[1] + dw[2] * q0[2]);
        v1[dx] = w * dh1 + (dw[0] * q1[0] + dw[1] * q1[1] + dw[2] * q1[2]);
        emit[dx] = valid;
        if (dedup) {
          // contiguous runs of equal vertex index across the wave: run heads by ballot, run start = highest
          // head bit at or below this lane, then a shuffle scan limited to the run
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
          emit[dx] = valid && (lane == 63 || ((heads >> (lane + 1)) & 1ull));  // last lane of the run
        }
      }
#define NSIM_QUAD_ISSUE(I)                                                                                   \
  {                                                                                                          \
    const uint32_t i0 = quad_bcast<I>(idx[0]), i1 = quad_bcast<I>(idx[1]);                                   \
    const float a0 = quad_bcast<I>(v0[0]), a1 = quad_bcast<I>(v1[0]);                                        \
    const float b0 = quad_bcast<I>(v0[1]), b1 = quad_bcast<I>(v1[1]);                                        \
    const int e0 = quad_bcast<I>((int)emit[0]), e1 = quad_bcast<I>((int)emit[1]);                            \
    const uint32_t ii = rq < 2 ? i0 : i1;                                                                    \
    const float vv = rq == 0 ? a0 : (rq == 1 ? a1 : (rq == 2 ? b0 : b1));                                    \
    const int ee = rq < 2 ? e0 : e1;                                                                         \
    if (ee) atomicAdd(base + 2 * (int64_t)ii + (rq & 1), vv);                                                \
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
        NSIM_QUAD_ISSUE(0)
        NSIM_QUAD_ISSUE(1)
        NSIM_QUAD_ISSUE(2)
        NSIM_QUAD_ISSUE(3)
      }
#undef NSIM_QUAD_ISSUE
    }
  }
}

// ------------------------------------------------------------------------------------- pose refinement
// dL/dx of the normals' own position dependence: nablas_c' = sum_f g_f J_f,c'(x), and inside a cell the trilinear
// interpolant has mixed second derivatives only:  H_f[c'][c] = dscale_c' dscale_c sum_corner s_c' s_c w_third grid_f
// (c != c').  dx[c] += sum_f g_f sum_{c' != c} gn_c' H_f[c'][c].  One lane per sample over all levels (point-major:
// this launch exists only when the rays carry gradients, e.g. StreetSurf's pose refinement,
// withmask_withlidar_joint.240219.yaml:338-352), g from the hand-off planes of the SDF-branch backward.
struct HessArgs {
  LotdDev lotd;
  const f16* grid;
  const float *x, *rays_o, *rays_d, *t;
  const int64_t* ridx;
  const int64_t* ray_goff;
  int64_t S;
  const float *g_pl, *gn;
  float* dx;
};

__global__ void __launch_bounds__(256) k_lotd_hess_dx(HessArgs a) {
  const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= a.S) return;
  float xx[3];
  uint32_t goff = 0u;
  if (a.x) {
    xx[0] = a.x[3 * s]; xx[1] = a.x[3 * s + 1]; xx[2] = a.x[3 * s + 2];
    if (a.ray_goff) goff = (uint32_t)a.ray_goff[a.ridx[s]];
  } else {
    const int64_t ray = a.ridx[s];
    const float tt = a.t[s];
#pragma unroll
    for (int c = 0; c < 3; ++c) xx[c] = a.rays_o[3 * ray + c] + tt * a.rays_d[3 * ray + c];
    if (a.ray_goff) goff = (uint32_t)a.ray_goff[ray];
  }
  const float gn[3] = {a.gn[3 * s], a.gn[3 * s + 1], a.gn[3 * s + 2]};
  const GridRef gref = grid_ref(a.grid);
  float acc[3] = {0.f, 0.f, 0.f};
  for (int l = 0; l < a.lotd.n_active; ++l) {
    const LotdRes R = a.lotd.res[l];
    const LotdCell c = lotd_cell(xx, R, a.lotd);
    const float* gp = a.g_pl + ((int64_t)l * a.S + s) * 2;
    const float g0 = gp[0], g1 = gp[1];
    float hxy = 0.f, hxz = 0.f, hyz = 0.f;      // already contracted with g over the two features
#pragma unroll
    for (int corner = 0; corner < 8; ++corner) {
      const int bx = corner & 1, by = (corner >> 1) & 1, bz = (corner >> 2) & 1;
      const uint32_t idx = lotd_index(c.c0[0] + bx, c.c0[1] + by, c.c0[2] + bz, R, a.lotd.type[l], a.lotd.size[l]);
      float f0, f1;
      lotd_load2(gref, (uint32_t)a.lotd.offset[l] + goff + 2u * idx, f0, f1);
      const float v = g0 * f0 + g1 * f1;
      const float sx = bx ? 1.0f : -1.0f, sy = by ? 1.0f : -1.0f, sz = bz ? 1.0f : -1.0f;
      const float wx = bx ? c.w[0] : 1.0f - c.w[0], wy = by ? c.w[1] : 1.0f - c.w[1], wz = bz ? c.w[2] : 1.0f - c.w[2];
      hxy = hxy + sx * sy * wz * v;
      hxz = hxz + sx * sz * wy * v;
      hyz = hyz + sy * sz * wx * v;
    }
    hxy = hxy * (c.dscale[0] * c.dscale[1]);
    hxz = hxz * (c.dscale[0] * c.dscale[2]);
    hyz = hyz * (c.dscale[1] * c.dscale[2]);
    acc[0] = acc[0] + gn[1] * hxy + gn[2] * hxz;
    acc[1] = acc[1] + gn[0] * hxy + gn[2] * hyz;
    acc[2] = acc[2] + gn[0] * hxz + gn[1] * hyz;
  }
#pragma unroll
  for (int c = 0; c < 3; ++c) a.dx[3 * s + c] = a.dx[3 * s + c] + acc[c];
}

// x_s = o_r + t_s d_r, v_s = d_r:  d o_r += dx_s,  d d_r += t_s dx_s + dv_s   (samples of a ray are neighbours: the
// atomics of one wave hit a handful of addresses)
__global__ void __launch_bounds__(256) k_ray_grad_reduce(const float* __restrict__ dx, const float* __restrict__ dv,
                                                         const float* __restrict__ t, const int64_t* __restrict__ ridx,
                                                         int64_t S, float* __restrict__ d_o, float* __restrict__ d_d) {
  const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= S) return;
  const int64_t r = ridx[s];
  const float tt = t[s];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float g = dx[3 * s + c];
    if (d_o) atomicAdd(&d_o[3 * r + c], g);
    if (d_d) atomicAdd(&d_d[3 * r + c], tt * g + (dv ? dv[3 * s + c] : 0.f));
  }
}

// ------------------------------------------------------------------------------------- self test
__global__ void __launch_bounds__(64) k_selftest_mfma(const float* __restrict__ A, const float* __restrict__ B,
                                                       float* __restrict__ D, int use_f32) {
  // D[32x32] = A[32x16] * B[16x32] (row-major), computed only through the documented operand contract:
  // lane (i,hi) supplies row i of A / column i of B for the K slots (hi, e) <-> k = 8*hi + e.
  const int lane = nsim_lane(), i = lane & 31, hi = lane >> 5;
  f32x16 acc = zero16();
  if (!use_f32) {
    f16x8 a, b;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      a[e] = (f16)A[i * 16 + 8 * hi + e];
      b[e] = (f16)B[(8 * hi + e) * 32 + i];
    }
    acc = mfma_32x32x16_f16(a, b, acc);
  } else {
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int k = 2 * q + hi;
      acc = mfma_32x32x2_f32(A[i * 16 + k], B[k * 32 + i], acc);
    }
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) D[mfma_row(r, hi) * 32 + i] = acc[r];
}

// ================================================================================== C ABI
static int field_meta_check(const NsimFieldMeta* m) {
  if (!m) return 20;
  const int rc = lotd_meta_check(&m->lotd);
  if (rc) return rc;
  if (m->lotd.num_levels < 1 || m->lotd.num_levels > 32) return 21;
  if (m->sdf_D != 1 && m->sdf_D != 2) return 22;
  if (m->precision != 0 && m->precision != 1 && m->precision != 2) return 23;
  if (m->embed_E < 0 || m->embed_E > 63 || (m->embed_E > 0 && (m->embed_E - 3) % 6 != 0)) return 36;
  return 0;
}
// precision 2 (split f16: f32-equivalent arithmetic on the f16 matrix cores) exists for the no-grad SDF query only
static int field_meta_check_full(const NsimFieldMeta* m) {
  const int rc = field_meta_check(m);
  return rc ? rc : (m->precision == 2 ? 23 : 0);
}

// ------------------------------------------------------------------------------------ weight-gradient replicas
// The joint backward kernels end with every workgroup adding its partial weight gradients into the SAME ~6 k floats.
// 256-512 same-address atomics per address serialise in L2 at ~0.1 us each: s_memtime stamps (tools/ktime.py, round 3)
// show the last group of the radiance backward waiting ~55 us of a 100 us launch behind that storm.  With a caller-owned,
// zeroed scratch registered for the stream (nsim_set_grad_scratch), workgroup b flushes into replica b % R instead and a
// small second launch folds the R copies into the caller's gradient buffers (and zeroes the scratch again).
struct GradScratch {
  int device;      // the registry is keyed by (device, stream): the default stream has handle 0 on EVERY device of a process
  void* stream;
  float* buf;
  int64_t floats;
};
static GradScratch g_grad_scratch[64];
static int g_grad_scratch_n = 0;

__global__ void __launch_bounds__(256) k_rep_reduce(float* __restrict__ sc, int R, int64_t stride, int64_t n_w,
                                                    float* __restrict__ dst_w, float* __restrict__ dst_b, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float s = 0.f;
  for (int r = 0; r < R; ++r) {
    s += sc[r * stride + i];
    sc[r * stride + i] = 0.f;
  }
  if (s != 0.f) {
    float* d = i < n_w ? dst_w + i : dst_b + (i - n_w);
    *d = *d + s;
  }
}

static int grad_replicas_min_wg() {      // launches with fewer workgroups have no storm to avoid (tests lower it)
  const char* e = getenv("NSIM_GRAD_REPLICAS_MIN_WG");
  return e ? atoi(e) : 64;
}

static int grad_replicas() {
  static const int r = [] {
    const char* e = getenv("NSIM_GRAD_REPLICAS");
    int v = e ? atoi(e) : 16;
    int p = 1;
    while (p * 2 <= v && p < 64) p *= 2;      // a power of two, 1 = off
    return v <= 1 ? 1 : p;
  }();
  return r;
}

// the registered scratch of ``stream`` if it holds R x n floats, else NULL (= flush into the caller's buffers directly)
static int current_device() {
  int d = 0;
  if (hipGetDevice(&d) != hipSuccess) d = 0;
  return d;
}

static float* grad_scratch(void* stream, int64_t n, int& R) {
  R = grad_replicas();
  if (R <= 1) return nullptr;
  const int dev = current_device();
  for (int i = 0; i < g_grad_scratch_n; ++i)
    if (g_grad_scratch[i].device == dev && g_grad_scratch[i].stream == stream && g_grad_scratch[i].buf &&
        g_grad_scratch[i].floats >= (int64_t)R * n)
      return g_grad_scratch[i].buf;
  return nullptr;
}

static void grad_scratch_fold(float* sc, int R, int64_t n_w, int64_t n_b, float* dst_w, float* dst_b, hipStream_t stream) {
  const int64_t n = n_w + n_b;
  hipLaunchKernelGGL(k_rep_reduce, dim3(nsim_blocks(n, 256)), dim3(256), 0, stream, sc, R, n, n_w, dst_w, dst_b, n);
}

static FieldArgs field_args(const NsimFieldMeta* meta) {
  FieldArgs a = FieldArgs();
  a.lotd = lotd_dev(&meta->lotd);
  a.lay = field_layout(meta->precision, field_ni(meta));
  a.embed_E = meta->embed_E;
  a.beta = meta->softplus_beta > 0.f ? meta->softplus_beta : 1e30f;      // (<= 0: relu, see softplus_exact)
  static const int code_pf = getenv("NSIM_CODE_PREFETCH") ? atoi(getenv("NSIM_CODE_PREFETCH")) : 0;
  a.code_pf = code_pf;
  return a;
}

static unsigned field_grid(int64_t S, int64_t max_blocks, int waves = FIELD_WAVES) {
  const int64_t tiles = (S + 31) / 32;
  int64_t b = (tiles + waves - 1) / waves;
  if (b > max_blocks) b = max_blocks;
  if (b < 1) b = 1;
  return (unsigned)b;
}

static size_t weights_lds_bytes(const NsimFieldMeta* meta, int first = 0, int count = M_COUNT) {
  if (meta->precision == 1) return 0;
  const FieldLayout L = field_layout(meta->precision, field_ni(meta));
  const int64_t m1 = (first + count < M_COUNT) ? L.mat[first + count] : L.vec[0];
  return (size_t)(((m1 - L.mat[first]) + (L.total - L.vec[0]) + 15) & ~15);
}
static size_t stage_bytes(const NsimFieldMeta* meta) {
  return meta->precision == 0 ? stage_bytes_per_wave<0>() : stage_bytes_per_wave<1>();
}

// waves per workgroup of k_field<., ., MODE, NC> (see NW in the kernel)
static int field_waves(const NsimFieldMeta* meta, int mode) {
  return (mode == 2 && meta->precision == 0 && field_nc(meta->lotd.num_levels) == 2) ? 3 : FIELD_WAVES;
}

template <int MODE>
static int field_launch(const NsimFieldMeta* meta, const FieldArgs& a, size_t shmem, int64_t max_blocks,
                        hipStream_t stream, bool gl2 = false) {
  const int nw = field_waves(meta, MODE);
  const dim3 grid(field_grid(a.S, max_blocks, nw)), block(64 * nw);
  const int key = meta->precision * 2 + (meta->sdf_D - 1);
  if (field_ne(meta)) {      // embedded-position block behind the features: the forward on the planes only
    if constexpr (MODE == 3) {
      switch ((field_nc(meta->lotd.num_levels) - 1) * 4 + key) {
        case 0: hipLaunchKernelGGL((k_field<0, 1, 3, 1, false, 2>), grid, block, shmem, stream, a); break;
        case 1: hipLaunchKernelGGL((k_field<0, 2, 3, 1, false, 2>), grid, block, shmem, stream, a); break;
        case 2: hipLaunchKernelGGL((k_field<1, 1, 3, 1, false, 2>), grid, block, shmem, stream, a); break;
        case 3: hipLaunchKernelGGL((k_field<1, 2, 3, 1, false, 2>), grid, block, shmem, stream, a); break;
        case 4: hipLaunchKernelGGL((k_field<0, 1, 3, 2, false, 2>), grid, block, shmem, stream, a); break;
        case 5: hipLaunchKernelGGL((k_field<0, 2, 3, 2, false, 2>), grid, block, shmem, stream, a); break;
        case 6: hipLaunchKernelGGL((k_field<1, 1, 3, 2, false, 2>), grid, block, shmem, stream, a); break;
        case 7: hipLaunchKernelGGL((k_field<1, 2, 3, 2, false, 2>), grid, block, shmem, stream, a); break;
      }
      NSIM_CHECK_LAUNCH();
      return 0;
    } else {
      return 36;
    }
  }
  if (field_nc(meta->lotd.num_levels) == 2) {
    if constexpr (MODE == 3) {
      if (gl2 && meta->precision == 0) {      // plane image in LDS (fp16 mode, <= 23 levels)
        if (meta->sdf_D == 1) hipLaunchKernelGGL((k_field<0, 1, 3, 2, true>), grid, block, shmem, stream, a);
        else hipLaunchKernelGGL((k_field<0, 2, 3, 2, true>), grid, block, shmem, stream, a);
        NSIM_CHECK_LAUNCH();
        return 0;
      }
    }
    if constexpr (MODE == 2 || MODE == 3) {
      switch (key) {
        case 0: hipLaunchKernelGGL((k_field<0, 1, MODE, 2>), grid, block, shmem, stream, a); break;
        case 1: hipLaunchKernelGGL((k_field<0, 2, MODE, 2>), grid, block, shmem, stream, a); break;
        case 2: hipLaunchKernelGGL((k_field<1, 1, MODE, 2>), grid, block, shmem, stream, a); break;
        case 3: hipLaunchKernelGGL((k_field<1, 2, MODE, 2>), grid, block, shmem, stream, a); break;
      }
      NSIM_CHECK_LAUNCH();
      return 0;
    } else {
      return 33;      // more than 16 levels: only the level-major (planes) path exists
    }
  }
  switch (key) {
    case 0: hipLaunchKernelGGL((k_field<0, 1, MODE>), grid, block, shmem, stream, a); break;
    case 1: hipLaunchKernelGGL((k_field<0, 2, MODE>), grid, block, shmem, stream, a); break;
    case 2: hipLaunchKernelGGL((k_field<1, 1, MODE>), grid, block, shmem, stream, a); break;
    case 3: hipLaunchKernelGGL((k_field<1, 2, MODE>), grid, block, shmem, stream, a); break;
  }
  NSIM_CHECK_LAUNCH();
  return 0;
}

static int bwd_ablate() {
  const char* ab = getenv("NSIM_ABLATE");
  return ab ? atoi(ab) : 0;
}

// persistent grids: the packed weights (58 KB) are staged into LDS once per workgroup
#define FIELD_GRID_FWD 1024
#define FIELD_GRID_BWD 256      // one resident workgroup per CU (LDS-limited): persistent waves amortise the accumulator flush

extern "C" {

int64_t nsim_field_wpack_bytes(const NsimFieldMeta* meta) {
  if (field_meta_check(meta)) return -1;
  return field_layout(meta->precision, field_ni(meta)).total;
}

int nsim_field_pack_weights(const NsimFieldMeta* meta, const float* sdf_w, const float* sdf_b, const float* rad_w,
                            const float* rad_b, void* wpack, void* stream) {
  const int rc = field_meta_check(meta);
  if (rc) return rc;
  const int nc = field_ni(meta);
  const FieldLayout L = field_layout(meta->precision, nc);
  PackDims dims;
  dims.E = meta->embed_E;
  dims.EB = 32 * field_nc(meta->lotd.num_levels);
  int64_t total = 0;
  for (int m = 0; m < M_COUNT; ++m) {
    dims.uo[m] = mat_uo(m, nc);
    dims.ui[m] = mat_ui(m, nc);
    total += (int64_t)dims.uo[m] * dims.ui[m];
  }
  total += (int64_t)V_COUNT * 64;
  hipLaunchKernelGGL(k_field_pack, dim3(nsim_blocks(total, 256)), dim3(256), 0, (hipStream_t)stream, L, dims,
                     meta->sdf_D, 2 * meta->lotd.num_levels, sdf_w, sdf_b, rad_w, rad_b, (char*)wpack);
  NSIM_CHECK_LAUNCH();
  return 0;
}

int nsim_jplane_elem_bytes(const NsimFieldMeta* meta) { return meta ? NSIM_J_ELEM_BYTES(meta->precision) : 0; }

int nsim_field_pack_weights2(const NsimFieldMeta* meta_a, void* wpack_a, const NsimFieldMeta* meta_b, void* wpack_b,
                             const float* sdf_w, const float* sdf_b, const float* rad_w, const float* rad_b, void* stream) {
  int rc = field_meta_check(meta_a);
  if (rc) return rc;
  rc = field_meta_check(meta_b);
  if (rc) return rc;
  if (meta_a->sdf_D != meta_b->sdf_D || meta_a->lotd.num_levels != meta_b->lotd.num_levels || meta_a->embed_E != meta_b->embed_E) return 2;
  if (!wpack_a || !wpack_b) return 4;
  const int nc = field_ni(meta_a);
  PackTwo p;
  const NsimFieldMeta* ms[2] = {meta_a, meta_b};
  int64_t most = 0;
  for (int q = 0; q < 2; ++q) {
    p.L[q] = field_layout(ms[q]->precision, nc);
    int64_t total = (int64_t)V_COUNT * 64;
    p.dims[q].E = meta_a->embed_E;
    p.dims[q].EB = 32 * field_nc(meta_a->lotd.num_levels);
    for (int m = 0; m < M_COUNT; ++m) {
      p.dims[q].uo[m] = mat_uo(m, nc);
      p.dims[q].ui[m] = mat_ui(m, nc);
      total += (int64_t)p.dims[q].uo[m] * p.dims[q].ui[m];
    }
    most = total > most ? total : most;
  }
  p.out[0] = (char*)wpack_a;
  p.out[1] = (char*)wpack_b;
  hipLaunchKernelGGL(k_field_pack2, dim3(nsim_blocks(most, 256), 2), dim3(256), 0, (hipStream_t)stream, p, meta_a->sdf_D,
                     2 * meta_a->lotd.num_levels, sdf_w, sdf_b, rad_w, rad_b);
  NSIM_CHECK_LAUNCH();
  return 0;
}

// Deal the levels to the XCDs so that every XCD streams few tables through its L2 AND the XCDs finish together:
// a hashed level costs 1, a dense one ~0.35 (its lines mostly hit L1); levels go, most expensive first, onto the least
// loaded XCD -- whole if that keeps the XCD within 8 % of the ideal load, otherwise split into two halves of the point
// range on two XCDs (3 of the 11 hashed levels of the default pyramid end up split: max load 1.7 instead of 2.0).
static void deal_levels(const NsimFieldMeta* meta, FieldArgs& a) {
  // the plane arrays hold 16 nc levels; NC == 2: the ones past num_levels are neither written here nor read by the decoders
  // 17..32 levels: the pyramid's own levels + zeros up to the next multiple of 8 (the sampling decoder's K-step), see LV_OK
  const int NL = field_nc(meta->lotd.num_levels) == 2 ? ((meta->lotd.num_levels + 7) & ~7) : 16;
  float cost[32], load[8] = {0, 0, 0, 0, 0, 0, 0, 0}, total = 0.f;
  bool used[32] = {false};
  for (int l = 0; l < NL; ++l) {
    cost[l] = l >= meta->lotd.num_levels ? (NL > 16 ? 0.05f : 0.35f) : (meta->lotd.type[l] == NSIM_LOTD_HASH ? 1.0f : 0.35f);
    total += cost[l];
  }
  const float limit = total / 8.0f * 1.08f;
  for (int xc = 0; xc < 8; ++xc) a.glm_n[xc] = 0;
  auto least = [&]() {
    int tx = 0;
    for (int xc = 1; xc < 8; ++xc)
      if (load[xc] < load[tx]) tx = xc;
    return tx;
  };
  auto put = [&](int xc, int l, int half, float c) {
    a.glm_lv[xc][(int)a.glm_n[xc]] = (signed char)l;
    a.glm_half[xc][(int)a.glm_n[xc]++] = (signed char)half;
    load[xc] += c;
  };
  auto tsize = [&](int l) { return l < meta->lotd.num_levels ? meta->lotd.size[l] : 0u; };
  for (int it = 0; it < NL; ++it) {
    int best = -1;
    for (int l = 0; l < NL; ++l)
      if (!used[l] && (best < 0 || cost[l] > cost[best] || (cost[l] == cost[best] && tsize(l) > tsize(best)))) best = l;
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

int nsim_lotd_gather_lm(const NsimFieldMeta* meta, const void* grid_f16, const float* x, const float* rays_o,
                        const float* rays_d, const float* t, const int64_t* ridx, const int64_t* ray_goff, int64_t S,
                        const int64_t* n_dev, int64_t n_add, void* feat_planes, void* stream) {
  const int rc = field_meta_check(meta);
  if (rc) return rc;
  if (S <= 0) return 0;
  if (!x && !(rays_o && rays_d && t && ridx)) return 24;
  if (ray_goff && !ridx) return 29;
  if (!feat_planes || !grid_f16) return 4;
  FieldArgs a = field_args(meta);
  a.grid = (const f16*)grid_f16;
  a.x = x; a.rays_o = rays_o; a.rays_d = rays_d; a.t = t; a.ridx = ridx;
  a.ray_goff = ray_goff;
  a.S = S;
  a.S_dev = n_dev;
  a.S_add = n_add;
  a.feat_pl = feat_planes;
  a.PS = NSIM_PLANE_PITCH(S);
  deal_levels(meta, a);
  if (meta->precision == 0) launch_gather_lm<0, false>(a, S, (hipStream_t)stream);
  else launch_gather_lm<1, false>(a, S, (hipStream_t)stream);   // f32 planes (1, 2)
  NSIM_CHECK_LAUNCH();
  return 0;
}

int nsim_field_sdf(const NsimFieldMeta* meta, const void* grid_f16, const void* wpack, const float* x,
                   const float* rays_o, const float* rays_d, const float* t, const int64_t* ridx,
                   const int64_t* ray_goff, int64_t S, const int64_t* n_dev, int64_t n_add, float* sdf,
                   const void* feat_planes, float* occ_val, const NsimOccMeta* occ_meta, float occ_inv_s, void* stream) {
  const int rc = field_meta_check(meta);
  if (rc) return rc;
  if (field_ne(meta)) return 36;      // a model with an embedded-position block queries through nsim_wide_sdf
  if (S <= 0) return 0;
  if (occ_val && !(x && occ_meta)) return 24;
  if (!feat_planes && !x && !(rays_o && rays_d && t && ridx)) return 24;
  if (ray_goff && !ridx) return 29;
  void* feat_scratch = const_cast<void*>(feat_planes);
  FieldArgs a = field_args(meta);
  a.grid = (const f16*)grid_f16;
  a.wpack = (const char*)wpack;
  a.x = x; a.rays_o = rays_o; a.rays_d = rays_d; a.t = t; a.ridx = ridx;
  a.ray_goff = ray_goff;
  a.S_dev = n_dev;                            // (round 5: the point-major form honours a device-side point count as well)
  a.S_add = n_add;
  a.S = S;
  a.PS = NSIM_PLANE_PITCH(S);
  a.sdf = sdf;
  a.feat_pl = feat_scratch;
  if (occ_val) {
    a.occ_val = occ_val;
    a.occ = occ_dev(occ_meta);
    a.occ_inv_s = occ_inv_s;
  }
  // 512 persistent workgroups (two per CU): at 2048 a wave staged 26 KB of weights for ~1.2 tiles of work
  // (0.0506 -> 0.0463 ms per 0.31 M points; 1024: 0.0483, 256: 0.0706)
  static const int sdf_grid_env = getenv("NSIM_SDF_GRID") ? atoi(getenv("NSIM_SDF_GRID")) : 0;
  // ... and about five tiles per wave beyond that (1.39 M points of the street step: 2048 workgroups 0.094 ms, 512: 0.134 --
  // with more tiles per wave the static stride balances worse than the dispatcher's backfill of finished workgroups)
  // (the 16-level kernel measures best at 512 for every launch size of the object step: 0.0463 ms avg against 0.0523 with
  // the scaled grid; the scaling is the 17..32-level kernel's)
  // (round 4, LDS-DMA form -- 128 VGPR, 42 KB LDS, three workgroups fit a CU: 768 -> 0.0352 ms, 512: 0.0378, 1024: 0.0372)
  int64_t sdf_grid = sdf_grid_env > 0 ? sdf_grid_env : (feat_planes ? 768 : 512);
  if (sdf_grid_env <= 0 && field_nc(meta->lotd.num_levels) == 2) {
    sdf_grid = ((S + 31) / 32) / (FIELD_WAVES * 5);
    sdf_grid = sdf_grid < 512 ? 512 : (sdf_grid > 4096 ? 4096 : sdf_grid);
  }
  const dim3 grid(field_grid(S, sdf_grid)), block(64 * FIELD_WAVES);
  size_t shmem = weights_lds_bytes(meta, 0, 2);
  const int key = meta->precision * 2 + (meta->sdf_D - 1);
  // (split precision without planes -- the fused point-major form -- exists for <= 16 levels since round 5: small launches)
  static const bool sdf_glds = !(getenv("NSIM_SDF_GLDS") && atoi(getenv("NSIM_SDF_GLDS")) == 0);
  const int nc = field_nc(meta->lotd.num_levels);
  const bool gl = sdf_glds && feat_scratch;
  if (gl)     // the LDS plane image(s) of every wave behind the weights: 2 x (2 | 4) KB for <= 16 levels, 1 x (4 | 8) KB above
    shmem = ((shmem + 15) & ~(size_t)15) + (size_t)FIELD_WAVES * (nc == 1 ? NSIM_SDF_NBUF : 1) * 16 * nc * (meta->precision == 0 ? 128 : 256);
  hipStream_t st = (hipStream_t)stream;
#define NSIM_SDF_LAUNCH(P, D, N)                                                                                  \
  do {                                                                                                            \
    if (gl) hipLaunchKernelGGL((k_field_sdf<P, D, true, N, true>), grid, block, shmem, st, a);                    \
    else hipLaunchKernelGGL((k_field_sdf<P, D, true, N, false>), grid, block, shmem, st, a);                      \
  } while (0)
  if (nc == 2) {
    if (!feat_scratch) return 33;     // more than 16 levels: level-major path only
    switch (key) {
      case 0: NSIM_SDF_LAUNCH(0, 1, 2); break;
      case 1: NSIM_SDF_LAUNCH(0, 2, 2); break;
      case 2: NSIM_SDF_LAUNCH(1, 1, 2); break;
      case 3: NSIM_SDF_LAUNCH(1, 2, 2); break;
      case 4: NSIM_SDF_LAUNCH(2, 1, 2); break;
      case 5: NSIM_SDF_LAUNCH(2, 2, 2); break;
    }
  } else if (feat_scratch) {   // decoder on the planes gathered by nsim_lotd_gather_lm
    switch (key) {
      case 0: NSIM_SDF_LAUNCH(0, 1, 1); break;
      case 1: NSIM_SDF_LAUNCH(0, 2, 1); break;
      case 2: NSIM_SDF_LAUNCH(1, 1, 1); break;
      case 3: NSIM_SDF_LAUNCH(1, 2, 1); break;
      case 4: NSIM_SDF_LAUNCH(2, 1, 1); break;
      case 5: NSIM_SDF_LAUNCH(2, 2, 1); break;
    }
#undef NSIM_SDF_LAUNCH
  } else {
    switch (key) {
      case 0: hipLaunchKernelGGL((k_field_sdf<0, 1, false>), grid, block, shmem, (hipStream_t)stream, a); break;
      case 1: hipLaunchKernelGGL((k_field_sdf<0, 2, false>), grid, block, shmem, (hipStream_t)stream, a); break;
      case 2: hipLaunchKernelGGL((k_field_sdf<1, 1, false>), grid, block, shmem, (hipStream_t)stream, a); break;
      case 3: hipLaunchKernelGGL((k_field_sdf<1, 2, false>), grid, block, shmem, (hipStream_t)stream, a); break;
      case 4: hipLaunchKernelGGL((k_field_sdf<2, 1, false>), grid, block, shmem, (hipStream_t)stream, a); break;
      case 5: hipLaunchKernelGGL((k_field_sdf<2, 2, false>), grid, block, shmem, (hipStream_t)stream, a); break;
    }
  }
  NSIM_CHECK_LAUNCH();
  return 0;
}

int nsim_field_fwd(const NsimFieldMeta* meta, const void* grid_f16, const void* wpack, const float* x,
                   const float* rays_o, const float* rays_d, const float* t, const int64_t* ridx,
                   const int64_t* ray_goff, const float* h_appear, int64_t S, float* sdf, float* nablas, float* rgb,
                   float* h_planes, void* J_planes, const int64_t* n_dev, int64_t n_add, void* stream) {
  const int rc = field_meta_check_full(meta);
  if (rc) return rc;
  if (S <= 0) return 0;
  if (ray_goff && !ridx) return 29;
  if (n_dev && !h_planes) return 28;          // the device-side point count is a feature of the level-major path
  if (!x && !(rays_o && rays_d && t && ridx)) return 24;
  if (rgb && !(rays_d && ridx)) return 25;
  if ((h_planes != nullptr) != (J_planes != nullptr)) return 28;
  FieldArgs a = field_args(meta);
  a.grid = (const f16*)grid_f16;
  a.wpack = (const char*)wpack;
  a.x = x; a.rays_o = rays_o; a.rays_d = rays_d; a.t = t; a.ridx = ridx;
  a.h_appear = h_appear;
  a.ray_goff = ray_goff;
  a.S = S;
  a.PS = NSIM_PLANE_PITCH(S);
  a.S_dev = n_dev;
  a.S_add = n_add;
  a.sdf = sdf; a.nablas = nablas; a.rgb = rgb;
  a.h_pl = h_planes; a.J_pl = J_planes;
  a.has_rgb = rgb ? 1 : 0;
  static const bool fused = getenv("NSIM_FWD_FUSED") && atoi(getenv("NSIM_FWD_FUSED")) == 1;
  if (!grid_f16 && !h_planes) return 4;
  if (h_planes && (!grid_f16 || !fused || n_dev || field_nc(meta->lotd.num_levels) == 2)) {      // training: level-major gather into the planes, then the decoders on the planes
    if (grid_f16) {      // (NULL: the caller's encoding has filled the planes already -- nsim_permuto_gather)
      deal_levels(meta, a);
      if (meta->precision == 0) launch_gather_lm<0, true>(a, S, (hipStream_t)stream);
      else launch_gather_lm<1, true>(a, S, (hipStream_t)stream);
      if (!wpack) {      // gather only: the caller runs another decoder on the planes (wide_field.hip)
        NSIM_CHECK_LAUNCH();
        return 0;
      }
    }
    // <= 16 levels: + one 16 KB plane-prefetch buffer per wave (k_field GLDS); 17..32 levels, fp16: 1 KB per level and wave
    // where that still fits the 160 KB of a CU (NSIM_FWD_GL2=0: the direct-load kernel)
    size_t wl = weights_lds_bytes(meta);
    if (meta->precision != 0) wl = 0;
    if (field_ne(meta))      // the forward matrices only, no plane image: 66 KB (<= 16 levels) -> two workgroups per CU
      return field_launch<3>(meta, a, weights_lds_bytes(meta, 0, M_R3 + 1), 512, (hipStream_t)stream, false);
    size_t pf_bytes = field_nc(meta->lotd.num_levels) == 1 ? (size_t)FIELD_WAVES * (NSIM_FWD_JDIRECT ? 4096 : 16384) : 0;
    bool gl2 = false;
    if (field_nc(meta->lotd.num_levels) == 2 && meta->precision == 0) {
      static const bool gl2_on = !(getenv("NSIM_FWD_GL2") && atoi(getenv("NSIM_FWD_GL2")) == 0);
      const size_t img = (size_t)FIELD_WAVES * 1024 * meta->lotd.num_levels;
      if (gl2_on && wl + img <= 160 * 1024) {
        gl2 = true;
        pf_bytes = img;
      }
    }
    // persistent workgroups: with the plane-prefetch buffers one workgroup fits a CU (124 KB of LDS), 1024 of them ran as
    // four rounds that each staged the weights and took a cold first tile (s_memtime stamps: 22.5 k vs 16.7 k ticks) --
    // one round of 256: 0.197 -> 0.189 ms per 0.31 M points (gather included); the 17..32-level kernel fits two per CU
    static const int fwd_grid_env = getenv("NSIM_FWD_GRID") ? atoi(getenv("NSIM_FWD_GRID")) : 0;
    const int fwd_grid = fwd_grid_env > 0 ? fwd_grid_env : ((pf_bytes && !(NSIM_FWD_JDIRECT && !gl2)) ? 256 : 512);
    return field_launch<3>(meta, a, wl + pf_bytes, fwd_grid, (hipStream_t)stream, gl2);
  }
  if (field_ne(meta)) return 36;      // the embedded-position block exists on the level-major path (h_planes / J_planes) only
  return field_launch<1>(meta, a, weights_lds_bytes(meta), FIELD_GRID_FWD, (hipStream_t)stream);
}

int nsim_set_grad_scratch(float* buf, int64_t floats, void* stream) {
  const int dev = current_device();
  for (int i = 0; i < g_grad_scratch_n; ++i)
    if (g_grad_scratch[i].device == dev && g_grad_scratch[i].stream == stream) {
      g_grad_scratch[i].buf = buf;
      g_grad_scratch[i].floats = buf ? floats : 0;
      return 0;
    }
  if (!buf) return 0;
  if (g_grad_scratch_n >= 64) return 34;
  g_grad_scratch[g_grad_scratch_n].device = dev;
  g_grad_scratch[g_grad_scratch_n].stream = stream;
  g_grad_scratch[g_grad_scratch_n].buf = buf;
  g_grad_scratch[g_grad_scratch_n++].floats = floats;
  return 0;
}

int nsim_field_bwd_rad(const NsimFieldMeta* meta, const void* wpack, const float* nablas_fwd, const float* rgb_fwd,
                       const float* x, const float* rays_o, const float* rays_d, const float* t, const int64_t* ridx,
                       const float* h_appear, int64_t S, const float* dnablas, const float* drgb, float* gn_out,
                       float* drad_w, float* drad_b, float* dh_appear, float* dx, float* dv, void* stream) {
  const int rc = field_meta_check_full(meta);
  if (rc) return rc;
  if (S <= 0) return 0;
  if (!x && !(rays_o && rays_d && t && ridx)) return 24;
  if (!(rays_d && ridx)) return 25;
  if (!drad_w || !drad_b) return 26;
  if (!(nablas_fwd && rgb_fwd && gn_out && drgb)) return 27;
  FieldArgs a = field_args(meta);
  a.wpack = (const char*)wpack;
  a.x = x; a.rays_o = rays_o; a.rays_d = rays_d; a.t = t; a.ridx = ridx;
  a.h_appear = h_appear;
  a.S = S;
  a.dnablas = dnablas; a.drgb = drgb;
  a.nablas_fwd = nablas_fwd; a.rgb_fwd = rgb_fwd; a.dnab_total = gn_out;
  a.drad_w = drad_w; a.drad_b = drad_b; a.dh_appear = dh_appear;
  a.dx = dx; a.dv = dv;
  a.has_rgb = 1;
  a.ablate = bwd_ablate();
  const int64_t tiles = (S + 31) / 32;
  // workgroup-joint weight gradients: weights + one staging area in LDS, two workgroups per CU
  const size_t row = meta->precision == 0 ? jstage_row_bytes<0>() : jstage_row_bytes<1>();
  const size_t shmem = weights_lds_bytes(meta, M_R1, 6) + (meta->precision == 0 ? 160 : 128) * row;      // (+ the early rin area)
  int64_t nb = (tiles + JOINT_WAVES - 1) / JOINT_WAVES;
  nb = nb > 512 ? 512 : (nb < 1 ? 1 : nb);          // two resident workgroups per CU
  const dim3 grid((unsigned)nb), block(64 * JOINT_WAVES);
  const SrcOff so = src_off(1);
  int R = 1;
  float* sc = nb >= grad_replicas_min_wg() ? grad_scratch(stream, so.n_rad_w + so.n_rad_b, R) : nullptr;
  if (sc) {
    a.drad_w = sc; a.drad_b = sc + so.n_rad_w;
    a.rep_mask = R - 1; a.rep_stride = so.n_rad_w + so.n_rad_b;
  }
  if (meta->precision == 0) hipLaunchKernelGGL((k_rad_bwd_j<0>), grid, block, shmem, (hipStream_t)stream, a);
  else hipLaunchKernelGGL((k_rad_bwd_j<1>), grid, block, shmem, (hipStream_t)stream, a);
  if (sc) grad_scratch_fold(sc, R, so.n_rad_w, so.n_rad_b, drad_w, drad_b, (hipStream_t)stream);
  NSIM_CHECK_LAUNCH();
  return 0;
}

}  // extern "C"

// nsim_field_bwd_sdf (pos == NULL, no embedded-position block) | nsim_wide_bwd_sdf (pos = the sample positions of a model with one)
struct BwdPos {
  const float *x, *rays_o, *rays_d, *t;
  const int64_t* ridx;
};
static int field_bwd_sdf(const NsimFieldMeta* meta, const void* wpack, const BwdPos* pos, const float* h_planes,
                         const void* J_planes, int64_t S, const float* dsdf, const float* gn, float* dh_planes, float* g_planes,
                         float* dsdf_w, float* dsdf_b, float* dx, int64_t plane_pitch, void* stream) {
  const int rc = field_meta_check_full(meta);
  if (rc) return rc;
  if (S <= 0) return 0;
  if (!dsdf_w || !dsdf_b) return 26;
  if (!h_planes || !J_planes) return 28;
  if ((dh_planes != nullptr) != (g_planes != nullptr)) return 28;
  const int ne = field_ne(meta);
  if ((ne != 0) != (pos != nullptr)) return 36;      // the block is regenerated from the positions: nsim_wide_bwd_sdf
  if (ne && dx) return 36;
  FieldArgs a = field_args(meta);
  a.wpack = (const char*)wpack;
  if (pos) {
    if (!pos->x && !(pos->rays_o && pos->rays_d && pos->t && pos->ridx)) return 24;
    a.x = pos->x; a.rays_o = pos->rays_o; a.rays_d = pos->rays_d; a.t = pos->t; a.ridx = pos->ridx;
  }
  a.S = S;
  if (plane_pitch != 0 && (plane_pitch < S || (plane_pitch & 31))) return 28;
  a.PS = plane_pitch ? plane_pitch : NSIM_PLANE_PITCH(S);
  a.dsdf = dsdf; a.dnablas = gn;
  a.h_pl = const_cast<float*>(h_planes); a.J_pl = const_cast<void*>(J_planes);
  a.dh_pl = dh_planes; a.g_pl = g_planes;
  a.dsdf_w = dsdf_w; a.dsdf_b = dsdf_b;
  a.dx = dx;
  a.ablate = bwd_ablate();
  const int nc = field_nc(meta->lotd.num_levels), nw = field_waves(meta, 2);
  // 16-level pyramids: the workgroup-joint kernel (k_field_bwd_j).  Measured on 278 k points (MI355X): k_field<0,2,2>
  // 0.232 ms -> 0.176 ms (weights in LDS instead of L2 reads -- the recomputed forward alone was 35 % of a tile --
  // and no accumulator read-modify-write); forced into 256 registers (two waves per SIMD) it spills 177 and takes 0.275 ms,
  // so the two-hidden-layer instantiation runs one wave per SIMD.  NSIM_SDF_BWD_OLD=1: the round-1 kernel (A/B aid).
  const char* oldp = getenv("NSIM_SDF_BWD_OLD");
  if (ne || !(oldp && atoi(oldp) == 1)) {
    // workgroup-joint weight gradients: weights + three staging areas in LDS, two workgroups per CU
    const size_t row = meta->precision == 0 ? jstage_row_bytes<0>() : jstage_row_bytes<1>();
    // + one 16 KB plane-prefetch buffer per wave in fp16 mode (k_field_bwd_j GLDS)
    const int ni = nc + ne;      // staging: A 64 rows, B 32 ni (>= 64) rows, C 64 rows
    const bool dbuf = NSIM_BWD_DBUF && meta->precision == 0 && !ne && (nc == 2 || NSIM_BWD_JDIRECT);      // (DB in the kernel)
    const size_t shmem = weights_lds_bytes(meta, 0, 4) + (dbuf ? 2 : 1) * (128 + (ni > 2 ? 32 * ni : 64)) * row +
                         (meta->precision == 0 && nc == 1 && !ne ? (size_t)JOINT_WAVES * (NSIM_BWD_JDIRECT ? 4096 : 16384) : 0);
    const int64_t tiles = (S + 31) / 32;
    int64_t nb = (tiles + JOINT_WAVES - 1) / JOINT_WAVES;
    const char* gcap = getenv("NSIM_SDF_BWD_GRID");
    const int64_t cap = gcap ? atoi(gcap) : 256;          // one resident workgroup per CU (register-limited)
    nb = nb > cap ? cap : (nb < 1 ? 1 : nb);
    const dim3 grid((unsigned)nb), block(64 * JOINT_WAVES);
    const SrcOff so = src_off(meta->sdf_D, 2 * meta->lotd.num_levels + meta->embed_E);
    int R = 1;
    float* sc = nb >= grad_replicas_min_wg() ? grad_scratch(stream, so.n_sdf_w + so.n_sdf_b, R) : nullptr;
    if (sc) {
      a.dsdf_w = sc; a.dsdf_b = sc + so.n_sdf_w;
      a.rep_mask = R - 1; a.rep_stride = so.n_sdf_w + so.n_sdf_b;
    }
    switch ((ne ? 8 : 0) + (nc - 1) * 4 + meta->precision * 2 + (meta->sdf_D - 1)) {
      case 8: hipLaunchKernelGGL((k_field_bwd_j<0, 1, 1, 2>), grid, block, shmem, (hipStream_t)stream, a); break;
      case 9: hipLaunchKernelGGL((k_field_bwd_j<0, 2, 1, 2>), grid, block, shmem, (hipStream_t)stream, a); break;
      case 10: hipLaunchKernelGGL((k_field_bwd_j<1, 1, 1, 2>), grid, block, shmem, (hipStream_t)stream, a); break;
      case 11: hipLaunchKernelGGL((k_field_bwd_j<1, 2, 1, 2>), grid, block, shmem, (hipStream_t)stream, a); break;
      case 12: hipLaunchKernelGGL((k_field_bwd_j<0, 1, 2, 2>), grid, block, shmem, (hipStream_t)stream, a); break;
      case 13: hipLaunchKernelGGL((k_field_bwd_j<0, 2, 2, 2>), grid, block, shmem, (hipStream_t)stream, a); break;
      case 14: hipLaunchKernelGGL((k_field_bwd_j<1, 1, 2, 2>), grid, block, shmem, (hipStream_t)stream, a); break;
      case 15: hipLaunchKernelGGL((k_field_bwd_j<1, 2, 2, 2>), grid, block, shmem, (hipStream_t)stream, a); break;
      case 0: hipLaunchKernelGGL((k_field_bwd_j<0, 1>), grid, block, shmem, (hipStream_t)stream, a); break;
      case 1: hipLaunchKernelGGL((k_field_bwd_j<0, 2>), grid, block, shmem, (hipStream_t)stream, a); break;
      case 2: hipLaunchKernelGGL((k_field_bwd_j<1, 1>), grid, block, shmem, (hipStream_t)stream, a); break;
      case 3: hipLaunchKernelGGL((k_field_bwd_j<1, 2>), grid, block, shmem, (hipStream_t)stream, a); break;
      case 4: hipLaunchKernelGGL((k_field_bwd_j<0, 1, 2>), grid, block, shmem, (hipStream_t)stream, a); break;
      case 5: hipLaunchKernelGGL((k_field_bwd_j<0, 2, 2>), grid, block, shmem, (hipStream_t)stream, a); break;
      case 6: hipLaunchKernelGGL((k_field_bwd_j<1, 1, 2>), grid, block, shmem, (hipStream_t)stream, a); break;
      case 7: hipLaunchKernelGGL((k_field_bwd_j<1, 2, 2>), grid, block, shmem, (hipStream_t)stream, a); break;
    }
    if (sc) grad_scratch_fold(sc, R, so.n_sdf_w, so.n_sdf_b, dsdf_w, dsdf_b, (hipStream_t)stream);
    NSIM_CHECK_LAUNCH();
    return 0;
  }
  // NSIM_SDF_BWD_OLD=1 (A/B aid): the round-1 kernel; more than 16 levels: fp16: one private accumulator copy per wave +
  // staging, weights from L2; f32: staged weights + one shared accumulator
  const size_t acc_bytes = ((6400 + 2048 * (nc - 1)) * 4 + 15) & ~15;
  const size_t shmem = meta->precision == 0 ? weights_lds_bytes(meta, 0, 0) + nw * acc_bytes + nw * stage_bytes(meta)
                                            : weights_lds_bytes(meta, 0, 4) + acc_bytes + FIELD_WAVES * stage_bytes(meta);
  return field_launch<2>(meta, a, shmem, FIELD_GRID_BWD, (hipStream_t)stream);
}

extern "C" {

int nsim_field_bwd_sdf(const NsimFieldMeta* meta, const void* wpack, const float* h_planes, const void* J_planes,
                       int64_t S, const float* dsdf, const float* gn, float* dh_planes, float* g_planes, float* dsdf_w,
                       float* dsdf_b, float* dx, int64_t plane_pitch, void* stream) {
  return field_bwd_sdf(meta, wpack, nullptr, h_planes, J_planes, S, dsdf, gn, dh_planes, g_planes, dsdf_w, dsdf_b, dx, plane_pitch,
                       stream);
}

int nsim_wide_bwd_sdf(const NsimFieldMeta* meta, const void* wpack, const float* x, const float* rays_o, const float* rays_d,
                      const float* t, const int64_t* ridx, int64_t S, const float* h_planes, const void* J_planes,
                      int64_t plane_pitch, const float* dsdf, const float* gn, float* dh_planes, float* g_planes, float* dsdf_w,
                      float* dsdf_b, void* stream) {
  const BwdPos pos = {x, rays_o, rays_d, t, ridx};
  return field_bwd_sdf(meta, wpack, &pos, h_planes, J_planes, S, dsdf, gn, dh_planes, g_planes, dsdf_w, dsdf_b, nullptr, plane_pitch,
                       stream);
}

int nsim_lotd_scatter(const NsimLotdMeta* meta, const float* x, const float* rays_o, const float* rays_d, const float* t,
                      const int64_t* ridx, const int64_t* ray_goff, int64_t S, const float* dh_planes,
                      const float* g_planes, const float* gn, float* dgrid, int level_begin, int level_count,
                      void* stream) {
  const int rc = lotd_meta_check(meta);
  if (rc) return rc;
  if (S <= 0) return 0;
  if (level_count <= 0) {       // all levels
    level_begin = 0;
    level_count = meta->num_levels;
  }
  if (level_begin < 0 || level_begin + level_count > meta->num_levels) return 12;
  if (!x && !(rays_o && rays_d && t && ridx)) return 24;
  if (!dh_planes || !g_planes || !dgrid) return 28;
  if (ray_goff && !ridx) return 29;
  if (bwd_ablate() & 1) return 0;
  ScatterArgs sa;
  sa.lotd = lotd_dev(meta);
  sa.x = x; sa.rays_o = rays_o; sa.rays_d = rays_d; sa.t = t; sa.ridx = ridx;
  sa.S = S;
  sa.dh_pl = dh_planes; sa.g_pl = g_planes; sa.gn = gn;
  sa.ray_goff = ray_goff;
  sa.dgrid = dgrid;
  sa.dedup_max_res = 1 << 30;   // every level (compressed query mode keeps neighbouring fine samples: 0.345 -> 0.319 ms; it was 600 for the un-compressed mode)
  const char* e = getenv("NSIM_DEDUP_MAX_RES");
  if (e) sa.dedup_max_res = atoi(e);
  const char* ep = getenv("NSIM_SCATTER_PARITY");      // 0: slots by corner offset, as rounds 1-5 (A/B aid)
  sa.parity_slots = !(ep && atoi(ep) == 0);
  const int64_t chunks = (S + 63) / 64;
  sa.level_begin = level_begin;
  const dim3 grid(nsim_blocks(chunks, 4, 4096), level_count);
  const char* eg = getenv("NSIM_SCATTER_GROUP");       // (read per launch: A/B runs flip it inside one process)
  const bool consec = !(eg && atoi(eg) == 0);
  if (consec)
    hipLaunchKernelGGL(k_lotd_scatter<true>, grid, dim3(256), 0, (hipStream_t)stream, sa);
  else
    hipLaunchKernelGGL(k_lotd_scatter<false>, grid, dim3(256), 0, (hipStream_t)stream, sa);
  NSIM_CHECK_LAUNCH();
  return 0;
}

int nsim_lotd_hess_dx(const NsimLotdMeta* meta, const void* grid_f16, const float* x, const float* rays_o,
                      const float* rays_d, const float* t, const int64_t* ridx, const int64_t* ray_goff, int64_t S,
                      const float* g_planes, const float* gn, float* dx, void* stream) {
  const int rc = lotd_meta_check(meta);
  if (rc) return rc;
  if (S <= 0) return 0;
  if (!x && !(rays_o && rays_d && t && ridx)) return 24;
  if (ray_goff && !ridx) return 29;
  if (!grid_f16 || !g_planes || !gn || !dx) return 4;
  HessArgs ha;
  ha.lotd = lotd_dev(meta);
  ha.grid = (const f16*)grid_f16;
  ha.x = x; ha.rays_o = rays_o; ha.rays_d = rays_d; ha.t = t; ha.ridx = ridx;
  ha.ray_goff = ray_goff;
  ha.S = S;
  ha.g_pl = g_planes; ha.gn = gn; ha.dx = dx;
  hipLaunchKernelGGL(k_lotd_hess_dx, dim3(nsim_blocks(S, 256)), dim3(256), 0, (hipStream_t)stream, ha);
  NSIM_CHECK_LAUNCH();
  return 0;
}

int nsim_ray_grad_reduce(const float* dx, const float* dv, const float* t, const int64_t* ridx, int64_t S,
                         float* d_rays_o, float* d_rays_d, void* stream) {
  if (S < 0) return 2;
  if (S == 0) return 0;
  if (!dx || !t || !ridx) return 4;
  hipLaunchKernelGGL(k_ray_grad_reduce, dim3(nsim_blocks(S, 256)), dim3(256), 0, (hipStream_t)stream, dx, dv, t, ridx, S,
                     d_rays_o, d_rays_d);
  NSIM_CHECK_LAUNCH();
  return 0;
}

int nsim_selftest_mfma(const float* a, const float* b, float* d, int use_f32, void* stream) {
  hipLaunchKernelGGL(k_selftest_mfma, dim3(1), dim3(64), 0, (hipStream_t)stream, a, b, d, use_f32);
  NSIM_CHECK_LAUNCH();
  return 0;
}

}  // extern "C"
