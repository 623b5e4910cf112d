// decoder_chain_bench.hip -- what CAN the 32 -> 64 -> 64 -> 1 SDF decoder chains reach on gfx950?  (round 5)
//
// The product's decoder kernels (csrc/field.hip: k_field_sdf, k_field MODE 3, k_field_bwd_j) report 6.5-10 % MFMA busy against
// the 2.5 PFLOP/s dense f16 peak.  That peak is not their ceiling: a 64-wide layer with a softplus(beta) activation spends two
// quarter-rate transcendentals and ~5 full-rate VALU operations per accumulator element next to 1/8 of an MFMA.  This
// benchmark measures the ceiling of the CHAIN ITSELF: the same sequence of dense products (v_mfma_f32_32x32x16_f16, or
// v_mfma_f32_16x16x32_f16 tiles) and activations on REGISTER-RESIDENT tiles -- inputs generated in registers, weights as MFMA
// A-fragments in LDS (staged once per workgroup), no feature planes, no LDS staging of activations, no weight-gradient
// products, no global traffic but one word per wave at the end -- at 1 / 2 / 4 waves per SIMD (__launch_bounds__ second
// argument = minimum waves per execution unit; the register budget follows: 512 / 256 / 128 per lane).
//
//   chain 0  forward (sampling / evaluation query): a1 = sp(W1 h), a2 = sp(W2 a1), sdf = wh . a2
//            12 MFMA 32x32x16 + 64 softplus per 32 points                                   (12 416 FLOP per point)
//   chain 1  forward with first-order normals' seed: chain 0 + d2 = sig(a2) wh, d1 = sig(a1) (W2^T d2), g = W1^T d1
//            24 MFMA + 64 softplus + 64 sigmoid                                                (the with-grad forward decoder)
//   chain 2  second-order backward of the SDF branch, the product sequence of k_field_bwd_j<0, 2, 1> without its
//            weight-gradient tiles: W1, W2, W2^T, W1^T (recomputed forward + g), W1 gh, W2 eh1, W2^T dz2, W1^T dz1
//            48 MFMA + 64 softplus + 192 sigmoid-from-softplus per 32 points
//   chain 3  chain 2 + the 24 weight-gradient MFMAs of the product (bf16 operands taken from registers: the MFMA work of the
//            joint dW tiles without their LDS staging and barriers)
//
// Output: per (chain, tile shape, waves / SIMD): VGPR / AGPR / scratch of the instantiation (from the code object), points per
// second chip-wide, ns per 32-point tile and wave, achieved MFMA TFLOP/s and its fraction of 2 500.
// Build: hipcc --offload-arch=gfx950 -O3 tools/decoder_chain_bench.hip -o tools/decoder_chain_bench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef _Float16 f16;
typedef f16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define LOG2E 1.4426950408889634f
#define LN2 0.6931471805599453f
__device__ __forceinline__ float sp(float z, float beta, float inv_beta) {      // softplus on the raw base-2 pipes (field.hip)
  const float t = __builtin_amdgcn_exp2f(-fabsf(z) * (beta * LOG2E));
  return fmaxf(z, 0.f) + __builtin_amdgcn_logf(1.0f + t) * (inv_beta * LN2);
}
__device__ __forceinline__ float sg(float a, float beta) { return 1.0f - __builtin_amdgcn_exp2f(-a * (beta * LOG2E)); }

// weight fragments in LDS: matrix m, MFMA index i (A operand of the i-th product of that matrix), lane
//   32x32x16: W1 [64x32] = 2 tiles x 2 K-steps = 4, W2 / W2T [64x64] = 2 x 4 = 8, W1T [32x64] = 1 x 4 = 4     -> 24 fragments
//   16x16x32: W1 = 4 tiles x 1 = 4, W2 / W2T = 4 x 2 = 8, W1T = 2 x 2 = 4                                      -> 24 fragments
#define NFRAG 24
#define OFF_W1 0
#define OFF_W2 4
#define OFF_W2T 12
#define OFF_W1T 20

__device__ __forceinline__ f16x8 wfrag(const f16x8* W, int i) { return W[i * 64 + (threadIdx.x & 63)]; }

// ------------------------------------------------------------------------------------------- 32-point tiles (32x32x16)
// dense: out [NO x 16 accumulators] = W (NO tiles of 32 units, K = 16 * NK) . in (NK B-fragments)
template <int NO, int NK>
__device__ __forceinline__ void dense32(float (&out)[16 * NO], const f16x8* W, int off, const float (&in)[8 * NK], float scale) {
  f16x8 b[NK];
#pragma unroll
  for (int k = 0; k < NK; ++k)
#pragma unroll
    for (int e = 0; e < 8; ++e) b[k][e] = (f16)(in[8 * k + e] * scale);
#pragma unroll
  for (int o = 0; o < NO; ++o) {
    f32x16 acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int k = 0; k < NK; ++k) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wfrag(W, off + o * NK + k), b[k], acc, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 16; ++r) out[16 * o + r] = acc[r];
  }
}

template <int CHAIN>
__device__ __forceinline__ float tile32(const f16x8* W, float seed, float beta, float inv_beta, f32x16& dwa, f32x16& dwb) {
  const int lane = threadIdx.x & 63;
  float h[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) h[r] = __builtin_sinf(seed + 0.37f * (float)r + 0.011f * (float)lane) * 0.05f;      // v_sin: 1 transcendental
  float a1[32], a2[32];
  dense32<2, 2>(a1, W, OFF_W1, h, 1.0f);
#pragma unroll
  for (int k = 0; k < 32; ++k) a1[k] = sp(a1[k] + 0.01f, beta, inv_beta);
  dense32<2, 4>(a2, W, OFF_W2, a1, 1.0f);
#pragma unroll
  for (int k = 0; k < 32; ++k) a2[k] = sp(a2[k] - 0.01f, beta, inv_beta);
  float out = 0.f;
#pragma unroll
  for (int k = 0; k < 32; ++k) out += a2[k] * (0.03f + 0.001f * (float)k);
  if constexpr (CHAIN == 0) return out;
  float d1[32];
  {
    float d2[32];
#pragma unroll
    for (int k = 0; k < 32; ++k) d2[k] = sg(a2[k], beta) * (0.03f + 0.001f * (float)k);
    dense32<2, 4>(d1, W, OFF_W2T, d2, 1.0f);
  }
#pragma unroll
  for (int k = 0; k < 32; ++k) d1[k] = sg(a1[k], beta) * d1[k];
  float g[16];
  dense32<1, 4>(g, W, OFF_W1T, d1, 1.0f);
#pragma unroll
  for (int r = 0; r < 16; ++r) out += g[r];
  if constexpr (CHAIN == 1) return out;
  // ---- second order: gh = J . gn (3 FMAs per feature), then the product sequence of k_field_bwd_j
  float gh[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) gh[r] = h[r] * 0.3f + g[r] * 0.2f + seed * 0.1f;
  float dh1[32];
  dense32<2, 2>(dh1, W, OFF_W1, gh, 1.0f);
  float dz1[32], eh1[32];
#pragma unroll
  for (int k = 0; k < 32; ++k) {
    const float s1 = sg(a1[k], beta);
    dz1[k] = dh1[k] * d1[k] * (beta * (1.0f - s1));
    eh1[k] = dh1[k] * s1;
  }
  if constexpr (CHAIN == 3) {      // dW2 += d2 (x) eh1 : one 32x32 tile over K = 128 points per wave in the product = 8 MFMA
    bf16x8 x, y;
#pragma unroll
    for (int e = 0; e < 8; ++e) { x[e] = (__bf16)eh1[e]; y[e] = (__bf16)d1[e]; }
#pragma unroll
    for (int q = 0; q < 8; ++q) dwa = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, dwa, 0, 0, 0);
  }
  float dh2[32];
  dense32<2, 4>(dh2, W, OFF_W2, eh1, 1.0f);
  float dz2[32];
#pragma unroll
  for (int k = 0; k < 32; ++k) {
    const float s2 = sg(a2[k], beta);
    const float wh = 0.03f + 0.001f * (float)k;
    out += dh2[k] * s2 + seed * a2[k];
    dz2[k] = seed * wh * s2 + dh2[k] * wh * (beta * s2 * (1.0f - s2));
  }
  if constexpr (CHAIN == 3) {      // dW2 += dz2 (x) a1 (8 MFMA) and dW1 += d1 (x) gh (4 MFMA)
    bf16x8 x, y;
#pragma unroll
    for (int e = 0; e < 8; ++e) { x[e] = (__bf16)dz2[e]; y[e] = (__bf16)a1[e]; }
#pragma unroll
    for (int q = 0; q < 8; ++q) dwa = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, dwa, 0, 0, 0);
#pragma unroll
    for (int q = 0; q < 4; ++q) dwb = __builtin_amdgcn_mfma_f32_32x32x16_bf16(y, x, dwb, 0, 0, 0);
  }
  float da1[32];
  dense32<2, 4>(da1, W, OFF_W2T, dz2, 1.0f);
#pragma unroll
  for (int k = 0; k < 32; ++k) dz1[k] = dz1[k] + da1[k] * sg(a1[k], beta);
  if constexpr (CHAIN == 3) {      // dW1 += dz1 (x) h (4 MFMA)
    bf16x8 x, y;
#pragma unroll
    for (int e = 0; e < 8; ++e) { x[e] = (__bf16)dz1[e]; y[e] = (__bf16)h[e]; }
#pragma unroll
    for (int q = 0; q < 4; ++q) dwb = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, dwb, 0, 0, 0);
  }
  float dh[16];
  dense32<1, 4>(dh, W, OFF_W1T, dz1, 1.0f);
#pragma unroll
  for (int r = 0; r < 16; ++r) out += dh[r];
  return out;
}

// ------------------------------------------------------------------------------------------- 16-point tiles (16x16x32)
// B fragment of 16x16x32: lane (n = lane & 15, kg = lane >> 4) holds k = 8 kg + e, e < 8 -> K = 32 per MFMA.
// accumulators: 4 per 16-unit tile and lane.  dense16: out [4 NO] = W (NO tiles of 16 units, K = 32 NK) . in [8 NK]
template <int NO, int NK>
__device__ __forceinline__ void dense16(float (&out)[4 * NO], const f16x8* W, int off, const float (&in)[8 * NK]) {
  f16x8 b[NK];
#pragma unroll
  for (int k = 0; k < NK; ++k)
#pragma unroll
    for (int e = 0; e < 8; ++e) b[k][e] = (f16)in[8 * k + e];
#pragma unroll
  for (int o = 0; o < NO; ++o) {
    f32x4 acc = {0, 0, 0, 0};
#pragma unroll
    for (int k = 0; k < NK; ++k) acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(wfrag(W, (off + o * NK + k) % NFRAG), b[k], acc, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 4; ++r) out[4 * o + r] = acc[r];
  }
}

// one 16-point tile: features 32 -> 8 per lane (1 K-step), hidden 64 -> 16 per lane (4 tiles x 4 = the 2 K-steps of the next layer)
template <int CHAIN>
__device__ __forceinline__ float tile16(const f16x8* W, float seed, float beta, float inv_beta) {
  const int lane = threadIdx.x & 63;
  float h[8];
#pragma unroll
  for (int r = 0; r < 8; ++r) h[r] = __builtin_sinf(seed + 0.37f * (float)r + 0.011f * (float)lane) * 0.05f;
  float a1[16], a2[16];
  dense16<4, 1>(a1, W, 0, h);
#pragma unroll
  for (int k = 0; k < 16; ++k) a1[k] = sp(a1[k] + 0.01f, beta, inv_beta);
  dense16<4, 2>(a2, W, 4, a1);
#pragma unroll
  for (int k = 0; k < 16; ++k) a2[k] = sp(a2[k] - 0.01f, beta, inv_beta);
  float out = 0.f;
#pragma unroll
  for (int k = 0; k < 16; ++k) out += a2[k] * (0.03f + 0.001f * (float)k);
  if constexpr (CHAIN == 0) return out;
  float d1[16];
  {
    float d2[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) d2[k] = sg(a2[k], beta) * (0.03f + 0.001f * (float)k);
    dense16<4, 2>(d1, W, 12, d2);
  }
#pragma unroll
  for (int k = 0; k < 16; ++k) d1[k] = sg(a1[k], beta) * d1[k];
  float g[8];
  dense16<2, 2>(g, W, 20, d1);
#pragma unroll
  for (int r = 0; r < 8; ++r) out += g[r];
  if constexpr (CHAIN == 1) return out;
  float gh[8];
#pragma unroll
  for (int r = 0; r < 8; ++r) gh[r] = h[r] * 0.3f + g[r] * 0.2f + seed * 0.1f;
  float dh1[16];
  dense16<4, 1>(dh1, W, 0, gh);
  float dz1[16], eh1[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    const float s1 = sg(a1[k], beta);
    dz1[k] = dh1[k] * d1[k] * (beta * (1.0f - s1));
    eh1[k] = dh1[k] * s1;
  }
  float dh2[16];
  dense16<4, 2>(dh2, W, 4, eh1);
  float dz2[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    const float s2 = sg(a2[k], beta);
    const float wh = 0.03f + 0.001f * (float)k;
    out += dh2[k] * s2 + seed * a2[k];
    dz2[k] = seed * wh * s2 + dh2[k] * wh * (beta * s2 * (1.0f - s2));
  }
  float da1[16];
  dense16<4, 2>(da1, W, 12, dz2);
#pragma unroll
  for (int k = 0; k < 16; ++k) dz1[k] = dz1[k] + da1[k] * sg(a1[k], beta);
  float dh[8];
  dense16<2, 2>(dh, W, 20, dz1);
#pragma unroll
  for (int r = 0; r < 8; ++r) out += dh[r];
  return out;
}

// ------------------------------------------------------------------------------------------- kernels
template <int CHAIN, int TILE, int WPS>
__global__ void __launch_bounds__(256, WPS) k_chain(const f16* __restrict__ wsrc, int tiles_per_wave, float beta, float* __restrict__ out) {
  __shared__ f16x8 W[NFRAG * 64];      // 24 KB
  for (int i = threadIdx.x; i < NFRAG * 64; i += 256) {
    f16x8 v;
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = wsrc[(i * 8 + e) % 4096];
    W[i] = v;
  }
  __syncthreads();
  const float inv_beta = 1.0f / beta;
  float acc = 0.f;
  f32x16 dwa = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, dwb = dwa;
  float seed = 0.001f * (float)(blockIdx.x * 4 + (threadIdx.x >> 6));
  for (int t = 0; t < tiles_per_wave; ++t) {
    if constexpr (TILE == 32) {
      acc += tile32<CHAIN>(W, seed, beta, inv_beta, dwa, dwb);
    } else {      // two 16-point tiles = the same 32 points
      acc += tile16<CHAIN>(W, seed, beta, inv_beta);
      acc += tile16<CHAIN>(W, seed + 0.5f, beta, inv_beta);
    }
    seed += 0.013f;
  }
  if constexpr (CHAIN == 3) {
#pragma unroll
    for (int r = 0; r < 16; ++r) acc += dwa[r] + dwb[r];
  }
  if (acc == 123.456f) out[blockIdx.x * 256 + threadIdx.x] = acc;
}

struct Res { int vgpr, agpr, scratch; };
template <int CHAIN, int TILE, int WPS>
static void run(const f16* w, float* out, int cus, const char* name) {
  hipFuncAttributes fa;
  hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(k_chain<CHAIN, TILE, WPS>));
  const int tiles = 256;
  const int blocks = cus * WPS;
  hipLaunchKernelGGL((k_chain<CHAIN, TILE, WPS>), dim3(blocks), dim3(256), 0, 0, w, tiles, 100.0f, out);
  hipDeviceSynchronize();
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int nrep = 5;
  hipEventRecord(e0);
  for (int r = 0; r < nrep; ++r) hipLaunchKernelGGL((k_chain<CHAIN, TILE, WPS>), dim3(blocks), dim3(256), 0, 0, w, tiles, 100.0f, out);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  ms /= nrep;
  static const int mfma32[4] = {12, 24, 48, 72};      // 32x32x16-equivalent MFMAs per 32-point tile
  const double tiles_total = (double)blocks * 4 * tiles;
  const double pts_s = tiles_total * 32 / (ms * 1e-3);
  const double tflops = tiles_total * mfma32[CHAIN] * 32768.0 / (ms * 1e-3) * 1e-12;
  const double ns_tile = ms * 1e6 / tiles;             // per 32-point tile and wave
  printf("%-28s tile %2d  waves/SIMD %d  regs %3d  scratch %4d B  %8.2f G pts/s  %8.1f ns/tile/wave  %7.1f TFLOP/s  %5.1f %% of 2500\n",
         name, TILE, WPS, (int)fa.numRegs, (int)fa.localSizeBytes, pts_s * 1e-9, ns_tile, tflops, 100.0 * tflops / 2500.0);
}

int main(int argc, char** argv) {
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  const int cus = p.multiProcessorCount;
  printf("device %s, %d CUs, %d MHz\n", p.name, cus, p.clockRate / 1000);
  f16* w;
  float* out;
  hipMalloc(&w, 4096 * sizeof(f16));
  hipMalloc(&out, sizeof(float) * 256 * cus * 8);
  f16 hw[4096];
  for (int i = 0; i < 4096; ++i) hw[i] = (f16)(0.02f * (float)((i * 37) % 19 - 9));
  hipMemcpy(w, hw, sizeof(hw), hipMemcpyHostToDevice);
#define RUN3(C, T, NAME) run<C, T, 1>(w, out, cus, NAME); run<C, T, 2>(w, out, cus, NAME); run<C, T, 4>(w, out, cus, NAME);
  RUN3(0, 32, "forward (sdf query)")
  RUN3(0, 16, "forward (sdf query)")
  RUN3(1, 32, "forward + d sdf / d h")
  RUN3(1, 16, "forward + d sdf / d h")
  RUN3(2, 32, "2nd-order backward chain")
  RUN3(2, 16, "2nd-order backward chain")
  RUN3(3, 32, "2nd-order chain + dW MFMAs")
  return 0;
}
