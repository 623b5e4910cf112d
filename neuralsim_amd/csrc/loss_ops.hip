// loss_ops.hip -- fused reductions of the losses the training step evaluates on the path's outputs (SURVEY sec. 8 row
// a18) and the gradient of the per-frame appearance-embedding lookup.
//   * eikonal: mean((|nablas| - 1)^2)                 (app/loss/eikonal.py:96-105 with safe_mse off, alpha_reg_zero 0)
//   * photometric mse: mean((pred - gt)^2)            (app/loss/photometric.py:88-146, fn_type mse, no mask)
//   * rows_scatter_add: d embed[idx[i], :] += g[i, :]  (app/models/scene/image_embeddings.py:23-80: ``embed[fidx]``)
// The reference evaluates these with a handful of torch ops each; here one launch per direction, because at 8192
// rays per iteration the step is bounded by launch count, not by bytes.
#include "nsim_common.h"

#define LOSS_BLOCK 256
#define LOSS_MAX_BLOCKS 256

__device__ __forceinline__ void block_sum_atomic(float v, float scale, float* out) {
  __shared__ float red[LOSS_BLOCK / 64];
  v = wave_sum(v);
  if (nsim_lane() == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0) {
    float tot = 0.f;
    for (int w = 0; w < LOSS_BLOCK / 64; ++w) tot += red[w];
    if (tot != 0.f) atomicAdd(out, tot * scale);
  }
}

__global__ void __launch_bounds__(LOSS_BLOCK) k_eikonal_fwd(const float* __restrict__ nab, int64_t S, float inv_S,
                                                            float* __restrict__ out) {
  float acc = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * LOSS_BLOCK + threadIdx.x; i < S; i += (int64_t)gridDim.x * LOSS_BLOCK) {
    const float a = nab[3 * i], b = nab[3 * i + 1], c = nab[3 * i + 2];
    const float e = sqrtf(a * a + b * b + c * c) - 1.0f;
    acc += e * e;
  }
  block_sum_atomic(acc, inv_S, out);
}

__global__ void __launch_bounds__(LOSS_BLOCK) k_eikonal_bwd(const float* __restrict__ nab, int64_t S, float inv_S,
                                                            const float* __restrict__ gout, float* __restrict__ dnab) {
  const int64_t i = (int64_t)blockIdx.x * LOSS_BLOCK + threadIdx.x;
  if (i >= S) return;
  const float a = nab[3 * i], b = nab[3 * i + 1], c = nab[3 * i + 2];
  const float nrm = sqrtf(a * a + b * b + c * c);
  // d/dn (|n|-1)^2 = 2 (|n|-1) n/|n|; the sub-gradient at n = 0 is 0 (as torch.norm's backward)
  const float k = nrm > 0.f ? gout[0] * inv_S * 2.0f * (nrm - 1.0f) / nrm : 0.f;
  dnab[3 * i] = k * a;
  dnab[3 * i + 1] = k * b;
  dnab[3 * i + 2] = k * c;
}

__global__ void __launch_bounds__(LOSS_BLOCK) k_mse_fwd(const float* __restrict__ a, const float* __restrict__ b,
                                                        int64_t n, float inv_n, float* __restrict__ out) {
  float acc = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * LOSS_BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * LOSS_BLOCK) {
    const float e = a[i] - b[i];
    acc += e * e;
  }
  block_sum_atomic(acc, inv_n, out);
}

__global__ void __launch_bounds__(LOSS_BLOCK) k_mse_bwd(const float* __restrict__ a, const float* __restrict__ b,
                                                        int64_t n, float inv_n, const float* __restrict__ gout,
                                                        float* __restrict__ da) {
  const int64_t i = (int64_t)blockIdx.x * LOSS_BLOCK + threadIdx.x;
  if (i >= n) return;
  da[i] = gout[0] * inv_n * 2.0f * (a[i] - b[i]);
}

// The training step's loss head in ONE launch (six otherwise): acc[0] += mse(pred, gt), acc[1] += eikonal(nab[:S]),
// acc[2] += eikonal(nab[S:S+M]) and the gradients of  mse + w_eik (eik + eik)  -- d_img = 2 (pred - gt) / n and
// dnab = w_eik 2 (|n| - 1) n / (|n| count).  None of the gradients needs the loss VALUE, so there is no second pass.
__global__ void __launch_bounds__(LOSS_BLOCK) k_loss_head(const float* __restrict__ pred, const float* __restrict__ gt,
                                                          int64_t n_img, const float* __restrict__ nab, int64_t S,
                                                          int64_t M, float w_eik, float* __restrict__ acc,
                                                          float* __restrict__ d_img, float* __restrict__ dnab) {
  __shared__ float red[3][LOSS_BLOCK / 64];
  const float inv_n = 1.0f / (float)n_img, inv_S = 1.0f / (float)(S > 0 ? S : 1), inv_M = 1.0f / (float)(M > 0 ? M : 1);
  const int64_t St = S + M, top = n_img > St ? n_img : St;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * LOSS_BLOCK + threadIdx.x; i < top; i += (int64_t)gridDim.x * LOSS_BLOCK) {
    if (i < n_img) {
      const float e = pred[i] - gt[i];
      a0 += e * e;
      d_img[i] = inv_n * 2.0f * e;
    }
    if (i < St) {
      const float a = nab[3 * i], b = nab[3 * i + 1], c = nab[3 * i + 2];
      const float nrm = sqrtf(a * a + b * b + c * c);
      const float e = nrm - 1.0f;
      const bool ren = i < S;
      if (ren) a1 += e * e;
      else a2 += e * e;
      const float k = nrm > 0.f ? w_eik * (ren ? inv_S : inv_M) * 2.0f * (nrm - 1.0f) / nrm : 0.f;
      dnab[3 * i] = k * a;
      dnab[3 * i + 1] = k * b;
      dnab[3 * i + 2] = k * c;
    }
  }
  a0 = wave_sum(a0);
  a1 = wave_sum(a1);
  a2 = wave_sum(a2);
  if (nsim_lane() == 0) {
    red[0][threadIdx.x >> 6] = a0;
    red[1][threadIdx.x >> 6] = a1;
    red[2][threadIdx.x >> 6] = a2;
  }
  __syncthreads();
  if (threadIdx.x < 3) {
    float tot = 0.f;
    for (int w = 0; w < LOSS_BLOCK / 64; ++w) tot += red[threadIdx.x][w];
    const float sc = threadIdx.x == 0 ? inv_n : (threadIdx.x == 1 ? inv_S : inv_M);
    if (tot != 0.f) atomicAdd(acc + threadIdx.x, tot * sc);
  }
}

// rows * C <= ROWS_LDS_MAX: per-block LDS histogram (the few hundred frame embeddings are hit by thousands of rays),
// flushed with one global atomic per touched entry; larger tables go straight to global atomics.
#define ROWS_LDS_MAX 4096
__global__ void __launch_bounds__(LOSS_BLOCK) k_rows_scatter_add(const float* __restrict__ g,
                                                                 const int64_t* __restrict__ idx, int64_t n, int C,
                                                                 int64_t rows, float* __restrict__ out) {
  __shared__ float acc[ROWS_LDS_MAX];
  const int64_t tot = rows * C;
  const bool use_lds = tot <= ROWS_LDS_MAX;
  if (use_lds) {
    for (int j = threadIdx.x; j < tot; j += LOSS_BLOCK) acc[j] = 0.f;
    __syncthreads();
  }
  const int64_t nC = n * C;
  for (int64_t i = (int64_t)blockIdx.x * LOSS_BLOCK + threadIdx.x; i < nC; i += (int64_t)gridDim.x * LOSS_BLOCK) {
    const int64_t r = idx[i / C];
    if (r < 0 || r >= rows) continue;
    const int64_t j = r * C + (i % C);
    if (use_lds) atomicAdd(&acc[j], g[i]);
    else atomicAdd(&out[j], g[i]);
  }
  if (use_lds) {
    __syncthreads();
    for (int j = threadIdx.x; j < tot; j += LOSS_BLOCK) {
      const float v = acc[j];
      if (v != 0.f) atomicAdd(&out[j], v);
    }
  }
}

// analytic image of a sphere at the origin along unit rays: 0.5 + 0.5 n at the first hit, black elsewhere
// (synthetic multi-view-consistent supervision of bench.py; one launch instead of ~10 elementwise torch ops)
__global__ void __launch_bounds__(LOSS_BLOCK) k_sphere_image(const float* __restrict__ o, const float* __restrict__ d,
                                                             int64_t N, float radius, float* __restrict__ rgb) {
  const int64_t i = (int64_t)blockIdx.x * LOSS_BLOCK + threadIdx.x;
  if (i >= N) return;
  const float ox = o[3 * i], oy = o[3 * i + 1], oz = o[3 * i + 2];
  const float dx = d[3 * i], dy = d[3 * i + 1], dz = d[3 * i + 2];
  const float b = ox * dx + oy * dy + oz * dz;
  const float c = ox * ox + oy * oy + oz * oz - radius * radius;
  const float disc = b * b - c;
  const float t = -b - sqrtf(fmaxf(disc, 0.f));
  const bool hit = disc > 0.f && t > 0.f;
  const float inv = 1.0f / radius;
  rgb[3 * i] = hit ? 0.5f + 0.5f * (ox + t * dx) * inv : 0.f;
  rgb[3 * i + 1] = hit ? 0.5f + 0.5f * (oy + t * dy) * inv : 0.f;
  rgb[3 * i + 2] = hit ? 0.5f + 0.5f * (oz + t * dz) * inv : 0.f;
}

// compaction of the AABB-tested rays: (o, d, near, far)[idx] in one launch (model.ray_test)
__global__ void __launch_bounds__(LOSS_BLOCK) k_gather_rays(const float* __restrict__ o, const float* __restrict__ d,
                                                            const float* __restrict__ near, const float* __restrict__ far,
                                                            const int64_t* __restrict__ idx, int64_t R,
                                                            float* __restrict__ o_out, float* __restrict__ d_out,
                                                            float* __restrict__ near_out, float* __restrict__ far_out) {
  const int64_t i = (int64_t)blockIdx.x * LOSS_BLOCK + threadIdx.x;
  if (i >= R) return;
  const int64_t r = idx[i];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    o_out[3 * i + c] = o[3 * r + c];
    d_out[3 * i + c] = d[3 * r + c];
  }
  near_out[i] = near[r];
  far_out[i] = far[r];
}

static inline dim3 loss_grid(int64_t n) {
  int64_t b = nsim_blocks(n, LOSS_BLOCK);
  return dim3((unsigned)(b > LOSS_MAX_BLOCKS ? LOSS_MAX_BLOCKS : b));
}

extern "C" {

// out[0] must be zero on entry (the caller's memset is part of the op); out[0] += mean((|nab_i| - 1)^2)
int nsim_eikonal_loss_fwd(const float* nablas, int64_t S, float* out, void* stream) {
  if (S < 0) return 2;
  if (!out) return 4;
  if (S == 0) return 0;
  hipLaunchKernelGGL(k_eikonal_fwd, loss_grid(S), dim3(LOSS_BLOCK), 0, (hipStream_t)stream, nablas, S,
                     1.0f / (float)S, out);
  NSIM_CHECK_LAUNCH();
  return 0;
}

int nsim_eikonal_loss_bwd(const float* nablas, int64_t S, const float* gout, float* dnablas, void* stream) {
  if (S < 0) return 2;
  if (S == 0) return 0;
  if (!dnablas || !gout) return 26;
  hipLaunchKernelGGL(k_eikonal_bwd, dim3(nsim_blocks(S, LOSS_BLOCK)), dim3(LOSS_BLOCK), 0, (hipStream_t)stream, nablas,
                     S, 1.0f / (float)S, gout, dnablas);
  NSIM_CHECK_LAUNCH();
  return 0;
}

int nsim_mse_loss_fwd(const float* pred, const float* gt, int64_t n, float* out, void* stream) {
  if (n < 0) return 2;
  if (!out) return 4;
  if (n == 0) return 0;
  hipLaunchKernelGGL(k_mse_fwd, loss_grid(n), dim3(LOSS_BLOCK), 0, (hipStream_t)stream, pred, gt, n, 1.0f / (float)n,
                     out);
  NSIM_CHECK_LAUNCH();
  return 0;
}

int nsim_mse_loss_bwd(const float* pred, const float* gt, int64_t n, const float* gout, float* dpred, void* stream) {
  if (n < 0) return 2;
  if (n == 0) return 0;
  if (!dpred || !gout) return 26;
  hipLaunchKernelGGL(k_mse_bwd, dim3(nsim_blocks(n, LOSS_BLOCK)), dim3(LOSS_BLOCK), 0, (hipStream_t)stream, pred, gt, n,
                     1.0f / (float)n, gout, dpred);
  NSIM_CHECK_LAUNCH();
  return 0;
}

// acc [3] must be zero on entry.  n_img = number of image VALUES (rays * 3); nablas [S + M, 3]
int nsim_train_loss_head(const float* pred, const float* gt, int64_t n_img, const float* nablas, int64_t S, int64_t M,
                         float w_eikonal, float* acc, float* d_pred, float* d_nablas, void* stream) {
  if (n_img <= 0 || S < 0 || M < 0) return 2;
  if (!pred || !gt || !acc || !d_pred || (S + M > 0 && (!nablas || !d_nablas))) return 4;
  const int64_t top = n_img > S + M ? n_img : S + M;
  int64_t b = nsim_blocks(top, LOSS_BLOCK);
  if (b > LOSS_MAX_BLOCKS) b = LOSS_MAX_BLOCKS;     // three single-address atomics per block: they serialise at L2
  hipLaunchKernelGGL(k_loss_head, dim3((unsigned)b), dim3(LOSS_BLOCK), 0, (hipStream_t)stream, pred, gt, n_img, nablas, S,
                     M, w_eikonal, acc, d_pred, d_nablas);
  NSIM_CHECK_LAUNCH();
  return 0;
}

// out [rows, C] must be initialised by the caller (zeros for a plain gradient)
int nsim_rows_scatter_add(const float* g, const int64_t* idx, int64_t n, int C, int64_t rows, float* out, void* stream) {
  if (n < 0 || rows < 0) return 2;
  if (C <= 0) return 3;
  if (n == 0) return 0;
  if (!out) return 4;
  hipLaunchKernelGGL(k_rows_scatter_add, loss_grid(n * C), dim3(LOSS_BLOCK), 0, (hipStream_t)stream, g, idx, n, C, rows,
                     out);
  NSIM_CHECK_LAUNCH();
  return 0;
}

// out[i, :] = table[idx[i], :] for i < n, zeros for n <= i < n + tail (the forward of ``embed[fidx]`` with the zero rows of
// the step's appended free points in the same launch)
__global__ void __launch_bounds__(LOSS_BLOCK) k_rows_gather(const float* __restrict__ table, const int64_t* __restrict__ idx,
                                                            int64_t n, int C, int64_t rows, int64_t tail, float* __restrict__ out) {
  const int64_t tot = (n + tail) * C;
  for (int64_t i = (int64_t)blockIdx.x * LOSS_BLOCK + threadIdx.x; i < tot; i += (int64_t)gridDim.x * LOSS_BLOCK) {
    const int64_t r = i / C;
    float v = 0.f;
    if (r < n) {
      const int64_t k = idx[r];
      if (k >= 0 && k < rows) v = table[k * C + (i % C)];
    }
    out[i] = v;
  }
}

int nsim_rows_gather(const float* table, const int64_t* idx, int64_t n, int C, int64_t rows, int64_t tail, float* out,
                     void* stream) {
  if (n < 0 || tail < 0 || C <= 0) return 2;
  if (n + tail == 0) return 0;
  if (!out || (n > 0 && (!table || !idx))) return 4;
  hipLaunchKernelGGL(k_rows_gather, loss_grid((n + tail) * C), dim3(LOSS_BLOCK), 0, (hipStream_t)stream, table, idx, n, C, rows,
                     tail, out);
  NSIM_CHECK_LAUNCH();
  return 0;
}

int nsim_sphere_image(const float* rays_o, const float* rays_d, int64_t N, float radius, float* rgb, void* stream) {
  if (N < 0) return 2;
  if (N == 0) return 0;
  if (!rgb || !(radius > 0.f)) return 4;
  hipLaunchKernelGGL(k_sphere_image, dim3(nsim_blocks(N, LOSS_BLOCK)), dim3(LOSS_BLOCK), 0, (hipStream_t)stream, rays_o,
                     rays_d, N, radius, rgb);
  NSIM_CHECK_LAUNCH();
  return 0;
}

int nsim_gather_rays(const float* rays_o, const float* rays_d, const float* near, const float* far, const int64_t* idx,
                     int64_t R, float* o_out, float* d_out, float* near_out, float* far_out, void* stream) {
  if (R < 0) return 2;
  if (R == 0) return 0;
  if (!o_out || !d_out || !near_out || !far_out) return 4;
  hipLaunchKernelGGL(k_gather_rays, dim3(nsim_blocks(R, LOSS_BLOCK)), dim3(LOSS_BLOCK), 0, (hipStream_t)stream, rays_o,
                     rays_d, near, far, idx, R, o_out, d_out, near_out, far_out);
  NSIM_CHECK_LAUNCH();
  return 0;
}

}  // extern "C"
