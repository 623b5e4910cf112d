// optim.hip -- fused Adam step for gfx950 (f32 master params + optional fp16 shadow used by the kernels).
// Replaces torch.optim.Adam as configured by the reference's ``training_cfg{lr, eps 1e-15, betas [.9,.99]}``
// (code_single/configs/object_centric/lotd_neus.dtu.230814.yaml:178-184; step site
// code_single/tools/train.py:1494-1502).  Pure HBM streaming: 16 B read + 12..14 B written per parameter.
#include "nsim_common.h"

__global__ void __launch_bounds__(256) k_adam(float* __restrict__ p, f16* __restrict__ p16, float* __restrict__ grad,
                                               float* __restrict__ m, float* __restrict__ v, int64_t n, float lr,
                                               float b1, float b2, float eps, float bias1, float bias2,
                                               float grad_scale, int zero_grad) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  // zero_grad: bit 0 = zero the gradient; bit 1 = TOUCHED-ENTRIES ("lazy") mode -- an entry whose gradient is exactly zero
  // this step is left alone: no moment decay, no update, nothing but its gradient word read (the rule of
  // torch.optim.SparseAdam).  Opt-in: a dense Adam moves such an entry by its decaying first moment, so the trajectory
  // differs from the reference's optimizer.
  const bool lazy = (zero_grad & 2) != 0;
  zero_grad &= 1;
  for (; i < n; i += stride) {
    const float g = grad[i] * grad_scale;
    if (lazy && g == 0.f) continue;
    const float mi = b1 * m[i] + (1.0f - b1) * g;
    const float vi = b2 * v[i] + (1.0f - b2) * g * g;
    m[i] = mi;
    v[i] = vi;
    // torch.optim.Adam: p -= lr/bias1 * m / (sqrt(v)/sqrt(bias2) + eps)
    const float denom = sqrtf(vi) / sqrtf(bias2) + eps;
    const float pi = p[i] - (lr / bias1) * (mi / denom);
    p[i] = pi;
    if (p16) p16[i] = (f16)pi;
    if (zero_grad) grad[i] = 0.f;
  }
}

// The small parameter tensors of a step (decoder weights and biases, inv_s, appearance codes: seven launches of a few
// thousand elements each) in ONE launch: blockIdx.y selects the tensor.
struct AdamMulti {
  NsimAdamTensor t[NSIM_ADAM_MULTI_MAX];
};

__global__ void __launch_bounds__(256) k_adam_multi(AdamMulti a, float lr, float eps, float grad_scale, int zero_grad) {
  const NsimAdamTensor t = a.t[blockIdx.y];
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  f16* p16 = (f16*)t.p16;
  for (; i < t.n; i += stride) {
    const float g = t.grad[i] * grad_scale;
    const float mi = t.beta1 * t.m[i] + (1.0f - t.beta1) * g;
    const float vi = t.beta2 * t.v[i] + (1.0f - t.beta2) * g * g;
    t.m[i] = mi;
    t.v[i] = vi;
    const float denom = sqrtf(vi) / sqrtf(t.bias2) + eps;
    const float pi = t.p[i] - (lr * t.lr_scale / t.bias1) * (mi / denom);
    t.p[i] = pi;
    if (p16) p16[i] = (f16)pi;
    if (zero_grad) t.grad[i] = 0.f;
  }
}

extern "C" int nsim_adam_multi(const NsimAdamTensor* tensors, int n_tensors, float lr, float eps, float grad_scale,
                               int zero_grad, void* stream) {
  if (n_tensors <= 0) return 0;
  if (!tensors || n_tensors > NSIM_ADAM_MULTI_MAX) return 2;
  AdamMulti a;
  int64_t nmax = 0;
  for (int k = 0; k < n_tensors; ++k) {
    a.t[k] = tensors[k];
    if (!a.t[k].p || !a.t[k].grad || !a.t[k].m || !a.t[k].v) return 4;
    nmax = a.t[k].n > nmax ? a.t[k].n : nmax;
  }
  if (nmax <= 0) return 0;
  hipLaunchKernelGGL(k_adam_multi, dim3(nsim_blocks(nmax, 256, 1024), n_tensors), dim3(256), 0, (hipStream_t)stream, a, lr,
                     eps, grad_scale, zero_grad);
  NSIM_CHECK_LAUNCH();
  return 0;
}

extern "C" int nsim_adam_step(float* p, void* p16, float* grad, float* m, float* v, int64_t n, float lr, float beta1,
                              float beta2, float eps, float bias1, float bias2, float grad_scale, int zero_grad,
                              void* stream) {
  if (n <= 0) return 0;
  hipLaunchKernelGGL(k_adam, dim3(nsim_blocks(n, 256, 8192)), dim3(256), 0, (hipStream_t)stream, p, (f16*)p16, grad, m, v,
                     n, lr, beta1, beta2, eps, bias1, bias2, grad_scale, zero_grad);
  NSIM_CHECK_LAUNCH();
  return 0;
}
