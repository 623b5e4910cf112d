// atomic_bench.hip -- micro-benchmark: f32 atomic-add throughput on MI355X by memory scope and placement.
// Development aid for the LoTD gradient scatter design.  Build: hipcc --offload-arch=gfx950 -O3.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

__device__ __forceinline__ unsigned xcc_id() { return __builtin_amdgcn_s_getreg((3 << 11) | 20) & 0xf; }

template <int SCOPE, int REPLICA>
__global__ void k_atomic(float* tab, unsigned n_entries, int per_thread, unsigned seed, unsigned hot_mask) {
  unsigned tid = blockIdx.x * blockDim.x + threadIdx.x;
  unsigned x = tid * 2654435761u + seed;
  float* base = tab + (REPLICA ? (size_t)xcc_id() * n_entries : 0);
  for (int i = 0; i < per_thread; ++i) {
    x = x * 1664525u + 1013904223u;
    unsigned idx = (x >> 8) & hot_mask;
    if (SCOPE == 0) atomicAdd(base + idx, 1.0f);
    else if (SCOPE == 1) __hip_atomic_fetch_add(base + idx, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    else __hip_atomic_fetch_add(base + idx, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
  }
}

__global__ void k_xcc(unsigned* out) { if (threadIdx.x == 0) out[blockIdx.x] = xcc_id(); }

template <int SCOPE, int REPLICA>
void run(const char* name, float* tab, unsigned n_entries, unsigned hot_mask, int nrep) {
  const int blocks = 2048, threads = 256, per_thread = 256;
  hipMemset(tab, 0, sizeof(float) * (size_t)n_entries * 8);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k_atomic<SCOPE, REPLICA>), dim3(blocks), dim3(threads), 0, 0, tab, n_entries, per_thread, 1u, hot_mask);
  hipDeviceSynchronize();
  hipMemset(tab, 0, sizeof(float) * (size_t)n_entries * 8);
  hipEventRecord(e0);
  for (int r = 0; r < nrep; ++r)
    hipLaunchKernelGGL((k_atomic<SCOPE, REPLICA>), dim3(blocks), dim3(threads), 0, 0, tab, n_entries, per_thread, 7u + r, hot_mask);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double total = (double)blocks * threads * per_thread * nrep;
  // correctness: the sum over the table (all replicas) must equal the number of atomics
  std::vector<float> h((size_t)n_entries * 8);
  hipMemcpy(h.data(), tab, sizeof(float) * h.size(), hipMemcpyDeviceToHost);
  double sum = 0; for (float v : h) sum += v;
  printf("%-34s hot_mask=%08x : %8.2f G atomics/s   sum/expected = %.6f\n", name, hot_mask, total / ms * 1e-6, sum / total);
}

int main() {
  const unsigned n_entries = 1u << 24;   // 16 Mi floats = 64 MiB per replica
  float* tab; hipMalloc(&tab, sizeof(float) * (size_t)n_entries * 8);
  unsigned* x; hipMalloc(&x, 64 * 4);
  hipLaunchKernelGGL(k_xcc, dim3(16), dim3(64), 0, 0, x);
  unsigned hx[16]; hipMemcpy(hx, x, 64, hipMemcpyDeviceToHost);
  printf("xcc ids of blocks 0..15:"); for (int i = 0; i < 16; ++i) printf(" %u", hx[i]); printf("\n");
  for (unsigned mask : {0xffffffu, 0xfffffu, 0xfffu, 0xffu}) {
    run<0, 0>("agent scope, shared table", tab, n_entries, mask, 3);
    run<1, 0>("workgroup scope, shared table", tab, n_entries, mask, 3);
    run<1, 1>("workgroup scope, per-XCD replica", tab, n_entries, mask, 3);
    run<0, 1>("agent scope, per-XCD replica", tab, n_entries, mask, 3);
    run<2, 1>("wavefront scope, per-XCD replica", tab, n_entries, mask, 3);
  }
  return 0;
}
