// atomic_bench2.hip -- does the f32 atomic rate depend on intra-instruction address locality?
// Groups of G adjacent lanes add to G adjacent dwords of one random base.  Development aid.
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdio.h>
#include <vector>

template <int G, int MODE>
__global__ void k_atomic(float* tab, unsigned n_entries, int per_thread, unsigned seed) {
  unsigned tid = blockIdx.x * blockDim.x + threadIdx.x;
  unsigned grp = tid / G, sub = tid % G;
  unsigned x = grp * 2654435761u + seed;
  for (int i = 0; i < per_thread; ++i) {
    x = x * 1664525u + 1013904223u;
    unsigned idx = (((x >> 8) & (n_entries - 1)) & ~(unsigned)(G - 1)) + sub;
    if (MODE == 0) atomicAdd(tab + idx, 1.0f);
    else if (MODE == 1) {  // packed f16 pair
      __half2 v = __floats2half2_rn(1.0f, 1.0f);
      unsafeAtomicAdd(reinterpret_cast<__half2*>(tab) + idx, v);
    } else if (MODE == 2) {
      atomicAdd(reinterpret_cast<unsigned long long*>(tab) + (idx >> 1), 1ull);
    }
  }
}

template <int G, int MODE>
void run(const char* name, float* tab, unsigned n_entries) {
  const int blocks = 2048, threads = 256, per_thread = 256, nrep = 3;
  hipMemset(tab, 0, sizeof(float) * (size_t)n_entries);
  hipLaunchKernelGGL((k_atomic<G, MODE>), dim3(blocks), dim3(threads), 0, 0, tab, n_entries, per_thread, 1u);
  hipDeviceSynchronize();
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  for (int r = 0; r < nrep; ++r)
    hipLaunchKernelGGL((k_atomic<G, MODE>), dim3(blocks), dim3(threads), 0, 0, tab, n_entries, per_thread, 7u + r);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double total = (double)blocks * threads * per_thread * nrep;
  printf("%-28s G=%2d : %8.2f G lane-atomics/s\n", name, G, total / ms * 1e-6);
}

int main() {
  const unsigned n_entries = 1u << 24;
  float* tab; hipMalloc(&tab, sizeof(float) * (size_t)n_entries);
  run<1, 0>("f32 add", tab, n_entries);
  run<2, 0>("f32 add", tab, n_entries);
  run<4, 0>("f32 add", tab, n_entries);
  run<8, 0>("f32 add", tab, n_entries);
  run<16, 0>("f32 add", tab, n_entries);
  run<64, 0>("f32 add", tab, n_entries);
  run<1, 1>("pk f16x2 add", tab, n_entries);
  run<4, 1>("pk f16x2 add", tab, n_entries);
  run<1, 2>("u64 add", tab, n_entries);
  run<4, 2>("u64 add", tab, n_entries);
  return 0;
}
