// atomic_alloc_bench.hip -- round 5: does the RATE of float atomics depend on how the gradient table was
// allocated?  k_lotd_scatter sits at 0.91 of the 20.7 G requests/s the atomic micro-benchmarks measure, and the L2 counters of
// its launches read TCC_MISS ~ request count although the 48.8 MB table is the only thing it writes.  Rounds 1-4 showed the rate
// is independent of memory scope, of the table size (4 MB .. 64 MB: tools/atomic_bench4) and of partitioning the table by XCD.
// This probe varies the ALLOCATION: hipMalloc (coarse-grained, what torch's caching allocator hands out), fine-grained and
// uncached device memory (hipExtMallocWithFlags), managed memory with the preferred-location / coarse-grain advice -- same
// kernel: the scatter's quad-transposed issue (4 lanes = one x-adjacent vertex pair = one 16-byte request), random vertices.
// Also: the same adds as plain (non-atomic, racy) read-modify-write stores -- what the memory system does for an ordinary
// scattered RMW stream of the same shape, as an upper reference.
// Build: hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics tools/atomic_alloc_bench.hip -o tools/atomic_alloc_bench
#include <hip/hip_runtime.h>
#include <stdio.h>

template <int MODE>
__global__ void __launch_bounds__(256) k(float* tab, unsigned mask, int per_thread, unsigned seed) {
  const unsigned tid = blockIdx.x * blockDim.x + threadIdx.x;
  unsigned x = (tid >> 2) * 2654435761u + seed;
  for (int i = 0; i < per_thread; ++i) {
    x = x * 1664525u + 1013904223u;
    const unsigned v = (x >> 8) & mask;
    float* p = tab + 2 * (v & ~1u) + (tid & 3);
    if (MODE == 0) atomicAdd(p, 1.0f);
    else *p = *p + 1.0f;                      // racy plain RMW: reference only
  }
}

template <int MODE>
static double run(float* tab, unsigned mask) {
  const int blocks = 8192, per = 64, nrep = 4;
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, tab, mask, per, 1u);
  hipDeviceSynchronize();
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipEventRecord(e0);
  for (int r = 0; r < nrep; ++r) hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, tab, mask, per, 7u + r);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  return (double)blocks * 256 * per * nrep / 4.0 / (ms * 1e-3) * 1e-9;      // G 16-byte requests / s
}

int main() {
  const unsigned nv = 1u << 22;                 // 4 Mi vertices x 2 floats = 32 MB table
  const size_t bytes = (size_t)nv * 2 * sizeof(float);
  struct { const char* name; int kind; } allocs[] = {{"hipMalloc (coarse-grained: torch's allocator)", 0},
                                                     {"hipExtMallocWithFlags(FineGrained)", 1},
                                                     {"hipExtMallocWithFlags(Uncached)", 2},
                                                     {"hipMallocManaged + preferred location + coarse-grain advice", 3}};
  for (auto& al : allocs) {
    float* tab = nullptr;
    hipError_t e = hipSuccess;
    if (al.kind == 0) e = hipMalloc(&tab, bytes);
    else if (al.kind == 1) e = hipExtMallocWithFlags((void**)&tab, bytes, hipDeviceMallocFinegrained);
    else if (al.kind == 2) e = hipExtMallocWithFlags((void**)&tab, bytes, hipDeviceMallocUncached);
    else {
      e = hipMallocManaged(&tab, bytes);
      if (e == hipSuccess) {
        hipMemAdvise(tab, bytes, hipMemAdviseSetPreferredLocation, 0);
        hipMemAdvise(tab, bytes, hipMemAdviseSetCoarseGrain, 0);
        hipMemPrefetchAsync(tab, bytes, 0, 0);
      }
    }
    if (e != hipSuccess || !tab) {
      printf("%-62s allocation failed (%s)\n", al.name, hipGetErrorString(e));
      continue;
    }
    hipMemset(tab, 0, bytes);
    const double a = run<0>(tab, nv - 1);
    const double p = run<1>(tab, nv - 1);
    printf("%-62s atomicAdd %6.2f G requests/s    plain racy RMW %6.2f G requests/s\n", al.name, a, p);
    hipFree(tab);
  }
  return 0;
}
