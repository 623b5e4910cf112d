// atomic_bench4.hip -- micro-benchmark + correctness probe (round 4): do float atomics of a LOWER memory scope retire faster
// than the agent-scope atomicAdd the gradient scatters issue, and are they exact when every address is only ever touched
// from ONE XCD?  Agent-scope atomics of a multi-XCD device cannot execute in an XCD's (non-coherent) L2; workgroup-scope
// ones may.  The scatter can guarantee single-XCD ownership WITHOUT assuming a block -> XCD placement: a workgroup reads the
// XCD it actually runs on (s_getreg_b32 HW_REG_XCC_ID) and only works on the table slices dealt to that XCD.
//   mode A  agent scope, any block touches any address                       (what k_lotd_scatter does today)
//   mode W  workgroup scope, addresses partitioned by the ACTUAL XCC_ID       (candidate)
//   mode P  agent scope, same partition                                        (separates the partition from the scope)
// Every mode issues the scatter's quad-transposed pattern (4 lanes = one x-adjacent vertex pair = one 16-byte request).
// Correctness: the table must sum to the number of adds issued (each add is +1.0, counts stay < 2^24).
// Build: hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics tools/atomic_bench4.hip -o tools/atomic_bench4
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

__device__ __forceinline__ unsigned xcc_id() {
  unsigned v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
  return v & 0xfu;
}

template <int MODE>
__global__ void k(float* tab, unsigned slice_mask, int per_thread, unsigned seed, unsigned* xcd_blocks) {
  const unsigned tid = blockIdx.x * blockDim.x + threadIdx.x;
  const unsigned xcd = xcc_id();
  if (threadIdx.x == 0) atomicAdd(xcd_blocks + xcd, 1u);
  unsigned x = (tid >> 2) * 2654435761u + seed;
  // MODE A: the whole table (8 slices) is one address space; W / P: this XCD's slice only
  float* base = (MODE == 0) ? tab : tab + (size_t)xcd * 2 * (slice_mask + 1);
  const unsigned mask = (MODE == 0) ? (8 * (slice_mask + 1) - 1) : slice_mask;
  for (int i = 0; i < per_thread; ++i) {
    x = x * 1664525u + 1013904223u;
    const unsigned v = (x >> 8) & mask;
    float* p = base + 2 * (v & ~1u) + (tid & 3);
    if (MODE == 1) __hip_atomic_fetch_add(p, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    else atomicAdd(p, 1.0f);
  }
}

template <int MODE>
void run(const char* name, float* tab, unsigned slice_log2, unsigned* xcd_blocks) {
  const int blocks = 2048, threads = 256, per_thread = 128, nrep = 3;
  const size_t n = (size_t)16 << slice_log2;      // 8 slices x 2 floats
  hipMemset(tab, 0, n * sizeof(float));
  hipMemset(xcd_blocks, 0, 16 * sizeof(unsigned));
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, tab, (1u << slice_log2) - 1, per_thread, 1u, xcd_blocks);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int r = 0; r < nrep; ++r)
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, tab, (1u << slice_log2) - 1, per_thread, 7u + r, xcd_blocks);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  float* h = (float*)malloc(n * sizeof(float));
  hipMemcpy(h, tab, n * sizeof(float), hipMemcpyDeviceToHost);
  double sum = 0.0;
  for (size_t i = 0; i < n; ++i) sum += h[i];
  free(h);
  unsigned xb[16];
  hipMemcpy(xb, xcd_blocks, sizeof(xb), hipMemcpyDeviceToHost);
  const double adds = (double)blocks * threads * per_thread * (nrep + 1);
  printf("%-34s slice 2^%u vertices (%5.1f MB / XCD): %7.2f G lane-atomics/s = %6.2f G vertices/s (%6.2f ms)  sum %s (%.0f of %.0f)  blocks/xcd",
         name, slice_log2, 8.0 * (1u << slice_log2) / 1048576.0, (adds * nrep / (nrep + 1)) / ms * 1e-6,
         (adds * nrep / (nrep + 1)) / 2.0 / ms * 1e-6, ms, sum == adds ? "EXACT" : "WRONG", sum, adds);
  for (int i = 0; i < 8; ++i) printf(" %u", xb[i]);
  printf("\n");
}

int main() {
  float* tab; unsigned* xb;
  hipMalloc(&tab, sizeof(float) * ((size_t)16 << 20));
  hipMalloc(&xb, 16 * sizeof(unsigned));
  for (unsigned lg : {16u, 19u, 20u}) {      // per-XCD slice: 0.5 MB (L2-resident), 4 MB (one hashed level, T = 2^19), 8 MB (street, T = 2^20)
    run<0>("A agent scope, unpartitioned", tab, lg, xb);
    run<2>("P agent scope, XCC_ID partition", tab, lg, xb);
    run<1>("W workgroup scope, XCC_ID partition", tab, lg, xb);
  }
  return 0;
}
