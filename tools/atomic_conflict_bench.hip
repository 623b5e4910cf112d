// atomic_conflict_bench.hip -- what does it cost when several QUADS of one wave instruction add to the SAME four dwords?
// Background (profiles/round4_atomic_line_bench.txt): the float-atomic rate of the MI355X is paid per distinct 64-byte sector
// per wave instruction (~21 G sectors/s), however many lanes of the instruction fall into the sector.  k_lotd_scatter issues
// one quad (16 bytes: an x-pair of corners x two features) per sample; when it puts CONSECUTIVE samples of a ray into one
// instruction, samples in the same cell hit the SAME 16 bytes.  This measures whether such same-address lanes ride for free
// (one sector request) or serialise.
//   K = quads of an instruction sharing one address group (K = 1: all 16 quads distinct sectors).
// Build: hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics tools/atomic_conflict_bench.hip -o tools/atomic_conflict_bench
#include <hip/hip_runtime.h>
#include <stdio.h>

// mode 0: the K quads hit the same 16 bytes.  mode 1: the K quads hit K different 16-byte pieces of ONE 64-byte sector (K <= 4).
__global__ void __launch_bounds__(256) k(float* tab, unsigned sector_mask, int K, int mode, int iters, unsigned seed) {
  const unsigned tid = blockIdx.x * blockDim.x + threadIdx.x;
  const unsigned wave = tid >> 6, lane = tid & 63u, quad = lane >> 2;
  const unsigned grp = quad / (unsigned)K, sub = quad % (unsigned)K;
  unsigned x = (wave * 16u + grp) * 2654435761u + seed;
  for (int i = 0; i < iters; ++i) {
    x = x * 1664525u + 1013904223u;
    const unsigned sector = (x >> 7) & sector_mask;
    const unsigned piece = mode == 0 ? 0u : (sub & 3u);
    atomicAdd(tab + (size_t)sector * 16 + 4u * piece + (lane & 3u), 1.0f);
  }
}

int main() {
  float* tab;
  const unsigned sectors = 1u << 19;      // 2^19 sectors x 64 B = 32 MB
  (void)hipMalloc(&tab, (size_t)sectors * 64);
  (void)hipMemset(tab, 0, (size_t)sectors * 64);
  const int blocks = 4096, iters = 64, nrep = 3;
  for (int mode : {0, 1})
    for (int K : {1, 2, 4, 8, 16}) {
      if (mode == 1 && K > 4) continue;
      hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, tab, sectors - 1, K, mode, iters, 1u);
      (void)hipDeviceSynchronize();
      hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
      (void)hipEventRecord(e0);
      for (int r = 0; r < nrep; ++r) hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, tab, sectors - 1, K, mode, iters, 7u + r);
      (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
      float ms; (void)hipEventElapsedTime(&ms, e0, e1);
      const double quads = (double)blocks * 256 / 4 * iters * nrep;
      printf("%s, K = %2d quads per group: %7.2f G quads/s = %6.2f G sectors/s  (%.3f ms)\n",
             mode == 0 ? "same 16 bytes      " : "same sector, pieces", K, quads / ms * 1e-6, quads / K / ms * 1e-6, ms);
    }
  return 0;
}
