// atomic_line_bench.hip -- what is the unit the float-atomic rate is paid in: lanes, 16-byte quads, or 128-byte lines?
// Every wave instruction updates 64 / G random lines of a 32 MB f32 table, G consecutive lanes adding to G consecutive dwords of
// the same line (G = 1 .. 32).  If the atomic unit retired LINES per instruction, lane-atomics/s would grow with G; if it
// retires 16-byte quads, it saturates at G = 4.  (k_lotd_scatter issues quads: DESIGN.md sec. 4, round 4.)
// Build: hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics tools/atomic_line_bench.hip -o tools/atomic_line_bench
#include <hip/hip_runtime.h>
#include <stdio.h>

__global__ void __launch_bounds__(256) k(float* tab, unsigned line_mask, int G, int iters, unsigned seed) {
  const unsigned tid = blockIdx.x * blockDim.x + threadIdx.x;
  const unsigned grp = tid / (unsigned)G, sub = tid % (unsigned)G;
  unsigned x = grp * 2654435761u + seed;
  for (int i = 0; i < iters; ++i) {
    x = x * 1664525u + 1013904223u;
    const unsigned line = (x >> 7) & line_mask;
    atomicAdd(tab + (size_t)line * 32 + sub, 1.0f);
  }
}

// non-adjacent lanes of one instruction in the same 64-byte sector: quad q (lanes 4q .. 4q+3) and quad q + 8 share a sector
// (dwords 0..3 and 4..7 of it); MODE 1: the same pairing but the two quads sit next to each other (lanes 8p .. 8p+7)
__global__ void __launch_bounds__(256) k2(float* tab, unsigned line_mask, int mode, int iters, unsigned seed) {
  const unsigned tid = blockIdx.x * blockDim.x + threadIdx.x;
  const unsigned wave = tid >> 6, lane = tid & 63u, quad = lane >> 2;
  const unsigned pair = mode == 0 ? (quad & 7u) : (quad >> 1), half = mode == 0 ? (quad >> 3) : (quad & 1u);
  unsigned x = (wave * 8u + pair) * 2654435761u + seed;
  for (int i = 0; i < iters; ++i) {
    x = x * 1664525u + 1013904223u;
    const unsigned line = (x >> 7) & line_mask;
    atomicAdd(tab + (size_t)line * 32 + 4u * half + (lane & 3u), 1.0f);
  }
}

int main() {
  float* tab;
  const unsigned lines = 1u << 18;      // 2^18 lines x 128 B = 32 MB
  (void)hipMalloc(&tab, (size_t)lines * 128);
  (void)hipMemset(tab, 0, (size_t)lines * 128);
  const int blocks = 4096, iters = 64, nrep = 3;
  for (int G : {1, 2, 4, 8, 16, 32}) {
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, tab, lines - 1, G, iters, 1u);
    (void)hipDeviceSynchronize();
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0);
    for (int r = 0; r < nrep; ++r) hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, tab, lines - 1, G, iters, 7u + r);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double lane_atomics = (double)blocks * 256 * iters * nrep;
    printf("G = %2d lanes per line: %7.1f G lane-atomics/s = %6.2f G lines/s = %6.2f G 16-byte quads/s\n", G,
           lane_atomics / ms * 1e-6, lane_atomics / G / ms * 1e-6, lane_atomics / (G < 4 ? G : 4) / ms * 1e-6 / (G < 4 ? 1 : 1));
  }
  for (int mode : {1, 0}) {
    hipLaunchKernelGGL(k2, dim3(blocks), dim3(256), 0, 0, tab, lines - 1, mode, iters, 1u);
    (void)hipDeviceSynchronize();
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0);
    for (int r = 0; r < nrep; ++r) hipLaunchKernelGGL(k2, dim3(blocks), dim3(256), 0, 0, tab, lines - 1, mode, iters, 7u + r);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("two quads per 64-byte sector, %s: %7.1f G lane-atomics/s\n", mode ? "adjacent lanes (8p .. 8p+7)" : "32 lanes apart (quad q and q + 8)",
           (double)blocks * 256 * iters * nrep / ms * 1e-6);
  }
  return 0;
}
