// gather_dedup_bench.hip -- does wavefront-level de-duplication pay for the hash-grid GATHER (VERDICT r3 item 5 /
// north_star "hash gather with wavefront-level dedup")?  Lanes of a wave are consecutive samples of a ray; on a level whose
// cell is wider than the sample spacing, runs of K neighbouring lanes fall into the same cell and read the same 8 corners.
//   plain : every lane issues its 8 corner loads (what k_lotd_gather_lm does; same-address lanes of one instruction meet in
//           the texture unit / L1)
//   dedup : run heads are found with a ballot, ONLY the head of a run loads the 8 corners (exec-masked loads), the other
//           lanes of the run fetch the 8 words from their head through ds_bpermute
// One hashed level of T = 2^19 entries (2 MB: resident in the 4 MB L2 of an XCD), cells drawn at random per RUN, runs of
// K = 1, 2, 4, 8, 16 lanes.  Reports G lane-corner-values/s delivered.  Development aid.
// Build: hipcc --offload-arch=gfx950 -O3 tools/gather_dedup_bench.hip -o tools/gather_dedup_bench
#include <hip/hip_runtime.h>
#include <stdio.h>

__device__ __forceinline__ unsigned hash3(unsigned x, unsigned y, unsigned z, unsigned mask) {
  return (x ^ (y * 2654435761u) ^ (z * 805459861u)) & mask;
}

template <bool DEDUP>
__global__ void __launch_bounds__(256) k(const unsigned* __restrict__ tab, unsigned mask, int iters, int K, unsigned seed, unsigned* out) {
  const unsigned tid = blockIdx.x * blockDim.x + threadIdx.x;
  const int lane = threadIdx.x & 63;
  const unsigned run = tid / (unsigned)K;                     // lanes of a run share the cell
  unsigned x = run * 2654435761u + seed;
  unsigned acc = 0;
  for (int i = 0; i < iters; ++i) {
    x = x * 1664525u + 1013904223u; const unsigned cx = x >> 12;
    x = x * 1664525u + 1013904223u; const unsigned cy = x >> 12;
    x = x * 1664525u + 1013904223u; const unsigned cz = x >> 12;
    unsigned v[8];
    if (DEDUP) {
      const unsigned key = hash3(cx, cy, cz, 0xffffffffu);
      const unsigned prev = __shfl(key, lane - 1, 64);
      const bool head = lane == 0 || prev != key;
      const unsigned long long heads = __ballot(head);
      const unsigned long long below = heads & ((2ull << lane) - 1ull);
      const int src = 63 - __builtin_clzll(below);          // the head of this lane's run
      if (head) {
#pragma unroll
        for (int c = 0; c < 8; ++c) v[c] = tab[hash3(cx + (c & 1), cy + ((c >> 1) & 1), cz + (c >> 2), mask)];
      }
#pragma unroll
      for (int c = 0; c < 8; ++c) v[c] = __shfl(head ? v[c] : 0u, src, 64);
    } else {
#pragma unroll
      for (int c = 0; c < 8; ++c) v[c] = tab[hash3(cx + (c & 1), cy + ((c >> 1) & 1), cz + (c >> 2), mask)];
    }
#pragma unroll
    for (int c = 0; c < 8; ++c) acc += v[c] * (unsigned)(c + 1 + lane);
  }
  if (acc == 0x12345678u) out[tid] = acc;
}

template <bool DEDUP>
static double run(const unsigned* tab, unsigned mask, int K, unsigned* out) {
  const int blocks = 4096, iters = 64, nrep = 3;
  hipLaunchKernelGGL(k<DEDUP>, dim3(blocks), dim3(256), 0, 0, tab, mask, iters, K, 1u, out);
  hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  for (int r = 0; r < nrep; ++r) hipLaunchKernelGGL(k<DEDUP>, dim3(blocks), dim3(256), 0, 0, tab, mask, iters, K, 7u + r, out);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return (double)blocks * 256 * iters * 8 * nrep / ms * 1e-6;      // G corner values delivered per second
}

int main() {
  unsigned *tab, *out;
  const unsigned n = 1u << 19;
  hipMalloc(&tab, n * 4); hipMemset(tab, 1, n * 4);
  hipMalloc(&out, 4096 * 256 * 4);
  printf("hashed level T = 2^19 (2 MB), 4096 x 256 lanes x 64 points; G corner values / s delivered to lanes\n");
  for (int K : {1, 2, 4, 8, 16}) {
    const double p = run<false>(tab, n - 1, K, out), d = run<true>(tab, n - 1, K, out);
    printf("run length K = %2d : plain %8.1f   dedup %8.1f   (dedup / plain = %.2f)\n", K, p, d, d / p);
  }
  return 0;
}
