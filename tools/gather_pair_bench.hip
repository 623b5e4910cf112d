// gather_pair_bench.hip -- is the hash-grid GATHER bound by load INSTRUCTION-lanes or by distinct cache lines, and does
// fetching the two x-corners of a cell as ONE 8-byte load pay?  (round 4)
// The spatial hash of a LoTD level is  cx ^ cy * p1 ^ cz * p2  (lotd_dev.h): the x-neighbour of an EVEN cx differs in bit 0 of
// the index only, i.e. the two entries (4 B each: two f16 features) are one naturally aligned 8-byte pair; for an odd cx the
// neighbour lives in another pair.  On dense levels the x-neighbour is always the next entry (8 bytes at 4-byte alignment).
//   mode 0 : 8 x global_load_dword per point                        (what k_lotd_gather_lm does)
//   mode 1 : hashed level, 4 x aligned dwordx2 for every lane + 4 x dwordx2 on the odd-cx lanes only (exec-masked)
//   mode 2 : dense level, 4 x dwordx2 at 4-byte alignment
//   mode 3 : 4 x aligned dwordx2 only (lower bound of mode 1: every cx even)
// Table 2^19 entries (2 MB), random cells per lane.  Reports G points/s (8 corner values delivered per point).
// Build: hipcc --offload-arch=gfx950 -O3 tools/gather_pair_bench.hip -o tools/gather_pair_bench
#include <hip/hip_runtime.h>
#include <stdio.h>

__device__ __forceinline__ unsigned hash3(unsigned x, unsigned y, unsigned z, unsigned mask) {
  return (x ^ (y * 2654435761u) ^ (z * 805459861u)) & mask;
}
struct __attribute__((packed, aligned(4))) U2 { unsigned a, b; };

// 8 x buffer_load_dword with cache-policy bits AUX (gfx940: bit 0 sc0, bit 1 nt, bit 4 sc1)
template <int AUX>
__global__ void __launch_bounds__(256) kaux(const unsigned* __restrict__ tab, unsigned mask, int iters, unsigned seed, unsigned* out) {
  const unsigned tid = blockIdx.x * blockDim.x + threadIdx.x;
  const int lane = threadIdx.x & 63;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned*>(tab), 0, 0xffffffff, 0x00020000);
  unsigned x = tid * 2654435761u + seed;
  unsigned acc = 0;
  for (int i = 0; i < iters; ++i) {
    x = x * 1664525u + 1013904223u; unsigned cx = x >> 12;
    x = x * 1664525u + 1013904223u; const unsigned cy = x >> 12;
    x = x * 1664525u + 1013904223u; const unsigned cz = x >> 12;
    unsigned v[8];
#pragma unroll
    for (int c = 0; c < 8; ++c)
      v[c] = (unsigned)__builtin_amdgcn_raw_buffer_load_b32(rs, (int)(4u * hash3(cx + (c & 1), cy + ((c >> 1) & 1), cz + (c >> 2), mask)), 0, AUX);
#pragma unroll
    for (int c = 0; c < 8; ++c) acc += v[c] * (unsigned)(c + 1 + lane);
  }
  if (acc == 0x12345678u) out[tid] = acc;
}
template <int AUX>
static double run_aux(const unsigned* tab, unsigned mask, unsigned* out) {
  const int blocks = 4096, iters = 64, nrep = 3;
  hipLaunchKernelGGL(kaux<AUX>, dim3(blocks), dim3(256), 0, 0, tab, mask, iters, 1u, out);
  hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  for (int r = 0; r < nrep; ++r) hipLaunchKernelGGL(kaux<AUX>, dim3(blocks), dim3(256), 0, 0, tab, mask, iters, 7u + r, out);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return (double)blocks * 256 * iters * nrep / ms * 1e-6;
}

template <int MODE>
__global__ void __launch_bounds__(256) k(const unsigned* __restrict__ tab, unsigned mask, int iters, unsigned seed, unsigned* out) {
  const unsigned tid = blockIdx.x * blockDim.x + threadIdx.x;
  const int lane = threadIdx.x & 63;
  unsigned x = tid * 2654435761u + seed;
  unsigned acc = 0;
  for (int i = 0; i < iters; ++i) {
    x = x * 1664525u + 1013904223u; unsigned cx = x >> 12;
    x = x * 1664525u + 1013904223u; const unsigned cy = x >> 12;
    x = x * 1664525u + 1013904223u; const unsigned cz = x >> 12;
    unsigned v[8];
    if (MODE == 0) {
#pragma unroll
      for (int c = 0; c < 8; ++c) v[c] = tab[hash3(cx + (c & 1), cy + ((c >> 1) & 1), cz + (c >> 2), mask)];
    } else if (MODE == 1 || MODE == 3) {
      if (MODE == 3) cx &= ~1u;
      const bool odd = cx & 1u;
      unsigned i0[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        i0[c] = hash3(cx, cy + (c & 1), cz + (c >> 1), mask);
        const uint2 p = *reinterpret_cast<const uint2*>(tab + (i0[c] & ~1u));
        v[2 * c] = (i0[c] & 1u) ? p.y : p.x;
        v[2 * c + 1] = (i0[c] & 1u) ? p.x : p.y;          // the x-neighbour when cx is even
      }
      if (odd) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const unsigned i1 = hash3(cx + 1u, cy + (c & 1), cz + (c >> 1), mask);
          const uint2 p = *reinterpret_cast<const uint2*>(tab + (i1 & ~1u));
          v[2 * c + 1] = (i1 & 1u) ? p.y : p.x;
        }
      }
    } else {
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const unsigned i0 = hash3(cx, cy + (c & 1), cz + (c >> 1), mask - 1u);
        const U2 p = *reinterpret_cast<const U2*>(tab + i0);
        v[2 * c] = p.a;
        v[2 * c + 1] = p.b;
      }
    }
#pragma unroll
    for (int c = 0; c < 8; ++c) acc += v[c] * (unsigned)(c + 1 + lane);
  }
  if (acc == 0x12345678u) out[tid] = acc;
}

template <int MODE>
static double run(const unsigned* tab, unsigned mask, unsigned* out) {
  const int blocks = 4096, iters = 64, nrep = 3;
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, tab, mask, iters, 1u, out);
  hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  for (int r = 0; r < nrep; ++r) hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, tab, mask, iters, 7u + r, out);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return (double)blocks * 256 * iters * nrep / ms * 1e-6;      // G points per second
}

int main() {
  unsigned *tab, *out;
  hipMalloc(&out, 4096 * 256 * 4);
  for (int lg : {19, 22, 25}) {
    const unsigned n = 1u << lg;
    hipMalloc(&tab, (size_t)n * 4); hipMemset(tab, 1, (size_t)n * 4);
    printf("table 2^%d entries (%d MB): G points/s  8 x dword %.1f | hashed pairs (4 + 4 masked) %.1f | dense unaligned pairs %.1f | 4 aligned pairs %.1f\n",
           lg, (int)(n >> 18), run<0>(tab, n - 1, out), run<1>(tab, n - 1, out), run<2>(tab, n - 1, out), run<3>(tab, n - 1, out));
    printf("   8 x buffer_load_dword, cache policy: none %.1f | sc0 %.1f | nt %.1f | sc0 nt %.1f | sc1 %.1f | sc0 sc1 %.1f | sc1 nt %.1f | sc0 sc1 nt %.1f\n",
           run_aux<0>(tab, n - 1, out), run_aux<1>(tab, n - 1, out), run_aux<2>(tab, n - 1, out), run_aux<3>(tab, n - 1, out),
           run_aux<16>(tab, n - 1, out), run_aux<17>(tab, n - 1, out), run_aux<18>(tab, n - 1, out), run_aux<19>(tab, n - 1, out));
    hipFree(tab);
  }
  return 0;
}
