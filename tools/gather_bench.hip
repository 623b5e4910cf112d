// gather_bench.hip -- what does a random 4-byte (fp16x2) gather cost on MI355X?  Development aid.
// Each lane issues ILP independent loads per iteration from a table of n_entries dwords; addresses are
//   MODE 0: fully random per lane
//   MODE 1: random base per lane, 2 loads at (base, base+1)      -- the x / x+1 corner pair, two dword loads
//   MODE 2: random base per lane, one dwordx2 load at base&~1     -- the pair as one 8-byte load
//   MODE 3: 8 corners of a random cell of a dense R^3 level (R = cbrt(n)), 8 dword loads
//   MODE 4: same cell, 4 dwordx2 loads (x pairs)
// Reports G lane-loads/s (dword-equivalents fetched usefully).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>

template <int MODE, int ILP>
__global__ void __launch_bounds__(256) k_gather(const unsigned* __restrict__ tab, unsigned n_entries, int iters, unsigned seed,
                                                 unsigned* out, int R) {
  unsigned tid = blockIdx.x * blockDim.x + threadIdx.x;
  unsigned x = tid * 2654435761u + seed;
  unsigned acc = 0;
  for (int i = 0; i < iters; ++i) {
    unsigned v[ILP * 2];
#pragma unroll
    for (int k = 0; k < ILP; ++k) {
      x = x * 1664525u + 1013904223u;
      const unsigned idx = (x >> 8) & (n_entries - 1);
      if (MODE == 0) {
        v[2 * k] = tab[idx]; v[2 * k + 1] = 0;
      } else if (MODE == 1) {
        const unsigned b = idx & (n_entries - 2);
        v[2 * k] = tab[b]; v[2 * k + 1] = tab[b + 1];
      } else if (MODE == 2) {
        const uint2 p = *reinterpret_cast<const uint2*>(tab + (idx & ~1u));
        v[2 * k] = p.x; v[2 * k + 1] = p.y;
      }
    }
#pragma unroll
    for (int k = 0; k < ILP * 2; ++k) acc += v[k];
  }
  if (acc == 0x12345678u) out[tid] = acc;
}

// random dword gather through a buffer resource with cache-policy bits AUX (gfx94x/95x: 1 = sc0, 2 = nt, 16 = sc1)
template <int AUX, int ILP>
__global__ void __launch_bounds__(256) k_gather_buf(const unsigned* __restrict__ tab, unsigned n_entries, int iters,
                                                     unsigned seed, unsigned* out) {
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned*>(tab), 0, 0xffffffff, 0x00020000);
  unsigned tid = blockIdx.x * blockDim.x + threadIdx.x;
  unsigned x = tid * 2654435761u + seed;
  unsigned acc = 0;
  for (int i = 0; i < iters; ++i) {
    unsigned v[ILP];
#pragma unroll
    for (int k = 0; k < ILP; ++k) {
      x = x * 1664525u + 1013904223u;
      const unsigned idx = (x >> 8) & (n_entries - 1);
      v[k] = __builtin_amdgcn_raw_buffer_load_b32(r, (int)(idx * 4u), 0, AUX);
    }
#pragma unroll
    for (int k = 0; k < ILP; ++k) acc += v[k];
  }
  if (acc == 0x12345678u) out[tid] = acc;
}

template <int MODE>
__global__ void __launch_bounds__(256) k_cell(const unsigned* __restrict__ tab, int R, int iters, unsigned seed, unsigned* out) {
  unsigned tid = blockIdx.x * blockDim.x + threadIdx.x;
  unsigned x = tid * 2654435761u + seed;
  unsigned acc = 0;
  for (int i = 0; i < iters; ++i) {
    x = x * 1664525u + 1013904223u; const unsigned cx = (x >> 8) % (unsigned)(R - 1);
    x = x * 1664525u + 1013904223u; const unsigned cy = (x >> 8) % (unsigned)(R - 1);
    x = x * 1664525u + 1013904223u; const unsigned cz = (x >> 8) % (unsigned)(R - 1);
    unsigned v[8];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const unsigned b = cx + R * ((cy + (c & 1)) + R * (cz + (c >> 1)));
      if (MODE == 3) { v[2 * c] = tab[b]; v[2 * c + 1] = tab[b + 1]; }
      else {
        // 4-byte aligned 8-byte load (global loads only need dword alignment)
        const uint2 p = *reinterpret_cast<const uint2*>(tab + b);
        v[2 * c] = p.x; v[2 * c + 1] = p.y;
      }
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) acc += v[k];
  }
  if (acc == 0x12345678u) out[tid] = acc;
}

static double timeit(void (*launch)(int), int nrep) {
  launch(0);
  hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  for (int r = 0; r < nrep; ++r) launch(r + 1);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms / nrep;
}

static unsigned* g_tab; static unsigned* g_out; static unsigned g_n; static int g_R; static int g_blocks;
template <int MODE, int ILP> void launch_g(int r) {
  hipLaunchKernelGGL((k_gather<MODE, ILP>), dim3(g_blocks), dim3(256), 0, 0, g_tab, g_n, 64, 7u + r, g_out, g_R);
}
template <int AUX> void launch_b(int r) {
  hipLaunchKernelGGL((k_gather_buf<AUX, 8>), dim3(g_blocks), dim3(256), 0, 0, g_tab, g_n, 64, 7u + r, g_out);
}
template <int MODE> void launch_c(int r) {
  hipLaunchKernelGGL((k_cell<MODE>), dim3(g_blocks), dim3(256), 0, 0, g_tab, g_R, 64, 7u + r, g_out);
}

int main() {
  const size_t maxn = 1u << 26;
  hipMalloc(&g_tab, maxn * 4); hipMemset(g_tab, 1, maxn * 4);
  hipMalloc(&g_out, 4096 * 256 * 4);
  for (int blocks : {1024, 4096}) {
    g_blocks = blocks;
    for (unsigned lg : {16u, 19u, 21u, 23u, 25u}) {   // 256 KB, 2 MB (one hash level), 8 MB, 32 MB, 128 MB
      g_n = 1u << lg;
      const double threads = (double)blocks * 256, it = 64;
      double t;
      t = timeit(launch_g<0, 8>, 3);  printf("blocks %d table %6.1f MB  random dword        ILP8 : %7.1f G loads/s\n", blocks, g_n * 4 / 1048576.0, threads * it * 8 / t / 1e6);
      t = timeit(launch_g<1, 4>, 3);  printf("blocks %d table %6.1f MB  pair 2x dword       ILP4 : %7.1f G dwords/s\n", blocks, g_n * 4 / 1048576.0, threads * it * 8 / t / 1e6);
      t = timeit(launch_g<2, 4>, 3);  printf("blocks %d table %6.1f MB  pair dwordx2        ILP4 : %7.1f G dwords/s\n", blocks, g_n * 4 / 1048576.0, threads * it * 8 / t / 1e6);
      t = timeit(launch_b<0>, 3);   printf("blocks %d table %6.1f MB  buffer dword aux 0 (default)  : %7.1f G loads/s\n", blocks, g_n * 4 / 1048576.0, threads * it * 8 / t / 1e6);
      t = timeit(launch_b<1>, 3);   printf("blocks %d table %6.1f MB  buffer dword aux 1 (sc0)      : %7.1f G loads/s\n", blocks, g_n * 4 / 1048576.0, threads * it * 8 / t / 1e6);
      t = timeit(launch_b<2>, 3);   printf("blocks %d table %6.1f MB  buffer dword aux 2 (nt)       : %7.1f G loads/s\n", blocks, g_n * 4 / 1048576.0, threads * it * 8 / t / 1e6);
      t = timeit(launch_b<3>, 3);   printf("blocks %d table %6.1f MB  buffer dword aux 3 (sc0 nt)   : %7.1f G loads/s\n", blocks, g_n * 4 / 1048576.0, threads * it * 8 / t / 1e6);
      t = timeit(launch_b<16>, 3);  printf("blocks %d table %6.1f MB  buffer dword aux 16 (sc1)     : %7.1f G loads/s\n", blocks, g_n * 4 / 1048576.0, threads * it * 8 / t / 1e6);
      t = timeit(launch_b<17>, 3);  printf("blocks %d table %6.1f MB  buffer dword aux 17 (sc0 sc1) : %7.1f G loads/s\n", blocks, g_n * 4 / 1048576.0, threads * it * 8 / t / 1e6);
      g_R = (int)floor(cbrt((double)g_n));
      t = timeit(launch_c<3>, 3);     printf("blocks %d table %6.1f MB  dense cell 8x dword (R=%d): %7.1f G dwords/s\n", blocks, g_n * 4 / 1048576.0, g_R, threads * it * 8 / t / 1e6);
      t = timeit(launch_c<4>, 3);     printf("blocks %d table %6.1f MB  dense cell 4x dwordx2     : %7.1f G dwords/s\n", blocks, g_n * 4 / 1048576.0, threads * it * 8 / t / 1e6);
    }
  }
  return 0;
}
