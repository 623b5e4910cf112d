// atomic_bench3.hip -- micro-benchmark: does a packed 2 x bf16 atomic add (global_atomic_pk_add_bf16, one 32-bit slot =
// both features of a LoTD vertex) retire faster than the two f32 atomics the gradient scatter issues per vertex today?
// The scatter's bound is the number of 64-byte-line REQUESTS (tools/atomic_bench2.hip: ~21 G/s chip-wide, lanes of one
// instruction hitting adjacent dwords share a request), so the question is whether halving the bytes per vertex halves
// anything the atomic unit counts.  Random vertices of a 2^19-entry table (one hashed level), one vertex per lane:
//   f32 pair : two instructions, lanes write {2v, 2v+1}            (8 B per vertex: the current layout)
//   f32 quad : four lanes per vertex pair, 16 B per request          (the scatter's quad-transposed issue)
//   pk bf16  : one instruction, lane writes the packed dword v       (4 B per vertex)
// Development aid.  Build: hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ void k(float* tab32, bf16x2* tab16, unsigned mask, int per_thread, unsigned seed) {
  unsigned tid = blockIdx.x * blockDim.x + threadIdx.x;
  unsigned x = (MODE == 1 ? (tid >> 2) : tid) * 2654435761u + seed;
  for (int i = 0; i < per_thread; ++i) {
    x = x * 1664525u + 1013904223u;
    unsigned v = (x >> 8) & mask;
    if (MODE == 0) {
      atomicAdd(tab32 + 2 * v, 1.0f);
      atomicAdd(tab32 + 2 * v + 1, 1.0f);
    } else if (MODE == 1) {      // 4 lanes serve the x-adjacent vertex pair (v & ~1, v | 1): one 16-byte request
      atomicAdd(tab32 + 2 * (v & ~1u) + (tid & 3), 1.0f);
    } else {
      const bf16x2 one = {(__bf16)1.0f, (__bf16)1.0f};
      __builtin_amdgcn_global_atomic_fadd_v2bf16((bf16x2*)(tab16 + v), one);
    }
  }
}

template <int MODE>
void run(const char* name, float* t32, bf16x2* t16, unsigned mask) {
  const int blocks = 2048, threads = 256, per_thread = 128, nrep = 3;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, t32, t16, mask, per_thread, 1u);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int r = 0; r < nrep; ++r) hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, t32, t16, mask, per_thread, 7u + r);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double lanes = (double)blocks * threads * per_thread * nrep;
  const double vertices = MODE == 1 ? lanes / 2.0 : lanes;      // quad mode: 4 lanes cover 2 vertices
  printf("%-28s table 2^%d vertices : %7.2f G vertex-updates/s  (%6.2f ms)\n", name, 32 - __builtin_clz(mask), vertices / ms * 1e-6, ms);
}

int main() {
  float* t32; bf16x2* t16;
  hipMalloc(&t32, sizeof(float) * 2 * (1u << 22));
  hipMalloc(&t16, sizeof(bf16x2) * (1u << 22));
  hipMemset(t32, 0, sizeof(float) * 2 * (1u << 22));
  hipMemset(t16, 0, sizeof(bf16x2) * (1u << 22));
  for (unsigned mask : {(1u << 19) - 1, (1u << 22) - 1}) {
    run<0>("f32 pair (2 instr/vertex)", t32, t16, mask);
    run<1>("f32 quad-transposed", t32, t16, mask);
    run<2>("packed bf16x2 (1 instr)", t32, t16, mask);
  }
  return 0;
}
