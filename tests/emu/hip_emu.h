// hip_emu.h -- TEST INFRASTRUCTURE ONLY (never linked into the product library).
//
// A tiny SIMT emulator that lets the *same* kernel sources under
// neuralsim_amd/csrc/ be compiled for the host (clang++ -I tests/emu first: nsim_prims.h of this directory) so that
// kernel logic (indexing, wave-level scans, MFMA fragment bookkeeping, packed
// segment handling) can be checked against the oracle in the CPU-only authoring
// container.  Every GPU thread is a fiber; a 64-lane wavefront executes in
// lock-step at every cross-lane primitive (shfl / ballot / mfma), a workgroup at
// every __syncthreads().  Blocks run one after another, so "atomics" are plain
// read-modify-writes.
//
// The MFMA emulation encodes only what the CDNA4 guide states:
//   * A operand of a 32x32xK instruction: lane l supplies row (l & 31),
//     B operand: lane l supplies column (l & 31); the K slots are indexed by
//     (l >> 5, element) *identically* for A and B (so any kernel that is
//     correct here is correct for the hardware's actual slot order);
//   * C/D layout: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5).
// Nothing in the product (neuralsim_amd/_lib.py) can load the emulator build.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <vector>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __shared__ static

struct dim3 {
  unsigned x, y, z;
  dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
typedef void* hipStream_t;
typedef int hipError_t;
#define hipSuccess 0
static inline hipError_t hipGetLastError() { return 0; }
static inline hipError_t hipGetDevice(int* d) { *d = 0; return 0; }
static inline const char* hipGetErrorString(hipError_t) { return "emu"; }
static inline hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t) {
  memset(p, v, n);
  return 0;
}

namespace emu {

struct Wave {
  int nlive = 0, arrived = 0;
  uint64_t gen = 0;
  bool alive[64];
  alignas(16) unsigned char xbuf[64][512];
};

struct Fiber {
  void* sp = nullptr;
  char* stack = nullptr;
  bool done = true;
  dim3 tid;
  int lane = 0;
  Wave* wave = nullptr;
};

struct State {
  Fiber* cur = nullptr;
  void* sched_sp = nullptr;
  dim3 bIdx, bDim, gDim;
  // block barrier
  int b_nlive = 0, b_arrived = 0;
  uint64_t b_gen = 0;
  uint64_t progress = 0;
  char* dyn_smem = nullptr;
  const std::function<void()>* body = nullptr;
};

State& st();
void yield();
void launch(dim3 grid, dim3 block, size_t shmem, const std::function<void()>& body);

inline void wave_barrier() {
  State& s = st();
  Wave& w = *s.cur->wave;
  uint64_t g = w.gen;
  s.progress++;
  if (++w.arrived >= w.nlive) {
    w.arrived = 0;
    w.gen++;
  } else {
    while (w.gen == g) yield();
  }
}

inline void block_barrier() {
  State& s = st();
  uint64_t g = s.b_gen;
  s.progress++;
  if (++s.b_arrived >= s.b_nlive) {
    s.b_arrived = 0;
    s.b_gen++;
  } else {
    while (s.b_gen == g) yield();
  }
}

inline int lane_id() { return st().cur->lane; }

template <class T>
inline T shfl(T v, int src) {
  static_assert(sizeof(T) <= 512, "xbuf too small");
  Fiber* f = st().cur;
  Wave& w = *f->wave;
  memcpy(w.xbuf[f->lane], &v, sizeof(T));
  wave_barrier();
  T r;
  memcpy(&r, w.xbuf[src & 63], sizeof(T));
  wave_barrier();
  return r;
}

inline unsigned long long ballot(int pred) {
  Fiber* f = st().cur;
  Wave& w = *f->wave;
  int p = pred ? 1 : 0;
  memcpy(w.xbuf[f->lane], &p, sizeof(int));
  wave_barrier();
  unsigned long long m = 0;
  for (int l = 0; l < 64; ++l) {
    if (!w.alive[l]) continue;
    int q;
    memcpy(&q, w.xbuf[l], sizeof(int));
    if (q) m |= (1ull << l);
  }
  wave_barrier();
  return m;
}

// D = A*B + C for one wave; KS = K slots held per lane (8 for 32x32x16 f16, 1 for 32x32x2 f32).
template <class TA, int KS>
inline void mfma32(const TA* a, const TA* b, const float* c, float* d) {
  struct Rec {
    TA a[KS], b[KS];
    float c[16];
  };
  static_assert(sizeof(Rec) <= 512, "xbuf too small");
  Fiber* f = st().cur;
  Wave& w = *f->wave;
  Rec rec;
  for (int e = 0; e < KS; ++e) { rec.a[e] = a[e]; rec.b[e] = b[e]; }
  for (int r = 0; r < 16; ++r) rec.c[r] = c[r];
  memcpy(w.xbuf[f->lane], &rec, sizeof(Rec));
  wave_barrier();
  const int lane = f->lane, j = lane & 31, hi = lane >> 5;
  for (int r = 0; r < 16; ++r) {
    const int i = (r & 3) + 8 * (r >> 2) + 4 * hi;
    float acc = rec.c[r];
    for (int h2 = 0; h2 < 2; ++h2) {
      Rec ra, rb;
      memcpy(&ra, w.xbuf[i + 32 * h2], sizeof(Rec));
      memcpy(&rb, w.xbuf[j + 32 * h2], sizeof(Rec));
      for (int e = 0; e < KS; ++e) acc += (float)ra.a[e] * (float)rb.b[e];
    }
    d[r] = acc;
  }
  wave_barrier();
}

}  // namespace emu

#define threadIdx (emu::st().cur->tid)
#define blockIdx (emu::st().bIdx)
#define blockDim (emu::st().bDim)
#define gridDim (emu::st().gDim)
#define __syncthreads() emu::block_barrier()

#define hipLaunchKernelGGL(kern, grid, block, shmem, stream, ...) \
  emu::launch((grid), (block), (shmem), [=]() { kern(__VA_ARGS__); })

// ---- atomics (blocks and fibers are serialised, so plain RMW is atomic) ----
template <class T>
static inline T atomicAdd(T* p, T v) { T o = *p; *p = o + v; return o; }
template <class T>
static inline T atomicMax(T* p, T v) { T o = *p; *p = o > v ? o : v; return o; }
template <class T>
static inline T atomicMin(T* p, T v) { T o = *p; *p = o < v ? o : v; return o; }
template <class T>
static inline T atomicOr(T* p, T v) { T o = *p; *p = o | v; return o; }
template <class T>
static inline T atomicExch(T* p, T v) { T o = *p; *p = v; return o; }

static inline int __popcll(unsigned long long x) { return __builtin_popcountll(x); }
static inline int __popc(unsigned x) { return __builtin_popcount(x); }
static inline float rsqrtf(float x) { return 1.0f / sqrtf(x); }
