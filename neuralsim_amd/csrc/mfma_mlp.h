// mfma_mlp.h -- building blocks of the fused tiny-MLP kernels (field.hip: NeuS SDF + radiance; nerf_field.hip: the
// NeRF++ distant model) on the gfx950 matrix cores.
//
// Activation-register convention (one wave = 32 points): lane (j = l&31, hi = l>>5) owns point j and the units
// U(m,r,hi) = 32m + (r&3) + 8(r>>2) + 4hi of every M-tile m -- the C/D fragment of v_mfma_f32_32x32x*.  Layers are
// computed transposed (Out^T = W . In^T, weights = A operand), so a lane's accumulator registers are its B fragment for
// the next layer.  Weight gradients contract over points and go through an LDS transpose ([unit][point]).
#pragma once
#include "nsim_common.h"

__device__ __forceinline__ int unit_of(int m, int r, int hi) { return 32 * m + (r & 3) + 8 * (r >> 2) + 4 * hi; }

// ------------------------------------------------------------------------------------- MFMA helpers
// (wave_sync_lds: nsim_prims.h)

__device__ __forceinline__ f32x16 zero16() {
  f32x16 z;
#pragma unroll
  for (int r = 0; r < 16; ++r) z[r] = 0.f;
  return z;
}

// exact power-of-two scale bringing the wave-wide max |v| to ~16 (1 when PREC==1 or all-zero)
template <int PREC, int N>
__device__ __forceinline__ float dyn_scale(const float (&v)[N]) {
  if constexpr (PREC == 1) {
    return 1.0f;
  } else {
    float m = 0.f;
#pragma unroll
    for (int i = 0; i < N; ++i) m = fmaxf(m, fabsf(v[i]));
    m = wave_max(m);
    if (!(m > 0.f) || !(m < 3.0e38f)) return 1.0f;
    uint32_t bits;
    memcpy(&bits, &m, 4);
    const int ex = (int)((bits >> 23) & 0xffu) - 126;  // m = f * 2^ex, f in [0.5,1)
    int k = 4 - ex;
    k = k > 60 ? 60 : (k < -60 ? -60 : k);
    const uint32_t sb = (uint32_t)(127 + k) << 23;
    float sc;
    memcpy(&sc, &sb, 4);
    return sc;
  }
}

// acc[mo] += W[32mo.., :] . In^T   with In given in activation-register order (NI M-tiles of 16 regs),
// multiplied by in_scale before the f16 conversion.  Caller multiplies the result by 1/in_scale.
template <int PREC, int MO, int NI>
__device__ __forceinline__ void contract(f32x16 (&acc)[MO], const char* wmat, const float (&in)[NI * 16],
                                         float in_scale) {
  const int lane = nsim_lane();
  if constexpr (PREC == 0) {
    const f16x8* A = reinterpret_cast<const f16x8*>(wmat);
#pragma unroll
    for (int s = 0; s < 2 * NI; ++s) {
      f16x8 b;
#pragma unroll
      for (int e = 0; e < 8; ++e) b[e] = (f16)(in[(s >> 1) * 16 + 8 * (s & 1) + e] * in_scale);
#pragma unroll
      for (int mo = 0; mo < MO; ++mo) acc[mo] = mfma_32x32x16_f16(A[(mo * 2 * NI + s) * 64 + lane], b, acc[mo]);
    }
  } else {
    const float* A = reinterpret_cast<const float*>(wmat);
#pragma unroll
    for (int mi = 0; mi < NI; ++mi) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float b = in[mi * 16 + r];
#pragma unroll
        for (int mo = 0; mo < MO; ++mo)
          acc[mo] = mfma_32x32x2_f32(A[((mo * NI + mi) * 16 + r) * 64 + lane], b, acc[mo]);
      }
    }
  }
}

// out[m*16+r] = (W . In^T)[unit(m,r,hi)][pt]
template <int PREC, int MO, int NI>
__device__ __forceinline__ void dense(float (&out)[MO * 16], const char* wmat, const float (&in)[NI * 16],
                                      bool dynamic) {
  f32x16 acc[MO];
#pragma unroll
  for (int mo = 0; mo < MO; ++mo) acc[mo] = zero16();
  const float sc = dynamic ? dyn_scale<PREC, NI * 16>(in) : 1.0f;
  contract<PREC, MO, NI>(acc, wmat, in, sc);
  const float inv = 1.0f / sc;
#pragma unroll
  for (int mo = 0; mo < MO; ++mo)
#pragma unroll
    for (int r = 0; r < 16; ++r) out[mo * 16 + r] = acc[mo][r] * inv;
}

// ---- LDS staging [unit][point] for the weight-gradient contractions (contract over the tile's 32 points)
template <int PREC>
struct StageT {
  typedef f16 T;
  static constexpr int PITCH = 40;
};
template <>
struct StageT<1> {
  typedef float T;
  static constexpr int PITCH = 33;
};

// bytes of the two per-wave staging arrays (A side + B side, 64 rows each)
template <int PREC>
__host__ __device__ constexpr int stage_bytes_per_wave() {
  return 2 * 64 * StageT<PREC>::PITCH * (int)sizeof(typename StageT<PREC>::T);
}

template <int PREC, int NM>
__device__ __forceinline__ void stage(void* st, const float (&v)[NM * 16], float scale) {
  typedef typename StageT<PREC>::T T;
  T* p = reinterpret_cast<T*>(st);
  const int lane = nsim_lane(), j = lane & 31, hi = lane >> 5;
#pragma unroll
  for (int m = 0; m < NM; ++m)
#pragma unroll
    for (int r = 0; r < 16; ++r) p[unit_of(m, r, hi) * StageT<PREC>::PITCH + j] = (T)(v[m * 16 + r] * scale);
}

// C[32mo + row][32no + col] = sum_pt A[32mo+row][pt] * B[32no+col][pt]
template <int PREC>
__device__ __forceinline__ f32x16 dw_tile(const void* stA, int mo, const void* stB, int no) {
  typedef typename StageT<PREC>::T T;
  constexpr int P = StageT<PREC>::PITCH;
  const T* a = reinterpret_cast<const T*>(stA);
  const T* b = reinterpret_cast<const T*>(stB);
  const int lane = nsim_lane(), i = lane & 31, hi = lane >> 5;
  f32x16 acc = zero16();
  if constexpr (PREC == 0) {
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const f16x8 av = *reinterpret_cast<const f16x8*>(a + (32 * mo + i) * P + 16 * s + 8 * hi);
      const f16x8 bv = *reinterpret_cast<const f16x8*>(b + (32 * no + i) * P + 16 * s + 8 * hi);
      acc = mfma_32x32x16_f16(av, bv, acc);
    }
  } else {
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const float av = a[(32 * mo + i) * P + 2 * q + hi];
      const float bv = b[(32 * no + i) * P + 2 * q + hi];
      acc = mfma_32x32x2_f32(av, bv, acc);
    }
  }
  return acc;
}

// accumulate a dW tile into an LDS accumulator: dst[(32mo+row)*ld + 32no + col].
// PRIV = the accumulator belongs to THIS wave alone -> plain read-add-write.  Measured on MI355X: ds_add_f32 costs
// ~800 cycles per wave instruction (the 192 LDS float atomics per 32-point tile were 80 % of the SDF-branch backward),
// a ds_read / v_add / ds_write triple a few tens.  PRIV = false keeps the shared accumulator + atomics (f32 test mode).
template <bool PRIV = false>
__device__ __forceinline__ void dw_flush(float* dst, int ld, int rows, int cols, int mo, int no, const f32x16& acc,
                                         float unscale) {
  const int lane = nsim_lane(), col = 32 * no + (lane & 31), hi = lane >> 5;
  if (col >= cols) return;
  if constexpr (PRIV) {
    float cur[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = 32 * mo + mfma_row(r, hi);
      cur[r] = row < rows ? dst[row * ld + col] : 0.f;
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = 32 * mo + mfma_row(r, hi);
      if (row < rows) dst[row * ld + col] = cur[r] + acc[r] * unscale;
    }
  } else {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = 32 * mo + mfma_row(r, hi);
      if (row < rows) atomicAdd(&dst[row * ld + col], acc[r] * unscale);
    }
  }
}

// dW[rows x cols] += A (NMA m-tiles) (x) B (NMB m-tiles) over the tile's points; db[rows] += rowsum(A)
template <int PREC, int NMA, int NMB, bool PRIV = false>
__device__ __forceinline__ void dw_product(void* stA, void* stB, const float (&A)[NMA * 16], const float (&B)[NMB * 16],
                                           float* dW, int ld, int rows, int cols, float* db) {
  const float sa = dyn_scale<PREC, NMA * 16>(A), sb = dyn_scale<PREC, NMB * 16>(B);
  wave_sync_lds();
  stage<PREC, NMA>(stA, A, sa);
  stage<PREC, NMB>(stB, B, sb);
  wave_sync_lds();
  const float un = 1.0f / (sa * sb);
#pragma unroll
  for (int mo = 0; mo < NMA; ++mo)
#pragma unroll
    for (int no = 0; no < NMB; ++no) {
      const f32x16 acc = dw_tile<PREC>(stA, mo, stB, no);
      dw_flush<PRIV>(dW, ld, rows, cols, mo, no, acc, un);
    }
  if (db) {
    typedef typename StageT<PREC>::T T;
    const T* a = reinterpret_cast<const T*>(stA);
    const int lane = nsim_lane();
    if (lane < NMA * 32 && lane < rows) {
      float s = 0.f;
      for (int j = 0; j < 32; ++j) s += (float)a[lane * StageT<PREC>::PITCH + j];
      if constexpr (PRIV) db[lane] = db[lane] + s / sa;
      else atomicAdd(&db[lane], s / sa);
    }
  }
}

// row sums of an activation (for vector-shaped gradients such as the SDF head weights)
template <int PREC, int NM, bool PRIV = false>
__device__ __forceinline__ void rowsum_acc(void* stA, const float (&A)[NM * 16], float* dst, int rows) {
  const float sa = dyn_scale<PREC, NM * 16>(A);
  wave_sync_lds();
  stage<PREC, NM>(stA, A, sa);
  wave_sync_lds();
  typedef typename StageT<PREC>::T T;
  const T* a = reinterpret_cast<const T*>(stA);
  const int lane = nsim_lane();
  if (lane < NM * 32 && lane < rows) {
    float s = 0.f;
    for (int j = 0; j < 32; ++j) s += (float)a[lane * StageT<PREC>::PITCH + j];
    if constexpr (PRIV) dst[lane] = dst[lane] + s / sa;
    else atomicAdd(&dst[lane], s / sa);
  }
}


// ===================================================================== workgroup-joint weight-gradient products
// The weight gradients contract over POINTS.  Instead of one 32-point product per wave and tile (with its accumulator
// read-add-written in LDS after every tile: 16 ds_read + 16 ds_write per lane and output tile, four private 25 KB
// accumulator copies -> one workgroup per CU), the NW waves of a workgroup stage their activations side by side
// ([unit][32 NW points]) and every wave owns a FIXED subset of the output tiles, which it keeps in MFMA accumulator
// registers for the whole launch: no read-modify-write at all, no accumulators in LDS, one flush per wave at the end.
// fp16 mode stages bf16 (v_mfma_f32_32x32x16_bf16, same rate as f16 on gfx950): the f32 exponent range makes the
// per-tile power-of-two re-scaling of the f16 operands unnecessary, which is what allows the accumulation to run
// ACROSS tiles; the operands keep 8 significant bits, the sum is f32.  f32 mode stages f32 (exact, validation).
#ifndef NSIM_STAGE_PAIRS
#define NSIM_STAGE_PAIRS 1      // jstage, bf16: v_cvt_pk_bf16_f32 converts two staged values per instruction
#endif
#define JOINT_WAVES 4
#define JOINT_PTS (32 * JOINT_WAVES)

template <int PREC>
struct JStageT {
  typedef bf16 T;
  static constexpr int PITCH = JOINT_PTS + 8;     // 272 B rows: 16-byte aligned, conflict-free ds_read_b128
};
template <>
struct JStageT<1> {
  typedef float T;
  static constexpr int PITCH = JOINT_PTS + 1;
};

template <int PREC>
__host__ __device__ constexpr int jstage_row_bytes() {
  return JStageT<PREC>::PITCH * (int)sizeof(typename JStageT<PREC>::T);
}

// write this wave's 32 points of an activation (NM m-tiles in activation-register order) into rows [0, 32 NM) of ``st``
// PAIRS (bf16): two values per conversion instruction (nsim_cvt2_bf16; MI355X: nsim_field_bwd_sdf 0.1216 -> 0.1135 ms on the bench
// step -- but the 17..32-level one-hidden-layer backward of the street step got 5 % slower, its register allocation moved 78 more
// values through AGPRs: that instantiation passes PAIRS = false)
template <int PREC, int NM, bool PAIRS = true>
__device__ __forceinline__ void jstage(void* st, const float (&v)[NM * 16], int wave) {
  typedef typename JStageT<PREC>::T T;
  T* p = reinterpret_cast<T*>(st);
  const int lane = nsim_lane(), j = lane & 31, hi = lane >> 5;
  if constexpr (PREC == 0 && NSIM_STAGE_PAIRS && PAIRS) {
#pragma unroll
    for (int m = 0; m < NM; ++m)
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        bf16 lo, hi16;
        nsim_cvt2_bf16(v[m * 16 + r], v[m * 16 + r + 1], lo, hi16);
        p[unit_of(m, r, hi) * JStageT<PREC>::PITCH + 32 * wave + j] = lo;
        p[unit_of(m, r + 1, hi) * JStageT<PREC>::PITCH + 32 * wave + j] = hi16;
      }
    return;
  }
#pragma unroll
  for (int m = 0; m < NM; ++m)
#pragma unroll
    for (int r = 0; r < 16; ++r) p[unit_of(m, r, hi) * JStageT<PREC>::PITCH + 32 * wave + j] = (T)v[m * 16 + r];
}

// the same for v * scale (a chain carried in scaled form is un-scaled where it is staged: no second copy of it in registers)
template <int PREC, int NM>
__device__ __forceinline__ void jstage_scaled(void* st, const float (&v)[NM * 16], float scale, int wave) {
  typedef typename JStageT<PREC>::T T;
  T* p = reinterpret_cast<T*>(st);
  const int lane = nsim_lane(), j = lane & 31, hi = lane >> 5;
  if constexpr (PREC == 0 && NSIM_STAGE_PAIRS) {
#pragma unroll
    for (int m = 0; m < NM; ++m)
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        bf16 lo, hi16;
        nsim_cvt2_bf16(v[m * 16 + r] * scale, v[m * 16 + r + 1] * scale, lo, hi16);
        p[unit_of(m, r, hi) * JStageT<PREC>::PITCH + 32 * wave + j] = lo;
        p[unit_of(m, r + 1, hi) * JStageT<PREC>::PITCH + 32 * wave + j] = hi16;
      }
    return;
  }
#pragma unroll
  for (int m = 0; m < NM; ++m)
#pragma unroll
    for (int r = 0; r < 16; ++r) p[unit_of(m, r, hi) * JStageT<PREC>::PITCH + 32 * wave + j] = (T)(v[m * 16 + r] * scale);
}

// acc += A[32 mo + row][:] . B[32 no + col][:]^T over the staged points [16 S0, 16 S1)  (default: all JOINT_PTS)
template <int PREC, int S0 = 0, int S1 = JOINT_PTS / 16>
__device__ __forceinline__ f32x16 jdw_tile(const void* stA, int mo, const void* stB, int no, f32x16 acc) {
  typedef typename JStageT<PREC>::T T;
  constexpr int P = JStageT<PREC>::PITCH;
  const T* a = reinterpret_cast<const T*>(stA);
  const T* b = reinterpret_cast<const T*>(stB);
  const int lane = nsim_lane(), i = lane & 31, hi = lane >> 5;
  if constexpr (PREC == 0) {
#pragma unroll
    for (int s = S0; s < S1; ++s) {
      const bf16x8 av = *reinterpret_cast<const bf16x8*>(a + (32 * mo + i) * P + 16 * s + 8 * hi);
      const bf16x8 bv = *reinterpret_cast<const bf16x8*>(b + (32 * no + i) * P + 16 * s + 8 * hi);
      acc = mfma_32x32x16_bf16(av, bv, acc);
    }
  } else {
#pragma unroll 8
    for (int q = 8 * S0; q < 8 * S1; ++q) {
      const float av = a[(32 * mo + i) * P + 2 * q + hi];
      const float bv = b[(32 * no + i) * P + 2 * q + hi];
      acc = mfma_32x32x2_f32(av, bv, acc);
    }
  }
  return acc;
}

// sum of row ``lane`` over THIS wave's 32 staged points (bias gradients; every wave keeps its own partial sum, so no
// wave becomes the straggler of the next barrier).  bf16: v_dot2c_f32_bf16 against (1, 1) adds two points per issue.
template <int PREC>
__device__ __forceinline__ float jrow_sum(const void* stA, int rows, int wave) {
  typedef typename JStageT<PREC>::T T;
  constexpr int P = JStageT<PREC>::PITCH;
  const T* a = reinterpret_cast<const T*>(stA);
  const int lane = nsim_lane();
  float s = 0.f;
  if (lane < rows) {
    if constexpr (PREC == 0) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const bf16x8 v = *reinterpret_cast<const bf16x8*>(a + lane * P + 32 * wave + 8 * q);
        s = nsim_bf16x8_sum(v, s);
      }
    } else {
      for (int q = 0; q < 32; ++q) s += a[lane * P + 32 * wave + q];
    }
  }
  return s;
}

// flush an accumulator tile with global atomics: dst[(32 mo + row) * ld + 32 no + col], rows < rows, cols < cols
__device__ __forceinline__ void jflush_tile(float* dst, int ld, int rows, int cols, int mo, int no, const f32x16& acc) {
  const int lane = nsim_lane(), col = 32 * no + (lane & 31), hi = lane >> 5;
  if (col >= cols) return;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = 32 * mo + mfma_row(r, hi);
    if (row < rows && acc[r] != 0.f) atomicAdd(&dst[row * ld + col], acc[r]);
  }
}

// Sum v0 / v1 over the runs of equal ``key`` among the 32 lanes of each wave half (keys arrive grouped: consecutive
// samples of a ray are neighbouring lanes); the run total is valid on the LAST lane of the run, for which the function
// returns true.  Used to issue ONE atomic per (ray, channel) and wave instead of one per sample: same-address atomics of
// one instruction are separate requests to the atomic unit (21 G requests/s chip-wide) and serialise in L2.
__device__ __forceinline__ bool halfwave_run_sum2(int64_t key, bool valid, float& v0, float& v1) {
  const int lane = nsim_lane();
  const int64_t k = valid ? key : (int64_t)-1 - lane;          // invalid lanes: runs of their own
  const int64_t pk = wave_shfl(k, lane - 1);
  const unsigned long long heads = wave_ballot((lane & 31) == 0 || pk != k);
  const unsigned long long below = heads & ((2ull << lane) - 1ull);
  const int run_start = 63 - __builtin_clzll(below);
#pragma unroll
  for (int d = 1; d < 32; d <<= 1) {
    const float o0 = wave_shfl(v0, lane - d), o1 = wave_shfl(v1, lane - d);
    if (lane - d >= run_start) {
      v0 += o0;
      v1 += o1;
    }
  }
  return valid && ((lane & 31) == 31 || ((heads >> (lane + 1)) & 1ull));
}
