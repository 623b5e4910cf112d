// pack_ops.hip -- segmented ("packed") tensor ops and the fused volume-integration kernels for gfx950.
//
// Replaces the native half of nr3d_lib.graphics.pack_ops / nr3d_lib.graphics.nerf as the reference calls
// them (app/renderers/single_volume_renderer.py:73-102, app/renderers/buffer_compose_renderer.py:644-723,
// app/loss/lidar.py:102-110).  Design: ONE 64-lane wavefront owns one pack (ray); lanes stride over the
// pack's samples (coalesced 256-B rows), segmented scans are wave-level shuffle scans with a scalar carry
// between 64-sample chunks.  Everything here is HBM-bound: 4-44 B per sample in, 4-28 B out.
#include "nsim_common.h"

#define PACK_WAVES_PER_BLOCK 4
#define PACK_BLOCK (64 * PACK_WAVES_PER_BLOCK)

__device__ __forceinline__ int64_t pack_wave_id() {
  return (int64_t)blockIdx.x * PACK_WAVES_PER_BLOCK + (threadIdx.x >> 6);
}
static inline dim3 pack_grid(int64_t P) { return dim3(nsim_blocks(P, PACK_WAVES_PER_BLOCK)); }

// ------------------------------------------------------------------------------ pack_infos_from_n
// One workgroup of 16 waves walks the counts in super-chunks of PI_THREADS * PI_PER = 8192.
#define PI_THREADS 1024
#define PI_PER 8
// Blocked arrangement: thread t owns PI_PER CONSECUTIVE counts of the super-chunk, so a super-chunk of 8192 counts
// needs one serial 8-element prefix per thread and ONE block scan (wave scan + 16 wave totals) instead of PI_PER of
// them -- the 8192-ray batch of the training step is exactly one super-chunk.
__global__ void __launch_bounds__(PI_THREADS) k_pack_infos_from_n(const int64_t* __restrict__ n, int64_t P,
                                                                    int64_t* __restrict__ pi,
                                                                    int64_t* __restrict__ total, int64_t cap,
                                                                    int64_t* notify, int64_t seq) {
  __shared__ int64_t wtot[PI_THREADS / 64];
  const int tid = threadIdx.x, lane = nsim_lane(), wave = tid >> 6;
  int64_t carry = 0;
  for (int64_t base = 0; base < P; base += (int64_t)PI_THREADS * PI_PER) {
    const int64_t i0 = base + (int64_t)tid * PI_PER;
    int64_t v[PI_PER];
    int64_t mine = 0;
#pragma unroll
    for (int k = 0; k < PI_PER; ++k) {
      v[k] = (i0 + k) < P ? n[i0 + k] : 0;
      mine += v[k];
    }
    const int64_t incl = wave_incl_sum(mine);
    if (lane == 63) wtot[wave] = incl;
    __syncthreads();
    int64_t before = 0, chunk = 0;
#pragma unroll
    for (int w = 0; w < PI_THREADS / 64; ++w) {
      const int64_t x = wtot[w];
      before += (w < wave) ? x : 0;
      chunk += x;
    }
    int64_t run = carry + before + incl - mine;
#pragma unroll
    for (int k = 0; k < PI_PER; ++k) {
      const int64_t i = i0 + k;
      if (i < P) {
        // cap >= 0: the caller sized its buffers speculatively; a pack that would end beyond cap is emptied and every
        // start stays <= cap, so that each consumer -- including those that append per-pack data at start + const * i
        // -- stays in bounds (the caller sees total > cap at its next sync and redoes the pass)
        const bool over = cap >= 0 && run + v[k] > cap;
        pi[2 * i] = (cap >= 0 && run > cap) ? cap : run;
        pi[2 * i + 1] = over ? 0 : v[k];
      }
      run += v[k];
    }
    carry += chunk;
    __syncthreads();
  }
  if (tid == 0 && total) total[0] = carry;
  if (tid == 0 && notify) {  // host-mapped words: the host spins on notify[1] == seq instead of a stream sync + copy
    nsim_store_system(notify, carry, false);
    nsim_store_system(notify + 1, seq, true);
  }
}

// ------------------------------------------------------------------------------------ live-ray ranks
// ``upsample_on_marched_only`` (ray_query_cfg.query_param): coarse and fine samples go only to the rays whose occupancy march
// found something.  From the march counts n [R] (one workgroup, the scan structure of k_pack_infos_from_n):
//   live_rank[r] = q  (>= 0)  r is the q-th live ray (n[r] > 0);   ~q (< 0)  r is dead, q live rays precede it
//   live_idx[q]  = r          the live rays in order (entries q >= R' are set to 0)
//   cnts[0] = R', cnts[1] = sum(n) + R' C (points of the first SDF query), cnts[2 + k] = R' nf[k] (points of draw k), cnts[6] = sum(n)
// and optionally (R', seq) to host-mapped ``notify`` words (see k_pack_infos_from_n).
struct LiveArgs {
  int C;
  int nf[4];
};
// pi (may be NULL): the pack infos of n as k_pack_infos_from_n writes them (cap, total, notify_total alike) -- the sampling
// pass needs both from the same counts, one launch instead of two single-workgroup scans back to back (9 us each).
__global__ void __launch_bounds__(PI_THREADS) k_live_rank(const int64_t* __restrict__ n, int64_t R, LiveArgs la,
                                                           int64_t* __restrict__ live_rank, int64_t* __restrict__ live_idx,
                                                           int64_t* __restrict__ cnts, int64_t* notify, int64_t seq,
                                                           int64_t* __restrict__ pi, int64_t* __restrict__ total, int64_t cap,
                                                           int64_t* notify_total, int64_t seq_total) {
  // two sums ride one pass: the sample counts (int64) and the live-ray count.  (Rounds 4-5 packed both into one int64 scan with
  // the live count in bits 44..63: 2^19 live rays reached the sign bit -- an 800x800 image queried in one chunk has 640 k rays.)
  __shared__ int64_t wtot[PI_THREADS / 64];
  __shared__ int wlive[PI_THREADS / 64];
  const int tid = threadIdx.x, lane = nsim_lane(), wave = tid >> 6;
  int64_t carry = 0, carry_live = 0;
  for (int64_t base = 0; base < R; base += (int64_t)PI_THREADS * PI_PER) {
    const int64_t i0 = base + (int64_t)tid * PI_PER;
    int64_t v[PI_PER];
    int64_t mine = 0;
    int mine_live = 0;      // (at most PI_THREADS * PI_PER per super-chunk: the 32-bit DPP scan; the running total is 64-bit)
#pragma unroll
    for (int k = 0; k < PI_PER; ++k) {
      v[k] = (i0 + k) < R ? n[i0 + k] : 0;
      mine += v[k];
      mine_live += v[k] > 0 ? 1 : 0;
    }
    const int64_t incl = wave_incl_sum(mine);
    const int incl_live = wave_incl_sum(mine_live);
    if (lane == 63) {
      wtot[wave] = incl;
      wlive[wave] = incl_live;
    }
    __syncthreads();
    int64_t before = 0, chunk = 0;
    int before_live = 0, chunk_live = 0;
#pragma unroll
    for (int w = 0; w < PI_THREADS / 64; ++w) {
      const int64_t x = wtot[w];
      const int y = wlive[w];
      before += (w < wave) ? x : 0;
      chunk += x;
      before_live += (w < wave) ? y : 0;
      chunk_live += y;
    }
    int64_t run = carry + before + incl - mine;
    int64_t q = carry_live + before_live + incl_live - mine_live;
#pragma unroll
    for (int k = 0; k < PI_PER; ++k) {
      const int64_t i = i0 + k;
      if (i < R) {
        const bool live = v[k] > 0;
        live_rank[i] = live ? q : ~q;
        if (live && live_idx) live_idx[q] = i;
        if (pi) {      // (k_pack_infos_from_n's rule for a speculative capacity)
          const bool over = cap >= 0 && run + v[k] > cap;
          pi[2 * i] = (cap >= 0 && run > cap) ? cap : run;
          pi[2 * i + 1] = over ? 0 : v[k];
        }
        q += live ? 1 : 0;
      }
      run += v[k];
    }
    carry += chunk;
    carry_live += chunk_live;
    __syncthreads();
  }
  const int64_t Rl = carry_live, M = carry;
  if (live_idx)
    for (int64_t i = Rl + tid; i < R; i += PI_THREADS) live_idx[i] = 0;
  if (tid == 0) {
    cnts[0] = Rl;
    cnts[1] = M + Rl * la.C;
#pragma unroll
    for (int k = 0; k < 4; ++k) cnts[2 + k] = Rl * la.nf[k];
    cnts[6] = M;
    cnts[7] = 0;
    if (total) total[0] = M;
    if (notify) {
      nsim_store_system(notify, Rl, false);
      nsim_store_system(notify + 1, seq, true);
    }
    if (notify_total) {
      nsim_store_system(notify_total, M, false);
      nsim_store_system(notify_total + 1, seq_total, true);
    }
  }
}

// ------------------------------------------------------------------------------------ packed_sum
__global__ void __launch_bounds__(PACK_BLOCK) k_packed_sum(const float* __restrict__ x, int C,
                                                             const int64_t* __restrict__ pi, int64_t P,
                                                             float* __restrict__ out) {
  const int64_t p = pack_wave_id();
  if (p >= P) return;
  const int lane = nsim_lane();
  const int64_t st = pi[2 * p], n = pi[2 * p + 1];
  for (int c = 0; c < C; ++c) {
    float acc = 0.f;
    for (int64_t i = lane; i < n; i += 64) acc += x[(st + i) * C + c];
    acc = wave_sum(acc);
    if (lane == 0) out[p * C + c] = acc;
  }
}

__global__ void __launch_bounds__(PACK_BLOCK) k_packed_binary(const float* __restrict__ x, int C,
                                                                const float* __restrict__ pp, int Cp,
                                                                const int64_t* __restrict__ pi, int64_t P,
                                                                int op, float* __restrict__ out) {
  const int64_t p = pack_wave_id();
  if (p >= P) return;
  const int lane = nsim_lane();
  const int64_t st = pi[2 * p], n = pi[2 * p + 1];
  const int64_t tot = n * C;
  for (int64_t j = lane; j < tot; j += 64) {
    const int c = (int)(j % C);
    const float b = pp[p * Cp + (Cp == 1 ? 0 : c)];
    const float a = x ? x[st * C + j] : 1.0f;
    float r;
    if (op == 0) r = a * b;
    else if (op == 1) r = a / b;
    else if (op == 2) r = a + b;
    else r = a - b;
    out[st * C + j] = r;
  }
}

__global__ void __launch_bounds__(PACK_BLOCK) k_packed_cmp(const float* __restrict__ x,
                                                             const float* __restrict__ pp,
                                                             const int64_t* __restrict__ pi, int64_t P, int op,
                                                             uint8_t* __restrict__ out) {
  const int64_t p = pack_wave_id();
  if (p >= P) return;
  const int lane = nsim_lane();
  const int64_t st = pi[2 * p], n = pi[2 * p + 1];
  const float b = pp[p];
  for (int64_t i = lane; i < n; i += 64) {
    const float a = x[st + i];
    bool r = (op == 0) ? (a >= b) : (op == 1) ? (a <= b) : (op == 2) ? (a < b) : (a > b);
    out[st + i] = r ? 1 : 0;
  }
}

__global__ void __launch_bounds__(PACK_BLOCK) k_packed_matmul3(const float* __restrict__ x,
                                                                 const float* __restrict__ rot,
                                                                 const int64_t* __restrict__ pi, int64_t P,
                                                                 int transpose, float* __restrict__ out) {
  const int64_t p = pack_wave_id();
  if (p >= P) return;
  const int lane = nsim_lane();
  const int64_t st = pi[2 * p], n = pi[2 * p + 1];
  float r[9];
  for (int k = 0; k < 9; ++k) r[k] = rot[p * 9 + k];
  for (int64_t i = lane; i < n; i += 64) {
    const float a = x[(st + i) * 3 + 0], b = x[(st + i) * 3 + 1], c = x[(st + i) * 3 + 2];
    float o0, o1, o2;
    if (!transpose) {  // (rot * x[None,:]).sum(-1): broadcast-multiply-sum, no reduced-precision matmul
      o0 = r[0] * a + r[1] * b + r[2] * c;
      o1 = r[3] * a + r[4] * b + r[5] * c;
      o2 = r[6] * a + r[7] * b + r[8] * c;
    } else {
      o0 = r[0] * a + r[3] * b + r[6] * c;
      o1 = r[1] * a + r[4] * b + r[7] * c;
      o2 = r[2] * a + r[5] * b + r[8] * c;
    }
    out[(st + i) * 3 + 0] = o0;
    out[(st + i) * 3 + 1] = o1;
    out[(st + i) * 3 + 2] = o2;
  }
}

__global__ void __launch_bounds__(PACK_BLOCK) k_interleave_linstep(const int64_t* __restrict__ start,
                                                                     const int64_t* __restrict__ pi,
                                                                     int64_t P, int64_t step,
                                                                     int64_t* __restrict__ out) {
  const int64_t p = pack_wave_id();
  if (p >= P) return;
  const int lane = nsim_lane();
  const int64_t st = pi[2 * p], n = pi[2 * p + 1], s0 = start[p];
  for (int64_t i = lane; i < n; i += 64) out[st + i] = s0 + i * step;
}

// ------------------------------------------------------------------------------------ packed_sort
// Stable sort inside one pack: rank(i) = #{j : x_j < x_i  or (x_j == x_i and j < i)}.
// The packs of this path are CONCATENATIONS OF SORTED RUNS -- the depths of one ray as collected object by object
// (buffer_compose_renderer.py:648-695: street samples, the samples of every vehicle the ray crosses, the distant shells) -- so
// the pack is staged in LDS, its run heads (x_i < x_{i-1}) are found with ballots, and
//   * one run: the pack is in order already (most rays: background + distant shells) -> identity;
//   * <= PSORT_RUNS runs: rank(i) = offset in its own run + sum over the other runs of a binary search (elements <= x_i of
//     the runs before it, < x_i of the runs after it: the same stable order) -- O(n runs log n) instead of O(n^2);
//   * more runs (unstructured input): every lane ranks up to four elements per sweep over the staged pack, O(n^2 / 64).
// Packs beyond the staging capacity take the direct O(n^2) path on global memory.
// Round 4, multi-object step: 0.77 ms -> see profiles (2.6 M samples in 16 k packs).
#define PSORT_CAP 1024
#define PSORT_RUNS 16
__device__ __forceinline__ int lds_upper_bound(const float* a, int n, float v) {      // #elements <= v
  int lo = 0, hi = n;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (a[mid] <= v) lo = mid + 1; else hi = mid;
  }
  return lo;
}
__device__ __forceinline__ int lds_lower_bound(const float* a, int n, float v) {      // #elements < v
  int lo = 0, hi = n;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (a[mid] < v) lo = mid + 1; else hi = mid;
  }
  return lo;
}
// The staged sort: stable ranks of the nn values sx[0 .. nn) (nn <= PSORT_CAP, one wave), emit(i, rank, x_i) once per element.
template <class Emit>
__device__ __forceinline__ void psort_staged(const float* sx, int* srun, int nn, int lane, Emit emit) {
  int nrun = 0;
  for (int i0 = 0; i0 < nn; i0 += 64) {
    const int i = i0 + lane;
    const bool head = i < nn && (i == 0 || sx[i] < sx[i - 1]);
    const unsigned long long m = wave_ballot(head);
    if (head) {
      const int pos = nrun + __popcll(m & ((1ull << lane) - 1ull));
      if (pos < PSORT_RUNS) srun[pos] = i;
    }
    nrun += __popcll(m);
  }
  if (nrun <= 1) {
    for (int i = lane; i < nn; i += 64) emit(i, i, sx[i]);
    return;
  }
  if (nrun <= PSORT_RUNS) {
    if (lane == 0) srun[nrun] = nn;
    wave_sync_lds();
    for (int i = lane; i < nn; i += 64) {
      const float xi = sx[i];
      int rank = 0;
      for (int r = 0; r < nrun; ++r) {
        const int b = srun[r], e = srun[r + 1];
        if (e <= i) rank += lds_upper_bound(&sx[b], e - b, xi);
        else if (b > i) rank += lds_lower_bound(&sx[b], e - b, xi);
        else rank += i - b;
      }
      emit(i, rank, xi);
    }
    return;
  }
  for (int i0 = 0; i0 < nn; i0 += 256) {      // four elements per lane and sweep
    float xi[4];
    int ii[4], rank[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      ii[k] = i0 + lane + 64 * k;
      xi[k] = ii[k] < nn ? sx[ii[k]] : 0.f;
      rank[k] = 0;
    }
    for (int j = 0; j < nn; ++j) {
      const float xj = sx[j];
#pragma unroll
      for (int k = 0; k < 4; ++k) rank[k] += (xj < xi[k] || (xj == xi[k] && j < ii[k])) ? 1 : 0;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (ii[k] < nn) emit(ii[k], rank[k], xi[k]);
  }
}

__global__ void __launch_bounds__(PACK_BLOCK) k_packed_sort(const float* __restrict__ x,
                                                              const int64_t* __restrict__ pi, int64_t P,
                                                              float* __restrict__ sorted,
                                                              int64_t* __restrict__ indices) {
  __shared__ float sx[PACK_WAVES_PER_BLOCK][PSORT_CAP];
  __shared__ int srun[PACK_WAVES_PER_BLOCK][PSORT_RUNS + 1];
  const int64_t p = pack_wave_id();
  if (p >= P) return;
  const int lane = nsim_lane();
  const int w = (int)(threadIdx.x >> 6);
  const int64_t st = pi[2 * p], n = pi[2 * p + 1];
  if (n <= PSORT_CAP) {
    for (int64_t i = lane; i < n; i += 64) sx[w][i] = x[st + i];
    wave_sync_lds();
    psort_staged(sx[w], srun[w], (int)n, lane, [&](int i, int rank, float xi) {
      sorted[st + rank] = xi;
      indices[st + rank] = st + i;
    });
    return;
  }
  for (int64_t i = lane; i < n; i += 64) {
    const float xi = x[st + i];
    int64_t rank = 0;
    for (int64_t j = 0; j < n; ++j) {
      const float xj = x[st + j];
      rank += (xj < xi || (xj == xi && j < i)) ? 1 : 0;
    }
    sorted[st + rank] = xi;
    indices[st + rank] = st + i;
  }
}

// ------------------------------------------------------------------------------------ compose: collect + sort in one launch
// BufferComposeRenderer's collect and sort steps (code_multi/app/renderers/buffer_compose_renderer.py:648-695) as ONE kernel:
// the reference copies every object's per-ray packs behind a running per-ray cursor into one buffer (interleave_linstep +
// index_put per object and attribute), sorts every ray's pack by depth (packed_sort) and permutes every attribute.  Here a
// wave owns a ray: it finds the ray's pack in every source (the sources' ray lists are ascending: binary search, one lane per
// source), stages the depths source after source -- the reference's cursor order -- and ranks them with the staged sort above,
// so the order (ties included) is the reference's; it writes the sorted depths and, per source sample, its FINAL position
// ``dst`` -- every attribute then needs one index_put per source straight into sorted order, and an object's weights in the
// context of the whole scene are vw[dst] (no unsorted buffer, no permutation, no inverse permutation, no per-source host
// read of a sample count).
#define COMPOSE_MAX_SRC 64      // one lane per source
struct ComposeSrcDev {
  const float* t;
  const int64_t *ric, *pi;
  int64_t P;
  int64_t* dst;
};
struct ComposeArgs {
  int K;
  ComposeSrcDev src[COMPOSE_MAX_SRC];
  const int64_t* total_pi;      // [N, 2] (start, count) of every ray in the merged buffer
  int64_t N;
  float* t_sorted;
};

__global__ void __launch_bounds__(PACK_BLOCK) k_compose_collect_sort(ComposeArgs a) {
  __shared__ float sx[PACK_WAVES_PER_BLOCK][PSORT_CAP];
  __shared__ int srun[PACK_WAVES_PER_BLOCK][PSORT_RUNS + 1];
  __shared__ int s_off[PACK_WAVES_PER_BLOCK][COMPOSE_MAX_SRC + 1];      // offsets of the sources in the ray's unsorted list
  __shared__ int64_t s_stg[PACK_WAVES_PER_BLOCK][COMPOSE_MAX_SRC];      // start of the ray's pack in each source
  const int64_t r = pack_wave_id();
  if (r >= a.N) return;
  const int lane = nsim_lane();
  const int w = (int)(threadIdx.x >> 6);
  const int64_t st = a.total_pi[2 * r], n = a.total_pi[2 * r + 1];
  if (n <= 0) return;
  // lane k < K: the pack of ray r in source k (start, count), or count 0
  int64_t my_st = 0;
  int my_n = 0;
  if (lane < a.K) {
    const ComposeSrcDev& sk = a.src[lane];
    int64_t lo = 0, hi = sk.P;
    while (lo < hi) {
      const int64_t mid = (lo + hi) >> 1;
      if (sk.ric[mid] < r) lo = mid + 1; else hi = mid;
    }
    if (lo < sk.P && sk.ric[lo] == r) {
      my_st = sk.pi[2 * lo];
      my_n = (int)sk.pi[2 * lo + 1];
    }
  }
  const int incl = wave_incl_sum(my_n);
  s_off[w][lane] = incl - my_n;
  if (lane == 63) s_off[w][64] = incl;
  s_stg[w][lane] = my_st;
  wave_sync_lds();
  const int K = a.K;
  // unsorted element j -> its source's dst entry / its depth (sources without a pack on this ray have empty ranges)
  auto source_of = [&](int j) -> int {
    int lo = 0, hi = K - 1;      // the last k with s_off[k] <= j
    while (lo < hi) {
      const int mid = (lo + hi + 1) >> 1;
      if (s_off[w][mid] <= j) lo = mid; else hi = mid - 1;
    }
    return lo;
  };
  if (n <= PSORT_CAP) {
    for (int k = 0; k < K; ++k) {
      const int ok = s_off[w][k], nk = s_off[w][k + 1] - ok;
      const float* tk = a.src[k].t + s_stg[w][k];
      for (int i = lane; i < nk; i += 64) sx[w][ok + i] = tk[i];
    }
    wave_sync_lds();
    psort_staged(sx[w], srun[w], (int)n, lane, [&](int i, int rank, float xi) {
      a.t_sorted[st + rank] = xi;
      const int k = source_of(i);
      a.src[k].dst[s_stg[w][k] + (i - s_off[w][k])] = st + rank;
    });
    return;
  }
  // a ray with more samples than the staging area holds: ranks straight from the sources (O(n^2) reads through the caches)
  for (int64_t i = lane; i < n; i += 64) {
    const int ki = source_of((int)i);
    const float xi = a.src[ki].t[s_stg[w][ki] + (i - s_off[w][ki])];
    int64_t rank = 0;
    for (int k = 0; k < K; ++k) {
      const int ok = s_off[w][k], nk = s_off[w][k + 1] - ok;
      const float* tk = a.src[k].t + s_stg[w][k];
      for (int jj = 0; jj < nk; ++jj) {
        const float xj = tk[jj];
        const int64_t j = ok + jj;
        rank += (xj < xi || (xj == xi && j < i)) ? 1 : 0;
      }
    }
    a.t_sorted[st + rank] = xi;
    a.src[ki].dst[s_stg[w][ki] + (i - s_off[w][ki])] = st + rank;
  }
}

// lower_bound / upper_bound on a sorted global array segment
__device__ __forceinline__ int64_t seg_lower_bound(const float* a, int64_t n, float v) {  // #elements < v
  int64_t lo = 0, hi = n;
  while (lo < hi) {
    const int64_t mid = (lo + hi) >> 1;
    if (a[mid] < v) lo = mid + 1; else hi = mid;
  }
  return lo;
}
__device__ __forceinline__ int64_t seg_upper_bound(const float* a, int64_t n, float v) {  // #elements <= v
  int64_t lo = 0, hi = n;
  while (lo < hi) {
    const int64_t mid = (lo + hi) >> 1;
    if (a[mid] <= v) lo = mid + 1; else hi = mid;
  }
  return lo;
}

// merge_two_packs_sorted: one wave per pack of a and per pack of b (two launches share this kernel).
// For a pack of `self` on union slot u, every element's position in the merged pack is its own index plus
// the number of `other` elements that precede it (strictly smaller; ties: a first).
__global__ void __launch_bounds__(PACK_BLOCK) k_merge_two_packs(
    const float* __restrict__ vs, const int64_t* __restrict__ pis, const int64_t* __restrict__ slot_s,
    int64_t Ps, const float* __restrict__ vo, const int64_t* __restrict__ pio,
    const int64_t* __restrict__ slot_o, int64_t Po, const int64_t* __restrict__ pi_out, int self_is_a,
    int64_t* __restrict__ pidx_s) {
  const int64_t p = pack_wave_id();
  if (p >= Ps) return;
  const int lane = nsim_lane();
  const int64_t st = pis[2 * p], n = pis[2 * p + 1], u = slot_s[p];
  // find the other side's pack on the same union slot (slot_o ascending & unique)
  int64_t lo = 0, hi = Po;
  while (lo < hi) {
    const int64_t mid = (lo + hi) >> 1;
    if (slot_o[mid] < u) lo = mid + 1; else hi = mid;
  }
  const bool has_other = (lo < Po) && (slot_o[lo] == u);
  const int64_t ost = has_other ? pio[2 * lo] : 0, on = has_other ? pio[2 * lo + 1] : 0;
  const int64_t out_st = pi_out[2 * u];
  for (int64_t i = lane; i < n; i += 64) {
    const float v = vs[st + i];
    const int64_t before = self_is_a ? seg_lower_bound(vo + ost, on, v) : seg_upper_bound(vo + ost, on, v);
    pidx_s[st + i] = out_st + i + before;
  }
}

// ------------------------------------------------------------------------------- alpha -> vw
// chunked exclusive product scan of f = 1 - alpha + eps with a scalar carry
__device__ __forceinline__ void vw_chunk(float a, bool valid, float eps, float& carry, float& T) {
  const int lane = nsim_lane();
  const float f = valid ? (1.0f - a + eps) : 1.0f;
  const float incl = wave_incl_prod(f);
  float excl = wave_shfl(incl, lane - 1);
  if (lane == 0) excl = 1.0f;
  T = carry * excl;
  carry = carry * wave_shfl(incl, 63);
}

__global__ void __launch_bounds__(PACK_BLOCK) k_alpha_to_vw_fwd(const float* __restrict__ alpha,
                                                                  const int64_t* __restrict__ pi, int64_t P,
                                                                  float* __restrict__ vw,
                                                                  float* __restrict__ trans) {
  const int64_t p = pack_wave_id();
  if (p >= P) return;
  const int lane = nsim_lane();
  const int64_t st = pi[2 * p], n = pi[2 * p + 1];
  float carry = 1.0f;
  for (int64_t base = 0; base < n; base += 64) {
    const int64_t i = base + lane;
    const bool valid = i < n;
    const float a = valid ? alpha[st + i] : 0.f;
    float T;
    vw_chunk(a, valid, 1e-10f, carry, T);
    if (valid) {
      vw[st + i] = a * T;
      if (trans) trans[st + i] = T;
    }
  }
}

// d alpha_i = dvw_i * T_i - (sum_{k>i} dvw_k vw_k) / (1 - alpha_i + eps), chunks walked back to front
__global__ void __launch_bounds__(PACK_BLOCK) k_alpha_to_vw_bwd(
    const float* __restrict__ alpha, const float* __restrict__ trans, const float* __restrict__ vw,
    const float* __restrict__ dvw, const int64_t* __restrict__ pi, int64_t P, float* __restrict__ dalpha) {
  const int64_t p = pack_wave_id();
  if (p >= P) return;
  const int lane = nsim_lane();
  const int64_t st = pi[2 * p], n = pi[2 * p + 1];
  float carry = 0.f;  // sum of g over all later chunks
  const int64_t nchunks = (n + 63) / 64;
  for (int64_t c = nchunks - 1; c >= 0; --c) {
    // lanes hold the chunk BACK TO FRONT, so the inclusive scan over lanes is a true suffix sum: the sum over the later
    // samples is accumulated from small terms upwards.  (total - prefix cancels catastrophically behind an opaque
    // sample, and the result is divided by 1 - alpha + 1e-10.)
    const int64_t i = c * 64 + (63 - lane);
    const bool valid = i < n;
    const float g = valid ? dvw[st + i] * vw[st + i] : 0.f;
    const float incl = wave_incl_sum(g);
    const float tot = wave_shfl(incl, 63);
    float excl = wave_shfl(incl, lane - 1);
    if (lane == 0) excl = 0.f;
    const float suffix = carry + excl;
    if (valid) {
      const float a = alpha[st + i];
      dalpha[st + i] = dvw[st + i] * trans[st + i] - suffix / (1.0f - a + 1e-10f);
    }
    carry += tot;
  }
}

// ------------------------------------------------------------------------- fused compositing
__device__ __forceinline__ float nsim_sigmoid(float x) { return 1.0f / (1.0f + expf(-x)); }

__device__ __forceinline__ float neus_inv_s(const float* ln_inv_s, float factor, float forward_inv_s) {
  return forward_inv_s > 0.f ? forward_inv_s : expf(ln_inv_s[0] * factor);
}

// FROM_SDF: the NeuS opacities are computed here from ``sdf`` (the arithmetic of k_neus_alpha_fwd) and written to
// ``alpha_out`` for the backward -- one launch for sdf -> alpha -> vw -> images.
template <bool FROM_SDF>
__global__ void __launch_bounds__(PACK_BLOCK) k_composite_fwd(
    const float* __restrict__ alpha, const float* __restrict__ t, const float* __restrict__ rgb,
    const float* __restrict__ nrm, const int64_t* __restrict__ pi, int64_t P, int normalized_depth,
    float* __restrict__ vw, float* __restrict__ trans, float* __restrict__ mask, float* __restrict__ depth,
    float* __restrict__ rgb_out, float* __restrict__ nrm_out, const int64_t* __restrict__ out_idx,
    const float* __restrict__ sdf, const float* __restrict__ ln_inv_s, float factor, float forward_inv_s,
    float* __restrict__ alpha_out) {
  const int64_t p = pack_wave_id();
  if (p >= P) return;
  const int lane = nsim_lane();
  const int64_t st = pi[2 * p], n = pi[2 * p + 1];
  const int64_t q = out_idx ? out_idx[p] : p;      // row of the per-ray outputs (scatter to all-rays images)
  const float s = FROM_SDF ? neus_inv_s(ln_inv_s, factor, forward_inv_s) : 0.f;
  float carry = 1.0f;
  float am = 0.f, ad = 0.f, ar[3] = {0.f, 0.f, 0.f}, an[3] = {0.f, 0.f, 0.f};
  for (int64_t base = 0; base < n; base += 64) {
    const int64_t i = base + lane;
    const bool valid = i < n;
    float a = 0.f;
    if (FROM_SDF) {
      if (i + 1 < n) {
        const float c0 = nsim_sigmoid(sdf[st + i] * s), c1 = nsim_sigmoid(sdf[st + i + 1] * s);
        a = (c0 - c1 + 1e-5f) / (c0 + 1e-5f);
        a = fminf(fmaxf(a, 0.f), 1.f);
      }
      if (valid) alpha_out[st + i] = a;
    } else {
      a = valid ? alpha[st + i] : 0.f;
    }
    float T;
    vw_chunk(a, valid, 1e-10f, carry, T);
    if (valid) {
      const float w = a * T;
      vw[st + i] = w;
      trans[st + i] = T;
      am += w;
      ad += w * t[st + i];
      if (rgb) {
        ar[0] += w * rgb[(st + i) * 3 + 0];
        ar[1] += w * rgb[(st + i) * 3 + 1];
        ar[2] += w * rgb[(st + i) * 3 + 2];
      }
      if (nrm) {
        an[0] += w * nrm[(st + i) * 3 + 0];
        an[1] += w * nrm[(st + i) * 3 + 1];
        an[2] += w * nrm[(st + i) * 3 + 2];
      }
    }
  }
  am = wave_sum(am);
  ad = wave_sum(ad);
  if (rgb) for (int c = 0; c < 3; ++c) ar[c] = wave_sum(ar[c]);
  if (nrm) for (int c = 0; c < 3; ++c) an[c] = wave_sum(an[c]);
  if (lane == 0) {
    mask[q] = am;
    depth[q] = normalized_depth ? ad / (am + 1e-10f) : ad;
    if (rgb) for (int c = 0; c < 3; ++c) rgb_out[q * 3 + c] = ar[c];
    if (nrm) for (int c = 0; c < 3; ++c) nrm_out[q * 3 + c] = an[c];
  }
}

__global__ void __launch_bounds__(PACK_BLOCK) k_composite_bwd(
    const float* __restrict__ alpha, const float* __restrict__ trans, const float* __restrict__ vw,
    const float* __restrict__ t, const float* __restrict__ rgb, const float* __restrict__ nrm,
    const int64_t* __restrict__ pi, int64_t P, int normalized_depth, const float* __restrict__ mask,
    const float* __restrict__ depth, const float* __restrict__ dmask, const float* __restrict__ ddepth,
    const float* __restrict__ drgb_out, const float* __restrict__ dnrm_out,
    const float* __restrict__ dvw_ext, float* __restrict__ dalpha, float* __restrict__ drgb,
    float* __restrict__ dnrm, const int64_t* __restrict__ out_idx) {
  const int64_t p = pack_wave_id();
  if (p >= P) return;
  const int lane = nsim_lane();
  const int64_t st = pi[2 * p], n = pi[2 * p + 1];
  const int64_t q = out_idx ? out_idx[p] : p;
  float gm = dmask ? dmask[q] : 0.f;
  float gd = ddepth ? ddepth[q] : 0.f;
  if (normalized_depth) {
    // depth = D / (m + eps): dD = gd / (m+eps) ; dm += -gd * depth / (m+eps)
    const float den = mask[q] + 1e-10f;
    gm += -gd * depth[q] / den;
    gd = gd / den;
  }
  float gr[3] = {0.f, 0.f, 0.f}, gn[3] = {0.f, 0.f, 0.f};
  if (rgb && drgb_out) for (int c = 0; c < 3; ++c) gr[c] = drgb_out[q * 3 + c];
  if (nrm && dnrm_out) for (int c = 0; c < 3; ++c) gn[c] = dnrm_out[q * 3 + c];
  float carry = 0.f;
  const int64_t nchunks = (n + 63) / 64;
  for (int64_t c = nchunks - 1; c >= 0; --c) {
    const int64_t i = c * 64 + (63 - lane);      // back to front: see k_alpha_to_vw_bwd
    const bool valid = i < n;
    float gvw = 0.f, w = 0.f;
    if (valid) {
      w = vw[st + i];
      gvw = gm + gd * t[st + i];
      if (dvw_ext) gvw += dvw_ext[st + i];
      if (rgb) {
        const float r0 = rgb[(st + i) * 3 + 0], r1 = rgb[(st + i) * 3 + 1], r2 = rgb[(st + i) * 3 + 2];
        gvw += gr[0] * r0 + gr[1] * r1 + gr[2] * r2;
        if (drgb) {
          drgb[(st + i) * 3 + 0] = w * gr[0];
          drgb[(st + i) * 3 + 1] = w * gr[1];
          drgb[(st + i) * 3 + 2] = w * gr[2];
        }
      }
      if (nrm) {
        const float n0 = nrm[(st + i) * 3 + 0], n1 = nrm[(st + i) * 3 + 1], n2 = nrm[(st + i) * 3 + 2];
        gvw += gn[0] * n0 + gn[1] * n1 + gn[2] * n2;
        if (dnrm) {
          dnrm[(st + i) * 3 + 0] = w * gn[0];
          dnrm[(st + i) * 3 + 1] = w * gn[1];
          dnrm[(st + i) * 3 + 2] = w * gn[2];
        }
      }
    }
    const float g = gvw * w;
    const float incl = wave_incl_sum(g);
    const float tot = wave_shfl(incl, 63);
    float excl = wave_shfl(incl, lane - 1);
    if (lane == 0) excl = 0.f;
    const float suffix = carry + excl;
    if (valid) {
      const float a = alpha[st + i];
      dalpha[st + i] = gvw * trans[st + i] - suffix / (1.0f - a + 1e-10f);
    }
    carry += tot;
  }
}

// ------------------------------------------------------------------------------- NeuS sdf -> alpha
__global__ void __launch_bounds__(PACK_BLOCK) k_neus_alpha_fwd(const float* __restrict__ sdf,
                                                                 const int64_t* __restrict__ pi, int64_t P,
                                                                 const float* __restrict__ ln_inv_s,
                                                                 float factor, float forward_inv_s,
                                                                 float* __restrict__ alpha) {
  const int64_t p = pack_wave_id();
  if (p >= P) return;
  const int lane = nsim_lane();
  const int64_t st = pi[2 * p], n = pi[2 * p + 1];
  const float s = neus_inv_s(ln_inv_s, factor, forward_inv_s);
  for (int64_t i = lane; i < n; i += 64) {
    float a = 0.f;
    if (i + 1 < n) {
      const float c0 = nsim_sigmoid(sdf[st + i] * s), c1 = nsim_sigmoid(sdf[st + i + 1] * s);
      a = (c0 - c1 + 1e-5f) / (c0 + 1e-5f);
      a = fminf(fmaxf(a, 0.f), 1.f);
    }
    alpha[st + i] = a;
  }
}

__global__ void __launch_bounds__(PACK_BLOCK) k_neus_alpha_bwd(
    const float* __restrict__ sdf, const float* __restrict__ dalpha, const int64_t* __restrict__ pi, int64_t P,
    const float* __restrict__ ln_inv_s, float factor, float forward_inv_s, float* __restrict__ dsdf,
    float* __restrict__ d_ln_inv_s) {
  // a capped grid walks the packs wave by wave; d(ln_inv_s) is reduced per block before the single-address atomic
  // (one atomic per pack serialises at L2: 8192 rays cost ~0.1 ms)
  __shared__ float red[PACK_WAVES_PER_BLOCK];
  const int lane = nsim_lane();
  const float s = neus_inv_s(ln_inv_s, factor, forward_inv_s);
  const int64_t nw = (int64_t)gridDim.x * PACK_WAVES_PER_BLOCK;
  float ds_acc = 0.f;
  for (int64_t p = pack_wave_id(); p < P; p += nw) {
  const int64_t st = pi[2 * p], n = pi[2 * p + 1];
  for (int64_t base = 0; base < n; base += 64) {
    const int64_t i = base + lane;
    float g = 0.f;
    if (i < n) {
      const float x0 = sdf[st + i];
      const float c0 = nsim_sigmoid(x0 * s);
      // as the left end of interval i
      if (i + 1 < n) {
        const float x1 = sdf[st + i + 1];
        const float c1 = nsim_sigmoid(x1 * s);
        const float raw = (c0 - c1 + 1e-5f) / (c0 + 1e-5f);
        if (raw >= 0.f && raw <= 1.f) {
          const float ga = dalpha[st + i];
          const float den = c0 + 1e-5f;
          const float da_dc0 = (c1) / (den * den);
          const float da_dc1 = -1.0f / den;
          g += ga * da_dc0 * s * c0 * (1.f - c0);
          ds_acc += ga * (da_dc0 * x0 * c0 * (1.f - c0) + da_dc1 * x1 * c1 * (1.f - c1));
        }
      }
      // as the right end of interval i-1
      if (i >= 1) {
        const float xm = sdf[st + i - 1];
        const float cm = nsim_sigmoid(xm * s);
        const float raw = (cm - c0 + 1e-5f) / (cm + 1e-5f);
        if (raw >= 0.f && raw <= 1.f) {
          const float ga = dalpha[st + i - 1];
          g += ga * (-1.0f / (cm + 1e-5f)) * s * c0 * (1.f - c0);
        }
      }
      dsdf[st + i] = g;
    }
  }
  }
  if (d_ln_inv_s && forward_inv_s <= 0.f) {
    ds_acc = wave_sum(ds_acc);
    if (lane == 0) red[threadIdx.x >> 6] = ds_acc;
    __syncthreads();
    if (threadIdx.x == 0) {
      float tot = 0.f;
      for (int w = 0; w < PACK_WAVES_PER_BLOCK; ++w) tot += red[w];
      if (tot != 0.f) atomicAdd(d_ln_inv_s, tot * s * factor);
    }
  }
}

// ------------------------------------------------------------------------------- the training step's render head
// ONE launch for the differentiable tail of the fused training step (round 4; four launches before: nsim_neus_composite_fwd,
// nsim_train_loss_head, nsim_composite_bwd, nsim_neus_alpha_bwd -- 8 + 8 + 7 + 13 us under rocprofv3, each waiting for its
// predecessor's drain): per hit ray one wave runs  sdf -> alpha -> visibility weights -> images  (the arithmetic of
// k_composite_fwd<true>), forms that pixel's photometric-mse gradient, walks back through the compositing (k_composite_bwd)
// and the sdf -> alpha map (k_neus_alpha_bwd); the samples of a ray are handed between the passes through the per-sample
// buffers the backward kernels further down read anyway (alpha / vw / trans / dalpha), ordered by wave-level fences.  The
// remaining workgroups of the SAME launch do the element-wise part of the loss head: sum gt^2 over the rays OUTSIDE the packs (a
// ray that hits nothing renders black: its residual is gt) and the eikonal terms on the S render samples and the M free points with
// their gradients.  acc[0] = mse, acc[1] = eikonal(render samples), acc[2] = eikonal(free points), as nsim_train_loss_head.
struct RenderHeadArgs {
  const float *sdf, *ln_inv_s, *t, *rgb, *nab, *gt;
  const int64_t *pi, *out_idx;
  int64_t P, N, S, M;
  float factor, forward_inv_s, w_eik;
  int normalized_depth, ray_blocks;
  float *alpha, *vw, *trans, *mask, *depth, *rgb_out, *nrm_out, *acc, *dalpha, *dsdf, *drgb, *dnab, *dln;
};

__global__ void __launch_bounds__(PACK_BLOCK) k_render_head(RenderHeadArgs a) {
  __shared__ float red[4][PACK_WAVES_PER_BLOCK];
  const int lane = nsim_lane(), wave = (int)(threadIdx.x >> 6);
  float r0 = 0.f, r1 = 0.f, r2 = 0.f, r3 = 0.f;      // block partial sums: mse, eik render, eik free, d ln_inv_s
  const float inv_n = 1.0f / (float)(3 * a.N);
  if ((int)blockIdx.x < a.ray_blocks) {
    const int64_t p = (int64_t)blockIdx.x * PACK_WAVES_PER_BLOCK + wave;
    if (p < a.P) {
      const int64_t st = a.pi[2 * p], n = a.pi[2 * p + 1];
      const int64_t q = a.out_idx ? a.out_idx[p] : p;
      const float s = neus_inv_s(a.ln_inv_s, a.factor, a.forward_inv_s);
      // ---- forward: sdf -> alpha -> vw -> images
      float carry = 1.0f, am = 0.f, ad = 0.f, ar[3] = {0.f, 0.f, 0.f}, an[3] = {0.f, 0.f, 0.f};
      for (int64_t base = 0; base < n; base += 64) {
        const int64_t i = base + lane;
        const bool valid = i < n;
        float al = 0.f;
        if (i + 1 < n) {
          const float c0 = nsim_sigmoid(a.sdf[st + i] * s), c1 = nsim_sigmoid(a.sdf[st + i + 1] * s);
          al = (c0 - c1 + 1e-5f) / (c0 + 1e-5f);
          al = fminf(fmaxf(al, 0.f), 1.f);
        }
        if (valid) a.alpha[st + i] = al;
        float T;
        vw_chunk(al, valid, 1e-10f, carry, T);
        if (valid) {
          const float w = al * T;
          a.vw[st + i] = w;
          a.trans[st + i] = T;
          am += w;
          ad += w * a.t[st + i];
#pragma unroll
          for (int c = 0; c < 3; ++c) {
            ar[c] += w * a.rgb[(st + i) * 3 + c];
            an[c] += w * a.nab[(st + i) * 3 + c];
          }
        }
      }
      am = wave_sum(am);
      ad = wave_sum(ad);
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        ar[c] = wave_sum(ar[c]);
        an[c] = wave_sum(an[c]);
      }
      // ---- this pixel's photometric term (pred - gt)^2; the element-wise workgroups add gt^2 for the rays OUTSIDE the packs
      // (round 4 added gt^2 for every ray there and e^2 - g^2 here: two O(1) sums cancelling to an O(mse) result, ~1e-6 of
      // order-dependent noise on the reported mse -- ADVICE r4)
      float gr[3];
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const float g = a.gt[q * 3 + c], e = ar[c] - g;
        gr[c] = inv_n * 2.0f * e;
        if (lane == 0) r0 += e * e;
      }
      if (lane == 0) {
        a.mask[q] = am;
        a.depth[q] = a.normalized_depth ? ad / (am + 1e-10f) : ad;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          a.rgb_out[q * 3 + c] = ar[c];
          a.nrm_out[q * 3 + c] = an[c];
        }
      }
      nsim_wave_fence();
      // ---- backward of the compositing: d alpha, d rgb (back to front: see k_alpha_to_vw_bwd)
      float bcarry = 0.f;
      const int64_t nchunks = (n + 63) / 64;
      for (int64_t c = nchunks - 1; c >= 0; --c) {
        const int64_t i = c * 64 + (63 - lane);
        const bool valid = i < n;
        float gvw = 0.f, w = 0.f;
        if (valid) {
          w = a.vw[st + i];
          const float x0 = a.rgb[(st + i) * 3 + 0], x1 = a.rgb[(st + i) * 3 + 1], x2 = a.rgb[(st + i) * 3 + 2];
          gvw = gr[0] * x0 + gr[1] * x1 + gr[2] * x2;
          a.drgb[(st + i) * 3 + 0] = w * gr[0];
          a.drgb[(st + i) * 3 + 1] = w * gr[1];
          a.drgb[(st + i) * 3 + 2] = w * gr[2];
        }
        const float g = gvw * w;
        const float incl = wave_incl_sum(g);
        const float tot = wave_shfl(incl, 63);
        float excl = wave_shfl(incl, lane - 1);
        if (lane == 0) excl = 0.f;
        const float suffix = bcarry + excl;
        if (valid) a.dalpha[st + i] = gvw * a.trans[st + i] - suffix / (1.0f - a.alpha[st + i] + 1e-10f);
        bcarry += tot;
      }
      nsim_wave_fence();
      // ---- backward of sdf -> alpha
      float ds_acc = 0.f;
      for (int64_t base = 0; base < n; base += 64) {
        const int64_t i = base + lane;
        if (i < n) {
          float g = 0.f;
          const float x0 = a.sdf[st + i];
          const float c0 = nsim_sigmoid(x0 * s);
          if (i + 1 < n) {
            const float x1 = a.sdf[st + i + 1];
            const float c1 = nsim_sigmoid(x1 * s);
            const float raw = (c0 - c1 + 1e-5f) / (c0 + 1e-5f);
            if (raw >= 0.f && raw <= 1.f) {
              const float ga = a.dalpha[st + i];
              const float den = c0 + 1e-5f;
              const float da_dc0 = (c1) / (den * den);
              const float da_dc1 = -1.0f / den;
              g += ga * da_dc0 * s * c0 * (1.f - c0);
              ds_acc += ga * (da_dc0 * x0 * c0 * (1.f - c0) + da_dc1 * x1 * c1 * (1.f - c1));
            }
          }
          if (i >= 1) {
            const float xm = a.sdf[st + i - 1];
            const float cm = nsim_sigmoid(xm * s);
            const float raw = (cm - c0 + 1e-5f) / (cm + 1e-5f);
            if (raw >= 0.f && raw <= 1.f) g += a.dalpha[st + i - 1] * (-1.0f / (cm + 1e-5f)) * s * c0 * (1.f - c0);
          }
          a.dsdf[st + i] = g;
        }
      }
      if (a.dln && a.forward_inv_s <= 0.f) r3 = wave_sum(ds_acc) * s * a.factor;
    }
  } else {
    // ---- element-wise part: sum gt^2 over all rays, eikonal terms + gradients on the S + M samples
    const int64_t eb = (int64_t)blockIdx.x - a.ray_blocks, neb = (int64_t)gridDim.x - a.ray_blocks;
    const int64_t St = a.S + a.M, n_img = 3 * a.N, top = n_img > St ? n_img : St;
    const float inv_S = 1.0f / (float)(a.S > 0 ? a.S : 1), inv_M = 1.0f / (float)(a.M > 0 ? a.M : 1);
    float a0 = 0.f, a1 = 0.f, a2 = 0.f;
    for (int64_t i = eb * PACK_BLOCK + threadIdx.x; i < top; i += neb * PACK_BLOCK) {
      if (i < n_img) {
        // a ray outside the packs renders black: its residual is gt.  Rows of the packs: out_idx [P] ASCENDING (the hit-ray
        // compaction's order), or the first P rays
        const int64_t ray = i / 3;
        bool in_pack;
        if (a.out_idx) {
          int64_t lo = 0, hi = a.P;
          while (lo < hi) {
            const int64_t mid = (lo + hi) >> 1;
            if (a.out_idx[mid] < ray) lo = mid + 1; else hi = mid;
          }
          in_pack = lo < a.P && a.out_idx[lo] == ray;
        } else {
          in_pack = ray < a.P;
        }
        if (!in_pack) {
          const float g = a.gt[i];
          a0 += g * g;
        }
      }
      if (i < St) {
        const float x = a.nab[3 * i], y = a.nab[3 * i + 1], z = a.nab[3 * i + 2];
        const float nrm = sqrtf(x * x + y * y + z * z);
        const float e = nrm - 1.0f;
        const bool ren = i < a.S;
        if (ren) a1 += e * e;
        else a2 += e * e;
        const float k = nrm > 0.f ? a.w_eik * (ren ? inv_S : inv_M) * 2.0f * (nrm - 1.0f) / nrm : 0.f;
        a.dnab[3 * i] = k * x;
        a.dnab[3 * i + 1] = k * y;
        a.dnab[3 * i + 2] = k * z;
      }
    }
    r0 = wave_sum(a0);
    r1 = wave_sum(a1) * inv_S;
    r2 = wave_sum(a2) * inv_M;
  }
  if (lane == 0) {
    red[0][wave] = r0;
    red[1][wave] = r1;
    red[2][wave] = r2;
    red[3][wave] = r3;
  }
  __syncthreads();
  if (threadIdx.x < 4) {
    float tot = 0.f;
    for (int w = 0; w < PACK_WAVES_PER_BLOCK; ++w) tot += red[threadIdx.x][w];
    if (tot != 0.f) {
      if (threadIdx.x == 0) atomicAdd(a.acc, tot * inv_n);
      else if (threadIdx.x < 3) atomicAdd(a.acc + threadIdx.x, tot);
      else if (a.dln) atomicAdd(a.dln, tot);
    }
  }
}

// ================================================================================== C ABI
extern "C" {

int nsim_pack_infos_from_n(const int64_t* n, int64_t P, int64_t* pack_infos, int64_t* total, int64_t cap,
                           void* stream) {
  if (P < 0) return 2;
  hipLaunchKernelGGL(k_pack_infos_from_n, dim3(1), dim3(PI_THREADS), 0, (hipStream_t)stream, n, P, pack_infos, total, cap,
                     (int64_t*)nullptr, (int64_t)0);
  NSIM_CHECK_LAUNCH();
  return 0;
}

int nsim_pack_infos_from_n_notify(const int64_t* n, int64_t P, int64_t* pack_infos, int64_t* total, int64_t cap,
                                  int64_t* notify, int64_t seq, void* stream) {
  if (P <= 0) return 2;
  if (!notify) return 4;
  hipLaunchKernelGGL(k_pack_infos_from_n, dim3(1), dim3(PI_THREADS), 0, (hipStream_t)stream, n, P, pack_infos, total, cap,
                     notify, seq);
  NSIM_CHECK_LAUNCH();
  return 0;
}

int nsim_live_rank(const int64_t* n, int64_t R, int C, int nf0, int nf1, int nf2, int nf3, int64_t* live_rank, int64_t* live_idx,
                   int64_t* cnts, int64_t* notify, int64_t seq, int64_t* pack_infos, int64_t* total, int64_t cap,
                   int64_t* notify_total, int64_t seq_total, void* stream) {
  if (R <= 0) return 2;
  if (!n || !live_rank || !cnts || C < 0) return 4;
  LiveArgs la;
  la.C = C;
  la.nf[0] = nf0, la.nf[1] = nf1, la.nf[2] = nf2, la.nf[3] = nf3;
  hipLaunchKernelGGL(k_live_rank, dim3(1), dim3(PI_THREADS), 0, (hipStream_t)stream, n, R, la, live_rank, live_idx, cnts, notify,
                     seq, pack_infos, total, cap, notify_total, seq_total);
  NSIM_CHECK_LAUNCH();
  return 0;
}

int nsim_packed_sum(const float* x, int C, const int64_t* pack_infos, int64_t P, float* out, void* stream) {
  if (P <= 0) return 0;
  if (C < 1) return 3;
  hipLaunchKernelGGL(k_packed_sum, pack_grid(P), dim3(PACK_BLOCK), 0, (hipStream_t)stream, x, C, pack_infos, P, out);
  NSIM_CHECK_LAUNCH();
  return 0;
}

int nsim_packed_binary(const float* x, int C, const float* per_pack, int Cp, const int64_t* pack_infos,
                       int64_t P, int op, float* out, void* stream) {
  if (P <= 0) return 0;
  if (C < 1 || (Cp != 1 && Cp != C) || op < 0 || op > 3) return 3;
  hipLaunchKernelGGL(k_packed_binary, pack_grid(P), dim3(PACK_BLOCK), 0, (hipStream_t)stream, x, C, per_pack, Cp,
                     pack_infos, P, op, out);
  NSIM_CHECK_LAUNCH();
  return 0;
}

int nsim_packed_cmp(const float* x, const float* per_pack, const int64_t* pack_infos, int64_t P, int op,
                    uint8_t* out, void* stream) {
  if (P <= 0) return 0;
  if (op < 0 || op > 3) return 3;
  hipLaunchKernelGGL(k_packed_cmp, pack_grid(P), dim3(PACK_BLOCK), 0, (hipStream_t)stream, x, per_pack, pack_infos, P, op, out);
  NSIM_CHECK_LAUNCH();
  return 0;
}

int nsim_packed_matmul3(const float* x, const float* rot, const int64_t* pack_infos, int64_t P, int transpose,
                        float* out, void* stream) {
  if (P <= 0) return 0;
  hipLaunchKernelGGL(k_packed_matmul3, pack_grid(P), dim3(PACK_BLOCK), 0, (hipStream_t)stream, x, rot, pack_infos, P,
                     transpose, out);
  NSIM_CHECK_LAUNCH();
  return 0;
}

int nsim_packed_sort(const float* x, const int64_t* pack_infos, int64_t P, float* sorted, int64_t* indices,
                     void* stream) {
  if (P <= 0) return 0;
  hipLaunchKernelGGL(k_packed_sort, pack_grid(P), dim3(PACK_BLOCK), 0, (hipStream_t)stream, x, pack_infos, P, sorted, indices);
  NSIM_CHECK_LAUNCH();
  return 0;
}

int nsim_compose_collect_sort(const NsimComposeSrc* src, int32_t K, const int64_t* total_pack_infos, int64_t N,
                              float* t_sorted, void* stream) {
  if (K < 0 || K > COMPOSE_MAX_SRC) return 37;
  if (N <= 0 || K == 0) return 0;
  if (!src || !total_pack_infos || !t_sorted) return 2;
  ComposeArgs a;
  memset(&a, 0, sizeof(a));
  a.K = K;
  for (int k = 0; k < K; ++k) {
    if (src[k].P < 0 || (src[k].P > 0 && !(src[k].t && src[k].rays_inds && src[k].pack_infos && src[k].dst))) return 2;
    a.src[k].t = src[k].t;
    a.src[k].ric = src[k].rays_inds;
    a.src[k].pi = src[k].pack_infos;
    a.src[k].P = src[k].P;
    a.src[k].dst = src[k].dst;
  }
  a.total_pi = total_pack_infos;
  a.N = N;
  a.t_sorted = t_sorted;
  hipLaunchKernelGGL(k_compose_collect_sort, pack_grid(N), dim3(PACK_BLOCK), 0, (hipStream_t)stream, a);
  NSIM_CHECK_LAUNCH();
  return 0;
}

int nsim_interleave_linstep(const int64_t* start, const int64_t* pack_infos, int64_t P, int64_t step,
                            int64_t* out, void* stream) {
  if (P <= 0) return 0;
  hipLaunchKernelGGL(k_interleave_linstep, pack_grid(P), dim3(PACK_BLOCK), 0, (hipStream_t)stream, start, pack_infos, P, step, out);
  NSIM_CHECK_LAUNCH();
  return 0;
}

int nsim_merge_two_packs(const float* va, const int64_t* pia, const int64_t* slot_a, int64_t Pa,
                         const float* vb, const int64_t* pib, const int64_t* slot_b, int64_t Pb,
                         const int64_t* pack_infos_out, int64_t U, int64_t* pidx_a, int64_t* pidx_b,
                         void* stream) {
  (void)U;
  if (Pa > 0) {
    hipLaunchKernelGGL(k_merge_two_packs, pack_grid(Pa), dim3(PACK_BLOCK), 0, (hipStream_t)stream, va, pia, slot_a, Pa,
                       vb, pib, slot_b, Pb, pack_infos_out, 1, pidx_a);
    NSIM_CHECK_LAUNCH();
  }
  if (Pb > 0) {
    hipLaunchKernelGGL(k_merge_two_packs, pack_grid(Pb), dim3(PACK_BLOCK), 0, (hipStream_t)stream, vb, pib, slot_b, Pb,
                       va, pia, slot_a, Pa, pack_infos_out, 0, pidx_b);
    NSIM_CHECK_LAUNCH();
  }
  return 0;
}

int nsim_alpha_to_vw_fwd(const float* alpha, const int64_t* pack_infos, int64_t P, float* vw, float* trans,
                         void* stream) {
  if (P <= 0) return 0;
  hipLaunchKernelGGL(k_alpha_to_vw_fwd, pack_grid(P), dim3(PACK_BLOCK), 0, (hipStream_t)stream, alpha, pack_infos, P, vw, trans);
  NSIM_CHECK_LAUNCH();
  return 0;
}

int nsim_alpha_to_vw_bwd(const float* alpha, const float* trans, const float* vw, const float* dvw,
                         const int64_t* pack_infos, int64_t P, float* dalpha, void* stream) {
  if (P <= 0) return 0;
  hipLaunchKernelGGL(k_alpha_to_vw_bwd, pack_grid(P), dim3(PACK_BLOCK), 0, (hipStream_t)stream, alpha, trans, vw, dvw,
                     pack_infos, P, dalpha);
  NSIM_CHECK_LAUNCH();
  return 0;
}

int nsim_composite_fwd(const float* alpha, const float* t, const float* rgb, const float* nrm,
                       const int64_t* pack_infos, int64_t P, int normalized_depth, float* vw, float* trans,
                       float* mask, float* depth, float* rgb_out, float* nrm_out, const int64_t* out_idx,
                       void* stream) {
  if (P <= 0) return 0;
  if (!vw || !trans || !mask || !depth) return 4;
  hipLaunchKernelGGL((k_composite_fwd<false>), pack_grid(P), dim3(PACK_BLOCK), 0, (hipStream_t)stream, alpha, t, rgb, nrm,
                     pack_infos, P, normalized_depth, vw, trans, mask, depth, rgb_out, nrm_out, out_idx, nullptr, nullptr,
                     0.f, 0.f, nullptr);
  NSIM_CHECK_LAUNCH();
  return 0;
}

int nsim_neus_composite_fwd(const float* sdf, const float* ln_inv_s, float ln_inv_s_factor, float forward_inv_s,
                            const float* t, const float* rgb, const float* nrm, const int64_t* pack_infos, int64_t P,
                            int normalized_depth, float* alpha, float* vw, float* trans, float* mask, float* depth,
                            float* rgb_out, float* nrm_out, const int64_t* out_idx, void* stream) {
  if (P <= 0) return 0;
  if (!sdf || !alpha || !vw || !trans || !mask || !depth) return 4;
  if (!ln_inv_s && !(forward_inv_s > 0.f)) return 4;
  hipLaunchKernelGGL((k_composite_fwd<true>), pack_grid(P), dim3(PACK_BLOCK), 0, (hipStream_t)stream, nullptr, t, rgb, nrm,
                     pack_infos, P, normalized_depth, vw, trans, mask, depth, rgb_out, nrm_out, out_idx, sdf, ln_inv_s,
                     ln_inv_s_factor, forward_inv_s, alpha);
  NSIM_CHECK_LAUNCH();
  return 0;
}

int nsim_render_head(const float* sdf, const float* ln_inv_s, float ln_inv_s_factor, float forward_inv_s, const float* t,
                     const float* rgb, const float* nablas, const int64_t* pack_infos, int64_t P, int normalized_depth,
                     const float* gt, int64_t N, int64_t S, int64_t M, float w_eikonal, const int64_t* out_idx, float* alpha,
                     float* vw, float* trans, float* mask, float* depth, float* rgb_out, float* nrm_out, float* acc,
                     float* dalpha, float* dsdf, float* drgb, float* dnablas, float* d_ln_inv_s, void* stream) {
  if (P < 0 || N <= 0 || S < 0 || M < 0) return 2;
  if (!gt || !acc || (S + M > 0 && (!nablas || !dnablas))) return 4;
  if (P > 0 && (!sdf || !t || !rgb || !pack_infos || !alpha || !vw || !trans || !mask || !depth || !rgb_out || !nrm_out ||
                !dalpha || !dsdf || !drgb)) return 4;
  if (!ln_inv_s && !(forward_inv_s > 0.f)) return 4;
  RenderHeadArgs a;
  a.sdf = sdf; a.ln_inv_s = ln_inv_s; a.t = t; a.rgb = rgb; a.nab = nablas; a.gt = gt;
  a.pi = pack_infos; a.out_idx = out_idx;
  a.P = P; a.N = N; a.S = S; a.M = M;
  a.factor = ln_inv_s_factor; a.forward_inv_s = forward_inv_s; a.w_eik = w_eikonal;
  a.normalized_depth = normalized_depth;
  a.ray_blocks = (int)nsim_blocks(P > 0 ? P : 1, PACK_WAVES_PER_BLOCK);
  if (P == 0) a.ray_blocks = 0;
  a.alpha = alpha; a.vw = vw; a.trans = trans; a.mask = mask; a.depth = depth; a.rgb_out = rgb_out; a.nrm_out = nrm_out;
  a.acc = acc; a.dalpha = dalpha; a.dsdf = dsdf; a.drgb = drgb; a.dnab = dnablas; a.dln = d_ln_inv_s;
  const int64_t top = 3 * N > S + M ? 3 * N : S + M;
  int64_t eb = nsim_blocks(top, PACK_BLOCK);
  if (eb > 128) eb = 128;                      // three single-address atomics per workgroup
  hipLaunchKernelGGL(k_render_head, dim3((unsigned)(a.ray_blocks + eb)), dim3(PACK_BLOCK), 0, (hipStream_t)stream, a);
  NSIM_CHECK_LAUNCH();
  return 0;
}

int nsim_composite_bwd(const float* alpha, const float* trans, const float* vw, const float* t,
                       const float* rgb, const float* nrm, const int64_t* pack_infos, int64_t P,
                       int normalized_depth, const float* mask, const float* depth, const float* dmask,
                       const float* ddepth, const float* drgb_out, const float* dnrm_out,
                       const float* dvw_ext, float* dalpha, float* drgb, float* dnrm, const int64_t* out_idx,
                       void* stream) {
  if (P <= 0) return 0;
  hipLaunchKernelGGL(k_composite_bwd, pack_grid(P), dim3(PACK_BLOCK), 0, (hipStream_t)stream, alpha, trans, vw, t, rgb,
                     nrm, pack_infos, P, normalized_depth, mask, depth, dmask, ddepth, drgb_out, dnrm_out, dvw_ext,
                     dalpha, drgb, dnrm, out_idx);
  NSIM_CHECK_LAUNCH();
  return 0;
}

int nsim_neus_alpha_fwd(const float* sdf, const int64_t* pack_infos, int64_t P, const float* ln_inv_s,
                        float ln_inv_s_factor, float forward_inv_s, float* alpha, void* stream) {
  if (P <= 0) return 0;
  hipLaunchKernelGGL(k_neus_alpha_fwd, pack_grid(P), dim3(PACK_BLOCK), 0, (hipStream_t)stream, sdf, pack_infos, P,
                     ln_inv_s, ln_inv_s_factor, forward_inv_s, alpha);
  NSIM_CHECK_LAUNCH();
  return 0;
}

int nsim_neus_alpha_bwd(const float* sdf, const float* dalpha, const int64_t* pack_infos, int64_t P,
                        const float* ln_inv_s, float ln_inv_s_factor, float forward_inv_s, float* dsdf,
                        float* d_ln_inv_s, void* stream) {
  if (P <= 0) return 0;
  dim3 grid = pack_grid(P);
  if (grid.x > 512) grid.x = 512;
  hipLaunchKernelGGL(k_neus_alpha_bwd, grid, dim3(PACK_BLOCK), 0, (hipStream_t)stream, sdf, dalpha, pack_infos, P,
                     ln_inv_s, ln_inv_s_factor, forward_inv_s, dsdf, d_ln_inv_s);
  NSIM_CHECK_LAUNCH();
  return 0;
}

}  // extern "C"
