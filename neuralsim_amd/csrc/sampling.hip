// sampling.hip -- ray generation, AABB ray test, occupancy-grid marching and NeuS up-sampling for gfx950.
//
// Replaces the native half of Camera.get_selected_rays (app/resources/observers/cameras.py:281-310),
// AABBSpace.ray_test, OccGridAccel.ray_march / OccGridEma (accel_cfg, march_cfg in
// code_single/configs/object_centric/lotd_neus.dtu.230814.yaml:140-173) and the multi-stage inverse-CDF
// up-sampler of NeusRendererMixin (num_fine, upsample_inv_s, upsample_use_estimate_alpha).
//
// Design: one wavefront per ray.  Marching tests 64 lattice steps per wave instruction against the
// occupancy BITFIELD (64^3 bits = 32 KiB: L1/LDS resident), ballots the hits and emits them already
// sorted; up-sampling is a chunked wave scan (alpha -> transmittance -> cdf) followed by one binary search
// per new sample.  All arithmetic that decides sample membership is written mul-then-add (the library is
// built with -ffp-contract=off) so the sample set is bit-identical to the oracle's.
#include "nsim_common.h"
#include "occ_dev.h"

#define SMP_WAVES_PER_BLOCK 4
#define SMP_BLOCK (64 * SMP_WAVES_PER_BLOCK)
#define SMP_LDS_FLOATS 512   // per-wave LDS window of the up-sampling / merge kernels (longer rays: global memory)

__device__ __forceinline__ int64_t smp_wave_id() {
  return (int64_t)blockIdx.x * SMP_WAVES_PER_BLOCK + (threadIdx.x >> 6);
}
static inline dim3 smp_grid(int64_t R) { return dim3(nsim_blocks(R, SMP_WAVES_PER_BLOCK)); }

// ``upsample_on_marched_only`` (nsim_live_rank, pack_ops.hip): live_rank[r] = q >= 0 -- ray r is the q-th ray whose march found
// occupied voxels ("live"); ~q < 0 -- it found none (q live rays precede it).  Only live rays get coarse and fine samples; every
// per-live-ray array ([R', C] coarse depths, [R', n_fine] draws and their positions / SDFs) is indexed by q, so the points of the
// SDF queries are compact: R' n instead of R n.  live_rank == NULL: every ray is live and q = r (the dense layout).
__device__ __forceinline__ bool smp_live(const int64_t* __restrict__ live_rank, int64_t r, int64_t& q) {
  if (!live_rank) {
    q = r;
    return true;
  }
  const int64_t e = live_rank[r];
  q = e >= 0 ? e : ~e;
  return e >= 0;
}

// ----------------------------------------------------------------------------------- ray generation
// ``intr.lift(u, v, 1)``: pixel -> direction in the camera frame.  Pinhole: ((u - cx) / fx, (v - cy) / fy, 1).  OpenCV
// model (camera_model 'opencv', app/resources/observers/cameras.py:84-87; Waymo's calibration, ``consider_distortion:
// true`` in the street configs): the pinhole coordinates are the DISTORTED ones, x_d = x (1 + k1 r2 + k2 r4 + k3 r6) +
// 2 p1 x y + p2 (r2 + 2 x2) (y alike), dist = (k1, k2, p1, p2, k3); the undistorted (x, y) come from the fixed-point
// iteration of cv::undistortPoints, n_iters rounds (OpenCV runs 5).  nr3d_lib's OpenCVCameraMatHW is absent: the
// iteration and its count are fixed here.  Written mul-then-add (no contraction) so that the oracle reproduces it.
// Fisheye model (camera_model 'fisheye', cameras.py:88-92; the OpenCV fisheye / Kannala-Brandt equidistant model the
// reference's app/resources/observers/fisheye.py:36-42 applies: theta_d = theta (1 + k1 theta^2 + k2 theta^4 + k3 theta^6 +
// k4 theta^8), pixel = f (x_d, y_d) + c with (x_d, y_d) = theta_d (a, b) / r): n_iters < 0 selects it, dist = (k1..k4) with
// stride 4, -n_iters Newton rounds on theta (cv::fisheye::undistortPoints runs 10) from theta = theta_d.  The lifted
// direction is (sin theta x_d / theta_d, sin theta y_d / theta_d, cos theta) -- (a, b, 1) cos theta, valid past 90 degrees.
__device__ __forceinline__ void raygen_lift_fisheye(const float* K, const float* dist, int n_iters, float w, float h, float l[3]) {
  const float xd = (w - K[2]) / K[0], yd = (h - K[5]) / K[4];
  const float td = sqrtf(xd * xd + yd * yd);
  const float k1 = dist[0], k2 = dist[1], k3 = dist[2], k4 = dist[3];
  float th = td;
  for (int it = 0; it < n_iters; ++it) {
    const float t2 = th * th;
    const float f = th * (1.0f + (((k4 * t2 + k3) * t2 + k2) * t2 + k1) * t2) - td;
    const float fp = 1.0f + (((9.0f * k4 * t2 + 7.0f * k3) * t2 + 5.0f * k2) * t2 + 3.0f * k1) * t2;
    th = th - f / fp;
  }
  const float sc = td > 1e-8f ? sinf(th) / td : 1.0f;
  l[0] = xd * sc;
  l[1] = yd * sc;
  l[2] = cosf(th);
}

__device__ __forceinline__ void raygen_lift(const float* K, const float* dist, int n_iters, float w, float h, float l[3]) {
  if (n_iters < 0) {
    raygen_lift_fisheye(K, dist, -n_iters, w, h, l);
    return;
  }
  const float x0 = (w - K[2]) / K[0], y0 = (h - K[5]) / K[4];
  float x = x0, y = y0;
  if (dist) {
    const float k1 = dist[0], k2 = dist[1], p1 = dist[2], p2 = dist[3], k3 = dist[4];
    for (int it = 0; it < n_iters; ++it) {
      const float r2 = x * x + y * y;
      const float icd = 1.0f / (1.0f + ((k3 * r2 + k2) * r2 + k1) * r2);
      const float dx = 2.0f * p1 * x * y + p2 * (r2 + 2.0f * x * x);
      const float dy = p1 * (r2 + 2.0f * y * y) + 2.0f * p2 * x * y;
      x = (x0 - dx) * icd;
      y = (y0 - dy) * icd;
    }
  }
  l[0] = x;
  l[1] = y;
  l[2] = 1.0f;
}

__global__ void __launch_bounds__(256) k_raygen_pinhole(const float* __restrict__ xy,
                                                         const int64_t* __restrict__ fidx,
                                                         const float* __restrict__ intr,
                                                         const float* __restrict__ c2w,
                                                         const int64_t* __restrict__ WH, int64_t N, int snap,
                                                         float* __restrict__ rays_o, float* __restrict__ rays_d,
                                                         const float* __restrict__ dist, int n_iters) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  const int64_t f = fidx[i];
  const float W = (float)WH[2 * f], H = (float)WH[2 * f + 1];
  float w = xy[2 * i] * W, h = xy[2 * i + 1] * H;
  if (snap) {  // (xy*WH).long().clamp(0, WH-1) + 0.5   (cameras.py:302-303)
    float wi = truncf(w), hi = truncf(h);
    wi = fminf(fmaxf(wi, 0.f), W - 1.f);
    hi = fminf(fmaxf(hi, 0.f), H - 1.f);
    w = wi + 0.5f;
    h = hi + 0.5f;
  }
  float l[3];
  raygen_lift(intr + f * 9, dist ? dist + f * (n_iters < 0 ? 4 : 5) : nullptr, n_iters, w, h, l);
  const float dx = l[0], dy = l[1], dz = l[2];      // (dz == 1.0f for the pinhole / OpenCV models)
  const float* M = c2w + f * 16;
  // broadcast-multiply-sum, never a reduced-precision matmul (cameras.py:355-359)
  float d0 = M[0] * dx + M[1] * dy + M[2] * dz;
  float d1 = M[4] * dx + M[5] * dy + M[6] * dz;
  float d2 = M[8] * dx + M[9] * dy + M[10] * dz;
  const float nrm = fmaxf(sqrtf(d0 * d0 + d1 * d1 + d2 * d2), 1e-12f);
  rays_d[3 * i + 0] = d0 / nrm;
  rays_d[3 * i + 1] = d1 / nrm;
  rays_d[3 * i + 2] = d2 / nrm;
  rays_o[3 * i + 0] = M[3];
  rays_o[3 * i + 1] = M[7];
  rays_o[3 * i + 2] = M[11];
}

// Backward of the ray generation w.r.t. the camera poses (pose refinement: the reference's LearnableParams feed refined
// c2w matrices into Camera.get_selected_rays, withmask_withlidar_joint.240219.yaml:338-352):
//   rays_o = T,  rays_d = R l / |R l|   =>   dT[f] += d_o,  dR[f] += ((I - d d^T) d_d / |R l|) (x) l
// d_c2w [V,4,4] is accumulated with atomics (a few thousand rays onto a few hundred poses; the bottom row stays zero).
__global__ void __launch_bounds__(256) k_raygen_pinhole_bwd(const float* __restrict__ xy,
                                                             const int64_t* __restrict__ fidx,
                                                             const float* __restrict__ intr,
                                                             const float* __restrict__ c2w,
                                                             const int64_t* __restrict__ WH, int64_t N, int snap,
                                                             const float* __restrict__ d_o,
                                                             const float* __restrict__ d_d,
                                                             float* __restrict__ d_c2w,
                                                             const float* __restrict__ dist, int n_iters) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  const int64_t f = fidx[i];
  const float W = (float)WH[2 * f], H = (float)WH[2 * f + 1];
  float w = xy[2 * i] * W, h = xy[2 * i + 1] * H;
  if (snap) {
    float wi = truncf(w), hi = truncf(h);
    wi = fminf(fmaxf(wi, 0.f), W - 1.f);
    hi = fminf(fmaxf(hi, 0.f), H - 1.f);
    w = wi + 0.5f;
    h = hi + 0.5f;
  }
  float l[3];
  raygen_lift(intr + f * 9, dist ? dist + f * (n_iters < 0 ? 4 : 5) : nullptr, n_iters, w, h, l);
  const float* M = c2w + f * 16;
  float dw[3];
#pragma unroll
  for (int r = 0; r < 3; ++r) dw[r] = M[4 * r] * l[0] + M[4 * r + 1] * l[1] + M[4 * r + 2] * l[2];
  const float nrm = fmaxf(sqrtf(dw[0] * dw[0] + dw[1] * dw[1] + dw[2] * dw[2]), 1e-12f);
  float* G = d_c2w + f * 16;
  if (d_d) {
    const float u[3] = {dw[0] / nrm, dw[1] / nrm, dw[2] / nrm};
    const float g[3] = {d_d[3 * i], d_d[3 * i + 1], d_d[3 * i + 2]};
    const float dot = g[0] * u[0] + g[1] * u[1] + g[2] * u[2];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      const float gr = (g[r] - dot * u[r]) / nrm;
#pragma unroll
      for (int c = 0; c < 3; ++c) atomicAdd(&G[4 * r + c], gr * l[c]);
    }
  }
  if (d_o) {
#pragma unroll
    for (int r = 0; r < 3; ++r) atomicAdd(&G[4 * r + 3], d_o[3 * i + r]);
  }
}

// ------------------------------------------------------------------------------------ AABB ray test
__global__ void __launch_bounds__(256) k_aabb_ray_test(const float* __restrict__ rays_o,
                                                        const float* __restrict__ rays_d, int64_t N, OccDev m,
                                                        float near, float far, float* __restrict__ near_out,
                                                        float* __restrict__ far_out, uint8_t* __restrict__ hit) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  float tmin = -INFINITY, tmax = INFINITY;
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const float o = rays_o[3 * i + a];
    float d = rays_d[3 * i + a];
    if (fabsf(d) < 1e-12f) d = (d < 0.f) ? -1e-12f : 1e-12f;
    const float inv = 1.0f / d;
    const float t1 = (m.mn[a] - o) * inv, t2 = (m.mx[a] - o) * inv;
    tmin = fmaxf(tmin, fminf(t1, t2));
    tmax = fminf(tmax, fmaxf(t1, t2));
  }
  const float n = fmaxf(tmin, near);
  const float f = (far >= 0.f) ? fminf(tmax, far) : tmax;
  near_out[i] = n;
  far_out[i] = f;
  hit[i] = (f > n) ? 1 : 0;
}

// ----------------------------------------------------------------------------------- occupancy grid
__global__ void __launch_bounds__(256) k_occ_decay(float* __restrict__ val, int64_t n, float decay) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) val[i] = val[i] * decay;
}

// scatter-max of f(sdf) = 4 sig(s x)(1 - sig(s x)) >= 0: integer atomicMax on the bit pattern is exact
// n_dev (may be NULL): the number of valid points is min(n, *n_dev + n_add) -- 0 when that exceeds n (a speculatively
// sized sampling pass that overflowed: the caller redoes it, see k_lotd_gather_lm)
__global__ void __launch_bounds__(256) k_occ_update(float* __restrict__ val, const float* __restrict__ pts,
                                                     const float* __restrict__ sdf, int64_t n, OccDev m,
                                                     float inv_s, const int64_t* __restrict__ n_dev, int64_t n_add) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (n_dev) {
    const int64_t nd = n_dev[0] + n_add;
    n = nd <= n ? nd : 0;
  }
  const bool ok = i < n;
  occ_collect_wave(val, m, ok, ok ? pts[3 * i] : 0.f, ok ? pts[3 * i + 1] : 0.f, ok ? pts[3 * i + 2] : 0.f, ok ? sdf[i] : 0.f,
                   inv_s);
}

__global__ void __launch_bounds__(256) k_occ_pack_bits(const float* __restrict__ val, int64_t nvox, float thre,
                                                        uint32_t* __restrict__ bits) {
  const int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t nwords = (nvox + 31) / 32;
  if (w >= nwords) return;
  uint32_t b = 0;
  for (int k = 0; k < 32; ++k) {
    const int64_t v = w * 32 + k;
    if (v < nvox && val[v] > thre) b |= (1u << k);
  }
  bits[w] = b;
}

// ------------------------------------------------------------------------------------------ marching
struct MarchRay {
  float o[3], d[3], near, far, jit;
  int K;
};

__device__ __forceinline__ MarchRay march_load(const float* rays_o, const float* rays_d, const float* near,
                                               const float* far, const float* jitter, int64_t r, float step,
                                               int max_steps) {
  MarchRay m;
  for (int a = 0; a < 3; ++a) {
    m.o[a] = rays_o[3 * r + a];
    m.d[a] = rays_d[3 * r + a];
  }
  m.near = near[r];
  m.far = far[r];
  m.jit = jitter ? jitter[r] : 0.5f;
  float kf = ceilf((m.far - m.near) / step);
  kf = fminf(fmaxf(kf, 0.f), (float)max_steps);
  m.K = (int)kf;
  return m;
}

__device__ __forceinline__ bool march_test(const MarchRay& m, const OccDev& occ, const uint32_t* bits, int k,
                                           float step, float& t) {
  t = m.near + ((float)k + m.jit) * step;
  if (k >= m.K || !(t < m.far)) return false;
  const float px = m.o[0] + t * m.d[0], py = m.o[1] + t * m.d[1], pz = m.o[2] + t * m.d[2];
  int64_t flat;
  if (!occ_voxel(occ, px, py, pz, flat)) return false;
  return (bits[flat >> 5] >> (flat & 31)) & 1u;
}

__global__ void __launch_bounds__(SMP_BLOCK) k_march_count(
    const float* __restrict__ rays_o, const float* __restrict__ rays_d, const float* __restrict__ near,
    const float* __restrict__ far, const float* __restrict__ jitter, int64_t R, const uint32_t* __restrict__ bits,
    const int64_t* __restrict__ ray_word_off, OccDev occ, float step, int max_steps, int64_t* __restrict__ counts) {
  const int64_t r = smp_wave_id();
  if (r >= R) return;
  const int lane = nsim_lane();
  if (ray_word_off) bits += ray_word_off[r];      // batched occupancy grid: this ray's instance
  const MarchRay m = march_load(rays_o, rays_d, near, far, jitter, r, step, max_steps);
  int cnt = 0;
  for (int base = 0; base < m.K; base += 64) {
    float t;
    const bool hit = march_test(m, occ, bits, base + lane, step, t);
    cnt += __popcll(wave_ballot(hit));
  }
  if (lane == 0) counts[r] = cnt;
}

__global__ void __launch_bounds__(SMP_BLOCK) k_march_emit(
    const float* __restrict__ rays_o, const float* __restrict__ rays_d, const float* __restrict__ near,
    const float* __restrict__ far, const float* __restrict__ jitter, int64_t R, const uint32_t* __restrict__ bits,
    const int64_t* __restrict__ ray_word_off, OccDev occ, float step, int max_steps, const int64_t* __restrict__ pi,
    float* __restrict__ t_out) {
  const int64_t r = smp_wave_id();
  if (r >= R) return;
  const int lane = nsim_lane();
  if (ray_word_off) bits += ray_word_off[r];
  if (pi[2 * r + 1] == 0) return;      // nothing to emit (or a pack emptied by a speculative capacity, pack_ops.hip)
  const MarchRay m = march_load(rays_o, rays_d, near, far, jitter, r, step, max_steps);
  const int64_t st = pi[2 * r];
  int cnt = 0;
  for (int base = 0; base < m.K; base += 64) {
    float t;
    const bool hit = march_test(m, occ, bits, base + lane, step, t);
    const unsigned long long mask = wave_ballot(hit);
    if (hit) {
      const int before = __popcll(mask & ((1ull << lane) - 1ull));
      t_out[st + cnt + before] = t;
    }
    cnt += __popcll(mask);
  }
}

__global__ void __launch_bounds__(256) k_coarse_depths(const float* __restrict__ near, const float* __restrict__ far,
                                                        const float* __restrict__ jc, int64_t R, int C,
                                                        float* __restrict__ out, const int64_t* __restrict__ live_rank) {
  const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= R * C) return;
  const int64_t r = j / C;
  const int i = (int)(j % C);
  int64_t q;
  if (!smp_live(live_rank, r, q)) return;
  const float u = jc ? jc[j] : 0.5f;
  out[q * C + i] = near[r] + (far[r] - near[r]) * (((float)i + u) / (float)C);
}

// ---------------------------------------------------------------------------------------- up-sampling
__device__ __forceinline__ float smp_sigmoid(float x) { return 1.0f / (1.0f + expf(-x)); }

// interval opacity used for up-sampling (NeuS `up_sample`, or the plain consecutive-sdf form)
__device__ __forceinline__ float upsample_alpha(const float* t, const float* sdf, int64_t i, float inv_s,
                                                int use_est) {
  const float s0 = sdf[i], s1 = sdf[i + 1];
  if (use_est) {
    const float t0 = t[i], t1 = t[i + 1];
    const float dist = t1 - t0;
    float cs = (s1 - s0) / (dist + 1e-5f);
    float prev = 0.f;
    if (i > 0) prev = (s0 - sdf[i - 1]) / ((t0 - t[i - 1]) + 1e-5f);
    cs = fminf(prev, cs);
    cs = fminf(fmaxf(cs, -1e3f), 0.f);
    const float mid = (s0 + s1) * 0.5f;
    const float half = cs * dist * 0.5f;
    const float pc = smp_sigmoid((mid - half) * inv_s), nc = smp_sigmoid((mid + half) * inv_s);
    return (pc - nc + 1e-5f) / (pc + 1e-5f);
  } else {
    const float c0 = smp_sigmoid(s0 * inv_s), c1 = smp_sigmoid(s1 * inv_s);
    const float a = (c0 - c1 + 1e-5f) / (c0 + 1e-5f);
    return fminf(fmaxf(a, 0.f), 1.f);
  }
}

// one ray (one wave): n_fine new depths drawn from the interval weights of the ray's samples tt / ss [n]; ``lds`` holds the
// running sums when the ray fits (SMP_LDS_FLOATS intervals), the global scratch ``cs_glob`` otherwise
__device__ __forceinline__ void upsample_ray(const float* tt, const float* ss, int64_t n, int64_t r, int64_t q, float inv_s, int n_fine,
                                             int use_est, float* lds, float* cs_glob, float* __restrict__ t_new,
                                             const float* __restrict__ rays_o, const float* __restrict__ rays_d,
                                             float* __restrict__ x_new) {
  const int lane = nsim_lane();
  const int64_t ni = n - 1;  // intervals
  float ro[3] = {0.f, 0.f, 0.f}, rd[3] = {0.f, 0.f, 0.f};
  if (x_new) {
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      ro[c] = rays_o[3 * r + c];
      rd[c] = rays_d[3 * r + c];
    }
  }
  auto body = [&](auto cs) {
    float carry_T = 1.0f, carry_S = 0.f;
    for (int64_t base = 0; base < ni; base += 64) {
      const int64_t i = base + lane;
      const bool valid = i < ni;
      const float a = valid ? upsample_alpha(tt, ss, i, inv_s, use_est) : 0.f;
      const float f = valid ? (1.0f - a + 1e-7f) : 1.0f;
      const float incl = wave_incl_prod(f);
      float excl = wave_shfl(incl, lane - 1);
      if (lane == 0) excl = 1.0f;
      const float T = carry_T * excl;
      carry_T = carry_T * wave_shfl(incl, 63);
      const float w = valid ? (a * T + 1e-5f) : 0.f;
      const float ws = wave_incl_sum(w);
      if (valid) cs[i] = carry_S + ws;
      carry_S = carry_S + wave_shfl(ws, 63);
    }
    nsim_wave_fence();
    const float wsum = carry_S;
    for (int kb = 0; kb < n_fine; kb += 64) {
      const int k = kb + lane;
      if (k >= n_fine) continue;
      float tn = (n > 0) ? tt[0] : 0.f;
      if (ni > 0) {
        const float u = ((float)k + 0.5f) / (float)n_fine;
        // first interval i with cdf[i+1] = cs[i]/wsum > u, clamped to the last interval
        int64_t lo = 0, hi = ni;
        while (lo < hi) {
          const int64_t mid = (lo + hi) >> 1;
          if (cs[mid] / wsum <= u) lo = mid + 1; else hi = mid;
        }
        if (lo > ni - 1) lo = ni - 1;
        const float c_lo = (lo == 0) ? 0.f : cs[lo - 1] / wsum;
        const float c_hi = cs[lo] / wsum;
        float den = c_hi - c_lo;
        if (den < 1e-5f) den = 1.0f;
        float frac = (u - c_lo) / den;
        frac = fminf(fmaxf(frac, 0.f), 1.f);
        const float b_lo = tt[lo], b_hi = tt[lo + 1];
        tn = b_lo + frac * (b_hi - b_lo);
      }
      t_new[q * n_fine + k] = tn;
      if (x_new) {   // the sample's position, so that the level-major query loads 12 B instead of re-deriving it per XCD
#pragma unroll
        for (int c = 0; c < 3; ++c) x_new[(q * n_fine + k) * 3 + c] = ro[c] + tn * rd[c];
      }
    }
  };
  if (ni <= SMP_LDS_FLOATS) body(lds);
  else body(cs_glob);
}

__global__ void __launch_bounds__(SMP_BLOCK) k_upsample_stage(const float* __restrict__ t,
                                                                const float* __restrict__ sdf,
                                                                const int64_t* __restrict__ pi, int64_t R,
                                                                float inv_s, int n_fine, int use_est,
                                                                float* __restrict__ csum,
                                                                float* __restrict__ t_new,
                                                                const float* __restrict__ rays_o,
                                                                const float* __restrict__ rays_d,
                                                                float* __restrict__ x_new,
                                                                const int64_t* __restrict__ live_rank) {
  // the running sums of a ray's interval weights live in LDS (<= SMP_LDS_FLOATS intervals; longer rays use the global
  // scratch): the inverse-CDF search below is eight DEPENDENT reads per new sample -- 13 us per launch from L2
  __shared__ float smp_lds[SMP_WAVES_PER_BLOCK][SMP_LDS_FLOATS];
  const int64_t r = smp_wave_id();
  if (r >= R) return;
  int64_t q;
  if (!smp_live(live_rank, r, q)) return;      // no marched samples: no draws
  const int64_t st = pi[2 * r], n = pi[2 * r + 1];
  upsample_ray(t + st, sdf + st, n, r, q, inv_s, n_fine, use_est, &smp_lds[threadIdx.x >> 6][0], csum + st, t_new, rays_o, rays_d,
               x_new);
}

// -------------------------------------------------------------------------------- sorted merge
__device__ __forceinline__ int64_t smp_lower_bound(const float* a, int64_t n, float v) {
  int64_t lo = 0, hi = n;
  while (lo < hi) {
    const int64_t mid = (lo + hi) >> 1;
    if (a[mid] < v) lo = mid + 1; else hi = mid;
  }
  return lo;
}
__device__ __forceinline__ int64_t smp_upper_bound(const float* a, int64_t n, float v) {
  int64_t lo = 0, hi = n;
  while (lo < hi) {
    const int64_t mid = (lo + hi) >> 1;
    if (a[mid] <= v) lo = mid + 1; else hi = mid;
  }
  return lo;
}

// one ray (one wave): merge its sorted depth lists a [na] (values v_a) and b [nb] (values v_b) into the packed output at
// ``so``; ``lds`` holds both lists when they fit
__device__ __forceinline__ void merge_ray(const float* __restrict__ t_a, const float* __restrict__ v_a, int64_t sa, int64_t na,
                                          const float* __restrict__ t_b, const float* __restrict__ v_b, int64_t r, int64_t q, int nb,
                                          int64_t so, float* lds, float* __restrict__ t_out, float* __restrict__ v_out,
                                          int64_t* __restrict__ ridx_out, const float* __restrict__ rays_o,
                                          const float* __restrict__ rays_d, float* __restrict__ x_out) {
  const int lane = nsim_lane();
  float ro[3] = {0.f, 0.f, 0.f}, rd[3] = {0.f, 0.f, 0.f};
  if (x_out) {
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      ro[c] = rays_o[3 * r + c];
      rd[c] = rays_d[3 * r + c];
    }
  }
  // both depth lists of the ray in LDS when they fit (a first, then b): the rank searches are 6-8 DEPENDENT reads each
  const bool in_lds = na + nb <= SMP_LDS_FLOATS;
  float* la = lds;
  float* lb = la + na;
  if (in_lds) {
    for (int64_t i = lane; i < na; i += 64) la[i] = t_a[sa + i];
    for (int64_t j = lane; j < nb; j += 64) lb[j] = t_b[q * (int64_t)nb + j];
    nsim_wave_fence();
  }
  auto body = [&](auto a, auto b) {
    for (int64_t i = lane; i < na; i += 64) {  // a first on ties: b elements strictly smaller precede
      const float v = a[i];
      const int64_t pos = i + smp_lower_bound(b, nb, v);
      t_out[so + pos] = v;
      if (v_out) v_out[so + pos] = v_a ? v_a[sa + i] : 0.f;
      if (ridx_out) ridx_out[so + pos] = r;
      if (x_out) {
#pragma unroll
        for (int c = 0; c < 3; ++c) x_out[(so + pos) * 3 + c] = ro[c] + v * rd[c];
      }
    }
    for (int64_t j = lane; j < nb; j += 64) {
      const float v = b[j];
      const int64_t pos = j + smp_upper_bound(a, na, v);
      t_out[so + pos] = v;
      if (v_out) v_out[so + pos] = v_b ? v_b[q * (int64_t)nb + j] : 0.f;
      if (ridx_out) ridx_out[so + pos] = r;
      if (x_out) {
#pragma unroll
        for (int c = 0; c < 3; ++c) x_out[(so + pos) * 3 + c] = ro[c] + v * rd[c];
      }
    }
  };
  if (in_lds) body((const float*)la, (const float*)lb);
  else body(t_a + sa, t_b + q * (int64_t)nb);
}

__global__ void __launch_bounds__(SMP_BLOCK) k_merge_sorted(const float* __restrict__ t_a,
                                                              const float* __restrict__ v_a,
                                                              const int64_t* __restrict__ pia,
                                                              const float* __restrict__ t_b,
                                                              const float* __restrict__ v_b, int64_t R, int nb,
                                                              float* __restrict__ t_out, float* __restrict__ v_out,
                                                              int64_t* __restrict__ pio, int64_t* __restrict__ ridx_out,
                                                              const float* __restrict__ rays_o,
                                                              const float* __restrict__ rays_d,
                                                              float* __restrict__ x_out,
                                                              const int64_t* __restrict__ live_rank) {
  __shared__ float smp_lds[SMP_WAVES_PER_BLOCK][SMP_LDS_FLOATS];
  const int64_t r = smp_wave_id();
  if (r >= R) return;
  int64_t q;
  const bool live = smp_live(live_rank, r, q);
  const int64_t sa = pia[2 * r], na = live ? pia[2 * r + 1] : 0;
  const int64_t so = sa + q * (int64_t)nb;       // q live rays precede r, each with nb list-b samples
  const int nb_r = live ? nb : 0;
  if (nsim_lane() == 0) {
    pio[2 * r] = so;
    pio[2 * r + 1] = na + nb_r;
  }
  if (!live) return;
  merge_ray(t_a, v_a, sa, na, t_b, v_b, r, q, nb, so, &smp_lds[threadIdx.x >> 6][0], t_out, v_out, ridx_out, rays_o, rays_d, x_out);
}

// merge of up-sampling stage k AND the draw of stage k + 1 in one launch (round 4): both are one-wave-per-ray; the merged
// samples of a ray are handed from the merge to the draw through t_out / v_out (written and read by the SAME wave, ordered by
// the wave-level fence), the LDS row is used by the merge first and by the draw's running sums afterwards
__global__ void __launch_bounds__(SMP_BLOCK) k_merge_upsample(const float* __restrict__ t_a, const float* __restrict__ v_a,
                                                                const int64_t* __restrict__ pia, const float* __restrict__ t_b,
                                                                const float* __restrict__ v_b, int64_t R, int nb,
                                                                float* __restrict__ t_out, float* __restrict__ v_out,
                                                                int64_t* __restrict__ pio, int64_t* __restrict__ ridx_out,
                                                                float inv_s, int n_fine, int use_est, float* __restrict__ csum,
                                                                float* __restrict__ t_new, const float* __restrict__ rays_o,
                                                                const float* __restrict__ rays_d, float* __restrict__ x_new,
                                                                const int64_t* __restrict__ live_rank) {
  __shared__ float smp_lds[SMP_WAVES_PER_BLOCK][SMP_LDS_FLOATS];
  const int64_t r = smp_wave_id();
  if (r >= R) return;
  int64_t q;
  const bool live = smp_live(live_rank, r, q);
  const int64_t sa = pia[2 * r], na = live ? pia[2 * r + 1] : 0;
  const int64_t so = sa + q * (int64_t)nb;
  const int nb_r = live ? nb : 0;
  if (nsim_lane() == 0) {
    pio[2 * r] = so;
    pio[2 * r + 1] = na + nb_r;
  }
  if (!live) return;
  float* lds = &smp_lds[threadIdx.x >> 6][0];
  merge_ray(t_a, v_a, sa, na, t_b, v_b, r, q, nb, so, lds, t_out, v_out, ridx_out, nullptr, nullptr, nullptr);
  nsim_wave_fence();
  upsample_ray(t_out + so, v_out + so, na + nb, r, q, inv_s, n_fine, use_est, lds, csum + so, t_new, rays_o, rays_d, x_new);
}

// ------------------------------------------------------------------ compressed query mode
// ``query_mode: march_occ_multi_upsample_compressed`` (lotd_neus.dtu.230814.yaml:157): after up-sampling every
// sample carries a no-grad SDF; samples whose visibility weight is negligible are dropped before the expensive
// with-grad query.  A sample is kept iff one of the two intervals it bounds has vw > thre; surviving neighbours
// form merged intervals (the NeuS opacity telescopes: 1 - alpha(i,j) = Phi(s_j)/Phi(s_i)).
// One wave per ray: alpha -> transmittance scan -> keep flags -> ballot compaction.
__device__ __forceinline__ float compress_alpha(const float* sdf, int64_t i, int64_t n, float s) {
  if (i + 1 >= n) return 0.f;
  const float c0 = smp_sigmoid(sdf[i] * s), c1 = smp_sigmoid(sdf[i + 1] * s);
  const float a = (c0 - c1 + 1e-5f) / (c0 + 1e-5f);
  return fminf(fmaxf(a, 0.f), 1.f);
}

template <bool EMIT>
__global__ void __launch_bounds__(SMP_BLOCK) k_compress(const float* __restrict__ sdf, const float* __restrict__ t,
                                                         const int64_t* __restrict__ pi, int64_t R,
                                                         const float* __restrict__ ln_inv_s, float factor,
                                                         float forward_inv_s, float thre, int64_t* __restrict__ counts,
                                                         const int64_t* __restrict__ pi_out, float* __restrict__ t_out,
                                                         int64_t* __restrict__ ridx_out, int64_t tail_n) {
  const int64_t r = smp_wave_id();
  if (r >= R) return;
  const int lane = nsim_lane();
  if (EMIT && tail_n > 0) {  // tail_n zero-length samples on the pseudo-rays R, R+1, ... behind the kept set
    const int64_t S = pi_out[2 * (R - 1)] + pi_out[2 * (R - 1) + 1];
    for (int64_t i = r * 64 + lane; i < tail_n; i += R * 64) {
      t_out[S + i] = 0.f;
      ridx_out[S + i] = R + i;
    }
  }
  const int64_t st = pi[2 * r], n = pi[2 * r + 1];
  const float s = forward_inv_s > 0.f ? forward_inv_s : expf(ln_inv_s[0] * factor);
  const float* ss = sdf + st;
  float carry = 1.0f;
  bool prev_sig = false;  // significance of the interval ending at the first sample of the chunk
  int cnt = 0;
  const int64_t ost = EMIT ? pi_out[2 * r] : 0;
  for (int64_t base = 0; base < n; base += 64) {
    const int64_t i = base + lane;
    const bool valid = i < n;
    const float a = valid ? compress_alpha(ss, i, n, s) : 0.f;
    const float f = valid ? (1.0f - a + 1e-10f) : 1.0f;
    const float incl = wave_incl_prod(f);
    float excl = wave_shfl(incl, lane - 1);
    if (lane == 0) excl = 1.0f;
    const float vw = a * (carry * excl);
    carry = carry * wave_shfl(incl, 63);
    const bool sig = valid && (vw > thre);
    int psig = wave_shfl((int)sig, lane - 1);
    if (lane == 0) psig = prev_sig ? 1 : 0;
    prev_sig = wave_shfl((int)sig, 63) != 0;
    const bool keep = valid && (sig || psig != 0);
    const unsigned long long m = wave_ballot(keep);
    if (EMIT && keep) {
      const int before = __popcll(m & ((1ull << lane) - 1ull));
      t_out[ost + cnt + before] = t[st + i];
      ridx_out[ost + cnt + before] = r;
    }
    cnt += __popcll(m);
  }
  if (!EMIT && lane == 0) counts[r] = cnt;
}

// ================================================================================== C ABI
extern "C" {

int nsim_raygen_pinhole(const float* xy, const int64_t* fidx, const float* intr, const float* c2w,
                        const int64_t* WH, int64_t N, int snap, float* rays_o, float* rays_d, void* stream) {
  if (N <= 0) return 0;
  hipLaunchKernelGGL(k_raygen_pinhole, dim3(nsim_blocks(N, 256)), dim3(256), 0, (hipStream_t)stream, xy, fidx, intr, c2w,
                     WH, N, snap, rays_o, rays_d, (const float*)nullptr, 0);
  NSIM_CHECK_LAUNCH();
  return 0;
}

int nsim_raygen_opencv(const float* xy, const int64_t* fidx, const float* intr, const float* distortion, int n_iters,
                       const float* c2w, const int64_t* WH, int64_t N, int snap, float* rays_o, float* rays_d,
                       void* stream) {
  if (N <= 0) return 0;
  if (!distortion || n_iters < 0 || n_iters > 64) return 4;
  hipLaunchKernelGGL(k_raygen_pinhole, dim3(nsim_blocks(N, 256)), dim3(256), 0, (hipStream_t)stream, xy, fidx, intr, c2w,
                     WH, N, snap, rays_o, rays_d, distortion, n_iters);
  NSIM_CHECK_LAUNCH();
  return 0;
}

int nsim_raygen_fisheye(const float* xy, const int64_t* fidx, const float* intr, const float* distortion, int n_iters,
                        const float* c2w, const int64_t* WH, int64_t N, int snap, float* rays_o, float* rays_d,
                        void* stream) {
  if (N <= 0) return 0;
  if (!distortion || n_iters < 1 || n_iters > 64) return 4;
  hipLaunchKernelGGL(k_raygen_pinhole, dim3(nsim_blocks(N, 256)), dim3(256), 0, (hipStream_t)stream, xy, fidx, intr, c2w,
                     WH, N, snap, rays_o, rays_d, distortion, -n_iters);
  NSIM_CHECK_LAUNCH();
  return 0;
}

int nsim_raygen_fisheye_bwd(const float* xy, const int64_t* fidx, const float* intr, const float* distortion, int n_iters,
                            const float* c2w, const int64_t* WH, int64_t N, int snap, const float* d_rays_o,
                            const float* d_rays_d, float* d_c2w, void* stream) {
  if (N <= 0) return 0;
  if (!d_c2w || !distortion || n_iters < 1 || n_iters > 64) return 4;
  hipLaunchKernelGGL(k_raygen_pinhole_bwd, dim3(nsim_blocks(N, 256)), dim3(256), 0, (hipStream_t)stream, xy, fidx, intr,
                     c2w, WH, N, snap, d_rays_o, d_rays_d, d_c2w, distortion, -n_iters);
  NSIM_CHECK_LAUNCH();
  return 0;
}

int nsim_raygen_pinhole_bwd(const float* xy, const int64_t* fidx, const float* intr, const float* c2w,
                            const int64_t* WH, int64_t N, int snap, const float* d_rays_o, const float* d_rays_d,
                            float* d_c2w, void* stream) {
  if (N <= 0) return 0;
  if (!d_c2w) return 4;
  hipLaunchKernelGGL(k_raygen_pinhole_bwd, dim3(nsim_blocks(N, 256)), dim3(256), 0, (hipStream_t)stream, xy, fidx, intr,
                     c2w, WH, N, snap, d_rays_o, d_rays_d, d_c2w, (const float*)nullptr, 0);
  NSIM_CHECK_LAUNCH();
  return 0;
}

int nsim_raygen_opencv_bwd(const float* xy, const int64_t* fidx, const float* intr, const float* distortion, int n_iters,
                           const float* c2w, const int64_t* WH, int64_t N, int snap, const float* d_rays_o,
                           const float* d_rays_d, float* d_c2w, void* stream) {
  if (N <= 0) return 0;
  if (!d_c2w || !distortion || n_iters < 0 || n_iters > 64) return 4;
  hipLaunchKernelGGL(k_raygen_pinhole_bwd, dim3(nsim_blocks(N, 256)), dim3(256), 0, (hipStream_t)stream, xy, fidx, intr,
                     c2w, WH, N, snap, d_rays_o, d_rays_d, d_c2w, distortion, n_iters);
  NSIM_CHECK_LAUNCH();
  return 0;
}

int nsim_aabb_ray_test(const float* rays_o, const float* rays_d, int64_t N, const NsimOccMeta* meta, float near,
                       float far, float* near_out, float* far_out, uint8_t* hit, void* stream) {
  if (N <= 0) return 0;
  if (!meta) return 5;
  hipLaunchKernelGGL(k_aabb_ray_test, dim3(nsim_blocks(N, 256)), dim3(256), 0, (hipStream_t)stream, rays_o, rays_d, N,
                     occ_dev(meta), near, far, near_out, far_out, hit);
  NSIM_CHECK_LAUNCH();
  return 0;
}

int nsim_occ_decay(float* val, int64_t nvox, float decay, void* stream) {
  if (nvox <= 0) return 0;
  hipLaunchKernelGGL(k_occ_decay, dim3(nsim_blocks(nvox, 256)), dim3(256), 0, (hipStream_t)stream, val, nvox, decay);
  NSIM_CHECK_LAUNCH();
  return 0;
}

int nsim_occ_update(float* val, const float* pts, const float* sdf, int64_t n, const NsimOccMeta* meta,
                    float inv_s, void* stream) {
  if (n <= 0) return 0;
  if (!meta) return 5;
  hipLaunchKernelGGL(k_occ_update, dim3(nsim_blocks(n, 256)), dim3(256), 0, (hipStream_t)stream, val, pts, sdf, n,
                     occ_dev(meta), inv_s, (const int64_t*)nullptr, (int64_t)0);
  NSIM_CHECK_LAUNCH();
  return 0;
}

int nsim_occ_collect(float* val, const float* pts, const float* sdf, int64_t n, const int64_t* n_dev, int64_t n_add,
                     const NsimOccMeta* meta, float inv_s, void* stream) {
  if (n <= 0) return 0;
  if (!meta) return 5;
  hipLaunchKernelGGL(k_occ_update, dim3(nsim_blocks(n, 256)), dim3(256), 0, (hipStream_t)stream, val, pts, sdf, n,
                     occ_dev(meta), inv_s, n_dev, n_add);
  NSIM_CHECK_LAUNCH();
  return 0;
}

int nsim_occ_pack_bits(const float* val, int64_t nvox, float thre, uint32_t* bits, void* stream) {
  if (nvox <= 0) return 0;
  hipLaunchKernelGGL(k_occ_pack_bits, dim3(nsim_blocks((nvox + 31) / 32, 256)), dim3(256), 0, (hipStream_t)stream, val,
                     nvox, thre, bits);
  NSIM_CHECK_LAUNCH();
  return 0;
}

int nsim_march_count(const float* rays_o, const float* rays_d, const float* near, const float* far,
                     const float* jitter, int64_t R, const uint32_t* bits, const int64_t* ray_word_off,
                     const NsimOccMeta* meta, float step, int max_steps, int64_t* counts, void* stream) {
  if (R <= 0) return 0;
  if (!meta || !(step > 0.f)) return 5;
  hipLaunchKernelGGL(k_march_count, smp_grid(R), dim3(SMP_BLOCK), 0, (hipStream_t)stream, rays_o, rays_d, near, far,
                     jitter, R, bits, ray_word_off, occ_dev(meta), step, max_steps, counts);
  NSIM_CHECK_LAUNCH();
  return 0;
}

int nsim_march_emit(const float* rays_o, const float* rays_d, const float* near, const float* far,
                    const float* jitter, int64_t R, const uint32_t* bits, const int64_t* ray_word_off,
                    const NsimOccMeta* meta, float step, int max_steps, const int64_t* pack_infos, float* t_out,
                    void* stream) {
  if (R <= 0) return 0;
  if (!meta || !(step > 0.f)) return 5;
  hipLaunchKernelGGL(k_march_emit, smp_grid(R), dim3(SMP_BLOCK), 0, (hipStream_t)stream, rays_o, rays_d, near, far, jitter,
                     R, bits, ray_word_off, occ_dev(meta), step, max_steps, pack_infos, t_out);
  NSIM_CHECK_LAUNCH();
  return 0;
}

int nsim_coarse_depths(const float* near, const float* far, const float* jitter_c, int64_t R, int C, float* out,
                       const int64_t* live_rank, void* stream) {
  if (R <= 0 || C <= 0) return 0;
  hipLaunchKernelGGL(k_coarse_depths, dim3(nsim_blocks(R * C, 256)), dim3(256), 0, (hipStream_t)stream, near, far,
                     jitter_c, R, C, out, live_rank);
  NSIM_CHECK_LAUNCH();
  return 0;
}

int nsim_upsample_stage(const float* t, const float* sdf, const int64_t* pack_infos, int64_t R, float inv_s,
                        int n_fine, int use_estimate_alpha, float* scratch, float* t_new, const float* rays_o,
                        const float* rays_d, float* x_new, const int64_t* live_rank, void* stream) {
  if (R <= 0 || n_fine <= 0) return 0;
  if (x_new && !(rays_o && rays_d)) return 24;
  hipLaunchKernelGGL(k_upsample_stage, smp_grid(R), dim3(SMP_BLOCK), 0, (hipStream_t)stream, t, sdf, pack_infos, R, inv_s,
                     n_fine, use_estimate_alpha, scratch, t_new, rays_o, rays_d, x_new, live_rank);
  NSIM_CHECK_LAUNCH();
  return 0;
}

int nsim_merge_sorted(const float* t_a, const float* v_a, const int64_t* pack_infos_a, const float* t_b,
                      const float* v_b, int64_t R, int nb, float* t_out, float* v_out, int64_t* pack_infos_out,
                      int64_t* ridx_out, const float* rays_o, const float* rays_d, float* x_out, const int64_t* live_rank,
                      void* stream) {
  if (R <= 0) return 0;
  if (x_out && !(rays_o && rays_d)) return 24;
  hipLaunchKernelGGL(k_merge_sorted, smp_grid(R), dim3(SMP_BLOCK), 0, (hipStream_t)stream, t_a, v_a, pack_infos_a, t_b, v_b,
                     R, nb, t_out, v_out, pack_infos_out, ridx_out, rays_o, rays_d, x_out, live_rank);
  NSIM_CHECK_LAUNCH();
  return 0;
}

int nsim_merge_upsample(const float* t_a, const float* v_a, const int64_t* pack_infos_a, const float* t_b, const float* v_b,
                        int64_t R, int nb, float* t_out, float* v_out, int64_t* pack_infos_out, int64_t* ridx_out, float inv_s,
                        int n_fine, int use_estimate_alpha, float* scratch, float* t_new, const float* rays_o,
                        const float* rays_d, float* x_new, const int64_t* live_rank, void* stream) {
  if (R <= 0) return 0;
  if (n_fine <= 0 || !v_a || !v_b || !v_out || !scratch || !t_new) return 4;
  if (x_new && !(rays_o && rays_d)) return 24;
  hipLaunchKernelGGL(k_merge_upsample, smp_grid(R), dim3(SMP_BLOCK), 0, (hipStream_t)stream, t_a, v_a, pack_infos_a, t_b, v_b, R,
                     nb, t_out, v_out, pack_infos_out, ridx_out, inv_s, n_fine, use_estimate_alpha, scratch, t_new, rays_o,
                     rays_d, x_new, live_rank);
  NSIM_CHECK_LAUNCH();
  return 0;
}

int nsim_compress_count(const float* sdf, const int64_t* pack_infos, int64_t R, const float* ln_inv_s,
                        float ln_inv_s_factor, float forward_inv_s, float thre, int64_t* counts, void* stream) {
  if (R <= 0) return 0;
  hipLaunchKernelGGL((k_compress<false>), smp_grid(R), dim3(SMP_BLOCK), 0, (hipStream_t)stream, sdf, nullptr, pack_infos, R,
                     ln_inv_s, ln_inv_s_factor, forward_inv_s, thre, counts, nullptr, nullptr, nullptr, (int64_t)0);
  NSIM_CHECK_LAUNCH();
  return 0;
}

int nsim_compress_emit(const float* sdf, const float* t, const int64_t* pack_infos, int64_t R, const float* ln_inv_s,
                       float ln_inv_s_factor, float forward_inv_s, float thre, const int64_t* pack_infos_out,
                       float* t_out, int64_t* ridx_out, int64_t tail_n, void* stream) {
  if (R <= 0) return 0;
  hipLaunchKernelGGL((k_compress<true>), smp_grid(R), dim3(SMP_BLOCK), 0, (hipStream_t)stream, sdf, t, pack_infos, R,
                     ln_inv_s, ln_inv_s_factor, forward_inv_s, thre, nullptr, pack_infos_out, t_out, ridx_out, tail_n);
  NSIM_CHECK_LAUNCH();
  return 0;
}

}  // extern "C"
