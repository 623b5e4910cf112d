/* nsim.h -- C ABI of libnsim_hip.so, the MI355X (gfx950) implementation of the NeuS / StreetSurf
 * volume-render hot path of PJLab-ADG/neuralsim.
 *
 * The reference has no C/FFI plugin interface: its extension mechanism is Python class substitution
 * (``import_str(cfg.model_class)(**cfg.model_params)``, app/resources/asset_bank.py:129-138) and the
 * native work sits behind the Python API of the (un-vendored) nr3d_lib.  Every entry point below
 * therefore names the nr3d_lib Python symbol / reference call site whose native half it replaces; the
 * binding a maintainer adds is the ctypes stub shown in INTEGRATION.md (neuralsim_amd/_lib.py).
 *
 * Conventions:
 *   - plain pointers + sizes, no C++/torch types; all pointers are DEVICE pointers unless marked host;
 *   - never allocates, never synchronises; work is enqueued on ``stream`` (a hipStream_t);
 *   - returns 0 on success, a positive code otherwise (1000 + hipError_t for launch failures,
 *     1..99 for argument errors); nsim_strerror() explains argument errors;
 *   - "packed" arrays: ``pack_infos`` is int64 [P,2] = (first index, count) per ray
 *     (nr3d_lib.graphics.pack_ops.get_pack_infos_from_n; buffer_compose_renderer.py:991).
 */
#ifndef NSIM_H
#define NSIM_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NSIM_MAX_LEVELS 32
#define NSIM_LOTD_DENSE 0
#define NSIM_LOTD_HASH 1

/* LoTD level table (host struct, passed by pointer, copied by value into the launch).
 * nr3d_lib.models.grid_encodings.lotd ``lotd_cfg{lod_res, lod_n_feats, lod_types, hashmap_size}``
 * (code_single/configs/object_centric/lotd_neus.dtu.230814.yaml:96-101). n_feats is 2 for every level. */
typedef struct NsimLotdMeta {
  int32_t num_levels;
  int32_t n_feats;                     /* must be 2 */
  int32_t n_active_levels;             /* hardmask level annealing (encoding_cfg.anneal_cfg{type: hardmask},
                                        * lotd_neus.dtu.230814.yaml:104-108): levels >= n_active_levels yield zero
                                        * features and receive no gradient; 0 or >= num_levels = all active */
  int32_t res[NSIM_MAX_LEVELS][3];     /* vertices per axis (x, y, z); equal for cubic levels, per-axis for
                                        * ``lotd_use_cuboid`` (withmask_withlidar_joint.240219.yaml:160) */
  int32_t type[NSIM_MAX_LEVELS];       /* NSIM_LOTD_DENSE | NSIM_LOTD_HASH */
  uint32_t size[NSIM_MAX_LEVELS];      /* entries (vertices or hash slots) */
  int64_t offset[NSIM_MAX_LEVELS];     /* offset of the level in the flat param tensor, in scalars */
  float x_scale[3];                    /* position -> unit coordinate of the pyramid, per axis: u = x * x_scale + x_shift */
  float x_shift[3];                    /* = the model's AABB normalisation (nr3d_lib AABBSpace: lo -> 0, hi -> 1);
                                        * x_scale all zero = the [-1,1]^3 cube (0.5, 0.5) */
} NsimLotdMeta;

/* Occupancy-grid / AABB description (host struct). nr3d_lib.models.accelerations.OccGridAccel +
 * nr3d_lib.models.spatial.AABBSpace (``accel_cfg{type: occ_grid, resolution: [64,64,64]}``,
 * lotd_neus.dtu.230814.yaml:140-155). scale = res / (aabb_max - aabb_min). */
typedef struct NsimOccMeta {
  float aabb_min[3];
  float aabb_max[3];
  float scale[3];
  int32_t res[3];
} NsimOccMeta;

const char* nsim_strerror(int code);
int nsim_version(void);

/* ---------------------------------------------------------------- pack ops (graphics.pack_ops) */
/* get_pack_infos_from_n(n) ; total (device int64[1], may be NULL) receives sum(n).
 * cap < 0: plain.  cap >= 0: the caller sized its sample buffers for cap elements WITHOUT knowing sum(n); a pack
 * that would end beyond cap gets count 0 (its start is kept), so consumers stay in bounds; total still receives the
 * true sum and the caller redoes the pass when it later reads total > cap. */
int nsim_pack_infos_from_n(const int64_t* n, int64_t P, int64_t* pack_infos, int64_t* total, int64_t cap,
                           void* stream);
/* The same, and the kernel also stores (sum(n), seq) to ``notify`` [2] -- HOST-MAPPED pinned words (system-scope
 * stores, seq last with release order): the host polls notify[1] == seq and reads the size without a stream
 * synchronisation or a copy, so work queued behind this launch is not drained by the size read.  P > 0. */
int nsim_pack_infos_from_n_notify(const int64_t* n, int64_t P, int64_t* pack_infos, int64_t* total, int64_t cap,
                                  int64_t* notify, int64_t seq, void* stream);
/* ``ray_query_cfg.query_param.upsample_on_marched_only`` (the [R'] hit set of the reference's volume buffers,
 * single_volume_renderer.py:209-220,289-300: ``rays_inds_hit`` / ``pack_infos_hit`` cover the rays that produced samples): from
 * the occupancy-march counts n [R]: live_rank [R] = q >= 0 for the q-th ray with n > 0 ("live"), ~q < 0 for a ray with n == 0
 * (q live rays precede it); live_idx [R] (may be NULL) = the live rays in order, zero-filled past R'; cnts [8] (device) =
 * {R', sum(n) + R' C, R' nf0, R' nf1, R' nf2, R' nf3, sum(n), 0} -- the device-side point counts of the sampling pass's SDF
 * queries; notify (may be NULL): host-mapped words receiving (R', seq) as nsim_pack_infos_from_n_notify.  pack_infos [R,2]
 * (may be NULL) + total + cap + notify_total / seq_total: the outputs of nsim_pack_infos_from_n[_notify](n, cap) from the same
 * launch (the sampling pass needs both scans of the same counts). */
int nsim_live_rank(const int64_t* n, int64_t R, int C, int nf0, int nf1, int nf2, int nf3, int64_t* live_rank,
                   int64_t* live_idx, int64_t* cnts, int64_t* notify, int64_t seq, int64_t* pack_infos, int64_t* total,
                   int64_t cap, int64_t* notify_total, int64_t seq_total, void* stream);
/* packed_sum(x [S,C], pack_infos) -> out [P,C]   (single_volume_renderer.py:84-101) */
int nsim_packed_sum(const float* x, int C, const int64_t* pack_infos, int64_t P, float* out, void* stream);
/* out[s,c] = x[s,c] (op) per_pack[p, c or 0]; op 0 mul, 1 div, 2 add, 3 sub; x may be NULL (treated as 1 for
 * mul => broadcast, the backward of packed_sum).  packed_div: single_volume_renderer.py:86 */
int nsim_packed_binary(const float* x, int C, const float* per_pack, int Cp, const int64_t* pack_infos,
                       int64_t P, int op, float* out, void* stream);
/* packed_geq / packed_leq / packed_lt / packed_gt (op 0..3) -> uint8 mask (app/loss/lidar.py:102-110) */
int nsim_packed_cmp(const float* x, const float* per_pack, const int64_t* pack_infos, int64_t P, int op,
                    uint8_t* out, void* stream);
/* packed_matmul(x [S,3], rot [P,3,3]) : out[s] = rot[p] @ x[s] (transpose != 0: rot[p]^T @ x[s], the backward)
 * (app/renderers/utils.py:25) */
int nsim_packed_matmul3(const float* x, const float* rot, const int64_t* pack_infos, int64_t P,
                        int transpose, float* out, void* stream);
/* packed_sort(x, pack_infos) -> (sorted, GLOBAL indices)  (buffer_compose_renderer.py:1043-1047) */
int nsim_packed_sort(const float* x, const int64_t* pack_infos, int64_t P, float* sorted,
                     int64_t* indices, void* stream);
/* The collect + sort steps of BufferComposeRenderer.ray_query (code_multi/app/renderers/buffer_compose_renderer.py:648-695) in
 * one launch.  K <= 64 sources (the per-object volume buffers, in the reference's collect order); source k: depths t [S_k] in
 * packs ``pack_infos`` [P_k, 2] that live on the rays ``rays_inds`` [P_k] (ascending, one pack per ray: the reference's
 * ``rays_inds_collect`` / ``pack_infos_collect``).  total_pack_infos [N, 2]: (start, count) of every one of the N rays in the merged
 * buffer (count = the ray's samples over all sources; 0: no samples).  Outputs: t_sorted [S] -- every ray's samples ordered
 * by depth, ties in collect order (the reference's stable packed_sort of the concatenation) -- and, per source, dst [S_k]: the
 * position of each of its samples in that order (= ranks[pidx_in_total] of the reference's bookkeeping), so an attribute is
 * placed with ONE indexed store per source and an object's weights in the scene are vw[dst]. */
typedef struct NsimComposeSrc {
  const float* t;
  const int64_t* rays_inds;
  const int64_t* pack_infos;
  int64_t P;
  int64_t* dst;
} NsimComposeSrc;
int nsim_compose_collect_sort(const NsimComposeSrc* src, int32_t K, const int64_t* total_pack_infos, int64_t N,
                              float* t_sorted, void* stream);
/* interleave_linstep(start [P], n [P], step): pack_infos = get_pack_infos_from_n(n)  (buffer_compose_renderer.py:1036) */
int nsim_interleave_linstep(const int64_t* start, const int64_t* pack_infos, int64_t P, int64_t step,
                            int64_t* out, void* stream);
/* merge_two_packs_sorted (single_volume_renderer.py:341-344): packs of a / b live on rays slot_a / slot_b
 * (index into the union ray list, ascending); pack_infos_out [U,2] given (from the summed counts).
 * a-first on ties. */
int nsim_merge_two_packs(const float* va, const int64_t* pia, const int64_t* slot_a, int64_t Pa,
                         const float* vb, const int64_t* pib, const int64_t* slot_b, int64_t Pb,
                         const int64_t* pack_infos_out, int64_t U, int64_t* pidx_a, int64_t* pidx_b,
                         void* stream);

/* --------------------------------------------------------------- graphics.nerf / compositing */
/* packed_alpha_to_vw (single_volume_renderer.py:79-83): vw_i = alpha_i * prod_{j<i}(1-alpha_j+1e-10).
 * trans (may be NULL) receives the transmittance T_i, saved for the backward. */
int nsim_alpha_to_vw_fwd(const float* alpha, const int64_t* pack_infos, int64_t P, float* vw, float* trans,
                         void* stream);
int nsim_alpha_to_vw_bwd(const float* alpha, const float* trans, const float* vw, const float* dvw,
                         const int64_t* pack_infos, int64_t P, float* dalpha, void* stream);
/* The differentiable tail of a training step in ONE launch (no reference counterpart: the reference evaluates these with a
 * dozen torch / nr3d_lib calls -- ``_volume_integration`` app/renderers/single_volume_renderer.py:73-102, the photometric mse
 * app/loss/photometric.py:88-146, the eikonal term app/loss/eikonal.py:185-253 -- and autograd's backward of each): for the P
 * hit rays (packs ``pack_infos``, image row ``out_idx[p]`` or p) sdf -> alpha -> visibility weights -> mask / depth / rgb / normal
 * images; loss = mean((rgb image - gt)^2) over ALL N rays (rays outside the packs render black) + w_eikonal (mean over the S
 * render samples + mean over the M free points of (|nablas| - 1)^2); acc[0..2] += (mse, eikonal_render, eikonal_free) -- the mse
 * as a sum of squares only (e^2 of the pack rays + gt^2 of the others; out_idx must be ASCENDING: membership by search); and the
 * backward of all of it: dalpha [S], dsdf [S] (rows S.. are the caller's zeros), drgb [S,3], dnablas [S+M,3] (eikonal part
 * only), d_ln_inv_s += (may be NULL).  alpha / vw / trans [S] are written (the later backward launches read them). */
int nsim_render_head(const float* sdf, const float* ln_inv_s, float ln_inv_s_factor, float forward_inv_s, const float* t,
                     const float* rgb, const float* nablas, const int64_t* pack_infos, int64_t P, int normalized_depth,
                     const float* gt, int64_t N, int64_t S, int64_t M, float w_eikonal, const int64_t* out_idx, float* alpha,
                     float* vw, float* trans, float* mask, float* depth, float* rgb_out, float* nrm_out, float* acc,
                     float* dalpha, float* dsdf, float* drgb, float* dnablas, float* d_ln_inv_s, void* stream);
/* Fused SingleVolumeRenderer._volume_integration (single_volume_renderer.py:73-102): alpha -> vw -> mask,
 * depth (optionally normalised), rgb, normals per ray.  rgb / nrm (and their outputs) may be NULL.
 * out_idx [P] (may be NULL): row of the per-ray outputs that pack p writes / reads -- the scatter of the hit rays into
 * the all-rays images (``rendered[k][rays_inds_hit] = ...``, :88-101) fused into the same launch; the caller zeroes
 * the images. */
int nsim_composite_fwd(const float* alpha, const float* t, const float* rgb, const float* nrm,
                       const int64_t* pack_infos, int64_t P, int normalized_depth, float* vw, float* trans,
                       float* mask, float* depth, float* rgb_out, float* nrm_out, const int64_t* out_idx,
                       void* stream);
int nsim_composite_bwd(const float* alpha, const float* trans, const float* vw, const float* t,
                       const float* rgb, const float* nrm, const int64_t* pack_infos, int64_t P,
                       int normalized_depth, const float* mask, const float* depth, const float* dmask,
                       const float* ddepth, const float* drgb_out, const float* dnrm_out,
                       const float* dvw_ext, float* dalpha, float* drgb, float* dnrm, const int64_t* out_idx,
                       void* stream);
/* NeuS sdf -> opacity (NeusRendererMixin): alpha_i for interval (i,i+1), 0 at the
 * last sample of each pack.  inv_s = exp(ln_inv_s[0] * ln_inv_s_factor) unless forward_inv_s > 0. */
int nsim_neus_alpha_fwd(const float* sdf, const int64_t* pack_infos, int64_t P, const float* ln_inv_s,
                        float ln_inv_s_factor, float forward_inv_s, float* alpha, void* stream);
int nsim_neus_alpha_bwd(const float* sdf, const float* dalpha, const int64_t* pack_infos, int64_t P,
                        const float* ln_inv_s, float ln_inv_s_factor, float forward_inv_s, float* dsdf,
                        float* d_ln_inv_s, void* stream);
/* nsim_neus_alpha_fwd + nsim_composite_fwd in one launch (identical arithmetic; alpha [S] is written for the
 * backward passes above). */
int nsim_neus_composite_fwd(const float* sdf, const float* ln_inv_s, float ln_inv_s_factor, float forward_inv_s,
                            const float* t, const float* rgb, const float* nrm, const int64_t* pack_infos, int64_t P,
                            int normalized_depth, float* alpha, float* vw, float* trans, float* mask, float* depth,
                            float* rgb_out, float* nrm_out, const int64_t* out_idx, void* stream);

/* ------------------------------------------------------------------------------ ray generation */
/* Camera._get_selected_rays_from_ixy (app/resources/observers/cameras.py:281-310) */
int nsim_raygen_pinhole(const float* xy, const int64_t* fidx, const float* intr /*[V,3,3]*/,
                        const float* c2w /*[V,4,4]*/, const int64_t* WH /*[V,2]*/, int64_t N, int snap,
                        float* rays_o, float* rays_d, void* stream);
/* Its backward w.r.t. the poses (pose refinement, LearnableParams -> refined c2w,
 * withmask_withlidar_joint.240219.yaml:338-352): d_c2w [V,4,4] (initialised by the caller) += the gradients of
 * rays_o = T and rays_d = R l / |R l|; d_rays_o / d_rays_d [N,3] may each be NULL. */
int nsim_raygen_pinhole_bwd(const float* xy, const int64_t* fidx, const float* intr, const float* c2w,
                            const int64_t* WH, int64_t N, int snap, const float* d_rays_o, const float* d_rays_d,
                            float* d_c2w, void* stream);
/* The same two with the OpenCV camera model (``camera_model: opencv`` -> ``OpenCVCameraMatHW.lift``,
 * app/resources/observers/cameras.py:84-87; the street configs' ``consider_distortion: true``,
 * withmask_withlidar_joint.240219.yaml:141, Waymo calibration (k1, k2, p1, p2, k3),
 * dataio/autonomous_driving/waymo/preprocess.py:172): distortion [V,5]; the undistorted direction comes from n_iters
 * rounds of the fixed-point iteration of cv::undistortPoints (5 there).  Implementation in the absent nr3d_lib:
 * semantics fixed here. */
int nsim_raygen_opencv(const float* xy, const int64_t* fidx, const float* intr, const float* distortion /*[V,5]*/,
                       int n_iters, const float* c2w, const int64_t* WH, int64_t N, int snap, float* rays_o,
                       float* rays_d, void* stream);
int nsim_raygen_opencv_bwd(const float* xy, const int64_t* fidx, const float* intr, const float* distortion, int n_iters,
                           const float* c2w, const int64_t* WH, int64_t N, int snap, const float* d_rays_o,
                           const float* d_rays_d, float* d_c2w, void* stream);
/* ... and with the fisheye camera model (``camera_model: fisheye`` -> ``FisheyeCameraMatHW.lift``,
 * app/resources/observers/cameras.py:88-92; dataio/autonomous_driving/custom/custom_autodrive_dataset.py:512,581): the
 * OpenCV fisheye (Kannala-Brandt equidistant) model theta_d = theta (1 + k1 theta^2 + k2 theta^4 + k3 theta^6 + k4 theta^8)
 * the reference applies in app/resources/observers/fisheye.py:36-42; distortion [V,4]; theta from n_iters Newton rounds
 * (cv::fisheye::undistortPoints runs 10), direction (sin theta x_d / theta_d, sin theta y_d / theta_d, cos theta).
 * Implementation in the absent nr3d_lib: semantics fixed here. */
int nsim_raygen_fisheye(const float* xy, const int64_t* fidx, const float* intr, const float* distortion /*[V,4]*/,
                        int n_iters, const float* c2w, const int64_t* WH, int64_t N, int snap, float* rays_o,
                        float* rays_d, void* stream);
int nsim_raygen_fisheye_bwd(const float* xy, const int64_t* fidx, const float* intr, const float* distortion, int n_iters,
                            const float* c2w, const int64_t* WH, int64_t N, int snap, const float* d_rays_o,
                            const float* d_rays_d, float* d_c2w, void* stream);
/* AABBSpace.ray_test (call site single_volume_renderer.py:235-238): far < 0 means "no far". */
int nsim_aabb_ray_test(const float* rays_o, const float* rays_d, int64_t N, const NsimOccMeta* meta,
                       float near, float far, float* near_out, float* far_out, uint8_t* hit, void* stream);

/* ------------------------------------------------------------------------ occupancy + sampling */
/* OccGridEma update (lotd_neus.dtu.230814.yaml:140-155): val = max(val*decay, 4 sig(s sdf)(1-sig)) */
int nsim_occ_decay(float* val, int64_t nvox, float decay, void* stream);
int nsim_occ_update(float* val, const float* pts, const float* sdf, int64_t n, const NsimOccMeta* meta,
                    float inv_s, void* stream);
int nsim_occ_pack_bits(const float* val, int64_t nvox, float thre, uint32_t* bits, void* stream);
/* ``accel_cfg.update_from_samples_cfg: {}`` (lotd_neus.dtu.230814.yaml:158; hook app/resources/asset_bank.py:291-298): the
 * SDFs the sampling pass of a training step computes anyway are max-folded into the value grid (no decay: the periodic
 * refresh decays and re-thresholds).  n_dev (may be NULL): valid points = min(n, *n_dev + n_add), 0 if that exceeds n. */
int nsim_occ_collect(float* val, const float* pts, const float* sdf, int64_t n, const int64_t* n_dev, int64_t n_add,
                     const NsimOccMeta* meta, float inv_s, void* stream);
/* OccGridAccel.ray_march (march_cfg{step_size,max_steps}): lattice t_k = near + (k + jitter) * step.
 * ray_word_off (may be NULL): batched occupancy grid (``accel_cfg{type: occ_grid_batched}``,
 * code_multi/.../no_fg_occ.221218.yaml:369-377) -- offset in 32-bit words of ray r's instance inside ``bits``. */
int nsim_march_count(const float* rays_o, const float* rays_d, const float* near, const float* far,
                     const float* jitter, int64_t R, const uint32_t* bits, const int64_t* ray_word_off,
                     const NsimOccMeta* meta, float step, int max_steps, int64_t* counts, void* stream);
int nsim_march_emit(const float* rays_o, const float* rays_d, const float* near, const float* far,
                    const float* jitter, int64_t R, const uint32_t* bits, const int64_t* ray_word_off,
                    const NsimOccMeta* meta, float step, int max_steps, const int64_t* pack_infos, float* t_out,
                    void* stream);
/* coarse_step_cfg{step_mode: linear}: t = near + (far-near) * ((i + u)/C); jitter_c NULL => u = 0.5 */
/* live_rank (here and in the three entry points below; may be NULL = every ray is live, dense [R, n] layout): the ranks of
 * nsim_live_rank -- only live rays get samples, and every per-live-ray array (out here; t_b / v_b, t_new, x_new below) is
 * indexed by the rank q instead of r, i.e. holds R' n compact entries. */
int nsim_coarse_depths(const float* near, const float* far, const float* jitter_c, int64_t R, int C,
                       float* out, const int64_t* live_rank, void* stream);
/* One NeuS up-sampling stage (num_fine[i], upsample_inv_s * factor[i], upsample_use_estimate_alpha).
 * scratch: float [S]. t_new: [R, n_fine] ascending.  x_new [R, n_fine, 3] (may be NULL; needs rays_o / rays_d [R,3]):
 * positions o + t d of the new samples, so that the level-major query reads 12 B per point instead of re-deriving
 * the point from (ridx, t, o, d) once per XCD. */
int nsim_upsample_stage(const float* t, const float* sdf, const int64_t* pack_infos, int64_t R, float inv_s,
                        int n_fine, int use_estimate_alpha, float* scratch, float* t_new, const float* rays_o,
                        const float* rays_d, float* x_new, const int64_t* live_rank, void* stream);
/* Sorted merge of packed (t_a, v_a) with batched (t_b, v_b) [R,nb]; packs must tile the arrays in order.
 * v_a / v_b / v_out may be NULL. pack_infos_out [R,2] is written; ridx_out [S_out] (may be NULL) receives the
 * ray (pack) index of every merged sample; x_out [S_out, 3] (may be NULL; needs rays_o / rays_d) their positions. */
int nsim_merge_sorted(const float* t_a, const float* v_a, const int64_t* pack_infos_a, const float* t_b,
                      const float* v_b, int64_t R, int nb, float* t_out, float* v_out,
                      int64_t* pack_infos_out, int64_t* ridx_out, const float* rays_o, const float* rays_d,
                      float* x_out, const int64_t* live_rank, void* stream);
/* nsim_merge_sorted of up-sampling stage k followed, in the same launch, by nsim_upsample_stage of stage k + 1 on the merged
 * samples (both one wave per ray; v_a / v_b / v_out = the no-grad SDFs, required): t_new [R, n_fine] (+ x_new) are the next
 * stage's draws; scratch [>= merged sample count] as nsim_upsample_stage.  Same values as the two calls. */
int nsim_merge_upsample(const float* t_a, const float* v_a, const int64_t* pack_infos_a, const float* t_b, const float* v_b,
                        int64_t R, int nb, float* t_out, float* v_out, int64_t* pack_infos_out, int64_t* ridx_out, float inv_s,
                        int n_fine, int use_estimate_alpha, float* scratch, float* t_new, const float* rays_o,
                        const float* rays_d, float* x_new, const int64_t* live_rank, void* stream);

/* ``query_mode: march_occ_multi_upsample_compressed`` (lotd_neus.dtu.230814.yaml:157): from the no-grad SDF of all
 * samples keep those that bound an interval with visibility weight > thre.  count -> counts [R]; emit (given
 * pack_infos_out from the counts) -> compacted t_out / ridx_out; emit also appends ``tail_n`` (>= 0) zero-depth samples
 * on the pseudo-rays R .. R+tail_n-1 behind the kept set (the free eikonal points of the with-grad query), so
 * t_out / ridx_out hold total + tail_n entries. */
int nsim_compress_count(const float* sdf, const int64_t* pack_infos, int64_t R, const float* ln_inv_s,
                        float ln_inv_s_factor, float forward_inv_s, float thre, int64_t* counts, void* stream);
int nsim_compress_emit(const float* sdf, const float* t, const int64_t* pack_infos, int64_t R, const float* ln_inv_s,
                       float ln_inv_s_factor, float forward_inv_s, float thre, const int64_t* pack_infos_out,
                       float* t_out, int64_t* ridx_out, int64_t tail_n, void* stream);

/* ------------------------------------------------------------------- LoTD encoding (standalone) */
/* LoTDEncoding.forward / forward_dydx (inspect_rendering.py:468-474): x [S,3] in [-1,1], grid fp16.
 * out f32 [S, L*2]; dydx (may be NULL) f32 [S, L*2, 3] = d out / d x. */
int nsim_lotd_fwd(const float* x, const void* grid_f16, const NsimLotdMeta* meta, int64_t S, float* out,
                  float* dydx, void* stream);
/* LoTD backward: dgrid (f32 [n_params], accumulated with atomics) += d L/d out . d out/d grid
 *   + (if dL_ddydx != NULL) d L / d dydx . d dydx / d grid  (the second-order path used by nablas). */
int nsim_lotd_bwd(const float* x, const float* dL_dout, const float* dL_ddydx, const NsimLotdMeta* meta,
                  int64_t S, float* dgrid, void* stream);

/* ------------------------------------------------------- permutohedral-lattice encoding */
/* nr3d_lib.models.grid_encodings.permuto.PermutoEncoding(in_dim, permuto_auto_compute_cfg{type: multi_res, coarsest_res,
 * finest_res, n_levels, n_feats, log2_hashmap_size, apply_random_shifts_per_level}) -- call sites
 * app/models/single/neus.py:64-76 (PermutoNeuSObj), docs/exps/exp_permuto_3d_modulated.py:52-60,
 * code_multi/configs/exps/fg_neus=permuto/all_occ.240201.yaml:438-446.  The implementation is in the absent nr3d_lib; the
 * kernels follow the published algorithm (Adams et al. 2010, Rosu & Behnke 2023) as restated in oracle/permuto.py:
 * level l: cf_i = (x_i + shift[l][i]) * scale[l][i] with scale[l][i] = res_l / sqrt((i+1)(i+2)), elevation, nearest
 * remainder-0 point, ranks, barycentric weights of the in_dim + 1 simplex vertices, vertex hash
 * k <- (k + key_i) * 2531011 (uint32) over the first in_dim key coordinates, modulo hashmap_size.
 * Table: fp16, level l at [l * hashmap_size * 2, (l+1) * hashmap_size * 2), two features per entry. */
typedef struct NsimPermutoMeta {
  int32_t in_dim;                      /* 2..8 (3 = positions; 3 + k = positions with a k-dim condition concatenated) */
  int32_t num_levels;                  /* 1..32 */
  int32_t n_feats;                     /* must be 2 */
  uint32_t hashmap_size;               /* entries per level, a power of two */
  float scale[NSIM_MAX_LEVELS][8];
  float shift[NSIM_MAX_LEVELS][8];     /* ``apply_random_shifts_per_level`` (zeros when off) */
} NsimPermutoMeta;
/* PermutoEncoding.forward / forward_dydx: x [S,in_dim] -> out f32 [S, L*2]; dydx (may be NULL) f32 [S, L*2, in_dim]. */
int nsim_permuto_fwd(const NsimPermutoMeta* meta, const void* grid_f16, const float* x, int64_t S, float* out, float* dydx,
                     void* stream);
/* backward w.r.t. the table: dgrid (f32 [L * hashmap_size * 2], accumulated with atomics) += dL/dout . dout/dgrid.
 * (dL/dx = sum dL/dout . dydx is a host-side contraction of the forward's dydx.) */
int nsim_permuto_bwd(const NsimPermutoMeta* meta, const float* x, int64_t S, const float* dL_dout, float* dgrid,
                     void* stream);
/* The NeuS field's front end (PermutoNeuSObj; GenerativePermutoConcat with z): positions x [S,3] or rays
 * (rays_o, rays_d [R,3], t [S], ridx [S]); z [R, in_dim - 3] (may be NULL = zeros; needs ridx) is the per-ray condition
 * concatenated to the position.  Writes LEVEL-MAJOR planes in the layout of nsim_lotd_gather_lm / nsim_field_fwd, so
 * that nsim_field_sdf (feat_planes) and nsim_field_fwd / _bwd_sdf (h_planes, J_planes; pass grid_f16 = NULL there: "the
 * planes are already gathered") run unchanged on a permutohedral model: exactly one of
 *   feat_planes [NL][P] (f16x2 scaled by 1024 when feat_f32 = 0, f32x2 when 1)      -- no-grad query, or
 *   h_planes [NL][P][2] + J_planes [NL][P][2][3], P = NSIM_PLANE_PITCH(S)            -- with-grad query; J_planes in the
 *       element type the decoders' meta asks for (nsim_jplane_elem_bytes): feat_f32 = 0 writes f16 (2 bytes), 1 writes f32
 * (NL = 16 for <= 16 levels, else 32; levels past num_levels are not written).  n_dev / n_add as nsim_lotd_gather_lm. */
int nsim_permuto_gather(const NsimPermutoMeta* meta, const void* grid_f16, const float* x, const float* rays_o,
                        const float* rays_d, const float* t, const int64_t* ridx, const float* z, int64_t S,
                        const int64_t* n_dev, int64_t n_add, void* feat_planes, int feat_f32, float* h_planes,
                        void* J_planes, void* stream);
/* Backward to the table from the decoders' hand-off planes (nsim_field_bwd_sdf): dgrid[v][f] += w_v dL/dh[f]
 * + g[f] (dw_v/dx . gn)  -- gn [S,3] (may be NULL, then g_planes may be NULL too) is the total dL/dnablas; the weights
 * are piecewise linear in x, so this is the whole second-order term. */
int nsim_permuto_scatter(const NsimPermutoMeta* meta, const float* x, const float* rays_o, const float* rays_d,
                         const float* t, const int64_t* ridx, const float* z, int64_t S, const float* dh_planes,
                         const float* g_planes, const float* gn, float* dgrid, void* stream);
/* Backward to the CONDITION of a conditioned field (in_dim > 3; GenerativePermutoConcat's learned per-instance codes,
 * app/models/shared/batched_neus.py:295-407 ``z_ins_all``): dz[ray][c] += sum over the ray's samples, levels and features of
 * dL/dh . dh/dz_c (dz [R, in_dim - 3], caller-zeroed; z and ridx required).  The normals carry no z term: dh/dx is constant
 * inside a simplex. */
int nsim_permuto_dz(const NsimPermutoMeta* meta, const void* grid_f16, const float* x, const float* rays_o,
                    const float* rays_d, const float* t, const int64_t* ridx, const float* z, int64_t S,
                    const float* dh_planes, float* dz, void* stream);

/* ------------------------------------------------------- fused NeuS field (LoTD + MLPs, MFMA) */
/* Network description (host struct).  LoTDNeuSModel = LoTDSDF + RadianceNet
 * (app/models/single/neus.py:24-62; lotd_neus.dtu.230814.yaml:92-139). */
typedef struct NsimFieldMeta {
  NsimLotdMeta lotd;     /* 1..32 levels x 2 feats (<= 64 input features; W1 is [64 x 2 num_levels]); more than 16
                          * levels: level-major planes of 32 levels, the planes arguments are then mandatory */
  int32_t sdf_D;         /* hidden layers of the SDF decoder: 1 or 2 (width 64, softplus beta) */
  int32_t precision;     /* 0: fp16 MFMA (v_mfma_f32_32x32x16_f16), 1: exact f32 MFMA (32x32x2 f32) */
  float softplus_beta;   /* 100; a negative value selects relu (decoder_cfg.activation: relu) */
  int32_t embed_E;       /* width of the embedded-position block appended to the SDF decoder's input: 0 = none, else
                          * 3 + 6 n_frequencies <= 63 (``surface_cfg.extra_pos_embed_cfg{type: sinusoidal_legacy}`` of the
                          * StyleLoTD Vehicle block, code_multi/configs/exps/fg_neus=hyper_lotd/no_fg_occ.221218.yaml:319-321).
                          * W1 is then [64 x (2 num_levels + embed_E)] and the packed first-layer matrices carry two more
                          * 32-input MFMA chunks; nsim_field_fwd (level-major path) and nsim_wide_bwd_sdf read it */
} NsimFieldMeta;

/* size in bytes of the packed-fragment weight buffer for a given meta */
int64_t nsim_field_wpack_bytes(const NsimFieldMeta* meta);
/* Re-pack f32 master weights into MFMA A-fragment order (once per optimizer step).
 * sdf_w: [W1 (64 x 2 num_levels), (W2 (64x64)), Wout (1x64)] concatenated row-major; sdf_b likewise [64,(64),1];
 * rad_w: [Wr1 (64x26), Wr2 (64x64), Wr3 (3x64)], rad_b [64,64,3]. */
int nsim_field_pack_weights(const NsimFieldMeta* meta, const float* sdf_w, const float* sdf_b,
                            const float* rad_w, const float* rad_b, void* wpack, void* stream);
/* Two packs of the same master weights in ONE launch (a model whose sampling pass runs at another precision than its field
 * keeps two: every optimizer step re-packs both). */
int nsim_field_pack_weights2(const NsimFieldMeta* meta_a, void* wpack_a, const NsimFieldMeta* meta_b, void* wpack_b,
                             const float* sdf_w, const float* sdf_b, const float* rad_w, const float* rad_b, void* stream);
/* No-grad SDF query (model.query_sdf / forward_sdf; inspect_rendering.py:120-128).
 * Points are x[s] (if x != NULL) or rays_o[ridx[s]] + t[s] * rays_d[ridx[s]].
 * feat_planes == NULL: one fused point-major kernel (gather + decoder).
 * feat_planes != NULL: decoder only, on the level-major feature planes written by nsim_lotd_gather_lm for the same
 * points (x / rays / grid are then unused and may be NULL).  Same values either way.
 * Speculatively sized buffers (level-major path): S is the CAPACITY (the plane pitch is NSIM_PLANE_PITCH(S)); when n_dev != NULL the
 * number of valid points is *n_dev + n_add, read on the device (0 if that exceeds S: the caller under-sized its
 * buffers and redoes the pass) -- the host never learns the size of the marched sample set before launching its
 * first query (one host sync less per step).
 * occ_val != NULL (needs x and occ_meta): the SDFs are also folded into the occupancy value grid in the same launch --
 * nsim_occ_collect's update (``update_from_samples_cfg``) without a pass of its own. */
int nsim_field_sdf(const NsimFieldMeta* meta, const void* grid_f16, const void* wpack, const float* x,
                   const float* rays_o, const float* rays_d, const float* t, const int64_t* ridx,
                   const int64_t* ray_goff, int64_t S, const int64_t* n_dev, int64_t n_add, float* sdf,
                   const void* feat_planes, float* occ_val, const NsimOccMeta* occ_meta, float occ_inv_s, void* stream);
/* Level-major LoTD gather of the no-grad query (the encoding half of forward_sdf): feat_planes [NLP][P] (NLP = 16 for <= 16 levels, 32 above;
 * P = NSIM_PLANE_PITCH(S): a 32-point tile of a level is one aligned 128 B | 256 B piece, which nsim_field_sdf copies
 * straight into LDS, global_load_lds_dwordx4, one tile ahead) of
 * (fp16 x 2, pre-scaled for the fp16 MFMA decoder | f32 x 2) = NLP * P * (4 | 8) bytes, caller-owned.  Every wave
 * walks the levels in one order and the levels are dealt to the XCDs, so a level's table is read through ONE L2. */
int nsim_lotd_gather_lm(const NsimFieldMeta* meta, const void* grid_f16, const float* x, const float* rays_o,
                        const float* rays_d, const float* t, const int64_t* ridx, const int64_t* ray_goff, int64_t S,
                        const int64_t* n_dev, int64_t n_add, void* feat_planes, void* stream);
/* Batched / multi-instance models (``batched_ray_query`` + ``set_condition``,
 * app/renderers/buffer_compose_renderer.py:222-265, app/models/shared/batched_neus.py:380-407): the tables of all
 * instances live in ONE flat tensor and ``ray_goff[r]`` (may be NULL) is the offset, in scalars (even), of ray r's
 * instance; every kernel that touches the table takes it next to ``ridx`` (required then, also with ``x``).
 *
 * With-grad query: forward_sdf_nablas + radiance. v: view dirs per sample taken from
 * rays_d[ridx]; h_appear [R,4] per ray (may be NULL => zeros). Outputs sdf [S], nablas [S,3], rgb [S,3]
 * (rgb may be NULL: with_rgb=False, code_single/tools/train.py:896-902).
 * h_planes [NLP,P,2] (f32) / J_planes [NLP,P,2,3] (f16 | f32: nsim_jplane_elem_bytes) with the pitch P = NSIM_PLANE_PITCH(S) = S rounded up to 32 -- every 32-point
 * tile of a level is then one 16-byte-aligned 256 B / 768 B piece, which the decoder kernels prefetch straight into LDS
 * (global_load_lds_dwordx4) -- (both or neither; NLP = 16 for <= 16 levels, 32 above): when given, the gathered features and their
 * derivative w.r.t. x are saved level-major for the backward launches (which then never gather again).
 * n_dev (may be NULL; needs the planes): the launch is sized for a CAPACITY S while the number of valid points,
 * n_dev[0] + n_add <= S, is read on the device -- the training step queues this launch before the host has read the
 * size of the kept sample set (a count above the capacity touches nothing; the caller redoes the launch).
 * wpack == NULL (needs grid_f16 and the planes): GATHER ONLY -- the planes are written, no decoder runs (the caller's
 * decoder is another one: nsim_wide_fwd); sdf / nablas / rgb are then unused. */
#define NSIM_PLANE_PITCH(S) ((((int64_t)(S)) + 31) & ~(int64_t)31)
int nsim_field_fwd(const NsimFieldMeta* meta, const void* grid_f16, const void* wpack, const float* x,
                   const float* rays_o, const float* rays_d, const float* t, const int64_t* ridx,
                   const int64_t* ray_goff, const float* h_appear, int64_t S, float* sdf, float* nablas, float* rgb,
                   float* h_planes, void* J_planes, const int64_t* n_dev, int64_t n_add, void* stream);
/* Element size of the dh/dx planes (J_planes) for this meta: 2 (f16: the fp16 field mode -- 192 B per point instead of 384,
 * the dominant plane traffic of the with-grad gather and of the two decoders that read them) or 4 (f32: the f32 mode).
 * J_planes holds NLP * P * 6 such elements. */
int nsim_jplane_elem_bytes(const NsimFieldMeta* meta);
/* Optional scratch for the weight-gradient flush of the backward launches (1) and (2) below.  Every workgroup of those
 * launches ends by adding its partial dW / db into the same few thousand floats; hundreds of same-address atomics per
 * address serialise in L2 (measured: ~55 us of a 100 us radiance backward).  With a caller-owned, ZEROED buffer of
 * ``floats`` floats registered for (the calling thread's current device, ``stream``) (>= 16 x 8448 covers every decoder shape), workgroup b adds into replica
 * b % 16 of it and a small second launch folds the replicas into dW / db and zeroes the buffer again: the buffer is zero
 * whenever no launch of that stream is in flight.  buf = NULL unregisters.  Without a registration (or with
 * NSIM_GRAD_REPLICAS=1) the launches add into dW / db directly, as before.  The reference has no counterpart: its
 * autograd backward of the tcnn-style MLP reduces weight gradients inside one GEMM. */
int nsim_set_grad_scratch(float* buf, int64_t floats, void* stream);
/* Backward of nsim_field_fwd = three launches (each its own entry point so that callers can time / overlap them):
 *
 * (1) radiance branch: given dL/drgb [S,3], the saved forward nablas_fwd / rgb_fwd [S,3] and the upstream
 *     dL/dnablas (may be NULL): accumulates drad_w / drad_b (layouts of nsim_field_pack_weights), dh_appear [R,4]
 *     (may be NULL) and writes gn_out [S,3] = dL/dnablas + d(radiance)/d nablas.
 *     Pose refinement (LearnableParams, withmask_withlidar_joint.240219.yaml:338-352 -- the rays carry gradients):
 *     dx [S,3] (may be NULL) is SET to the radiance net's gradient w.r.t. the sample position, dv [S,3] (may be
 *     NULL) to its gradient w.r.t. the view direction (through SH-4). */
int nsim_field_bwd_rad(const NsimFieldMeta* meta, const void* wpack, const float* nablas_fwd, const float* rgb_fwd,
                       const float* x, const float* rays_o, const float* rays_d, const float* t, const int64_t* ridx,
                       const float* h_appear, int64_t S, const float* dnablas, const float* drgb, float* gn_out,
                       float* drad_w, float* drad_b, float* dh_appear, float* dx, float* dv, void* stream);
/* (2) SDF-decoder branch on the saved h / J planes: given dL/dsdf [S] and the total dL/dnablas gn [S,3] (either may
 *     be NULL) accumulates dsdf_w / dsdf_b -- including the double-backward terms of nablas w.r.t. the decoder
 *     weights (app/loss/eikonal.py:216-251) -- and writes the hand-off planes dh_planes = dL/dh and
 *     g_planes = d sdf/d h, both [NLP,S,2] (both or neither).  dx [S,3] (may be NULL; initialised by the caller or
 *     by (1)) += (dh/dx)^T dL/dh, the first-order position gradient (LoTD ``dL/dx``).
 *     plane_pitch: pitch of h_planes / J_planes when the forward ran at a capacity above S (0 = NSIM_PLANE_PITCH(S)). */
int nsim_field_bwd_sdf(const NsimFieldMeta* meta, const void* wpack, const float* h_planes, const void* J_planes,
                       int64_t S, const float* dsdf, const float* gn, float* dh_planes, float* g_planes,
                       float* dsdf_w, float* dsdf_b, float* dx, int64_t plane_pitch, void* stream);
/* The SDF decoder with an embedded-position block appended to its input (meta->embed_E > 0):
 * ``surface_cfg.extra_pos_embed_cfg{type: sinusoidal_legacy, n_frequencies: N}`` of the StyleLoTD Vehicle block
 * (code_multi/configs/exps/fg_neus=hyper_lotd/no_fg_occ.221218.yaml:319-321).  Input = [2 num_levels features | x_n (3),
 * sin(2^k x_n) (3), cos(2^k x_n) (3), k = 0..N-1], x_n = the AABB-normalised position in [-1, 1] (meta->lotd.x_scale /
 * x_shift); sdf_w = [W1 (64 x FIN), (W2 (64 x 64)), w_head (64)], FIN = 2 num_levels + 3 + 6 N.
 *   with-grad forward: nsim_field_fwd on the level-major planes (h_planes / J_planes given) with the meta's wpack -- the first
 *     layer contracts over the feature chunks + two embedding chunks on the matrix cores (k_field<.., NE = 2>, csrc/field.hip);
 *   backward of the SDF branch: nsim_wide_bwd_sdf = nsim_field_bwd_sdf + the sample positions (the embedding and its
 *     x-derivative are regenerated in the kernel): k_field_bwd_j<.., NE = 2>; no dL/dx output;
 *   no-grad query (sampling, occupancy): nsim_wide_sdf on the f32 MASTER weights, f32 arithmetic on the VALU
 *     (csrc/wide_field.hip; meta->precision is not read) over f32 feature planes of nsim_lotd_gather_lm.
 * The hand-off planes of the backward feed nsim_lotd_scatter, the radiance backward is nsim_field_bwd_rad.  Points, n_dev /
 * n_add, plane pitches: as for nsim_field_sdf / _fwd / _bwd_sdf. */
int nsim_wide_sdf(const NsimFieldMeta* meta, int32_t n_freq, const float* sdf_w, const float* sdf_b, const float* x,
                  const float* rays_o, const float* rays_d, const float* t, const int64_t* ridx, int64_t S,
                  const int64_t* n_dev, int64_t n_add, const float* feat_planes, float* sdf, void* stream);
int nsim_wide_bwd_sdf(const NsimFieldMeta* meta, const void* wpack, const float* x, const float* rays_o,
                      const float* rays_d, const float* t, const int64_t* ridx, int64_t S, const float* h_planes,
                      const void* J_planes, int64_t plane_pitch, const float* dsdf, const float* gn, float* dh_planes,
                      float* g_planes, float* dsdf_w, float* dsdf_b, void* stream);
/* (3) LoTD scatter (LoTD backward incl. the dy/dx path): dgrid[level][vertex][f] (f32, atomics) +=
 *     w_c * dh[f] + g[f] * (d w_c/d x . gn).  gn may be NULL (no second-order term).
 *     Levels [level_begin, level_begin + level_count) only (level_count <= 0: all) -- a data-parallel caller scatters
 *     the pyramid in two halves and starts the all-reduce of the first half's gradient while the second is computed. */
int nsim_lotd_scatter(const NsimLotdMeta* meta, const float* x, const float* rays_o, const float* rays_d,
                      const float* t, const int64_t* ridx, const int64_t* ray_goff, int64_t S,
                      const float* dh_planes, const float* g_planes, const float* gn, float* dgrid, int level_begin,
                      int level_count, void* stream);
/* (4, pose refinement only) position gradient of the normals' own dependence on x: nablas = (d sdf/d h) . dh/dx(x), and
 *     inside a cell the trilinear interpolant has mixed second derivatives:
 *     dx[s][c] += sum_{l,f} g[l][s][f] sum_{c' != c} gn[s][c'] d2 h_{l,f} / dx_c' dx_c.   g_planes from (2), gn from (1). */
int nsim_lotd_hess_dx(const NsimLotdMeta* meta, const void* grid_f16, const float* x, const float* rays_o,
                      const float* rays_d, const float* t, const int64_t* ridx, const int64_t* ray_goff, int64_t S,
                      const float* g_planes, const float* gn, float* dx, void* stream);
/* (5, pose refinement only) x_s = o_r + t_s d_r, v_s = d_r  =>  d_rays_o[r] += dx_s, d_rays_d[r] += t_s dx_s + dv_s
 *     (dv, d_rays_o, d_rays_d may each be NULL); the outputs [R,3] are initialised by the caller. */
int nsim_ray_grad_reduce(const float* dx, const float* dv, const float* t, const int64_t* ridx, int64_t S,
                         float* d_rays_o, float* d_rays_d, void* stream);

/* ------------------------------------------------ NeRF++ distant-view model (LoTDNeRFDistant) */
/* 4-D LoTD level table of ``lotd_auto_compute_cfg{type: ngp4d}`` (lotd_neus.dtu.230814.yaml:193-200): level l has
 * res_xyz^3 * res_w vertices (Dense) or a 2^k-entry hash table; 2 features per level; <= 16 levels.
 * ``lotd_use_cuboid: true`` (withmask_withlidar_joint.240219.yaml:256): per-axis vertex counts -- res_xyz is the x
 * count, res_y / res_z the other two; 0 in res_y / res_z means "same as res_xyz" (cubic level). */
typedef struct NsimLotd4Meta {
  int32_t num_levels;
  int32_t res_xyz[16];
  int32_t res_w[16];
  int32_t type[16];
  uint32_t size[16];
  int64_t offset[16];
  int32_t res_y[16];
  int32_t res_z[16];
} NsimLotd4Meta;

typedef struct NsimDistantMeta {
  NsimLotd4Meta lotd;
  int32_t precision;     /* 0: fp16 MFMA, 1: exact f32 MFMA */
} NsimDistantMeta;

int64_t nsim_distant_wpack_bytes(const NsimDistantMeta* meta);
/* den_w: [D1 (64 x F), Dhead (1 x 64)], den_b: [64, 1]; rad_w: [Q1 (64 x (F+20)), Q2 (64x64), Q3 (3x64)], rad_b [64,64,3];
 * F = 2 * num_levels; radiance input = [features F, SH-4 (16), appearance (4)]  (yaml :221-234). */
int nsim_distant_pack_weights(const NsimDistantMeta* meta, const float* den_w, const float* den_b, const float* rad_w,
                              const float* rad_b, void* wpack, void* stream);
/* ``ray_query_cfg{query_mode: march, march_cfg{sample_mode: box, max_steps: K}}`` with ``radius_scale_min/max``:
 * K shells per ray, 1/r uniform; t [N,K] = exit depth of the AABB (host float[6] = min,max) scaled by r about its
 * centre, u4 [N,K,4] = network input in [0,1]^4, valid [N,K] (shell crossed beyond near[ray]). jitter [N,K] or NULL. */
int nsim_distant_shells(const float* rays_o, const float* rays_d, const float* near, const float* jitter, int64_t N,
                        int K, const float* aabb, float r_min, float r_max, float* t, float* u4, uint8_t* valid,
                        void* stream);
/* alpha = 1 - exp(-sigma * delta), delta = t[k+1]-t[k]; last shell: 1e10 when include_inf_distance (object-centric
 * configs, lotd_neus.dtu.230814.yaml:236) else the previous interval repeated (street config with a sky model,
 * withmask_withlidar_joint.240219.yaml:294); 0 if !valid */
int nsim_density_alpha_fwd(const float* sigma, const float* t, const uint8_t* valid, int64_t N, int K,
                           int include_inf_distance, float* alpha, void* stream);
int nsim_density_alpha_bwd(const float* sigma, const float* t, const uint8_t* valid, const float* dalpha, int64_t N,
                           int K, int include_inf_distance, float* dsigma, void* stream);
/* fused 4-D gather + density MLP + radiance MLP on S = N*K points (ray = s / K): sigma [S], rgb [S,3];
 * h_planes [16,S,2] (may be NULL) saves the features for the backward. */
int nsim_distant_fwd(const NsimDistantMeta* meta, const void* grid_f16, const void* wpack, const float* u4,
                     const float* rays_d, const float* h_appear, int64_t S, int K, float* sigma, float* rgb,
                     float* h_planes, void* stream);
/* backward of nsim_distant_fwd: accumulates dden_w/dden_b/drad_w/drad_b (layouts above), dh_appear [N,4] (may be
 * NULL) and writes dh_planes [16,S,2] for nsim_lotd4_scatter. */
int nsim_distant_bwd(const NsimDistantMeta* meta, const void* wpack, const float* h_planes, const float* sigma_fwd,
                     const float* rgb_fwd, const float* rays_d, const float* h_appear, const uint8_t* valid,
                     int64_t S, int K, const float* dsigma, const float* drgb, float* dh_planes, float* dden_w,
                     float* dden_b, float* drad_w, float* drad_b, float* dh_appear, void* stream);
int nsim_lotd4_scatter(const NsimLotd4Meta* meta, const float* u4, const uint8_t* valid, int64_t S,
                       const float* dh_planes, float* dgrid, void* stream);

/* ------------------------------------------------------------------------------- sky MLP (row a16) */
/* ``SimpleSky`` (app/models/env/sky.py:16-51) as configured by the street configs
 * (code_single/configs/waymo/streetsurf/withmask_withlidar_joint.240219.yaml:312-322): sinusoidal embedding of the
 * unit view direction [v, {sin(2^f v), cos(2^f v)}_{f<F}] (3 + 6F dims) ++ h_appear (A dims) -> 256 -> 256 -> 3,
 * ReLU hidden, sigmoid output (D = 2, W = 256 are fixed; 3 + 6F + A <= 96).  One query per RAY
 * (call site app/renderers/single_volume_renderer.py:449-457).
 * Flat weights in layer order: w = [W1 (256 x IN), W2 (256 x 256), W3 (3 x 256)] row-major, b = [256, 256, 3],
 * IN = 3 + 6F + A.  planes_fwd: [(96 + 256 + 256) * pitch] floats, planes_bwd: [(32 + 256 + 256) * pitch] floats
 * scratch, pitch = nsim_sky_plane_pitch(N) (N rounded up to 128). */
typedef struct NsimSkyMeta {
  int32_t n_frequencies;  /* F */
  int32_t n_appear;       /* A */
  int32_t precision;      /* 0: fp16 MFMA, 1: exact f32 MFMA */
} NsimSkyMeta;

int64_t nsim_sky_wpack_bytes(const NsimSkyMeta* meta);
int64_t nsim_sky_plane_pitch(int64_t N);
int nsim_sky_pack_weights(const NsimSkyMeta* meta, const float* w, const float* b, void* wpack, void* stream);
/* rgb [N,3] = sky(v [N,3], h_appear [N,A] (NULL iff A == 0)); planes_fwd (may be NULL) saves the activations. */
int nsim_sky_fwd(const NsimSkyMeta* meta, const void* wpack, const float* v, const float* h_appear, int64_t N,
                 float* rgb, float* planes_fwd, void* stream);
/* accumulates dw / db (layouts above, caller zeroes) and writes dh_appear [N,A] (may be NULL). */
int nsim_sky_bwd(const NsimSkyMeta* meta, const void* wpack, const float* rgb_fwd, const float* drgb, int64_t N,
                 const float* planes_fwd, float* planes_bwd, float* dw, float* db, float* dh_appear, void* stream);

/* ------------------------------------------------------------------------------- losses (row a18) */
/* Fused reductions of the two losses of the object-centric configs and the gradient of the appearance-embedding
 * lookup.  *out is a device scalar the caller zeroes; the kernels ADD the mean into it.
 *   eikonal: mean((|nablas_i| - 1)^2), nablas [S,3]          replaces app/loss/eikonal.py:96-105 (fn, plain mse) + .mean()
 *   mse:     mean((pred - gt)^2) over n floats               replaces app/loss/photometric.py:88-146 (fn_type mse)
 *   rows_scatter_add: out[idx[i], :] += g[i, :], g [n,C], out [rows,C]
 *                                                            backward of ``embed[fidx]``, app/models/scene/image_embeddings.py:23-80 */
int nsim_eikonal_loss_fwd(const float* nablas, int64_t S, float* out, void* stream);
int nsim_eikonal_loss_bwd(const float* nablas, int64_t S, const float* gout, float* dnablas, void* stream);
int nsim_mse_loss_fwd(const float* pred, const float* gt, int64_t n, float* out, void* stream);
int nsim_mse_loss_bwd(const float* pred, const float* gt, int64_t n, const float* gout, float* dpred, void* stream);
int nsim_rows_scatter_add(const float* g, const int64_t* idx, int64_t n, int C, int64_t rows, float* out, void* stream);
/* its forward: out[i, :] = table[idx[i], :] (i < n; an index outside [0, rows) gives zeros), followed by ``tail`` zero rows
 * (the free points a training step appends to its rays carry no appearance code) -- out [n + tail, C]. */
int nsim_rows_gather(const float* table, const int64_t* idx, int64_t n, int C, int64_t rows, int64_t tail, float* out,
                     void* stream);
/* The loss head of one training step in a single launch (the reference's total = mse + w (eikonal(render samples) +
 * eikonal(uniform points)), code_single/tools/train.py:1411-1423 with app/loss/photometric.py + eikonal.py):
 *   acc[0] += mse(pred, gt) over n_img floats; acc[1] += eikonal(nablas[:S]); acc[2] += eikonal(nablas[S:S+M]);
 *   d_pred = d total / d pred;  d_nablas [S+M,3] = d total / d nablas.   acc [3] zero on entry. */
int nsim_train_loss_head(const float* pred, const float* gt, int64_t n_img, const float* nablas, int64_t S, int64_t M,
                         float w_eikonal, float* acc, float* d_pred, float* d_nablas, void* stream);
/* Compaction of the AABB-tested rays, (o, d, near, far)[idx] -> [R,...] in one launch
 * (model.ray_test, app/renderers/single_volume_renderer.py:235-238). */
int nsim_gather_rays(const float* rays_o, const float* rays_d, const float* near, const float* far, const int64_t* idx,
                     int64_t R, float* o_out, float* d_out, float* near_out, float* far_out, void* stream);
/* Synthetic supervision of bench.py (no dataset in this environment): analytic image of a sphere of ``radius`` at the
 * origin along unit rays -- rgb = 0.5 + 0.5 normal at the first hit, 0 elsewhere. */
int nsim_sphere_image(const float* rays_o, const float* rays_d, int64_t N, float radius, float* rgb, void* stream);

/* ------------------------------------------------------------------------------- optimizer */
/* Adam (training_cfg{eps 1e-15, betas [.9,.99]}, lotd_neus.dtu.230814.yaml:178-184) on f32 master params;
 * p16 (may be NULL) receives the fp16 copy used by the kernels; grad is scaled by grad_scale; zero_grad bit 0: the gradient is
 * zeroed; bit 1 (opt-in, SURVEY sec. 8f-3 "Adam only on touched hash entries"): entries whose gradient is exactly 0 this step
 * are skipped entirely -- no moment decay, no update (torch.optim.SparseAdam's rule; NOT the reference's dense optimizer). */
int nsim_adam_step(float* p, void* p16, float* grad, float* m, float* v, int64_t n, float lr, float beta1,
                   float beta2, float eps, float bias1, float bias2, float grad_scale, int zero_grad,
                   void* stream);

/* The same update for up to NSIM_ADAM_MULTI_MAX small tensors in one launch (the decoder weights / biases, inv_s and
 * appearance codes of a step); per tensor: its own betas / bias corrections and a learning-rate factor. */
#define NSIM_ADAM_MULTI_MAX 12
typedef struct {
  float* p;
  void* p16;        /* may be NULL */
  float* grad;
  float* m;
  float* v;
  int64_t n;
  float beta1, beta2, bias1, bias2, lr_scale;
} NsimAdamTensor;
int nsim_adam_multi(const NsimAdamTensor* tensors, int n_tensors, float lr, float eps, float grad_scale, int zero_grad,
                    void* stream);

/* MFMA layout self-test (tests only): writes D = A(32x16 f16) * B(16x32 f16) with the wrappers used by the
 * field kernels; a, b given in plain row-major. d is 32x32 f32 row-major. */
int nsim_selftest_mfma(const float* a, const float* b, float* d, int use_f32, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* NSIM_H */
