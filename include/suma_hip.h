/*
 * suma_hip.h -- C-ABI of the MI355X (gfx950) projective-ICP + surfel-fusion core.
 *
 * This is the drop-in boundary for the hot path of SuMa++ (PRBonn/semantic_suma).  Every entry
 * point below replaces one method of the reference's C++ classes in src/core (Preprocessing,
 * Frame2Model/Objective + LieGaussNewton, SurfelMap, and the per-scan sequencing of
 * SurfelMapping); the reference-side adapter that keeps those class signatures and calls these
 * functions is shown in INTEGRATION.md and include/suma_adapter.hpp.
 *
 * Conventions
 *  - extern "C", plain pointers and sizes; no exceptions cross the boundary.
 *  - return value: 0 = SUMA_OK, negative = error (suma_last_error() gives the text).
 *  - one suma_ctx = one HIP device + one HIP stream; calls on a ctx are serialised by the caller
 *    (as in the reference, whose methods all run on the thread that owns the GL context);
 *    different ctxs may be driven from different threads / processes (one per GPU).  suma_ctx_create makes its
 *    device the calling thread's current HIP device; a thread that drives ctxs on DIFFERENT devices must make the
 *    ctx's device current (hipSetDevice) before calling into it -- the entry points do not switch devices.
 *  - matrices are column-major 4x4 (Eigen::Matrix4f / Matrix4d default storage).
 *  - host pointers unless the name says _device.
 *  - there is NO CPU fallback: without a gfx950 device suma_ctx_create fails.
 */
#ifndef SUMA_HIP_H_
#define SUMA_HIP_H_

#include <stdint.h>

#include "suma_types.h"

#ifdef __cplusplus
extern "C" {
#endif

enum {
  SUMA_OK = 0,
  SUMA_ERR_INVALID = -1,  /* bad argument */
  SUMA_ERR_HIP = -2,      /* HIP runtime error (no device, launch failure, ...) */
  SUMA_ERR_CAPACITY = -3, /* surfel / pose / cache capacity exceeded (reference: silent TF truncation,
                             SurfelMap.cpp:726-727) */
  SUMA_ERR_NOMEM = -4
};

typedef struct suma_ctx suma_ctx;
typedef struct suma_frame suma_frame;       /* reference: class Frame, src/core/Frame.h:21-79 */
typedef struct suma_pipeline suma_pipeline; /* reference: class SurfelMapping, src/core/SurfelMapping.h */

const char* suma_version(void);
/* last error text of this ctx (or of the failed create when ctx == NULL) */
const char* suma_last_error(const suma_ctx* ctx);

/* ---- context: constructors / setParameters of Preprocessing, Frame2Model, LieGaussNewton,
 *      SurfelMap (Preprocessing.cpp:16-118, Frame2Model.cpp:14-110, LieGaussNewton.cpp:81-91,
 *      SurfelMap.cpp:8-457).  Image sizes and capacities are fixed at creation (textures and
 *      buffers are sized in the reference's constructors); all other keys may be re-sent. */
int suma_ctx_create(const suma_params* params, int hip_device, suma_ctx** out);
void suma_ctx_destroy(suma_ctx* ctx);
int suma_set_params(suma_ctx* ctx, const suma_params* params);
int suma_synchronize(suma_ctx* ctx);
/* the hipStream_t the work of this ctx is enqueued on (for event timing by the caller).  One exception: a scan
 * pipeline runs the preprocessing K1-K3 of a scan on a side stream of its own (it overlaps the surfel passes of the
 * previous scan) and joins it to this stream before the first reader.  Consequence for suma_pipeline_*_device: the scan
 * buffers must be COMPLETE when the call is made -- work the caller has enqueued on this stream to produce them is not
 * waited for by the side stream (synchronise it first, or stage through suma_pipeline_prefetch_scan, whose upload the
 * preprocessing does wait for).  SUMA_NO_SIDE_STREAM=1 in the environment keeps everything on this one stream. */
void* suma_ctx_stream(suma_ctx* ctx);

/* ---- frames: Frame::Frame / Frame::copy (Frame.h:26-61); which = SUMA_MAP_* */
int suma_frame_create(suma_ctx* ctx, uint32_t width, uint32_t height, suma_frame** out);
void suma_frame_destroy(suma_frame* f);
int suma_frame_copy(suma_ctx* ctx, suma_frame* dst, const suma_frame* src);
int suma_frame_download(suma_ctx* ctx, const suma_frame* f, int which, suma_float4* host);
int suma_frame_upload(suma_ctx* ctx, suma_frame* f, int which, const suma_float4* host);
/* a frame whose maps were written through an exported device pointer (suma_frame_device_ptr / suma_frame_export) by
 * work the library did not enqueue: tells the render de-duplication and the fused K8 products that the contents
 * changed (every writing call of this API does so by itself) */
int suma_frame_touch(suma_ctx* ctx, suma_frame* f);
uint32_t suma_frame_width(const suma_frame* f);
uint32_t suma_frame_height(const suma_frame* f);
/* device address of one map (HIP->GL interop / zero-copy consumers) */
void* suma_frame_device_ptr(const suma_frame* f, int which);
/* exchange the contents of two frames of equal size (the shared_ptr swaps of SurfelMapping::initialize,
 * SurfelMapping.cpp:323-331): O(1), no copy */
int suma_frame_swap(suma_ctx* ctx, suma_frame* a, suma_frame* b);
/* viewer feed (ViewportWidget.cpp:404-434 binds Frame::vertex_map / normal_map / semantic_map as textures): the
 * device buffer of one map with its layout -- row-major, row 0 = lowest beam, RGBA32F texels, row_bytes = 16 * width.
 * The pointer stays valid for the life of the frame; contents change with the next call that writes the frame.
 * INTEGRATION.md shows the hipGraphicsGLRegisterImage / hipMemcpy2DToArrayAsync recipe. */
int suma_frame_export(suma_ctx* ctx, const suma_frame* f, int which, void** d_ptr, uint32_t* width, uint32_t* height,
                      uint32_t* row_bytes);

/* ---- Preprocessing::process (Preprocessing.h:55-56, Preprocessing.cpp:120-339): K1 z-buffered
 *      spherical scatter, K2 cross-stencil normals + label erosion, K3 label flood fill.
 *      points: n x (x,y,z,1) as rv::Point3f; labels / probs: n floats (may be NULL). */
int suma_preprocess(suma_ctx* ctx, const suma_float4* points, const float* labels, const float* probs, uint32_t n,
                    uint32_t timestamp, suma_frame* out);
/* same with the scan already resident in HBM */
int suma_preprocess_device(suma_ctx* ctx, const suma_float4* d_points, const float* d_labels, const float* d_probs,
                           uint32_t n, uint32_t timestamp, suma_frame* out);

/* ---- Objective::setData (Objective.h:58, Frame2Model.cpp:117-123) */
int suma_icp_set_data(suma_ctx* ctx, const suma_frame* current, const suma_frame* model);
/* ---- the parameters a Frame2Model OBJECT owns (Frame2Model::updateParameters / setParameter, Frame2Model.cpp:65-115).
 *      The reference builds two objectives with different gates -- objective_ and recovery_ = Frame2Model(fallback
 *      parameters), SurfelMapping.cpp:87-94 -- while one suma_ctx carries one parameter block.  An adapter object
 *      sends its own values before each launch; NULL returns to the ctx parameters (suma_params). */
typedef struct suma_icp_objective {
  float icp_max_distance; /* "icp-max-distance" */
  float icp_max_angle;    /* "icp-max-angle", degrees */
  int32_t weight_function; /* SUMA_WEIGHT_* ("weighting") */
  float factor;
  int32_t bilinear_sampling;
} suma_icp_objective;
int suma_icp_set_objective(suma_ctx* ctx, const suma_icp_objective* objective);
/* ---- Frame2Model::jacobianProducts (Frame2Model.h:50, Frame2Model.cpp:136-261): K6 at the given
 *      pose.  JtJ 6x6 column-major, Jtr 6; acc (optional) = the raw 2^-28 fixed-point sums,
 *      SUMA_ACC_WORDS int64.  Returns F in stats->error. */
int suma_icp_jacobian_products(suma_ctx* ctx, const double pose[16], uint32_t iteration, double JtJ[36], double Jtr[6],
                               int64_t* acc, suma_icp_stats* stats);
/* ---- LieGaussNewton::minimize (LieGaussNewton.h:32, LieGaussNewton.cpp:13-79) with
 *      Objective::increment / SE3::exp (Objective.h:45-48, lie_algebra.cpp:4-34): the whole
 *      Gauss-Newton loop runs on the device (no per-iteration readback).
 *      history (optional): history_cap x 16 doubles receive LieGaussNewton::history(); *n_hist = entries pushed. */
int suma_icp_minimize(suma_ctx* ctx, const double T0[16], double T_out[16], double* history, uint32_t history_cap,
                      uint32_t* n_hist, suma_icp_stats* stats);
/* Frame2Model::iteration_ the NEXT suma_icp_minimize starts with (one shot; suma_icp_set_data resets it to 0 like
 * Frame2Model::setData, Frame2Model.cpp:117-123).  The reference resets the counter in setData ONLY: a caller that
 * minimises several times on one setData -- the loop over the initial guesses in checkLoopClosure,
 * SurfelMapping.cpp:693-700 -- starts its later minimisations with iteration_ > 0, which the Tukey weight reads
 * (Frame2Model_jacobians.geom:129).  An adapter object passes its own counter here before every minimisation. */
int suma_icp_set_iteration(suma_ctx* ctx, uint32_t iteration);
/* LieGaussNewton::history() (LieGaussNewton.h:44) of the last suma_icp_minimize, fetched on demand: the device always
 * records it, the copy is only paid by callers that look at it (the reference's caller keeps it for drawing,
 * SurfelMapping.cpp:391).  *n_hist = entries the minimisation pushed; min(that, history_cap) x 16 doubles are copied. */
int suma_icp_history(suma_ctx* ctx, double* history, uint32_t history_cap, uint32_t* n_hist);
/* The history lives in ONE device buffer per context; every minimisation that records one (suma_icp_minimize, also inside
 * suma_loop_closure_verify_serial / suma_loop_closure_track; NOT the batched suma_loop_closure_verify nor the scan
 * pipeline's own minimisations, which record none) overwrites it and advances this counter.  The reference keeps history_ per optimizer
 * object (LieGaussNewton.h:72): an adapter object notes the counter after ITS minimisation and refuses to hand out
 * another chain's poses when it has moved (include/suma_adapter.hpp, LieGaussNewton::history). */
uint64_t suma_icp_history_sequence(const suma_ctx* ctx);
/* LieGaussNewton::information() (LieGaussNewton.h:50, LieGaussNewton.cpp:75,103-105): J^T W J of the last step of the
 * last suma_icp_minimize (or of the last suma_icp_jacobian_products), 6x6 column-major */
int suma_icp_information(suma_ctx* ctx, double information[36]);
/* n_hyp independent minimisations of the same frame pair from different T0 (the reference's
 * loop-closure verification pattern, SurfelMapping.cpp:662-779; BASELINE config 3) in one batch. */
int suma_icp_minimize_batch(suma_ctx* ctx, const double* T0s, uint32_t n_hyp, double* T_out, suma_icp_stats* stats);

/* ---- SurfelMap (SurfelMap.h:36-78) */
int suma_map_reset(suma_ctx* ctx);                                                  /* SurfelMap.cpp:473-482 */
int suma_map_update(suma_ctx* ctx, const float pose[16], const suma_frame* frame); /* SurfelMap.cpp:492-584 */
/* render(pose_old, pose_new, frame, ct), SurfelMap.cpp:847-1021; also fills the OLD / NEW frames */
int suma_map_render(suma_ctx* ctx, const float pose_old[16], const float pose_new[16], float conf_threshold,
                    suma_frame* out);
int suma_map_render_active(suma_ctx* ctx, const float pose[16], float conf_threshold);   /* SurfelMap.cpp:1023-1069 */
int suma_map_render_inactive(suma_ctx* ctx, const float pose[16], float conf_threshold); /* SurfelMap.cpp:1071-1114 */
int suma_map_render_composed(suma_ctx* ctx, const float pose_old[16], const float pose_new[16],
                             float conf_threshold);                                      /* SurfelMap.cpp:1116-1165 */
/* oldMapFrame() / newMapFrame() / composedFrame(), SurfelMap.h:59-61; which = SUMA_FRAME_* */
suma_frame* suma_map_frame(suma_ctx* ctx, int which);
int suma_map_update_poses(suma_ctx* ctx, const float* poses16, uint32_t n); /* SurfelMap.cpp:485-490 */
int suma_map_size(suma_ctx* ctx, uint32_t* n);                              /* SurfelMap::size() */
int suma_map_timestamp(suma_ctx* ctx, uint32_t* t);
/* getAllSurfels(), SurfelMap.cpp:1232-1237: copies min(size, cap) surfels; *n = size */
int suma_map_download(suma_ctx* ctx, suma_surfel* host, uint32_t cap, uint32_t* n);
/* viewer feed / getModelSurfels() / getDataSurfels() (SurfelMap.h:64-67; SurfelMap::draw reads the surfel VBO,
 * SurfelMap.cpp:1167-1230): the device buffer of the active map, *n records of 64 bytes in the layout of
 * suma_surfel (= the reference's Surfel / its VAO layout, SurfelMap.cpp:46-55).  The map is double buffered: the
 * pointer is valid (and its contents stable) until the next suma_map_update / suma_map_upload / suma_map_reset or
 * pipeline scan.  The data surfels of the last update are the tail [*first, *first + *n_data) of the same buffer:
 * the new surfels that survived the active-area copy (the reference's data_surfels_ also holds the ones K11
 * dropped).  INTEGRATION.md shows the hipGraphicsGLRegisterBuffer recipe. */
int suma_map_export_surfels(suma_ctx* ctx, void** d_ptr, uint32_t* n);
int suma_map_export_data_surfels(suma_ctx* ctx, void** d_ptr, uint32_t* first, uint32_t* n_data);
/* checkpoint / resume: replace the active map */
int suma_map_upload(suma_ctx* ctx, const suma_surfel* host, uint32_t n, uint32_t timestamp);
/* intermediates of the last update, for stage-by-stage parity tests */
int suma_map_download_index_map(suma_ctx* ctx, uint32_t* host);      /* P, surfel id + 1 (uint32, not float) */
int suma_map_download_radius_conf(suma_ctx* ctx, suma_float4* host); /* P */
int suma_map_download_integrated(suma_ctx* ctx, uint8_t* host);      /* P */
int suma_map_counts(suma_ctx* ctx, uint32_t* n_updated, uint32_t* n_new, uint32_t* n_cached, int32_t origin_ij[2]);
/* SurfelMap::poses_ (SurfelMap.h:205-208): the pose table, entries 0 .. timestamp - 1 (16 floats each, column-major; after
 * suma_map_update_poses the optimised ones).  *n = entries in the table; min(*n, capacity) are copied. */
int suma_map_download_poses(suma_ctx* ctx, float* host, uint32_t capacity, uint32_t* n);
/* the submap cache in HBM (the reference pages tiles to host vectors, SurfelMap.h:186, SurfelMap.cpp:733-734): surfels
 * allocated from the arena (live tiles + the blocks that re-extracted tiles left behind), its capacity
 * (suma_params.cache_surfels), and how often it has been compacted -- when it runs full the live tiles are copied into
 * a fresh arena; SUMA_ERR_CAPACITY only if the live tiles alone do not fit */
int suma_map_cache_stats(suma_ctx* ctx, uint32_t* used, uint32_t* capacity, uint32_t* compactions);
/* one parked tile, as the reference keeps it in submapCache_(i, j).surfels (SurfelMap.h:186, filled by extractSurfels,
 * SurfelMap.cpp:733-734): *n = its surfels (0 for a tile that was never extracted); the first min(*n, capacity)
 * records are copied to `host` (may be NULL with capacity 0 to ask for the size) */
int suma_map_download_cached_tile(suma_ctx* ctx, int32_t i, int32_t j, suma_surfel* host, uint32_t capacity, uint32_t* n);

/* ---- SurfelMapping::processScan (SurfelMapping.h:47, SurfelMapping.cpp:175-210) without the
 *      loop-closure / pose-graph part: initialize, preprocess, updatePose
 *      (incl. the frame-to-frame fallback, :434-449), updateMap.
 *      fixed_iterations > 0 runs exactly that many GN iterations (bench mode). */
int suma_pipeline_create(const suma_params* params, int hip_device, suma_pipeline** out);
void suma_pipeline_destroy(suma_pipeline* s);
/* SurfelMapping::reset (SurfelMapping.cpp:131-169): empty map, identity poses, timestamp 0 */
int suma_pipeline_reset(suma_pipeline* s);
suma_ctx* suma_pipeline_ctx(suma_pipeline* s);
int suma_pipeline_process_scan(suma_pipeline* s, const suma_float4* points, const float* labels, const float* probs,
                               uint32_t n, int32_t fixed_iterations);
int suma_pipeline_process_scan_device(suma_pipeline* s, const suma_float4* d_points, const float* d_labels,
                                      const float* d_probs, uint32_t n, int32_t fixed_iterations);
/* ---- device-side scan ingest (KITTIReader::read hands over host vectors, KITTIReader.cpp:136-203 ->
 *      SurfelMapping::processScan(const rv::Laserscan&), SurfelMapping.cpp:175): three pinned staging slots, a copy
 *      stream and an ingest thread.  prefetch stages the scan (host copy into pinned memory + async H2D, both off
 *      the caller's thread) and returns at once; process_prefetched runs the oldest staged scan, its first kernel
 *      waiting on the upload's event.  Calling prefetch(scan k+1) before process_prefetched(scan k) overlaps the
 *      upload of k+1 with the kernels of k.  At most two scans may be staged; the host arrays must stay valid
 *      until the matching process call returns.  suma_pipeline_process_scan_async = prefetch (unless that very
 *      scan is already staged) + process_prefetched. */
int suma_pipeline_prefetch_scan(suma_pipeline* s, const suma_float4* points, const float* labels, const float* probs,
                                uint32_t n);
int suma_pipeline_process_prefetched(suma_pipeline* s, int32_t fixed_iterations);
int suma_pipeline_process_scan_async(suma_pipeline* s, const suma_float4* points, const float* labels,
                                     const float* probs, uint32_t n, int32_t fixed_iterations);
/* ---- the phases of SurfelMapping::processScan as calls of their own (SurfelMapping.cpp:175-204), for hosts that run
 *      loop closures between them -- config/default.xml:71 ships close-loops = true, and then processScan is
 *        integrateLoopClosures (:179) -> initialize + preprocess (:181-187) -> updatePose (:192) ->
 *        checkLoopClosure (:196) -> updateMap (:201) -> timestamp_ += 1 (:209).
 *      begin_scan = initialize + preprocess (K1-K3 on the side stream, the pre-ICP render(pose_old, pose_new));
 *      update_pose = updatePose (no-op while timestamp == 0, as :190); update_map = updateMap + timestamp_ += 1.
 *      suma_pipeline_process_scan* IS these three back to back: same launches, same side stream, fused K7 / K8, lazy
 *      statistics and render de-duplication -- a host with loop closures on loses none of them.
 *      Between update_pose and update_map the host may call suma_pipeline_verify_loop_closure /
 *      suma_pipeline_track_loop_closure (the device sides of checkLoopClosure) and suma_pipeline_set_pose_old;
 *      before begin_scan, suma_pipeline_integrate_loop_closures. */
int suma_pipeline_begin_scan(suma_pipeline* s, const suma_float4* points, const float* labels, const float* probs,
                             uint32_t n);
int suma_pipeline_begin_scan_device(suma_pipeline* s, const suma_float4* d_points, const float* d_labels,
                                    const float* d_probs, uint32_t n);
/* the oldest scan staged with suma_pipeline_prefetch_scan */
int suma_pipeline_begin_prefetched(suma_pipeline* s);
int suma_pipeline_update_pose(suma_pipeline* s, int32_t fixed_iterations);
int suma_pipeline_update_map(suma_pipeline* s);
/* integrateLoopClosures (SurfelMapping.cpp:211-250) once the pose graph has been optimised: map_->updatePoses(poses)
 * (:236), currentPose_ = difference * currentPose_ (:239), currentPose_new_ = currentPose_old_ = currentPose_ (:243).
 * poses16: n x 16 floats (casted_poses); difference: poses_opt[beforeID_] * beforeOptimizationPose_^-1 (:229). */
int suma_pipeline_integrate_loop_closures(suma_pipeline* s, const float* poses16, uint32_t n, const double difference[16]);
/* currentPose_old_ = ... (SurfelMapping.cpp:582 after a tracked closure, :744 after a verified candidate): the pose the
 * NEXT scan's render() uses for the inactive ("old") part of the map */
int suma_pipeline_set_pose_old(suma_pipeline* s, const double pose_old[16]);
/* which: 0 currentPose_, 1 currentPose_old_, 2 currentPose_new_, 3 lastPose_old_ (:456), 4 lastPose_ */
int suma_pipeline_get_pose(const suma_pipeline* s, int which, double pose[16]);
/* result_new_ of updatePose (:417-423): the statistics pass of the current scan (resolves the lazy read-back) --
 * what checkLoopClosure compares candidates against (:538-539, :582, :727-729) */
int suma_pipeline_result_new(suma_pipeline* s, suma_icp_stats* st);

/* ---- several pose hypotheses per scan (BASELINE config 3; the reference runs several minimisations of one frame pair
 *      from different starts in its loop-closure verification, SurfelMapping.cpp:662-779): between begin_scan and
 *      update_map, INSTEAD of update_pose.  minimize_hypotheses runs n_hyp (<= 64) device-resident Gauss-Newton chains
 *      as one batch against the rendered model (map_->newMapFrame(), as updatePose does, :384); apply_increment does
 *      updatePose's pose bookkeeping (:453-474) for the increment the caller chose.  T0s / T_out: n_hyp x 16 doubles. */
int suma_pipeline_minimize_hypotheses(suma_pipeline* s, const double* T0s, uint32_t n_hyp, int32_t fixed_iterations,
                                      double* T_out, suma_icp_stats* stats);
int suma_pipeline_apply_increment(suma_pipeline* s, const double increment[16]);

int suma_pipeline_pose(const suma_pipeline* s, double pose[16]);
int suma_pipeline_last_increment(const suma_pipeline* s, double inc[16]);
int suma_pipeline_last_stats(const suma_pipeline* s, suma_icp_stats* st);
/* the frame-to-model minimisation of the last suma_pipeline_update_pose as gn_ left it (iterations = gn_->iterationCount(),
 * SurfelMapping.cpp:394; converged; the objective's counters after the last step) -- known when update_pose returns,
 * unlike the statistics pass behind it (suma_pipeline_last_stats), which is read back lazily */
int suma_pipeline_minimize_stats(const suma_pipeline* s, suma_icp_stats* st);
uint32_t suma_pipeline_timestamp(const suma_pipeline* s);
/* number of scans on which the frame-to-frame fallback minimisation ran (trackLoss_, SurfelMapping.cpp:441) */
uint32_t suma_pipeline_track_loss(const suma_pipeline* s);
/* which: 0 current data frame, 1 last model frame, 2 current model frame */
suma_frame* suma_pipeline_frame(suma_pipeline* s, int which);

/* ---- loop-closure verification, the device side of SurfelMapping::checkLoopClosure
 *      (SurfelMapping.cpp:662-757): render the inactive map from a candidate pose, run the
 *      frame-to-model minimisation from each initial guess, and -- for guesses that pass the
 *      valid / outlier gates -- render the composed (old + new) view and evaluate the objective at
 *      identity against it.  Faithful to the reference's sequencing, including its quirk that after
 *      a passing guess the objective keeps pointing at the composed frame for the remaining guesses
 *      (setData is called once before the loop, :693, and again inside the branch, :719).
 *      The candidate search, thresholds on the returned ratios and the pose-graph edges stay with the
 *      caller (SurfelMapping / gtsam are out of scope). */
typedef struct suma_loop_result {
  double gn_pose[16];            /* LieGaussNewton::pose() of this guess (relative to pose_prior) */
  suma_icp_stats after_minimize; /* jacobianProducts at that pose (:705): valid / outlier ratios */
  int32_t passed;                /* valid_ratio > min_valid_ratio && outlier_ratio < max_outlier_ratio (:713) */
  float pose_old[16];            /* (pose_prior * gn_pose).cast<float>() (:714) */
  suma_icp_stats composed;       /* jacobianProducts at identity against composedFrame (:723); zero if !passed */
  double JtJ[36];                /* information matrix of that evaluation (result_old_.information, :744) */
} suma_loop_result;
int suma_loop_closure_verify(suma_ctx* ctx, const suma_frame* current, const double pose_prior[16],
                             const double* initializations, uint32_t n_init, const float pose_new[16],
                             float conf_threshold, float min_valid_ratio, float max_outlier_ratio,
                             suma_loop_result* out);
/* suma_loop_closure_verify minimises the initial guesses as ONE batched Gauss-Newton chain (grid.y = guess) with the
 * per-guess evaluation (:705) riding on the same chain states: one chain of launches and one synchronisation when no
 * guess (or only the last) passes; after a guess that passes, the later guesses -- which the reference minimises
 * against composedFrame() (:718-719) -- are redone as a batch against that frame.  This entry is the reference's
 * sequencing literally (one minimisation, one evaluation, two host round trips per guess): identical results, kept as
 * the cross-check of the batched form and for the A/B timing in bench.py. */
int suma_loop_closure_verify_serial(suma_ctx* ctx, const suma_frame* current, const double pose_prior[16],
                                    const double* initializations, uint32_t n_init, const float pose_new[16],
                                    float conf_threshold, float min_valid_ratio, float max_outlier_ratio,
                                    suma_loop_result* out);

/* the same on a pipeline's own state between suma_pipeline_update_pose and suma_pipeline_update_map: current frame,
 * currentPose_new_ and getConfidenceThreshold() are the pipeline's (SurfelMapping.cpp:679-719) */
int suma_pipeline_verify_loop_closure(suma_pipeline* s, const double pose_prior[16], const double* initializations,
                                      uint32_t n_init, float min_valid_ratio, float max_outlier_ratio,
                                      suma_loop_result* out);

/* ---- the other device part of checkLoopClosure: re-verifying a closure that is being tracked, on the scans that
 *      follow its detection (SurfelMapping.cpp:546-574): render the inactive map from lastPose_old_, minimise from
 *      lastIncrement_, gate on the valid / outlier ratios of the objective as the minimisation left it (:557-558) and on
 *      |log(lastIncrement_) - log(increment_old)| (:561-563; SE3::log, lie_algebra.cpp:36-71), then render the composed
 *      view at lastPose_old_ * increment_old and evaluate the objective at identity against it (:566-572).  The
 *      reference's gates are the literals 0.2 / 0.85 / 0.1. */
typedef struct suma_loop_track {
  double increment_old[16];      /* gn_->pose() (:560) */
  suma_icp_stats after_minimize; /* objective_->valid() / outlier() / inlier() / invalid() after minimize (:557-558) */
  float increment_difference;    /* (:561) */
  int32_t passed;                /* (:563) */
  double pose_old[16];           /* lastPose_old_ * increment_old (:564, :581): what currentPose_old_ becomes */
  suma_icp_stats composed;       /* jacobianProducts at identity against composedFrame (:570-572); zero if !passed */
  double JtJ[36];
} suma_loop_track;
int suma_loop_closure_track(suma_ctx* ctx, const suma_frame* current, const double last_pose_old[16],
                            const double last_increment[16], const float pose_new[16], float conf_threshold,
                            double min_valid_ratio, double max_outlier_ratio, double max_increment_difference,
                            suma_loop_track* out);
/* on a pipeline's own state (lastPose_old_, lastIncrement_, currentPose_new_, current frame); the reference's
 * gates are 0.2, 0.85, 0.1 */
int suma_pipeline_track_loop_closure(suma_pipeline* s, double min_valid_ratio, double max_outlier_ratio,
                                     double max_increment_difference, suma_loop_track* out);
/* SE3::log (lie_algebra.cpp:36-71), host side: x = (v, omega) */
void suma_se3_log(const double T[16], double x[6]);

/* ---- device scratch for callers that keep scans resident in HBM (bench, replay) */
int suma_device_alloc(suma_ctx* ctx, uint64_t bytes, void** d_ptr);
int suma_device_free(suma_ctx* ctx, void* d_ptr);
int suma_device_upload(suma_ctx* ctx, void* d_dst, const void* host_src, uint64_t bytes);
int suma_device_download(suma_ctx* ctx, void* host_dst, const void* d_src, uint64_t bytes);

/* ---- per-kernel timing (rv::Stopwatch / SurfelMapping::Stats, SurfelMapping.cpp:183-207):
 *      on = 1: every kernel group is bracketed by HIP events on the ctx stream; on = 2: only the
 *      Gauss-Newton chain (the kernel with the largest share of GPU time), which costs two event
 *      records per scan (each costs the stream a ~6 us bubble); on = 3: the same on every 4th scan; on = 0: off.
 *      suma_profile_get fills up to cap entries, returns the number of distinct kernels. */
typedef struct suma_kernel_time {
  char name[48];
  uint64_t launches;
  double total_ms;
  double bytes; /* algorithmic bytes summed over the launches (SURVEY.md 8d formulas) */
} suma_kernel_time;
/* Where the calls of the blocking host-vector entry (suma_pipeline_process_scan / suma_pipeline_begin_scan: the
 * reference's processScan(const rv::Laserscan&), SurfelMapping.cpp:175, 323-331) spent their time ON THE CALLER'S THREAD,
 * summed since the last reset: out = {calls, whole calls, wait for the staging slot, pageable -> pinned copies, upload
 * enqueue, kernel enqueue, wait for the minimisation result, threads that share the copies}; seconds.  The copy helpers
 * are sized from the CPUs the process may use (affinity mask and cgroup quota; SUMA_COPY_HELPERS overrides). */
int suma_pipeline_host_entry_times(suma_pipeline* s, double out[8], int reset);
int suma_profile_enable(suma_ctx* ctx, int on);
int suma_profile_reset(suma_ctx* ctx);
int suma_profile_get(suma_ctx* ctx, suma_kernel_time* out, uint32_t cap);

#ifdef __cplusplus
}
#endif
#endif /* SUMA_HIP_H_ */
