/*
 * oracle/suma_oracle.h -- TEST INFRASTRUCTURE.  CPU restatement of the SuMa++ projective-ICP +
 * surfel-fusion hot path (reference: PRBonn/semantic_suma, src/core + src/shader).
 *
 * PINNED TO THE REFERENCE'S SOURCE TEXT: the reference ships no tests, golden vectors or fixtures
 * and its host side cannot be built here (OpenGL context, glow, Eigen, gtsam, Qt), but its GLSL shaders and
 * lie_algebra.cpp are compiled with g++ where they lie (oracle/ref_build.py -> oracle/_ref/libsuma_ref.so) and
 * tests/test_ref_shaders.py demands equal values between them and this restatement for every stage (K1-K12) on
 * the live inputs of a scan sequence.  What stays modelled (GL leaves it to the driver): triangle coverage of K4,
 * transcendental last bits, blend order; see the deviation tests in that file.  Each function cites the
 * file:line it follows; analytic known-answer tests are in tests/test_oracle_kat.py.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 * The product (semantic_suma_amd/) never links, imports or calls it.
 */
#ifndef SUMA_ORACLE_H_
#define SUMA_ORACLE_H_

#include <stdint.h>

#include "../include/suma_types.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ora_frame {
  uint32_t width, height;
  suma_float4* vertex;
  suma_float4* normal;
  suma_float4* semantic;
} ora_frame;

typedef struct ora_ctx ora_ctx;

ora_ctx* ora_create(const suma_params* p);
void ora_destroy(ora_ctx* c);
void ora_set_params(ora_ctx* c, const suma_params* p);
/* number of OpenMP threads used by the pixel / surfel loops (1 = faithful sequential) */
void ora_set_threads(ora_ctx* c, int n);

ora_frame* ora_frame_create(uint32_t w, uint32_t h);
void ora_frame_destroy(ora_frame* f);
suma_float4* ora_frame_map(ora_frame* f, int which);
void ora_frame_copy(ora_frame* dst, const ora_frame* src);

/* K1-K3: Preprocessing::process (src/core/Preprocessing.cpp:120-339) */
void ora_preprocess(ora_ctx* c, const suma_float4* points, const float* labels, const float* probs, uint32_t n,
                    uint32_t timestamp, ora_frame* out);

/* K6: Frame2Model::jacobianProducts (src/core/Frame2Model.cpp:136-261).  acc = raw int64 sums
 * (SUMA_ACC_WORDS), JtJ column-major 6x6. Returns F. */
double ora_icp_jacobian_products(ora_ctx* c, const ora_frame* current, const ora_frame* model, const double pose[16],
                                 uint32_t iteration, int64_t* acc, double* JtJ, double* Jtr, suma_icp_stats* st);
/* LieGaussNewton::minimize (src/core/LieGaussNewton.cpp:13-79): history gets (n_hist) 4x4 doubles */
void ora_icp_minimize(ora_ctx* c, const ora_frame* current, const ora_frame* model, const double T0[16],
                      double T_out[16], double* history, uint32_t history_cap, uint32_t* n_hist, suma_icp_stats* st);
/* the same with Frame2Model::iteration_ starting at iteration0 (a minimisation that is NOT the first one behind a
 * setData: SurfelMapping.cpp:693-700) */
void ora_icp_minimize_from(ora_ctx* c, const ora_frame* current, const ora_frame* model, const double T0[16],
                           uint32_t iteration0, double T_out[16], double* history, uint32_t history_cap,
                           uint32_t* n_hist, suma_icp_stats* st);

/* SurfelMap (src/core/SurfelMap.cpp) */
void ora_map_reset(ora_ctx* c);
void ora_map_update(ora_ctx* c, const float pose[16], const ora_frame* frame);
void ora_map_render(ora_ctx* c, const float pose_old[16], const float pose_new[16], float conf_threshold,
                    ora_frame* out);
void ora_map_render_active(ora_ctx* c, const float pose[16], float conf_threshold);
void ora_map_render_inactive(ora_ctx* c, const float pose[16], float conf_threshold);
void ora_map_render_composed(ora_ctx* c, const float pose_old[16], const float pose_new[16], float conf_threshold);
ora_frame* ora_map_frame(ora_ctx* c, int which);
void ora_map_update_poses(ora_ctx* c, const float* poses16, uint32_t n);
uint32_t ora_map_size(const ora_ctx* c);
uint32_t ora_map_timestamp(const ora_ctx* c);
const suma_surfel* ora_map_surfels(const ora_ctx* c);
void ora_map_upload(ora_ctx* c, const suma_surfel* s, uint32_t n, uint32_t timestamp);
/* intermediates of the last update, for stage-by-stage parity */
const uint32_t* ora_map_index_map(const ora_ctx* c);       /* P, surfel id + 1, 0 = none */
const suma_float4* ora_map_radius_conf(const ora_ctx* c);  /* P */
const uint8_t* ora_map_integrated(const ora_ctx* c);       /* P */
uint32_t ora_map_last_updated_count(const ora_ctx* c);     /* S' */
uint32_t ora_map_last_new_count(const ora_ctx* c);         /* D  */
uint32_t ora_map_cached_surfels(const ora_ctx* c);         /* surfels parked in submap caches */
const suma_surfel* ora_map_updated_surfels(const ora_ctx* c); /* K9 output (S' records), before K11 */
const suma_surfel* ora_map_data_surfels(const ora_ctx* c);    /* K10 output (D records), before K11 */
const float* ora_map_poses(const ora_ctx* c);                 /* pose table, max_poses x 16 */
uint32_t ora_map_pending_extractions(const ora_ctx* c);
const suma_surfel* ora_map_cache_tile(const ora_ctx* c, int32_t i, int32_t j, uint32_t* n);
uint32_t ora_map_last_extraction(const ora_ctx* c, int32_t* ij); /* returns the number of K12 extractions so far */
/* K4 vertex + geometry stage per surfel, before rasterisation (tests/test_ref_shaders.py) */
void ora_debug_render_quads(const ora_ctx* c, const float pose[16], float conf_threshold, int mode, int32_t thr,
                            uint8_t* emitted, float* corners, float* pn);
/* the triangle rasteriser / the depth quantisation alone (tests/test_gl_reference.py compares them with a real GL) */
void ora_debug_raster_quads(int32_t W, int32_t H, const float* corners, const uint32_t* ids, uint32_t n, int use_disc,
                            int use_depth, int64_t* winner);
void ora_debug_depth24(const float* zw, uint32_t n, uint32_t* out);
void ora_map_submap_origin(const ora_ctx* c, int32_t* ij);

/* results of the two device-side parts of SurfelMapping::checkLoopClosure (field for field the product's
 * suma_loop_result / suma_loop_track, include/suma_hip.h) */
typedef struct ora_loop_result {
  double gn_pose[16];
  suma_icp_stats after_minimize;
  int32_t passed;
  float pose_old[16];
  suma_icp_stats composed;
  double JtJ[36];
} ora_loop_result;
typedef struct ora_loop_track {
  double increment_old[16];
  suma_icp_stats after_minimize;
  float increment_difference;
  int32_t passed;
  double pose_old[16];
  suma_icp_stats composed;
  double JtJ[36];
} ora_loop_track;
void ora_loop_closure_verify(ora_ctx* c, const ora_frame* current, const double pose_prior[16], const double* inits,
                             uint32_t n_init, const float pose_new[16], float conf_threshold, float min_valid_ratio,
                             float max_outlier_ratio, ora_loop_result* out); /* SurfelMapping.cpp:679-757 */
void ora_loop_closure_track(ora_ctx* c, const ora_frame* current, const double last_pose_old[16],
                            const double last_increment[16], const float pose_new[16], float conf_threshold,
                            double min_valid_ratio, double max_outlier_ratio, double max_increment_difference,
                            ora_loop_track* out); /* SurfelMapping.cpp:546-574 */
void ora_se3_log(const double T[16], double x[6]); /* lie_algebra.cpp:36-71 */

/* SurfelMapping::processScan (src/core/SurfelMapping.cpp:175-210): one call, or its three phases with the loop-closure
 * hooks between them */
typedef struct ora_pipeline ora_pipeline;
ora_pipeline* ora_pipeline_create(const suma_params* p);
void ora_pipeline_destroy(ora_pipeline* s);
ora_ctx* ora_pipeline_ctx(ora_pipeline* s);
/* fixed_iterations > 0: run exactly that many GN iterations (bench mode) */
void ora_pipeline_process_scan(ora_pipeline* s, const suma_float4* points, const float* labels, const float* probs,
                               uint32_t n, int32_t fixed_iterations);
void ora_pipeline_begin_scan(ora_pipeline* s, const suma_float4* points, const float* labels, const float* probs,
                             uint32_t n);
void ora_pipeline_update_pose(ora_pipeline* s, int32_t fixed_iterations);
void ora_pipeline_update_map(ora_pipeline* s);
void ora_pipeline_integrate_loop_closures(ora_pipeline* s, const float* poses16, uint32_t n, const double difference[16]);
void ora_pipeline_set_pose_old(ora_pipeline* s, const double pose_old[16]);
/* which: 0 currentPose_, 1 currentPose_old_, 2 currentPose_new_, 3 lastPose_old_, 4 lastPose_ */
void ora_pipeline_get_pose(const ora_pipeline* s, int which, double pose[16]);
void ora_pipeline_verify_loop_closure(ora_pipeline* s, const double pose_prior[16], const double* inits, uint32_t n_init,
                                      float min_valid_ratio, float max_outlier_ratio, ora_loop_result* out);
void ora_pipeline_track_loop_closure(ora_pipeline* s, double min_valid_ratio, double max_outlier_ratio,
                                     double max_increment_difference, ora_loop_track* out); /* reference: 0.2, 0.85, 0.1 */
void ora_pipeline_pose(const ora_pipeline* s, double pose[16]);
void ora_pipeline_last_increment(const ora_pipeline* s, double inc[16]);
void ora_pipeline_last_stats(const ora_pipeline* s, suma_icp_stats* st);
uint32_t ora_pipeline_track_loss(const ora_pipeline* s);
ora_frame* ora_pipeline_frame(ora_pipeline* s, int which); /* 0 current data, 1 last model, 2 current model */

/* host math exposed for tests: SE3::exp (lie_algebra.cpp:4-34), 6x6 LDLT solve */
void ora_se3_exp(const double x[6], double T[16]);
void ora_solve6(const double* JtJ, const double* Jtr, double* dx);

#ifdef __cplusplus
}
#endif
#endif
