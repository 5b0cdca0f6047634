/*
 * oracle/o_pipeline.c -- TEST INFRASTRUCTURE (CPU oracle).
 * Context / frame management and the per-scan sequencing of SurfelMapping::processScan
 * (reference src/core/SurfelMapping.cpp:175-210, 323-358, 372-476, 797-804) in its three phases, the pose
 * bookkeeping of integrateLoopClosures (:211-250) and the two device-side parts of checkLoopClosure (:546-574 a
 * tracked closure verified again, :662-757 a candidate verified from three initial guesses).  The candidate
 * search, the pose graph and its optimiser (gtsam) are not restated: a test scripts their outputs.
 */
#include "o_ctx.h"

ora_frame* ora_frame_create(uint32_t w, uint32_t h) {
  ora_frame* f = (ora_frame*)malloc(sizeof(ora_frame));
  size_t P = (size_t)w * h;
  f->width = w;
  f->height = h;
  f->vertex = (suma_float4*)calloc(P, sizeof(suma_float4));
  f->normal = (suma_float4*)calloc(P, sizeof(suma_float4));
  f->semantic = (suma_float4*)calloc(P, sizeof(suma_float4));
  return f;
}
void ora_frame_destroy(ora_frame* f) {
  if (!f) return;
  free(f->vertex);
  free(f->normal);
  free(f->semantic);
  free(f);
}
suma_float4* ora_frame_map(ora_frame* f, int which) {
  return which == SUMA_MAP_VERTEX ? f->vertex : (which == SUMA_MAP_NORMAL ? f->normal : f->semantic);
}
void ora_frame_copy(ora_frame* dst, const ora_frame* src) {
  size_t P = (size_t)src->width * src->height;
  memcpy(dst->vertex, src->vertex, P * sizeof(suma_float4));
  memcpy(dst->normal, src->normal, P * sizeof(suma_float4));
  memcpy(dst->semantic, src->semantic, P * sizeof(suma_float4));
}

ora_ctx* ora_create(const suma_params* p) {
  ora_ctx* c = (ora_ctx*)calloc(1, sizeof(ora_ctx));
  c->p = *p;
  c->threads = 1;
  o_map_alloc(c);
  return c;
}
void ora_destroy(ora_ctx* c) {
  if (!c) return;
  o_map_free(c);
  free(c);
}
/* geometry / capacities must not change (as in the reference, where textures are sized in ctors) */
void ora_set_params(ora_ctx* c, const suma_params* p) { c->p = *p; }
void ora_set_threads(ora_ctx* c, int n) { c->threads = n < 1 ? 1 : n; }

struct ora_pipeline {
  ora_ctx* c;
  ora_frame *last_frame, *current_frame, *current_model, *last_model;
  double current_pose[16], last_pose[16], pose_old[16], pose_new[16], last_increment[16];
  double last_pose_old[16]; /* lastPose_old_, SurfelMapping.cpp:456 */
  uint32_t timestamp;
  float log_unstable;
  suma_icp_stats stats;
  uint32_t track_loss;
};

static void o_eye(double* T) {
  for (int i = 0; i < 16; ++i) T[i] = (i % 5 == 0) ? 1.0 : 0.0;
}
static void o_mul4(const double* A, const double* B, double* C) {
  for (int c = 0; c < 4; ++c)
    for (int r = 0; r < 4; ++r)
      C[4 * c + r] =
          ((A[r] * B[4 * c] + A[4 + r] * B[4 * c + 1]) + A[8 + r] * B[4 * c + 2]) + A[12 + r] * B[4 * c + 3];
}
static void o_rigid_inv_d(const double* m, double* out) {
  for (int c = 0; c < 3; ++c)
    for (int r = 0; r < 3; ++r) out[4 * c + r] = m[4 * r + c];
  for (int r = 0; r < 3; ++r) out[12 + r] = -((m[4 * r] * m[12] + m[4 * r + 1] * m[13]) + m[4 * r + 2] * m[14]);
  out[3] = out[7] = out[11] = 0.0;
  out[15] = 1.0;
}
static void o_cast(const double* T, float* out) {
  for (int i = 0; i < 16; ++i) out[i] = (float)T[i];
}

ora_pipeline* ora_pipeline_create(const suma_params* p) {
  ora_pipeline* s = (ora_pipeline*)calloc(1, sizeof(ora_pipeline));
  s->c = ora_create(p);
  s->last_frame = ora_frame_create(p->data_width, p->data_height);
  s->current_frame = ora_frame_create(p->data_width, p->data_height);
  s->current_model = ora_frame_create(p->model_width, p->model_height);
  s->last_model = ora_frame_create(p->model_width, p->model_height);
  o_eye(s->current_pose);
  o_eye(s->last_pose);
  o_eye(s->pose_old);
  o_eye(s->pose_new);
  o_eye(s->last_increment);
  o_eye(s->last_pose_old);
  /* SurfelMapping.cpp:108-109 */
  float p_unstable = 0.1f;
  s->log_unstable = (float)log((double)(p_unstable / (1.0f - p_unstable)));
  return s;
}
void ora_pipeline_destroy(ora_pipeline* s) {
  if (!s) return;
  ora_frame_destroy(s->last_frame);
  ora_frame_destroy(s->current_frame);
  ora_frame_destroy(s->current_model);
  ora_frame_destroy(s->last_model);
  ora_destroy(s->c);
  free(s);
}
ora_ctx* ora_pipeline_ctx(ora_pipeline* s) { return s->c; }
void ora_pipeline_pose(const ora_pipeline* s, double pose[16]) { memcpy(pose, s->current_pose, 16 * sizeof(double)); }
void ora_pipeline_last_increment(const ora_pipeline* s, double inc[16]) {
  memcpy(inc, s->last_increment, 16 * sizeof(double));
}
void ora_pipeline_last_stats(const ora_pipeline* s, suma_icp_stats* st) { *st = s->stats; }
uint32_t ora_pipeline_track_loss(const ora_pipeline* s) { return s->track_loss; } /* trackLoss_, SurfelMapping.cpp:441 */
ora_frame* ora_pipeline_frame(ora_pipeline* s, int which) {
  return which == 0 ? s->current_frame : (which == 1 ? s->last_model : s->current_model);
}

/* SurfelMapping::getConfidenceThreshold, SurfelMapping.cpp:333-340 (time_init = 10, SurfelMapping.h:192) */
static float o_conf_threshold(const ora_pipeline* s) {
  float ct = s->c->p.confidence_threshold;
  const uint32_t time_init = 10;
  if (s->timestamp < time_init) {
    float alpha = (float)s->timestamp / (float)time_init;
    ct = (float)((1.0 - (double)alpha) * (double)s->log_unstable + (double)(alpha * s->c->p.confidence_threshold));
  }
  return ct;
}

static void o_minimize_cfg(ora_pipeline* s, const ora_frame* cur, const ora_frame* model, const double* T0, double* T,
                           int32_t fixed_iterations, suma_icp_stats* st) {
  suma_params saved = s->c->p;
  if (fixed_iterations > 0) {
    s->c->p.max_iterations = (uint32_t)fixed_iterations;
    s->c->p.stopping_threshold = 0.0f;
    s->c->p.delta = 0.0f;
  }
  ora_icp_minimize(s->c, cur, model, T0, T, NULL, 0, NULL, st);
  s->c->p = saved;
}

/* SurfelMapping::updatePose, SurfelMapping.cpp:372-476 */
static void o_update_pose(ora_pipeline* s, int32_t fixed_iterations) {
  ora_ctx* c = s->c;
  double T0[16], increment[16];
  if (!c->p.initialize_identity)
    memcpy(T0, s->last_increment, sizeof(T0));
  else
    o_eye(T0);
  suma_icp_stats mst;
  o_minimize_cfg(s, s->current_frame, ora_map_frame(c, SUMA_FRAME_NEW), T0, increment, fixed_iterations, &mst);

  double inv_last[16], delta[16], posed[16], I[16];
  float posef[16];
  o_rigid_inv_d(s->last_increment, inv_last);
  o_mul4(inv_last, increment, delta);
  o_mul4(s->pose_new, increment, posed);
  o_cast(posed, posef);
  ora_map_render_active(c, posef, o_conf_threshold(s));        /* :406 */
  ora_frame_copy(s->last_model, ora_map_frame(c, SUMA_FRAME_NEW)); /* :407 */
  o_eye(I);
  suma_icp_stats st;
  ora_icp_jacobian_products(c, s->current_frame, ora_map_frame(c, SUMA_FRAME_NEW), I, 0, NULL, NULL, NULL, &st); /* :411-413 */
  st.iterations = mst.iterations;
  st.converged = mst.converged;
  s->stats = st;

  float t_err = (float)sqrt((delta[12] * delta[12] + delta[13] * delta[13]) + delta[14] * delta[14]);
  float angle = (float)(0.5 * (((delta[0] + delta[5]) + delta[10]) - 1.0));
  float r_err = (float)acos((double)fmaxf(fminf(angle, 1.0f), -1.0f));
  if (s->timestamp > 1 && ((double)t_err > 0.4 || (double)r_err > 0.1) && c->p.fallback_mode) { /* :438-449: float against the double literals */
    s->track_loss += 1;
    suma_params saved = c->p;
    c->p.icp_max_distance = c->p.fallback_max_distance;
    c->p.icp_max_angle = c->p.fallback_max_angle;
    o_minimize_cfg(s, s->current_frame, s->last_frame, T0, increment, fixed_iterations, &mst);
    c->p = saved;
  }
  memcpy(s->last_pose, s->current_pose, sizeof(s->last_pose));
  double np[16];
  o_mul4(s->current_pose, increment, np);
  memcpy(s->current_pose, np, sizeof(np));
  memcpy(s->last_pose_old, s->pose_old, sizeof(np)); /* :456 */
  memcpy(s->pose_old, np, sizeof(np));
  memcpy(s->pose_new, np, sizeof(np));
  memcpy(s->last_increment, increment, sizeof(increment));
}

/* initialize() + preprocess(), SurfelMapping.cpp:181-187 */
void ora_pipeline_begin_scan(ora_pipeline* s, const suma_float4* points, const float* labels, const float* probs,
                             uint32_t n) {
  ora_ctx* c = s->c;
  /* initialize(), SurfelMapping.cpp:323-331 */
  ora_frame* t = s->last_frame;
  s->last_frame = s->current_frame;
  s->current_frame = t;
  t = s->last_model;
  s->last_model = s->current_model;
  s->current_model = t;
  /* preprocess(), :342-358 */
  ora_preprocess(c, points, labels, probs, n, s->timestamp, s->current_frame);
  float po[16], pn[16];
  o_cast(s->pose_old, po);
  o_cast(s->pose_new, pn);
  ora_map_render(c, po, pn, o_conf_threshold(s), s->last_model);
}
/* :190-193 */
void ora_pipeline_update_pose(ora_pipeline* s, int32_t fixed_iterations) {
  if (s->timestamp > 0) o_update_pose(s, fixed_iterations);
}
/* updateMap(), :797-804, and timestamp_ += 1 (:209) */
void ora_pipeline_update_map(ora_pipeline* s) {
  ora_ctx* c = s->c;
  float pc[16];
  o_cast(s->current_pose, pc);
  ora_map_update(c, pc, s->current_frame);
  ora_map_render(c, pc, pc, o_conf_threshold(s), s->current_model);
  s->timestamp += 1;
}
void ora_pipeline_process_scan(ora_pipeline* s, const suma_float4* points, const float* labels, const float* probs,
                               uint32_t n, int32_t fixed_iterations) {
  ora_pipeline_begin_scan(s, points, labels, probs, n);
  ora_pipeline_update_pose(s, fixed_iterations);
  ora_pipeline_update_map(s);
}

/* integrateLoopClosures, SurfelMapping.cpp:211-250, the part behind the optimiser's future: poses16 = casted_poses,
 * difference = poses_opt[beforeID_] * beforeOptimizationPose_.inverse() (:229) */
void ora_pipeline_integrate_loop_closures(ora_pipeline* s, const float* poses16, uint32_t n, const double difference[16]) {
  ora_map_update_poses(s->c, poses16, n); /* :236 */
  double np[16];
  o_mul4(difference, s->current_pose, np); /* :239 */
  memcpy(s->current_pose, np, sizeof(np));
  memcpy(s->pose_old, np, sizeof(np)); /* :243 */
  memcpy(s->pose_new, np, sizeof(np));
}
void ora_pipeline_set_pose_old(ora_pipeline* s, const double pose_old[16]) { memcpy(s->pose_old, pose_old, 16 * sizeof(double)); }
void ora_pipeline_get_pose(const ora_pipeline* s, int which, double pose[16]) {
  const double* src[5] = {s->current_pose, s->pose_old, s->pose_new, s->last_pose_old, s->last_pose};
  memcpy(pose, src[which], 16 * sizeof(double));
}

/* SE3::log, lie_algebra.cpp:36-71 */
void ora_se3_log(const double T[16], double x[6]) {
  /* R(r, c) = T[4 * c + r]; R.trace() adds the diagonal in order */
  const double d = 0.5 * (((T[0] + T[5]) + T[10]) - 1.0); /* :44 */
  double W[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};             /* omega_skew, row-major */
  for (int i = 0; i < 6; ++i) x[i] = 0.0;
  if (d < 1 - 1e-10) { /* :47 */
    const double theta = acos(d);
    const double f = theta / (2 * sin(theta)); /* :49 */
    for (int r = 0; r < 3; ++r)
      for (int cc = 0; cc < 3; ++cc) W[3 * r + cc] = f * (T[4 * cc + r] - T[4 * r + cc]);
    x[3] = W[3 * 2 + 1]; /* :51-53 */
    x[4] = W[3 * 0 + 2];
    x[5] = W[3 * 1 + 0];
  }
  const double theta = sqrt((x[3] * x[3] + x[4] * x[4]) + x[5] * x[5]); /* :56 */
  const double t[3] = {T[12], T[13], T[14]};
  x[0] = t[0];
  x[1] = t[1];
  x[2] = t[2];
  if (fabs(theta) > 1e-10) { /* :60 */
    const double half_theta = 0.5 * theta;
    const double alpha = -0.5;
    const double beta = 1 / (theta * theta) * (1 - theta * cos(half_theta) / (2 * sin(half_theta))); /* :64 */
    double W2[9], Vi[9];
    for (int r = 0; r < 3; ++r)
      for (int cc = 0; cc < 3; ++cc)
        W2[3 * r + cc] = (W[3 * r] * W[cc] + W[3 * r + 1] * W[3 + cc]) + W[3 * r + 2] * W[6 + cc];
    for (int r = 0; r < 3; ++r)
      for (int cc = 0; cc < 3; ++cc) Vi[3 * r + cc] = ((r == cc ? 1.0 : 0.0) + alpha * W[3 * r + cc]) + beta * W2[3 * r + cc];
    for (int r = 0; r < 3; ++r) x[r] = (Vi[3 * r] * t[0] + Vi[3 * r + 1] * t[1]) + Vi[3 * r + 2] * t[2]; /* :67 */
  }
}

/* checkLoopClosure, the candidate loop (SurfelMapping.cpp:679-757): render the inactive map from pose_prior, minimise
 * from every initial guess, and for guesses that pass the ratio gates render the composed view and evaluate the
 * objective at identity against it.  As in the reference the objective keeps pointing at the composed frame for the
 * guesses that follow a passing one (setData at :693 is outside the loop, the one at :719 inside). */
void ora_loop_closure_verify(ora_ctx* c, const ora_frame* current, const double pose_prior[16], const double* inits,
                             uint32_t n_init, const float pose_new[16], float conf_threshold, float min_valid_ratio,
                             float max_outlier_ratio, ora_loop_result* out) {
  float prior_f[16];
  o_cast(pose_prior, prior_f);
  ora_map_render_inactive(c, prior_f, conf_threshold); /* :679 */
  const ora_frame* model = ora_map_frame(c, SUMA_FRAME_OLD); /* :693 */
  /* Frame2Model::iteration_: reset by the setData calls at :693 and :719 only, advanced by every increment -- it keeps
   * counting ACROSS the guesses (the Tukey weight of a later guess is on from its first step) */
  uint32_t iteration = 0;
  for (uint32_t k = 0; k < n_init; ++k) {
    ora_loop_result* o = &out[k];
    memset(o, 0, sizeof(*o));
    suma_icp_stats mst;
    ora_icp_minimize_from(c, current, model, inits + 16 * (size_t)k, iteration, o->gn_pose, NULL, 0, NULL, &mst); /* :700 */
    iteration += mst.iterations + (mst.converged ? 1u : 0u); /* one increment per step, also the converged one */
    ora_icp_jacobian_products(c, current, model, o->gn_pose, iteration, NULL, NULL, NULL, &o->after_minimize); /* :705 */
    o->after_minimize.iterations = mst.iterations;
    o->after_minimize.converged = mst.converged;
    const suma_icp_stats* s0 = &o->after_minimize;
    const float valid_ratio = (float)s0->valid / (float)(s0->valid + s0->invalid);       /* :707 */
    const float outlier_ratio = (float)s0->outlier / (float)(s0->outlier + s0->inlier); /* :708 */
    double pd[16];
    o_mul4(pose_prior, o->gn_pose, pd);
    o_cast(pd, o->pose_old); /* :714 */
    o->passed = (valid_ratio > min_valid_ratio && outlier_ratio < max_outlier_ratio) ? 1 : 0; /* :713 */
    if (o->passed) {
      ora_map_render_composed(c, o->pose_old, pose_new, conf_threshold); /* :717 */
      model = ora_map_frame(c, SUMA_FRAME_COMPOSED);                     /* :719 */
      iteration = 0;                                                      /* setData, Frame2Model.cpp:122 */
      double I[16];
      o_eye(I);
      ora_icp_jacobian_products(c, current, model, I, 0, NULL, o->JtJ, NULL, &o->composed); /* :720-723 */
    }
  }
}

/* checkLoopClosure, part 1 (SurfelMapping.cpp:546-574): a tracked closure is verified again on the next scans */
void ora_loop_closure_track(ora_ctx* c, const ora_frame* current, const double last_pose_old[16],
                            const double last_increment[16], const float pose_new[16], float conf_threshold,
                            double min_valid_ratio, double max_outlier_ratio, double max_increment_difference,
                            ora_loop_track* o) {
  memset(o, 0, sizeof(*o));
  float pf[16];
  o_cast(last_pose_old, pf);                      /* :548 */
  ora_map_render_inactive(c, pf, conf_threshold); /* :550 */
  const ora_frame* model = ora_map_frame(c, SUMA_FRAME_OLD);
  ora_icp_minimize(c, current, model, last_increment, o->increment_old, NULL, 0, NULL, &o->after_minimize); /* :553-554 */
  const suma_icp_stats* s0 = &o->after_minimize;
  const float valid_ratio = (float)s0->valid / (float)(s0->valid + s0->invalid);       /* :557 */
  const float outlier_ratio = (float)s0->outlier / (float)(s0->outlier + s0->inlier); /* :558 */
  double la[6], lb[6], sq = 0.0;
  ora_se3_log(last_increment, la);
  ora_se3_log(o->increment_old, lb);
  for (int i = 0; i < 6; ++i) sq += (la[i] - lb[i]) * (la[i] - lb[i]);
  o->increment_difference = (float)sqrt(sq); /* :561 */
  o_mul4(last_pose_old, o->increment_old, o->pose_old);
  o->passed = ((double)valid_ratio > min_valid_ratio && (double)outlier_ratio < max_outlier_ratio &&
               (double)o->increment_difference < max_increment_difference) ? 1 : 0; /* :563 */
  if (o->passed) {
    float po[16];
    o_cast(o->pose_old, po);                                      /* :564 */
    ora_map_render_composed(c, po, pose_new, conf_threshold);    /* :567 */
    double I[16];
    o_eye(I);
    ora_icp_jacobian_products(c, current, ora_map_frame(c, SUMA_FRAME_COMPOSED), I, 0, NULL, o->JtJ, NULL, &o->composed); /* :569-572 */
  }
}

void ora_pipeline_verify_loop_closure(ora_pipeline* s, const double pose_prior[16], const double* inits, uint32_t n_init,
                                      float min_valid_ratio, float max_outlier_ratio, ora_loop_result* out) {
  float pn[16];
  o_cast(s->pose_new, pn); /* currentPose_new_.cast<float>(), :717 */
  ora_loop_closure_verify(s->c, s->current_frame, pose_prior, inits, n_init, pn, o_conf_threshold(s), min_valid_ratio,
                          max_outlier_ratio, out);
}
void ora_pipeline_track_loop_closure(ora_pipeline* s, double min_valid_ratio, double max_outlier_ratio,
                                     double max_increment_difference, ora_loop_track* out) {
  float pn[16];
  o_cast(s->pose_new, pn);
  ora_loop_closure_track(s->c, s->current_frame, s->last_pose_old, s->last_increment, pn, o_conf_threshold(s),
                         min_valid_ratio, max_outlier_ratio, max_increment_difference, out); /* :563: 0.2, 0.85, 0.1 */
}
