/*
 * oracle/o_pipeline.c -- TEST INFRASTRUCTURE (CPU oracle).
 * Context / frame management and the per-scan sequencing of SurfelMapping::processScan
 * (reference src/core/SurfelMapping.cpp:175-210, 323-358, 372-476, 797-804), without loop closures
 * (SurfelMapping.cpp:527-795 is SURVEY 8(f)-1, "next").
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
  memcpy(s->pose_old, np, sizeof(np));
  memcpy(s->pose_new, np, sizeof(np));
  memcpy(s->last_increment, increment, sizeof(increment));
}

void ora_pipeline_process_scan(ora_pipeline* s, const suma_float4* points, const float* labels, const float* probs,
                               uint32_t n, int32_t fixed_iterations) {
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
  if (s->timestamp > 0) o_update_pose(s, fixed_iterations);
  /* updateMap(), :797-804 */
  float pc[16];
  o_cast(s->current_pose, pc);
  ora_map_update(c, pc, s->current_frame);
  ora_map_render(c, pc, pc, o_conf_threshold(s), s->current_model);
  s->timestamp += 1;
}
