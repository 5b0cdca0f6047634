/*
 * oracle/o_icp.c -- TEST INFRASTRUCTURE (CPU oracle).
 * Restates
 *   K6  Frame2Model::jacobianProducts (reference src/core/Frame2Model.cpp:136-261) and
 *       Frame2Model_jacobians.geom:53-247 (association, gating, weights, JtWJ / JtWr sums),
 *   LieGaussNewton::minimize/step (src/core/LieGaussNewton.cpp:13-79),
 *   Objective::increment (src/core/Objective.h:45-48), SE3::exp (src/core/lie_algebra.cpp:4-34).
 *
 * Deviation from the reference, on purpose (DESIGN.md "K6"): the reference sums fp32 terms with
 * ROP blending in an undefined order (Frame2Model.cpp:189-190), so it is not reproducible even
 * against itself.  Here every per-pixel fp32 term is converted to a 2^-28 fixed-point int64
 * (round to nearest even) and the integers are summed: exact, order independent, and therefore
 * bit-identical between this loop and any GPU reduction tree.
 */
#include "o_ctx.h"

/* GL_LINEAR + CLAMP_TO_BORDER fetch of a rectangle texture at continuous coords (GL 3.3 spec
 * 3.8.11): texel centres at integer + 0.5, border texels are (0,0,0,0); all four channels are
 * filtered, including validity and label (quirk B-5). */
static inline suma_float4 o_bilinear(const suma_float4* map, int32_t w, int32_t h, float x, float y) {
  float u = x - 0.5f, v = y - 0.5f;
  float fu = sdm_floor(u), fv = sdm_floor(v);
  float a = u - fu, b = v - fv;
  int32_t i0 = (int32_t)fu, j0 = (int32_t)fv;
  suma_float4 t00 = o_texel(map, w, h, i0, j0);
  suma_float4 t10 = o_texel(map, w, h, i0 + 1, j0);
  suma_float4 t01 = o_texel(map, w, h, i0, j0 + 1);
  suma_float4 t11 = o_texel(map, w, h, i0 + 1, j0 + 1);
  float w00 = (1.0f - a) * (1.0f - b), w10 = a * (1.0f - b), w01 = (1.0f - a) * b, w11 = a * b;
  suma_float4 r;
  r.x = ((w00 * t00.x + w10 * t10.x) + w01 * t01.x) + w11 * t11.x;
  r.y = ((w00 * t00.y + w10 * t10.y) + w01 * t01.y) + w11 * t11.y;
  r.z = ((w00 * t00.z + w10 * t10.z) + w01 * t01.z) + w11 * t11.z;
  r.w = ((w00 * t00.w + w10 * t10.w) + w01 * t01.w) + w11 * t11.w;
  return r;
}
static inline suma_float4 o_nearest(const suma_float4* map, int32_t w, int32_t h, float x, float y) {
  return o_texel(map, w, h, (int32_t)sdm_floor(x), (int32_t)sdm_floor(y));
}

static inline int64_t o_fix(float term) { return (int64_t)llrint((double)term * SUMA_ACC_SCALE); }

/* accumulator layout (SUMA_ACC_WORDS = 32):
 *   0..20  upper triangle of JtWJ, row-major: (0,0) (0,1) .. (0,5) (1,1) .. (5,5)
 *   21..26 JtWr
 *   27 F = sum w r^2 (valid)   28 sum w r^2 (inliers)   29 n_valid  30 n_outlier  31 n_invalid */
double ora_icp_jacobian_products(ora_ctx* c, const ora_frame* current, const ora_frame* model, const double pose[16],
                                 uint32_t iteration, int64_t* acc_out, double* JtJ, double* Jtr,
                                 suma_icp_stats* st) {
  const suma_params* p = &c->p;
  const int32_t W = (int32_t)current->width, H = (int32_t)current->height;
  const int32_t Wm = (int32_t)model->width, Hm = (int32_t)model->height;
  /* Frame2Model.cpp:66-67,82-83 */
  const float angle_thresh = (float)cos((double)p->icp_max_angle * M_PI / 180.0);
  const float distance_thresh = p->icp_max_distance;
  const float fov_up = fabsf(p->data_fov_up), fov_down = fabsf(p->data_fov_down);
  const float fov = fabsf(fov_up) + fabsf(fov_down);
  const float factor = p->factor;
  const int32_t wf = p->weight_function;
  float T[16];
  for (int i = 0; i < 16; ++i) T[i] = (float)pose[i]; /* pose_.cast<float>(), Frame2Model.cpp:194 */

  int64_t acc[SUMA_ACC_WORDS];
  memset(acc, 0, sizeof(acc));

  /* exact integer sums: any summation order (any number of threads) gives the same words */
#pragma omp parallel for num_threads(c->threads) schedule(static) reduction(+ : acc[:SUMA_ACC_WORDS])
  for (int32_t y = 0; y < H; ++y) {
    for (int32_t x = 0; x < W; ++x) {
      size_t pix = (size_t)y * W + x;
      suma_float4 vd4 = current->vertex[pix], nd4 = current->normal[pix];
      float e_d = vd4.w + nd4.w;
      ov3 v_d = om4_point(T, ov3_make(vd4.x, vd4.y, vd4.z));
      ov3 n_d = om4_dir(T, ov3_make(nd4.x, nd4.y, nd4.z));
      /* project2model, Frame2Model_jacobians.geom:53-65 */
      float depth = ov3_len(v_d);
      float yaw = sdm_atan2(v_d.y, v_d.x);
      float pitch = -sdm_asin(v_d.z / depth);
      float ix = (0.5f * ((-yaw * SUMA_INV_PI_F) + 1.0f)) * (float)Wm;
      float iy = (1.0f - ((pitch * SUMA_RAD2DEG_F) + fov_up) / fov) * (float)Hm;
      if (ix < 0.0f || ix >= (float)Wm || iy < 0.0f || iy >= (float)Hm) e_d = 0.0f;
      int in_image = (ix >= 0.0f && ix < (float)Wm && iy >= 0.0f && iy < (float)Hm); /* false for NaN */
      suma_float4 vm4, nm4, sm4;
      if (!in_image) {
        /* NaN coordinates: GL result undefined; defined here as the border colour */
        vm4 = nm4 = sm4 = o_f4(0.f, 0.f, 0.f, 0.f);
        e_d = 0.0f;
      } else if (p->bilinear_sampling) {
        vm4 = o_bilinear(model->vertex, Wm, Hm, ix, iy);
        nm4 = o_bilinear(model->normal, Wm, Hm, ix, iy);
        sm4 = o_bilinear(model->semantic, Wm, Hm, ix, iy);
      } else {
        vm4 = o_nearest(model->vertex, Wm, Hm, ix, iy);
        nm4 = o_nearest(model->normal, Wm, Hm, ix, iy);
        sm4 = o_nearest(model->semantic, Wm, Hm, ix, iy);
      }
      float e_m = vm4.w + nm4.w;
      if ((e_m > 1.5f) && (e_d > 1.5f)) {
        ov3 v_m = ov3_make(vm4.x, vm4.y, vm4.z), n_m = ov3_make(nm4.x, nm4.y, nm4.z);
        int inlier = 1;
        if (ov3_len(ov3_sub(v_m, v_d)) > distance_thresh) inlier = 0;
        if (ov3_dot(n_m, n_d) < angle_thresh) inlier = 0;
        float residual = ov3_dot(n_m, ov3_sub(v_d, v_m));
        ov3 n = n_m;
        ov3 cp = ov3_cross(v_d, n_m);
        float weight = 1.0f;
        if (wf == 4 || wf == 1) {
          if (fabsf(residual) > factor) weight = factor / fabsf(residual);
        } else if (wf == 2 && iteration > 0) {
          if (fabsf(residual) > factor) {
            weight = 0.0f;
          } else {
            float alpha = residual / factor;
            weight = (1.0f - alpha * alpha);
            weight = weight * weight;
          }
        }
        /* semantic weighting, Frame2Model_jacobians.geom:143-158 */
        suma_float4 sd4 = current->semantic[pix];
        float data_label = sd4.x * 255.0f, data_prob = sd4.w;
        float model_label = sm4.x * 255.0f;
        if (o_is_dynamic_label(model_label)) {
          if (sdm_round(data_label) != sdm_round(model_label))
            weight *= (1.0f - data_prob);
          else
            weight *= data_prob;
        }
        float wr2 = (weight * residual) * residual;
        if (inlier) {
          float J[6] = {n.x, n.y, n.z, cp.x, cp.y, cp.z};
          int k = 0;
          for (int i = 0; i < 6; ++i)
            for (int j = i; j < 6; ++j) acc[k++] += o_fix((weight * J[i]) * J[j]);
          for (int i = 0; i < 6; ++i) acc[21 + i] += o_fix((weight * residual) * J[i]);
          acc[27] += o_fix(wr2);
          acc[28] += o_fix(wr2);
          acc[29] += 1;
        } else {
          acc[27] += o_fix(wr2);
          acc[29] += 1;
          acc[30] += 1;
        }
      } else {
        acc[31] += 1;
      }
    }
  }

  if (acc_out) memcpy(acc_out, acc, sizeof(acc));
  const double inv = 1.0 / SUMA_ACC_SCALE;
  if (JtJ) {
    int k = 0;
    for (int i = 0; i < 6; ++i)
      for (int j = i; j < 6; ++j) {
        double v = (double)acc[k++] * inv;
        JtJ[6 * j + i] = v;
        JtJ[6 * i + j] = v;
      }
  }
  if (Jtr)
    for (int i = 0; i < 6; ++i) Jtr[i] = (double)acc[21 + i] * inv;
  double F = (double)acc[27] * inv;
  if (st) {
    st->error = F;
    st->inlier_residual = (double)acc[28] * inv;
    st->valid = (uint32_t)acc[29];
    st->outlier = (uint32_t)acc[30];
    st->inlier = st->valid - st->outlier;
    st->invalid = (uint32_t)acc[31];
  }
  return F;
}

/* JtJ.ldlt().solve(-Jtf), LieGaussNewton.cpp:60.  Unpivoted LDL^T in fp64 (Eigen pivots; for
 * the SPD systems of this problem both return the same solution to fp64 rounding). */
void ora_solve6(const double* A, const double* b, double* x) {
  double L[36], D[6], y[6];
  memset(L, 0, sizeof(L));
  for (int j = 0; j < 6; ++j) {
    double d = A[6 * j + j];
    for (int k = 0; k < j; ++k) d -= (L[6 * k + j] * L[6 * k + j]) * D[k];
    D[j] = d;
    L[6 * j + j] = 1.0;
    for (int i = j + 1; i < 6; ++i) {
      double s = A[6 * j + i];
      for (int k = 0; k < j; ++k) s -= (L[6 * k + i] * L[6 * k + j]) * D[k];
      L[6 * j + i] = s / d;
    }
  }
  for (int i = 0; i < 6; ++i) { /* L y = -b */
    double s = -b[i];
    for (int k = 0; k < i; ++k) s -= L[6 * k + i] * y[k];
    y[i] = s;
  }
  for (int i = 0; i < 6; ++i) y[i] = y[i] / D[i];
  for (int i = 5; i >= 0; --i) { /* L^T x = y */
    double s = y[i];
    for (int k = i + 1; k < 6; ++k) s -= L[6 * i + k] * x[k];
    x[i] = s;
  }
}

static void o_mul4d(const double* A, const double* B, double* C) {
  for (int c = 0; c < 4; ++c)
    for (int r = 0; r < 4; ++r)
      C[4 * c + r] =
          ((A[r] * B[4 * c] + A[4 + r] * B[4 * c + 1]) + A[8 + r] * B[4 * c + 2]) + A[12 + r] * B[4 * c + 3];
}

/* SE3::exp, lie_algebra.cpp:4-34; x = (v, omega); column-major output */
void ora_se3_exp(const double x[6], double T[16]) {
  for (int i = 0; i < 16; ++i) T[i] = (i % 5 == 0) ? 1.0 : 0.0;
  const double v[3] = {x[0], x[1], x[2]}, o[3] = {x[3], x[4], x[5]};
  double theta = sqrt((o[0] * o[0] + o[1] * o[1]) + o[2] * o[2]);
  if (theta > 1e-10) {
    /* K = skew(omega), row-major K[r][c] */
    double K[9] = {0, -o[2], o[1], o[2], 0, -o[0], -o[1], o[0], 0};
    double K2[9];
    for (int r = 0; r < 3; ++r)
      for (int cc = 0; cc < 3; ++cc)
        K2[3 * r + cc] = (K[3 * r] * K[cc] + K[3 * r + 1] * K[3 + cc]) + K[3 * r + 2] * K[6 + cc];
    double alpha = sdm_sin_d(theta) / theta;
    double beta = (1 - sdm_cos_d(theta)) / (theta * theta);
    double gamma = (1.0 - sdm_cos_d(theta)) / (theta * theta);
    double delta = (theta - sdm_sin_d(theta)) / (theta * theta * theta);
    for (int r = 0; r < 3; ++r) {
      double t = 0.0;
      for (int cc = 0; cc < 3; ++cc) {
        double I = (r == cc) ? 1.0 : 0.0;
        T[4 * cc + r] = (I + alpha * K[3 * r + cc]) + beta * K2[3 * r + cc];
        double Vrc = (I + gamma * K[3 * r + cc]) + delta * K2[3 * r + cc];
        t += Vrc * v[cc];
      }
      T[12 + r] = t;
    }
  } else {
    T[12] = v[0];
    T[13] = v[1];
    T[14] = v[2];
  }
}

/* LieGaussNewton::minimize / step.  history receives every Tk pushed at LieGaussNewton.cpp:24.
 * iteration0 = Frame2Model::iteration_ when the minimisation starts.  ONLY Frame2Model::setData resets that counter
 * (Frame2Model.cpp:117-123); LieGaussNewton::initialize -> Objective::initialize (LieGaussNewton.cpp:36-51,
 * Objective.h:58) sets the pose and nothing else, and every Objective::increment advances it (Objective.h:45-48).  A
 * caller that minimises twice on one setData -- the loop over the initial guesses of checkLoopClosure,
 * SurfelMapping.cpp:693-700 -- therefore starts its later minimisations with iteration_ > 0, which the Tukey weight
 * reads (Frame2Model_jacobians.geom:129: `iteration > 0`). */
void ora_icp_minimize_from(ora_ctx* c, const ora_frame* current, const ora_frame* model, const double T0[16],
                           uint32_t iteration0, double T_out[16], double* history, uint32_t history_cap,
                           uint32_t* n_hist, suma_icp_stats* st) {
  const uint32_t max_iter = c->p.max_iterations;
  const double epsilon = (double)c->p.stopping_threshold, delta = (double)c->p.delta;
  double Tk[16];
  memcpy(Tk, T0, sizeof(Tk));
  double last_error = (double)3.402823466e+38f; /* std::numeric_limits<float>::max(), :48 */
  uint32_t iteration = iteration0;
  uint32_t k = 0, nh = 0, converged = 0;
  suma_icp_stats s;
  memset(&s, 0, sizeof(s));
  for (;;) {
    if (history && nh < history_cap) memcpy(history + 16 * (size_t)nh, Tk, sizeof(Tk));
    nh++;
    if (max_iter > 0 && k >= max_iter) break;
    double JtJ[36], Jtr[6], dx[6];
    double err = ora_icp_jacobian_products(c, current, model, Tk, iteration, NULL, JtJ, Jtr, &s);
    ora_solve6(JtJ, Jtr, dx);
    int result = 1;
    double linf = 0.0, maxc = Jtr[0];
    for (int i = 0; i < 6; ++i) {
      if (fabs(dx[i]) > linf) linf = fabs(dx[i]);
      if (Jtr[i] > maxc) maxc = Jtr[i];
    }
    if (linf < delta) result = 0;                                            /* :64 */
    if (fabs(maxc) < epsilon) result = 0;                                    /* :65 (quirk B-4) */
    if (err < last_error && fabs(err - last_error) < epsilon) result = 0;    /* :66 */
    double E[16], Tn[16];
    ora_se3_exp(dx, E);
    o_mul4d(E, Tk, Tn); /* pose_ = SE3::exp(delta) * pose_, Objective.h:46 -- applied even when converged */
    memcpy(Tk, Tn, sizeof(Tk));
    iteration += 1;
    last_error = err;
    if (result == 0) {
      converged = 1;
      break;
    }
    ++k;
  }
  memcpy(T_out, Tk, sizeof(Tk));
  if (n_hist) *n_hist = nh;
  if (st) {
    *st = s;
    st->iterations = k;
    st->converged = converged;
  }
}

/* the minimisation right behind a setData (iteration_ = 0): every call site of the scan path (SurfelMapping.cpp:384-388,
 * 443-445, 553-554) */
void ora_icp_minimize(ora_ctx* c, const ora_frame* current, const ora_frame* model, const double T0[16],
                      double T_out[16], double* history, uint32_t history_cap, uint32_t* n_hist,
                      suma_icp_stats* st) {
  ora_icp_minimize_from(c, current, model, T0, 0u, T_out, history, history_cap, n_hist, st);
}
