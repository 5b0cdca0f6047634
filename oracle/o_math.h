/*
 * oracle/o_math.h -- TEST INFRASTRUCTURE (CPU oracle), never shipped or linked by the product.
 *
 * Small fp32 vector / matrix helpers with a FIXED operation order, restating the GLSL built-ins
 * the reference's shaders use (dot, cross, length, normalize, mat4*vec4, mat4*mat4).  The
 * operation order below IS the specification; the gfx950 kernels restate the same order so the
 * two can be compared bit for bit.  Compiled with -ffp-contract=off.
 *
 * Round 5: the sums of products inside these built-ins are chains of EXPLICIT fused multiply-adds
 * (dot = fma(z, z', fma(y, y', x * x')), mat * vec = fma over the columns in order), and vector / scalar
 * is three multiplies by ONE correctly rounded reciprocal -- the forms a GPU's GLSL compiler
 * emits, and the fastest on gfx950 (include/suma_detmath.h, SDM_MA).  Statements of the shaders
 * themselves (a * b + c written out in the shader text) stay unfused.
 */
#ifndef ORACLE_O_MATH_H_
#define ORACLE_O_MATH_H_

#include <math.h>
#include <stdint.h>
#include <string.h>

#include "../include/suma_detmath.h"
#include "../include/suma_types.h"

#ifdef ORACLE_USE_LIBM
/* independence cross-check build: glibc transcendental functions instead of the shared
 * deterministic ones (results then agree only to tolerance, never used for bit parity) */
#define sdm_atan2 atan2f
#define sdm_asin asinf
#define sdm_acos acosf
#define sdm_sin sinf
#define sdm_exp expf
#define sdm_log logf
#define sdm_sin_d sin
#define sdm_cos_d cos
#endif

typedef struct {
  float x, y, z;
} ov3;

static inline ov3 ov3_make(float x, float y, float z) {
  ov3 r = {x, y, z};
  return r;
}
#define O_FMA(a, b, c) __builtin_fmaf((a), (b), (c))
static inline float ov3_dot(ov3 a, ov3 b) { return O_FMA(a.z, b.z, O_FMA(a.y, b.y, a.x * b.x)); }
static inline float ov3_len(ov3 a) { return sdm_sqrt(ov3_dot(a, a)); }
static inline ov3 ov3_sub(ov3 a, ov3 b) { return ov3_make(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline ov3 ov3_add(ov3 a, ov3 b) { return ov3_make(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline ov3 ov3_scale(float s, ov3 a) { return ov3_make(s * a.x, s * a.y, s * a.z); }
/* GLSL vec / float: one correctly rounded reciprocal, three multiplies */
static inline ov3 ov3_divs(ov3 a, float s) {
  const float r = 1.0f / s;
  return ov3_make(a.x * r, a.y * r, a.z * r);
}
static inline ov3 ov3_neg(ov3 a) { return ov3_make(-a.x, -a.y, -a.z); }
/* GLSL normalize(): v / length(v) */
static inline ov3 ov3_normalize(ov3 a) { return ov3_divs(a, ov3_len(a)); }
static inline ov3 ov3_cross(ov3 a, ov3 b) {
  return ov3_make(O_FMA(a.y, b.z, -(a.z * b.y)), O_FMA(a.z, b.x, -(a.x * b.z)), O_FMA(a.x, b.y, -(a.y * b.x)));
}

/* column-major 4x4 (Eigen::Matrix4f / GLSL mat4): element (row r, col c) = m[4*c + r] */
/* M * (p, 1) : fma(col2, z, fma(col1, y, col0 * x)) + col3 (= fma(col3, 1, .)), xyz rows only */
static inline ov3 om4_point(const float* m, ov3 p) {
  ov3 r;
  r.x = O_FMA(m[8], p.z, O_FMA(m[4], p.y, m[0] * p.x)) + m[12];
  r.y = O_FMA(m[9], p.z, O_FMA(m[5], p.y, m[1] * p.x)) + m[13];
  r.z = O_FMA(m[10], p.z, O_FMA(m[6], p.y, m[2] * p.x)) + m[14];
  return r;
}
/* M * (d, 0) */
static inline ov3 om4_dir(const float* m, ov3 d) {
  ov3 r;
  r.x = O_FMA(m[8], d.z, O_FMA(m[4], d.y, m[0] * d.x));
  r.y = O_FMA(m[9], d.z, O_FMA(m[5], d.y, m[1] * d.x));
  r.z = O_FMA(m[10], d.z, O_FMA(m[6], d.y, m[2] * d.x));
  return r;
}
/* C = A * B, each element fma(a3, b3, fma(a2, b2, fma(a1, b1, a0 * b0))) */
static inline void om4_mul(const float* A, const float* B, float* C) {
  for (int c = 0; c < 4; ++c)
    for (int r = 0; r < 4; ++r)
      C[4 * c + r] = O_FMA(A[12 + r], B[4 * c + 3], O_FMA(A[8 + r], B[4 * c + 2], O_FMA(A[4 + r], B[4 * c + 1], A[r] * B[4 * c])));
}
/* Inverse of a rigid transform, evaluated in double from the fp32 matrix and rounded once to
 * fp32: R^T, -R^T t.  (The reference uses Eigen's / GLSL's general inverse on rigid poses,
 * SurfelMap.cpp:497,875 and update_surfels.vert:197; those agree with this to fp32 rounding.) */
static inline void om4_rigid_inverse(const float* m, float* out) {
  double R[9], t[3];
  for (int c = 0; c < 3; ++c)
    for (int r = 0; r < 3; ++r) R[3 * c + r] = (double)m[4 * c + r];
  for (int r = 0; r < 3; ++r) t[r] = (double)m[12 + r];
  for (int c = 0; c < 3; ++c)
    for (int r = 0; r < 3; ++r) out[4 * c + r] = (float)R[3 * r + c];
  for (int r = 0; r < 3; ++r) {
    double s = (R[3 * r + 0] * t[0] + R[3 * r + 1] * t[1]) + R[3 * r + 2] * t[2];
    out[12 + r] = (float)(-s);
  }
  out[3] = out[7] = out[11] = 0.0f;
  out[15] = 1.0f;
}

static inline float of_clamp(float x, float lo, float hi) {
  float t = (x < lo) ? lo : x; /* max(x, lo) */
  return (t > hi) ? hi : t;    /* min(t, hi) */
}
static inline float of_min(float a, float b) { return (b < a) ? b : a; }
static inline float of_max(float a, float b) { return (a < b) ? b : a; }

/* color.glsl:31-37 pack() */
static inline float o_pack(float r, float g, float b) {
  int32_t rgb = (int32_t)sdm_round(r * 255.0f);
  rgb = (rgb << 8) + (int32_t)sdm_round(g * 255.0f);
  rgb = (rgb << 8) + (int32_t)sdm_round(b * 255.0f);
  return (float)rgb;
}

/* color_map.glsl:8-17: the movable classes the shaders single out */
static inline int o_is_dynamic_label(float label) {
  return label == 10.0f || label == 11.0f || label == 13.0f || label == 15.0f || label == 18.0f || label == 20.0f ||
         label == 30.0f || label == 31.0f || label == 32.0f;
}

/* 24-bit unorm depth of a window-space z in [0,1] (GL_DEPTH24_STENCIL8 renderbuffers,
 * Preprocessing.cpp:56, SurfelMap.cpp:103,113,172) */
/* window depth -> 24-bit unorm of a GL_DEPTH24_STENCIL8 renderbuffer: the fp32 product rounded to the nearest integer,
 * ties to even.  Pinned in round 4 against Mesa llvmpipe running through oracle/glref.py (tests/test_gl_reference.py:
 * 16384 pairs of fragments a few 2^-25 apart, zero disagreements); rounds 1-3 used (uint32_t)(zw * 16777215.0f + 0.5f),
 * whose fp32 ADD rounds a second time above 2^23 (odd values came out one too large). */
static inline uint32_t o_depth24(float zw) { return (uint32_t)__builtin_rintf(zw * 16777215.0f); }

#endif
