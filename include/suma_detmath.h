/*
 * suma_detmath.h -- deterministic single-precision elementary functions.
 *
 * WHY THIS EXISTS.  The reference computes atan/asin/acos/sin/exp/log inside GLSL shaders
 * (e.g. src/shader/gen_vertexmap.vert:80-81, update_surfels.vert:116-121,238-243), where the
 * GL driver picks the implementation and its last-ulp behaviour.  The projective pipeline then
 * feeds those values through floor() to obtain pixel indices, so a 1-ulp difference between two
 * implementations flips integer outputs (pixel assignment, surfel counts).  To make "bit-exact
 * surfel indices/counts" a testable property between the CPU oracle and the gfx950 kernels,
 * the transcendental functions are part of the *specification*: both sides evaluate the very
 * same sequence of IEEE-754 binary32 +,-,*,/,sqrt,fma operations (no FMA CONTRACTION: every
 * translation unit including this header is compiled with -ffp-contract=off, the fused steps
 * below are explicit; hipcc's default correctly rounded fp32 divide/sqrt is relied upon and
 * checked by tests/test_detmath.py).
 *
 * Algorithms: classic Cephes single-precision kernels (S. Moshier, public domain algorithms:
 * atanf/asinf/sinf/expf/logf range reductions + minimax polynomials), restated here.
 * Accuracy (measured in tests/test_detmath.py against libm in double): <= 2 ulp on the
 * domains used by the pipeline.
 *
 * C99 / C++ / HIP compatible, header-only.
 */
#ifndef SUMA_DETMATH_H_
#define SUMA_DETMATH_H_

#include <stdint.h>

#if defined(__HIPCC__)
#define SUMA_HD __host__ __device__ static inline
#else
#define SUMA_HD static inline
#endif

/* One multiply-add step of a polynomial kernel: FUSED, a * b + c with one rounding (round 5; rounds 1-4 specified the
 * unfused form).  The arithmetic specification of this repository is a choice -- GLSL leaves the evaluation of its
 * built-in functions to the implementation, and a real GL (Mesa llvmpipe) was measured to differ from any fixed choice
 * in 1-2 % of the texels through its own asin alone -- so it is chosen where the target is
 * fastest: v_fma_f32 is one instruction on gfx950, mul + add are two (profiles/r05_spec_v2_experiment.txt: +3.2 %
 * scans/s for the whole change).  Fused multiply-adds appear ONLY where they are written out (here, and in the vector /
 * matrix helpers of oracle/o_math.h, oracle/glsl_compat.hpp and csrc/dev_math.h); every translation unit is still
 * compiled with -ffp-contract=off, so the compiler forms none of its own.  On the host __builtin_fmaf is the
 * correctly rounded fma (vfmadd with -mfma, glibc's fmaf otherwise): the same bits as the device's. */
#define SDM_MA(a, b, c) __builtin_fmaf((a), (b), (c))

#define SUMA_PI_F 3.14159265358979323846f
#define SUMA_PI_2_F 1.57079632679489661923f
#define SUMA_PI_4_F 0.78539816339744830962f
/* the literal used by every projection in the reference (gen_vertexmap.vert:19) */
#define SUMA_INV_PI_F 0.31830988618379067154f
/* GLSL degrees(): 180/pi */
#define SUMA_RAD2DEG_F 57.295779513082320877f

SUMA_HD uint32_t sdm_f2u(float f) {
  uint32_t u;
  __builtin_memcpy(&u, &f, 4);
  return u;
}
SUMA_HD float sdm_u2f(uint32_t u) {
  float f;
  __builtin_memcpy(&f, &u, 4);
  return f;
}
SUMA_HD float sdm_abs(float x) { return sdm_u2f(sdm_f2u(x) & 0x7fffffffu); }
SUMA_HD int sdm_isnan(float x) { return (sdm_f2u(x) & 0x7fffffffu) > 0x7f800000u; }

/* exact floor for |x| < 2^31, NaN -> NaN, large values returned unchanged */
SUMA_HD float sdm_floor(float x) {
  if (!(sdm_abs(x) < 8388608.0f)) return x; /* already integral, inf or NaN */
  float t = (float)(int32_t)x;              /* truncation */
  return (t > x) ? (t - 1.0f) : t;
}

/* GLSL round(): "the fraction 0.5 will round in a direction chosen by the implementation".  Ties DO occur on this path
 * -- pack(vec3(0.3)) and pack(vec3(0, 0.7, 0)) of color.glsl are 76.5 and 178.5 -- and the implementations that exist
 * round them to EVEN (the hardware's round-to-nearest instruction; Mesa llvmpipe does, measured by
 * tests/test_gl_reference.py through the reference's own update_surfels.vert), not away from zero as C's roundf does
 * (rounds 1-3 of this repository).  Exact for |x| < 2^23, larger values are integral already. */
SUMA_HD float sdm_round(float x) {
  if (!(sdm_abs(x) < 8388608.0f)) return x;
  float a = sdm_abs(x);
  float t = (float)(int32_t)a; /* truncation; a - t is exact */
  float f = a - t;
  if (f > 0.5f || (f == 0.5f && (((int32_t)t) & 1))) t = t + 1.0f;
  return (x < 0.0f) ? -t : t;
}

SUMA_HD float sdm_atan(float xx) {
  float x = sdm_abs(xx);
  float y;
  if (x > 2.414213562373095f) { /* tan(3pi/8) */
    y = SUMA_PI_2_F;
    x = -(1.0f / x);
  } else if (x > 0.4142135623730950f) { /* tan(pi/8) */
    y = SUMA_PI_4_F;
    x = (x - 1.0f) / (x + 1.0f);
  } else {
    y = 0.0f;
  }
  float z = x * x;
  float p = SDM_MA(8.05374449538e-2f, z, -1.38776856032e-1f);
  p = SDM_MA(p, z, 1.99777106478e-1f);
  p = SDM_MA(p, z, -3.33329491539e-1f);
  p = SDM_MA(p * z, x, x);
  y = y + p;
  return (xx < 0.0f) ? -y : y;
}

/* atan2 with the C quadrant conventions; (0,0) -> 0 */
SUMA_HD float sdm_atan2(float y, float x) {
  if (sdm_isnan(x) || sdm_isnan(y)) return x + y;
  if (x == 0.0f) {
    if (y > 0.0f) return SUMA_PI_2_F;
    if (y < 0.0f) return -SUMA_PI_2_F;
    return 0.0f;
  }
  float z = sdm_atan(y / x);
  if (x < 0.0f) {
    if (y < 0.0f)
      z = z - SUMA_PI_F;
    else
      z = z + SUMA_PI_F;
  }
  return z;
}

/* correctly rounded on both sides: hipcc lowers __builtin_sqrtf to the IEEE sequence under its default
 * -fhip-fp32-correctly-rounded-divide-sqrt (HIP's __fsqrt_rn is NOT: it maps to the native
 * approximation, measured 15 % 1-ulp mismatches on gfx950 -- tools/fpcheck.hip) */
SUMA_HD float sdm_sqrt(float x) { return __builtin_sqrtf(x); }

SUMA_HD float sdm_asin(float xx) {
  float a = sdm_abs(xx);
  if (!(a <= 1.0f)) return sdm_u2f(0x7fc00000u); /* NaN, also for NaN input */
  float x, z;
  int flag = 0;
  if (a < 1.0e-4f) return xx;
  if (a > 0.5f) {
    z = 0.5f * (1.0f - a);
    x = sdm_sqrt(z);
    flag = 1;
  } else {
    x = a;
    z = x * x;
  }
  float p = SDM_MA(4.2163199048e-2f, z, 2.4181311049e-2f);
  p = SDM_MA(p, z, 4.5470025998e-2f);
  p = SDM_MA(p, z, 7.4953002686e-2f);
  p = SDM_MA(p, z, 1.6666752422e-1f);
  z = SDM_MA(p * z, x, x);
  if (flag) {
    z = z + z;
    z = SUMA_PI_2_F - z;
  }
  return (xx < 0.0f) ? -z : z;
}

SUMA_HD float sdm_acos(float x) {
  if (!(sdm_abs(x) <= 1.0f)) return sdm_u2f(0x7fc00000u);
  if (x < -0.5f) return SUMA_PI_F - 2.0f * sdm_asin(sdm_sqrt(0.5f * (1.0f + x)));
  if (x > 0.5f) return 2.0f * sdm_asin(sdm_sqrt(0.5f * (1.0f - x)));
  return SUMA_PI_2_F - sdm_asin(x);
}

/* sin / cos for |x| <= 8192 (the pipeline uses [0, pi]) */
SUMA_HD float sdm_sincos_core(float xx, int want_cos) {
  const float DP1 = 0.78515625f, DP2 = 2.4187564849853515625e-4f, DP3 = 3.77489497744594108e-8f;
  const float FOPI = 1.27323954473516f; /* 4/pi */
  int sign = 1;
  float x = xx;
  if (x < 0.0f) {
    x = -x;
    if (!want_cos) sign = -1;
  }
  if (!(x <= 8192.0f)) return sdm_u2f(0x7fc00000u);
  int32_t j = (int32_t)(FOPI * x);
  float y = (float)j;
  if (j & 1) {
    j += 1;
    y += 1.0f;
  }
  j &= 7;
  if (j > 3) {
    sign = -sign;
    j -= 4;
  }
  if (want_cos && j > 1) sign = -sign;
  x = ((x - y * DP1) - y * DP2) - y * DP3;
  float z = x * x;
  float r;
  int use_cos_poly = want_cos ? !(j == 1 || j == 2) : (j == 1 || j == 2);
  if (use_cos_poly) {
    float p = SDM_MA(2.443315711809948e-5f, z, -1.388731625493765e-3f);
    p = SDM_MA(p, z, 4.166664568298827e-2f);
    r = p * z * z;
    r = r - 0.5f * z;
    r = r + 1.0f;
  } else {
    float p = SDM_MA(-1.9515295891e-4f, z, 8.3321608736e-3f);
    p = SDM_MA(p, z, -1.6666654611e-1f);
    r = SDM_MA(p * z, x, x);
  }
  return (sign < 0) ? -r : r;
}
SUMA_HD float sdm_sin(float x) { return sdm_sincos_core(x, 0); }
SUMA_HD float sdm_cos(float x) { return sdm_sincos_core(x, 1); }

/* 2^n * x for results in the normal range, exact */
SUMA_HD float sdm_ldexp(float x, int32_t n) {
  /* split the scaling so each factor is a normal power of two */
  while (n > 127) {
    x = x * sdm_u2f(0x7f000000u); /* 2^127 */
    n -= 127;
  }
  while (n < -126) {
    x = x * sdm_u2f(0x00800000u); /* 2^-126 */
    n += 126;
  }
  return x * sdm_u2f((uint32_t)(n + 127) << 23);
}

SUMA_HD float sdm_exp(float xx) {
  if (sdm_isnan(xx)) return xx;
  if (xx > 88.72283905206835f) return sdm_u2f(0x7f800000u);
  if (xx < -103.278929903431851103f) return 0.0f;
  const float LOG2EF = 1.44269504088896341f;
  const float C1 = 0.693359375f, C2 = -2.12194440e-4f;
  float x = xx;
  float z = sdm_floor(LOG2EF * x + 0.5f);
  x = x - z * C1;
  x = x - z * C2;
  int32_t n = (int32_t)z;
  z = x * x;
  float p = SDM_MA(1.9875691500e-4f, x, 1.3981999507e-3f);
  p = SDM_MA(p, x, 8.3334519073e-3f);
  p = SDM_MA(p, x, 4.1665795894e-2f);
  p = SDM_MA(p, x, 1.6666665459e-1f);
  p = SDM_MA(p, x, 5.0000001201e-1f);
  z = SDM_MA(p, z, x) + 1.0f;
  return sdm_ldexp(z, n);
}

SUMA_HD float sdm_log(float xx) {
  if (sdm_isnan(xx)) return xx;
  if (xx < 0.0f) return sdm_u2f(0x7fc00000u);
  if (xx == 0.0f) return sdm_u2f(0xff800000u);
  if (sdm_f2u(xx) == 0x7f800000u) return xx;
  /* frexp: x = m * 2^e, m in [0.5, 1) */
  uint32_t u = sdm_f2u(xx);
  int32_t e = 0;
  if ((u >> 23) == 0) { /* subnormal: scale up by 2^24 */
    xx = xx * 16777216.0f;
    u = sdm_f2u(xx);
    e = -24;
  }
  e += (int32_t)(u >> 23) - 126;
  float x = sdm_u2f((u & 0x007fffffu) | 0x3f000000u);
  if (x < 0.707106781186547524f) {
    e -= 1;
    x = x + x - 1.0f;
  } else {
    x = x - 1.0f;
  }
  float z = x * x;
  float p = SDM_MA(7.0376836292e-2f, x, -1.1514610310e-1f);
  p = SDM_MA(p, x, 1.1676998740e-1f);
  p = SDM_MA(p, x, -1.2420140846e-1f);
  p = SDM_MA(p, x, 1.4249322787e-1f);
  p = SDM_MA(p, x, -1.6668057665e-1f);
  p = SDM_MA(p, x, 2.0000714765e-1f);
  p = SDM_MA(p, x, -2.4999993993e-1f);
  p = SDM_MA(p, x, 3.3333331174e-1f);
  float y = p * x * z;
  float fe = (float)e;
  y = y + -2.12194440e-4f * fe;
  y = y - 0.5f * z;
  z = x + y;
  z = z + 0.693359375f * fe;
  return z;
}

/* ---- double-precision sin / cos (SE3::exp of the Gauss-Newton update, reference
 * src/core/lie_algebra.cpp:4-34, runs on the device in the gfx950 build and on the host in the
 * oracle; both must produce the same pose bits).  Cephes double kernels restated; |x| <= 2^30;
 * accuracy ~1 ulp (tests/test_detmath.py). */
SUMA_HD double sdm_floor_d(double x) {
  if (!((x < 0 ? -x : x) < 4503599627370496.0)) return x;
  double t = (double)(int64_t)x;
  return (t > x) ? (t - 1.0) : t;
}
SUMA_HD double sdm_sincos_core_d(double xx, int want_cos) {
  const double DP1 = 7.85398125648498535156E-1, DP2 = 3.77489470793079817668E-8, DP3 = 2.69515142907905952645E-15;
  const double FOPI = 1.27323954473516268615; /* 4/pi */
  int sign = 1;
  double x = xx;
  if (x < 0.0) {
    x = -x;
    if (!want_cos) sign = -1;
  }
  if (!(x <= 1073741824.0)) return xx - xx; /* NaN for inf / NaN, 0 beyond the supported range */
  double y = sdm_floor_d(x * FOPI);
  int64_t j = (int64_t)y;
  if (j & 1) {
    j += 1;
    y += 1.0;
  }
  j &= 7;
  if (j > 3) {
    sign = -sign;
    j -= 4;
  }
  if (want_cos && j > 1) sign = -sign;
  double z = ((x - y * DP1) - y * DP2) - y * DP3;
  double zz = z * z;
  int use_cos_poly = want_cos ? !(j == 1 || j == 2) : (j == 1 || j == 2);
  double r;
  if (use_cos_poly) {
    double p = -1.13585365213876817300E-11 * zz + 2.08757008419747316778E-9;
    p = p * zz - 2.75573141792967388112E-7;
    p = p * zz + 2.48015872888517045348E-5;
    p = p * zz - 1.38888888888730564116E-3;
    p = p * zz + 4.16666666666665929218E-2;
    r = (1.0 - 0.5 * zz) + (zz * zz) * p;
  } else {
    double p = 1.58962301576546568060E-10 * zz - 2.50507477628578072866E-8;
    p = p * zz + 2.75573136213857245213E-6;
    p = p * zz - 1.98412698295895385996E-4;
    p = p * zz + 8.33333333332211858878E-3;
    p = p * zz - 1.66666666666666307295E-1;
    r = z + (z * zz) * p;
  }
  return (sign < 0) ? -r : r;
}
SUMA_HD double sdm_sin_d(double x) { return sdm_sincos_core_d(x, 0); }
SUMA_HD double sdm_cos_d(double x) { return sdm_sincos_core_d(x, 1); }
SUMA_HD double sdm_sqrt_d(double x) { return __builtin_sqrt(x); }

#endif /* SUMA_DETMATH_H_ */
