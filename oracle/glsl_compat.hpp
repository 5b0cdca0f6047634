/*
 * oracle/glsl_compat.hpp -- TEST INFRASTRUCTURE.  Never shipped, never linked by the product.
 *
 * A small GLSL 3.30 -> C++ compatibility layer, so that the reference's OWN shader sources
 * (/root/reference/src/shader/ *.vert|.geom|.frag, read at build time by oracle/ref_build.py, never copied into
 * this repository) can be compiled with g++ and executed on the CPU: "the reference compiled here" for the
 * per-vertex / per-fragment arithmetic (oracle/_ref/libsuma_ref.so).
 *
 * What is restated here is the GLSL language / built-in library, not the reference:
 *   vec2/vec3/vec4 (with xyzw and rgba swizzles as lvalues and rvalues), bvec3, mat3, mat4, the component-wise
 *   operators, the built-in functions the hot-path shaders call, and rectangle / buffer samplers with
 *   NEAREST / LINEAR filtering and CLAMP_TO_BORDER (border colour 0), GL 3.3 core spec 3.8.
 * Every operation is IEEE binary32 in the natural left-to-right order of the GLSL specification's formulas
 * (dot = x*x' + y*y' + z*z', normalize = v / length(v), mat*vec = sum of column * component); since round 5 the sums
 * of products INSIDE the built-ins are explicit fused multiply-add chains and vector / scalar multiplies by one
 * correctly rounded reciprocal (what a GPU's GLSL compiler emits; the arithmetic of the shader TEXT stays unfused:
 * compiled with -ffp-contract=off).  Transcendental functions are a build switch: REF_USE_LIBM selects glibc's, otherwise the
 * repository's deterministic ones (include/suma_detmath.h) -- GL leaves their last bits to the driver.
 */
#ifndef ORACLE_GLSL_COMPAT_HPP_
#define ORACLE_GLSL_COMPAT_HPP_

#include <cmath>
#include <cstdint>
#include <cstring>
#include <type_traits>

#include "../include/suma_detmath.h"

namespace glsl {

struct vec2;
struct vec3;
struct vec4;

/* ---- swizzle proxies: empty structs living in a union with the components ---- */
template <int A, int B>
struct swz2 {
  const float* p() const { return reinterpret_cast<const float*>(this); }
  float* p() { return reinterpret_cast<float*>(this); }
  inline operator vec2() const;
  inline swz2& operator=(const vec2& v);
  inline swz2& operator=(const swz2& o);
};
template <int A, int B, int C>
struct swz3 {
  const float* p() const { return reinterpret_cast<const float*>(this); }
  float* p() { return reinterpret_cast<float*>(this); }
  inline operator vec3() const;
  inline swz3& operator=(const vec3& v);
  inline swz3& operator=(const swz3& o);
};
template <int A, int B, int C, int D>
struct swz4 {
  const float* p() const { return reinterpret_cast<const float*>(this); }
  float* p() { return reinterpret_cast<float*>(this); }
  inline operator vec4() const;
  inline swz4& operator=(const vec4& v);
  inline swz4& operator=(const swz4& o);
};

struct vec2 {
  enum { N = 2 };
  union {
    struct {
      float x, y;
    };
    struct {
      float r, g;
    };
    float d[2];
#include "glsl_swz_vec2.inc"
  };
  vec2() : x(0.f), y(0.f) {}
  explicit vec2(float s) : x(s), y(s) {}
  vec2(float a, float b) : x(a), y(b) {}
  vec2(const vec2& o) : x(o.x), y(o.y) {}
  inline explicit vec2(const vec3& o);
  inline explicit vec2(const vec4& o);
  vec2& operator=(const vec2& o) {
    x = o.x;
    y = o.y;
    return *this;
  }
  float& operator[](int i) { return d[i]; }
  const float& operator[](int i) const { return d[i]; }
  template <class B>
  vec2& operator+=(const B& b);
  template <class B>
  vec2& operator-=(const B& b);
  template <class B>
  vec2& operator*=(const B& b);
  template <class B>
  vec2& operator/=(const B& b);
};

struct vec3 {
  enum { N = 3 };
  union {
    struct {
      float x, y, z;
    };
    struct {
      float r, g, b;
    };
    float d[3];
#include "glsl_swz_vec3.inc"
  };
  vec3() : x(0.f), y(0.f), z(0.f) {}
  explicit vec3(float s) : x(s), y(s), z(s) {}
  vec3(float a, float b_, float c) : x(a), y(b_), z(c) {}
  vec3(const vec2& a, float c) : x(a.x), y(a.y), z(c) {}
  vec3(float a, const vec2& bc) : x(a), y(bc.x), z(bc.y) {}
  vec3(const vec3& o) : x(o.x), y(o.y), z(o.z) {}
  inline explicit vec3(const vec4& o);
  vec3& operator=(const vec3& o) {
    x = o.x;
    y = o.y;
    z = o.z;
    return *this;
  }
  float& operator[](int i) { return d[i]; }
  const float& operator[](int i) const { return d[i]; }
  template <class B>
  vec3& operator+=(const B& b);
  template <class B>
  vec3& operator-=(const B& b);
  template <class B>
  vec3& operator*=(const B& b);
  template <class B>
  vec3& operator/=(const B& b);
};

struct vec4 {
  enum { N = 4 };
  union {
    struct {
      float x, y, z, w;
    };
    struct {
      float r, g, b, a;
    };
    float d[4];
#include "glsl_swz_vec4.inc"
  };
  vec4() : x(0.f), y(0.f), z(0.f), w(0.f) {}
  explicit vec4(float s) : x(s), y(s), z(s), w(s) {}
  vec4(float a_, float b_, float c, float e) : x(a_), y(b_), z(c), w(e) {}
  vec4(const vec3& v, float e) : x(v.x), y(v.y), z(v.z), w(e) {}
  vec4(float a_, const vec3& v) : x(a_), y(v.x), z(v.y), w(v.z) {}
  vec4(const vec2& v, float c, float e) : x(v.x), y(v.y), z(c), w(e) {}
  vec4(const vec2& u, const vec2& v) : x(u.x), y(u.y), z(v.x), w(v.y) {}
  vec4(const vec4& o) : x(o.x), y(o.y), z(o.z), w(o.w) {}
  vec4& operator=(const vec4& o) {
    x = o.x;
    y = o.y;
    z = o.z;
    w = o.w;
    return *this;
  }
  float& operator[](int i) { return d[i]; }
  const float& operator[](int i) const { return d[i]; }
  /* GLSL: a scalar constructed from a vector takes its first component (int(texture(..)), update_surfels.vert) */
  explicit operator int() const { return (int)x; }
  explicit operator float() const { return x; }
  template <class B>
  vec4& operator+=(const B& b);
  template <class B>
  vec4& operator-=(const B& b);
  template <class B>
  vec4& operator*=(const B& b);
  template <class B>
  vec4& operator/=(const B& b);
};

inline vec2::vec2(const vec3& o) : x(o.x), y(o.y) {}
inline vec2::vec2(const vec4& o) : x(o.x), y(o.y) {}
inline vec3::vec3(const vec4& o) : x(o.x), y(o.y), z(o.z) {}

template <int A, int B>
inline swz2<A, B>::operator vec2() const {
  return vec2(p()[A], p()[B]);
}
template <int A, int B>
inline swz2<A, B>& swz2<A, B>::operator=(const vec2& v) {
  static_assert(A != B, "repeated component in swizzle assignment");
  p()[A] = v.x;
  p()[B] = v.y;
  return *this;
}
template <int A, int B>
inline swz2<A, B>& swz2<A, B>::operator=(const swz2& o) {
  return *this = vec2(o);
}
template <int A, int B, int C>
inline swz3<A, B, C>::operator vec3() const {
  return vec3(p()[A], p()[B], p()[C]);
}
template <int A, int B, int C>
inline swz3<A, B, C>& swz3<A, B, C>::operator=(const vec3& v) {
  static_assert(A != B && A != C && B != C, "repeated component in swizzle assignment");
  p()[A] = v.x;
  p()[B] = v.y;
  p()[C] = v.z;
  return *this;
}
template <int A, int B, int C>
inline swz3<A, B, C>& swz3<A, B, C>::operator=(const swz3& o) {
  return *this = vec3(o);
}
template <int A, int B, int C, int D>
inline swz4<A, B, C, D>::operator vec4() const {
  return vec4(p()[A], p()[B], p()[C], p()[D]);
}
template <int A, int B, int C, int D>
inline swz4<A, B, C, D>& swz4<A, B, C, D>::operator=(const vec4& v) {
  static_assert(A != B && A != C && A != D && B != C && B != D && C != D, "repeated component");
  p()[A] = v.x;
  p()[B] = v.y;
  p()[C] = v.z;
  p()[D] = v.w;
  return *this;
}
template <int A, int B, int C, int D>
inline swz4<A, B, C, D>& swz4<A, B, C, D>::operator=(const swz4& o) {
  return *this = vec4(o);
}

/* ---- component-wise operators over {vecN, N-swizzle, scalar} ---- */
template <class T>
struct vt {
  enum { n = 0 };
  typedef void type;
};
template <>
struct vt<vec2> {
  enum { n = 2 };
  typedef vec2 type;
};
template <>
struct vt<vec3> {
  enum { n = 3 };
  typedef vec3 type;
};
template <>
struct vt<vec4> {
  enum { n = 4 };
  typedef vec4 type;
};
template <int A, int B>
struct vt<swz2<A, B> > {
  enum { n = 2 };
  typedef vec2 type;
};
template <int A, int B, int C>
struct vt<swz3<A, B, C> > {
  enum { n = 3 };
  typedef vec3 type;
};
template <int A, int B, int C, int D>
struct vt<swz4<A, B, C, D> > {
  enum { n = 4 };
  typedef vec4 type;
};

template <class A, class B>
struct binres {
  enum {
    na = vt<A>::n,
    nb = vt<B>::n,
    sa = std::is_arithmetic<A>::value,
    sb = std::is_arithmetic<B>::value,
    ok = (na > 0 && (nb == na || sb)) || (sa && nb > 0)
  };
  typedef typename std::conditional<(na > 0), typename vt<A>::type, typename vt<B>::type>::type type;
};
template <class A, class B>
using binres_t = typename std::enable_if<binres<A, B>::ok, typename binres<A, B>::type>::type;
template <class A>
using unres_t = typename std::enable_if<(vt<A>::n > 0), typename vt<A>::type>::type;

/* broadcast / convert an operand to the result vector type R */
template <class R, class T>
inline typename std::enable_if<std::is_arithmetic<T>::value, R>::type bc(const T& s) {
  return R((float)s);
}
template <class R, class T>
inline typename std::enable_if<(vt<T>::n > 0), R>::type bc(const T& v) {
  return R(v);
}

#define GLSL_BINOP(OP)                                      \
  template <class A, class B>                               \
  inline binres_t<A, B> operator OP(const A& a, const B& b) { \
    typedef binres_t<A, B> R;                               \
    R x = bc<R>(a), y = bc<R>(b), r;                        \
    for (int i = 0; i < R::N; ++i) r[i] = x[i] OP y[i];     \
    return r;                                               \
  }
GLSL_BINOP(+)
GLSL_BINOP(-)
GLSL_BINOP(*)
#undef GLSL_BINOP
/* operator/ : vector / SCALAR is a multiplication by ONE correctly rounded reciprocal (round 5: what a GPU's GLSL
 * compiler emits for it -- v * rcp(s) -- with the rcp exact; oracle/o_math.h ov3_divs and csrc/dev_math.h divs3 are the
 * same three operations); vector / vector and scalar / vector divide component by component */
template <class A, class B>
inline binres_t<A, B> operator/(const A& a, const B& b) {
  typedef binres_t<A, B> R;
  R x = bc<R>(a), y = bc<R>(b), r;
  if (vt<A>::n > 0 && std::is_arithmetic<B>::value) {
    const float rcp = 1.0f / y[0];
    for (int i = 0; i < R::N; ++i) r[i] = x[i] * rcp;
  } else {
    for (int i = 0; i < R::N; ++i) r[i] = x[i] / y[i];
  }
  return r;
}
template <class A>
inline unres_t<A> operator-(const A& a) {
  typedef unres_t<A> R;
  R x = R(a), r;
  for (int i = 0; i < R::N; ++i) r[i] = -x[i];
  return r;
}
#define GLSL_COMPOUND(V)                 \
  template <class B>                     \
  inline V& V::operator+=(const B& b) {  \
    return *this = *this + b;            \
  }                                      \
  template <class B>                     \
  inline V& V::operator-=(const B& b) {  \
    return *this = *this - b;            \
  }                                      \
  template <class B>                     \
  inline V& V::operator*=(const B& b) {  \
    return *this = *this * b;            \
  }                                      \
  template <class B>                     \
  inline V& V::operator/=(const B& b) {  \
    return *this = *this / b;            \
  }
GLSL_COMPOUND(vec2)
GLSL_COMPOUND(vec3)
GLSL_COMPOUND(vec4)
#undef GLSL_COMPOUND

struct bvec3 {
  bool x, y, z;
};
inline bvec3 lessThan(const vec3& a, const vec3& b) { return bvec3{a.x < b.x, a.y < b.y, a.z < b.z}; }
inline bvec3 greaterThanEqual(const vec3& a, const vec3& b) { return bvec3{a.x >= b.x, a.y >= b.y, a.z >= b.z}; }
inline bool all(const bvec3& b) { return b.x && b.y && b.z; }
inline bool any(const bvec3& b) { return b.x || b.y || b.z; }

/* ---- scalar built-ins ---- */
#ifdef REF_USE_LIBM
inline float sin(float x) { return ::sinf(x); }
inline float cos(float x) { return ::cosf(x); }
inline float asin(float x) { return ::asinf(x); }
inline float acos(float x) { return ::acosf(x); }
inline float atan(float y, float x) { return ::atan2f(y, x); }
inline float atan(float x) { return ::atanf(x); }
inline float exp(float x) { return ::expf(x); }
inline float log(float x) { return ::logf(x); }
#else
inline float sin(float x) { return sdm_sin(x); }
inline float cos(float x) { return sdm_cos(x); }
inline float asin(float x) { return sdm_asin(x); }
inline float acos(float x) { return sdm_acos(x); }
inline float atan(float y, float x) { return sdm_atan2(y, x); }
inline float atan(float x) { return sdm_atan(x); }
inline float exp(float x) { return sdm_exp(x); }
inline float log(float x) { return sdm_log(x); }
#endif
inline float sqrt(float x) { return __builtin_sqrtf(x); }
inline float abs(float x) { return __builtin_fabsf(x); }
inline float floor(float x) { return __builtin_floorf(x); }
inline float fract(float x) { return x - __builtin_floorf(x); }
/* round(): GLSL leaves the direction for x.5 to the implementation; to even, as the GL implementations do (llvmpipe:
 * measured, tests/test_gl_reference.py) -- nearbyint in the default rounding mode */
inline float round(float x) { return __builtin_rintf(x); }
inline float min(float a, float b) { return (b < a) ? b : a; }                 /* GLSL: y < x ? y : x */
inline float max(float a, float b) { return (a < b) ? b : a; }                 /* GLSL: x < y ? y : x */
inline float clamp(float x, float lo, float hi) { return min(max(x, lo), hi); } /* GLSL definition */
inline float mix(float a, float b, float t) { return a * (1.0f - t) + b * t; }
inline float step(float edge, float x) { return x < edge ? 0.0f : 1.0f; }
inline float degrees(float r) { return r * 57.295779513082320877f; }
inline float radians(float d) { return d * 0.017453292519943295769f; }

/* ---- vector built-ins ---- */
/* sums of products inside the built-ins: chains of explicit fused multiply-adds in component order (round 5,
 * include/suma_detmath.h SDM_MA; the compiler forms none of its own: -ffp-contract=off) */
inline float fma_(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
inline float dot(const vec2& a, const vec2& b) { return fma_(a.y, b.y, a.x * b.x); }
inline float dot(const vec3& a, const vec3& b) { return fma_(a.z, b.z, fma_(a.y, b.y, a.x * b.x)); }
inline float dot(const vec4& a, const vec4& b) { return fma_(a.w, b.w, fma_(a.z, b.z, fma_(a.y, b.y, a.x * b.x))); }
inline float length(const vec2& a) { return sqrt(dot(a, a)); }
inline float length(const vec3& a) { return sqrt(dot(a, a)); }
inline float length(const vec4& a) { return sqrt(dot(a, a)); }
inline float length(float a) { return abs(a); }
inline vec2 normalize(const vec2& a) { return a / length(a); }
inline vec3 normalize(const vec3& a) { return a / length(a); }
inline vec4 normalize(const vec4& a) { return a / length(a); }
inline float distance(const vec3& a, const vec3& b) { return length(a - b); }
inline vec3 cross(const vec3& a, const vec3& b) {
  /* GLSL spec 8.5: x[1] y[2] - y[1] x[2], ...; the first product fused into the subtraction */
  return vec3(fma_(a.y, b.z, -(b.y * a.z)), fma_(a.z, b.x, -(b.z * a.x)), fma_(a.x, b.y, -(b.x * a.y)));
}
#define GLSL_MAP1(F)                                                   \
  inline vec2 F(const vec2& a) { return vec2(F(a.x), F(a.y)); }        \
  inline vec3 F(const vec3& a) { return vec3(F(a.x), F(a.y), F(a.z)); } \
  inline vec4 F(const vec4& a) { return vec4(F(a.x), F(a.y), F(a.z), F(a.w)); }
GLSL_MAP1(abs)
GLSL_MAP1(floor)
GLSL_MAP1(fract)
GLSL_MAP1(round)
#undef GLSL_MAP1
inline vec3 clamp(const vec3& v, float lo, float hi) { return vec3(clamp(v.x, lo, hi), clamp(v.y, lo, hi), clamp(v.z, lo, hi)); }
inline vec4 clamp(const vec4& v, float lo, float hi) {
  return vec4(clamp(v.x, lo, hi), clamp(v.y, lo, hi), clamp(v.z, lo, hi), clamp(v.w, lo, hi));
}
inline vec3 mix(const vec3& a, const vec3& b, float t) { return a * (1.0f - t) + b * t; }
inline vec4 mix(const vec4& a, const vec4& b, float t) { return a * (1.0f - t) + b * t; }

/* ---- matrices (column major) ---- */
struct mat4;
struct mat3 {
  vec3 c[3];
  mat3() {}
  explicit mat3(float s) {
    c[0] = vec3(s, 0, 0);
    c[1] = vec3(0, s, 0);
    c[2] = vec3(0, 0, s);
  }
  mat3(const vec3& a, const vec3& b, const vec3& d) {
    c[0] = a;
    c[1] = b;
    c[2] = d;
  }
  inline explicit mat3(const mat4& m);
  vec3& operator[](int i) { return c[i]; }
  const vec3& operator[](int i) const { return c[i]; }
};
struct mat4 {
  vec4 c[4];
  mat4() {}
  explicit mat4(float s) {
    c[0] = vec4(s, 0, 0, 0);
    c[1] = vec4(0, s, 0, 0);
    c[2] = vec4(0, 0, s, 0);
    c[3] = vec4(0, 0, 0, s);
  }
  mat4(const vec4& a, const vec4& b, const vec4& d, const vec4& e) {
    c[0] = a;
    c[1] = b;
    c[2] = d;
    c[3] = e;
  }
  explicit mat4(const mat3& m) {
    c[0] = vec4(m.c[0], 0.f);
    c[1] = vec4(m.c[1], 0.f);
    c[2] = vec4(m.c[2], 0.f);
    c[3] = vec4(0.f, 0.f, 0.f, 1.f);
  }
  vec4& operator[](int i) { return c[i]; }
  const vec4& operator[](int i) const { return c[i]; }
};
inline mat3::mat3(const mat4& m) {
  c[0] = vec3(m.c[0]);
  c[1] = vec3(m.c[1]);
  c[2] = vec3(m.c[2]);
}
/* linear-algebraic products, accumulated in column order with fused multiply-adds: fma(c3, w, fma(c2, z, fma(c1, y, c0 x))) */
inline vec4 operator*(const mat4& m, const vec4& v) {
  vec4 r;
  for (int i = 0; i < 4; ++i) r[i] = fma_(m.c[3][i], v.w, fma_(m.c[2][i], v.z, fma_(m.c[1][i], v.y, m.c[0][i] * v.x)));
  return r;
}
inline vec3 operator*(const mat3& m, const vec3& v) {
  vec3 r;
  for (int i = 0; i < 3; ++i) r[i] = fma_(m.c[2][i], v.z, fma_(m.c[1][i], v.y, m.c[0][i] * v.x));
  return r;
}
inline mat4 operator*(const mat4& a, const mat4& b) { return mat4(a * b.c[0], a * b.c[1], a * b.c[2], a * b.c[3]); }
inline mat3 operator*(const mat3& a, const mat3& b) { return mat3(a * b.c[0], a * b.c[1], a * b.c[2]); }
inline mat3 operator-(const mat3& a) { return mat3(-a.c[0], -a.c[1], -a.c[2]); }
inline mat3 transpose(const mat3& m) {
  return mat3(vec3(m.c[0].x, m.c[1].x, m.c[2].x), vec3(m.c[0].y, m.c[1].y, m.c[2].y), vec3(m.c[0].z, m.c[1].z, m.c[2].z));
}
/* inverse(mat4): GLSL does not prescribe an algorithm.  inverse_cofactor: cofactor expansion (adjugate /
 * determinant) in binary32, the form GLSL compilers commonly lower it to. */
inline mat4 inverse_cofactor(const mat4& M) {
  float m[16], inv[16];
  for (int c = 0; c < 4; ++c)
    for (int r = 0; r < 4; ++r) m[4 * c + r] = M.c[c][r];
  inv[0] = m[5] * m[10] * m[15] - m[5] * m[11] * m[14] - m[9] * m[6] * m[15] + m[9] * m[7] * m[14] + m[13] * m[6] * m[11] - m[13] * m[7] * m[10];
  inv[4] = -m[4] * m[10] * m[15] + m[4] * m[11] * m[14] + m[8] * m[6] * m[15] - m[8] * m[7] * m[14] - m[12] * m[6] * m[11] + m[12] * m[7] * m[10];
  inv[8] = m[4] * m[9] * m[15] - m[4] * m[11] * m[13] - m[8] * m[5] * m[15] + m[8] * m[7] * m[13] + m[12] * m[5] * m[11] - m[12] * m[7] * m[9];
  inv[12] = -m[4] * m[9] * m[14] + m[4] * m[10] * m[13] + m[8] * m[5] * m[14] - m[8] * m[6] * m[13] - m[12] * m[5] * m[10] + m[12] * m[6] * m[9];
  inv[1] = -m[1] * m[10] * m[15] + m[1] * m[11] * m[14] + m[9] * m[2] * m[15] - m[9] * m[3] * m[14] - m[13] * m[2] * m[11] + m[13] * m[3] * m[10];
  inv[5] = m[0] * m[10] * m[15] - m[0] * m[11] * m[14] - m[8] * m[2] * m[15] + m[8] * m[3] * m[14] + m[12] * m[2] * m[11] - m[12] * m[3] * m[10];
  inv[9] = -m[0] * m[9] * m[15] + m[0] * m[11] * m[13] + m[8] * m[1] * m[15] - m[8] * m[3] * m[13] - m[12] * m[1] * m[11] + m[12] * m[3] * m[9];
  inv[13] = m[0] * m[9] * m[14] - m[0] * m[10] * m[13] - m[8] * m[1] * m[14] + m[8] * m[2] * m[13] + m[12] * m[1] * m[10] - m[12] * m[2] * m[9];
  inv[2] = m[1] * m[6] * m[15] - m[1] * m[7] * m[14] - m[5] * m[2] * m[15] + m[5] * m[3] * m[14] + m[13] * m[2] * m[7] - m[13] * m[3] * m[6];
  inv[6] = -m[0] * m[6] * m[15] + m[0] * m[7] * m[14] + m[4] * m[2] * m[15] - m[4] * m[3] * m[14] - m[12] * m[2] * m[7] + m[12] * m[3] * m[6];
  inv[10] = m[0] * m[5] * m[15] - m[0] * m[7] * m[13] - m[4] * m[1] * m[15] + m[4] * m[3] * m[13] + m[12] * m[1] * m[7] - m[12] * m[3] * m[5];
  inv[14] = -m[0] * m[5] * m[14] + m[0] * m[6] * m[13] + m[4] * m[1] * m[14] - m[4] * m[2] * m[13] - m[12] * m[1] * m[6] + m[12] * m[2] * m[5];
  inv[3] = -m[1] * m[6] * m[11] + m[1] * m[7] * m[10] + m[5] * m[2] * m[11] - m[5] * m[3] * m[10] - m[9] * m[2] * m[7] + m[9] * m[3] * m[6];
  inv[7] = m[0] * m[6] * m[11] - m[0] * m[7] * m[10] - m[4] * m[2] * m[11] + m[4] * m[3] * m[10] + m[8] * m[2] * m[7] - m[8] * m[3] * m[6];
  inv[11] = -m[0] * m[5] * m[11] + m[0] * m[7] * m[9] + m[4] * m[1] * m[11] - m[4] * m[3] * m[9] - m[8] * m[1] * m[7] + m[8] * m[3] * m[5];
  inv[15] = m[0] * m[5] * m[10] - m[0] * m[6] * m[9] - m[4] * m[1] * m[10] + m[4] * m[2] * m[9] + m[8] * m[1] * m[6] - m[8] * m[2] * m[5];
  float det = m[0] * inv[0] + m[1] * inv[4] + m[2] * inv[8] + m[3] * inv[12];
  float id = 1.0f / det;
  mat4 R;
  for (int c = 0; c < 4; ++c)
    for (int r = 0; r < 4; ++r) R.c[c][r] = inv[4 * c + r] * id;
  return R;
}

/* The hot path only inverts rigid poses (update_surfels.vert:197).  Like the transcendental functions, the
 * implementation is a build switch: REF_USE_LIBM takes the general cofactor form above; otherwise the rigid form
 * of the repository (R^T, -R^T t evaluated in double and rounded once -- oracle/o_math.h om4_rigid_inverse), which is
 * the correctly-rounded-er of the two on rigid input (tests/test_ref_shaders.py measures the difference). */
inline mat4 inverse(const mat4& M) {
#ifdef REF_USE_LIBM
  return inverse_cofactor(M);
#else
  double R[3][3], t[3];
  for (int c = 0; c < 3; ++c)
    for (int r = 0; r < 3; ++r) R[c][r] = (double)M.c[c][r];
  for (int r = 0; r < 3; ++r) t[r] = (double)M.c[3][r];
  mat4 O(0.0f);
  for (int c = 0; c < 3; ++c)
    for (int r = 0; r < 3; ++r) O.c[c][r] = (float)R[r][c];
  for (int r = 0; r < 3; ++r) O.c[3][r] = (float)(-((R[r][0] * t[0] + R[r][1] * t[1]) + R[r][2] * t[2]));
  O.c[3][3] = 1.0f;
  return O;
#endif
}

/* ---- samplers: rectangle textures (unnormalised coordinates); CLAMP_TO_BORDER with border colour 0 (the sampler
 * objects the reference binds) or CLAMP_TO_EDGE (the GL initial state of a rectangle texture) ---- */
enum { GLSL_NEAREST = 0, GLSL_LINEAR = 1, GLSL_CLAMP_TO_EDGE = 2 /* flag, or-ed into the filter argument */ };
struct sampler2DRect {
  const float* data;
  int w, h, ch; /* ch = 1 (R32F: (r,0,0,1)) or 4 (RGBA32F) */
  int filter;
  int edge;
  sampler2DRect() : data(nullptr), w(0), h(0), ch(4), filter(GLSL_NEAREST), edge(0) {}
};
inline vec4 glsl_texel(const sampler2DRect& s, int i, int j) {
  if (s.edge && s.data) {
    i = i < 0 ? 0 : (i >= s.w ? s.w - 1 : i);
    j = j < 0 ? 0 : (j >= s.h ? s.h - 1 : j);
  }
  if (!s.data || i < 0 || j < 0 || i >= s.w || j >= s.h) return vec4(0.f, 0.f, 0.f, 0.f);
  const float* t = s.data + ((size_t)j * (size_t)s.w + (size_t)i) * (size_t)s.ch;
  return s.ch == 4 ? vec4(t[0], t[1], t[2], t[3]) : vec4(t[0], 0.f, 0.f, 1.f);
}
inline vec2 textureSize(const sampler2DRect& s) { return vec2((float)s.w, (float)s.h); }
inline vec4 texture(const sampler2DRect& s, const vec2& c) {
  if (!(c.x == c.x) || !(c.y == c.y)) return vec4(0.f, 0.f, 0.f, 0.f); /* NaN: undefined in GL; border here */
  if (s.filter == GLSL_NEAREST) {
    float fi = floor(c.x), fj = floor(c.y);
    if (!(fi >= -1.0f && fi <= (float)s.w && fj >= -1.0f && fj <= (float)s.h)) return vec4(0.f, 0.f, 0.f, 0.f);
    return glsl_texel(s, (int)fi, (int)fj);
  }
  /* GL 3.3 core 3.8.11: i0 = floor(u - 1/2), alpha = frac(u - 1/2); tau = (1-a)(1-b) t00 + a(1-b) t10 + (1-a) b t01 + ab t11 */
  float u = c.x - 0.5f, v = c.y - 0.5f;
  float fu = floor(u), fv = floor(v);
  if (!(fu >= -2.0f && fu <= (float)s.w + 1.0f && fv >= -2.0f && fv <= (float)s.h + 1.0f)) return vec4(0.f, 0.f, 0.f, 0.f);
  float a = u - fu, b = v - fv;
  int i0 = (int)fu, j0 = (int)fv;
  vec4 t00 = glsl_texel(s, i0, j0), t10 = glsl_texel(s, i0 + 1, j0), t01 = glsl_texel(s, i0, j0 + 1), t11 = glsl_texel(s, i0 + 1, j0 + 1);
  /* a tap whose weight is exactly 0 contributes nothing, whatever it holds: at a texel centre (a = b = 0) the
   * fetch is the texel itself even beside a NaN / inf texel (GL 3.3 core 2.1.1 leaves arithmetic on NaN / inf
   * undefined; 0 * NaN = NaN would poison every texel left of / below a NaN one).  The remaining products are added
   * in the order of the spec's formula. */
  const float w[4] = {(1.0f - a) * (1.0f - b), a * (1.0f - b), (1.0f - a) * b, a * b};
  const vec4* t[4] = {&t00, &t10, &t01, &t11};
  vec4 r(0.f, 0.f, 0.f, 0.f);
  bool any = false;
  for (int k = 0; k < 4; ++k) {
    if (w[k] == 0.0f) continue;
    r = any ? r + *t[k] * w[k] : *t[k] * w[k];
    any = true;
  }
  return r;
}
struct samplerBuffer {
  const float* data; /* RGBA32F texels */
  int n;
  samplerBuffer() : data(nullptr), n(0) {}
};
inline vec4 texelFetch(const samplerBuffer& s, int i) {
  if (!s.data || i < 0 || i >= s.n) return vec4(0.f, 0.f, 0.f, 0.f);
  return vec4(s.data[4 * i], s.data[4 * i + 1], s.data[4 * i + 2], s.data[4 * i + 3]);
}

}  // namespace glsl
#endif
