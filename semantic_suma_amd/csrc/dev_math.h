/*
 * dev_math.h -- fp32 vector / matrix helpers of the gfx950 kernels.
 *
 * The projective pipeline turns floats into pixel indices through floor(), so the ORDER of the
 * fp32 operations is part of the specification (DESIGN.md "Determinism"): every expression below
 * is written as an explicit sequence of IEEE binary32 +,-,*,/,sqrt,fma with the association the
 * reference's GLSL built-ins imply (dot, cross, length, normalize, mat4*vec4, mat4*mat4;
 * e.g. src/shader/gen_vertexmap.vert:73-103).  All translation units are compiled with
 * -ffp-contract=off so the compiler forms no FMA of its own; hipcc's fp32 divide / sqrt are correctly rounded.
 * Transcendentals come from include/suma_detmath.h.
 */
#ifndef SUMA_DEV_MATH_H_
#define SUMA_DEV_MATH_H_

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/suma_detmath.h"
#include "../../include/suma_types.h"

#define SDEV __device__ __forceinline__

struct v3 {
  float x, y, z;
};

SDEV v3 mk3(float x, float y, float z) {
  v3 r;
  r.x = x;
  r.y = y;
  r.z = z;
  return r;
}
SDEV v3 xyz(const float4& a) { return mk3(a.x, a.y, a.z); }
/* Round 5: the sums of products inside the GLSL built-ins are chains of EXPLICIT fused multiply-adds (v_fma_f32: one
 * instruction where mul + add are two), and vector / scalar is three multiplies by ONE correctly rounded reciprocal --
 * the same operations, in the same order, as the CPU checker and the compiled reference shaders use (profiles/r05_spec_v2_experiment.txt:
 * -7 ... -21 % VALU instructions in the map kernels, +3.2 % scans/s).  The compiler still forms no fma of its own. */
#define SDEV_FMA(a, b, c) __builtin_fmaf((a), (b), (c))
SDEV float dot3(v3 a, v3 b) { return SDEV_FMA(a.z, b.z, SDEV_FMA(a.y, b.y, a.x * b.x)); }
SDEV float len3(v3 a) { return sdm_sqrt(dot3(a, a)); }
SDEV v3 sub3(v3 a, v3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
SDEV v3 add3(v3 a, v3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
SDEV v3 scale3(float s, v3 a) { return mk3(s * a.x, s * a.y, s * a.z); }
/* GLSL vec / float */
SDEV v3 divs3(v3 a, float s) {
  const float r = 1.0f / s;
  return mk3(a.x * r, a.y * r, a.z * r);
}
SDEV v3 neg3(v3 a) { return mk3(-a.x, -a.y, -a.z); }
SDEV v3 normalize3(v3 a) { return divs3(a, len3(a)); } /* GLSL normalize(): v / length(v) */
SDEV v3 cross3(v3 a, v3 b) {
  return mk3(SDEV_FMA(a.y, b.z, -(a.z * b.y)), SDEV_FMA(a.z, b.x, -(a.x * b.z)), SDEV_FMA(a.x, b.y, -(a.y * b.x)));
}

/* column-major 4x4 by value (kernel argument / register resident) */
struct m4 {
  float m[16];
};

/* M * (p, 1): fma(col2, z, fma(col1, y, col0 * x)) + col3 */
SDEV v3 m4_point(const float* m, v3 p) {
  v3 r;
  r.x = SDEV_FMA(m[8], p.z, SDEV_FMA(m[4], p.y, m[0] * p.x)) + m[12];
  r.y = SDEV_FMA(m[9], p.z, SDEV_FMA(m[5], p.y, m[1] * p.x)) + m[13];
  r.z = SDEV_FMA(m[10], p.z, SDEV_FMA(m[6], p.y, m[2] * p.x)) + m[14];
  return r;
}
/* M * (d, 0) */
SDEV v3 m4_dir(const float* m, v3 d) {
  v3 r;
  r.x = SDEV_FMA(m[8], d.z, SDEV_FMA(m[4], d.y, m[0] * d.x));
  r.y = SDEV_FMA(m[9], d.z, SDEV_FMA(m[5], d.y, m[1] * d.x));
  r.z = SDEV_FMA(m[10], d.z, SDEV_FMA(m[6], d.y, m[2] * d.x));
  return r;
}
/* C = A * B: each element fma(a3, b3, fma(a2, b2, fma(a1, b1, a0 * b0))) */
SDEV void m4_mul(const float* A, const float* B, float* C) {
#pragma unroll
  for (int c = 0; c < 4; ++c) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
      C[4 * c + r] = SDEV_FMA(A[12 + r], B[4 * c + 3], SDEV_FMA(A[8 + r], B[4 * c + 2], SDEV_FMA(A[4 + r], B[4 * c + 1], A[r] * B[4 * c])));
  }
}

SDEV float fclamp(float x, float lo, float hi) {
  float t = (x < lo) ? lo : x;
  return (t > hi) ? hi : t;
}
SDEV float fmin_(float a, float b) { return (b < a) ? b : a; }
SDEV float fmax_(float a, float b) { return (a < b) ? b : a; }

/* color.glsl:31-37 pack() */
SDEV float pack_rgb(float r, float g, float b) {
  int32_t rgb = (int32_t)sdm_round(r * 255.0f);
  rgb = (rgb << 8) + (int32_t)sdm_round(g * 255.0f);
  rgb = (rgb << 8) + (int32_t)sdm_round(b * 255.0f);
  return (float)rgb;
}

/* color_map.glsl:8-17: car, bicycle, bus, motorcycle, truck, other-vehicle, person, bicyclist,
 * motorcyclist -- the classes the shaders treat as movable */
SDEV bool is_dynamic_label(float l) {
  return l == 10.0f || l == 11.0f || l == 13.0f || l == 15.0f || l == 18.0f || l == 20.0f || l == 30.0f ||
         l == 31.0f || l == 32.0f;
}

/* 24-bit unorm depth (GL_DEPTH24_STENCIL8 renderbuffers of the reference) */
/* nearest-even of the fp32 product (v_rndne_f32): the rule of a real GL implementation (Mesa llvmpipe, pinned in
 * round 4: DESIGN.md section 2); rounds 1-3 used (uint32_t)(zw * 16777215.0f + 0.5f), whose fp32 add rounds twice */
SDEV uint32_t depth24(float zw) { return (uint32_t)__builtin_rintf(zw * 16777215.0f); }

/* spherical projection parameters of one image (data or model) */
struct proj_t {
  float fov_up, fov, min_depth, max_depth, width, height;
  int32_t W, H;
};

/* the projection repeated in five shaders (gen_indexmap.vert:37-52 et al.): (x01, y01, z01) */
SDEV v3 project01(const proj_t& q, v3 p) {
  float depth = len3(p);
  float yaw = sdm_atan2(p.y, p.x);
  float pitch = -sdm_asin(p.z / depth);
  v3 r;
  r.x = 0.5f * ((-yaw * SUMA_INV_PI_F) + 1.0f);
  r.y = 1.0f - ((pitch * SUMA_RAD2DEG_F) + q.fov_up) / q.fov;
  r.z = (depth - q.min_depth) / (q.max_depth - q.min_depth);
  return r;
}

SDEV float4 f4(float x, float y, float z, float w) { return make_float4(x, y, z, w); }
/* NEAREST + CLAMP_TO_BORDER texel fetch (border colour 0).  Branch-free on purpose: the load is
 * issued from a clamped (always valid) address and the border case is a select, so that a group
 * of fetches (5-point stencil, 4 bilinear taps x 3 maps) is ONE batch of loads in flight instead
 * of a chain of dependent round trips through exec-masked branches. */
SDEV float4 texel(const float4* __restrict__ map, int32_t w, int32_t h, int32_t x, int32_t y) {
  const bool inside = (x >= 0) & (y >= 0) & (x < w) & (y < h);
  const int32_t xc = min(max(x, 0), w - 1), yc = min(max(y, 0), h - 1);
  const float4 v = map[(size_t)yc * (size_t)w + (size_t)xc];
  return inside ? v : f4(0.f, 0.f, 0.f, 0.f);
}

/* inverse of a rigid transform evaluated in double from the fp32 matrix, rounded once:
 * R^T, -R^T t (the reference uses a general inverse, update_surfels.vert:197; equal to fp32
 * rounding on rigid poses) */
__device__ __forceinline__ void rigid_inverse_dev(const float* m, float* out) {
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

/* Streaming store for the surfel stream-out of K9 / K10 (tens of MB that nothing reads before the next pass
 * over the map): `nt` keeps the lines from piling up dirty in the XCD's L2.  Dirty lines are written back at
 * the END of a kernel, XCD by XCD, and the next kernel's workgroups do not start on an XCD before its
 * write-back is done (per-block start times of K10 behind K9: 0.6 us on the first XCD, 2.3 - 5.2 us on the
 * others; +1.4 % scans/s).  NOT for the frame maps / K8 products: the kernels that follow gather from them,
 * and with `nt` those reads got slower than the write-back they saved (-2 %). */
typedef float suma_v4f __attribute__((ext_vector_type(4)));
SDEV void store_stream(float4* p, const float4& v) {
  suma_v4f t = {v.x, v.y, v.z, v.w};
  __builtin_nontemporal_store(t, reinterpret_cast<suma_v4f*>(p));
}

/* Workgroup barrier that orders LDS traffic only: __syncthreads() also drains every outstanding global
 * load / store / atomic of the wave (s_waitcnt vmcnt(0)), which turns loads issued early on purpose and
 * fire-and-forget stores into stalls.  Use where the data exchanged across the barrier lives in LDS. */
SDEV void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

/* K8 for one measurement pixel (init_radiusConf.vert): radius / validity, cleared integration mark, and
 * everything K9 gathers for the pixel packed into ONE 64-byte line (vertex, normal, label, label probability,
 * radius) -- a surfel update then costs one random line instead of four.  Depends on the frame only, not on
 * the pose: besides k8_radius, the pipeline's statistics pass (which streams the same three maps anyway)
 * runs it. */
struct K8Out {
  float4* radius_conf;
  uint8_t* integrated;
  float4* pixrec;
  float pixel_size, angle_thresh, min_radius, max_radius;
};
SDEV void k8_pixel(const K8Out& o, uint32_t pix, float4 v, float4 n, float4 sem) {
  v3 vv = xyz(v), nn = xyz(n);
  float d = len3(vv);
  v3 view_dir = divs3(neg3(vv), d);
  float angle = dot3(nn, view_dir);
  float valid = 0.0f, radius = 0.0f;
  if (v.w > 0.5f && n.w > 0.5f && angle > o.angle_thresh) {
    valid = 1.0f;
    radius = ((1.41f * d) * o.pixel_size) / fclamp(dot3(nn, divs3(neg3(vv), d)), 0.5f, 1.0f);
    radius = fmin_(fmax_(radius, o.min_radius), o.max_radius);
  }
  o.radius_conf[pix] = f4(radius, 0.0f, 0.0f, valid); /* quirk B-3: the confidence channel stays 0 */
  o.integrated[pix] = 0;
  float4* r = o.pixrec + 4 * (size_t)pix;
  r[0] = v;
  r[1] = n;
  r[2] = f4(sem.x, sem.w, radius, 0.0f);
}

#define SUMA_EMPTY_KEY (~0ull)

/* Depth-tested write into a 64-bit z-buffer (key = depth24 << 32 | id, smaller wins).  A pixel that many
 * primitives hit (far-range rows of a dense map collect hundreds of surfels) would serialise that many
 * read-modify-writes on one address; the current value only ever decreases, so a coherent (L2-bypassing)
 * load first lets every primitive that cannot win skip the atomic altogether.  Same final value. */
SDEV void zbuf_min(unsigned long long* __restrict__ addr, unsigned long long key) {
  if (key < __hip_atomic_load(addr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMin(addr, key);
}

/* inclusive prefix sum over the 64 lanes of a wave as DPP moves (six VALU-rate steps; six __shfl_up are six
 * ds_bpermute round trips in a dependent chain): Kogge-Stone inside each row of 16 (row_shr:1,2,4,8; a lane without
 * a source keeps old = 0), then lane 15 of rows 0 / 2 into rows 1 / 3 (row_bcast:15) and lane 31 into the upper half
 * (row_bcast:31) */
__device__ __forceinline__ uint32_t wave_inclusive_scan(uint32_t v) {
  int x = (int)v;
  x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xF, 0xF, false);
  x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xF, 0xF, false);
  x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xF, 0xF, false);
  x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xF, 0xF, false);
  x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xA, 0xF, false);
  x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xC, 0xF, false);
  return (uint32_t)x;
}


/* -DSUMA_PHASE_TIMING (tools/phase_timeline.py builds such a library next to the product one): thread 0 of every block
 * accumulates wall_clock64 (100 MHz) between the stations of its tiles and adds the totals to a per-kernel symbol */
#ifdef SUMA_PHASE_TIMING
#define PH_BEGIN unsigned long long ph_t = wall_clock64(), ph_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}
#define PH(k)                                         \
  do {                                                \
    if (threadIdx.x == 0) {                           \
      const unsigned long long now_ = wall_clock64(); \
      ph_acc[k] += now_ - ph_t;                       \
      ph_t = now_;                                    \
    }                                                 \
  } while (0)
#define PH_BLOCKS 8192 /* per-block slots: hot global counters would serialise at one memory channel and distort the run */
#define PH_END(sym)                                                              \
  do {                                                                           \
    if (threadIdx.x == 0 && blockIdx.x < PH_BLOCKS) {                            \
      for (int k_ = 0; k_ < 8; ++k_) sym[blockIdx.x][k_] += ph_acc[k_];            \
      sym[blockIdx.x][8] += 1ull;                                                \
    }                                                                            \
  } while (0)
#else
#define PH_BEGIN ((void)0)
#define PH(k) ((void)0)
#define PH_END(sym) ((void)0)
#endif

#endif
