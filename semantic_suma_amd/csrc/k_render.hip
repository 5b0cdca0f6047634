/*
 * k_render.hip -- K4 + K5: surfel-map rendering on gfx950 (a compute rasteriser).
 *
 * Replaces (reference):
 *   SurfelMap::render / render_active / render_inactive / render_composed
 *       src/core/SurfelMap.cpp:847-1165
 *   src/shader/render_surfels.vert:42-54  (surfel -> sensor frame via the creation pose)
 *   src/shader/render_surfels.geom:76-123 (visibility / stability / age gating, 4-corner quad)
 *   src/shader/render_surfels.frag:19-33  (unit-disc test, flat outputs)
 *   src/shader/render_compose.frag:26-48  (K5: per-pixel select new vs. old rendering)
 *
 * GL structure replaced: the reference draws all S surfels as GL_POINTS, expands each to a
 * screen-space quad in a geometry shader, lets the fixed-function rasteriser + 24-bit depth test
 * pick the nearest surfel per pixel and writes three RGBA32F targets -- and does that 4 times per
 * render() (old, new, and two dead passes into composedFrame_, quirk B-8) plus a compose pass.
 * Here:
 *   - ONE streaming pass over the surfel buffer (64 B per lane, grid-stride, size read from HBM)
 *     feeds up to two z-buffers (old / new) at once;
 *   - the triangle rasteriser is explicit: window coordinates snapped to 1/256 pixel, 64-bit
 *     integer edge functions with an antisymmetric tie rule, affine interpolation of depth and
 *     disc coordinates (gl_Position.w = 1 in render_surfels.geom); pixel tests are distributed
 *     over the wave (see k_render), not looped per surfel;
 *   - the depth test is a 64-bit atomicMin of (depth24 << 32 | surfel id): GL_LESS with in-order
 *     primitives = smaller depth, then lower id;
 *   - the resolve pass gathers the winning surfel's attributes, applies K5 and leaves the
 *     z-buffers cleared for the next render (no separate clear launch).
 */
#include <cstring>

#include "suma_internal.h"
#define EXDIV_FN __device__ __forceinline__
#include "exact_div.h"

enum { TIE_LOW_INDEX = 0, TIE_HIGH_INDEX_OLD = 1, TIE_HIGH_INDEX_NEW = 2 };

struct RenderSlot {
  int enabled;
  int mode; /* 0: old surfels (creation < thr); 1: new surfels (creation >= thr || timestamp >= thr) */
  int tie;
  unsigned long long* zbuf;
  m4 inv_pose;
};

struct RenderArgs {
  const suma_surfel* surfels;
  const DevState* ds;
  const float* poses;
  proj_t q;
  float conf_threshold;
  int use_stability;
  int32_t thr;
  RenderSlot slot[2];
  /* both slots look from the SAME pose with the same tie rule (render() without loop closing: pose_old == pose_new
   * bit for bit): one trip per tile serves both -- transform, gating and the quad of a surfel are computed once, a
   * 2-bit mask per record says which z-buffers its fragments go to */
  int merged;
  /* optional K7 (gen_indexmap.vert:62-81) fused into this pass: the index-map splat of the same
   * surfels from slot[0]'s pose into the data-sized z-buffer */
  const float* inv_pose_dev; /* if set: slot 0 (and the fused K7) take the inverse pose from HBM */
  int k7_enabled;
  int k7_same_proj; /* data and model images share one projection: the centre is projected once */
  proj_t k7_q;
  unsigned long long* k7_zbuf;
};

struct rvtx {
  int32_t X, Y; /* window coordinates in 1/256 pixel (|X| < 2^21: 20 bits of image + seam unwrap) */
  float z, tu, tv;
};

/* Edge function of the directed edge a->b at point (px, py): an exact integer below 2^47 in magnitude (the four
 * differences are below 2^23).  Formed in binary64: each product of two such integers is exact (46 bits), and the fused
 * multiply-add delivers their exact difference (representable, so its one rounding rounds nothing) -- the same integer
 * the 64-bit form (b.X - a.X) * (py - a.Y) - (b.Y - a.Y) * (px - a.X) computes, as a double.  Why: the 64-bit form costs
 * two 32 x 32 -> 64-bit multiplies, add-with-carry pairs and 64-bit compares per edge, and its conversion to float is a
 * dozen instructions (count leading zeros, shifts, rounding) where v_cvt_f32_f64 is one; binary64 issues at the ordinary
 * VALU rate on gfx950 (profiles/r05_instr_rate_gfx950.txt).  Per pair of pixel tests 831 -> 684 VALU instructions; the
 * pass is 5 % faster where it is issue-bound (50 M surfels) and unchanged at the steady 1 M map
 * (profiles/r05_second_session_experiments.txt).  The one value the integer form cannot produce is -0 (two zero products
 * of unlike sign): comparisons do not see it, and edge_to_float() turns it into the +0 an integer 0 converts to. */
__device__ __forceinline__ double edge_fn(const rvtx& a, const rvtx& b, int32_t px, int32_t py) {
  const double ux = (double)(b.X - a.X), uy = (double)(b.Y - a.Y), vx = (double)(px - a.X), vy = (double)(py - a.Y);
  return __builtin_fma(ux, vy, -(uy * vx));
}
__device__ __forceinline__ float edge_to_float(double w) { return (float)(w + 0.0); } /* (float)(long long): -0 -> +0 */
/* ownership of a pixel centre exactly on an edge: antisymmetric in the edge direction, so a
 * pixel on the diagonal shared by the two strip triangles is produced exactly once */
__device__ __forceinline__ bool owns_edge(const rvtx& s, const rvtx& t) {
  int32_t dx = t.X - s.X, dy = t.Y - s.Y;
  return dy > 0 || (dy == 0 && dx < 0);
}

__device__ __forceinline__ unsigned long long render_key(uint32_t z24, uint32_t id, int tie) {
  /* GL_LESS, in order: equal depth keeps the earlier primitive (lower id).  GL_LEQUAL
   * (render_composed, SurfelMap.cpp:1126): equal depth takes the later primitive, and the "new"
   * pass is drawn after the "old" pass into the same depth buffer. */
  if (tie == TIE_LOW_INDEX) return ((unsigned long long)z24 << 32) | id;
  unsigned long long pass = (tie == TIE_HIGH_INDEX_OLD) ? 1u : 0u;
  return ((unsigned long long)z24 << 33) | (pass << 32) | (unsigned long long)(0xffffffffu - id);
}

/* One triangle at one pixel centre: coverage (top-left style ownership rule), affine interpolation
 * of depth and disc coordinates, disc + near/far tests.  Returns the z-buffer key of the fragment or
 * SUMA_EMPTY_KEY if the triangle produces none.  Every quantity is a function of the triangle and the
 * pixel only, so the work can be distributed freely over lanes. */
__device__ __forceinline__ unsigned long long raster_key(rvtx A, rvtx B, rvtx C, int32_t i, int32_t j, uint32_t id,
                                                         int tie) {
  double area = edge_fn(A, B, C.X, C.Y);
  if (area < 0) {
    rvtx t = B;
    B = C;
    C = t;
    area = -area;
  }
  const int32_t px = 256 * i + 128, py = 256 * j + 128;
  const double w0 = edge_fn(B, C, px, py), w1 = edge_fn(C, A, px, py), w2 = edge_fn(A, B, px, py);
  const bool covered = (area != 0) && (w0 > 0 || (w0 == 0 && owns_edge(B, C))) &&
                       (w1 > 0 || (w1 == 0 && owns_edge(C, A))) && (w2 > 0 || (w2 == 0 && owns_edge(A, B)));
  unsigned long long key = SUMA_EMPTY_KEY;
  if (covered) {
    const float fa = (float)area;
#ifdef SUMA_BARY_PLAIN_DIV
    float b0 = edge_to_float(w0) / fa, b1 = edge_to_float(w1) / fa, b2 = edge_to_float(w2) / fa;
#else
    /* three quotients by one denominator; the operands are integers in [0, 2^46] over a positive integer area (X, Y
     * below 2^21, so every edge function is below 2^45 in magnitude), and a zero numerator gives an exact 0: the short
     * sequence of exact_div.h is correctly rounded there (tools/div_study.c) -- 18 instead of 33 instructions, the
     * same bits as the three `/` (the parity suite is the check) */
    const float fr = exdiv_refine(fa, __builtin_amdgcn_rcpf(fa));
    float b0 = exdiv_quot(edge_to_float(w0), fa, fr), b1 = exdiv_quot(edge_to_float(w1), fa, fr),
          b2 = exdiv_quot(edge_to_float(w2), fa, fr);
#endif
    float tu = (b0 * A.tu + b1 * B.tu) + b2 * C.tu;
    float tv = (b0 * A.tv + b1 * B.tv) + b2 * C.tv;
    float z = (b0 * A.z + b1 * B.z) + b2 * C.z;
    /* render_surfels.frag:22 disc test; near / far clipping */
    if (!((tu * tu + tv * tv) > 1.0f) && (z >= 0.0f && z <= 1.0f)) key = render_key(depth24(z), id, tie);
  }
  return key;
}

/* (inv_pose * surfelPose) * v, render_surfels.vert:44-48 */
__device__ __forceinline__ void surfel_to_sensor(const float* __restrict__ poses, const float* inv_pose, float count,
                                                 v3 pos, v3 nrm, v3* p, v3* n) {
  float Ps[16], M[16];
  const float4* src = reinterpret_cast<const float4*>(poses + 16 * (size_t)(int32_t)count);
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    float4 col = src[k];
    Ps[4 * k] = col.x;
    Ps[4 * k + 1] = col.y;
    Ps[4 * k + 2] = col.z;
    Ps[4 * k + 3] = col.w;
  }
  m4_mul(inv_pose, Ps, M);
  *p = m4_point(M, pos);
  *n = m4_dir(M, nrm);
}

/* K4 in three block-cooperative phases.
 *  1a (lane per surfel)   transform to the sensor frame, stability / age / back-face / image gating,
 *                         projection of the centre (and the fused K7 splat).  Survivors -- about
 *                         40 % of the lanes -- are COMPACTED into an LDS list.
 *  1b (lane per survivor) tangent frame, projection of the four quad corners, clipped bounding box.
 *                         Running this on dense lanes instead of under a 40 %-full exec mask removes
 *                         most of the kernel's ALU time (5 spherical projections per surfel).
 *  2  (lane per pixel test) the pixel tests of all survivors of the block are laid end to end
 *                         (prefix sum of the box areas) and handed out 256 at a time: at 64x2048 the
 *                         median surfel covers no pixel centre, the mean box is 3.6 tests, but the mean
 *                         per-wave maximum is 14 -- a lane-per-surfel raster loop ran 7x longer than
 *                         the work it contained. */
#ifdef SUMA_PHASE_TIMING
__device__ unsigned long long g_k4_phase[PH_BLOCKS][9];
/* host: PH_BLOCKS x 9 words (8 phase totals in 10 ns units + the number of launches the block took part in) */
extern "C" int suma_debug_k4_phases(unsigned long long* host, int reset) {
  hipError_t e = hipMemcpyFromSymbol(host, HIP_SYMBOL(g_k4_phase), sizeof(g_k4_phase));
  if (e == hipSuccess && reset) {
    void* p = nullptr;
    e = hipGetSymbolAddress(&p, HIP_SYMBOL(g_k4_phase));
    if (e == hipSuccess) e = hipMemset(p, 0, sizeof(g_k4_phase));
  }
  return (int)e;
}
#endif

/* Block barriers of k_render: everything the phases exchange lives in LDS (rank counters, candidate list, quad records,
 * prefix), so with K4_PREFETCH they are LDS-only barriers -- a __syncthreads() also drains the vector-memory counter
 * (s_waitcnt vmcnt(0)), i.e. it would wait for the next tile's loads at the first barrier behind their issue */
#if !defined(K4_NO_PREFETCH)
#define K4_PREFETCH 1 /* default since round 5; -DK4_NO_PREFETCH builds the round-4 form for A/B runs */
#endif
#ifdef K4_PREFETCH
#define K4_BARRIER() lds_barrier()
#else
#define K4_BARRIER() __syncthreads()
#endif
#define RENDER_THREADS 256
#define RENDER_WAVES (RENDER_THREADS / 64)
#define RENDER_BATCH 2
#define SUMA_RENDER_MAX_BLOCKS 65536u

/* exclusive rank of `flag` among the block's threads + block total (all threads call) */
__device__ __forceinline__ uint32_t render_block_rank(bool flag, uint32_t* s_w, uint32_t* total) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const unsigned long long ball = __ballot(flag);
  if (lane == 0) s_w[wave] = __popcll(ball);
  K4_BARRIER();
  uint32_t off = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < RENDER_WAVES; ++w) {
    uint32_t c = s_w[w];
    if (w < wave) off += c;
    tot += c;
  }
  *total = tot;
  return off + __popcll(ball & ((1ull << lane) - 1ull));
}

template <bool BIG> /* see the row / column split in phase 2 */
__global__ void __launch_bounds__(RENDER_THREADS) __attribute__((amdgpu_waves_per_eu(5, 8))) k_render(RenderArgs a) {
  __shared__ float s_cand[RENDER_THREADS][9];   /* p.xyz, n.xyz, radius, pp.x, surfel id (bits) */
  __shared__ int32_t s_rec[RENDER_THREADS][17]; /* X0..3, Y0..3, z0..3 (float bits), i0, j0, w, surfel id; 17: a row stride of 16 words puts all lanes of a write on two banks */
  __shared__ uint32_t s_incl[RENDER_THREADS];
  __shared__ uint32_t s_w[2][RENDER_WAVES], s_w2[RENDER_WAVES];
  __shared__ uint8_t s_mask[RENDER_THREADS]; /* slot mask of a candidate, by candidate rank */
  const uint32_t S = a.ds->n_surfels;
  const float4* __restrict__ sf = reinterpret_cast<const float4*>(a.surfels);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  /* Tiles are taken from the END of the surfel array first: the array is in creation order, so its tail holds
   * the surfels around the sensor's recent positions -- the ones in view, with pixel tests to run -- and its
   * head mostly surfels that leave after phase 1a.  Dispatching the expensive tiles first keeps the cheap
   * ones for the kernel's tail. */
  const uint32_t ntile = (S + RENDER_THREADS - 1) / RENDER_THREADS;
  const int npass = a.merged ? 1 : 2;
  uint32_t trip = 0; /* counts rank barriers: the per-wave counters alternate between two sets (see the early-out) */
  /* The three 16-byte loads of a lane's surfel are issued ONE TILE AHEAD (K4_PREFETCH): a block walks its tiles one after
   * the other and used to start every trip with a cold round trip to HBM (2.3 of 15.5 us per trip, tools/phase_timeline.py).
   * The record's registers are dead once the last phase 1a of a trip has consumed them, so the next tile's loads are
   * issued right there and fly under phases 1b / 2 of this tile: no extra VGPRs (96, five waves per SIMD as before).
   * Bit-identical by construction.  Measured (profiles/r05_k4_prefetch_experiment.txt): 50 M surfels 2.32 -> 2.11 ms
   * (0.172 -> 0.190 of the HBM roofline); at the steady 1 M map, where a block makes ~3 trips and other blocks fill
   * the gap anyway, 44.4 -> 43.7 us (inside the noise of the scan rate). */
  float4 s0 = f4(0, 0, 0, 0), s1 = s0, s2 = s0;
  /* unconditional loads from a clamped (always valid) address: a lane beyond the map, or a trip beyond the last tile,
   * re-reads the last record and never uses it (`live` and the K7 splat test i < S) -- no exec-masked branch, no merge
   * with zeros, so the destination registers ARE the loop-carried ones and nothing has to wait for them here */
  auto fetch_tile = [&](uint32_t t) {
    const uint32_t tt = t < ntile ? t : ntile - 1u;
    uint32_t j = (ntile - 1u - tt) * RENDER_THREADS + threadIdx.x;
    j = j < S ? j : S - 1u;
    s0 = sf[4 * (size_t)j];
    s1 = sf[4 * (size_t)j + 1];
    s2 = sf[4 * (size_t)j + 2];
  };
#ifdef K4_PREFETCH
  if (ntile) fetch_tile(blockIdx.x);
  /* the pass of a trip after which the record is dead (kernel-uniform) */
  const int last_pass = a.merged ? 0 : (a.slot[1].enabled ? 1 : 0);
#endif
  PH_BEGIN;
  for (uint32_t tile = blockIdx.x; tile < ntile; tile += gridDim.x) {
    PH(7); /* loop overhead / previous tile's tail */
    const uint32_t blk0 = (ntile - 1u - tile) * RENDER_THREADS;
    const uint32_t i = blk0 + threadIdx.x;
#ifndef K4_PREFETCH
    fetch_tile(tile);
#endif
    const bool live = (i < S) && !(a.use_stability && !(s1.w > a.conf_threshold));
    const float radius = s0.w, count = s2.w;
    const int32_t creation = (int32_t)count;
    const int32_t ts = (int32_t)__float_as_uint(s2.x);
#ifdef SUMA_PHASE_TIMING
    if (live && ts == 0x7fffffff) ph_acc[7] += 1; /* consumes the loads before the stamp */
#endif
    PH(0); /* surfel loads arrived */
    for (int sl = 0; sl < npass; ++sl) {
      /* merged: one trip, slot[1]'s pose (== slot[0]'s); otherwise one trip per enabled slot */
      const RenderSlot& slot = a.slot[a.merged ? 1 : sl];
      if (!a.merged && !slot.enabled) continue; /* kernel-uniform */
      const float* inv_pose = (sl == 0 && a.inv_pose_dev != nullptr) ? a.inv_pose_dev : slot.inv_pose.m;
      /* ---- phase 1a ---- */
      uint32_t selmask; /* bit b: the record renders into slot[b].zbuf */
      {
        const bool sel_old = live && (creation < a.thr), sel_new = live && (creation >= a.thr || ts >= a.thr);
        if (a.merged)
          selmask = ((sel_old && a.slot[0].enabled) ? 1u : 0u) | (sel_new ? 2u : 0u);
        else
          selmask = ((slot.mode == 0) ? sel_old : sel_new) ? (1u << sl) : 0u;
      }
      const bool selected = selmask != 0;
      const bool k7 = (sl == 0) && a.k7_enabled && (i < S);
      bool cand = false;
      unsigned long long k7_key = SUMA_EMPTY_KEY;
      uint32_t k7_pix = 0;
      v3 p = mk3(0, 0, 0), n = p;
      float ppx = 0.f;
      if (selected || k7) {
        surfel_to_sensor(a.poses, inv_pose, count, xyz(s0), xyz(s1), &p, &n);
        float lp = len3(p);
        if (dot3(n, divs3(neg3(p), lp)) > 0.01f) { /* front facing (render_surfels.geom:83, gen_indexmap.vert:68) */
          v3 pp = mk3(0, 0, 0);
          if (selected || a.k7_same_proj) pp = project01(a.q, p);
          if (k7) {
            /* K7 for every surfel (no stability / age gating): nearest visible surfel per data pixel */
            const v3 pr = a.k7_same_proj ? pp : project01(a.k7_q, p);
            float fx = sdm_floor(pr.x * a.k7_q.width), fy = sdm_floor(pr.y * a.k7_q.height);
            float zn = 2.0f * pr.z - 1.0f;
            if (fx >= 0.0f && fx < a.k7_q.width && fy >= 0.0f && fy < a.k7_q.height && zn >= -1.0f && zn <= 1.0f) {
              /* depth-tested write deferred to phase 2, where its memory round trip overlaps the raster's */
              k7_key = ((unsigned long long)depth24(0.5f * zn + 0.5f) << 32) | i;
              k7_pix = (uint32_t)(int32_t)fy * (uint32_t)a.k7_q.W + (uint32_t)(int32_t)fx;
            }
          }
          if (selected) {
            cand = (pp.x >= 0.0f && pp.y >= 0.0f && pp.z >= 0.0f && pp.x < 1.0f && pp.y < 1.0f && pp.z < 1.0f);
            ppx = pp.x;
          }
        }
      }
#ifdef K4_PREFETCH
      if (sl == last_pass) fetch_tile(tile + gridDim.x); /* radius / count / stamps of THIS tile were copied out above */
#endif
      uint32_t ncand;
      const uint32_t crank = render_block_rank(cand, s_w[trip & 1u], &ncand);
      trip += 1;
      PH(1); /* phase 1a + rank barrier */
      if (ncand == 0) {
        /* Nothing of this tile renders into this slot (block-uniform: every thread totals the same counters) --
         * the rule for the "old" slot of render() outside loop closures, and for tiles that are out of view: no
         * candidate list, no prefix, no closing barrier.  The LDS lists are untouched; the rank counters of the next
         * trip live in the other set, and that trip's barrier separates this trip's reads from the set's next use. */
        if (k7_key != SUMA_EMPTY_KEY) zbuf_min(&a.k7_zbuf[k7_pix], k7_key);
        continue;
      }
      if (cand) {
        float* r = s_cand[crank];
        r[0] = p.x;
        r[1] = p.y;
        r[2] = p.z;
        r[3] = n.x;
        r[4] = n.y;
        r[5] = n.z;
        r[6] = radius;
        r[7] = ppx;
        r[8] = __uint_as_float(i);
        s_mask[crank] = (uint8_t)selmask;
      }
      K4_BARRIER();
      PH(2); /* candidate list written + barrier */
      /* ---- phase 1b: dense lanes ---- */
      uint32_t ntests = 0;
      if (threadIdx.x < ncand) {
        const float* r = s_cand[threadIdx.x];
        const v3 cp = mk3(r[0], r[1], r[2]), cn = mk3(r[3], r[4], r[5]);
        const float crad = r[6], cppx = r[7];
        v3 u = normalize3(mk3(cn.y - cn.z, -cn.x, cn.x));
        v3 v = normalize3(cross3(cn, u));
        v3 ru = scale3(crad, u), rv = scale3(crad, v);
        v3 corner[4];
        corner[0] = sub3(sub3(cp, ru), rv);
        corner[1] = sub3(add3(cp, ru), rv);
        corner[2] = add3(sub3(cp, ru), rv);
        corner[3] = add3(add3(cp, ru), rv);
        /* The spherical projection of the corners is evaluated in two halves: first range + pitch
         * (image row), and the yaw (atan2, image column) only if the rows spanned by the quad contain
         * a pixel-centre row at all -- at 64 rows over 28 degrees about half of the quads do not.
         * Identical values to project01(), just not computed when they cannot matter. */
        int32_t X[4], Y[4];
        float Z[4];
        bool bad = false;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float depth = len3(corner[k]);
          const float pitch = -sdm_asin(corner[k].z / depth);
          const float y01 = 1.0f - ((pitch * SUMA_RAD2DEG_F) + a.q.fov_up) / a.q.fov;
          /* window depth as shader (gl_Position.z = 2 z01 - 1, render_surfels.geom:104-117) and viewport (z_w = 0.5 z_ndc +
           * 0.5) form it: in fp32 that is z01 again for only 84 % of the values */
          Z[k] = 0.5f * (2.0f * ((depth - a.q.min_depth) / (a.q.max_depth - a.q.min_depth)) - 1.0f) + 0.5f;
          const float yw = y01 * a.q.height;
          if (sdm_isnan(yw) || sdm_isnan(Z[k])) bad = true;
          Y[k] = (int32_t)sdm_floor(yw * 256.0f + 0.5f);
        }
        const int32_t minY = min(min(Y[0], Y[1]), min(Y[2], Y[3])), maxY = max(max(Y[0], Y[1]), max(Y[2], Y[3]));
        int32_t j0 = (minY - 128 + 255) >> 8, j1 = (maxY - 128) >> 8; /* pixel-centre rows inside the box */
        j0 = max(j0, 0);
        j1 = min(j1, a.q.H - 1);
        if (!bad && j0 <= j1) {
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const float yaw = sdm_atan2(corner[k].y, corner[k].x);
            float x01 = 0.5f * ((-yaw * SUMA_INV_PI_F) + 1.0f);
            /* render_surfels.geom:67-69: keep the quad on the centre's side of the yaw seam */
            if (cppx - x01 > 0.5f) x01 += 1.0f;
            if (x01 - cppx > 0.5f) x01 -= 1.0f;
            const float xw = x01 * a.q.width;
            if (sdm_isnan(xw)) bad = true;
            X[k] = (int32_t)sdm_floor(xw * 256.0f + 0.5f);
          }
          if (!bad) {
            const int32_t minX = min(min(X[0], X[1]), min(X[2], X[3])), maxX = max(max(X[0], X[1]), max(X[2], X[3]));
            int32_t i0 = (minX - 128 + 255) >> 8, i1 = (maxX - 128) >> 8; /* pixel-centre columns inside the box */
            i0 = max(i0, 0);
            i1 = min(i1, a.q.W - 1);
            if (i0 <= i1) {
              const int32_t w = i1 - i0 + 1;
              ntests = (uint32_t)w * (uint32_t)(j1 - j0 + 1);
              int32_t* q = s_rec[threadIdx.x];
#pragma unroll
              for (int k = 0; k < 4; ++k) {
                q[k] = X[k];
                q[4 + k] = Y[k];
                q[8 + k] = __float_as_int(Z[k]);
              }
              q[12] = i0;
              q[13] = j0;
              q[14] = w;
              q[15] = __float_as_int(r[8]);
              q[16] = (int32_t)s_mask[threadIdx.x];
            }
          }
        }
      }
      PH(3); /* phase 1b */
      /* inclusive prefix of the test counts over the block */
      uint32_t incl = wave_inclusive_scan(ntests);
      if (lane == 63) s_w2[wave] = incl;
      K4_BARRIER();
      uint32_t woff = 0, total = 0;
#pragma unroll
      for (int w = 0; w < RENDER_WAVES; ++w) {
        uint32_t c = s_w2[w];
        if (w < wave) woff += c;
        total += c;
      }
      s_incl[threadIdx.x] = incl + woff;
      K4_BARRIER();
      PH(4); /* prefix over the block, two barriers */
      /* ---- phase 2 ---- */
      /* RENDER_BATCH tests per lane and trip: the fragments' keys are computed first, then the (device-
       * coherent, i.e. memory-side) z-buffer reads of the batch are in flight together and the atomics follow
       * -- one memory round trip per 512 tests instead of two per 256 (a batch of 4 is 0.5 % slower: 96 VGPRs
       * and 8 bytes of scratch against 93 and none).  The two strip triangles of a quad
       * write the same pixel, so their keys are min-combined into a single depth-tested write. */
      /* The K7 read sits at the head of phase 2, where the compiler waits for it on the spot (it keeps the 1-bit result
       * of the comparison below instead of the 64-bit value).  Issuing it EARLIER -- behind phase 1a, so that it flies
       * under 1b and the prefix -- was built and measured in round 5: the fused pass 70.4 -> 82.0 us, -4.7 % scans/s
       * (profiles/r05_k4_prefetch_experiment.txt): a staler value lets more splats through to the atomic, and the
       * atomics of a hot pixel serialise at its memory channel.  The read belongs as close to the write as it is. */
      unsigned long long k7_cur = 0;
      if (k7_key != SUMA_EMPTY_KEY)
        k7_cur = __hip_atomic_load(&a.k7_zbuf[k7_pix], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      unsigned long long* const zb0 = a.slot[0].zbuf;
      unsigned long long* const zb1 = a.slot[1].zbuf;
      for (uint32_t t0 = threadIdx.x; t0 < total; t0 += RENDER_BATCH * RENDER_THREADS) {
        unsigned long long key[RENDER_BATCH];
        uint32_t pix[RENDER_BATCH], msk[RENDER_BATCH];
#pragma unroll
        for (int u = 0; u < RENDER_BATCH; ++u) {
          const uint32_t t = t0 + (uint32_t)u * RENDER_THREADS;
          key[u] = SUMA_EMPTY_KEY;
          pix[u] = 0;
          msk[u] = 0;
          if (t < total) {
            /* source record: the first one whose inclusive prefix exceeds t */
            int lo = 0, hi = RENDER_THREADS - 1;
#pragma unroll
            for (int it = 0; it < 8; ++it) {
              int mid = (lo + hi) >> 1;
              if (s_incl[mid] > t)
                hi = mid;
              else
                lo = mid + 1;
            }
            const int src = lo;
            const uint32_t excl = src ? s_incl[src - 1] : 0u;
            const int32_t* r = s_rec[src];
            const uint32_t q = t - excl, w = (uint32_t)r[14];
            /* row / column of test q in a box w pixels wide.  An unsigned 32-bit division is ~25 instructions;
             * floor(q / w) = floor((q + 0.5) / w), and that quotient lies at least 0.5 / w away from every integer, which a
             * product with the hardware reciprocal (1 ulp; error of the product below 1.5 * 2^-23 of its value) cannot
             * bridge while q < 2^21 -- so the truncated product IS the row.  q is below the pixel count of the model
             * image; BIG (an image of 2 M pixels or more: none of the suites comes near) is the instantiation that divides. */
            const uint32_t qj = BIG ? q / w : (uint32_t)(((float)q + 0.5f) * __builtin_amdgcn_rcpf((float)w));
            const uint32_t qi = q - __umul24(qj, w); /* both below 2^24 */
            const int32_t pi = r[12] + (int32_t)qi, pj = r[13] + (int32_t)qj;
            rvtx vt[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              vt[k].X = r[k];
              vt[k].Y = r[4 + k];
              vt[k].z = __int_as_float(r[8 + k]);
              vt[k].tu = (k & 1) ? 1.0f : -1.0f;
              vt[k].tv = (k & 2) ? 1.0f : -1.0f;
            }
            const uint32_t id = (uint32_t)r[15];
            /* triangle strip: (v0,v1,v2), (v2,v1,v3) */
            const unsigned long long ka = raster_key(vt[0], vt[1], vt[2], pi, pj, id, slot.tie);
            const unsigned long long kb = raster_key(vt[2], vt[1], vt[3], pi, pj, id, slot.tie);
            key[u] = ka < kb ? ka : kb;
            pix[u] = __umul24((uint32_t)pj, (uint32_t)a.q.W) + (uint32_t)pi; /* row, width < 2^24 */
            msk[u] = (uint32_t)r[16];
          }
        }
        /* a z-buffer a fragment does not go to reads as 0: no key is smaller, no atomic follows */
        unsigned long long cur0[RENDER_BATCH], cur1[RENDER_BATCH];
#pragma unroll
        for (int u = 0; u < RENDER_BATCH; ++u) {
          cur0[u] = cur1[u] = 0;
          if (key[u] != SUMA_EMPTY_KEY) {
            if (msk[u] & 1u) cur0[u] = __hip_atomic_load(&zb0[pix[u]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (msk[u] & 2u) cur1[u] = __hip_atomic_load(&zb1[pix[u]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
        }
#pragma unroll
        for (int u = 0; u < RENDER_BATCH; ++u) {
          if (key[u] < cur0[u]) atomicMin(&zb0[pix[u]], key[u]);
          if (key[u] < cur1[u]) atomicMin(&zb1[pix[u]], key[u]);
        }
      }
      if (k7_key < k7_cur) atomicMin(&a.k7_zbuf[k7_pix], k7_key);
      PH(5); /* phase 2: pixel tests, z-buffer reads, atomics */
      K4_BARRIER(); /* the LDS lists are reused by the next slot / iteration */
      PH(6); /* closing barrier */
    }
  }
  PH_END(g_k4_phase);
}

__device__ __forceinline__ uint32_t key_id(unsigned long long key, int tie) {
  uint32_t low = (uint32_t)(key & 0xffffffffull);
  return tie == TIE_LOW_INDEX ? low : (0xffffffffu - low);
}

struct ResolveOut {
  float4 v, n, s;
};

/* attributes of the winning surfel (render_surfels.frag:30-32): surfel-centre position, normal,
 * semantic -- flat over the disc */
__device__ __forceinline__ ResolveOut resolve_pixel(unsigned long long key, int tie, const suma_surfel* surfels,
                                                    const float* poses, const float* inv_a, const float* inv_b) {
  ResolveOut o;
  if (key == SUMA_EMPTY_KEY) {
    o.v = o.n = o.s = f4(0.f, 0.f, 0.f, 0.f);
    return o;
  }
  uint32_t id = key_id(key, tie);
  const float4* sf = reinterpret_cast<const float4*>(surfels) + 4 * (size_t)id;
  float4 s0 = sf[0], s1 = sf[1], s2 = sf[2], s3 = sf[3];
  const float* inv = inv_a;
  if (tie != TIE_LOW_INDEX && ((key >> 32) & 1ull) == 0) inv = inv_b; /* composed: drawn by the new pass */
  v3 p, n;
  surfel_to_sensor(poses, inv, s2.w, xyz(s0), xyz(s1), &p, &n);
  o.v = f4(p.x, p.y, p.z, 1.0f);
  o.n = f4(n.x, n.y, n.z, 1.0f);
  o.s = s3;
  return o;
}

struct ResolveArgs {
  const suma_surfel* surfels;
  const float* poses;
  unsigned long long *zbuf_a, *zbuf_b;
  m4 inv_a, inv_b;
  int tie;
  uint32_t Pm;
  float max_distance;
  float4 *va, *na, *sa; /* targets of zbuf_a (old frame / single target); NULL = not written */
  float4 *vb, *nb, *sb; /* targets of zbuf_b (new frame) */
  float4 *vo, *no, *so; /* composed output (K5) */
  /* optional mirror of target a (Frame::copy fused into the resolve): vertex / normal as written,
   * semantic copied from sa_src (the semantic map is not re-rendered by render_active) */
  float4 *vm, *nm, *sm;
  const float4* sm_src;
  const float* inv_dev; /* if set: inverse pose from HBM instead of inv_a / inv_b */
};

/* single z-buffer -> up to three maps */
__global__ void __launch_bounds__(256) k_resolve(ResolveArgs a) {
  uint32_t pix = blockIdx.x * blockDim.x + threadIdx.x;
  if (pix >= a.Pm) return;
  unsigned long long key = a.zbuf_a[pix];
  a.zbuf_a[pix] = SUMA_EMPTY_KEY;
  const float* ia = a.inv_dev ? a.inv_dev : a.inv_a.m;
  const float* ib = a.inv_dev ? a.inv_dev : a.inv_b.m;
  ResolveOut o = resolve_pixel(key, a.tie, a.surfels, a.poses, ia, ib);
  if (a.va) a.va[pix] = o.v;
  if (a.na) a.na[pix] = o.n;
  if (a.sa) a.sa[pix] = o.s;
  if (a.vm) {
    a.vm[pix] = o.v;
    a.nm[pix] = o.n;
    a.sm[pix] = a.sm_src[pix];
  }
}

/* old + new z-buffers -> old frame, new frame and the K5 composition (render_compose.frag:26-48) */
__global__ void __launch_bounds__(256) k_resolve_compose(ResolveArgs a) {
  uint32_t pix = blockIdx.x * blockDim.x + threadIdx.x;
  if (pix >= a.Pm) return;
  unsigned long long ka = a.zbuf_a[pix], kb = a.zbuf_b[pix];
  a.zbuf_a[pix] = SUMA_EMPTY_KEY;
  a.zbuf_b[pix] = SUMA_EMPTY_KEY;
  ResolveOut o = resolve_pixel(ka, TIE_LOW_INDEX, a.surfels, a.poses, a.inv_a.m, a.inv_a.m);
  ResolveOut nw = resolve_pixel(kb, TIE_LOW_INDEX, a.surfels, a.poses, a.inv_b.m, a.inv_b.m);
  a.va[pix] = o.v;
  a.na[pix] = o.n;
  a.sa[pix] = o.s;
  a.vb[pix] = nw.v;
  a.nb[pix] = nw.n;
  a.sb[pix] = nw.s;
  float4 v = nw.v, n = nw.n, s = nw.s;
  bool valid = (o.v.w > 0.5f && o.n.w > 0.5f);
  bool new_valid = (v.w > 0.5f && n.w > 0.5f);
  if (!new_valid && valid && (v.w < 0.5f || len3(sub3(xyz(v), xyz(o.v))) < a.max_distance)) {
    v = o.v;
    n = o.n;
    s = o.s;
  }
  a.vo[pix] = v;
  a.no[pix] = n;
  a.so[pix] = s;
}

static void set_m4(m4& d, const float* s) {
  for (int i = 0; i < 16; ++i) d.m[i] = s[i];
}

/* the instantiation for this model image (see the row / column split in phase 2) */
static void run_render(suma_ctx* c, const RenderArgs& a, uint32_t grid) {
  /* decided on the image THIS launch renders into (a.q), which is what bounds q in phase 2; the rasteriser's other
   * range precondition -- window coordinates |X| < 2^21 in 1/256 pixel, x01 in [-0.5, 1.5] -- is a limit on the model
   * width that suma_ctx_create / suma_set_params enforce (SUMA_MAX_MODEL_WIDTH) */
  if ((size_t)a.q.W * (size_t)a.q.H >= ((size_t)1 << 21))
    k_render<true><<<grid, RENDER_THREADS, 0, c->ls>>>(a);
  else
    k_render<false><<<grid, RENDER_THREADS, 0, c->ls>>>(a);
}
static RenderArgs render_args(suma_ctx* c, float conf_threshold, int32_t thr) {
  RenderArgs a;
  a.surfels = c->surfels[c->cur];
  a.ds = c->ds;
  a.poses = c->poses;
  a.q = c->pm;
  a.conf_threshold = conf_threshold;
  a.use_stability = c->p.use_stability;
  a.thr = thr;
  a.slot[0].enabled = a.slot[1].enabled = 0;
  a.slot[0].zbuf = a.slot[1].zbuf = c->zbuf_a; /* valid addresses even for a slot that stays off */
  a.slot[0].tie = a.slot[1].tie = TIE_LOW_INDEX;
  a.merged = 0;
  a.inv_pose_dev = nullptr;
  a.k7_enabled = 0;
  a.k7_q = c->pd;
  a.k7_same_proj = (memcmp(&c->pd, &c->pm, sizeof(proj_t)) == 0) ? 1 : 0;
  a.k7_zbuf = c->zbuf_data;
  return a;
}
static uint32_t stream_grid(suma_ctx* c) {
  /* sized from the last surfel count the host has seen; the kernels grid-stride over the
   * device-resident count, so a stale value only changes the number of loop trips */
  uint64_t est = (uint64_t)c->known_surfels + 2 * c->P;
  uint64_t blocks = (est + 255) / 256;
  /* one tile per block while that stays a sane grid: the per-tile cost varies several-fold (pixel tests),
   * and the hardware dispatcher balances single-tile blocks for free; very large maps fall back to
   * grid-striding */
  if (blocks > SUMA_RENDER_MAX_BLOCKS) blocks = SUMA_RENDER_MAX_BLOCKS;
  if (blocks < 256) blocks = 256;
  return (uint32_t)blocks;
}
static ResolveArgs resolve_args(suma_ctx* c) {
  ResolveArgs r;
  memset(&r, 0, sizeof(r));
  r.surfels = c->surfels[c->cur];
  r.poses = c->poses;
  r.zbuf_a = c->zbuf_a;
  r.zbuf_b = c->zbuf_b;
  r.Pm = (uint32_t)c->Pm;
  r.max_distance = c->p.max_loop_closure_distance;
  return r;
}

/* SurfelMap::render(pose_old, pose_new, frame, ct), SurfelMap.cpp:847-1021 */
hipError_t launch_map_render(suma_ctx* c, const float* pose_old, const float* pose_new, float conf_threshold,
                             suma_frame* out) {
  float inv_old[16], inv_new[16];
  rigid_inverse_f(pose_old, inv_old);
  rigid_inverse_f(pose_new, inv_new);
  const double S = (double)c->known_surfels;
  const uint32_t Pm = (uint32_t)c->Pm;
  if (c->p.compose_rendering) {
    int32_t thr = (int32_t)(c->timestamp - 100u); /* SurfelMap.cpp:873, quirk B-7 */
    RenderArgs a = render_args(c, conf_threshold, thr);
    /* "old" surfels have creation stamp < thr: none can exist while thr <= 0 (the first 100 scans,
     * quirk B-7), so the pass is skipped; its z-buffer stays empty and resolves to the cleared frame */
    a.slot[0].enabled = thr > 0 ? 1 : 0;
    a.slot[0].mode = 0;
    a.slot[0].tie = TIE_LOW_INDEX;
    a.slot[0].zbuf = c->zbuf_a;
    set_m4(a.slot[0].inv_pose, inv_old);
    a.slot[1].enabled = 1;
    a.slot[1].mode = 1;
    a.slot[1].tie = TIE_LOW_INDEX;
    a.slot[1].zbuf = c->zbuf_b;
    set_m4(a.slot[1].inv_pose, inv_new);
    /* outside loop closures currentPose_old_ == currentPose_new_ (SurfelMapping.cpp:457-458): one trip per tile */
    a.merged = (memcmp(inv_old, inv_new, sizeof(inv_old)) == 0 && !getenv("SUMA_RENDER_NO_MERGE")) ? 1 : 0;
    {
      ProfScope ps(c, "k4_render_surfels", 64.0 * S);
      run_render(c, a, stream_grid(c));
    }
    ResolveArgs r = resolve_args(c);
    set_m4(r.inv_a, inv_old);
    set_m4(r.inv_b, inv_new);
    r.va = c->old_frame->map[0];
    r.na = c->old_frame->map[1];
    r.sa = c->old_frame->map[2];
    r.vb = c->new_frame->map[0];
    r.nb = c->new_frame->map[1];
    r.sb = c->new_frame->map[2];
    r.vo = out->map[0];
    r.no = out->map[1];
    r.so = out->map[2];
    {
      ProfScope ps(c, "k5_resolve_compose", (16.0 + 144.0) * Pm);
      k_resolve_compose<<<(Pm + 255) / 256, 256, 0, c->ls>>>(r);
    }
  } else {
    /* SurfelMap.cpp:976-1018: one pass, threshold 0, then two frame copies */
    RenderArgs a = render_args(c, conf_threshold, 0);
    a.slot[0].enabled = 1;
    a.slot[0].mode = 1;
    a.slot[0].tie = TIE_LOW_INDEX;
    a.slot[0].zbuf = c->zbuf_a;
    set_m4(a.slot[0].inv_pose, inv_old);
    {
      ProfScope ps(c, "k4_render_surfels", 64.0 * S);
      run_render(c, a, stream_grid(c));
    }
    ResolveArgs r = resolve_args(c);
    set_m4(r.inv_a, inv_old);
    set_m4(r.inv_b, inv_old);
    r.tie = TIE_LOW_INDEX;
    r.va = out->map[0];
    r.na = out->map[1];
    r.sa = out->map[2];
    {
      ProfScope ps(c, "k5_resolve", (8.0 + 48.0) * Pm);
      k_resolve<<<(Pm + 255) / 256, 256, 0, c->ls>>>(r);
    }
    size_t bytes = c->Pm * sizeof(float4);
    for (int m = 0; m < 3; ++m) {
      hipMemcpyAsync(c->new_frame->map[m], out->map[m], bytes, hipMemcpyDeviceToDevice, c->ls);
      hipMemcpyAsync(c->old_frame->map[m], out->map[m], bytes, hipMemcpyDeviceToDevice, c->ls);
    }
  }
  return hipGetLastError();
}

/* render_active (which = 1, SurfelMap.cpp:1023-1069) / render_inactive (which = 0, :1071-1114).
 * Only COLOR0 / COLOR1 are re-attached (:1047-1048): the semantic map of the target frame is NOT
 * refreshed by these calls; restated as such. */
hipError_t launch_map_render_single(suma_ctx* c, const float* pose, float conf_threshold, int active, int fuse_k7,
                                    suma_frame* mirror) {
  /* pose == NULL: the pose block written by the closing Gauss-Newton launch (HBM) is used */
  float inv[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
  if (pose) rigid_inverse_f(pose, inv);
  int32_t thr = (int32_t)(c->timestamp - 100u);
  RenderArgs a = render_args(c, conf_threshold, thr);
  if (!pose) a.inv_pose_dev = c->pose_block + 16;
  a.slot[0].enabled = 1;
  a.slot[0].mode = active ? 1 : 0;
  a.slot[0].tie = TIE_LOW_INDEX;
  a.slot[0].zbuf = c->zbuf_a;
  set_m4(a.slot[0].inv_pose, inv);
  a.k7_enabled = fuse_k7;
  {
    ProfScope ps(c, fuse_k7 ? "k4k7_render_indexmap" : "k4_render_surfels", 64.0 * (double)c->known_surfels);
    run_render(c, a, stream_grid(c));
  }
  ResolveArgs r = resolve_args(c);
  set_m4(r.inv_a, inv);
  set_m4(r.inv_b, inv);
  r.tie = TIE_LOW_INDEX;
  if (!pose) r.inv_dev = c->pose_block + 16;
  suma_frame* tgt = active ? c->new_frame : c->old_frame;
  r.va = tgt->map[0];
  r.na = tgt->map[1];
  r.sa = nullptr;
  if (mirror) { /* lastModelFrame_->copy(*map_->newMapFrame()), SurfelMapping.cpp:407, without a second pass */
    r.vm = mirror->map[0];
    r.nm = mirror->map[1];
    r.sm = mirror->map[2];
    r.sm_src = tgt->map[2];
  }
  {
    ProfScope ps(c, "k5_resolve", (8.0 + 32.0 + (mirror ? 64.0 : 0.0)) * (double)c->Pm);
    k_resolve<<<((uint32_t)c->Pm + 255) / 256, 256, 0, c->ls>>>(r);
  }
  return hipGetLastError();
}

/* render_composed, SurfelMap.cpp:1116-1165: old pass then new pass, GL_LEQUAL, one depth
 * buffer, only the vertex / normal attachments are switched to composedFrame_ */
hipError_t launch_map_render_composed(suma_ctx* c, const float* pose_old, const float* pose_new,
                                      float conf_threshold) {
  float inv_old[16], inv_new[16];
  rigid_inverse_f(pose_old, inv_old);
  rigid_inverse_f(pose_new, inv_new);
  int32_t thr = (int32_t)(c->timestamp - 100u);
  RenderArgs a = render_args(c, conf_threshold, thr);
  a.slot[0].enabled = 1;
  a.slot[0].mode = 0;
  a.slot[0].tie = TIE_HIGH_INDEX_OLD;
  a.slot[0].zbuf = c->zbuf_a;
  set_m4(a.slot[0].inv_pose, inv_old);
  a.slot[1].enabled = 1;
  a.slot[1].mode = 1;
  a.slot[1].tie = TIE_HIGH_INDEX_NEW;
  a.slot[1].zbuf = c->zbuf_a;
  set_m4(a.slot[1].inv_pose, inv_new);
  {
    ProfScope ps(c, "k4_render_surfels", 64.0 * (double)c->known_surfels);
    run_render(c, a, stream_grid(c));
  }
  ResolveArgs r = resolve_args(c);
  set_m4(r.inv_a, inv_old);
  set_m4(r.inv_b, inv_new);
  r.tie = TIE_HIGH_INDEX_OLD;
  r.va = c->composed_frame->map[0];
  r.na = c->composed_frame->map[1];
  r.sa = nullptr;
  {
    ProfScope ps(c, "k5_resolve", (8.0 + 32.0) * (double)c->Pm);
    k_resolve<<<((uint32_t)c->Pm + 255) / 256, 256, 0, c->ls>>>(r);
  }
  return hipGetLastError();
}
