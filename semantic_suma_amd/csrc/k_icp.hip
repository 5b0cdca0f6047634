/*
 * k_icp.hip -- K6 + the Gauss-Newton loop, device resident.
 *
 * Replaces (reference):
 *   Frame2Model::jacobianProducts   src/core/Frame2Model.cpp:136-261
 *     + src/shader/Frame2Model_jacobians.geom:53-247 (association, gating, Huber/Tukey and
 *       semantic weights, J^T W J / J^T W r / statistics)
 *   LieGaussNewton::minimize / step src/core/LieGaussNewton.cpp:13-79
 *   Objective::increment            src/core/Objective.h:45-48
 *   SE3::exp                        src/core/lie_algebra.cpp:4-34
 *
 * GL structure replaced: the reference draws ceil(W/64)*H geometry-shader threads that each
 * loop over 64 pixels and blend-add 16 RGB32F points into a 2x8 target (undefined summation
 * order), then glFinish + 192-byte readback + host LDLT per iteration.  Here:
 *   - one lane per data pixel (grid-stride), 3 coalesced float4 loads of the data maps, 12
 *     float4 bilinear taps of the model maps (L2 resident: 6 maps = 12.6 MB at 64x2048);
 *   - every fp32 term is converted to 2^-28 fixed point (round to nearest even, via the
 *     1.5*2^52 double trick) and summed as int64: exact and order independent, so the result is
 *     bit-identical for any reduction tree (and to the CPU oracle);
 *   - per-lane int64 accumulators -> wave butterfly (lane-swap stages on v_permlane32/16_swap, the row-local
 *     stages as DPP moves) -> LDS across the 8 waves -> the block's 32 sums are ADDED into one of ICP_RECORDS (8) rotating
 *     accumulator records per hypothesis with memory-side 64-bit atomics (exact integers: order immaterial);
 *   - no in-kernel hand-off: the records of launch j are consumed by the PROLOGUE of launch j+1 (kernel
 *     boundary = visibility), where every block redundantly totals them (2 KB), solves the 6x6 system by LDL^T
 *     in fp64, applies exp(delta) to the pose and evaluates the stopping tests while its own data-pixel loads
 *     are already in flight.  Three record sets rotate (read / add / zero for the next launch); the state is
 *     double buffered by launch parity; block 0 writes it.  The next iteration is just the next launch: no host
 *     round trip.  The closing launch (k_icp_finish, one block) only consumes and reports into a pinned host
 *     record; the objective-only statistics pass closes itself (last block totals and reports).
 */
#include <cstddef>

#include "suma_internal.h"

/* -DSUMA_GN_TIMING (tools/gn_timeline.py builds such a library next to the product one): every block stamps
 * wall_clock64 (100 MHz) at the stations of a launch into g_gn_timing[blockIdx.x][station] */
#ifdef SUMA_GN_TIMING
__device__ unsigned long long g_gn_timing[256][8];
#define GN_STAMP(k)                                                                            \
  do {                                                                                         \
    if (threadIdx.x == 0 && blockIdx.y == 0 && blockIdx.x < 256) g_gn_timing[blockIdx.x][k] = wall_clock64(); \
  } while (0)
extern "C" int suma_debug_gn_timing(unsigned long long* host /* 256 x 8 */) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_gn_timing), sizeof(g_gn_timing));
}
#else
#define GN_STAMP(k) ((void)0)
#endif

#define ICP_THREADS 512 /* 8 waves per block, one block per CU: 131072 lanes = one 64x2048 image in flight */
#define MAGIC_D 6755399441055744.0          /* 1.5 * 2^52 */
#define MAGIC_BITS 0x4338000000000000ll     /* its bit pattern */

struct IcpArgs {
  const float4 *Vd, *Nd, *Sd; /* data frame, exact texels */
  const float4 *Vm, *Nm, *Sm; /* model frame, bilinear / nearest */
  int32_t W, H, Wm, Hm;
  float fov_up, fov; /* of the data image (Frame2Model.cpp:82-99) */
  float angle_thresh, distance_thresh, factor;
  int32_t weight_function, bilinear;
  uint32_t P;
  /* statistics pass of the scan pipeline only: K8's per-pixel work for the frame that is being streamed
   * anyway (dev_math.h, k8_pixel) and the per-update counter resets ride along */
  int k8_enabled;
  K8Out k8;
  DevState* k8_ds;
};

__device__ __forceinline__ float4 bilinear_fetch(const float4* __restrict__ map, int32_t w, int32_t h, float x,
                                                 float y) {
  /* GL_LINEAR + CLAMP_TO_BORDER on a rectangle texture (GL 3.3 spec 3.8.11): texel centres at
   * integer + 0.5, border (0,0,0,0); all four channels filtered (quirk B-5) */
  float u = x - 0.5f, v = y - 0.5f;
  float fu = sdm_floor(u), fv = sdm_floor(v);
  float a = u - fu, b = v - fv;
  int32_t i0 = (int32_t)fu, j0 = (int32_t)fv;
  float4 t00 = texel(map, w, h, i0, j0);
  float4 t10 = texel(map, w, h, i0 + 1, j0);
  float4 t01 = texel(map, w, h, i0, j0 + 1);
  float4 t11 = texel(map, w, h, i0 + 1, j0 + 1);
  float w00 = (1.0f - a) * (1.0f - b), w10 = a * (1.0f - b), w01 = (1.0f - a) * b, w11 = a * b;
  float4 r;
  r.x = ((w00 * t00.x + w10 * t10.x) + w01 * t01.x) + w11 * t11.x;
  r.y = ((w00 * t00.y + w10 * t10.y) + w01 * t01.y) + w11 * t11.y;
  r.z = ((w00 * t00.z + w10 * t10.z) + w01 * t01.z) + w11 * t11.z;
  r.w = ((w00 * t00.w + w10 * t10.w) + w01 * t01.w) + w11 * t11.w;
  return r;
}

/* raw magic-number bits of round(term * 2^28); MAGIC_BITS is subtracted once per term after the
 * reduction (count * MAGIC_BITS, modulo 2^64) */
__device__ __forceinline__ long long fix_bits(float term, double scale, double magic) {
  /* one v_fma_f64: the product with 2^28 is exact, so fusing changes nothing numerically (this
   * file is otherwise compiled with -ffp-contract=off, which would split it into ldexp + add).
   * scale = 2^28 and magic = 1.5 * 2^52 are passed in REGISTERS (see fix_consts): with literal operands the
   * compiler emits a destructive v_fmac_f64 and re-materialises the magic number with two v_mov per term. */
  double d = __builtin_fma((double)term, scale, magic);
  return __double_as_longlong(d);
}
/* lane-trips of one pixel phase: the wave that starts at pixel 64 q runs a trip for every q with 64 q < P, all of its 64
 * lanes taking part (a lane beyond the image hands in zeros) -- whatever the grid is */
__device__ __forceinline__ unsigned long long lane_trips(uint32_t P) { return 64ull * ((P + 63u) / 64u); }
__device__ __forceinline__ void fix_consts(double* scale, double* magic) {
  double sc = SUMA_ACC_SCALE, mg = MAGIC_D;
  asm volatile("" : "+v"(sc), "+v"(mg)); /* opaque: keeps both in VGPR pairs */
  *scale = sc;
  *magic = mg;
}

__device__ __forceinline__ long long shfl_xor_ll(long long v, int mask) {
  int lo = __shfl_xor((int)(v & 0xffffffffll), mask, 64);
  int hi = __shfl_xor((int)(v >> 32), mask, 64);
  return ((long long)hi << 32) | (unsigned int)lo;
}

/* lane ^ MASK exchanges inside a row of 16 lanes as DPP moves (VALU rate; __shfl_xor lowers to ds_bpermute_b32, an LDS
 * crossbar round trip of ~60 cycles, and the butterfly below is a dependent chain of them):
 *   ^1, ^2  quad_perm [1,0,3,2] / [2,3,0,1];  ^8  row_ror:8;
 *   ^4      lanes 0-3 / 8-11 of a row take from lane + 4 (row_shl:4), the others from lane - 4 (row_shr:4) */
template <int MASK>
__device__ __forceinline__ int dpp_xor(int v) {
  static_assert(MASK == 1 || MASK == 2 || MASK == 4 || MASK == 8, "row-local exchanges only");
  if (MASK == 1) return __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, true);
  if (MASK == 2) return __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, true);
  if (MASK == 8) return __builtin_amdgcn_update_dpp(0, v, 0x128, 0xF, 0xF, true);
  /* both directions with bound_ctrl (a lane without an enabled source reads 0, as in the other three forms), then
   * the lane picks its side */
  const int from_up = __builtin_amdgcn_update_dpp(0, v, 0x104, 0xF, 0xF, true);   /* row_shl:4: lane + 4 */
  const int from_down = __builtin_amdgcn_update_dpp(0, v, 0x114, 0xF, 0xF, true); /* row_shr:4: lane - 4 */
  return (threadIdx.x & 4u) ? from_down : from_up;
}
template <int MASK>
__device__ __forceinline__ long long dpp_xor_ll(long long v) {
  const int lo = dpp_xor<MASK>((int)(v & 0xffffffffll));
  const int hi = dpp_xor<MASK>((int)(v >> 32));
  return ((long long)hi << 32) | (unsigned int)lo;
}

/* Butterfly reduction of 32 per-lane words over a 64-lane wave: at each stage a lane keeps half
 * of its words and trades the other half with its partner, so 32 -> 1 word per lane costs
 * 16+8+4+2+1 exchanges plus one final pairwise add.  Afterwards lane L holds the wave total of
 * word ((L>>5)&1)*16 + ((L>>4)&1)*8 + ((L>>3)&1)*4 + ((L>>2)&1)*2 + ((L>>1)&1). */
template <int H, int MASK>
__device__ __forceinline__ void reduce_stage(long long (&a)[SUMA_ACC_WORDS], int lane) {
  const bool upper = (lane & MASK) != 0;
#pragma unroll
  for (int i = 0; i < H; ++i) {
    long long keep = upper ? a[i + H] : a[i];
    long long send = upper ? a[i] : a[i + H];
    a[i] = keep + shfl_xor_ll(send, MASK);
  }
}
/* Stages 32 and 16 of the butterfly with gfx950's lane-swap instructions: v_permlane32_swap
 * exchanges the upper half of one register with the lower half of another (v_permlane16_swap: odd
 * rows with even rows), which is exactly "lower lanes keep word i and receive the partner's word
 * i, upper lanes keep word i+H and receive the partner's word i+H" -- the stage becomes
 * swap(lo), swap(hi), 64-bit add: no LDS crossbar, no selects. */
template <int H, bool SWAP32>
__device__ __forceinline__ void reduce_stage_swap(long long (&a)[SUMA_ACC_WORDS]) {
#pragma unroll
  for (int i = 0; i < H; ++i) {
    unsigned int x_lo = (unsigned int)a[i], x_hi = (unsigned int)(a[i] >> 32);
    unsigned int y_lo = (unsigned int)a[i + H], y_hi = (unsigned int)(a[i + H] >> 32);
    auto lo = SWAP32 ? __builtin_amdgcn_permlane32_swap(x_lo, y_lo, false, false)
                     : __builtin_amdgcn_permlane16_swap(x_lo, y_lo, false, false);
    auto hi = SWAP32 ? __builtin_amdgcn_permlane32_swap(x_hi, y_hi, false, false)
                     : __builtin_amdgcn_permlane16_swap(x_hi, y_hi, false, false);
    long long p = (long long)(((unsigned long long)hi[0] << 32) | lo[0]);
    long long q = (long long)(((unsigned long long)hi[1] << 32) | lo[1]);
    a[i] = p + q;
  }
}
__device__ __forceinline__ long long wave_reduce32(long long (&a)[SUMA_ACC_WORDS], int lane) {
  /* explicit stages: every index is a compile-time constant, the words stay in VGPRs */
  reduce_stage_swap<16, true>(a);
  reduce_stage_swap<8, false>(a);
  reduce_stage<4, 8>(a, lane);
  reduce_stage<2, 4>(a, lane);
  reduce_stage<1, 2>(a, lane);
  return a[0] + shfl_xor_ll(a[0], 1);
}
__device__ __forceinline__ int word_of_lane(int lane) {
  return ((lane >> 5) & 1) * 16 + ((lane >> 4) & 1) * 8 + ((lane >> 3) & 1) * 4 + ((lane >> 2) & 1) * 2 +
         ((lane >> 1) & 1);
}

/* The same butterfly for 16 words: halving stages 8 / 4 (lane swaps), 2 / 1 (xor 8, xor 4), then the two lane bits
 * that select no word are summed out.  Afterwards lane L holds the wave total of word
 * ((L>>5)&1)*8 + ((L>>4)&1)*4 + ((L>>3)&1)*2 + ((L>>2)&1)  (every word on four lanes). */
template <int H, bool SWAP32>
__device__ __forceinline__ void reduce16_stage_swap(long long (&a)[16]) {
#pragma unroll
  for (int i = 0; i < H; ++i) {
    unsigned int x_lo = (unsigned int)a[i], x_hi = (unsigned int)(a[i] >> 32);
    unsigned int y_lo = (unsigned int)a[i + H], y_hi = (unsigned int)(a[i + H] >> 32);
    auto lo = SWAP32 ? __builtin_amdgcn_permlane32_swap(x_lo, y_lo, false, false)
                     : __builtin_amdgcn_permlane16_swap(x_lo, y_lo, false, false);
    auto hi = SWAP32 ? __builtin_amdgcn_permlane32_swap(x_hi, y_hi, false, false)
                     : __builtin_amdgcn_permlane16_swap(x_hi, y_hi, false, false);
    long long p = (long long)(((unsigned long long)hi[0] << 32) | lo[0]);
    long long q = (long long)(((unsigned long long)hi[1] << 32) | lo[1]);
    a[i] = p + q;
  }
}
template <int H, int MASK>
__device__ __forceinline__ void reduce16_stage(long long (&a)[16], int lane) {
  const bool upper = (lane & MASK) != 0;
#pragma unroll
  for (int i = 0; i < H; ++i) {
    long long keep = upper ? a[i + H] : a[i];
    long long send = upper ? a[i] : a[i + H];
    a[i] = keep + dpp_xor_ll<MASK>(send);
  }
}
__device__ __forceinline__ long long wave_reduce16(long long (&a)[16], int lane) {
  reduce16_stage_swap<8, true>(a);
  reduce16_stage_swap<4, false>(a);
  reduce16_stage<2, 8>(a, lane);
  reduce16_stage<1, 4>(a, lane);
  long long t = a[0] + dpp_xor_ll<2>(a[0]);
  return t + dpp_xor_ll<1>(t);
}
__device__ __forceinline__ int word16_of_lane(int lane) {
  return ((lane >> 5) & 1) * 8 + ((lane >> 4) & 1) * 4 + ((lane >> 3) & 1) * 2 + ((lane >> 2) & 1);
}

__device__ __forceinline__ long long readlane_ll(long long v, int src) {
  const unsigned int lo = (unsigned int)__builtin_amdgcn_readlane((int)(unsigned int)v, src);
  const unsigned int hi = (unsigned int)__builtin_amdgcn_readlane((int)(unsigned int)((unsigned long long)v >> 32), src);
  return (long long)(((unsigned long long)hi << 32) | lo);
}

__device__ __forceinline__ double bcast_d(double v, int src) { /* value of lane `src`, wave-uniform */
  const long long b = __double_as_longlong(v);
  const unsigned int lo = (unsigned int)__builtin_amdgcn_readlane((int)(unsigned int)b, src);
  const unsigned int hi = (unsigned int)__builtin_amdgcn_readlane((int)(unsigned int)((unsigned long long)b >> 32), src);
  return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

/* JtJ.ldlt().solve(-Jtf), LieGaussNewton.cpp:60: unpivoted LDL^T in fp64 with the fixed operation order of
 * the CPU restatement this library is tested against, the six rows of L spread over lanes 0..5 of a wave (every lane of
 * the wave runs this; lanes >= 6 shadow lane 5).  Each element goes through exactly the serial algorithm's
 * operations, in the same order -- what changes is that the five quotients of a column (and the six of
 * y / D) are formed side by side: a one-lane version spends two thirds of its instructions in 21 dependent
 * fp64 divisions, this one issues 6 (fp64 instructions of the launch: 1025 -> 539).
 * val = the 21 packed upper-triangle entries of JtJ followed by Jtr (LDS); x comes back wave-uniform. */
__device__ __forceinline__ void solve6_wave(const double* val, int lane, double* x) {
  const int li = lane < 6 ? lane : 5;
  double Acol[6], Lrow[6], D[6];
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    const int lo = li < j ? li : j, hi = li < j ? j : li;
    Acol[j] = val[lo * 6 - (lo * (lo - 1)) / 2 + (hi - lo)]; /* JtJ(li, j), symmetric */
    Lrow[j] = 0.0;
  }
  double Dmine = 1.0;
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    double sacc = Acol[j];
#pragma unroll
    for (int k = 0; k < j; ++k) sacc -= (Lrow[k] * bcast_d(Lrow[k], j)) * D[k];
    D[j] = bcast_d(sacc, j); /* on lane j the sum above is d_j: (L[k][j] * L[k][j]) * D[k] */
    if (li == j) Dmine = sacc;
    Lrow[j] = sacc / D[j]; /* rows li > j; 1 on lane j, unused elsewhere */
  }
  double y = -val[21 + li];
#pragma unroll
  for (int k = 0; k < 5; ++k) {
    const double yk = bcast_d(y, k); /* final on lane k: it only ever subtracts terms k' < k */
    if (li > k) y -= Lrow[k] * yk;
  }
  y = y / Dmine;
#pragma unroll
  for (int i = 5; i >= 0; --i) {
    double sacc = bcast_d(y, i);
#pragma unroll
    for (int k = i + 1; k < 6; ++k) sacc -= bcast_d(Lrow[i], k) * x[k]; /* L[i][k] lives in row k */
    x[i] = sacc;
  }
}

/* SE3::exp, lie_algebra.cpp:4-34; x = (v, omega); column-major */
__device__ __forceinline__ void se3_exp(const double* x, double* T) {
  _Pragma("unroll") for (int i = 0; i < 16; ++i) T[i] = (i % 5 == 0) ? 1.0 : 0.0;
  const double v[3] = {x[0], x[1], x[2]}, o[3] = {x[3], x[4], x[5]};
  double theta = sdm_sqrt_d((o[0] * o[0] + o[1] * o[1]) + o[2] * o[2]);
  if (theta > 1e-10) {
    double K[9] = {0, -o[2], o[1], o[2], 0, -o[0], -o[1], o[0], 0};
    double K2[9];
    _Pragma("unroll") for (int r = 0; r < 3; ++r)
      _Pragma("unroll") for (int cc = 0; cc < 3; ++cc)
        K2[3 * r + cc] = (K[3 * r] * K[cc] + K[3 * r + 1] * K[3 + cc]) + K[3 * r + 2] * K[6 + cc];
    /* lie_algebra.cpp evaluates sin / cos twice each; same arguments, same values */
    const double st = sdm_sin_d(theta), ct = sdm_cos_d(theta);
    double alpha = st / theta;
    double beta = (1 - ct) / (theta * theta);
    double gamma = beta; /* (1.0 - cos(theta)) / (theta * theta) */
    double delta = (theta - st) / (theta * theta * theta);
    _Pragma("unroll") for (int r = 0; r < 3; ++r) {
      double t = 0.0;
      _Pragma("unroll") for (int cc = 0; cc < 3; ++cc) {
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

__device__ __forceinline__ void mul4d(const double* A, const double* B, double* C) {
  _Pragma("unroll") for (int c = 0; c < 4; ++c)
    _Pragma("unroll") for (int r = 0; r < 4; ++r)
      C[4 * c + r] =
          ((A[r] * B[4 * c] + A[4 + r] * B[4 * c + 1]) + A[8 + r] * B[4 * c + 2]) + A[12 + r] * B[4 * c + 3];
}

struct PoseD {
  double m[16];
};

__device__ __forceinline__ void gn_reset(GnState* g, int t, uint32_t iteration0) {
  if (t < SUMA_ACC_WORDS) g->acc[t] = 0;
  if (t == 0) {
    g->last_error = (double)3.402823466e+38f; /* LieGaussNewton.cpp:48 */
    g->F = g->F_inlier = 0.0;
    g->iteration = iteration0; /* Frame2Model::setData / initialize resets it to 0, Frame2Model.cpp:122 */
    g->k = 0;
    g->done = 0;
    g->converged = 0;
    g->valid = g->outlier = g->invalid = 0;
    g->n_hist = 1;
    g->pending = 0;
  }
}

/* LieGaussNewton::initialize (LieGaussNewton.cpp:36-51): one block per hypothesis */
__global__ void k_gn_init(GnState* gn, const double* T0s, double* history, uint32_t iteration0, uint32_t iteration0_rest) {
  GnState* g = gn + blockIdx.x;
  int t = threadIdx.x;
  if (t < 16) {
    g->Tk[t] = T0s[16 * blockIdx.x + t];
    if (history != nullptr && blockIdx.x == 0) history[t] = T0s[t];
  }
  gn_reset(g, t, blockIdx.x == 0 ? iteration0 : iteration0_rest);
}
#define ICP_RECORDS 8 /* accumulator records per hypothesis (power of two) */
struct IterArgs {
  IcpArgs a;
  const GnState* gin;  /* state written by the previous launch */
  GnState* gout;       /* state this launch writes */
  /* Block sums of a pixel phase are ADDED (memory-side 64-bit atomics; the sums are exact integers, so
   * the order is immaterial) into ICP_RECORDS accumulator records per hypothesis -- the next launch
   * totals 8 records instead of one per block.  Three buffers rotate: this launch reads pin (written by
   * the previous launch), adds into pout (zeroed by the previous launch) and zeroes pzero (read by the
   * previous launch) for the next one. */
  const long long* pin;
  long long* pout;
  long long* pzero;
  uint32_t zero_hyp; /* hypotheses whose records in pzero may be non-zero */
  uint32_t nblocks;    /* blocks of a pixel phase */
  uint32_t max_iter;
  double epsilon, delta_thr;
  int eval_only; /* Frame2Model::jacobianProducts only: no solve, no pose update */
  int pixel;     /* 0: consume-only launch (grid.x = 1) */
  double* history;
  uint32_t history_cap;
  /* first launch of a chain: LieGaussNewton::initialize (LieGaussNewton.cpp:36-51) is folded in --
   * the start state comes from the kernel arguments instead of gin */
  int init;
  uint32_t iteration0;
  PoseD T0;
  /* closing launch of the pipeline's minimisation: also emit the sensor pose the re-rendering
   * (SurfelMapping.cpp:406) uses, pose_base * increment as float + its rigid inverse, so that the
   * render pass can be enqueued without waiting for the host to read the increment back */
  int emit_pose;
  PoseD pose_base;
  float* pose_block; /* 16 floats pose, 16 floats inverse */
  /* closing launch: report to the host directly (pinned memory), see HostResult */
  HostResult* host_out;
  uint32_t host_seq;
  int host_full; /* class-by-class entries: the report also carries acc / JtJ / Jtr / n_hist (HostResult) */
  const DevState* ds;
  /* eval-only pixel launch that reports by itself (last block), see the end of icp_iter_body */
  HostResult* fused_report;
  uint32_t* fused_counter;
};

/* One launch of the Gauss-Newton chain.  grid = (nblocks or 1, n_hyp).
 *   prologue: if the previous launch left partial sums (pending), total them, and -- unless this is
 *             a jacobianProducts-only call -- solve, update the pose, run the stopping tests
 *             (LieGaussNewton::step, LieGaussNewton.cpp:53-79).  Done by every block on its own
 *             copy (identical inputs, identical code => identical results); block 0 records it.
 *   body:     unless finished, the K6 pixel phase at the current pose -> one partial per block. */
/* The two 4 x 4 double matrices of IterArgs (T0: the start pose of a fresh chain; pose_base: the sensor pose the closing
 * launch multiplies the increment onto) are 64 SGPRs that almost no launch needs: held for the whole kernel they pushed
 * the one-wave consume step of k_icp_finish over the 102 SGPRs of a wave -- 47 spilled, 118 v_readlane in a serial
 * chain (tools/isa_stats.py).  They are read from the kernel-argument segment at their one use instead (scalar loads;
 * the empty asm keeps them there).  karg_off = offset of the IterArgs parameter in the segment. */
typedef const IterArgs __attribute__((address_space(4))) * IterArgsK;
__device__ __forceinline__ IterArgsK gn_args_again(uint32_t karg_off) {
  const char __attribute__((address_space(4)))* p =
      (const char __attribute__((address_space(4)))*)__builtin_amdgcn_kernarg_segment_ptr() + karg_off;
  asm volatile("" : "+s"(p));
  return (IterArgsK)p;
}

template <bool PIXEL>
__device__ __forceinline__ void icp_iter_body(const IterArgs& g, const uint32_t karg_off) {
  const IcpArgs& a = g.a;
  const GnState* __restrict__ gin = g.gin + blockIdx.y;
  GnState* __restrict__ gout = g.gout + blockIdx.y;
  const bool writer = (blockIdx.x == 0 && threadIdx.x == 0);

  __shared__ long long s_wave[ICP_THREADS / 64][SUMA_ACC_WORDS];
  __shared__ long long s_tot[ICP_THREADS / 32][SUMA_ACC_WORDS];
  __shared__ double s_pose[16], s_E[16], s_val[SUMA_ACC_WORDS];
  __shared__ uint32_t s_flag[4]; /* done, iteration, history slot, last-block flag */

  if (blockIdx.x == 0 && threadIdx.x < ICP_RECORDS * SUMA_ACC_WORDS) /* for the next launch */
    for (uint32_t h = blockIdx.y; h < g.zero_hyp; h += gridDim.y)
      g.pzero[(size_t)h * ICP_RECORDS * SUMA_ACC_WORDS + threadIdx.x] = 0;

  GN_STAMP(0); /* block started */
  /* Everything this block needs from memory before it can compute is requested up front and TOGETHER -- the
   * previous launch's accumulator records (wave 0: four words per lane), the pose columns of lanes 0..15, this
   * lane's data-frame texels -- so that the prologue pays one cold round trip, not a chain of them: none of these
   * addresses depends on the state words read below.  (A launch that finds its chain finished has loaded in vain;
   * it exits right after.) */
  const long long* __restrict__ pin = g.pin + (size_t)blockIdx.y * ICP_RECORDS * SUMA_ACC_WORDS;
  long long rec[ICP_RECORDS / 2] = {0, 0, 0, 0};
  if (threadIdx.x < 64 && !g.init) {
    /* lane L of wave 0: word L & 31 of records (L >> 5) * 4 .. + 3 */
#pragma unroll
    for (int q = 0; q < ICP_RECORDS / 2; ++q) rec[q] = pin[((threadIdx.x >> 5) * (ICP_RECORDS / 2) + q) * SUMA_ACC_WORDS + (threadIdx.x & 31)];
  }
  /* lanes 0..15 form one element each of exp(delta) * pose_ at the end of the prologue: their column of the
   * current pose (and their own element, for launches that do not move the pose) comes straight from HBM */
  double tk_col[4] = {0.0, 0.0, 0.0, 0.0}, tk_self = 0.0;
  if (threadIdx.x < 16 && !g.init) {
#pragma unroll
    for (int q = 0; q < 4; ++q) tk_col[q] = gin->Tk[4 * (threadIdx.x >> 2) + q];
    tk_self = gin->Tk[threadIdx.x];
  }
  /* data-frame loads of this lane's first pixel do not depend on the pose */
  const uint32_t pix0 = blockIdx.x * ICP_THREADS + threadIdx.x;
  float4 vd4 = f4(0, 0, 0, 0), nd4 = vd4, sd4 = vd4;
  if (PIXEL && pix0 < a.P) {
    vd4 = a.Vd[pix0];
    nd4 = a.Nd[pix0];
    sd4 = a.Sd[pix0];
  }

  /* wave-uniform state (scalar loads), or the start state of a fresh chain */
  const uint32_t done_in = g.init ? 0u : gin->done, pending = g.init ? 0u : gin->pending;
  uint32_t iteration = g.init ? g.iteration0 : gin->iteration;
  double Tk[16];
  if (g.init) {
    const IterArgsK gk = gn_args_again(karg_off);
#pragma unroll
    for (int i = 0; i < 16; ++i) Tk[i] = gk->T0.m[i];
  } else {
#pragma unroll
    for (int i = 0; i < 16; ++i) Tk[i] = gin->Tk[i];
  }
  const bool want_px = PIXEL && !(done_in && !pending);

  float prefetch_sink = 0.0f;
  uint32_t done = done_in;
  if (pending) {
    if (threadIdx.x < 64) {
      /* ---- wave 0 totals the previous launch's accumulator records in registers (no LDS round trip, no block
       *      barrier: everything up to the pose broadcast below stays inside this wave, whose LDS traffic is in
       *      order), removes the fixed-point bias and converts: word w of the totals and its value in double land
       *      in LDS for the solve ---- */
      long long sum = (rec[0] + rec[1]) + (rec[2] + rec[3]);
      GN_STAMP(1); /* state words + records have arrived */
      sum += shfl_xor_ll(sum, 32);
      const int w = threadIdx.x & 31;
      /* every lane-trip of the pixel phase has added the magic number to each of the 29 fixed-point words (terms that
       * do not contribute are formed from a zero weight, not selected away: see the pixel phase) */
      if (w < 29) sum = (long long)((unsigned long long)sum - (unsigned long long)lane_trips(a.P) * (unsigned long long)MAGIC_BITS);
      if (threadIdx.x < SUMA_ACC_WORDS) {
        s_wave[0][w] = sum;
        s_val[w] = (double)sum * (1.0 / SUMA_ACC_SCALE);
      }
    }
    if (PIXEL && threadIdx.x >= 64 && want_px && pix0 < a.P && (vd4.w + nd4.w) > 1.5f) {
      /* The seven waves that now only wait for lane 0's solve warm the caches for their own pixel:
       * the model texels move by a fraction of a texel per iteration, so touching the lines the
       * PREVIOUS pose projects to turns most of the pixel phase's dependent gather into hits.
       * Pure prefetch -- nothing computed here reaches a result. */
      float To[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) To[i] = (float)Tk[i];
      const v3 q = m4_point(To, xyz(vd4));
      const float depth = len3(q);
      const float yaw = sdm_atan2(q.y, q.x);
      const float pitch = -sdm_asin(q.z / depth);
      const float ix = (0.5f * ((-yaw * SUMA_INV_PI_F) + 1.0f)) * (float)a.Wm - 0.5f;
      const float iy = (1.0f - ((pitch * SUMA_RAD2DEG_F) + a.fov_up) / a.fov) * (float)a.Hm - 0.5f;
      if (ix >= 0.0f && ix < (float)(a.Wm - 1) && iy >= 0.0f && iy < (float)(a.Hm - 1)) {
        const size_t t00 = (size_t)(int32_t)iy * (size_t)a.Wm + (size_t)(int32_t)ix;
        const float* pv = reinterpret_cast<const float*>(a.Vm);
        const float* pn = reinterpret_cast<const float*>(a.Nm);
        const float* ps = reinterpret_cast<const float*>(a.Sm);
        prefetch_sink = (pv[4 * t00] + pv[4 * (t00 + a.Wm) + 4]) + (pn[4 * t00] + pn[4 * (t00 + a.Wm) + 4]) +
                        (ps[4 * t00] + ps[4 * (t00 + a.Wm) + 4]);
      }
    }
    if (threadIdx.x < 64) {
      /* ---- LieGaussNewton::step on wave 0: every lane runs it on wave-uniform values (same cost as one
       *      lane), the 6x6 solve spreads its rows over lanes 0..5, lane 0 writes ---- */
      const int lane = threadIdx.x;
      /* s_wave / s_val were written by OTHER lanes of this wave: a wavefront-scope fence + wave barrier pins the
       * order for the compiler (the hardware executes one wave's LDS traffic in order; no instruction is emitted) */
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
      const double err = s_val[27];
      if (writer) {
        gout->F = err;
        gout->F_inlier = s_val[28];
        gout->valid = (uint32_t)s_wave[0][29];
        gout->outlier = (uint32_t)s_wave[0][30];
        gout->invalid = (uint32_t)s_wave[0][31];
        if (g.eval_only) /* Objective::jacobianProducts: the raw fixed-point words */
          for (int w = 0; w < SUMA_ACC_WORDS; ++w) gout->acc[w] = s_wave[0][w];
      }
      if (blockIdx.x == 0 && lane < 42) {
        /* JtJ / Jtf of this step, one element per lane: jacobianProducts' outputs and LieGaussNewton::information_
         * (LieGaussNewton.cpp:75); the symmetric matrix is mirrored from the packed upper triangle */
        const bool to_host = !PIXEL && g.host_full && g.host_out != nullptr && blockIdx.y == 0;
        if (lane < 36) {
          const int i = lane % 6, j = lane / 6, lo = i < j ? i : j, hi = i < j ? j : i;
          const double v = s_val[lo * 6 - (lo * (lo - 1)) / 2 + (hi - lo)];
          gout->JtJ[lane] = v;
          if (to_host) g.host_out->JtJ[lane] = v;
        } else {
          const double v = s_val[21 + (lane - 36)];
          gout->Jtr[lane - 36] = v;
          if (to_host) g.host_out->Jtr[lane - 36] = v;
        }
      }
      uint32_t k = gin->k, n_hist = gin->n_hist, converged = gin->converged, hist_slot = 0xffffffffu;
      double last_error = gin->last_error;
      if (!g.eval_only) {
        double dx[6];
        GN_STAMP(2); /* folded, converted, about to solve */
        solve6_wave(s_val, lane, dx);
        GN_STAMP(3); /* solved */
        int result = 1;
        double linf = 0.0, maxc = s_val[21];
#pragma unroll
        for (int i = 0; i < 6; ++i) {
          const double ad = dx[i] < 0 ? -dx[i] : dx[i], ji = s_val[21 + i];
          if (ad > linf) linf = ad;
          if (ji > maxc) maxc = ji;
        }
        if (linf < g.delta_thr) result = 0;                                   /* LieGaussNewton.cpp:64 */
        if ((maxc < 0 ? -maxc : maxc) < g.epsilon) result = 0;                /* :65 (quirk B-4) */
        double de = err - last_error;
        if (err < last_error && (de < 0 ? -de : de) < g.epsilon) result = 0;  /* :66 */
        double E[16];
        se3_exp(dx, E);
        GN_STAMP(7); /* exp(delta) formed */
        if (lane == 0) {
#pragma unroll
          for (int i = 0; i < 16; ++i) s_E[i] = E[i];
        }
        iteration += 1;
        last_error = err;
        if (result == 0) {
          converged = 1;
          done = 1;
        } else {
          k += 1;
          if (g.history != nullptr && blockIdx.y == 0 && n_hist < g.history_cap) hist_slot = n_hist;
          n_hist += 1;
          if (k >= g.max_iter) done = 1;
        }
      }
      if (lane == 0) {
        s_flag[0] = done;
        s_flag[1] = iteration;
      }
      if (writer) {
        gout->last_error = last_error;
        gout->iteration = iteration;
        gout->k = k;
        gout->n_hist = n_hist;
        gout->converged = converged;
        gout->done = done;
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); /* lane 0's s_E stores before the other lanes' loads */
      __builtin_amdgcn_wave_barrier();
      if (lane < 16) {
        /* pose_ = SE3::exp(delta) * pose_ -- applied even when converged (Objective.h:46): one element per
         * lane, the expression of mul4d().  LDS traffic of one wave is in order: lane 0's s_E stores above
         * are visible to these loads without a barrier.  (Round 6 tried the row picked out of the wave-uniform
         * registers by selects instead of this LDS round trip: pose ready 4.20 -> 4.27 us, not kept.) */
        double t = tk_self;
        if (!g.eval_only) {
          const int r = lane & 3;
          t = ((s_E[r] * tk_col[0] + s_E[4 + r] * tk_col[1]) + s_E[8 + r] * tk_col[2]) + s_E[12 + r] * tk_col[3];
          if (blockIdx.x == 0 && hist_slot != 0xffffffffu) g.history[16 * (size_t)hist_slot + lane] = t;
        }
        s_pose[lane] = t;
        if (blockIdx.x == 0) gout->Tk[lane] = t;
      }
    }
    GN_STAMP(4); /* wave 0: exp + pose product + state stores issued */
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 16; ++i) Tk[i] = s_pose[i];
    done = s_flag[0];
    iteration = s_flag[1];
  } else if (writer && g.init) {
    for (int i = 0; i < 16; ++i) gout->Tk[i] = Tk[i];
    if (g.history != nullptr && blockIdx.y == 0)
      for (int i = 0; i < 16; ++i) g.history[i] = Tk[i];
    gout->last_error = (double)3.402823466e+38f; /* LieGaussNewton.cpp:48 */
    gout->F = gout->F_inlier = 0.0;
    gout->iteration = iteration; /* Frame2Model::setData / initialize resets it, Frame2Model.cpp:122 */
    gout->k = 0;
    gout->n_hist = 1;
    gout->converged = 0;
    gout->done = 0;
    gout->valid = gout->outlier = gout->invalid = 0;
  } else if (blockIdx.x == 0 && threadIdx.x < 64 && !g.init) {
    /* Nothing to consume: carry the state over to the other buffer (the buffers alternate with the launch parity).
     * Round 6: the WHOLE record, one 8-byte word per lane and trip, all loads in flight together.  Round 5 had thread 0
     * copy it field by field -- ~90 dependent cold round trips, 11.5 us: a launch that found its chain CONVERGED took
     * longer than one that worked (9.2 us), and a minimisation in the reference's own mode (stopping tests on, 33
     * iterations at most: LieGaussNewton.cpp:23-33, default.xml:16) enqueues two dozen of them
     * (profiles/r06_gn_done_launch.txt).  The word that holds `pending` is left to thread 0, which stores that field on
     * every path below; acc / JtJ / Jtr of a chain that is not done are don't-cares (the next consume step rewrites
     * them) and travel along. */
    static_assert(sizeof(GnState) % 8 == 0 && offsetof(GnState, pending) % 8 == 0, "GnState is carried as 8-byte words");
    const unsigned long long* __restrict__ src = reinterpret_cast<const unsigned long long*>(gin);
    unsigned long long* __restrict__ dst = reinterpret_cast<unsigned long long*>(gout);
#pragma unroll
    for (int w0 = 0; w0 < (int)(sizeof(GnState) / 8); w0 += 64) {
      const int w = w0 + (int)threadIdx.x;
      if (w < (int)(sizeof(GnState) / 8) && w != (int)(offsetof(GnState, pending) / 8)) dst[w] = src[w];
    }
    if (done_in && !PIXEL && g.host_full && g.host_out != nullptr && blockIdx.y == 0 && threadIdx.x < 42) {
      /* information() of a chain that has converged */
      if (threadIdx.x < 36)
        g.host_out->JtJ[threadIdx.x] = gin->JtJ[threadIdx.x];
      else
        g.host_out->Jtr[threadIdx.x - 36] = gin->Jtr[threadIdx.x - 36];
    }
  }
  /* the record the closing report below reads this launch's statistics from: thread 0's own stores where it consumed
   * or initialised, the previous launch's record where the state was only carried (other lanes wrote the copy) */
  const GnState* __restrict__ rs = (pending || g.init) ? gout : gin;

  if (!PIXEL || (done && !g.eval_only) || (g.eval_only && pending)) {
    if (writer) {
      gout->pending = 0;
      if (!PIXEL && g.emit_pose && blockIdx.y == 0) {
        double Pd[16];
        double base[16];
        const IterArgsK gk = gn_args_again(karg_off);
        for (int i = 0; i < 16; ++i) base[i] = gk->pose_base.m[i];
        mul4d(base, Tk, Pd); /* currentPose_new_ * increment, in double as on the host */
        float Pf[16], Pinv[16];
        for (int i = 0; i < 16; ++i) Pf[i] = (float)Pd[i];
        rigid_inverse_dev(Pf, Pinv);
        for (int i = 0; i < 16; ++i) {
          g.pose_block[i] = Pf[i];
          g.pose_block[16 + i] = Pinv[i];
        }
      }
      if (!PIXEL && g.host_out != nullptr && blockIdx.y == 0) {
        HostResult* __restrict__ h = g.host_out;
        for (int i = 0; i < 16; ++i) h->Tk[i] = Tk[i];
        h->F = rs->F; /* own earlier stores, or the carried record's source */
        h->F_inlier = rs->F_inlier;
        h->valid = rs->valid;
        h->outlier = rs->outlier;
        h->invalid = rs->invalid;
        h->k = rs->k;
        h->converged = rs->converged;
        h->iteration = rs->iteration;
        h->n_hist = rs->n_hist;
        /* read here, behind the solve: requested up front with the prologue's loads the 64 bytes made the launch
         * 1.4 us LONGER (rocprofv3: 9.33 against 7.90 us; profiles/r04_late_experiments.txt) */
        h->ds = *g.ds;
        __threadfence_system();
        __hip_atomic_store(&h->seq, g.host_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
      }
    }
    return;
  }

  /* ---- K6 pixel phase at the current pose ---- */
  /* (Round 6 handed the pose to the waiting waves as 16 floats through a second LDS array: k_icp_step -1 % for one chain,
   * but 129 instead of 123 VGPRs -- over the 128 that let two 512-thread blocks share a CU, and batched chains, which
   * fill the chip, lost 19 % (BASELINE configs[2]: 1420 -> 1196 scans/s).  Not kept: profiles/r06_gn_pixel_phase.txt.) */
  float T[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) T[i] = (float)Tk[i]; /* pose_.cast<float>(), Frame2Model.cpp:194 */

  /* Per trip the 32 terms of a pixel are formed and wave-reduced in two halves of 16 words: a lane never holds more
   * than 16 int64 terms (32 VGPRs instead of 64), and what it carries from trip to trip is the wave total of ONE word
   * per half.  The sums are exact integers, so reducing per trip instead of once changes no bit. */
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  long long totA = 0, totB = 0; /* wave totals of word word16_of_lane(lane) and 16 + word16_of_lane(lane) */
  double fx_scale, fx_magic;
  fix_consts(&fx_scale, &fx_magic);

  const float fWm = (float)a.Wm, fHm = (float)a.Hm;
  /* The trip count is WAVE-uniform (the wave's first pixel decides): wave_reduce16 exchanges words between all 64
   * lanes, so a lane whose pixel lies beyond the image (P not a multiple of 64) must stay in the loop and hand in
   * zeros instead of leaving it. */
  for (uint32_t pix = pix0; pix - (threadIdx.x & 63u) < a.P; pix += gridDim.x * ICP_THREADS) {
    const bool valid_px = pix < a.P; /* this lane's trip processes a pixel of the image */
    if (pix != pix0) {
      vd4 = nd4 = sd4 = f4(0, 0, 0, 0);
      if (valid_px) {
        vd4 = a.Vd[pix];
        nd4 = a.Nd[pix];
        sd4 = a.Sd[pix];
      }
    }
    if (a.k8_enabled && valid_px) { /* k8_enabled is launch-uniform */
      k8_pixel(a.k8, pix, vd4, nd4, sd4);
      if (pix == 0) {
        a.k8_ds->n_updated = 0;
        a.k8_ds->n_data = 0;
        a.k8_ds->n_kept_updated = 0;
        a.k8_ds->n_kept_data = 0;
      }
    }
    float e_d = vd4.w + nd4.w;
    bool pair = false;
    float4 vm4, nm4, sm4;
    v3 v_d, n_d;
    if (e_d > 1.5f) {
      v_d = m4_point(T, xyz(vd4));
      n_d = m4_dir(T, xyz(nd4));
      /* project2model, Frame2Model_jacobians.geom:53-65 */
      float depth = len3(v_d);
      float yaw = sdm_atan2(v_d.y, v_d.x);
      float pitch = -sdm_asin(v_d.z / depth);
      float ix = (0.5f * ((-yaw * SUMA_INV_PI_F) + 1.0f)) * fWm;
      float iy = (1.0f - ((pitch * SUMA_RAD2DEG_F) + a.fov_up) / a.fov) * fHm;
      bool in_image = (ix >= 0.0f && ix < fWm && iy >= 0.0f && iy < fHm); /* false for NaN */
      if (in_image) {
        /* all 12 taps are issued together (the semantic taps are only needed for valid pairs, but
         * fetching them speculatively removes a dependent L2 round trip from every lane) */
        if (a.bilinear) {
          vm4 = bilinear_fetch(a.Vm, a.Wm, a.Hm, ix, iy);
          nm4 = bilinear_fetch(a.Nm, a.Wm, a.Hm, ix, iy);
          sm4 = bilinear_fetch(a.Sm, a.Wm, a.Hm, ix, iy);
        } else {
          int32_t tx = (int32_t)sdm_floor(ix), ty = (int32_t)sdm_floor(iy);
          vm4 = texel(a.Vm, a.Wm, a.Hm, tx, ty);
          nm4 = texel(a.Nm, a.Wm, a.Hm, tx, ty);
          sm4 = texel(a.Sm, a.Wm, a.Hm, tx, ty);
        }
        float e_m = vm4.w + nm4.w;
        pair = e_m > 1.5f;
      }
    }
    float J[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, wgt = 0.f, wr = 0.f, wr2 = 0.f;
    bool is_inlier = false;
    if (pair) {
      v3 v_m = xyz(vm4), n_m = xyz(nm4);
      bool inlier = true;
      if (len3(sub3(v_m, v_d)) > a.distance_thresh) inlier = false;
      if (dot3(n_m, n_d) < a.angle_thresh) inlier = false;
      float residual = dot3(n_m, sub3(v_d, v_m));
      v3 cp = cross3(v_d, n_m);
      float weight = 1.0f;
      if (a.weight_function == 4 || a.weight_function == 1) { /* Huber, .geom:120-128 */
        if (sdm_abs(residual) > a.factor) weight = a.factor / sdm_abs(residual);
      } else if (a.weight_function == 2 && iteration > 0) { /* Tukey, .geom:129-141 */
        if (sdm_abs(residual) > a.factor) {
          weight = 0.0f;
        } else {
          float alpha = residual / a.factor;
          weight = (1.0f - alpha * alpha);
          weight = weight * weight;
        }
      }
      /* semantic weighting, .geom:143-158 */
      float data_label = sd4.x * 255.0f, data_prob = sd4.w;
      float model_label = sm4.x * 255.0f;
      if (is_dynamic_label(model_label)) {
        if (sdm_round(data_label) != sdm_round(model_label))
          weight *= (1.0f - data_prob);
        else
          weight *= data_prob;
      }
      wr2 = (weight * residual) * residual;
      wr = weight * residual;
      is_inlier = inlier;
      J[0] = n_m.x;
      J[1] = n_m.y;
      J[2] = n_m.z;
      J[3] = cp.x;
      J[4] = cp.y;
      J[5] = cp.z;
      wgt = weight;
    }
    /* Round 6: a term that does not contribute is formed from a ZERO WEIGHT instead of being selected away behind its
     * conversion -- fix_bits(0) is the magic number itself, so every lane-trip adds exactly one magic number to each of
     * the 29 fixed-point words and the consume step removes lane_trips(P) of them (round 5: 29 64-bit selects per pixel
     * and a bias of n_inlier / n_valid magic numbers).  The sums are the same integers; ~55 VALU instructions per
     * pixel fewer in a phase that is issue bound at two waves per SIMD.  J is finite for every pair (both texels valid),
     * so 0 * J is a zero. */
    const bool in = pair && is_inlier;
    const float wgt_in = in ? wgt : 0.0f, wr_in = in ? wr : 0.0f, wr2_in = in ? wr2 : 0.0f;
    /* words 0..15: the first 16 entries of the upper triangle of J^T W J (row-major: (0,0)..(0,5), (1,1)..(1,5),
     * (2,2)..(2,5), (3,3)) */
    {
      long long h[16];
      int k = 0;
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        const float wJi = wgt_in * J[i];
#pragma unroll
        for (int j = i; j < 6; ++j) {
          if (k < 16) h[k] = fix_bits(wJi * J[j], fx_scale, fx_magic);
          ++k;
        }
      }
      totA += wave_reduce16(h, lane);
    }
    /* words 16..31: the last five entries of the triangle ((3,4) (3,5) (4,4) (4,5) (5,5)), J^T W r, F, F over the
     * inliers, the three counters */
    {
      long long h[16];
      h[0] = fix_bits((wgt_in * J[3]) * J[4], fx_scale, fx_magic);
      h[1] = fix_bits((wgt_in * J[3]) * J[5], fx_scale, fx_magic);
      h[2] = fix_bits((wgt_in * J[4]) * J[4], fx_scale, fx_magic);
      h[3] = fix_bits((wgt_in * J[4]) * J[5], fx_scale, fx_magic);
      h[4] = fix_bits((wgt_in * J[5]) * J[5], fx_scale, fx_magic);
#pragma unroll
      for (int i = 0; i < 6; ++i) h[5 + i] = fix_bits(wr_in * J[i], fx_scale, fx_magic);
      h[11] = fix_bits(wr2, fx_scale, fx_magic);    /* word 27: F over all pairs (wr2 is 0 without a pair) */
      h[12] = fix_bits(wr2_in, fx_scale, fx_magic); /* word 28: F over the inliers */
      h[13] = pair ? 1ll : 0ll;                     /* word 29: valid */
      h[14] = (pair && !is_inlier) ? 1ll : 0ll;     /* word 30: outlier */
      h[15] = (valid_px && !pair) ? 1ll : 0ll;      /* word 31: invalid */
      totB += wave_reduce16(h, lane);
    }
  }

  GN_STAMP(5); /* pixel phase + wave reductions done (thread 0's wave) */
  /* wave totals -> LDS -> block sums -> accumulator record */
  if ((lane & 3) == 0) {
    s_wave[wave][word16_of_lane(lane)] = totA;
    s_wave[wave][16 + word16_of_lane(lane)] = totB;
  }
  __syncthreads();
  if (threadIdx.x < SUMA_ACC_WORDS) {
    long long s = 0;
#pragma unroll
    for (int w = 0; w < ICP_THREADS / 64; ++w) s += s_wave[w][threadIdx.x];
    unsigned long long* pout = reinterpret_cast<unsigned long long*>(g.pout) +
                               ((size_t)blockIdx.y * ICP_RECORDS + (blockIdx.x & (ICP_RECORDS - 1))) * SUMA_ACC_WORDS;
    atomicAdd(&pout[threadIdx.x], (unsigned long long)s);
  }
  if (PIXEL && g.fused_report != nullptr) {
    /* Objective-value-only pass (eval_only, one hypothesis) that closes itself: the block that finds every
     * other block's sums already added totals the records and reports to the host, so the pass needs no
     * consume-only launch behind it.  The 32 adding lanes sit in wave 0: the wait below completes all of
     * them (memory-side atomics) before lane 0 takes its number. */
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); /* completion of the adds, without an L2 write-back */
    if (threadIdx.x == 0)
      s_flag[3] = (__hip_atomic_fetch_add(g.fused_counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) ==
                   gridDim.x - 1u);
    __syncthreads();
    if (s_flag[3]) {
      if (threadIdx.x < ICP_RECORDS * SUMA_ACC_WORDS)
        s_tot[threadIdx.x >> 5][threadIdx.x & 31] =
            __hip_atomic_load(g.pout + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __syncthreads();
      if (threadIdx.x < SUMA_ACC_WORDS) {
        long long s = 0;
#pragma unroll
        for (int q = 0; q < ICP_RECORDS; ++q) s += s_tot[q][threadIdx.x];
        const long long n_valid = readlane_ll(s, 29), n_outlier = readlane_ll(s, 30);
        const long long n_invalid = readlane_ll(s, 31);
        const int w = threadIdx.x;
        /* the same bias rule as the consume step above */
        if (w < 29) s = (long long)((unsigned long long)s - (unsigned long long)lane_trips(a.P) * (unsigned long long)MAGIC_BITS);
        const double v = (double)s * (1.0 / SUMA_ACC_SCALE);
        HostResult* __restrict__ h = g.fused_report;
        if (w == 27) h->F = v;
        if (w == 28) h->F_inlier = v;
        if (g.host_full) h->acc[w] = s; /* Frame2Model::jacobianProducts through the C-ABI: JtJ / Jtr are formed by the host */
        if (w == 0) {
          h->valid = (uint32_t)n_valid;
          h->outlier = (uint32_t)n_outlier;
          h->invalid = (uint32_t)n_invalid;
          h->k = 0;
          h->converged = 0;
          h->iteration = iteration;
          *g.fused_counter = 0; /* re-armed for the next pass (stream order) */
        }
        __threadfence_system();
        if (w == 0) __hip_atomic_store(&h->seq, g.host_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
      }
    }
  }
  GN_STAMP(6); /* adds issued */
  if (prefetch_sink == 1.2345678e-30f && writer) gout->pad[0] = 1; /* keeps the prefetch loads alive */
  if (writer) gout->pending = 1;
}

/* Where IterArgs lies in the kernel-argument segment of the two entry points below (gn_args_again): behind N leading
 * pointers, at the next multiple of its own alignment -- written as what the ABI computes, and pinned, so that a new
 * leading parameter or a wider member of IterArgs cannot silently make the launches read another matrix as T0
 * (round-5 advisor; both paths are covered by the Gauss-Newton parity tests: T0 by the first launch of every chain,
 * pose_base by the scan pipeline's closing launch). */
#define GN_KARG_OFFSET(n_leading_pointers) \
  ((uint32_t)(((n_leading_pointers) * sizeof(void*) + alignof(IterArgs) - 1) / alignof(IterArgs) * alignof(IterArgs)))
static_assert(alignof(IterArgs) == 8 && sizeof(void*) == 8, "IterArgs follows the leading pointers without padding");
static_assert(GN_KARG_OFFSET(5) == 40u && GN_KARG_OFFSET(2) == 16u, "kernarg offsets of k_icp_step / k_icp_finish");

/* two entry points so that profiles tell the pixel launches from the closing consume-only launch */
/* The pointers behind a launch's FIRST loads are leading scalar parameters: with kernarg preloading (Makefile:
 * -amdgpu-kernarg-preload-count) they arrive in SGPRs with the wave, so the record / state / data-texel loads of the
 * prologue do not wait for a round trip to the freshly written kernarg segment first.  The struct holds the same values. */
__global__ void __launch_bounds__(ICP_THREADS)
    k_icp_step(const long long* pin, const GnState* gin, const float4* Vd, const float4* Nd, const float4* Sd, IterArgs g) {
  g.pin = pin;
  g.gin = gin;
  g.a.Vd = Vd;
  g.a.Nd = Nd;
  g.a.Sd = Sd;
  icp_iter_body<true>(g, GN_KARG_OFFSET(5)); /* five leading pointers, then IterArgs */
}
__global__ void __launch_bounds__(ICP_THREADS) k_icp_finish(const long long* pin, const GnState* gin, IterArgs g) {
  g.pin = pin;
  g.gin = gin;
  icp_iter_body<false>(g, GN_KARG_OFFSET(2)); /* two leading pointers, then IterArgs */
}

static IcpArgs make_args(suma_ctx* c) {
  IcpArgs a;
  const suma_frame *cur = c->icp_current, *mod = c->icp_model;
  a.Vd = cur->map[SUMA_MAP_VERTEX];
  a.Nd = cur->map[SUMA_MAP_NORMAL];
  a.Sd = cur->map[SUMA_MAP_SEMANTIC];
  a.Vm = mod->map[SUMA_MAP_VERTEX];
  a.Nm = mod->map[SUMA_MAP_NORMAL];
  a.Sm = mod->map[SUMA_MAP_SEMANTIC];
  a.W = (int32_t)cur->width;
  a.H = (int32_t)cur->height;
  a.Wm = (int32_t)mod->width;
  a.Hm = (int32_t)mod->height;
  a.fov_up = c->pd.fov_up;
  a.fov = c->pd.fov;
  /* Frame2Model.cpp:66-67; an adapter-side Frame2Model object may carry its own values (suma_icp_set_objective) */
  const float max_angle = c->obj_set ? c->obj.icp_max_angle : c->p.icp_max_angle;
  a.angle_thresh = (float)cos((double)max_angle * M_PI / 180.0);
  a.distance_thresh = c->obj_set ? c->obj.icp_max_distance : c->p.icp_max_distance;
  a.factor = c->obj_set ? c->obj.factor : c->p.factor;
  a.k8_enabled = 0;
  a.k8 = launch_k8_out(c);
  a.k8_ds = c->ds;
  a.weight_function = c->obj_set ? c->obj.weight_function : c->p.weight_function;
  a.bilinear = c->obj_set ? c->obj.bilinear_sampling : c->p.bilinear_sampling;
  a.P = (uint32_t)a.W * (uint32_t)a.H;
  return a;
}

/* state / partial buffers alternate with the launch parity c->gn_launch */
static GnState* gn_buf(suma_ctx* c, uint32_t parity) { return c->gn + (size_t)(parity & 1u) * SUMA_MAX_HYP; }
static long long* part_buf(suma_ctx* c, uint32_t launch) {
  return (long long*)c->gn_partial + (size_t)(launch % 3u) * SUMA_MAX_HYP * ICP_RECORDS * SUMA_ACC_WORDS;
}

/* iteration0 / iteration0_rest: Frame2Model::iteration_ the first / every other chain of the batch starts with (0 right
 * behind a setData; > 0 for a minimisation that follows another one on the same setData, SurfelMapping.cpp:693-700) */
hipError_t launch_gn_init(suma_ctx* c, const double* h_T0s, uint32_t n_hyp, int with_history, uint32_t iteration0,
                          uint32_t iteration0_rest) {
  double* hist = with_history ? c->gn_history : nullptr;
  c->gn_launch = 0;
  c->gn_init_pending = 0;
  if (n_hyp == 1) {
    /* single chain: no launch here, the first k_icp_iter takes the start state by value */
    for (int i = 0; i < 16; ++i) c->gn_T0_host[i] = h_T0s[i];
    c->gn_iteration0 = iteration0;
    c->gn_init_pending = 1;
  } else {
    hipError_t e = hipMemcpyAsync(c->gn_T0s, h_T0s, (size_t)n_hyp * 16 * sizeof(double), hipMemcpyHostToDevice, c->ls);
    if (e != hipSuccess) return e;
    k_gn_init<<<n_hyp, 64, 0, c->ls>>>(gn_buf(c, 0), c->gn_T0s, hist, iteration0, iteration0_rest);
  }
  return hipGetLastError();
}

/* one launch of the chain; pixel = 0 is the closing consume-only launch (one block per hypothesis) */
hipError_t launch_icp_iteration(suma_ctx* c, uint32_t n_hyp, uint32_t max_iter, double epsilon, double delta,
                                int eval_only, int with_history, int pixel) {
  IterArgs g;
  g.a = make_args(c);
  g.a.k8_enabled = (pixel && eval_only && n_hyp == 1 && c->gn_fuse_k8) ? 1 : 0;
  g.gin = gn_buf(c, c->gn_launch);
  g.gout = gn_buf(c, c->gn_launch + 1);
  /* the accumulator rotation runs across chains (gn_launch restarts with every chain, this does not) */
  const uint32_t rot = c->gn_part_launch++;
  g.pin = part_buf(c, rot);
  g.pout = part_buf(c, rot + 1);
  g.pzero = part_buf(c, rot + 2);
  g.zero_hyp = c->gn_part_dirty[(rot + 2) % 3u];
  c->gn_part_dirty[(rot + 2) % 3u] = 0;
  if (pixel && c->gn_part_dirty[(rot + 1) % 3u] < n_hyp) c->gn_part_dirty[(rot + 1) % 3u] = n_hyp;
  g.nblocks = c->icp_blocks;
  g.max_iter = max_iter;
  g.epsilon = epsilon;
  g.delta_thr = delta;
  g.eval_only = eval_only;
  g.pixel = pixel;
  g.history = with_history ? c->gn_history : nullptr;
  g.history_cap = c->gn_history_cap;
  g.emit_pose = (!pixel && c->gn_emit_pose) ? 1 : 0;
  for (int i = 0; i < 16; ++i) g.pose_base.m[i] = c->gn_pose_base[i];
  g.pose_block = c->pose_block;
  g.host_out = pixel ? nullptr : c->gn_host_out;
  g.host_seq = c->gn_host_seq;
  g.host_full = c->gn_host_full;
  g.ds = c->ds;
  g.fused_report = (pixel && eval_only && n_hyp == 1) ? c->gn_fused_report : nullptr;
  g.fused_counter = &c->ds->reserved0;
  g.init = c->gn_init_pending;
  g.iteration0 = c->gn_iteration0;
  for (int i = 0; i < 16; ++i) g.T0.m[i] = c->gn_T0_host[i];
  c->gn_init_pending = 0;
  c->gn_launch += 1;
  dim3 grid(pixel ? c->icp_blocks : 1, n_hyp);
  if (pixel)
    k_icp_step<<<grid, ICP_THREADS, 0, c->ls>>>(g.pin, g.gin, g.a.Vd, g.a.Nd, g.a.Sd, g);
  else
    k_icp_finish<<<grid, ICP_THREADS, 0, c->ls>>>(g.pin, g.gin, g);
  return hipGetLastError();
}

/* where the result of the last launch lives */
const GnState* gn_result(suma_ctx* c) { return gn_buf(c, c->gn_launch); }
