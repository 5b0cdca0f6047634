// Issue rates of the VALU instructions the map kernels lean on, measured on the device (gfx950): every kernel runs
// ITER trips of 16 independent instructions of ONE kind per wave, 8 waves per SIMD resident on every CU, so that the
// figure is the issue rate and not a latency.  Prints cycles per wave-instruction per SIMD relative to v_fma_f32 (= 4
// on a 16-lane SIMD), i.e. "1.0" = full rate, "4.0" = quarter rate.
//   hipcc --offload-arch=gfx950 -O2 tools/instr_rate.hip -o /tmp/instr_rate && /tmp/instr_rate
// Why: round 5 moved k_render's edge functions from 64-bit integers to exact binary64 (-17 % static VALU in the pixel
// phase, 57 -> 2 quarter-rate multiplies) and the pass did not get faster; this table says what an instruction costs.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define ITER 2000
#define REP16(X) X X X X X X X X X X X X X X X X

#define KERNEL(name, decl, body, sink)                                   \
  __global__ void __launch_bounds__(512) name(float* out, int n) {      \
    decl;                                                                \
    for (int it = 0; it < n; ++it) { body }                              \
    if (threadIdx.x == 9999) out[0] = (float)(sink);                     \
  }

// sixteen independent accumulators per kind; inline asm pins the instruction
#define F32_DECL float a0 = threadIdx.x, a1 = 1, a2 = 2, a3 = 3, a4 = 4, a5 = 5, a6 = 6, a7 = 7, b = 1.0001f, c = 0.5f
#define F64_DECL double a0 = threadIdx.x, a1 = 1, a2 = 2, a3 = 3, a4 = 4, a5 = 5, a6 = 6, a7 = 7, b = 1.0001, c = 0.5
#define I32_DECL uint32_t a0 = threadIdx.x, a1 = 1, a2 = 2, a3 = 3, a4 = 4, a5 = 5, a6 = 6, a7 = 7, b = 12345u, c = 77u
#define I64_DECL unsigned long long a0 = threadIdx.x, a1 = 1, a2 = 2, a3 = 3, a4 = 4, a5 = 5, a6 = 6, a7 = 7; uint32_t b = 12345u, c = 77u

#define OP8(S, CON) \
  asm volatile(S : "+v"(a0) : CON(b), CON(c)); asm volatile(S : "+v"(a1) : CON(b), CON(c)); \
  asm volatile(S : "+v"(a2) : CON(b), CON(c)); asm volatile(S : "+v"(a3) : CON(b), CON(c)); \
  asm volatile(S : "+v"(a4) : CON(b), CON(c)); asm volatile(S : "+v"(a5) : CON(b), CON(c)); \
  asm volatile(S : "+v"(a6) : CON(b), CON(c)); asm volatile(S : "+v"(a7) : CON(b), CON(c));
#define V "v"

KERNEL(k_fma_f32, F32_DECL, OP8("v_fma_f32 %0, %0, %1, %2", V) OP8("v_fma_f32 %0, %0, %1, %2", V), a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7)
KERNEL(k_fma_f64, F64_DECL, OP8("v_fma_f64 %0, %0, %1, %2", V) OP8("v_fma_f64 %0, %0, %1, %2", V), a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7)
KERNEL(k_mul_f64, F64_DECL, OP8("v_mul_f64 %0, %0, %1", V) OP8("v_mul_f64 %0, %0, %1", V), a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7)
KERNEL(k_add_f64, F64_DECL, OP8("v_add_f64 %0, %0, %1", V) OP8("v_add_f64 %0, %0, %1", V), a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7)
KERNEL(k_mul_lo_u32, I32_DECL, OP8("v_mul_lo_u32 %0, %0, %1", V) OP8("v_mul_lo_u32 %0, %0, %1", V), a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7)
KERNEL(k_mul_hi_u32, I32_DECL, OP8("v_mul_hi_u32 %0, %0, %1", V) OP8("v_mul_hi_u32 %0, %0, %1", V), a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7)
KERNEL(k_mad_u32_u24, I32_DECL, OP8("v_mad_u32_u24 %0, %0, %1, %2", V) OP8("v_mad_u32_u24 %0, %0, %1, %2", V), a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7)
KERNEL(k_mad_u64_u32, I64_DECL, OP8("v_mad_u64_u32 %0, vcc, %1, %2, %0", V) OP8("v_mad_u64_u32 %0, vcc, %1, %2, %0", V), a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7)
KERNEL(k_lshl_b64, I64_DECL, OP8("v_lshlrev_b64 %0, 1, %0", V) OP8("v_lshlrev_b64 %0, 1, %0", V), a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7)
KERNEL(k_rcp_f32, F32_DECL, OP8("v_rcp_f32 %0, %0", V) OP8("v_rcp_f32 %0, %0", V), a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7)
KERNEL(k_sqrt_f32, F32_DECL, OP8("v_sqrt_f32 %0, %0", V) OP8("v_sqrt_f32 %0, %0", V), a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7)
KERNEL(k_and_b32, I32_DECL, OP8("v_and_b32 %0, %0, %1", V) OP8("v_and_b32 %0, %0, %1", V), a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7)
KERNEL(k_ffbh_u32, I32_DECL, OP8("v_ffbh_u32 %0, %0", V) OP8("v_ffbh_u32 %0, %0", V), a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7)
KERNEL(k_div_fixup_f32, F32_DECL, OP8("v_div_fixup_f32 %0, %0, %1, %2", V) OP8("v_div_fixup_f32 %0, %0, %1, %2", V), a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7)
KERNEL(k_div_scale_f32, F32_DECL, OP8("v_div_scale_f32 %0, vcc, %0, %1, %2", V) OP8("v_div_scale_f32 %0, vcc, %0, %1, %2", V), a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7)
KERNEL(k_cndmask, F32_DECL; asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(b), "v"(c) : "vcc"), OP8("v_cndmask_b32 %0, %0, %1, vcc", V) OP8("v_cndmask_b32 %0, %0, %1, vcc", V), a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7)

KERNEL(k_add_f32, F32_DECL, OP8("v_add_f32 %0, %0, %1", V) OP8("v_add_f32 %0, %0, %1", V), a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7)
KERNEL(k_mul_f32, F32_DECL, OP8("v_mul_f32 %0, %0, %1", V) OP8("v_mul_f32 %0, %0, %1", V), a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7)
KERNEL(k_max_f32, F32_DECL, OP8("v_max_f32 %0, %0, %1", V) OP8("v_max_f32 %0, %0, %1", V), a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7)
KERNEL(k_add_u32, I32_DECL, OP8("v_add_u32 %0, %0, %1", V) OP8("v_add_u32 %0, %0, %1", V), a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7)
KERNEL(k_lshl_b32, I32_DECL, OP8("v_lshlrev_b32 %0, 1, %0", V) OP8("v_lshlrev_b32 %0, 1, %0", V), a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7)
KERNEL(k_mov_b32, F32_DECL, OP8("v_mov_b32 %0, %1", V) OP8("v_mov_b32 %0, %2", V), a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7)
KERNEL(k_cvt_i32_f32, F32_DECL, OP8("v_cvt_i32_f32 %0, %1", V) OP8("v_cvt_i32_f32 %0, %2", V), a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7)
KERNEL(k_floor_f32, F32_DECL, OP8("v_floor_f32 %0, %1", V) OP8("v_floor_f32 %0, %2", V), a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7)
KERNEL(k_fma_f32_2src, F32_DECL, OP8("v_fma_f32 %0, %1, %2, %0", V) OP8("v_fma_f32 %0, %1, %2, %0", V), a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7)
KERNEL(k_pk_fma_f32, F64_DECL, OP8("v_pk_fma_f32 %0, %0, %1, %2", V) OP8("v_pk_fma_f32 %0, %0, %1, %2", V), a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7)
// the select forms: mask in vcc (VOP2) / in an SGPR pair (VOP3), destination in place / elsewhere
KERNEL(k_cndmask_sgpr, F32_DECL; unsigned long long m = 0x5555555555555555ull; asm volatile("" : "+s"(m)), OP8("v_cndmask_b32_e64 %0, %0, %1, s[2:3]", V) OP8("v_cndmask_b32_e64 %0, %0, %1, s[2:3]", V), a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (float)m)
KERNEL(k_cndmask_dst, F32_DECL; asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(b), "v"(c) : "vcc"), OP8("v_cndmask_b32 %0, %1, %2, vcc", V) OP8("v_cndmask_b32 %0, %2, %1, vcc", V), a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7)
// the idiom of the kernels: compare + select
KERNEL(k_cmp_cndmask, F32_DECL, OP8("v_cmp_lt_f32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %2, vcc", V), a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7)

// conversions and compares need mixed register classes: written out
__global__ void __launch_bounds__(512) k_cvt_f64_i32(float* out, int n) {
  double a0 = 0, a1 = 0, a2 = 0, a3 = 0, a4 = 0, a5 = 0, a6 = 0, a7 = 0;
  int b = threadIdx.x;
  for (int it = 0; it < n; ++it) {
#define CV(A) asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(A) : "v"(b));
    CV(a0) CV(a1) CV(a2) CV(a3) CV(a4) CV(a5) CV(a6) CV(a7) CV(a0) CV(a1) CV(a2) CV(a3) CV(a4) CV(a5) CV(a6) CV(a7)
  }
  if (threadIdx.x == 9999) out[0] = (float)(a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7);
}
__global__ void __launch_bounds__(512) k_cvt_f32_f64(float* out, int n) {
  float a0 = 0, a1 = 0, a2 = 0, a3 = 0, a4 = 0, a5 = 0, a6 = 0, a7 = 0;
  double b = threadIdx.x;
  for (int it = 0; it < n; ++it) {
#define CW(A) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(A) : "v"(b));
    CW(a0) CW(a1) CW(a2) CW(a3) CW(a4) CW(a5) CW(a6) CW(a7) CW(a0) CW(a1) CW(a2) CW(a3) CW(a4) CW(a5) CW(a6) CW(a7)
  }
  if (threadIdx.x == 9999) out[0] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}
__global__ void __launch_bounds__(512) k_cmp_f64(float* out, int n) {
  double b = threadIdx.x, c = 3.0;
  unsigned long long m = 0;
  for (int it = 0; it < n; ++it) {
#define CP asm volatile("v_cmp_lt_f64 vcc, %0, %1" : : "v"(b), "v"(c) : "vcc");
    REP16(CP)
  }
  asm volatile("s_mov_b64 %0, vcc" : "=s"(m));
  if (threadIdx.x == 9999) out[0] = (float)m;
}
__global__ void __launch_bounds__(512) k_cmp_i64(float* out, int n) {
  long long b = threadIdx.x, c = 3;
  unsigned long long m = 0;
  for (int it = 0; it < n; ++it) {
#define CQ asm volatile("v_cmp_lt_i64 vcc, %0, %1" : : "v"(b), "v"(c) : "vcc");
    REP16(CQ)
  }
  asm volatile("s_mov_b64 %0, vcc" : "=s"(m));
  if (threadIdx.x == 9999) out[0] = (float)m;
}
__global__ void __launch_bounds__(512) k_cmp_f32(float* out, int n) {
  float b = threadIdx.x, c = 3.0f;
  unsigned long long m = 0;
  for (int it = 0; it < n; ++it) {
#define CR asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(b), "v"(c) : "vcc");
    REP16(CR)
  }
  asm volatile("s_mov_b64 %0, vcc" : "=s"(m));
  if (threadIdx.x == 9999) out[0] = (float)m;
}

template <typename K>
static double run(K kern, float* d) {
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  // 256 CUs x 4 blocks of 512 threads: 8 waves per SIMD on every CU; best of 5 (the clock settles during the first)
  double best = 1e30;
  for (int rep = 0; rep < 5; ++rep) {
    hipEventRecord(a);
    kern<<<1024, 512>>>(d, ITER);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    if (ms < best) best = ms;
  }
  return best;
}

int main() {
  float* d;
  hipMalloc(&d, 64);
  run(k_fma_f32, d);
  const double base = run(k_fma_f32, d);
  printf("v_fma_f32 x %d x 16 per wave, 8 waves per SIMD on 1024 SIMDs: %.3f ms = %.2f GHz at four cycles per wave instruction\n", ITER, base,
         ITER * 16.0 * 8.0 * 4.0 / (base * 1e6));
#define ROW(k) printf("%-18s %5.2f\n", &#k[2], run(k, d) / base);
  ROW(k_fma_f32) ROW(k_and_b32) ROW(k_cndmask) ROW(k_cmp_f32) ROW(k_fma_f64) ROW(k_mul_f64) ROW(k_add_f64) ROW(k_cvt_f64_i32) ROW(k_cvt_f32_f64)
  ROW(k_cmp_f64) ROW(k_cmp_i64) ROW(k_mul_lo_u32) ROW(k_mul_hi_u32) ROW(k_mad_u32_u24) ROW(k_mad_u64_u32) ROW(k_lshl_b64)
  ROW(k_ffbh_u32) ROW(k_rcp_f32) ROW(k_sqrt_f32) ROW(k_div_scale_f32) ROW(k_div_fixup_f32) ROW(k_fma_f32)
  ROW(k_add_f32) ROW(k_mul_f32) ROW(k_max_f32) ROW(k_add_u32) ROW(k_lshl_b32) ROW(k_mov_b32) ROW(k_cvt_i32_f32) ROW(k_floor_f32)
  ROW(k_fma_f32_2src) ROW(k_pk_fma_f32) ROW(k_cndmask_sgpr) ROW(k_cndmask_dst)
  printf("(the next row: 8 compare + select PAIRS per trip = 16 instructions, as the other rows)\n");
  ROW(k_cmp_cndmask) ROW(k_fma_f32)
  return 0;
}
