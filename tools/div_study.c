/* Study for a cheaper bit-exact fp32 division on gfx950: is the UNSCALED core of the
 * compiler's division sequence -- rcp, two Newton fmas on the reciprocal, quotient, two fma corrections -- correctly
 * rounded for finite operands in a safe exponent range, WHATEVER the hardware's 1-ulp reciprocal approximation returns?
 * If so, three quotients by one denominator (v / length(v), the barycentrics of a triangle) cost 3 + 3 x 5 instructions
 * instead of 3 x 11, behind one exponent-range test.  CPU emulation with fmaf (exactly the GPU's v_fma_f32):
 *     gcc -O2 -ffp-contract=off -Itools tools/div_study.c -o /tmp/div_study -lm && /tmp/div_study 400000000 [hard]
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static inline float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static inline uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static uint64_t s = 0x9E3779B97F4A7C15ull;
static inline uint64_t rnd(void) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; }

#include "../semantic_suma_amd/csrc/exact_div.h" /* the very functions a kernel would use */

/* r0: any approximation of 1/y within 1 ulp (v_rcp_f32) */
static inline float div_core(float x, float y, float r0) { return exdiv_quot(x, y, exdiv_refine(y, r0)); }

int main(int argc, char** argv) {
  uint64_t n = argc > 1 ? strtoull(argv[1], 0, 10) : 100000000ull, bad = 0, tested = 0;
  const int ints = argc > 2 && argv[2][0] == 'i'; /* "int": integer operands as in k_render's barycentrics */
  const int hard = argc > 2 && !ints;             /* any other second argument: the near-boundary numerators */
  for (uint64_t i = 0; i < n; ++i) {
    uint64_t a = rnd(), b = rnd();
    /* sign, exponent in [-60, 60], random mantissa; every 8th pair with structured mantissas (all ones, one bit, equal) */
    uint32_t mx = (uint32_t)a & 0x7fffffu, my = (uint32_t)b & 0x7fffffu;
    if ((i & 7) == 0) { mx = (a >> 40) & 1 ? 0x7fffffu : (1u << ((a >> 41) % 23)); }
    if ((i & 15) == 0) { my = (b >> 40) & 1 ? 0x7fffffu : 0u; }
    if ((i & 31) == 0) { my = mx; }
    uint32_t ex = 127 - 60 + (uint32_t)((a >> 24) % 121), ey = 127 - 60 + (uint32_t)((b >> 24) % 121);
    float x = u2f(((uint32_t)(a >> 63) << 31) | (ex << 23) | mx), y = u2f(((uint32_t)(b >> 63) << 31) | (ey << 23) | my);
    if (ints) { /* the barycentrics of k_render: integer edge functions 0 <= w <= area < 2^46 converted to float */
      uint64_t area = 1 + (b >> (18 + (int)((a >> 58) % 40))), w = (a >> 18) % (area + 1);
      if ((i & 63) == 0) w = 0;
      if ((i & 127) == 1) w = area;
      x = (float)(long long)w;
      y = (float)(long long)area;
    } else if (hard) { /* numerators whose quotient lies next to a rounding boundary: x = RN(y * (q + ulp(q) / 2)) */
      float qm = u2f((127u << 23) | mx);
      double mid = (double)qm + ldexp(1.0, -24);
      x = (float)((double)y * mid);
    }
    float want = x / y, rc = 1.0f / y;
    for (int d = -1; d <= 1; ++d) {
      float r0 = u2f(f2u(rc) + d);
      float got = div_core(x, y, r0);
      tested++;
      if (f2u(got) != f2u(want)) {
        if (bad < 10) printf("MISMATCH x=%a y=%a r0=%a (d=%d): got %a want %a\n", x, y, r0, d, got, want);
        bad++;
      }
    }
  }
  printf("%llu %squotients (exponents of x, y in [-60, 60], reciprocal approximations -1 / 0 / +1 ulp): %llu mismatches\n",
         (unsigned long long)tested, ints ? "integer-operand " : hard ? "near-boundary " : "", (unsigned long long)bad);
  return bad != 0;
}
