/* exact_div.h -- correctly rounded fp32 quotients without v_div_scale / v_div_fmas / v_div_fixup.
 *
 * The core of the compiler's own division sequence -- hardware reciprocal (any approximation within 1 ulp), one Newton
 * fma pair on it, quotient, two fma corrections -- which tools/div_study.c finds equal to x / y on 1.5e9 emulated
 * quotients (random, structured and next to rounding boundaries; reciprocal perturbed by -1 / 0 / +1 ulp) as long as no
 * intermediate leaves the normal range: finite operands, y != 0, exponents of x and y within [-60, 60].  Outside that
 * range the caller divides plainly.  The reciprocal's Newton step depends on the denominator only: k quotients by one
 * denominator cost 3 + 5 k instructions instead of 11 k.  Used where the range is known BY CONSTRUCTION: the three
 * barycentrics of a covered triangle in k_render.hip (integers below 2^46 over a positive integer area) -- the same
 * bits as `/`, fewer instructions.  The fmas here are not an arithmetic-specification matter: they are an
 * implementation of the correctly rounded quotient the specification asks for.
 *
 * EXDIV_RCP(y): the 1-ulp reciprocal -- __builtin_amdgcn_rcpf on the device; the study passes perturbed values. */
#ifndef SUMA_EXACT_DIV_H
#define SUMA_EXACT_DIV_H
#include <stdint.h>
#include <string.h>

#ifndef EXDIV_FN
#define EXDIV_FN static inline
#endif

/* the refined reciprocal of y from a 1-ulp approximation r0 */
EXDIV_FN float exdiv_refine(float y, float r0) {
  float e = __builtin_fmaf(-y, r0, 1.0f);
  return __builtin_fmaf(e, r0, r0);
}
/* x / y given the refined reciprocal r of y */
EXDIV_FN float exdiv_quot(float x, float y, float r) {
  float q = x * r;
  float e1 = __builtin_fmaf(-y, q, x);
  q = __builtin_fmaf(e1, r, q);
  float e2 = __builtin_fmaf(-y, q, x);
  return __builtin_fmaf(e2, r, q);
}
/* all operands finite, y != 0, biased exponents of |v| within 127 +- 60: the fast path is exact there */
EXDIV_FN int exdiv_safe(float v) {
  uint32_t u;
  memcpy(&u, &v, 4);
  uint32_t e = (u >> 23) & 0xffu;
  return e >= 127u - 60u && e <= 127u + 60u;
}
#endif
