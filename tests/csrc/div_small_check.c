/* Host check of the quotient k_cg1b's Jacobi form computes without the division sequence (oi_kernels.hip: div_small):
 * for a neighbour count c = 1 .. 6 and y = RN(1 / c):  q = RN(a y);  r = fma(-c, q, a);  q' = fma(r, y, q)  must be the correctly
 * rounded a / c.  Random operands over 120 binades plus mantissa edge patterns; prints the number of operands and of mismatches.
 * usage: div_small_check <random operands per divisor> */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
static inline uint64_t rng(uint64_t* s) { uint64_t x = *s; x ^= x << 13; x ^= x >> 7; x ^= x << 17; return *s = x; }
static long check(double a, double dc, double y, long* bad) {
  const double q = a * y, r = fma(-dc, q, a), v = fma(r, y, q), w = a / dc;
  if (v != w && *bad < 5) printf("c %g a %a: %a vs %a\n", dc, a, v, w);
  if (v != w) ++*bad;
  return 1;
}
int main(int argc, char** argv) {
  const long nrand = argc > 1 ? atol(argv[1]) : 10000000L;
  uint64_t s = 88172645463325252ull;
  long bad = 0, n = 0;
  for (int c = 1; c <= 6; ++c) {
    const double dc = (double)c, y = 1.0 / dc;
    for (long it = 0; it < nrand; ++it) {
      const uint64_t m = rng(&s);
      const uint64_t bits = (m & 0x800fffffffffffffull) | ((uint64_t)(1023 - 60 + (int)((m >> 52) % 120)) << 52);
      double a;
      memcpy(&a, &bits, 8);
      n += check(a, dc, y, &bad);
    }
    for (uint64_t mant = 0; mant < (1ull << 18); ++mant)
      for (int hi = 0; hi < 4; ++hi) {
        const uint64_t bits = (1023ull << 52) | (hi == 0 ? mant : hi == 1 ? (0xfffffffffffffull - mant) : hi == 2 ? (mant << 34) : (0x8000000000000ull ^ mant));
        double a;
        memcpy(&a, &bits, 8);
        n += check(a, dc, y, &bad);
      }
  }
  printf("%ld quotients, %ld differ\n", n, bad);
  return bad != 0;
}
