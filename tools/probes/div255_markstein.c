// exhaustive: for every finite non-negative float a, compare RN(a/255) with Markstein's q' = fma(r, y, q), y = RN(1/255), q = RN(a*y), r = fma(-255, q, a)
#include <stdio.h>
#include <math.h>
#include <string.h>
#include <stdint.h>
#include <omp.h>
int main(void) {
  const float y = 1.0f / 255.0f;
  unsigned long long bad = 0, badfinal = 0, checked = 0;
  uint32_t first_bad = 0;
#pragma omp parallel for reduction(+:bad, badfinal, checked) schedule(static)
  for (uint64_t u = 0; u < 0x7f800000ull; ++u) {
    uint32_t b = (uint32_t)u; float a; memcpy(&a, &b, 4);
    volatile float q0 = a / 255.0f;
    float q = a * y;
    float r = fmaf(-255.0f, q, a);
    float q1 = fmaf(r, y, q);
    checked++;
    if (q1 != q0) { bad++; 
      float f0 = 2.f * q0 - 1.f, f1 = 2.f * q1 - 1.f;
      if (f0 != f1) { badfinal++; }
      if (bad == 1) { first_bad = b; }
    }
  }
  printf("checked %llu, quotient differs %llu, normalised value differs %llu (one example bits 0x%08x)\n", checked, bad, badfinal, first_bad);
  return 0;
}
