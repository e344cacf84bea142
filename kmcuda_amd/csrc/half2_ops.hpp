// half2_ops.hpp -- the reference's fp16x2 ARITHMETIC (src/fp_abstraction.h:100-182) as device functions: F = half2,
// every operation a binary16 operation rounded to nearest even, Kahan sums as two interleaved half accumulators.
// Shared by half2_strict.hip (Lloyd / Yinyang / seeding) and knn.hip (the strict k-NN).  Values travel as fp32 words
// that hold exactly representable halves.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "exact.hpp"

namespace kmx {

typedef _Float16 hf;

__device__ __forceinline__ hf h_fma(hf a, hf b, hf c) { return __builtin_fmaf16(a, b, c); }   // __hfma: one rounding
__device__ __forceinline__ hf h_ld(const float *p, size_t i) { return (hf)p[i]; }               // exact: p holds halves
// __int2half_rd: the largest half not above v (never +inf)
__device__ __forceinline__ hf h_from_int_rd(uint32_t v) {
  const float f = v > 65504u ? 65504.f : (float)v;   // exact below 2^24; above 65504 the answer is 65504
  hf h = (hf)f;                                      // nearest
  if ((float)h > f) {                                // step to the next half below
    unsigned short bits = __builtin_bit_cast(unsigned short, h);
    h = __builtin_bit_cast(hf, (unsigned short)(bits - 1));   // positive finite: the previous bit pattern
  }
  return h;
}
#define H2_KAHAN(acc, corr, a, b) do { const hf y__ = h_fma((a), (b), (corr)); const hf t__ = (acc) + y__; \
                                       (corr) = y__ - (t__ - (acc)); (acc) = t__; } while (0)

// metric_abstraction.h:55-57 (L2), :171-177 (angular): distance(sqr1, sqr2, prod) -> half
template <int METRIC>
__device__ __forceinline__ hf h2_distance3(hf sq_lo, hf sq_hi, hf p_lo, hf p_hi) {
  if (METRIC == 0) {
    const hf lo = h_fma((hf)-2.f, p_lo, (hf)0.f + sq_lo), hi = h_fma((hf)-2.f, p_hi, (hf)0.f + sq_hi);
    return hi + lo;   // _fin
  }
  const float fp = (float)(p_hi + p_lo);
  if (fp >= 1.f) return (hf)0.f;
  if (fp <= -1.f) return (hf)3.14159265358979323846f;
  return (hf)acosf(fp);
}
// METRIC::distance / distance_t / distance_tt (:59-101, :179-218) -> float
template <int METRIC>
__device__ __forceinline__ float h2_distance(const float *a, const float *b, uint32_t D) {
  hf s0 = 0, s1 = 0, c0 = 0, c1 = 0;
  if (METRIC == 0) {
    for (uint32_t f = 0; f + 1 < D; f += 2) {
      const hf d0 = h_ld(a, f) - h_ld(b, f), d1 = h_ld(a, f + 1) - h_ld(b, f + 1);
      H2_KAHAN(s0, c0, d0, d0);
      H2_KAHAN(s1, c1, d1, d1);
    }
    return sqrtf((float)(s1 + s0));   // _sqrt(_float(_fin(dist))): fp32 sqrt of the half sum
  }
  for (uint32_t f = 0; f + 1 < D; f += 2) {
    H2_KAHAN(s0, c0, h_ld(a, f), h_ld(b, f));
    H2_KAHAN(s1, c1, h_ld(a, f + 1), h_ld(b, f + 1));
  }
  return (float)h2_distance3<METRIC>((hf)1.f, (hf)1.f, s0, s1);
}

// METRIC::partial (metric_abstraction.h:103-118 / :220-232), F = half2: the Kahan sum of n fp32-held halves (n even:
// n / 2 half2 elements), folded by _fin and widened -- the caller adds such partials in fp32 and finalizes
template <int METRIC>
__device__ __forceinline__ float h2_partial(const float *a, const float *b, uint32_t n) {
  hf s0 = 0, s1 = 0, c0 = 0, c1 = 0;
  for (uint32_t f = 0; f + 1 < n; f += 2) {
    if (METRIC == 0) {
      const hf d0 = h_ld(a, f) - h_ld(b, f), d1 = h_ld(a, f + 1) - h_ld(b, f + 1);
      H2_KAHAN(s0, c0, d0, d0);
      H2_KAHAN(s1, c1, d1, d1);
    } else {
      H2_KAHAN(s0, c0, h_ld(a, f), h_ld(b, f));
      H2_KAHAN(s1, c1, h_ld(a, f + 1), h_ld(b, f + 1));
    }
  }
  return (float)(s1 + s0);
}

}  // namespace kmx
