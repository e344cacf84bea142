// exact.hpp -- bit-exact restatement, on gfx950, of the reference's fp32 arithmetic.
//
// The reference (src/fp_abstraction.h:23-98) defines, for F = float:
//   _fma(acc, a, b)  = __fmaf_rd(a, b, acc)  fused multiply-add rounded toward -inf
//   _add/_sub/_mul                            IEEE round-to-nearest-even
//   _sqrt = __fsqrt_rn, _reciprocal = __frcp_rn   correctly rounded
// gfx950 has no per-instruction rounding: the f32 rounding mode is MODE[1:0].  The
// helpers below flip it around *groups* of independent v_fma_f32 inside one asm statement
// (so the compiler cannot schedule a round-to-nearest op into the round-down window) and
// leave the wave in round-to-nearest, the mode hipcc assumes everywhere else.
//
// This translation unit family is compiled with -ffp-contract=off (Makefile): the Kahan
// steps "t = acc + y; corr = y - (t - acc)" must stay three separate RN operations.
// sqrtf() and '/' are correctly rounded on HIP by default
// (-fhip-fp32-correctly-rounded-divide-sqrt).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace kmx {

#define KMX_RD_ON "s_nop 0\n\ts_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 2\n\t"
#define KMX_RD_OFF "s_nop 1\n\ts_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 0\n\t"

// one round-down FMA: returns RD(a*b + c)
__device__ __forceinline__ float fma_rd(float a, float b, float c) {
  float r;
  asm volatile(KMX_RD_ON "v_fma_f32 %0, %1, %2, %3\n\t" KMX_RD_OFF
               : "=v"(r)
               : "v"(a), "v"(b), "v"(c));
  return r;
}

// four independent round-down FMAs sharing the multiplicand a:  y[j] = RD(a*b[j] + c[j])
__device__ __forceinline__ void fma_rd4(float a, const float (&b)[4], const float (&c)[4], float (&y)[4]) {
  asm volatile(KMX_RD_ON
               "v_fma_f32 %0, %4, %5, %9\n\t"
               "v_fma_f32 %1, %4, %6, %10\n\t"
               "v_fma_f32 %2, %4, %7, %11\n\t"
               "v_fma_f32 %3, %4, %8, %12\n\t" KMX_RD_OFF
               : "=&v"(y[0]), "=&v"(y[1]), "=&v"(y[2]), "=&v"(y[3])
               : "v"(a), "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]), "v"(c[0]), "v"(c[1]), "v"(c[2]),
                 "v"(c[3]));
}

// two independent round-down FMAs sharing the multiplicand a / two squares (one lane, two chains)
__device__ __forceinline__ void fma_rd2(float a, const float (&b)[2], const float (&c)[2], float (&y)[2]) {
  asm volatile(KMX_RD_ON
               "v_fma_f32 %0, %2, %3, %5\n\t"
               "v_fma_f32 %1, %2, %4, %6\n\t" KMX_RD_OFF
               : "=&v"(y[0]), "=&v"(y[1])
               : "v"(a), "v"(b[0]), "v"(b[1]), "v"(c[0]), "v"(c[1]));
}
__device__ __forceinline__ void sqfma_rd2(const float (&d)[2], const float (&c)[2], float (&y)[2]) {
  asm volatile(KMX_RD_ON
               "v_fma_f32 %0, %2, %2, %4\n\t"
               "v_fma_f32 %1, %3, %3, %5\n\t" KMX_RD_OFF
               : "=&v"(y[0]), "=&v"(y[1])
               : "v"(d[0]), "v"(d[1]), "v"(c[0]), "v"(c[1]));
}

// eight independent round-down FMAs sharing the multiplicand a
__device__ __forceinline__ void fma_rd8(float a, const float (&b)[8], const float (&c)[8], float (&y)[8]) {
  asm volatile(KMX_RD_ON
               "v_fma_f32 %0, %8, %9, %17\n\t"
               "v_fma_f32 %1, %8, %10, %18\n\t"
               "v_fma_f32 %2, %8, %11, %19\n\t"
               "v_fma_f32 %3, %8, %12, %20\n\t"
               "v_fma_f32 %4, %8, %13, %21\n\t"
               "v_fma_f32 %5, %8, %14, %22\n\t"
               "v_fma_f32 %6, %8, %15, %23\n\t"
               "v_fma_f32 %7, %8, %16, %24\n\t" KMX_RD_OFF
               : "=&v"(y[0]), "=&v"(y[1]), "=&v"(y[2]), "=&v"(y[3]), "=&v"(y[4]), "=&v"(y[5]),
                 "=&v"(y[6]), "=&v"(y[7])
               : "v"(a), "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]), "v"(b[4]), "v"(b[5]), "v"(b[6]),
                 "v"(b[7]), "v"(c[0]), "v"(c[1]), "v"(c[2]), "v"(c[3]), "v"(c[4]), "v"(c[5]),
                 "v"(c[6]), "v"(c[7]));
}

// four independent round-down squares-plus:  y[j] = RD(d[j]*d[j] + c[j])
__device__ __forceinline__ void sqfma_rd4(const float (&d)[4], const float (&c)[4], float (&y)[4]) {
  asm volatile(KMX_RD_ON
               "v_fma_f32 %0, %4, %4, %8\n\t"
               "v_fma_f32 %1, %5, %5, %9\n\t"
               "v_fma_f32 %2, %6, %6, %10\n\t"
               "v_fma_f32 %3, %7, %7, %11\n\t" KMX_RD_OFF
               : "=&v"(y[0]), "=&v"(y[1]), "=&v"(y[2]), "=&v"(y[3])
               : "v"(d[0]), "v"(d[1]), "v"(d[2]), "v"(d[3]), "v"(c[0]), "v"(c[1]), "v"(c[2]), "v"(c[3]));
}

// The reference's "Kahan summation with inverted c" step (kmeans.cu:335-340,
// metric_abstraction.h:59-132), given y = RD(a*b + corr) already computed.
__device__ __forceinline__ void kahan_fold(float y, float &acc, float &corr) {
  const float t = acc + y;
  corr = y - (t - acc);
  acc = t;
}

// Serial Kahan dot product, one chain (used where a thread owns a whole vector pair).
__device__ __forceinline__ float kahan_dot(const float *__restrict__ a, const float *__restrict__ b, uint32_t D) {
  float acc = 0.f, corr = 0.f;
  for (uint32_t f = 0; f < D; f++) kahan_fold(fma_rd(a[f], b[f], corr), acc, corr);
  return acc;
}

// metric_abstraction.h:59-72 / :179-191: distance(v1, v2) for contiguous vectors.
template <int METRIC>
__device__ __forceinline__ float distance_vv(const float *__restrict__ a, const float *__restrict__ b, uint32_t D);

__device__ __forceinline__ float angular_from_prod(float fp) {
  // metric_abstraction.h:171-177 / :248-253
  if (fp >= 1.f) return 0.f;
  if (fp <= -1.f) return 3.14159265358979323846f;
  return acosf(fp);
}

// Serial chains over contiguous vectors; 16-byte loads when both rows allow it (the chain order
// is unchanged: the four elements of a load are folded in feature order).
template <int METRIC>
__device__ __forceinline__ float chain_vv(const float *__restrict__ a, const float *__restrict__ b, uint32_t D) {
  float acc = 0.f, corr = 0.f;
  uint32_t f = 0;
  const bool aligned16 = ((((uintptr_t)a) | ((uintptr_t)b)) & 15u) == 0;
  if ((D & 7u) == 0 && aligned16) {
    // 8 features per group, the next group's four 16-byte loads issued before the current group's
    // dependent steps: both rows are one thread's own (scattered) lines, every group a round trip
    auto load8 = [&](uint32_t f0, float4 (&av)[2], float4 (&bv)[2]) {
#pragma unroll
      for (int t = 0; t < 2; t++) {
        av[t] = *reinterpret_cast<const float4 *>(a + f0 + 4 * t);
        bv[t] = *reinterpret_cast<const float4 *>(b + f0 + 4 * t);
      }
    };
    auto fold8 = [&](const float4 (&av)[2], const float4 (&bv)[2]) {
#pragma unroll
      for (int t = 0; t < 2; t++) {
        const float aa[4] = {av[t].x, av[t].y, av[t].z, av[t].w}, bb[4] = {bv[t].x, bv[t].y, bv[t].z, bv[t].w};
#pragma unroll
        for (int q = 0; q < 4; q++) {
          if (METRIC == 0) {
            const float d = aa[q] - bb[q];
            kahan_fold(fma_rd(d, d, corr), acc, corr);
          } else {
            kahan_fold(fma_rd(aa[q], bb[q], corr), acc, corr);
          }
        }
      }
    };
    float4 a0[2], b0[2], a1[2], b1[2];
    load8(0, a0, b0);
    for (; f + 16 <= D; f += 16) {
      load8(f + 8, a1, b1);
      fold8(a0, b0);
      if (f + 24 <= D) load8(f + 16, a0, b0);
      fold8(a1, b1);
    }
    if (f + 8 <= D) {  // an odd number of groups: the last one is already loaded
      fold8(a0, b0);
      f += 8;
    }
  } else if ((D & 3u) == 0 && aligned16) {
    for (; f < D; f += 4) {
      const float4 av = *reinterpret_cast<const float4 *>(a + f), bv = *reinterpret_cast<const float4 *>(b + f);
      const float aa[4] = {av.x, av.y, av.z, av.w}, bb[4] = {bv.x, bv.y, bv.z, bv.w};
#pragma unroll
      for (int q = 0; q < 4; q++) {
        if (METRIC == 0) {
          const float d = aa[q] - bb[q];
          kahan_fold(fma_rd(d, d, corr), acc, corr);
        } else {
          kahan_fold(fma_rd(aa[q], bb[q], corr), acc, corr);
        }
      }
    }
  }
  for (; f < D; f++) {
    if (METRIC == 0) {
      const float d = a[f] - b[f];
      kahan_fold(fma_rd(d, d, corr), acc, corr);
    } else {
      kahan_fold(fma_rd(a[f], b[f], corr), acc, corr);
    }
  }
  return acc;
}

template <>
__device__ __forceinline__ float distance_vv<0>(const float *__restrict__ a, const float *__restrict__ b, uint32_t D) {
  return sqrtf(chain_vv<0>(a, b, D));
}

template <>
__device__ __forceinline__ float distance_vv<1>(const float *__restrict__ a, const float *__restrict__ b, uint32_t D) {
  return angular_from_prod(chain_vv<1>(a, b, D));
}

// metric_abstraction.h:55-57 (L2): distance(0, csqr, prod) = RD(-2*prod + (0 + csqr));
// :171-177 (angular): acos rule on the product.
template <int METRIC>
__device__ __forceinline__ float lloyd_distance(float csqr, float prod) {
  if (METRIC == 0) return fma_rd(-2.f, prod, 0.f + csqr);
  return angular_from_prod(prod);
}

}  // namespace kmx
