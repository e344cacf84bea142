// filter_common.hpp -- the tail shared by the Lloyd filter kernels (lloyd.hip: f32 MFMA,
// lloyd_f16.hip: f16 MFMA on hi/lo-split operands): merge the two half-waves' running top-3,
// decide against the error bound, commit or hand the row to the exact kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace kmx {

// commit (kmeans.cu:358-363): prev[s] = old; if old != nearest { assign; ++changed }
__device__ __forceinline__ bool commit_row(uint32_t s, uint32_t nearest, uint32_t *__restrict__ assignments,
                                           uint32_t *__restrict__ assignments_prev) {
  const uint32_t old = assignments[s];
  assignments_prev[s] = old;
  if (old != nearest) {
    assignments[s] = nearest;
    return true;
  }
  return false;
}


// The angular metric's clamp (metric_abstraction.h:171-177): dist = p >= 1 ? 0 : p <= -1 ? pi : acos(p).  Every
// centroid whose product with a row reaches 1 sits at distance 0 and the ascending strict-'<' scan of
// kmeans_assign_lloyd (kmeans.cu:342-346) keeps the LOWEST INDEX among them, not the largest product (likewise at
// -1 / pi).  A filter decides on scores s(c) = p(c) - x.mu (x.mu constant per row: operands are centred by mu), so
// it may only decide a row when every centroid it rules out has a product provably BELOW 1 and the winner's is
// provably ABOVE -1; all other rows belong to the exact kernels, which follow the clamp (exact.hpp).
//   hi: a score below it is a product below 1 in the reference's arithmetic
//   lo: a score above it is a product above -1
// xm = x.mu as computed, exm bounds its error, ehalf bounds |score - reference score| (half the decision threshold).
// L2: no limits.  NaN operands give NaN limits: every comparison fails and the row goes to the exact kernels.
struct ClampLimits {
  float hi, lo;
};
__device__ __forceinline__ ClampLimits clamp_limits(bool angular, float xm, float exm, float ehalf) {
  ClampLimits l;
  if (!angular) {
    l.hi = INFINITY;
    l.lo = -INFINITY;
    return l;
  }
  // (2e-6: the roundings of this very arithmetic; 5 u |xm|: the subtraction's when x.mu is large)
  const float m = (ehalf + exm) * 1.001f + 2.0e-6f + 3.0e-7f * fabsf(xm);
  l.hi = (1.0f - m) - xm;
  l.lo = (m - 1.0f) - xm;
  return l;
}
// the error of an fp32 FMA sum of n products whose absolute values add up to at most `mass`
__device__ __forceinline__ float dot_error(int n, float mass) { return 6.1e-8f * (float)(n + 8) * mass; }

// x.mu and sum |x_f mu_f| over the features [f0, f1) of one row, for the kernels that do not have them on record:
// a loop of its own behind the matrix sweep (angular passes only), so that the L2 instantiations keep their registers.
template <typename T>
__device__ __forceinline__ void row_dot_mu(const T *__restrict__ xr, const float *__restrict__ mu, uint32_t f0,
                                           uint32_t f1, float &dot, float &mass) {
  float d = 0.f, m = 0.f;
#pragma unroll 4
  for (uint32_t f = f0; f < f1; f++) {
    const float x = (float)xr[f], u = mu[f];
    d = fmaf(x, u, d);
    m = fmaf(fabsf(x), fabsf(u), m);
  }
  dot = d;
  mass = m;
}

// codes: tile*16 + accumulator register; a lane's register r of tile t is centroid
// t*32 + (r&3) + 8*(r>>2) + 4*half.
//   v1 - v2 > thr : the reference's distance to i1 is strictly the smallest -> commit
//   v1 - v3 > thr : the minimum is i1 or i2 -> two exact Kahan distances settle it (pair list)
//   otherwise     : three or more contenders -> full exact scan (flagged list)
// lim (angular, clamp_limits above): the centroids a decision rules out must score below lim.hi, its winner above lim.lo
// thr = 2E with |score_filter - score_ref| <= E for every centroid up to a term constant in c
// (DESIGN.md 4.1); a NaN gap or NaN thr is "not certain".
__device__ __forceinline__ void filter_finish(float v1, float v2, float v3, uint32_t c1, uint32_t c2, int h, int lane,
                                              uint32_t s, uint32_t N, uint32_t K, bool insane, float thr,
                                              ClampLimits lim, uint32_t *__restrict__ assignments,
                                              uint32_t *__restrict__ assignments_prev,
                                              uint32_t *__restrict__ flagged, uint32_t *__restrict__ pairs,
                                              uint32_t *__restrict__ counters) {
  auto decode = [&](uint32_t code, int half) -> uint32_t {
    if (code == 0xFFFFFFFFu) return 0xFFFFFFFFu;
    const uint32_t r = code & 15u;
    return (code >> 4) * 32u + (r & 3u) + 8u * (r >> 2) + 4u * half;
  };
  uint32_t i1 = decode(c1, h), i2 = decode(c2, h);
  {
    const float pv1 = __shfl_xor(v1, 32), pv2 = __shfl_xor(v2, 32), pv3 = __shfl_xor(v3, 32);
    const uint32_t pi1 = __shfl_xor(i1, 32), pi2 = __shfl_xor(i2, 32);
    auto insert = [&](float v, uint32_t idx) {
      const bool g1 = v > v1, g2 = v > v2, g3 = v > v3;
      v3 = g2 ? v2 : (g3 ? v : v3);
      i2 = g1 ? i1 : (g2 ? idx : i2);
      v2 = g1 ? v1 : (g2 ? v : v2);
      i1 = g1 ? idx : i1;
      v1 = g1 ? v : v1;
    };
    insert(pv1, pi1);
    insert(pv2, pi2);
    insert(pv3, 0xFFFFFFFFu);  // can only land in third place
  }
  const bool certain = insane || (((v1 - v2) > thr) && (v2 < lim.hi) && (v1 > lim.lo));
  const bool two = !certain && ((v1 - v3) > thr) && (v3 < lim.hi) && (v1 > lim.lo) && i2 != 0xFFFFFFFFu;
  const bool mine = (h == 0) && (s < N);
  const bool commit_now = mine && certain;
  const bool pair_now = mine && two;
  const bool flag_now = mine && !certain && !two;
  bool changed = false;
  if (commit_now) changed = commit_row(s, insane ? K : i1, assignments, assignments_prev);
  const unsigned long long cm = __ballot(changed);
  const unsigned long long pm = __ballot(pair_now);
  const unsigned long long fm = __ballot(flag_now);
  if (lane == 0 && cm) atomicAdd(&counters[0], (uint32_t)__popcll(cm));
  if (pm) {
    uint32_t base = 0;
    if (lane == 0) base = atomicAdd(&counters[3], (uint32_t)__popcll(pm));
    base = __shfl(base, 0);
    if (pair_now) {
      const uint32_t slot = base + (uint32_t)__popcll(pm & ((1ull << lane) - 1ull));
      pairs[3 * (size_t)slot + 0] = s;
      pairs[3 * (size_t)slot + 1] = i1;
      pairs[3 * (size_t)slot + 2] = i2;
    }
  }
  if (fm) {
    uint32_t base = 0;
    if (lane == 0) base = atomicAdd(&counters[1], (uint32_t)__popcll(fm));
    base = __shfl(base, 0);
    if (flag_now) flagged[base + (uint32_t)__popcll(fm & ((1ull << lane) - 1ull))] = s;
  }
}

}  // namespace kmx
