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


// codes: tile*16 + accumulator register; a lane's register r of tile t is centroid
// t*32 + (r&3) + 8*(r>>2) + 4*half.
//   v1 - v2 > thr : the reference's distance to i1 is strictly the smallest -> commit
//   v1 - v3 > thr : the minimum is i1 or i2 -> two exact Kahan distances settle it (pair list)
//   otherwise     : three or more contenders -> full exact scan (flagged list)
// thr = 2E with |score_filter - score_ref| <= E for every centroid up to a term constant in c
// (DESIGN.md 4.1); a NaN gap or NaN thr is "not certain".
__device__ __forceinline__ void filter_finish(float v1, float v2, float v3, uint32_t c1, uint32_t c2, int h, int lane,
                                              uint32_t s, uint32_t N, uint32_t K, bool insane, float thr,
                                              uint32_t *__restrict__ assignments,
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
  const bool certain = insane || ((v1 - v2) > thr);
  const bool two = !certain && ((v1 - v3) > thr) && i2 != 0xFFFFFFFFu;
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
