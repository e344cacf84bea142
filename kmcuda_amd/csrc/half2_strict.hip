// half2_strict.hip -- the reference's fp16x2 ARITHMETIC on the GPU (KMCUDA_AMD_FP16_STRICT=1).
//
// The product's fp16x2 semantics are the fp32 reference arithmetic on the half values:
// fast, and within the reference's own fp16 noise, but not the reference's numbers.  This file is the
// verification mode that IS the reference's numbers: F = half2 throughout (src/fp_abstraction.h:100-182)
// -- every _add / _sub / _mul / _fma a packed binary16 operation rounded to nearest even (__hfma2 = one
// rounding), Kahan sums as TWO interleaved half accumulators (even / odd features) with their own
// compensation terms, _fin = hi + lo in half, _const<half2>(int) = __int2half_rd, _fmax = 65504, half
// compares -- restated kernel by kernel from src/kmeans.cu (assign :293-364, adjust :366-429, Yinyang
// :431-672, k-means++ :42-67, AFK-MC2 :69-183, average distance :674-691) and src/metric_abstraction.h.
// gfx950 executes binary16 add / mul / fma natively with IEEE rounding and denormals, so each operation
// below is one instruction with exactly the reference's result.  Deliberately plain -- one thread per
// sample (or per centroid), no matrix cores, no LDS: a parity mode, checked bit for bit against the CPU
// oracle's half2 restatement (oracle/kmcuda_oracle.c, pinned on the reference's five fp16 known answers),
// not a fast path.  Values travel as fp32 words that hold exactly representable halves (the engine's
// widened working copies), so the host orchestration is unchanged.
#include "exact.hpp"
#include "half2_ops.hpp"
#include "kernels.hpp"

namespace kmx {

// ---- Lloyd assignment, kmeans.cu:293-364 ----
template <int METRIC>
__global__ void h2_csqr_kernel(const float *__restrict__ centroids, uint32_t K, uint32_t D, float *__restrict__ sq2) {
  const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= K) return;
  hf s0 = 0, s1 = 0, c0 = 0, c1 = 0;
  if (METRIC == 0) {
    const float *v = centroids + (size_t)c * D;
    for (uint32_t f = 0; f + 1 < D; f += 2) {
      H2_KAHAN(s0, c0, h_ld(v, f), h_ld(v, f));
      H2_KAHAN(s1, c1, h_ld(v, f + 1), h_ld(v, f + 1));
    }
  } else {
    s0 = s1 = (hf)1.f;   // :149-158
  }
  sq2[2 * c] = (float)s0;
  sq2[2 * c + 1] = (float)s1;
}

template <int METRIC>
__global__ void h2_assign_kernel(const float *__restrict__ samples, uint32_t N, uint32_t D,
                                 const float *__restrict__ centroids, uint32_t K, const float *__restrict__ sq2,
                                 uint32_t *__restrict__ assignments, uint32_t *__restrict__ assignments_prev,
                                 uint32_t *__restrict__ counters) {
  const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
  bool changed = false;
  if (s < N) {
    const float *x = samples + (size_t)s * D;
    const bool insane = (x[0] != x[0]) || (x[1] != x[1]);   // _neq(half2, half2) = !__hbeq2
    hf min_dist = (hf)65504.f;                               // _fmax<half>()
    uint32_t nearest = 0xFFFFFFFFu;
    if (!insane)
      for (uint32_t c = 0; c < K; c++) {
        const float *cv = centroids + (size_t)c * D;
        hf p0 = 0, p1 = 0, c0 = 0, c1 = 0;
        for (uint32_t f = 0; f + 1 < D; f += 2) {
          H2_KAHAN(p0, c0, h_ld(x, f), h_ld(cv, f));
          H2_KAHAN(p1, c1, h_ld(x, f + 1), h_ld(cv, f + 1));
        }
        const hf dist = h2_distance3<METRIC>((hf)sq2[2 * c], (hf)sq2[2 * c + 1], p0, p1);
        if (dist < min_dist) { min_dist = dist; nearest = c; }   // __hlt
      }
    bool commit = true;
    if (nearest == 0xFFFFFFFFu) {   // kmeans.cu:349-357
      if (!insane) commit = false;
      else nearest = K;
    }
    if (commit) {
      const uint32_t ass = assignments[s];
      assignments_prev[s] = ass;
      if (ass != nearest) {
        assignments[s] = nearest;
        changed = true;
      }
    }
  }
  const unsigned long long m = __ballot(changed);
  if (m && (threadIdx.x & 63) == __ffsll((long long)m) - 1) atomicAdd(&counters[0], (uint32_t)__popcll(m));
}

// ---- centroid update, kmeans.cu:366-429 + normalize (metric_abstraction.h:138-144, :274-300) ----
template <int METRIC>
__global__ void h2_adjust_kernel(const float *__restrict__ samples, uint32_t N, uint32_t D, uint32_t K,
                                 const uint32_t *__restrict__ prev, const uint32_t *__restrict__ cur,
                                 float *__restrict__ centroids, uint32_t *__restrict__ ccounts) {
  const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= K) return;
  float *cen = centroids + (size_t)c * D;
  uint32_t my_count = ccounts[c];
  {
    const hf fmy = h_from_int_rd(my_count);
    for (uint32_t f = 0; f < D; f++) cen[f] = (float)(h_ld(cen, f) * fmy);
  }
  hf corr0 = 0, corr1 = 0;   // ONE half2 corr for all features and samples: one per lane
  for (uint32_t s = 0; s < N; s++) {
    const uint32_t p = prev[s], a = cur[s];
    hf fsign;
    if (p == c && a != c) { fsign = (hf)-1.f; my_count--; }
    else if (p != c && a == c) { fsign = (hf)1.f; my_count++; }
    else continue;
    const float *x = samples + (size_t)s * D;
    for (uint32_t f = 0; f + 1 < D; f += 2) {
      hf v = h_ld(cen, f);
      hf y = h_fma(h_ld(x, f), fsign, corr0), t = v + y;
      corr0 = y - (t - v);
      cen[f] = (float)t;
      v = h_ld(cen, f + 1);
      y = h_fma(h_ld(x, f + 1), fsign, corr1);
      t = v + y;
      corr1 = y - (t - v);
      cen[f + 1] = (float)t;
    }
  }
  if (METRIC == 0) {   // h2rcp(_const<half2>(count)): nearest half of the reciprocal
    const hf rc = (hf)(1.0f / (float)h_from_int_rd(my_count));
    for (uint32_t f = 0; f < D; f++) cen[f] = (float)(h_ld(cen, f) * rc);
  } else {             // fp32 norm, HIGH half of every pair first
    float norm = 0.f, ncorr = 0.f;
    for (uint32_t f = 0; f + 1 < D; f += 2) {
      kahan_fold(fma_rd(cen[f + 1], cen[f + 1], ncorr), norm, ncorr);
      kahan_fold(fma_rd(cen[f], cen[f], ncorr), norm, ncorr);
    }
    norm = 1.0f / sqrtf(norm);
    const hf norm2 = (hf)norm;
    for (uint32_t f = 0; f < D; f++) cen[f] = (float)(h_ld(cen, f) * norm2);
  }
  ccounts[c] = my_count;
}

// ---- distances to one row: k-means++ step (kmeans.cu:42-67) / AFK-MC2 q (:69-97) / members (:674-691) ----
// mode 0: dists[s] = min(dists[s], d) unless cc == 1 (k-means++; a NaN row counts as 0); 1: dists[s] = d * d;
template <int METRIC>
__global__ void h2_to_row_kernel(const float *__restrict__ samples, uint32_t N, uint32_t D,
                                 const float *__restrict__ row, uint32_t cc, int mode, float *__restrict__ dists) {
  const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= N) return;
  const float *x = samples + (size_t)s * D;
  if (mode == 1) {
    const float d = h2_distance<METRIC>(x, row, D);
    dists[s] = d * d;
    return;
  }
  float dist = 0.f;
  if (x[0] == x[0] && x[1] == x[1]) dist = h2_distance<METRIC>(x, row, D);
  if (cc == 1 || dist < dists[s]) dists[s] = dist;
}
template <int METRIC>
__global__ void h2_member_kernel(const float *__restrict__ samples, uint32_t N, uint32_t D,
                                 const float *__restrict__ centroids, const uint32_t *__restrict__ assignments,
                                 uint32_t K, float *__restrict__ dists) {
  const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= N) return;
  const uint32_t a = assignments[s];
  dists[s] = a < K ? h2_distance<METRIC>(samples + (size_t)s * D, centroids + (size_t)a * D, D) : 0.f;
}
template <int METRIC>
__global__ void h2_afk_min_dist_kernel(uint32_t m, uint32_t k, const float *__restrict__ samples, uint32_t D,
                                       const uint32_t *__restrict__ choices, const float *__restrict__ centroids,
                                       float *__restrict__ min_dists) {   // kmeans.cu:166-183
  const uint32_t chi = blockIdx.x * blockDim.x + threadIdx.x;
  if (chi >= m) return;
  float min_dist = 3.402823466e+38f;
  for (uint32_t c = 0; c < k; c++) {
    const float dist = h2_distance<METRIC>(samples + (size_t)choices[chi] * D, centroids + (size_t)c * D, D);
    if (dist < min_dist) min_dist = dist;
  }
  min_dists[chi] = min_dist * min_dist;
}

// ---- Yinyang, kmeans.cu:431-672 (bounds, drifts and their arithmetic are fp32; distances half2) ----
template <int METRIC>
__global__ void h2_yy_init_kernel(const float *__restrict__ samples, uint32_t len, uint32_t D, uint32_t K, uint32_t G,
                                  const float *__restrict__ centroids, const uint32_t *__restrict__ assignments,
                                  const uint32_t *__restrict__ groups, float *__restrict__ bounds) {
  const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= len) return;
  for (uint32_t i = 0; i < G + 1; i++) bounds[(size_t)len * i + s] = 3.402823466e+38f;
  const uint32_t nearest = assignments[s];
  for (uint32_t c = 0; c < K; c++) {
    const uint32_t group = groups[c];
    if (group >= G) continue;
    const float dist = h2_distance<METRIC>(samples + (size_t)s * D, centroids + (size_t)c * D, D);
    if (c != nearest) {
      const size_t gi = (size_t)len * (1 + group) + s;
      if (dist < bounds[gi]) bounds[gi] = dist;
    } else {
      bounds[s] = dist;
    }
  }
}
template <int METRIC>
__global__ void h2_yy_drifts_kernel(const float *__restrict__ centroids, uint32_t K, uint32_t D,
                                    float *__restrict__ drifts) {
  const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= K) return;
  drifts[(size_t)K * D + c] = h2_distance<METRIC>(centroids + (size_t)c * D, drifts + (size_t)c * D, D);
}
template <int METRIC>
__global__ void h2_yy_global_kernel(const float *__restrict__ samples, uint32_t len, uint32_t D, uint32_t K, uint32_t G,
                                    const float *__restrict__ centroids, const float *__restrict__ drifts,
                                    const float *__restrict__ gdrifts, const uint32_t *__restrict__ assignments,
                                    uint32_t *__restrict__ assignments_prev, float *__restrict__ bounds,
                                    uint32_t *__restrict__ passed, uint32_t *__restrict__ counters) {
  const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
  bool pass = false;
  if (s < len) {
    const uint32_t cluster = assignments[s];
    assignments_prev[s] = cluster;
    float upper_bound = bounds[s] + drifts[(size_t)K * D + cluster];
    float min_lower_bound = 3.402823466e+38f;
    for (uint32_t g = 0; g < G; g++) {
      const size_t gi = (size_t)len * (1 + g) + s;
      const float lower_bound = bounds[gi] - gdrifts[g];
      bounds[gi] = lower_bound;
      if (lower_bound < min_lower_bound) min_lower_bound = lower_bound;
    }
    if (min_lower_bound >= upper_bound) {
      bounds[s] = upper_bound;
    } else {
      upper_bound = h2_distance<METRIC>(samples + (size_t)s * D, centroids + (size_t)cluster * D, D);
      bounds[s] = upper_bound;
      pass = !(min_lower_bound >= upper_bound);
    }
  }
  const unsigned long long m = __ballot(pass);
  if (m) {
    const int lane = threadIdx.x & 63;
    uint32_t base = 0;
    if (lane == __ffsll((long long)m) - 1) base = atomicAdd(&counters[2], (uint32_t)__popcll(m));
    base = __shfl(base, __ffsll((long long)m) - 1);
    if (pass) passed[base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = s;
  }
}
template <int METRIC>
__global__ void h2_yy_local_kernel(const float *__restrict__ samples, uint32_t len, uint32_t D, uint32_t K, uint32_t G,
                                   const uint32_t *__restrict__ passed, const float *__restrict__ centroids,
                                   const uint32_t *__restrict__ groups, const float *__restrict__ drifts,
                                   const float *__restrict__ gdrifts, uint32_t *__restrict__ assignments,
                                   float *__restrict__ bounds, uint32_t *__restrict__ counters) {
  const uint32_t pi = blockIdx.x * blockDim.x + threadIdx.x;
  bool changed = false;
  if (pi < counters[2]) {
    const uint32_t s = passed[pi];
    const float *x = samples + (size_t)s * D;
    const float upper_bound = bounds[s];
    const uint32_t cluster = assignments[s];
    float min_dist = upper_bound, second_min_dist = 3.402823466e+38f;
    uint32_t nearest = cluster;
    for (uint32_t c = 0; c < K; c++) {
      if (c == cluster) continue;
      const uint32_t group = groups[c];
      if (group >= G) continue;
      float lower_bound = bounds[(size_t)len * (1 + group) + s];
      if (lower_bound >= upper_bound) {
        if (lower_bound < second_min_dist) second_min_dist = lower_bound;
        continue;
      }
      lower_bound += gdrifts[group] - drifts[(size_t)K * D + c];
      if (second_min_dist < lower_bound) continue;
      const float dist = h2_distance<METRIC>(x, centroids + (size_t)c * D, D);
      if (dist < min_dist) {
        second_min_dist = min_dist;
        min_dist = dist;
        nearest = c;
      } else if (dist < second_min_dist) {
        second_min_dist = dist;
      }
    }
    const uint32_t nearest_group = groups[nearest], previous_group = groups[cluster];
    bounds[(size_t)len * (1 + nearest_group) + s] = second_min_dist;
    if (nearest_group != previous_group) {
      const size_t gi = (size_t)len * (1 + previous_group) + s;
      if (bounds[gi] > upper_bound) bounds[gi] = upper_bound;
    }
    bounds[s] = min_dist;
    if (cluster != nearest) {
      assignments[s] = nearest;
      changed = true;
    }
  }
  const unsigned long long m = __ballot(changed);
  if (m && (threadIdx.x & 63) == __ffsll((long long)m) - 1) atomicAdd(&counters[0], (uint32_t)__popcll(m));
}

// ---------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------
#define H2_LAUNCH(metric, kernel, n, st, ...)                                                                  \
  do {                                                                                                         \
    if ((n) == 0) return hipSuccess;                                                                           \
    if ((metric) == 0) hipLaunchKernelGGL((kernel<0>), dim3(((n) + 127) / 128), dim3(128), 0, st, __VA_ARGS__); \
    else hipLaunchKernelGGL((kernel<1>), dim3(((n) + 127) / 128), dim3(128), 0, st, __VA_ARGS__);              \
    return hipGetLastError();                                                                                  \
  } while (0)

hipError_t launch_h2_csqr(int metric, const float *centroids, uint32_t K, uint32_t D, float *sq2, hipStream_t st) {
  H2_LAUNCH(metric, h2_csqr_kernel, K, st, centroids, K, D, sq2);
}
hipError_t launch_h2_assign(int metric, const float *samples, uint32_t N, uint32_t D, const float *centroids, uint32_t K,
                            const float *sq2, uint32_t *assignments, uint32_t *assignments_prev, uint32_t *counters,
                            hipStream_t st) {
  H2_LAUNCH(metric, h2_assign_kernel, N, st, samples, N, D, centroids, K, sq2, assignments, assignments_prev, counters);
}
hipError_t launch_h2_adjust(int metric, const float *samples, uint32_t N, uint32_t D, uint32_t K, const uint32_t *prev,
                            const uint32_t *cur, float *centroids, uint32_t *ccounts, hipStream_t st) {
  H2_LAUNCH(metric, h2_adjust_kernel, K, st, samples, N, D, K, prev, cur, centroids, ccounts);
}
hipError_t launch_h2_to_row(int metric, const float *samples, uint32_t N, uint32_t D, const float *row, uint32_t cc,
                            int mode, float *dists, hipStream_t st) {
  H2_LAUNCH(metric, h2_to_row_kernel, N, st, samples, N, D, row, cc, mode, dists);
}
hipError_t launch_h2_member(int metric, const float *samples, uint32_t N, uint32_t D, const float *centroids,
                            const uint32_t *assignments, uint32_t K, float *dists, hipStream_t st) {
  H2_LAUNCH(metric, h2_member_kernel, N, st, samples, N, D, centroids, assignments, K, dists);
}
hipError_t launch_h2_afk_min_dist(int metric, uint32_t m, uint32_t k, const float *samples, uint32_t D,
                                  const uint32_t *choices, const float *centroids, float *min_dists, hipStream_t st) {
  H2_LAUNCH(metric, h2_afk_min_dist_kernel, m, st, m, k, samples, D, choices, centroids, min_dists);
}
hipError_t launch_h2_yy_init(int metric, const float *samples, uint32_t len, uint32_t D, uint32_t K, uint32_t G,
                             const float *centroids, const uint32_t *assignments, const uint32_t *groups, float *bounds,
                             hipStream_t st) {
  H2_LAUNCH(metric, h2_yy_init_kernel, len, st, samples, len, D, K, G, centroids, assignments, groups, bounds);
}
hipError_t launch_h2_yy_drifts(int metric, const float *centroids, uint32_t K, uint32_t D, float *drifts, hipStream_t st) {
  H2_LAUNCH(metric, h2_yy_drifts_kernel, K, st, centroids, K, D, drifts);
}
hipError_t launch_h2_yy_global(int metric, const float *samples, uint32_t len, uint32_t D, uint32_t K, uint32_t G,
                               const float *centroids, const float *drifts, const float *gdrifts,
                               const uint32_t *assignments, uint32_t *assignments_prev, float *bounds, uint32_t *passed,
                               uint32_t *counters, hipStream_t st) {
  H2_LAUNCH(metric, h2_yy_global_kernel, len, st, samples, len, D, K, G, centroids, drifts, gdrifts, assignments,
            assignments_prev, bounds, passed, counters);
}
hipError_t launch_h2_yy_local(int metric, const float *samples, uint32_t len, uint32_t D, uint32_t K, uint32_t G,
                              const uint32_t *passed, const float *centroids, const uint32_t *groups, const float *drifts,
                              const float *gdrifts, uint32_t *assignments, float *bounds, uint32_t *counters,
                              hipStream_t st) {
  H2_LAUNCH(metric, h2_yy_local_kernel, len, st, samples, len, D, K, G, passed, centroids, groups, drifts, gdrifts,
            assignments, bounds, counters);
}

}  // namespace kmx
