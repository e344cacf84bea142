// yinyang.hip -- the Yinyang bound kernels (reference: src/kmeans.cu:431-672).
//
// These kernels are parity-critical in a different way from the Lloyd pass: the bounds they keep
// are floats produced by the reference's exact arithmetic (distance_t = sqrt of a Kahan /
// round-down-FMA sum of squared differences, metric_abstraction.h:73-86) and every later
// pruning decision compares against them, so all distances here use exact.hpp and the
// reference's evaluation order where order matters (local filter).  Layouts:
//   xt      feature-major copy of this shard's samples, xt[f*len + s]  (one transpose per call,
//           transpose.hip) -- thread-per-sample kernels read it fully coalesced
//   bounds  bounds[s] = upper bound, bounds[(1+g)*len + s] = lower bound to group g  (same
//           group-major layout as the reference: the streaming global filter is coalesced)
#include "exact.hpp"
#include "kernels.hpp"

namespace kmx {

// distance_t for a feature-major sample against a contiguous vector (metric_abstraction.h:73-86, :193-205)
template <int METRIC>
__device__ __forceinline__ float distance_t(const float *__restrict__ xt, size_t len, size_t s,
                                            const float *__restrict__ c, uint32_t D) {
  float acc = 0.f, corr = 0.f;
  if (METRIC == 0) {
    for (uint32_t f = 0; f < D; f++) {
      const float d = xt[(size_t)f * len + s] - c[f];
      kahan_fold(fma_rd(d, d, corr), acc, corr);
    }
    return sqrtf(acc);
  }
  for (uint32_t f = 0; f < D; f++) kahan_fold(fma_rd(xt[(size_t)f * len + s], c[f], corr), acc, corr);
  return angular_from_prod(acc);
}

// ---------------------------------------------------------------------------------------
// yy_init (kmeans.cu:431-485): bounds refresh.  Centroids are visited in GROUP-SORTED order
// (cperm, host-built) so the per-group minimum lives in a register and each bound is written
// once; a minimum does not depend on visiting order.  Four centroids per pass share every
// sample load and one rounding-mode window.
// ---------------------------------------------------------------------------------------
template <int METRIC>
__global__ __launch_bounds__(128) void yy_init_kernel(
    const float *__restrict__ xt, uint32_t len, uint32_t D, uint32_t G, const float *__restrict__ centroids,
    const uint32_t *__restrict__ assignments, const uint32_t *__restrict__ cperm,
    const uint32_t *__restrict__ gstart /* G+1 offsets into cperm */, float *__restrict__ bounds) {
  const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= len) return;
  const uint32_t nearest = assignments[s];
  float upper = 3.402823466e+38f;
  for (uint32_t g = 0; g < G; g++) {
    float gmin = 3.402823466e+38f;
    const uint32_t b = gstart[g], e = gstart[g + 1];
    for (uint32_t i = b; i < e; i += 4) {
      uint32_t cid[4];
      const float *cp[4];
#pragma unroll
      for (int j = 0; j < 4; j++) {
        cid[j] = cperm[(i + j < e) ? i + j : b];
        cp[j] = centroids + (size_t)cid[j] * D;
      }
      float acc[4] = {0.f, 0.f, 0.f, 0.f}, corr[4] = {0.f, 0.f, 0.f, 0.f};
      for (uint32_t f = 0; f < D; f++) {
        const float x = xt[(size_t)f * len + s];
        float y[4];
        if (METRIC == 0) {
          float d[4];
#pragma unroll
          for (int j = 0; j < 4; j++) d[j] = x - cp[j][f];
          sqfma_rd4(d, corr, y);
        } else {
          float cv[4];
#pragma unroll
          for (int j = 0; j < 4; j++) cv[j] = cp[j][f];
          fma_rd4(x, cv, corr, y);
        }
#pragma unroll
        for (int j = 0; j < 4; j++) kahan_fold(y[j], acc[j], corr[j]);
      }
#pragma unroll
      for (int j = 0; j < 4; j++) {
        if (i + j >= e) continue;
        const float dist = (METRIC == 0) ? sqrtf(acc[j]) : angular_from_prod(acc[j]);
        if (cid[j] != nearest) {
          if (dist < gmin) gmin = dist;
        } else {
          upper = dist;
        }
      }
    }
    bounds[(size_t)len * (1 + g) + s] = gmin;
  }
  bounds[s] = upper;
}

// kmeans.cu:487-499
template <int METRIC>
__global__ void yy_calc_drifts_kernel(const float *__restrict__ centroids, uint32_t K, uint32_t D,
                                      float *__restrict__ drifts) {
  const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= K) return;
  drifts[(size_t)K * D + c] = distance_vv<METRIC>(centroids + (size_t)c * D, drifts + (size_t)c * D, D);
}

// kmeans.cu:501-538 (writes into drifts[0..G), overlaying the old-centroid copy, :537)
__global__ void yy_group_max_drifts_kernel(const uint32_t *__restrict__ groups, uint32_t K, uint32_t D, uint32_t G,
                                           const float *__restrict__ drifts, float *__restrict__ gmax) {
  const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= G) return;
  float my_max = -3.402823466e+38f;
  for (uint32_t c = 0; c < K; c++) {
    if (groups[c] == g) {
      const float d = drifts[(size_t)K * D + c];
      if (my_max < d) my_max = d;
    }
  }
  gmax[g] = my_max;
}

// ---------------------------------------------------------------------------------------
// yy_global_filter (kmeans.cu:540-582): streams the bounds matrix; rows that survive both
// group-filter tries are appended to `passed` (order irrelevant: each is handled independently)
// ---------------------------------------------------------------------------------------
// Round 2: the tightening distance (row to its own centroid, needed by every row that fails try #1 --
// all of them on unstructured data) used to be walked by each thread along its own row and its own
// centroid row: every 16-byte load of a wave touched 64 different lines.  Now both come in through LDS
// in 32-feature chunks, 8 lanes per 128-byte line (the staging of kmpp_step2_kernel, seeding.hip), and
// the chain -- same operations, same order -- reads its operands from two 36-float-stride tiles.
template <int METRIC>
__global__ __launch_bounds__(256) void yy_global_filter_kernel(
    const float *__restrict__ samples, uint32_t len, uint32_t D, uint32_t K, uint32_t G,
    const float *__restrict__ centroids, const float *__restrict__ drifts, const float *__restrict__ gdrifts,
    const uint32_t *__restrict__ assignments, uint32_t *__restrict__ assignments_prev, float *__restrict__ bounds,
    uint32_t *__restrict__ passed, uint32_t *__restrict__ counters) {
  extern __shared__ __attribute__((aligned(16))) float gf_lds[];   // 2 x 256 x 36 floats + 256 cluster ids
  const uint32_t s = blockIdx.x * 256u + threadIdx.x;
  bool pass = false, need = false;
  float upper_bound = 0.f, min_lower_bound = 3.402823466e+38f;
  uint32_t cluster = 0;
  if (s < len) {
    cluster = assignments[s];
    assignments_prev[s] = cluster;
    upper_bound = bounds[s];
    const float cluster_drift = drifts[(size_t)K * D + cluster];
    upper_bound += cluster_drift;
    uint32_t g = 0;
    for (; g + 8 <= G; g += 8) {   // eight independent loads in flight per thread
      float lb[8];
#pragma unroll
      for (int q = 0; q < 8; q++) lb[q] = bounds[(size_t)len * (1 + g + q) + s];
#pragma unroll
      for (int q = 0; q < 8; q++) {
        const float lower_bound = lb[q] - gdrifts[g + q];
        bounds[(size_t)len * (1 + g + q) + s] = lower_bound;
        if (lower_bound < min_lower_bound) min_lower_bound = lower_bound;
      }
    }
    for (; g < G; g++) {
      const size_t gi = (size_t)len * (1 + g) + s;
      const float lower_bound = bounds[gi] - gdrifts[g];
      bounds[gi] = lower_bound;
      if (lower_bound < min_lower_bound) min_lower_bound = lower_bound;
    }
    if (min_lower_bound >= upper_bound) bounds[s] = upper_bound;  // group filter try #1
    else need = true;
  }
  if (__syncthreads_or(need ? 1 : 0)) {
    const bool staged = (D & 3u) == 0 && (((uintptr_t)samples | (uintptr_t)centroids) & 15u) == 0;
    if (staged) {
      float *tx = gf_lds, *tc = gf_lds + 256 * 36;
      uint32_t *cid = reinterpret_cast<uint32_t *>(gf_lds + 2 * 256 * 36);
      cid[threadIdx.x] = (s < len && cluster < K) ? cluster : 0xFFFFFFFFu;
      __syncthreads();
      const uint32_t nchunk = (D + 31) / 32;
      float4 sx[8], sc[8];
      auto fetch = [&](uint32_t ch) {
#pragma unroll
        for (int q = 0; q < 8; q++) {
          const uint32_t p = threadIdx.x + 256u * q, r = p >> 3, c4 = p & 7u;
          const uint32_t row = blockIdx.x * 256u + r, f = ch * 32 + c4 * 4;
          const uint32_t c = cid[r];
          const bool on = row < len && f < D;
          sx[q] = on ? *reinterpret_cast<const float4 *>(samples + (size_t)row * D + f) : make_float4(0.f, 0.f, 0.f, 0.f);
          sc[q] = (on && c != 0xFFFFFFFFu) ? *reinterpret_cast<const float4 *>(centroids + (size_t)c * D + f)
                                           : make_float4(0.f, 0.f, 0.f, 0.f);
        }
      };
      float acc = 0.f, corr = 0.f;
      fetch(0);
      for (uint32_t ch = 0; ch < nchunk; ch++) {
        __syncthreads();   // the previous chunk has been consumed
#pragma unroll
        for (int q = 0; q < 8; q++) {
          const uint32_t p = threadIdx.x + 256u * q, r = p >> 3, c4 = p & 7u;
          *reinterpret_cast<float4 *>(&tx[r * 36 + c4 * 4]) = sx[q];
          *reinterpret_cast<float4 *>(&tc[r * 36 + c4 * 4]) = sc[q];
        }
        __syncthreads();
        if (ch + 1 < nchunk) fetch(ch + 1);
        const uint32_t fmax = D - ch * 32 < 32u ? D - ch * 32 : 32u;   // multiple of 4
        for (uint32_t c4 = 0; c4 * 4 < fmax; c4++) {
          const float4 xv = *reinterpret_cast<const float4 *>(&tx[threadIdx.x * 36 + c4 * 4]);
          const float4 cv = *reinterpret_cast<const float4 *>(&tc[threadIdx.x * 36 + c4 * 4]);
          const float aa[4] = {xv.x, xv.y, xv.z, xv.w}, bb[4] = {cv.x, cv.y, cv.z, cv.w};
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
      if (need) upper_bound = METRIC == 0 ? sqrtf(acc) : angular_from_prod(acc);
    } else if (need) {
      upper_bound = distance_vv<METRIC>(samples + (size_t)s * D, centroids + (size_t)cluster * D, D);
    }
    if (need) {
      bounds[s] = upper_bound;
      pass = !(min_lower_bound >= upper_bound);  // try #2
    }
  }
  // survivors appended per BLOCK: one list cursor bump for the four waves (the cursor's cache line serves
  // same-address atomics one at a time; 125 K waves per launch each wanting their own base was measurable)
  const unsigned long long m = __ballot(pass);
  __shared__ uint32_t blk_cnt[4], blk_base;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  if (lane == 0) blk_cnt[wv] = (uint32_t)__popcll(m);
  __syncthreads();
  if (threadIdx.x == 0) {
    const uint32_t tot = blk_cnt[0] + blk_cnt[1] + blk_cnt[2] + blk_cnt[3];
    blk_base = tot ? atomicAdd(&counters[2], tot) : 0u;
  }
  __syncthreads();
  if (pass) {
    uint32_t base = blk_base;
    for (int w = 0; w < wv; w++) base += blk_cnt[w];
    passed[base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = s;
  }
}

// ---------------------------------------------------------------------------------------
// yy_local_filter (kmeans.cu:584-672).  The centroid scan is sequential per sample on purpose:
// which distances get evaluated depends on the running second_min_dist, and the bounds written
// back depend on exactly that set, so the reference's order is kept.  Row-major samples: each
// thread walks its own (gathered) row.
// ---------------------------------------------------------------------------------------
template <int METRIC>
__global__ __launch_bounds__(128) void yy_local_filter_kernel(
    const float *__restrict__ samples, uint32_t len, uint32_t D, uint32_t K, uint32_t G,
    const uint32_t *__restrict__ passed, const float *__restrict__ centroids, const uint32_t *__restrict__ groups,
    const float *__restrict__ drifts, const float *__restrict__ gdrifts, uint32_t *__restrict__ assignments,
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
      if (group >= G) continue;  // NaN centroid
      float lower_bound = bounds[(size_t)len * (1 + group) + s];
      if (lower_bound >= upper_bound) {
        if (lower_bound < second_min_dist) second_min_dist = lower_bound;
        continue;
      }
      lower_bound += gdrifts[group] - drifts[(size_t)K * D + c];
      if (second_min_dist < lower_bound) continue;
      const float dist = distance_vv<METRIC>(x, centroids + (size_t)c * D, D);
      if (dist < min_dist) {
        second_min_dist = min_dist;
        min_dist = dist;
        nearest = c;
      } else if (dist < second_min_dist) {
        second_min_dist = dist;
      }
    }
    const uint32_t nearest_group = groups[nearest];
    const uint32_t previous_group = groups[cluster];
    bounds[(size_t)len * (1 + nearest_group) + s] = second_min_dist;
    if (nearest_group != previous_group) {
      const size_t gi = (size_t)len * (1 + previous_group) + s;
      const float pb = bounds[gi];
      if (pb > upper_bound) bounds[gi] = upper_bound;
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
#define KMX_DISPATCH(metric, kernel, grid, block, st, ...)                                    \
  do {                                                                                        \
    if ((metric) == 0) hipLaunchKernelGGL((kernel<0>), grid, block, 0, st, __VA_ARGS__);      \
    else hipLaunchKernelGGL((kernel<1>), grid, block, 0, st, __VA_ARGS__);                    \
  } while (0)

hipError_t launch_yy_init(int metric, const float *xt, uint32_t len, uint32_t D, uint32_t G, const float *centroids,
                          const uint32_t *assignments, const uint32_t *cperm, const uint32_t *gstart, float *bounds,
                          hipStream_t st) {
  if (len == 0) return hipSuccess;
  KMX_DISPATCH(metric, yy_init_kernel, dim3((len + 127) / 128), dim3(128), st, xt, len, D, G, centroids, assignments,
               cperm, gstart, bounds);
  return hipGetLastError();
}

hipError_t launch_yy_group_max(uint32_t K, uint32_t D, uint32_t G, const uint32_t *groups, const float *drifts,
                               float *gdrifts, hipStream_t st) {
  hipLaunchKernelGGL(yy_group_max_drifts_kernel, dim3((G + 63) / 64), dim3(64), 0, st, groups, K, D, G, drifts,
                     gdrifts);
  return hipGetLastError();
}

hipError_t launch_yy_drifts(int metric, const float *centroids, uint32_t K, uint32_t D, uint32_t G,
                            const uint32_t *groups, float *drifts, float *gdrifts, hipStream_t st) {
  KMX_DISPATCH(metric, yy_calc_drifts_kernel, dim3((K + 63) / 64), dim3(64), st, centroids, K, D, drifts);
  return launch_yy_group_max(K, D, G, groups, drifts, gdrifts, st);
}

hipError_t launch_yy_global_filter(int metric, const float *samples, uint32_t len, uint32_t D, uint32_t K, uint32_t G,
                                   const float *centroids, const float *drifts, const float *gdrifts,
                                   const uint32_t *assignments, uint32_t *assignments_prev, float *bounds,
                                   uint32_t *passed, uint32_t *counters, hipStream_t st) {
  if (len == 0) return hipSuccess;
  const size_t lds = (2 * 256 * 36 + 256) * sizeof(float);
  static bool attr_set = false;   // > 64 KB of dynamic LDS
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void *)yy_global_filter_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipFuncSetAttribute((const void *)yy_global_filter_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set = true;
  }
  if (metric == 0)
    hipLaunchKernelGGL((yy_global_filter_kernel<0>), dim3((len + 255) / 256), dim3(256), lds, st, samples, len, D, K, G,
                       centroids, drifts, gdrifts, assignments, assignments_prev, bounds, passed, counters);
  else
    hipLaunchKernelGGL((yy_global_filter_kernel<1>), dim3((len + 255) / 256), dim3(256), lds, st, samples, len, D, K, G,
                       centroids, drifts, gdrifts, assignments, assignments_prev, bounds, passed, counters);
  return hipGetLastError();
}

hipError_t launch_yy_local_filter(int metric, const float *samples, uint32_t len, uint32_t D, uint32_t K, uint32_t G,
                                  const float *centroids, const uint32_t *groups, const float *drifts,
                                  const float *gdrifts, uint32_t *assignments, float *bounds, const uint32_t *passed,
                                  uint32_t *counters, hipStream_t st) {
  if (len == 0) return hipSuccess;
  // the passed count lives on the device: launch for the worst case, surplus threads exit at once
  KMX_DISPATCH(metric, yy_local_filter_kernel, dim3((len + 127) / 128), dim3(128), st, samples, len, D, K, G, passed,
               centroids, groups, drifts, gdrifts, assignments, bounds, counters);
  return hipGetLastError();
}

}  // namespace kmx
