// yinyang_mfma.hip -- the two distance-heavy Yinyang steps with the matrix cores in front of the
// reference's exact arithmetic (reference: src/kmeans.cu:431-485 kmeans_yy_init, :584-672
// kmeans_yy_local_filter).  yinyang.hip holds the same steps as plain exact kernels (the
// in-library cross-check, and the path for feature counts the filter is not instantiated for).
//
// Both kernels reuse the Lloyd filter's machinery (lloyd.hip): a wave keeps 32 rows, CENTRED
// (x' = x - mu), resident in VGPRs as the MFMA B operand and 32-centroid tiles of the centred
// panel (c' = c - mu) stream through LDS shared by the block's 4 waves.  With the accumulator
// seeded by the panel's bias:
//     ||x - c||^2 = ||x'||^2 - 2*acc          (L2)          x.c = acc + x.mu   (angular)
// and |acc - exact| <= 2 eps (||x'|| C'max + B'max).  The exact chains read the ORIGINAL row and
// centroid values from global memory (L1/L2 resident) in rolled loops.
// The approximate values only decide WHICH exact distances need evaluating; every number that is
// stored (bounds) or compared (min / second-min updates, skip tests) is the reference's exact
// arithmetic, evaluated in the reference's order where order matters.  Outputs are bit-identical
// to yinyang.hip's.
//
// yy_local_filter (kmeans.cu:584-672).  Per passed row the reference scans c = 0..K-1:
//     group bound >= upper bound      -> second_min = min(second_min, bound); skip          (a)
//     second_min < bound + drifts     -> skip                                                 (b)
//     else dist = exact; update (min, second_min, nearest) with strict '<'                    (c)
//   A centroid whose exact distance is >= second_min at its turn changes nothing whether it is
//   evaluated or skipped.  So (a) is replayed for every centroid (folded into a running minimum
//   between candidates: min is order free), and (b)/(c) only for centroids whose approximate
//   distance could be below second_min (threshold as of the last flush = a superset, second_min
//   only decreases).  Candidates are QUEUED per row, four deep; a flush evaluates the queued exact
//   distances as four interleaved chains (a distance has no side effects, so evaluating one the
//   reference would have skipped is harmless) and then replays the reference's tests and updates
//   in ascending c order with the live state -- the state evolves exactly as in the reference.
//
// yy_init (kmeans.cu:431-485): yinyang_init.hip.
#include "yinyang_tiles.hpp"

namespace kmx {

// ---------------------------------------------------------------------------------------
// yy_local_filter with the MFMA filter
// ---------------------------------------------------------------------------------------
template <int DP, int METRIC, bool FAST>
__global__ __launch_bounds__(256, 2) void yy_local_mfma_kernel(YyArgs a) {
  constexpr int NK = DP / 2, LDW = DP + 4, TILE = 32 * LDW, NST = (8 * DP + 255) / 256;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  auto tile_ptr = [&](int buf) { return lds + buf * TILE; };
  auto bias_ptr = [&](int buf) { return lds + 2 * TILE + buf * 32; };
  auto grp_ptr = [&](int buf) { return reinterpret_cast<uint32_t *>(lds + 2 * TILE + 64) + buf * 32; };

  const uint32_t npassed = *a.count_ptr;
  if (blockIdx.x * 128u >= npassed) return;  // block-uniform
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, col = lane & 31, h = lane >> 5;
  const uint32_t D = a.D, K = a.K, G = a.G, len = a.len;
  const uint32_t pi = blockIdx.x * 128u + wave * 32u + col;
  const bool live = pi < npassed;
  const uint32_t s = live ? a.passed[pi] : 0u;

  KMX_YY_LOAD_ROWS(a.samples, s, live)

  const float upper_bound = live ? a.bounds[s] : 0.f;
  const uint32_t cluster = live ? a.assignments[s] : 0xFFFFFFFFu;
  float min_dist = upper_bound, second_min = kFltMax;
  uint32_t nearest = cluster;

  // threshold in accumulator space: a centroid can only matter if acc >= amin
  const float cmaxc = sqrtf(__uint_as_float(a.stats[0])) * 1.000001f;
  const float bmaxc = __uint_as_float(a.stats[1]);
  const float xo = sqrtf(xo2) * 1.0001f, xc = sqrtf(xc2) * 1.0001f;
  const float e_mfma = 2.0f * a.eps * (xc * cmaxc + bmaxc) * 1.01f;
  // angular: x.c = acc + x.mu with x.mu evaluated in fp32 here
  const float e_cos = e_mfma + a.eps * xo * sqrtf(__uint_as_float(a.stats[3])) * 1.01f + 1e-6f;
  auto amin_of = [&](float sm) -> float {
    if (METRIC == 0) {
      const float T2 = sm * sm * 1.000002f;  // inf while second_min is still FLT_MAX
      return 0.5f * (xc2 - T2) - e_mfma - 1e-6f * (xc2 + T2);
    }
    if (sm >= 3.1415925f) return -INFINITY;
    return cosf(sm) - xmu - e_cos;
  };
  float amin = amin_of(second_min);

  f32x4 stage[NST];
  float bstage = 0.f;
  uint32_t gstage = 0;
  auto stage_load = [&](uint32_t tile) {
    const float *src = a.cfil + (size_t)tile * 32 * DP;
#pragma unroll
    for (int i = 0; i < NST; i++) {
      const int q = tid + i * 256;
      if (q < 8 * DP) stage[i] = reinterpret_cast<const f32x4 *>(src)[q];
    }
    if (tid < 32) {
      const uint32_t c = tile * 32 + tid;
      bstage = a.bias[c];
      gstage = c < K ? a.groups[c] : 0xFFFFFFFFu;
    }
  };
  auto stage_store = [&](int buf) {
#pragma unroll
    for (int i = 0; i < NST; i++) {
      const int q = tid + i * 256;
      if (q < 8 * DP) {
        const int row = q / (DP / 4), c4 = q % (DP / 4);
        *reinterpret_cast<f32x4 *>(tile_ptr(buf) + row * LDW + c4 * 4) = stage[i];
      }
    }
    if (tid < 32) {
      bias_ptr(buf)[tid] = bstage;
      grp_ptr(buf)[tid] = gstage;
    }
  };

  // queue of candidates (ascending c) + the minimum of the (a) bounds seen before each of them
  uint32_t qc[4] = {0, 0, 0, 0};
  float qpre[4] = {kFltMax, kFltMax, kFltMax, kFltMax};
  float tail_a = kFltMax;
  int qn = 0;
  auto flush = [&]() {  // wave-uniform call
    const float *crow[4];
#pragma unroll
    for (int i = 0; i < 4; i++) crow[i] = a.centroids + (size_t)(i < qn ? qc[i] : 0) * D;
    float dist[4];
    exact_distance4<NK, METRIC, FAST>(xrow, crow, D, h, col, dist, 4, qn);
#pragma unroll
    for (int i = 0; i < 4; i++) {
      if (i < qn) {
        if (qpre[i] < second_min) second_min = qpre[i];      // the (a) updates that preceded it
        const uint32_t c = qc[i];
        const uint32_t g = a.groups[c];
        float lb = a.bounds[(size_t)len * (1 + g) + s];
        lb += a.gdrifts[g] - a.drifts[(size_t)K * D + c];    // kmeans.cu:637
        if (!(second_min < lb)) {                            // :638-640
          const float d = dist[i];                           // :641-652
          if (d < min_dist) {
            second_min = min_dist;
            min_dist = d;
            nearest = c;
          } else if (d < second_min) {
            second_min = d;
          }
        }
      }
      qpre[i] = kFltMax;
    }
    if (tail_a < second_min) second_min = tail_a;
    tail_a = kFltMax;
    qn = 0;
    amin = amin_of(second_min);
  };

  const uint32_t ntiles = a.K_pad / 32;
  const bool wave_live = __ballot(live) != 0ull;
  stage_load(0);
  stage_store(0);
  __syncthreads();
  for (uint32_t t = 0; t < ntiles; t++) {
    const int buf = t & 1;
    if (t + 1 < ntiles) stage_load(t + 1);
    if (wave_live) {
      KMX_YY_MFMA_TILE(acc, buf)
      // (b)/(c) candidates by the filter, (a) group-skipped centroids by their bound
      uint32_t m16 = 0, a16 = 0;
      if (live) {
#pragma unroll
        for (int r = 0; r < 16; r++) {
          const uint32_t row = (r & 3) + 8 * (r >> 2) + 4 * h;
          const uint32_t c = t * 32 + row;
          const uint32_t g = grp_ptr(buf)[row];
          const bool valid = g < G && c != cluster;  // g >= G: NaN centroid or padding
          if (valid) {
            const float lb = a.bounds[(size_t)len * (1 + g) + s];
            if (lb >= upper_bound) a16 |= 1u << r;       // kmeans.cu:631-636
            else if (acc[r] >= amin) m16 |= 1u << r;
          }
        }
      }
      if (__ballot((m16 | a16) != 0u) != 0ull) {
        const uint32_t pm = __shfl_xor(m16, 32), pa = __shfl_xor(a16, 32);
        const uint32_t m0 = h ? pm : m16, m1 = h ? m16 : pm, a0 = h ? pa : a16, a1 = h ? a16 : pa;
        uint32_t bmask = 0, amask = 0;
#pragma unroll
        for (int g4 = 0; g4 < 4; g4++) {
          bmask |= (((m0 >> (4 * g4)) & 0xFu) << (8 * g4)) | (((m1 >> (4 * g4)) & 0xFu) << (8 * g4 + 4));
          amask |= (((a0 >> (4 * g4)) & 0xFu) << (8 * g4)) | (((a1 >> (4 * g4)) & 0xFu) << (8 * g4 + 4));
        }
        uint32_t rowmask = bmask | amask;
        while (__ballot(rowmask != 0u) != 0ull) {
          if (__ballot(qn == 4) != 0ull) flush();  // some row's queue is full
          const bool active = rowmask != 0u;
          const uint32_t rho = active ? (uint32_t)__ffs((int)rowmask) - 1u : 0u;
          rowmask &= rowmask - 1u;
          if (active) {
            if ((amask >> rho) & 1u) {
              const uint32_t g = grp_ptr(buf)[rho];
              const float lb = a.bounds[(size_t)len * (1 + g) + s];
              if (lb < tail_a) tail_a = lb;
            } else {
              const uint32_t c = t * 32 + rho;
#pragma unroll
              for (int i = 0; i < 4; i++)
                if (i == qn) {
                  qc[i] = c;
                  qpre[i] = tail_a;
                }
              tail_a = kFltMax;
              qn++;
              }
          }
        }
      }
    }
    if (t + 1 < ntiles) stage_store(buf ^ 1);
    __syncthreads();
  }
  if (wave_live && __ballot(qn > 0 || tail_a < kFltMax) != 0ull) flush();
  // write-back, kmeans.cu:653-671
  bool changed = false;
  if (live && h == 0) {
    const uint32_t nearest_group = a.groups[nearest], previous_group = a.groups[cluster];
    a.bounds[(size_t)len * (1 + nearest_group) + s] = second_min;
    if (nearest_group != previous_group) {
      const size_t gi = (size_t)len * (1 + previous_group) + s;
      const float pb = a.bounds[gi];
      if (pb > upper_bound) a.bounds[gi] = upper_bound;
    }
    a.bounds[s] = min_dist;
    if (cluster != nearest) {
      a.assignments[s] = nearest;
      changed = true;
    }
  }
  const unsigned long long cm = __ballot(changed);
  if (lane == 0 && cm) atomicAdd(&a.counters[0], (uint32_t)__popcll(cm));
}

// ---------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------
template <int DP, int METRIC>
static hipError_t launch_local_t(const YyArgs &a, hipStream_t st) {
  const size_t lds_bytes = (2 * 32 * (DP + 4) + 64 + 64) * sizeof(float);
  const uint32_t grid = (a.len + 127) / 128;  // worst case; blocks beyond the passed count exit at once
  if (a.D == (uint32_t)DP)
    hipLaunchKernelGGL((yy_local_mfma_kernel<DP, METRIC, true>), dim3(grid), dim3(256), lds_bytes, st, a);
  else
    hipLaunchKernelGGL((yy_local_mfma_kernel<DP, METRIC, false>), dim3(grid), dim3(256), lds_bytes, st, a);
  return hipGetLastError();
}
#define KMX_YY_SWITCH(fn)                                                              \
  switch (a.DP) {                                                                      \
    case 8: return metric == 0 ? fn<8, 0>(a, st) : fn<8, 1>(a, st);                    \
    case 16: return metric == 0 ? fn<16, 0>(a, st) : fn<16, 1>(a, st);                 \
    case 32: return metric == 0 ? fn<32, 0>(a, st) : fn<32, 1>(a, st);                 \
    case 64: return metric == 0 ? fn<64, 0>(a, st) : fn<64, 1>(a, st);                 \
    case 128: return metric == 0 ? fn<128, 0>(a, st) : fn<128, 1>(a, st);              \
    case 256: return metric == 0 ? fn<256, 0>(a, st) : fn<256, 1>(a, st);              \
    default: return hipErrorInvalidValue;                                              \
  }

hipError_t launch_yy_local_mfma(int metric, const YyArgs &a, hipStream_t st) {
  if (a.len == 0) return hipSuccess;
  KMX_YY_SWITCH(launch_local_t)
}

}  // namespace kmx
