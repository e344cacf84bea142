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
// yy_init (kmeans.cu:431-485).  bounds[1+g] = min over the group's centroids (other than the
//   row's own) of the exact distance: a minimum does not depend on the visiting order, so the panel
//   is streamed GROUP-SORTED (groups padded to multiples of 4 rows = one half-wave's accumulator
//   quad), each half-wave keeps a running top-3 of the approximate scores of the current group, and
//   at the group boundary the 1-2 contenders are queued (all members, evaluated at once, when
//   three or more are within the error bound); queued distances are evaluated four at a time.
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
  uint32_t n_flush = 0, n_cand = 0;  // statistics: counters[1] += flushes (per wave), [3] += candidates
  auto flush = [&]() {  // wave-uniform call
    n_flush++;
    const float *crow[4];
#pragma unroll
    for (int i = 0; i < 4; i++) crow[i] = a.centroids + (size_t)(i < qn ? qc[i] : 0) * D;
    float dist[4];
    exact_distance4<NK, METRIC, FAST>(xrow, crow, D, h, col, dist);
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
              n_cand++;
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
  {  // statistics (not part of the reference's state)
    uint32_t nc = (live && h == 0) ? n_cand : 0u;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) nc += __shfl_xor(nc, off);
    if (lane == 0) {
      atomicAdd(&a.counters[3], nc);
      atomicAdd(&a.counters[1], n_flush);
    }
  }
}

// ---------------------------------------------------------------------------------------
// yy_init with the MFMA filter: group-sorted panel, groups padded to multiples of 4 slots
// ---------------------------------------------------------------------------------------
// pids[slot]  centroid id of the slot or 0xFFFFFFFF (padding)
// pmeta[8*tile + ch]  (group << 1) | starts_new_group, for the 4-slot chunk ch of the tile
template <int DP, int METRIC, bool FAST>
__global__ __launch_bounds__(256, 2) void yy_init_mfma_kernel(YyArgs a) {
  constexpr int NK = DP / 2, LDW = DP + 4, TILE = 32 * LDW, NST = (8 * DP + 255) / 256;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  auto tile_ptr = [&](int buf) { return lds + buf * TILE; };
  auto bias_ptr = [&](int buf) { return lds + 2 * TILE + buf * 32; };
  auto id_ptr = [&](int buf) { return reinterpret_cast<uint32_t *>(lds + 2 * TILE + 64) + buf * 32; };
  auto meta_ptr = [&](int buf) { return reinterpret_cast<uint32_t *>(lds + 2 * TILE + 128) + buf * 8; };

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, col = lane & 31, h = lane >> 5;
  const uint32_t D = a.D, K = a.K, G = a.G, len = a.len;
  const uint32_t s = blockIdx.x * 128u + wave * 32u + col;
  const bool live = s < len;

  KMX_YY_LOAD_ROWS(a.samples, s, live)
  (void)xmu;

  const uint32_t nearest = live ? a.assignments[s] : 0xFFFFFFFFu;

  // two scores closer than thr cannot be ordered by the filter
  const float cmaxc = sqrtf(__uint_as_float(a.stats[0])) * 1.000001f;
  const float bmaxc = __uint_as_float(a.stats[1]);
  const float xo = sqrtf(xo2) * 1.0001f, xc = sqrtf(xc2) * 1.0001f;
  const float u = 5.9604645e-8f;
  float thr = 2.0f * (2.0f * a.eps * (xc * cmaxc + bmaxc)) * 1.01f;
  if (METRIC == 0) thr += 16.0f * u * (xc + cmaxc) * (xc + cmaxc);
  else thr += 16.0f * u * xo * sqrtf(__uint_as_float(a.stats[2])) + 2e-6f;

  f32x4 stage[NST];
  float bstage = 0.f;
  uint32_t istage = 0xFFFFFFFFu, mstage = 0;
  auto stage_load = [&](uint32_t tile) {
    const float *src = a.pfil + (size_t)tile * 32 * DP;
#pragma unroll
    for (int i = 0; i < NST; i++) {
      const int q = tid + i * 256;
      if (q < 8 * DP) stage[i] = reinterpret_cast<const f32x4 *>(src)[q];
    }
    if (tid < 32) {
      bstage = a.pbias[tile * 32 + tid];
      istage = a.pids[tile * 32 + tid];
    }
    if (tid < 8) mstage = a.pmeta[tile * 8 + tid];
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
      id_ptr(buf)[tid] = istage;
    }
    if (tid < 8) meta_ptr(buf)[tid] = mstage;
  };

  // queue of (group, centroid) distance evaluations; a group's entries are adjacent, its minimum
  // is carried across flushes and stored when the next group's first entry is replayed (or at the end)
  uint32_t qc[4] = {0, 0, 0, 0}, qg[4] = {0, 0, 0, 0};
  int qn = 0;
  uint32_t carry_g = 0xFFFFFFFFu;
  float carry_min = kFltMax;
  auto store_carry = [&]() __attribute__((always_inline)) {
    if (carry_g != 0xFFFFFFFFu && live && h == 0) a.bounds[(size_t)len * (1 + carry_g) + s] = carry_min;
  };
  // the exact chains run as a two-batch pipeline (exact_split.hpp): a flush starts the queued batch and
  // completes -- and replays -- the one before it
  uint32_t pqc[4] = {0, 0, 0, 0}, pqg[4] = {0, 0, 0, 0};
  int pqn = 0;
  ExactPipe4 pipe;
  auto flush = [&]() __attribute__((always_inline)) {  // wave-uniform call
    uint32_t cidx[4];
#pragma unroll
    for (int i = 0; i < 4; i++) cidx[i] = h ? (i < pqn ? pqc[i] : 0u) : (i < qn ? qc[i] : 0u);
    float dist[4];
    exact_distance4_pipe<NK, METRIC, FAST>(xrow, a.centroids, cidx, D, h, col, pipe, dist);
#pragma unroll
    for (int i = 0; i < 4; i++) {
      if (i < pqn) {
        if (pqg[i] != carry_g) {
          store_carry();
          carry_g = pqg[i];
          carry_min = kFltMax;
        }
        if (dist[i] < carry_min) carry_min = dist[i];   // kmeans.cu:477-481 (NaN never "less")
      }
      pqc[i] = qc[i];
      pqg[i] = qg[i];
    }
    pqn = qn;
    qn = 0;
  };
  auto drain = [&]() __attribute__((always_inline)) {  // wave-uniform call: nothing queued, nothing pending afterwards
    if (__ballot(qn > 0) != 0ull) flush();
    if (__ballot(pqn > 0) != 0ull) flush();
  };
  auto enqueue = [&](uint32_t g, uint32_t c, bool on) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 4; i++)
      if (on && i == qn) {
        qc[i] = c;
        qg[i] = g;
      }
    if (on) qn++;
  };

  // upper bound: exact distance to the row's own centroid (kmeans.cu:474-476); stays FLT_MAX if the
  // row has none (NaN row) or its centroid is in no group (NaN centroid)
  float upper = kFltMax;
  {
    const bool has = live && nearest < K && a.groups[nearest] < G;
    if (__ballot(has) != 0ull) {
      const float *crow[4];
#pragma unroll
      for (int i = 0; i < 4; i++) crow[i] = a.centroids + (size_t)(has ? nearest : 0) * D;
      float dist[4];
      exact_distance4<NK, METRIC, FAST>(xrow, crow, D, h, col, dist);
      if (has) upper = dist[0];
    }
  }

  // running top-3 (by score = smallest distance first) of the CURRENT group in this half-wave
  float v1 = -INFINITY, v2 = -INFINITY, v3 = -INFINITY;
  uint32_t c1 = 0xFFFFFFFFu, c2 = 0xFFFFFFFFu;
  uint32_t cur_group = 0xFFFFFFFFu;
  auto insert = [&](float v, uint32_t idx) {
    const bool g1 = v > v1, g2 = v > v2, g3 = v > v3;
    v3 = g2 ? v2 : (g3 ? v : v3);
    c2 = g1 ? c1 : (g2 ? idx : c2);
    v2 = g1 ? v1 : (g2 ? v : v2);
    c1 = g1 ? idx : c1;
    v1 = g1 ? v : v1;
  };
  auto finalize_group = [&]() __attribute__((always_inline)) {  // wave-uniform call
    if (cur_group == 0xFFFFFFFFu) return;
    {  // merge the partner half-wave's top-3
      const float pv1 = __shfl_xor(v1, 32), pv2 = __shfl_xor(v2, 32), pv3 = __shfl_xor(v3, 32);
      const uint32_t pc1 = __shfl_xor(c1, 32), pc2 = __shfl_xor(c2, 32);
      insert(pv1, pc1);
      insert(pv2, pc2);
      insert(pv3, 0xFFFFFFFFu);
      // Both lanes of a (col, col + 32) pair must now hold the SAME contenders: they evaluate one exact
      // chain between them (lower half: features [0, NK), upper half: the rest).  The merge above keeps a
      // lane's own entry ahead of an EQUAL score from the partner, so on an exact tie of two approximate
      // scores the two lanes disagreed on which centroid is first -- and the chain came out as the first
      // half of one centroid's distance and the second half of the other's (found by the 1M-row parity
      // test, tests/test_gpu_scale.py: 2 bounds in 1e8).  The lower half-wave's view wins.
      v1 = __shfl(v1, col); v2 = __shfl(v2, col); v3 = __shfl(v3, col);
      c1 = __shfl(c1, col); c2 = __shfl(c2, col);
    }
    const bool has1 = live && c1 != 0xFFFFFFFFu;
    const bool sure1 = has1 && ((v1 - v2) > thr);                      // NaN gap => not sure
    const bool sure2 = has1 && !sure1 && c2 != 0xFFFFFFFFu && ((v1 - v3) > thr);
    const bool scan = has1 && !sure1 && !sure2;
    if (__ballot(qn > 2) != 0ull) flush();                              // room for two more everywhere
    if (!has1) {
      // no member other than the row's own centroid (or an empty group): the bound stays FLT_MAX.
      // Replayed through the carry so that the store order stays one group at a time.
      if (carry_g != cur_group) {
        store_carry();
        carry_g = cur_group;
        carry_min = kFltMax;
      }
    }
    enqueue(cur_group, c1, has1 && !scan);
    enqueue(cur_group, c2, sure2);
    if (__ballot(scan) != 0ull) {  // three or more contenders: every member of the group, exactly
      drain();
      if (scan && carry_g != cur_group) {
        store_carry();
        carry_g = cur_group;
        carry_min = kFltMax;
      }
      const uint32_t gb = a.gstart[cur_group], ge = a.gstart[cur_group + 1];
      for (uint32_t i0 = gb; i0 < ge; i0 += 4) {
        const float *crow[4];
        bool on[4];
#pragma unroll
        for (int i = 0; i < 4; i++) {
          const uint32_t c = (i0 + i < ge) ? a.cperm[i0 + i] : a.cperm[gb];
          on[i] = scan && (i0 + i < ge) && c != nearest;
          crow[i] = a.centroids + (size_t)c * D;
        }
        float dist[4];
        exact_distance4<NK, METRIC, FAST>(xrow, crow, D, h, col, dist);
#pragma unroll
        for (int i = 0; i < 4; i++)
          if (on[i] && dist[i] < carry_min) carry_min = dist[i];
      }
    }
    v1 = v2 = v3 = -INFINITY;
    c1 = c2 = 0xFFFFFFFFu;
  };

  const uint32_t ntiles = a.nslots / 32;
  stage_load(0);
  stage_store(0);
  __syncthreads();
  for (uint32_t t = 0; t < ntiles; t++) {
    const int buf = t & 1;
    if (t + 1 < ntiles) stage_load(t + 1);
    KMX_YY_MFMA_TILE(acc, buf)
#pragma unroll
    for (int ch = 0; ch < 8; ch++) {
      const uint32_t meta = meta_ptr(buf)[ch];
      if (meta & 1u) {  // this chunk starts a new group: close the previous one (wave-uniform)
        finalize_group();
        cur_group = meta >> 1;
      }
      if ((ch & 1) == h) {
#pragma unroll
        for (int q = 0; q < 4; q++) {
          const int r = 4 * (ch >> 1) + q;
          const uint32_t row = 8 * (ch >> 1) + q + 4 * h;
          const uint32_t id = id_ptr(buf)[row];
          const float v = (id != 0xFFFFFFFFu && id != nearest) ? acc[r] : -INFINITY;
          insert(v, id);
        }
      }
    }
    if (t + 1 < ntiles) stage_store(buf ^ 1);
    __syncthreads();
  }
  finalize_group();
  drain();
  store_carry();
  if (live && h == 0) a.bounds[s] = upper;
}

// group-sorted padded panel from the centred panel of centroid_prep
__global__ void yy_sorted_panel_kernel(const float *__restrict__ cfil, const float *__restrict__ bias, uint32_t DP,
                                       const uint32_t *__restrict__ pids, uint32_t nslots,
                                       float *__restrict__ pfil, float *__restrict__ pbias) {
  const uint32_t slot = blockIdx.x;
  const uint32_t id = pids[slot];
  for (uint32_t f = threadIdx.x; f < DP; f += blockDim.x)
    pfil[(size_t)slot * DP + f] = id != 0xFFFFFFFFu ? cfil[(size_t)id * DP + f] : 0.f;
  if (threadIdx.x == 0) pbias[slot] = id != 0xFFFFFFFFu ? bias[id] : -INFINITY;
  (void)nslots;
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
template <int DP, int METRIC>
static hipError_t launch_init_t(const YyArgs &a, hipStream_t st) {
  const size_t lds_bytes = (2 * 32 * (DP + 4) + 64 + 64 + 16) * sizeof(float);
  const uint32_t grid = (a.len + 127) / 128;
  if (a.D == (uint32_t)DP)
    hipLaunchKernelGGL((yy_init_mfma_kernel<DP, METRIC, true>), dim3(grid), dim3(256), lds_bytes, st, a);
  else
    hipLaunchKernelGGL((yy_init_mfma_kernel<DP, METRIC, false>), dim3(grid), dim3(256), lds_bytes, st, a);
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

hipError_t launch_yy_init_mfma(int metric, const YyArgs &a, hipStream_t st) {
  if (a.len == 0) return hipSuccess;
  KMX_YY_SWITCH(launch_init_t)
}

hipError_t launch_yy_sorted_panel(const float *cfil, const float *bias, uint32_t DP, const uint32_t *pids,
                                  uint32_t nslots, float *pfil, float *pbias, hipStream_t st) {
  if (nslots == 0) return hipSuccess;
  hipLaunchKernelGGL(yy_sorted_panel_kernel, dim3(nslots), dim3(64), 0, st, cfil, bias, DP, pids, nslots, pfil, pbias);
  return hipGetLastError();
}

}  // namespace kmx
