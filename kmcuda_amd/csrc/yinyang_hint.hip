// yinyang_hint.hip -- kmeans_yy_local_filter (reference: src/kmeans.cu:584-672) with a per-row
// estimate of the FINAL second-best distance as the candidate threshold.
//
// yy_local_mfma_kernel (yinyang_mfma.hip) evaluates the reference's exact distance for every
// centroid whose approximate distance could be below the LIVE second minimum of the reference's
// scan.  The live value starts at FLT_MAX and falls as the scan meets closer centroids, so on data
// without cluster structure a row pays ~40 exact distances where 2 matter.  Here a first kernel
// (yy_hint_kernel: one f16 matrix-core product per 16 features, hi halves only, a best-two
// bookkeeping of 3 VALU operations per score) estimates the row's second-best centroid and the
// local filter takes candidates against  S' = max(upper bound, that estimate)  instead.
//
// S' is only a number: nothing below relies on its quality, only on  S' >= upper bound.  Write R
// for the reference's scan of the row and O for ours.  Both keep (min, nearest, second) where
// second is the second smallest of a multiset: {upper bound} + the exact distances evaluated so
// far + the group bounds folded by the (a) rule (group bound >= upper bound: second = min(second,
// bound), never evaluated).  O differs from R in three ways:
//   (1) it never looks at a centroid whose f32 matrix-core score rules out d <= min(second_O, S')
//       (rigorous bound, as in yinyang_mfma.hip);
//   (2) it folds the (a) bounds that are <= S' all at once, before the scan, and ignores the larger ones;
//   (3) it therefore applies the (b) test "second < bound + drifts -> skip" with its own second.
// Call X the values above S' that R's second has absorbed and O never saw (centroids of (1), bounds of
// (2)).  Claim: second_O <= second_R whenever second_R is not one of X, and min / nearest agree.
//   * What (1) drops cannot lower R's minimum (d > min(second_O, S') >= min) and lowers R's second only
//     with a value above S': an X.
//   * R skips, O evaluates: second_R < bound <= second_O, so second_R is an X and bound > S' -- FLAGGED (F1).
//   * O skips, R evaluates: second_O < bound <= second_R, so second_O is an early-folded group bound b
//     (everything else in O's multiset is in R's).  O has the exact distance anyway (a flush evaluates
//     its whole queue first): if d < bound the row is FLAGGED (F3); else d >= bound > b >= upper bound
//     >= min, so in R this centroid never becomes the minimum and stops mattering for the second
//     once R folds b itself -- until then second_O <= b < d keeps the claim.
//   * Both evaluate: the same exact distance and the same update on both sides.
//   * At the end R has folded every group bound, so second_R = min(second_O, X): equal to second_O iff
//     second_O <= S', else FLAGGED (F2).
// Flagged rows (also: no estimate) are left untouched and appended to flag_rows; the engine runs
// yy_local_mfma_kernel -- the reference's scan replayed state for state -- over that list.  Every
// row's outcome is therefore the reference's, bit for bit, whatever the estimate was.
#include "yinyang_tiles.hpp"

namespace kmx {

typedef _Float16 f16x8h __attribute__((ext_vector_type(8)));

// The (a) rule's share of the second minimum, folded up front (header): the smallest group bound that is
// >= the upper bound, belongs to a group the reference's scan meets (a member other than the row's own
// centroid), and is <= S' -- by walking all G bounds of the row (G lines, 4 N bytes apart).  Both
// half-waves return the same value.
__device__ __forceinline__ float low_bound_fold(const YyArgs &a, uint32_t s, int h, uint32_t cluster, float upper_bound,
                                                float hint, bool on) {
  const uint32_t G = a.G, len = a.len;
  float alow = kFltMax;
  if (__ballot(on) != 0ull) {
    float mine = kFltMax;
    if (on) {
      for (uint32_t g0 = h; g0 < G; g0 += 16) {
        float lb8[8];
#pragma unroll
        for (int q = 0; q < 8; q++) {
          const uint32_t g = g0 + 2 * q;
          lb8[q] = g < G ? a.bounds[(size_t)len * (1 + g) + s] : -INFINITY;
        }
#pragma unroll
        for (int q = 0; q < 8; q++) {
          const float lb = lb8[q];
          if (lb >= upper_bound && lb <= hint && lb < mine) {
            const uint32_t g = g0 + 2 * q;
            uint32_t p = a.gfirst[g];
            if (p == cluster) p = a.gsecond[g];
            if (p != 0xFFFFFFFFu) mine = lb;  // the group has a member the reference's scan meets
          }
        }
      }
    }
    const float other = __shfl_xor(mine, 32);
    const float both = other < mine ? other : mine;
    if (on) alow = both;
  }
  return alow;
}

// ---------------------------------------------------------------------------------------
// the estimate: best two coarse scores of every passed row
// ---------------------------------------------------------------------------------------
// A wave owns 64 rows as two 32-row operand sets, so every centroid fragment read from LDS feeds two
// matrix products (with one set the LDS reads alone are half the LDS bandwidth at full matrix rate).
template <int DP, int METRIC, bool FAST>
__global__ __launch_bounds__(256, 2) void yy_hint_kernel(YyArgs a) {
  constexpr int NK = DP / 2, KS = NK / 8, LDWH = DP / 2 + 4, TILE = 32 * LDWH, NSTH = (4 * DP + 255) / 256;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  auto tile_ptr = [&](int buf) { return lds + buf * TILE; };
  auto bias_ptr = [&](int buf) { return lds + 2 * TILE + buf * 32; };

  const uint32_t npassed = *a.count_ptr;
  if (blockIdx.x * 256u >= npassed) return;  // block-uniform
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, col = lane & 31, h = lane >> 5;
  const uint32_t D = a.D, K = a.K, G = a.G, len = a.len;
  uint32_t pi[2], s[2];
  bool live[2];
#pragma unroll
  for (int e = 0; e < 2; e++) {
    pi[e] = blockIdx.x * 256u + wave * 64u + 32u * e + col;
    live[e] = pi[e] < npassed;
    s[e] = live[e] ? a.passed[pi[e]] : 0u;
  }

  f16x8h xh[2][KS];
  float xc2[2], xmu[2];
#pragma unroll
  for (int e = 0; e < 2; e++) {
    float xc2e, xmue;
    {
      KMX_YY_LOAD_ROWS(a.samples, s[e], live[e])
      (void)xo2; (void)xrow;
#pragma unroll
      for (int j = 0; j < KS; j++) {
        f16x8h v;
#pragma unroll
        for (int q = 0; q < 8; q++) v[q] = (_Float16)xb[8 * j + q];
        xh[e][j] = v;
      }
      xc2e = xc2;
      xmue = xmu;
    }
    xc2[e] = xc2e;
    xmu[e] = xmue;
    // one set at a time: both sets' fp32 rows in flight at once would not fit the register file
#pragma unroll
    for (int j = 0; j < KS; j++) asm volatile("" : "+v"(xh[e][j]));
    __builtin_amdgcn_sched_barrier(0);
  }

  f32x4 stage[NSTH];
  float bstage = 0.f;
  auto stage_load = [&](uint32_t tile) {
    const f32x4 *src = reinterpret_cast<const f32x4 *>(reinterpret_cast<const _Float16 *>(a.panelhi) + (size_t)tile * 32 * DP);
#pragma unroll
    for (int i = 0; i < NSTH; i++) {
      const int q = tid + i * 256;
      if (q < 4 * DP) stage[i] = src[q];
    }
    if (tid < 32) bstage = a.bias[tile * 32 + tid];
  };
  auto stage_store = [&](int buf) {
#pragma unroll
    for (int i = 0; i < NSTH; i++) {
      const int q = tid + i * 256;
      if (q < 4 * DP) {
        const int row = q / (DP / 8), c8 = q % (DP / 8);
        *reinterpret_cast<f32x4 *>(tile_ptr(buf) + row * LDWH + c8 * 4) = stage[i];
      }
    }
    if (tid < 32) bias_ptr(buf)[tid] = bstage;
  };

  // best two scores of my 16 accumulator rows per tile; the register number rides in the low 4
  // mantissa bits, the tile index is noted once per tile
  float v1[2] = {-INFINITY, -INFINITY}, v2[2] = {-INFINITY, -INFINITY};
  uint32_t t1[2] = {0, 0}, t2[2] = {0, 0};
  const uint32_t ntiles = a.K_pad / 32;
  stage_load(0);
  stage_store(0);
  __syncthreads();
  for (uint32_t t = 0; t < ntiles; t++) {
    const int buf = t & 1;
    if (t + 1 < ntiles) stage_load(t + 1);
    f32x16 acc[2];
    {
      const float *bb = bias_ptr(buf) + 4 * h;
#pragma unroll
      for (int g4 = 0; g4 < 4; g4++) {
        const f32x4 b4 = *reinterpret_cast<const f32x4 *>(bb + 8 * g4);
#pragma unroll
        for (int e = 0; e < 2; e++) {
          acc[e][4 * g4 + 0] = b4.x; acc[e][4 * g4 + 1] = b4.y; acc[e][4 * g4 + 2] = b4.z; acc[e][4 * g4 + 3] = b4.w;
        }
      }
      const _Float16 *arow = reinterpret_cast<const _Float16 *>(tile_ptr(buf) + col * LDWH) + h * NK;
#pragma unroll
      for (int j = 0; j < KS; j++) {
        const f16x8h af = *reinterpret_cast<const f16x8h *>(arow + 8 * j);
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af, xh[0][j], acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af, xh[1][j], acc[1], 0, 0, 0);
      }
    }
#pragma unroll
    for (int e = 0; e < 2; e++) {
      const float o1 = v1[e], o2 = v2[e];
#pragma unroll
      for (int r = 0; r < 16; r++) {
        const float v = __uint_as_float((__float_as_uint(acc[e][r]) & ~15u) | (uint32_t)r);
        v2[e] = __builtin_amdgcn_fmed3f(v1[e], v2[e], v);
        v1[e] = fmaxf(v1[e], v);
      }
      const uint32_t n1 = (v1[e] == o1) ? t1[e] : ((v1[e] == o2) ? t2[e] : t);
      const uint32_t n2 = (v2[e] == o1) ? t1[e] : ((v2[e] == o2) ? t2[e] : t);
      t1[e] = n1;
      t2[e] = n2;
    }
    if (t + 1 < ntiles) stage_store(buf ^ 1);
    __syncthreads();
  }

#pragma unroll
  for (int e = 0; e < 2; e++) {
    // the better of the (up to four) noted centroids that is neither the row's own nor groupless
    const float pv1 = __shfl_xor(v1[e], 32), pv2 = __shfl_xor(v2[e], 32);
    const uint32_t pt1 = __shfl_xor(t1[e], 32), pt2 = __shfl_xor(t2[e], 32);
    if (!live[e] || h != 0) continue;
    const uint32_t row = s[e];
    const float upper_bound = a.bounds[row];
    const uint32_t cluster = a.assignments[row];
    const float cv[4] = {v1[e], v2[e], pv1, pv2};
    const uint32_t ct[4] = {t1[e], t2[e], pt1, pt2};
    float best = -INFINITY;
    uint32_t best_g = 0xFFFFFFFFu;
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const uint32_t r = __float_as_uint(cv[i]) & 15u;
      const uint32_t c = ct[i] * 32u + (r & 3u) + 8u * (r >> 2) + 4u * (i >= 2 ? 1u : 0u);
      if (c < K && c != cluster && cv[i] > best) {
        const uint32_t g = a.groups[c];
        if (g < G) {
          best = cv[i];
          best_g = g;
        }
      }
    }
    float hint = INFINITY;
    if (best_g < G) {
      const float lbg = a.bounds[(size_t)len * (1 + best_g) + row];
      float y;
      if (lbg >= upper_bound) {
        y = lbg;  // an (a) group: its bound itself enters the second minimum
      } else {
        // typical rounding of the hi.hi products (a fraction of the worst case 2^-10 ||x'|| C'max; the
        // fraction only trades candidates against handed-over rows), 16 ulp for the register number
        const float cmaxc = sqrtf(__uint_as_float(a.stats[0]));
        const float xc = sqrtf(xc2[e]);
        const float frac = fminf(1.0f, 4.0f / sqrtf((float)DP));
        const float e_h = 9.8e-4f * frac * xc * cmaxc + 4e-6f * fabsf(best) + 1e-6f * xc2[e];
        if (METRIC == 0) {
          y = sqrtf(fmaxf(xc2[e] - 2.0f * (best - e_h), 0.f)) * 1.00001f;
        } else {
          const float dot = best + xmu[e] - e_h;
          y = (dot >= 1.f ? 0.f : (dot <= -1.f ? 3.1415927f : acosf(dot))) * 1.00001f + 1e-6f;
        }
      }
      hint = fmaxf(upper_bound, y);  // a NaN y leaves the upper bound
      if (!(hint >= upper_bound)) hint = INFINITY;
    }
    a.hint[row] = hint;
  }
}

// ---------------------------------------------------------------------------------------
// the local filter against the estimate
// ---------------------------------------------------------------------------------------
// F16: the candidate sweep on the f16 matrix cores (hi halves only, like the estimate) with the
// coarse Lloyd stage's rigorous bound on the dropped parts (lloyd_f16.hip): the
// measured ||x' - hi(x')|| of the row and max ||c' - hi(c')|| of the panel (stats[5]).  !F16: the f32
// matrix cores of yinyang_mfma.hip (KMCUDA_AMD_YY_HINT=2, cross-check).
template <int DP, int METRIC, bool FAST, bool F16>
__global__ __launch_bounds__(256, 2) void yy_local_hint_kernel(YyArgs a) {
  constexpr int NK = DP / 2, KS = NK / 8;
  constexpr int LDW = F16 ? DP / 2 + 4 : DP + 4;          // LDS row in 4-byte words
  constexpr int TILE = 32 * LDW;
  constexpr int NST = ((F16 ? 4 : 8) * DP + 255) / 256;    // 16-byte pieces of a tile per thread
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
  f16x8h xh[F16 ? KS : 1];
  float dx2 = 0.f;  // ||x' - hi(x')||^2, measured
  if constexpr (F16) {
#pragma unroll
    for (int j = 0; j < KS; j++) {
      f16x8h v;
#pragma unroll
      for (int q = 0; q < 8; q++) {
        const _Float16 hi = (_Float16)xb[8 * j + q];
        const float r = xb[8 * j + q] - (float)hi;  // exact
        dx2 = fmaf(r, r, dx2);
        v[q] = hi;
      }
      xh[j] = v;
    }
    dx2 += __shfl_xor(dx2, 32);
  }

  const float upper_bound = live ? a.bounds[s] : 0.f;
  const uint32_t cluster = live ? a.assignments[s] : 0xFFFFFFFFu;
  const float hint = live ? a.hint[s] : INFINITY;
  float min_dist = upper_bound, second_min = kFltMax;
  uint32_t nearest = cluster;
  bool bad = !(hint < INFINITY);  // no estimate (or a dead lane): the plain kernel's row
  uint32_t why = bad ? 1u : 0u;   // statistics: first reason the row was handed over

  // (a) groups (bound >= upper bound) whose bound can still matter (<= S'): folded up front
  second_min = low_bound_fold(a, s, h, cluster, upper_bound, hint, !bad);

  // threshold in accumulator space (yinyang_mfma.hip): a centroid can only matter if acc >= amin
  const float cmaxc = sqrtf(__uint_as_float(a.stats[0])) * 1.000001f;
  const float bmaxc = __uint_as_float(a.stats[1]);
  const float xo = sqrtf(xo2) * 1.0001f, xc = sqrtf(xc2) * 1.0001f;
  float e_mfma = 2.0f * a.eps * (xc * cmaxc + bmaxc) * 1.01f;
  if constexpr (F16) {
    // x'.c' - hi(x').hi(c') = x'.dc + dx.c' - dx.dc, Cauchy-Schwarz on the measured residual norms; the
    // last term covers the absolute rounding of halves below the normal range
    const float dcmax = sqrtf(__uint_as_float(a.stats[5])) * 1.0001f;
    const float dx = sqrtf(dx2) * 1.0001f;
    e_mfma += (xc * dcmax + dx * cmaxc + dx * dcmax) * 1.001f + 6e-8f * sqrtf((float)DP) * (xc + cmaxc);
    if (!(xc < 6.0e4f && cmaxc < 6.0e4f && e_mfma < INFINITY) && !bad) {  // a half overflowed: no statement
      bad = true;
      why = 1u;
    }
  }
  const float e_cos = e_mfma + a.eps * xo * sqrtf(__uint_as_float(a.stats[3])) * 1.01f + 1e-6f;
  auto amin_of = [&](float sm) -> float {
    if (METRIC == 0) {
      const float T2 = sm * sm * 1.000002f;
      return 0.5f * (xc2 - T2) - e_mfma - 1e-6f * (xc2 + T2);
    }
    if (sm >= 3.1415925f) return -INFINITY;
    return cosf(sm) - xmu - e_cos;
  };
  float amin = amin_of(fminf(second_min, hint));

  f32x4 stage[NST];
  float bstage = 0.f;
  uint32_t gstage = 0;
  constexpr int PIECES = (F16 ? 4 : 8) * DP;   // 16-byte pieces per tile
  constexpr int PPR = F16 ? DP / 8 : DP / 4;    // ... per centroid row
  auto stage_load = [&](uint32_t tile) {
    const f32x4 *src = F16 ? reinterpret_cast<const f32x4 *>(reinterpret_cast<const _Float16 *>(a.panelhi) + (size_t)tile * 32 * DP)
                           : reinterpret_cast<const f32x4 *>(a.cfil + (size_t)tile * 32 * DP);
#pragma unroll
    for (int i = 0; i < NST; i++) {
      const int q = tid + i * 256;
      if (q < PIECES) stage[i] = src[q];
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
      if (q < PIECES) {
        const int row = q / PPR, c4 = q % PPR;
        *reinterpret_cast<f32x4 *>(tile_ptr(buf) + row * LDW + c4 * 4) = stage[i];
      }
    }
    if (tid < 32) {
      bias_ptr(buf)[tid] = bstage;
      grp_ptr(buf)[tid] = gstage;
    }
  };

  // queue of candidates (ascending c)
  uint32_t qc[4] = {0, 0, 0, 0};
  int qn = 0;
  uint32_t n_flush = 0, n_cand = 0;
  auto flush = [&]() {  // wave-uniform call
    n_flush++;
    const float *crow[4];
#pragma unroll
    for (int i = 0; i < 4; i++) crow[i] = a.centroids + (size_t)(i < qn ? qc[i] : 0) * D;
    float dist[4];
    // the fullest queue of the wave sets how many candidate rows are gathered (a row has 0.2 - 2 real ones)
    const int nq = __ballot(qn >= 4) ? 4 : (__ballot(qn >= 3) ? 3 : (__ballot(qn >= 2) ? 2 : 1));
    exact_distance4<NK, METRIC, FAST>(xrow, crow, D, h, col, dist, nq);
#pragma unroll
    for (int i = 0; i < 4; i++) {
      if (i < qn) {
        const uint32_t c = qc[i];
        const uint32_t g = a.groups[c];
        float lb = a.bounds[(size_t)len * (1 + g) + s];
        lb += a.gdrifts[g] - a.drifts[(size_t)K * D + c];    // kmeans.cu:637
        if (!(second_min < lb)) {                            // :638-640
          if (lb > hint) {                                   // F1: the reference may have skipped it
            bad = true;
            if (!why) why = 3u;
          }
          const float d = dist[i];                           // :641-652
          if (d < min_dist) {
            second_min = min_dist;
            min_dist = d;
            nearest = c;
          } else if (d < second_min) {
            second_min = d;
          }
        } else if (dist[i] < lb) {                           // F3: skipped here on the strength of a bound
          bad = true;                                        // that does not hold; the reference may not have
          if (!why) why = 2u;                                // skipped it (its second minimum lags ours)
        }
      }
    }
    qn = 0;
    amin = amin_of(fminf(second_min, hint));
  };

  const uint32_t ntiles = a.K_pad / 32;
  const bool wave_live = __ballot(live && !bad) != 0ull;
  stage_load(0);
  stage_store(0);
  __syncthreads();
  for (uint32_t t = 0; t < ntiles; t++) {
    const int buf = t & 1;
    if (t + 1 < ntiles) stage_load(t + 1);
    if (wave_live) {
      f32x16 acc;
      if constexpr (F16) {
        const float *bb = bias_ptr(buf) + 4 * h;
#pragma unroll
        for (int g4 = 0; g4 < 4; g4++) {
          const f32x4 b4 = *reinterpret_cast<const f32x4 *>(bb + 8 * g4);
          acc[4 * g4 + 0] = b4.x; acc[4 * g4 + 1] = b4.y; acc[4 * g4 + 2] = b4.z; acc[4 * g4 + 3] = b4.w;
        }
        const _Float16 *arow = reinterpret_cast<const _Float16 *>(tile_ptr(buf) + col * LDW) + h * NK;
#pragma unroll
        for (int j = 0; j < KS; j++) {
          const f16x8h af = *reinterpret_cast<const f16x8h *>(arow + 8 * j);
          acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(af, xh[j], acc, 0, 0, 0);
        }
      } else {
        KMX_YY_MFMA_TILE(acc32, buf)
        acc = acc32;
      }
      uint32_t m16 = 0;
      if (live && !bad) {
#pragma unroll
        for (int r = 0; r < 16; r++)
          if (acc[r] >= amin) m16 |= 1u << r;
      }
      if (__ballot(m16 != 0u) != 0ull) {
        const uint32_t pm = __shfl_xor(m16, 32);
        const uint32_t m0 = h ? pm : m16, m1 = h ? m16 : pm;
        uint32_t rowmask = 0;
#pragma unroll
        for (int g4 = 0; g4 < 4; g4++)
          rowmask |= (((m0 >> (4 * g4)) & 0xFu) << (8 * g4)) | (((m1 >> (4 * g4)) & 0xFu) << (8 * g4 + 4));
        while (__ballot(rowmask != 0u) != 0ull) {
          if (__ballot(qn == 4) != 0ull) flush();  // some row's queue is full
          const bool active = rowmask != 0u;
          const uint32_t rho = active ? (uint32_t)__ffs((int)rowmask) - 1u : 0u;
          rowmask &= rowmask - 1u;
          if (active) {
            const uint32_t c = t * 32 + rho;
            const uint32_t g = grp_ptr(buf)[rho];
            if (g < G && c != cluster) {  // g >= G: NaN centroid or padding
              const float lbg = a.bounds[(size_t)len * (1 + g) + s];
              if (!(lbg >= upper_bound)) {  // else an (a) centroid: an event if its bound is <= S'
#pragma unroll
                for (int i = 0; i < 4; i++)
                  if (i == qn) qc[i] = c;
                qn++;
                n_cand++;
              }
            }
          }
        }
      }
    }
    if (t + 1 < ntiles) stage_store(buf ^ 1);
    __syncthreads();
  }
  if (wave_live && __ballot(qn > 0) != 0ull) flush();
  if (!(second_min <= hint)) {  // F2: the reference's second minimum may be a value we never saw
    bad = true;
    if (!why) why = 4u;
  }

  // write-back, kmeans.cu:653-671 -- or hand the row to the plain kernel, untouched
  bool changed = false;
  const bool mine = live && h == 0;
  if (mine && !bad) {
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
  const unsigned long long fm = __ballot(mine && bad);
  if (fm) {
    uint32_t base = 0;
    if (lane == 0) {
      base = atomicAdd(&a.counters[5], (uint32_t)__popcll(fm));
      atomicAdd(&a.counters[7], (uint32_t)__popcll(fm));
    }
    base = __shfl(base, 0);
    if (mine && bad) a.flag_rows[base + (uint32_t)__popcll(fm & ((1ull << lane) - 1ull))] = s;
  }
  {  // statistics (not part of the reference's state)
    uint32_t nc = (mine && !bad) ? n_cand : 0u;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) nc += __shfl_xor(nc, off);
    const uint32_t nmine = (uint32_t)__popcll(__ballot(mine));
    uint32_t nwhy[4];
#pragma unroll
    for (int w = 0; w < 4; w++) nwhy[w] = (uint32_t)__popcll(__ballot(mine && why == (uint32_t)(w + 1)));
    if (lane == 0) {
      atomicAdd(&a.counters[3], nc);
      atomicAdd(&a.counters[1], n_flush);
      atomicAdd(&a.counters[6], nmine);
#pragma unroll
      for (int w = 0; w < 4; w++)
        if (nwhy[w]) atomicAdd(&a.counters[8 + w], nwhy[w]);
    }
  }
}

// ---------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------
bool yy_hint_supported(uint32_t DP) { return DP >= 16 && DP <= 256; }

template <int DP, int METRIC>
static hipError_t launch_hint_t(const YyArgs &a, hipStream_t st) {
  const size_t lds_bytes = (2 * 32 * (DP / 2 + 4) + 64) * sizeof(float);
  const uint32_t grid = (a.len + 255) / 256;  // worst case; blocks beyond the passed count exit at once
  if (a.D == (uint32_t)DP)
    hipLaunchKernelGGL((yy_hint_kernel<DP, METRIC, true>), dim3(grid), dim3(256), lds_bytes, st, a);
  else
    hipLaunchKernelGGL((yy_hint_kernel<DP, METRIC, false>), dim3(grid), dim3(256), lds_bytes, st, a);
  return hipGetLastError();
}
template <int DP, int METRIC>
static hipError_t launch_local_hint_t(const YyArgs &a, hipStream_t st) {
  const uint32_t grid = (a.len + 127) / 128;
  if (a.hint_f32_sweep) {
    const size_t lds_bytes = (2 * 32 * (DP + 4) + 64 + 64) * sizeof(float);
    if (a.D == (uint32_t)DP)
      hipLaunchKernelGGL((yy_local_hint_kernel<DP, METRIC, true, false>), dim3(grid), dim3(256), lds_bytes, st, a);
    else
      hipLaunchKernelGGL((yy_local_hint_kernel<DP, METRIC, false, false>), dim3(grid), dim3(256), lds_bytes, st, a);
  } else {
    const size_t lds_bytes = (2 * 32 * (DP / 2 + 4) + 64 + 64) * sizeof(float);
    if (a.D == (uint32_t)DP)
      hipLaunchKernelGGL((yy_local_hint_kernel<DP, METRIC, true, true>), dim3(grid), dim3(256), lds_bytes, st, a);
    else
      hipLaunchKernelGGL((yy_local_hint_kernel<DP, METRIC, false, true>), dim3(grid), dim3(256), lds_bytes, st, a);
  }
  return hipGetLastError();
}

#define KMX_YYH_SWITCH(fn)                                                             \
  switch (a.DP) {                                                                      \
    case 16: return metric == 0 ? fn<16, 0>(a, st) : fn<16, 1>(a, st);                 \
    case 32: return metric == 0 ? fn<32, 0>(a, st) : fn<32, 1>(a, st);                 \
    case 64: return metric == 0 ? fn<64, 0>(a, st) : fn<64, 1>(a, st);                 \
    case 128: return metric == 0 ? fn<128, 0>(a, st) : fn<128, 1>(a, st);              \
    case 256: return metric == 0 ? fn<256, 0>(a, st) : fn<256, 1>(a, st);              \
    default: return hipErrorInvalidValue;                                              \
  }

hipError_t launch_yy_hint(int metric, const YyArgs &a, hipStream_t st) {
  if (a.len == 0) return hipSuccess;
  KMX_YYH_SWITCH(launch_hint_t)
}

hipError_t launch_yy_local_hint(int metric, const YyArgs &a, hipStream_t st) {
  if (a.len == 0) return hipSuccess;
  KMX_YYH_SWITCH(launch_local_hint_t)
}

}  // namespace kmx
