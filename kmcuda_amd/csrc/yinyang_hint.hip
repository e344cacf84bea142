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
#include <type_traits>
#include "yinyang_tiles.hpp"

namespace kmx {

typedef _Float16 f16x8h __attribute__((ext_vector_type(8)));

// hand-issued LDS reads (raw LDS byte address) with counted waits: the compiler neither knows these reads nor
// drains the panel DMA in front of them (lloyd_f16.hip, knn_f16.hip)
__device__ __forceinline__ f16x8h yyh_frag_issue(uint32_t addr) {
  f16x8h f;
  asm volatile("ds_read_b128 %0, %1" : "=v"(f) : "v"(addr) : "memory");
  return f;
}
__device__ __forceinline__ f32x4 yyh_lds_read4(uint32_t addr) {
  f32x4 f;
  asm volatile("ds_read_b128 %0, %1" : "=v"(f) : "v"(addr) : "memory");
  return f;
}
__device__ __forceinline__ uint32_t yyh_lds_read1(uint32_t addr) {   // read and wait
  uint32_t v;
  asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
  return v;
}
template <int N>
__device__ __forceinline__ void yyh_frag_wait(f16x8h &f) {
  asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(f) : "n"(N));
}

// The (a) rule's share of the second minimum, folded up front (header): the smallest group bound that is
// >= the upper bound, belongs to a group the reference's scan meets (a member other than the row's own
// centroid), and is <= S' -- by walking all G bounds of the row (G lines, 4 N bytes apart).  Both
// half-waves return the same value.  amask (G <= 128): bit g set iff bound[g] >= upper bound -- the (a) test of a
// candidate's group without another trip to global memory (the row's bounds do not change until its write-back).
__device__ __forceinline__ float low_bound_fold(const YyArgs &a, uint32_t s, int h, uint32_t cluster, float upper_bound,
                                                float hint, bool on, uint32_t (&amask)[4]) {
  const uint32_t G = a.G, len = a.len;
  float alow = kFltMax;
#pragma unroll
  for (int w = 0; w < 4; w++) amask[w] = 0u;
  if (__ballot(on) != 0ull) {
    float mine = kFltMax;
    if (on) {
      // The bounds of 16 groups at a time -- and the two smallest members of each, which decide whether the reference's
      // scan meets the group at all -- are requested before the first is looked at, and looked at without a
      // branch: as a conditional load of gfirst / gsecond per qualifying group the fold was a chain of
      // dependent round trips (29 K of a wave's 167 K cycles, profiles/r2m_*timeline*).
      auto look = [&](float lb, uint32_t g, uint32_t gf, uint32_t gs) {
        const uint32_t bit = (lb >= upper_bound && g < 128u) ? 1u << (g & 31u) : 0u;
#pragma unroll
        for (int w = 0; w < 4; w++) amask[w] |= (g >> 5) == (uint32_t)w ? bit : 0u;
        const uint32_t p = gf == cluster ? gs : gf;   // 0xFFFFFFFF: no member the scan meets
        if (lb >= upper_bound && lb <= hint && lb < mine && p != 0xFFFFFFFFu) mine = lb;
      };
#pragma unroll 1
      for (int half = 0; half < 4; half++) {   // four batches of 16 groups per lane (rolled: 48 registers, not 192)
        float lbv[2][8];
        uint32_t gfv[2][8], gsv[2][8];
#pragma unroll
        for (int it = 0; it < 2; it++) {
#pragma unroll
          for (int q = 0; q < 8; q++) {
            const uint32_t g = (uint32_t)h + 16u * (2 * half + it) + 2u * q;
            lbv[it][q] = g < G ? a.bounds[(size_t)len * (1 + g) + s] : -INFINITY;
            gfv[it][q] = g < G ? a.gfirst[g] : 0xFFFFFFFFu;
            gsv[it][q] = g < G ? a.gsecond[g] : 0xFFFFFFFFu;
          }
        }
#pragma unroll
        for (int it = 0; it < 2; it++) {
#pragma unroll
          for (int q = 0; q < 8; q++)
            look(lbv[it][q], (uint32_t)h + 16u * (2 * half + it) + 2u * q, gfv[it][q], gsv[it][q]);
        }
      }
      for (uint32_t g0 = 128u + h; g0 < G; g0 += 16) {
        float lb8[8];
#pragma unroll
        for (int q = 0; q < 8; q++) {
          const uint32_t g = g0 + 2 * q;
          lb8[q] = g < G ? a.bounds[(size_t)len * (1 + g) + s] : -INFINITY;
        }
#pragma unroll
        for (int q = 0; q < 8; q++) {
          const uint32_t g = g0 + 2 * q;
          look(lb8[q], g, g < G ? a.gfirst[g] : 0xFFFFFFFFu, g < G ? a.gsecond[g] : 0xFFFFFFFFu);
        }
      }
    }
    const float other = __shfl_xor(mine, 32);
    const float both = other < mine ? other : mine;
    if (on) alow = both;
#pragma unroll
    for (int w = 0; w < 4; w++) amask[w] |= __shfl_xor(amask[w], 32);
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
  constexpr int NK = DP / 2, KS = NK / 8;
  constexpr int ROWB = DP * 2;              // bytes of one panel row (DP hi halves)
  constexpr int SUPB = 64 * ROWB;           // one super-tile: 64 centroids
  constexpr int NP = SUPB / 1024;           // 1-KB LDS-DMA pieces per super-tile
  constexpr int SWM = (KS < 16 ? KS : 16) - 1;
  constexpr int WV = 4, PPW = (NP + WV - 1) / WV;
  typedef __attribute__((address_space(3))) unsigned char lds_byte;
  extern __shared__ __attribute__((aligned(1024))) unsigned char lds2[];
  const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_byte *)lds2;
  if (lds0 & 1023u) __builtin_trap();
  const uint32_t bias0 = lds0 + 2 * SUPB;   // 2 x 64 floats

  const uint32_t npassed = *a.count_ptr;
  if (blockIdx.x * 256u >= npassed) return;  // block-uniform
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), col = lane & 31, h = lane >> 5;
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
  if (FAST && METRIC == 0 && a.xcache) {   // block-uniform: the Lloyd coarse stage's row cache (see yy_local_hint_kernel)
#pragma unroll
    for (int e = 0; e < 2; e++) {
      const uint32_t sr = live[e] ? s[e] : 0u;
      const f16x8h *src = reinterpret_cast<const f16x8h *>(a.xcache) + (size_t)(sr >> 5) * KS * 64 + (sr & 31u) + 32u * h;
#pragma unroll
      for (int j = 0; j < KS; j++) {
        xh[e][j] = src[(size_t)j * 64];
        if (!live[e]) {
#pragma unroll
          for (int q = 0; q < 8; q++) xh[e][j][q] = (_Float16)0.f;
        }
      }
      xc2[e] = live[e] ? reinterpret_cast<const float2 *>(a.xmeta)[sr].x : 0.f;
      xmu[e] = 0.f;   // angular only
    }
  } else {
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
  }

  const uint32_t nsuper = (a.K_pad + 63u) / 64u;
  const float *biashi = reinterpret_cast<const float *>(reinterpret_cast<const unsigned char *>(a.panelhi) + (size_t)nsuper * SUPB);
  // the panel streams as in yy_local_hint_kernel below (and lloyd_f16.hip): 64-centroid super-tiles by LDS-DMA
  auto issue_piece = [&](uint32_t sp, int buf, int i) {
    if (i == PPW) {
      if (wave == 0)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(biashi + sp * 64u + lane),
                                         (__attribute__((address_space(3))) void *)(uintptr_t)(bias0 + buf * 256), 4, 0, 0);
      return;
    }
    const int p = wave + WV * i;
    if (p >= NP) return;   // wave-uniform
    const unsigned char *src = reinterpret_cast<const unsigned char *>(a.panelhi) + (size_t)sp * SUPB;
    uint32_t P0 = (uint32_t)lane * 16u;
    asm volatile("" : "+v"(P0));
    const uint32_t P = (uint32_t)p * 1024u + P0;
    const uint32_t from = P ^ (((P / ROWB) & SWM) << 4);
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + from),
                                     (__attribute__((address_space(3))) void *)(uintptr_t)(lds0 + buf * SUPB + p * 1024), 16, 0, 0);
  };
  const uint32_t fragbase = lds0 + (uint32_t)col * ROWB + (uint32_t)h * (KS * 16);
  const uint32_t fragswz = (uint32_t)(col & SWM) * 16u;

  // best two scores of my 16 accumulator rows per tile; the register number rides in the low 4
  // mantissa bits, the tile index is noted once per tile
  float v1[2] = {-INFINITY, -INFINITY}, v2[2] = {-INFINITY, -INFINITY};
  uint32_t t1[2] = {0, 0}, t2[2] = {0, 0};
#pragma unroll
  for (int i = 0; i <= PPW; i++) issue_piece(0, 0, i);
  constexpr int DSTR = (2 * KS) / (PPW + 1) > 0 ? (2 * KS) / (PPW + 1) : 1;   // a piece every DSTR k-steps
  for (uint32_t sp = 0; sp < nsuper; sp++) {
    const int buf = sp & 1;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // my pieces of super-tile sp have landed
    __builtin_amdgcn_s_barrier();                      // everybody's have, and everybody is done with sp - 1
    asm volatile("" ::: "memory");
    const bool dma = sp + 1 < nsuper;
#pragma unroll
    for (int sub = 0; sub < 2; sub++) {
      const uint32_t t = sp * 2u + (uint32_t)sub;   // 32-centroid tile index
      const uint32_t tb = fragbase + (uint32_t)buf * SUPB + (uint32_t)sub * (32 * ROWB);
      const uint32_t bb = bias0 + (uint32_t)buf * 256u + (uint32_t)sub * 128u + 16u * h;
      f32x4 b4[4];
#pragma unroll
      for (int g4 = 0; g4 < 4; g4++) b4[g4] = yyh_lds_read4(bb + 32u * g4);
      constexpr int PD = KS < 3 ? KS : 3;   // fragments in flight
      f16x8h fr[PD + 1];
#pragma unroll
      for (int j = 0; j < PD; j++) fr[j] = yyh_frag_issue(tb + ((16u * j) ^ fragswz));
      asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(b4[0]), "+v"(b4[1]), "+v"(b4[2]), "+v"(b4[3]) : "n"(PD));
      f32x16 acc[2];
#pragma unroll
      for (int g4 = 0; g4 < 4; g4++) {
#pragma unroll
        for (int e = 0; e < 2; e++) {
          acc[e][4 * g4 + 0] = b4[g4].x; acc[e][4 * g4 + 1] = b4[g4].y; acc[e][4 * g4 + 2] = b4[g4].z; acc[e][4 * g4 + 3] = b4[g4].w;
        }
      }
#pragma unroll
      for (int j = 0; j < KS; j++) {
        if (j + PD < KS) fr[(j + PD) % (PD + 1)] = yyh_frag_issue(tb + ((16u * (j + PD)) ^ fragswz));
        const int behind = (KS - 1 - j) < PD ? (KS - 1 - j) : PD;
        f16x8h &f = fr[j % (PD + 1)];
        if (behind == 3) yyh_frag_wait<3>(f);
        else if (behind == 2) yyh_frag_wait<2>(f);
        else if (behind == 1) yyh_frag_wait<1>(f);
        else yyh_frag_wait<0>(f);
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f, xh[0][j], acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f, xh[1][j], acc[1], 0, 0, 0);
        {
          const int slot = sub * KS + j;   // compile-time after unrolling
          if (dma && slot % DSTR == 0 && slot / DSTR <= PPW) issue_piece(sp + 1, buf ^ 1, slot / DSTR);
        }
      }
      if (sub == 1 && dma) {   // pieces the slots did not cover (very short rows)
#pragma unroll
        for (int i = (2 * KS - 1) / DSTR + 1; i <= PPW; i++) issue_piece(sp + 1, buf ^ 1, i);
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
    }
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
// The candidate sweep runs on the f16 matrix cores (hi halves only, like the estimate) with the coarse Lloyd
// stage's rigorous bound on the dropped parts (lloyd_f16.hip): the measured ||x' - hi(x')|| of the
// row and max ||c' - hi(c')|| of the panel (stats[5]).  The panel streams like the Lloyd coarse stage's: 64-centroid
// super-tiles by LDS-DMA (bank-swizzled by source address, biases behind the panel), double buffered, the next
// super-tile's pieces issued between this one's matrix products, fragments read by hand with counted waits --
// round 1 staged 32-centroid tiles through registers with a barrier each: 7900 wave cycles per tile for 512
// cycles of products.
template <int DP, int METRIC, bool FAST>
__global__ __launch_bounds__(256, 2) void yy_local_hint_kernel(YyArgs a) {
  constexpr int NK = DP / 2, KS = NK / 8;
  constexpr int ROWB = DP * 2;              // bytes of one panel row (DP hi halves)
  constexpr int SUPB = 64 * ROWB;           // one super-tile: 64 centroids
  constexpr int NP = SUPB / 1024;           // 1-KB LDS-DMA pieces per super-tile
  constexpr int SWM = (KS < 16 ? KS : 16) - 1;
  constexpr int WV = 4, PPW = (NP + WV - 1) / WV;
  typedef __attribute__((address_space(3))) unsigned char lds_byte;
  extern __shared__ __attribute__((aligned(1024))) unsigned char lds2[];
  const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_byte *)lds2;
  if (lds0 & 1023u) __builtin_trap();
  const uint32_t bias0 = lds0 + 2 * SUPB;   // 2 x 64 floats
  const uint32_t grp0 = bias0 + 512;        // 2 x 64 group numbers

  const uint32_t npassed = *a.count_ptr;
  if (blockIdx.x * 128u >= npassed) return;  // block-uniform
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), col = lane & 31, h = lane >> 5;
  const uint32_t D = a.D, K = a.K, G = a.G, len = a.len;
  const uint32_t pi = blockIdx.x * 128u + wave * 32u + col;
  const bool live = pi < npassed;
  const uint32_t s = live ? a.passed[pi] : 0u;

  f16x8h xh[KS];
  float dx2 = 0.f;  // ||x' - hi(x')||^2, measured
  float xo2 = 0.f, xc2 = 0.f, xmu = 0.f;
  const float *xrow = a.samples + (size_t)(live ? s : 0) * D;
  if (FAST && METRIC == 0 && a.xcache) {   // block-uniform
    // the Lloyd coarse stage's row cache: hi(x - mu) in operand order, 2 D bytes per row instead of 4 D, nothing
    // to convert; its records carry the two norms the L2 bound needs (x.mu and ||x|| are angular-only)
    const uint32_t sr = live ? s : 0u;
    const f16x8h *src = reinterpret_cast<const f16x8h *>(a.xcache) + (size_t)(sr >> 5) * KS * 64 + (sr & 31u) + 32u * h;
#pragma unroll
    for (int j = 0; j < KS; j++) {
      xh[j] = src[(size_t)j * 64];
      if (!live) {
#pragma unroll
        for (int q = 0; q < 8; q++) xh[j][q] = (_Float16)0.f;
      }
    }
    const float2 m = reinterpret_cast<const float2 *>(a.xmeta)[sr];
    xc2 = live ? m.x : 0.f;
    dx2 = live ? (m.y < 0.f ? INFINITY : m.y) : 0.f;   // -1: a NaN first feature -- no bound, the plain kernel's row
  } else if constexpr (FAST && DP >= 64) {
    // coalesced through a per-wave LDS scratch (yinyang_tiles.hpp); the tile buffers are not in use yet
    constexpr int CH = NK / 4;
    const uint32_t s_any = live ? s : 0u;
    yy_rows_staged<DP>(a.samples, reinterpret_cast<float *>(lds2), (uint32_t)wave * (32u * (DP / 4 + 4)), lane,
                       [&](int r) { return (uint32_t)__shfl((int)s_any, r); },
                       [&](int c, int k, f32x4 v) {
                         const int jf = c * CH + 4 * k;   // first of four features of my half
                         const f32x4 m4 = *reinterpret_cast<const f32x4 *>(a.mu + h * NK + jf);
                         const float vv[4] = {v.x, v.y, v.z, v.w}, mm[4] = {m4.x, m4.y, m4.z, m4.w};
#pragma unroll
                         for (int q = 0; q < 4; q++) {
                           const float x = live ? vv[q] : 0.f;
                           const float xc = live ? vv[q] - mm[q] : 0.f;
                           xo2 = fmaf(x, x, xo2);
                           xc2 = fmaf(xc, xc, xc2);
                           xmu = fmaf(x, mm[q], xmu);
                           const _Float16 hi = (_Float16)xc;
                           const float r = xc - (float)hi;  // exact
                           dx2 = fmaf(r, r, dx2);
                           xh[(jf + q) / 8][(jf + q) % 8] = hi;
                         }
                       });
    xo2 += __shfl_xor(xo2, 32);
    xc2 += __shfl_xor(xc2, 32);
    xmu += __shfl_xor(xmu, 32);
    dx2 += __shfl_xor(dx2, 32);
    __syncthreads();   // the scratch is tile buffer space: no DMA before every wave is done with it
  } else {
    KMX_YY_LOAD_ROWS_INTO(a.samples, s, live, xb, xo2, xc2, xmu)
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
  uint32_t amask[4];
  second_min = low_bound_fold(a, s, h, cluster, upper_bound, hint, !bad, amask);

  // threshold in accumulator space (yinyang_mfma.hip): a centroid can only matter if acc >= amin
  const float cmaxc = sqrtf(__uint_as_float(a.stats[0])) * 1.000001f;
  const float bmaxc = __uint_as_float(a.stats[1]);
  const float xo = sqrtf(xo2) * 1.0001f, xc = sqrtf(xc2) * 1.0001f;
  float e_mfma = 2.0f * a.eps * (xc * cmaxc + bmaxc) * 1.01f;
  {
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

  const uint32_t nsuper = (a.K_pad + 63u) / 64u;
  const float *biashi = reinterpret_cast<const float *>(reinterpret_cast<const unsigned char *>(a.panelhi) + (size_t)nsuper * SUPB);
  // my i-th piece of super-tile sp (i = PPW: the 64 biases, wave 0).  Linear byte P of the super-tile lands in
  // LDS at P and is fetched from source byte P ^ (((P / ROWB) & SWM) << 4) (lloyd_f16.hip)
  auto issue_piece = [&](uint32_t sp, int buf, int i) {
    if (i == PPW) {
      if (wave == 0)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(biashi + sp * 64u + lane),
                                         (__attribute__((address_space(3))) void *)(uintptr_t)(bias0 + buf * 256), 4, 0, 0);
      return;
    }
    if (i == PPW + 1) {   // the 64 centroids' groups (yy_configure pads the array to whole super-tiles)
      if (wave == 1)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(a.groups + sp * 64u + lane),
                                         (__attribute__((address_space(3))) void *)(uintptr_t)(grp0 + buf * 256), 4, 0, 0);
      return;
    }
    const int p = wave + WV * i;
    if (p >= NP) return;   // wave-uniform
    const unsigned char *src = reinterpret_cast<const unsigned char *>(a.panelhi) + (size_t)sp * SUPB;
    uint32_t P0 = (uint32_t)lane * 16u;
    asm volatile("" : "+v"(P0));
    const uint32_t P = (uint32_t)p * 1024u + P0;
    const uint32_t from = P ^ (((P / ROWB) & SWM) << 4);
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + from),
                                     (__attribute__((address_space(3))) void *)(uintptr_t)(lds0 + buf * SUPB + p * 1024), 16, 0, 0);
  };
  const uint32_t fragbase = lds0 + (uint32_t)col * ROWB + (uint32_t)h * (KS * 16);
  const uint32_t fragswz = (uint32_t)(col & SWM) * 16u;

  // queue of candidates (ascending c)
  uint32_t qc[4] = {0, 0, 0, 0};
  int qn = 0;
  auto flush = [&](auto deep_c) {  // wave-uniform call; deep_c: std::true_type for the wave's last flush
    constexpr bool DEEPF = decltype(deep_c)::value;
    const float *crow[4];
#pragma unroll
    for (int i = 0; i < 4; i++) crow[i] = a.centroids + (size_t)(i < qn ? qc[i] : 0) * D;
    // what the replay below needs of each queued centroid, requested BEFORE the chains run: the group's bound
    // and drift, the centroid's drift (kmeans.cu:637) -- fetched one dependent load after the other inside the
    // replay they were 40 % of a flush
    float lbq[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
      lbq[i] = 0.f;
      if (i < qn) {
        const uint32_t c = qc[i];
        const uint32_t g = a.groups[c];
        lbq[i] = a.bounds[(size_t)len * (1 + g) + s] + (a.gdrifts[g] - a.drifts[(size_t)K * D + c]);
      }
    }
    float dist[4];
    // the fullest queue of the wave sets how many candidate rows are gathered (a row has 0.2 - 2 real ones)
    const int nq = __ballot(qn >= 4) ? 4 : (__ballot(qn >= 3) ? 3 : (__ballot(qn >= 2) ? 2 : 1));
    exact_distance4<NK, METRIC, FAST, DEEPF>(xrow, crow, D, h, col, dist, nq, qn);
#pragma unroll
    for (int i = 0; i < 4; i++) {
      if (i < qn) {
        const uint32_t c = qc[i];
        const float lb = lbq[i];                             // kmeans.cu:637
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

  const bool wave_live = __ballot(live && !bad) != 0ull;
#pragma unroll
  for (int i = 0; i <= PPW + 1; i++) issue_piece(0, 0, i);
  constexpr int DSTR = (2 * KS) / (PPW + 2) > 0 ? (2 * KS) / (PPW + 2) : 1;   // a piece every DSTR k-steps
  for (uint32_t sp = 0; sp < nsuper; sp++) {
    const int buf = sp & 1;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // my pieces of super-tile sp have landed
    __builtin_amdgcn_s_barrier();                      // everybody's have, and everybody is done with sp - 1
    asm volatile("" ::: "memory");
    const bool dma = sp + 1 < nsuper;
    if (dma && !wave_live) {
#pragma unroll
      for (int i = 0; i <= PPW + 1; i++) issue_piece(sp + 1, buf ^ 1, i);
    }
    if (wave_live) {
#pragma unroll
      for (int sub = 0; sub < 2; sub++) {
        const uint32_t tb = fragbase + (uint32_t)buf * SUPB + (uint32_t)sub * (32 * ROWB);
        const uint32_t bb = bias0 + (uint32_t)buf * 256u + (uint32_t)sub * 128u + 16u * h;
        f32x4 b4[4];
#pragma unroll
        for (int g4 = 0; g4 < 4; g4++) b4[g4] = yyh_lds_read4(bb + 32u * g4);
        constexpr int PD = KS < 4 ? KS : 4;   // fragments in flight
        f16x8h fr[PD + 1];
#pragma unroll
        for (int j = 0; j < PD; j++) fr[j] = yyh_frag_issue(tb + ((16u * j) ^ fragswz));
        asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(b4[0]), "+v"(b4[1]), "+v"(b4[2]), "+v"(b4[3]) : "n"(PD));
        f32x16 acc;
#pragma unroll
        for (int g4 = 0; g4 < 4; g4++) {
          acc[4 * g4 + 0] = b4[g4].x; acc[4 * g4 + 1] = b4[g4].y; acc[4 * g4 + 2] = b4[g4].z; acc[4 * g4 + 3] = b4[g4].w;
        }
#pragma unroll
        for (int j = 0; j < KS; j++) {
          if (j + PD < KS) fr[(j + PD) % (PD + 1)] = yyh_frag_issue(tb + ((16u * (j + PD)) ^ fragswz));
          const int behind = (KS - 1 - j) < PD ? (KS - 1 - j) : PD;
          f16x8h &f = fr[j % (PD + 1)];
          if (behind == 4) yyh_frag_wait<4>(f);
          else if (behind == 3) yyh_frag_wait<3>(f);
          else if (behind == 2) yyh_frag_wait<2>(f);
          else if (behind == 1) yyh_frag_wait<1>(f);
          else yyh_frag_wait<0>(f);
          acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(f, xh[j], acc, 0, 0, 0);
          {
            const int slot = sub * KS + j;   // compile-time after unrolling
            if (dma && slot % DSTR == 0 && slot / DSTR <= PPW + 1) issue_piece(sp + 1, buf ^ 1, slot / DSTR);
          }
        }
        if (sub == 1 && dma) {   // pieces the slots did not cover (very short rows)
#pragma unroll
          for (int i = (2 * KS - 1) / DSTR + 1; i <= PPW + 1; i++) issue_piece(sp + 1, buf ^ 1, i);
        }
        // the sub-tile's best score first: most sub-tiles hold no candidate for any row of the wave
        bool some = false;
        if (live && !bad) {
          const float m0 = __builtin_fmaxf(__builtin_fmaxf(acc[0], acc[1]), acc[2]);
          const float m1 = __builtin_fmaxf(__builtin_fmaxf(acc[3], acc[4]), acc[5]);
          const float m2 = __builtin_fmaxf(__builtin_fmaxf(acc[6], acc[7]), acc[8]);
          const float m3 = __builtin_fmaxf(__builtin_fmaxf(acc[9], acc[10]), acc[11]);
          const float m4 = __builtin_fmaxf(__builtin_fmaxf(acc[12], acc[13]), acc[14]);
          const float m5 = __builtin_fmaxf(__builtin_fmaxf(m0, m1), acc[15]);
          const float m6 = __builtin_fmaxf(__builtin_fmaxf(m2, m3), m4);
          some = __builtin_fmaxf(m5, m6) >= amin;
        }
        if (__ballot(some) != 0ull) {
          uint32_t m16 = 0;
          if (some) {
#pragma unroll
            for (int r = 0; r < 16; r++)
              if (acc[r] >= amin) m16 |= 1u << r;
          }
          const uint32_t pm = __shfl_xor(m16, 32);
          const uint32_t m0 = h ? pm : m16, m1 = h ? m16 : pm;
          uint32_t rowmask = 0;
#pragma unroll
          for (int g4 = 0; g4 < 4; g4++)
            rowmask |= (((m0 >> (4 * g4)) & 0xFu) << (8 * g4)) | (((m1 >> (4 * g4)) & 0xFu) << (8 * g4 + 4));
          while (__ballot(rowmask != 0u) != 0ull) {
            if (__ballot(qn == 4) != 0ull) flush(std::false_type());  // some row's queue is full
            const bool active = rowmask != 0u;
            const uint32_t rho = active ? (uint32_t)__ffs((int)rowmask) - 1u : 0u;
            rowmask &= rowmask - 1u;
            if (active) {
              const uint32_t c = sp * 64u + (uint32_t)sub * 32u + rho;
              const uint32_t g = yyh_lds_read1(grp0 + (uint32_t)buf * 256u + ((uint32_t)sub * 32u + rho) * 4u);
              if (g < G && c != cluster) {  // g >= G: NaN centroid or padding
                bool a_group;               // (a): group bound >= upper bound
                if (G <= 128u) {
                  const uint32_t word = g < 32u ? amask[0] : (g < 64u ? amask[1] : (g < 96u ? amask[2] : amask[3]));
                  a_group = (word >> (g & 31u)) & 1u;
                } else {
                  a_group = a.bounds[(size_t)len * (1 + g) + s] >= upper_bound;
                }
                if (!a_group) {  // else an (a) centroid: an event if its bound is <= S'
#pragma unroll
                  for (int i = 0; i < 4; i++)
                    if (i == qn) qc[i] = c;
                  qn++;
                }
              }
            }
          }
        }
      }
    }
  }
  if (wave_live && __ballot(qn > 0) != 0ull) flush(std::true_type());
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
    }
    base = __shfl(base, 0);
    if (mine && bad) a.flag_rows[base + (uint32_t)__popcll(fm & ((1ull << lane) - 1ull))] = s;
  }
  {  // statistics (not part of the reference's state), striped over 64 cache lines by block number and summed
     // by the reader (Engine::yy_hint_stats): as three to seven atomics per wave on the counters' own line they
     // were 20 % of this kernel (11.1 -> 8.8 ms per 8M rows without them) -- 1M same-line atomics per launch
    const uint32_t nmine = (uint32_t)__popcll(__ballot(mine));
    const uint32_t nbad = (uint32_t)__popcll(fm);
    uint32_t nwhy[4];
#pragma unroll
    for (int w = 0; w < 4; w++) nwhy[w] = (uint32_t)__popcll(__ballot(mine && why == (uint32_t)(w + 1)));
    if (lane == 0) {
      uint32_t *st = a.stat_stripes + (blockIdx.x & 63u) * 16u;
      atomicAdd(&st[6], nmine);
      if (nbad) atomicAdd(&st[7], nbad);
#pragma unroll
      for (int w = 0; w < 4; w++)
        if (nwhy[w]) atomicAdd(&st[8 + w], nwhy[w]);
    }
  }
}

// ---------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------
bool yy_hint_supported(uint32_t DP) { return DP >= 16 && DP <= 256; }

template <int DP, int METRIC>
static hipError_t launch_hint_t(const YyArgs &a, hipStream_t st) {
  const size_t lds_bytes = 2 * 64 * DP * 2 + 512;
  if (lds_bytes > 65536) {   // (per launch: the attribute belongs to the current device's copy of the kernel)
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&yy_hint_kernel<DP, METRIC, true>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e == hipSuccess)
      e = hipFuncSetAttribute(reinterpret_cast<const void *>(&yy_hint_kernel<DP, METRIC, false>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e != hipSuccess) return e;
  }
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
  const size_t lds_bytes = 2 * 64 * DP * 2 + 1024;
  if (lds_bytes > 65536) {   // (per launch: the attribute belongs to the current device's copy of the kernel)
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&yy_local_hint_kernel<DP, METRIC, true>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e == hipSuccess)
      e = hipFuncSetAttribute(reinterpret_cast<const void *>(&yy_local_hint_kernel<DP, METRIC, false>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e != hipSuccess) return e;
  }
  if (a.D == (uint32_t)DP)
    hipLaunchKernelGGL((yy_local_hint_kernel<DP, METRIC, true>), dim3(grid), dim3(256), lds_bytes, st, a);
  else
    hipLaunchKernelGGL((yy_local_hint_kernel<DP, METRIC, false>), dim3(grid), dim3(256), lds_bytes, st, a);
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
