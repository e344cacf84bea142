// lloyd_f16.hip -- the Lloyd assignment filter (reference: src/kmeans.cu:293-364) on the f16 matrix
// cores, for fp32 rows AND for the fp16x2 path's half rows (src/fp_abstraction.h:100-182).
//
// The filter only has to produce scores with a RIGOROUS error bound (lloyd.hip: rows it cannot
// decide go to the exact kernels), so nothing forces it onto the f32 MFMA (64 FLOP/clk/SIMD, 1/16
// of the f16 rate).  A centred fp32 operand split into two halves, a = a_hi + a_lo + r with
// |r| <= 2^-22 |a| + 2^-25, keeps 22 of its 24 significand bits, the products of halves are exact
// in the fp32 accumulator, and the residual is far below the bound the filter already carries.
// fp16x2 semantics of this implementation are the fp32 reference arithmetic on the
// half VALUES, so the half-row variant reproduces the fp32 path's decisions on the widened rows
// (the exact kernels read the widened copy): same contract, half the HBM bytes.
//
//   * rows are read as fp32 (HALF_ROWS = false) or as HALVES (512 B instead of 1 KB at D = 256);
//   * operands are CENTRED in fp32 (x' = x - mu, c' = c - mu: the centred bound, lloyd.hip) and
//     split into two halves each, a = a_hi + a_lo + r, |r| <= 2^-22 |a| + 2^-25, so that
//         x'.c' ~= x_hi.c_hi + x_hi.c_lo + x_lo.c_hi
//     with every product exact in the fp32 accumulator of v_mfma_f32_32x32x16_f16: three f16 MFMAs
//     (16 features each, 32 cycles) replace eight f32 MFMAs (2 features each, 64 cycles) -- 5.3x
//     fewer matrix-pipe cycles per (row, centroid) pair;
//   * the dropped lo.lo term and the split residuals are part of the error bound E.
//
// Layout: identical to the f32 filter (4 waves x 32 rows per block, 32-centroid tiles double
// buffered in LDS, the lower half-wave contracts features [0, DP/2), the upper [DP/2, DP)); a panel
// row holds DP hi halves followed by DP lo halves = the same 4*DP bytes as an f32 row.
#include <hip/hip_fp16.h>

#include "exact.hpp"
#include "filter_common.hpp"
#include "kernels.hpp"

namespace kmx {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// c' = c - mu split into halves: panel16[c] = [hi(c'_0..DP-1) | lo(c'_0..DP-1)]; zero rows for
// non-finite / padding centroids (their bias is -inf in the shared bias array).
__global__ void centroid_panel16_kernel(const float *__restrict__ centroids, uint32_t K, uint32_t D, uint32_t K_pad,
                                        uint32_t DP, const uint32_t *__restrict__ finite,
                                        const float *__restrict__ mu, _Float16 *__restrict__ panel16) {
  const uint32_t c = blockIdx.x;
  if (c >= K_pad) return;
  _Float16 *dst = panel16 + (size_t)c * 2 * DP;
  const bool ok = c < K && finite[c];
  for (uint32_t f = threadIdx.x; f < DP; f += blockDim.x) {
    float v = 0.f;
    if (ok && f < D) v = centroids[(size_t)c * D + f] - mu[f];
    const _Float16 hi = (_Float16)v;
    const _Float16 lo = (_Float16)(v - (float)hi);
    dst[f] = hi;
    dst[DP + f] = lo;
  }
}

template <int DP, bool HALF_ROWS, bool FAST>
__global__ __launch_bounds__(256, 2) void lloyd_filter_f16_kernel(
    const void *__restrict__ rows, uint32_t N, uint32_t D, const float *__restrict__ panel16f,
    const float *__restrict__ bias, const float *__restrict__ mu, uint32_t K_pad, uint32_t K,
    const uint32_t *__restrict__ stats, float eps, float tie_slack, uint32_t *__restrict__ assignments,
    uint32_t *__restrict__ assignments_prev, uint32_t *__restrict__ flagged, uint32_t *__restrict__ pairs,
    uint32_t *__restrict__ counters, const uint32_t *__restrict__ row_list, const uint32_t *__restrict__ n_list) {
  constexpr int NKH = DP / 2;          // features per half-wave
  constexpr int KS = NKH / 8;          // k-steps (8 features per lane per MFMA)
  constexpr int LDW = DP + 4;          // padded LDS row in 4-byte words (row = 2*DP halves)
  constexpr int TILE = 32 * LDW;
  constexpr int NST = (8 * DP + 255) / 256;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  auto tile_ptr = [&](int buf) { return lds + buf * TILE; };
  auto bias_ptr = [&](int buf) { return lds + 2 * TILE + buf * 32; };

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, col = lane & 31, h = lane >> 5;
  // all rows (row_list == nullptr: one 128-row group per block) or the rows the coarse stage could not
  // decide (a device-side list: the grid strides over its 128-row groups)
  const uint32_t total = row_list ? *n_list : N;
  for (uint32_t group = blockIdx.x; (size_t)group * 128u < total; group += gridDim.x) {
  const uint32_t pi = group * 128u + wave * 32u + col;
  const bool live = pi < total;
  const uint32_t s = row_list ? (live ? row_list[pi] : 0u) : pi;

  // ---- B operand: my half row, centred in fp32, split into hi / lo halves ----
  f16x8 xhi[KS], xlo[KS];
  float xn2 = 0.f, xo2 = 0.f, x0 = 0.f;
  {
    const size_t row = (size_t)(live ? s : 0);
    const float *m = mu + h * NKH;
#pragma unroll
    for (int j = 0; j < KS; j++) {
      float xv[8];
      if (FAST && HALF_ROWS) {
        const f16x8 raw = reinterpret_cast<const f16x8 *>(reinterpret_cast<const _Float16 *>(rows) + row * DP + h * NKH)[j];
#pragma unroll
        for (int q = 0; q < 8; q++) xv[q] = (float)raw[q];
      } else if (FAST) {
        const f32x4 *src = reinterpret_cast<const f32x4 *>(reinterpret_cast<const float *>(rows) + row * DP + h * NKH);
        const f32x4 a = src[2 * j], b = src[2 * j + 1];
        xv[0] = a.x; xv[1] = a.y; xv[2] = a.z; xv[3] = a.w;
        xv[4] = b.x; xv[5] = b.y; xv[6] = b.z; xv[7] = b.w;
      } else {
#pragma unroll
        for (int q = 0; q < 8; q++) {
          const uint32_t f = h * NKH + 8 * j + q;
          float v = 0.f;
          if (f < D) v = HALF_ROWS ? (float)reinterpret_cast<const _Float16 *>(rows)[row * D + f]
                                   : reinterpret_cast<const float *>(rows)[row * D + f];
          xv[q] = v;
        }
      }
      f16x8 hi, lo;
#pragma unroll
      for (int q = 0; q < 8; q++) {
        const bool on = live && (FAST || h * NKH + 8 * j + q < (int)D);
        const float x = on ? xv[q] : 0.f;
        const float xc = on ? x - m[8 * j + q] : 0.f;
        const _Float16 a = (_Float16)xc;
        hi[q] = a;
        lo[q] = (_Float16)(xc - (float)a);
        xo2 = fmaf(x, x, xo2);
        xn2 = fmaf(xc, xc, xn2);
        if (j == 0 && q == 0) x0 = live ? xv[q] : 0.f;
      }
      xhi[j] = hi;
      xlo[j] = lo;
    }
  }
  xn2 += __shfl_xor(xn2, 32);
  xo2 += __shfl_xor(xo2, 32);
  x0 = __shfl(x0, col);  // feature 0 lives in the lower half-wave
  const bool insane = (x0 != x0);  // kmeans.cu:312

  // ---- staging of panel tiles: byte-identical to the f32 filter's ----
  f32x4 stage[NST];
  float bstage = 0.f;
  auto stage_load = [&](uint32_t tile) {
    const float *src = panel16f + (size_t)tile * 32 * DP;
#pragma unroll
    for (int i = 0; i < NST; i++) {
      const int q = tid + i * 256;
      if (q < 8 * DP) stage[i] = reinterpret_cast<const f32x4 *>(src)[q];
    }
    if (tid < 32) bstage = bias[tile * 32 + tid];
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
    if (tid < 32) bias_ptr(buf)[tid] = bstage;
  };

  const uint32_t ntiles = K_pad / 32;
  stage_load(0);
  stage_store(0);
  __syncthreads();

  float v1 = -INFINITY, v2 = -INFINITY, v3 = -INFINITY;
  uint32_t c1 = 0xFFFFFFFFu, c2 = 0xFFFFFFFFu;
  for (uint32_t t = 0; t < ntiles; t++) {
    const int buf = t & 1;
    if (t + 1 < ntiles) stage_load(t + 1);
    // ONE accumulator on purpose: three independent ones (one per product kind) measured 15.4 ms
    // against 12.9 ms -- under the f16 matrix load the chip is power limited (1.75 GHz, PMC in
    // profiles/), extra matrix-level parallelism only lowers the clock further
    f32x16 acc0;
    {
      const float *bb = bias_ptr(buf) + 4 * h;
#pragma unroll
      for (int g = 0; g < 4; g++) {
        const f32x4 b4 = *reinterpret_cast<const f32x4 *>(bb + 8 * g);
        acc0[4 * g + 0] = b4.x; acc0[4 * g + 1] = b4.y; acc0[4 * g + 2] = b4.z; acc0[4 * g + 3] = b4.w;
      }
    }
    // my centroid row of the tile: hi halves at [0, DP), lo halves at [DP, 2 DP).  Fragments are
    // read TWO k-steps (6 MFMAs) ahead of their use and the order is pinned: left alone, hipcc sinks
    // each ds_read_b128 right in front of its first MFMA and the LDS latency lands on the pipe
    const _Float16 *arow = reinterpret_cast<const _Float16 *>(tile_ptr(buf) + col * LDW) + h * NKH;
    auto frag = [&](int j, int lo) { return *reinterpret_cast<const f16x8 *>(arow + lo * DP + 8 * j); };
    if constexpr (KS >= 2) {
      f16x8 h0 = frag(0, 0), l0 = frag(0, 1), h1 = frag(1, 0), l1 = frag(1, 1);
#pragma unroll
      for (int j = 0; j < KS; j++) {
        f16x8 h2 = h1, l2 = l1;
        if (j + 2 < KS) {
          h2 = frag(j + 2, 0);
          l2 = frag(j + 2, 1);
        }
        __builtin_amdgcn_sched_barrier(0);
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(h0, xhi[j], acc0, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(l0, xhi[j], acc0, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(h0, xlo[j], acc0, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        h0 = h1; l0 = l1;
        h1 = h2; l1 = l2;
      }
    } else {
      acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(frag(0, 0), xhi[0], acc0, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(frag(0, 1), xhi[0], acc0, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(frag(0, 0), xlo[0], acc0, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 16; r++) {
      const float v = acc0[r];
      const uint32_t code = t * 16u + r;
      // sorted insert with 3 value ops (v_med3 / v_max ignore a NaN operand, like the strict
      // compares do) + the two index selects
      const bool g1 = v > v1, g2 = v > v2;
      v3 = __builtin_amdgcn_fmed3f(v2, v3, v);
      c2 = g1 ? c1 : (g2 ? code : c2);
      v2 = __builtin_amdgcn_fmed3f(v1, v2, v);
      c1 = g1 ? code : c1;
      v1 = fmaxf(v1, v);
    }
    if (t + 1 < ntiles) stage_store(buf ^ 1);
    __syncthreads();
  }

  // ---- decide: the f32 filter's bound plus the hi/lo split terms ----
  //   accumulation of 3 DP exact products + bias in fp32:       gamma_{3DP+1} (||x'|| C'max + B'max)
  //   dropped lo.lo and split residuals (|r| <= 2^-22|a| + 2^-25): 3 * 2^-22 ||x'|| C'max
  //                                                              + 2^-25 sqrt(DP) (||x'|| + C'max)
  const float cmaxc = sqrtf(__uint_as_float(stats[0])) * 1.000001f;
  const float bmaxc = __uint_as_float(stats[1]);
  const float cmaxo = sqrtf(__uint_as_float(stats[2])) * 1.000001f;
  const float xn = sqrtf(xn2) * 1.0001f, xo = sqrtf(xo2) * 1.0001f;
  const float u = 5.9604645e-8f;
  const float e_mfma = 2.0f * (3.0f * eps) * (xn * cmaxc + bmaxc) + 12.0f * u * xn * cmaxc +
                       2.9802322e-8f * sqrtf((float)DP) * (xn + cmaxc);
  const float e_ref = u * (12.0f * xo * cmaxo + 4.0f * cmaxo * cmaxo);
  const float thr = 2.0f * (e_mfma + e_ref) * 1.001f + tie_slack;
  // filter_finish treats a row as present iff s < N: present rows carry their index, absent ones N
  filter_finish(v1, v2, v3, c1, c2, h, lane, live ? s : N, N, K, insane, thr, assignments, assignments_prev, flagged,
                pairs, counters);
  __syncthreads();  // the LDS tiles are reused by the next group
  }
}

// ---------------------------------------------------------------------------------------
// Stage 1 of the default filter: ONE f16 MFMA per 16 features (hi.hi only).  Under the f16 matrix
// load the chip is power limited (profiles/: 1.75 GHz), so the lever is fewer matrix operations:
// the coarse scores carry |error| <= E_c ~ 2^-10 ||x'|| C'max, enough to decide the rows whose
// best / second-best gap exceeds 2 E_c (the large majority); only the others go through the
// three-product kernel above.
// ---------------------------------------------------------------------------------------
constexpr int kCoarseThreads = 512;  // 8 waves share a super-tile: 2 blocks/CU = 4 waves per SIMD
template <int DP, bool HALF_ROWS, bool FAST>
__global__ __launch_bounds__(kCoarseThreads, 2) void lloyd_coarse_kernel(
    const void *__restrict__ rows, uint32_t N, uint32_t D, const float *__restrict__ panelhi,
    const float *__restrict__ bias, const float *__restrict__ mu, uint32_t K_pad, uint32_t K,
    const uint32_t *__restrict__ stats, float eps, float tie_slack, uint32_t *__restrict__ assignments,
    uint32_t *__restrict__ assignments_prev, uint32_t *__restrict__ undecided, uint32_t *__restrict__ counters) {
  constexpr int NKH = DP / 2;
  constexpr int KS = NKH / 8;
  constexpr int LDW = DP / 2 + 4;      // padded LDS row in 4-byte words (row = DP halves)
  constexpr int TILE = 32 * LDW;
  constexpr int BT = kCoarseThreads;
  constexpr int NST = (4 * DP + BT - 1) / BT;   // 16-byte pieces per thread per tile
  extern __shared__ __attribute__((aligned(16))) float lds[];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, col = lane & 31, h = lane >> 5;
  const uint32_t s = blockIdx.x * (uint32_t)(BT / 2) + wave * 32u + col;
  const bool live = s < N;

  f16x8 xhi[KS];
  float xn2 = 0.f, xo2 = 0.f, x0 = 0.f;
  {
    const size_t row = (size_t)(live ? s : 0);
    const float *m = mu + h * NKH;
#pragma unroll
    for (int j = 0; j < KS; j++) {
      float xv[8];
      if (FAST && HALF_ROWS) {
        const f16x8 raw = reinterpret_cast<const f16x8 *>(reinterpret_cast<const _Float16 *>(rows) + row * DP + h * NKH)[j];
#pragma unroll
        for (int q = 0; q < 8; q++) xv[q] = (float)raw[q];
      } else if (FAST) {
        const f32x4 *src = reinterpret_cast<const f32x4 *>(reinterpret_cast<const float *>(rows) + row * DP + h * NKH);
        const f32x4 a = src[2 * j], b = src[2 * j + 1];
        xv[0] = a.x; xv[1] = a.y; xv[2] = a.z; xv[3] = a.w;
        xv[4] = b.x; xv[5] = b.y; xv[6] = b.z; xv[7] = b.w;
      } else {
#pragma unroll
        for (int q = 0; q < 8; q++) {
          const uint32_t f = h * NKH + 8 * j + q;
          float v = 0.f;
          if (f < D) v = HALF_ROWS ? (float)reinterpret_cast<const _Float16 *>(rows)[row * D + f]
                                   : reinterpret_cast<const float *>(rows)[row * D + f];
          xv[q] = v;
        }
      }
      f16x8 hi;
#pragma unroll
      for (int q = 0; q < 8; q++) {
        const bool on = live && (FAST || h * NKH + 8 * j + q < (int)D);
        const float x = on ? xv[q] : 0.f;
        const float xc = on ? x - m[8 * j + q] : 0.f;
        hi[q] = (_Float16)xc;
        xo2 = fmaf(x, x, xo2);
        xn2 = fmaf(xc, xc, xn2);
        if (j == 0 && q == 0) x0 = live ? xv[q] : 0.f;
      }
      xhi[j] = hi;
    }
  }
  xn2 += __shfl_xor(xn2, 32);
  xo2 += __shfl_xor(xo2, 32);
  x0 = __shfl(x0, col);
  const bool insane = (x0 != x0);  // kmeans.cu:312

  // A tile's matrix work is only 16 MFMAs (512 cycles), so the barrier and the staging round trip
  // weigh as much as the tile itself: stage SUPER-TILES of 64 centroids (two 32-row MFMA tiles) per
  // barrier, double buffered (2 x 64 x (DP + 8) bytes of LDS, 2 blocks per CU).
  constexpr int NSS = 2 * NST;  // 16-byte pieces per thread per super-tile
  f32x4 stage[NSS];
  float bstage = 0.f;
  const uint32_t ntiles = K_pad / 32, nsuper = (ntiles + 1) / 2;
  auto sup_tile = [&](int buf, int sub) { return lds + (buf * 2 + sub) * TILE; };
  auto sup_bias = [&](int buf) { return lds + 4 * TILE + buf * 64; };
  auto stage_load = [&](uint32_t sp) {
    const float *src = panelhi + (size_t)sp * 64 * (DP / 2);
    const int limit = (2 * sp + 1 < ntiles) ? 8 * DP : 4 * DP;  // the last super-tile may hold one tile
#pragma unroll
    for (int i = 0; i < NSS; i++) {
      const int q = tid + i * BT;
      if (q < limit) stage[i] = reinterpret_cast<const f32x4 *>(src)[q];
    }
    if (tid < 64) bstage = (sp * 64 + tid < K_pad) ? bias[sp * 64 + tid] : -INFINITY;
  };
  auto stage_store = [&](int buf) {
#pragma unroll
    for (int i = 0; i < NSS; i++) {
      const int q = tid + i * BT;
      if (q < 8 * DP) {
        const int row = q / (DP / 8), c4 = q % (DP / 8);  // row 0..63
        *reinterpret_cast<f32x4 *>(sup_tile(buf, row >> 5) + (row & 31) * LDW + c4 * 4) = stage[i];
      }
    }
    if (tid < 64) sup_bias(buf)[tid] = bstage;
  };

  float v1 = -INFINITY, v2 = -INFINITY;
  uint32_t c1 = 0xFFFFFFFFu;
  auto compute_tile = [&](uint32_t t, int buf, int sub) {
    f32x16 acc;
    {
      const float *bb = sup_bias(buf) + 32 * sub + 4 * h;
#pragma unroll
      for (int g = 0; g < 4; g++) {
        const f32x4 b4 = *reinterpret_cast<const f32x4 *>(bb + 8 * g);
        acc[4 * g + 0] = b4.x; acc[4 * g + 1] = b4.y; acc[4 * g + 2] = b4.z; acc[4 * g + 3] = b4.w;
      }
    }
    const _Float16 *arow = reinterpret_cast<const _Float16 *>(sup_tile(buf, sub) + col * LDW) + h * NKH;
#pragma unroll
    for (int j = 0; j < KS; j++)
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(*reinterpret_cast<const f16x8 *>(arow + 8 * j), xhi[j], acc, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 16; r++) {
      const float v = acc[r];
      const bool g1 = v > v1;
      v2 = __builtin_amdgcn_fmed3f(v1, v2, v);
      c1 = g1 ? t * 16u + r : c1;
      v1 = fmaxf(v1, v);
    }
  };

  stage_load(0);
  stage_store(0);
  __syncthreads();
  for (uint32_t sp = 0; sp < nsuper; sp++) {
    const int buf = sp & 1;
    if (sp + 1 < nsuper) stage_load(sp + 1);
    compute_tile(2 * sp, buf, 0);
    if (2 * sp + 1 < ntiles) compute_tile(2 * sp + 1, buf, 1);
    if (sp + 1 < nsuper) stage_store(buf ^ 1);
    __syncthreads();
  }
  // merge the half-waves (same sample, disjoint centroid rows)
  uint32_t i1 = 0xFFFFFFFFu;
  if (c1 != 0xFFFFFFFFu) {
    const uint32_t r = c1 & 15u;
    i1 = (c1 >> 4) * 32u + (r & 3u) + 8u * (r >> 2) + 4u * h;
  }
  {
    const float pv1 = __shfl_xor(v1, 32), pv2 = __shfl_xor(v2, 32);
    const uint32_t pi1 = __shfl_xor(i1, 32);
    const bool g = pv1 > v1;
    const float second = fmaxf(g ? v1 : pv1, fmaxf(v2, pv2));  // fmax ignores a NaN operand
    i1 = g ? pi1 : i1;
    v1 = g ? pv1 : v1;
    v2 = second;
  }
  // |coarse score - reference score| <= E_c: the f32-accumulated hi.hi products (gamma_{DP+1}), the
  // dropped lo terms (|a_lo| <= 2^-11 |a|: (2^-10 + 2^-22) ||x'|| C'max) and half underflow, + E_ref
  const float cmaxc = sqrtf(__uint_as_float(stats[0])) * 1.000001f;
  const float bmaxc = __uint_as_float(stats[1]);
  const float cmaxo = sqrtf(__uint_as_float(stats[2])) * 1.000001f;
  const float xn = sqrtf(xn2) * 1.0001f, xo = sqrtf(xo2) * 1.0001f;
  const float u = 5.9604645e-8f;
  const float e_c = 2.0f * eps * (xn * cmaxc + bmaxc) + 9.8e-4f * xn * cmaxc +
                    6e-8f * sqrtf((float)DP) * (xn + cmaxc);
  const float e_ref = u * (12.0f * xo * cmaxo + 4.0f * cmaxo * cmaxo);
  const float thr = 2.0f * (e_c + e_ref) * 1.001f + tie_slack;
  const bool certain = insane || ((v1 - v2) > thr);  // NaN gap / thr => not certain
  const bool mine = (h == 0) && live;
  bool changed = false;
  if (mine && certain) changed = commit_row(s, insane ? K : i1, assignments, assignments_prev);
  const bool und = mine && !certain;
  const unsigned long long cm = __ballot(changed), um = __ballot(und);
  if (lane == 0 && cm) atomicAdd(&counters[0], (uint32_t)__popcll(cm));
  if (um) {
    uint32_t base = 0;
    if (lane == 0) base = atomicAdd(&counters[4], (uint32_t)__popcll(um));
    base = __shfl(base, 0);
    if (und) undecided[base + (uint32_t)__popcll(um & ((1ull << lane) - 1ull))] = s;
  }
}

// ---------------------------------------------------------------------------------------
// Stage 1, second generation.  PMC on the kernel above (profiles/r1e_pmc_summary.json): matrix pipe
// busy 22-28 %, waves parked 65 % of their cycles.  Three structural causes, three changes:
//   * every wave re-read the whole centroid panel from LDS for only 32 rows (1 KB of ds_read per
//     MFMA = half the LDS bandwidth at full matrix rate): a wave now owns 64 rows (two B-operand
//     sets), so each A fragment feeds two MFMAs;
//   * the top-2 bookkeeping of a tile (VALU) ran after its MFMAs with the matrix pipe idle: the
//     accumulators are double buffered and the bookkeeping of tile t-1 is interleaved with the MFMAs
//     of tile t; it is 3 ops per score instead of 4 -- the accumulator register number travels in
//     the low 4 mantissa bits of the score (<= 16 ulp, part of the bound), the tile index is noted
//     once per tile;
//   * staging went global -> 32 VGPRs -> ds_write_b128: tiles now arrive by LDS-DMA
//     (global_load_lds_dwordx4, no registers, no LDS-write issue slots).  The DMA writes lane-linear,
//     so the bank swizzle (16-byte chunk j of row r sits in slot j ^ (r & 15) of its half row) is
//     applied to the SOURCE address and again by the fragment reads.
// ---------------------------------------------------------------------------------------
constexpr int kCoarse2Threads = 256;   // 4 waves x 64 rows; 2 blocks per CU (2 x 66 KB of LDS)
template <int DP, bool HALF_ROWS, bool FAST>
__global__ __launch_bounds__(kCoarse2Threads, 2) void lloyd_coarse2_kernel(
    const void *__restrict__ rows, uint32_t N, uint32_t D, const float *__restrict__ panelhi,
    const float *__restrict__ bias, const float *__restrict__ mu, uint32_t K_pad, uint32_t K,
    const uint32_t *__restrict__ stats, float eps, float tie_slack, uint32_t *__restrict__ assignments,
    uint32_t *__restrict__ assignments_prev, uint32_t *__restrict__ undecided, uint32_t *__restrict__ counters) {
  constexpr int NKH = DP / 2;
  constexpr int KS = NKH / 8;               // k-steps = 16-byte chunks per half row
  constexpr int ROWB = DP * 2;              // bytes of one LDS row (DP hi halves)
  constexpr int SUPB = 64 * ROWB;           // one super-tile: 64 centroids
  constexpr int NP = SUPB / 1024;           // 1-KB LDS-DMA pieces per super-tile
  constexpr int SWM = (KS < 16 ? KS : 16) - 1;
  typedef __attribute__((address_space(3))) unsigned char lds_byte;
  extern __shared__ __attribute__((aligned(1024))) unsigned char lds2[];
  // raw LDS byte addresses (the fragment address is built with XOR: needs the 1-KB aligned base)
  const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_byte *)lds2;
  if (lds0 & 1023u) __builtin_trap();
  const uint32_t bias0 = lds0 + 2 * SUPB;   // 2 x 64 floats

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), col = lane & 31, h = lane >> 5;
  const uint32_t sA = blockIdx.x * 256u + wave * 64u + col, sB = sA + 32u;
  const bool liveA = sA < N, liveB = sB < N;

  f16x8 xa[KS], xb[KS];
  float xn2a = 0.f, x0a = 0.f, xn2b = 0.f, x0b = 0.f;
  // both rows of a lane per k-step, sharing the mean chunk.  Lanes without a row read row 0: an MFMA
  // column only feeds its own outputs and theirs are never committed, so nothing is masked.
  auto load_chunk = [&](uint32_t s, bool live, int j, float (&xv)[8]) {
    const size_t row = (size_t)(live ? s : 0);
    if (FAST && HALF_ROWS) {
      const f16x8 raw = reinterpret_cast<const f16x8 *>(reinterpret_cast<const _Float16 *>(rows) + row * DP + h * NKH)[j];
#pragma unroll
      for (int q = 0; q < 8; q++) xv[q] = (float)raw[q];
    } else if (FAST) {
      const f32x4 *src = reinterpret_cast<const f32x4 *>(reinterpret_cast<const float *>(rows) + row * DP + h * NKH);
      const f32x4 a = src[2 * j], b = src[2 * j + 1];
      xv[0] = a.x; xv[1] = a.y; xv[2] = a.z; xv[3] = a.w;
      xv[4] = b.x; xv[5] = b.y; xv[6] = b.z; xv[7] = b.w;
    } else {
#pragma unroll
      for (int q = 0; q < 8; q++) {
        const uint32_t f = h * NKH + 8 * j + q;
        float v = 0.f;
        if (f < D) v = HALF_ROWS ? (float)reinterpret_cast<const _Float16 *>(rows)[row * D + f]
                                 : reinterpret_cast<const float *>(rows)[row * D + f];
        xv[q] = v;
      }
    }
  };
  auto centre = [&](const float (&xv)[8], const float (&mm)[8], f16x8 &hi, float &xn2) {
#pragma unroll
    for (int q = 0; q < 8; q++) {
      const float xc = xv[q] - mm[q];
      hi[q] = (_Float16)xc;
      xn2 = fmaf(xc, xc, xn2);
    }
  };
  auto load_rows = [&]() {
    const f32x4 *mv = reinterpret_cast<const f32x4 *>(mu + h * NKH);  // DP floats, zero beyond D
#pragma unroll
    for (int j = 0; j < KS; j++) {
      float va[8], vb[8], mm[8];
      load_chunk(sA, liveA, j, va);
      load_chunk(sB, liveB, j, vb);
      const f32x4 m0 = mv[2 * j], m1 = mv[2 * j + 1];
      mm[0] = m0.x; mm[1] = m0.y; mm[2] = m0.z; mm[3] = m0.w;
      mm[4] = m1.x; mm[5] = m1.y; mm[6] = m1.z; mm[7] = m1.w;
      centre(va, mm, xa[j], xn2a);
      centre(vb, mm, xb[j], xn2b);
      if (j == 0) { x0a = va[0]; x0b = vb[0]; }
      asm volatile("" : "+v"(xa[j]), "+v"(xb[j]));  // convert NOW: hipcc parks the fp32 values in scratch otherwise
      // keep the loads of later k-steps behind the conversions of this group: hoisted all at once
      // they need far more registers than the kernel has
      if ((j & 1) == 1) __builtin_amdgcn_sched_barrier(0);
    }
    xn2a += __shfl_xor(xn2a, 32); x0a = __shfl(x0a, col);
    xn2b += __shfl_xor(xn2b, 32); x0b = __shfl(x0b, col);
  };

  // ---- LDS-DMA staging of super-tile sp into buffer buf ----
  const uint32_t nsuper = (K_pad + 63) / 64;
  const float *biashi = reinterpret_cast<const float *>(reinterpret_cast<const unsigned char *>(panelhi) + (size_t)nsuper * SUPB);
  auto stage_issue = [&](uint32_t sp, int buf) {
    // linear byte P of the super-tile image lands in LDS at P; it is fetched from source byte
    // P ^ (((P / ROWB) & SWM) << 4): the 16-byte chunk index XORed with the row's low bits (inside a
    // half row, SWM < KS).  Recomputed per call from one opaque register -- as loop invariants the
    // per-piece addresses cost 30 VGPRs the MFMA loop needs.
    uint32_t P0 = (uint32_t)lane * 16u;
    asm volatile("" : "+v"(P0));
    const unsigned char *src = reinterpret_cast<const unsigned char *>(panelhi) + (size_t)sp * SUPB;
#pragma unroll
    for (int i = 0; i < (NP + 3) / 4; i++) {
      const int p = i * 4 + wave;
      if (NP % 4 == 0 || p < NP) {
        const uint32_t P = (uint32_t)p * 1024u + P0;
        const uint32_t from = P ^ (((P / ROWB) & SWM) << 4);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + from),
                                         (__attribute__((address_space(3))) void *)(uintptr_t)(lds0 + buf * SUPB + p * 1024), 16, 0, 0);
      }
    }
    // the 64 biases of the super-tile (clamped copy behind the panel): one 4-byte DMA by wave 0.  No
    // ordinary global load lives in the loop: hipcc waits vmcnt(0) at its first use, draining the DMA
    if (wave == 0)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(biashi + sp * 64u + lane),
                                       (__attribute__((address_space(3))) void *)(uintptr_t)(bias0 + buf * 256), 4, 0, 0);
  };

  stage_issue(0, 0);
  load_rows();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  float v1a = -INFINITY, v2a = -INFINITY, v1b = -INFINITY, v2b = -INFINITY;
  uint32_t tba = 0, tbb = 0;
  // fragment address of k-step j: rowbase ^ swizzle ^ (16 j); (row, half) part fixed per lane
  const uint32_t fragbase = lds0 + (uint32_t)col * ROWB + (uint32_t)h * (KS * 16) + (uint32_t)((col & SWM) * 16);
  auto frag = [&](uint32_t fb, int j) {
    return *reinterpret_cast<const __attribute__((address_space(3))) f16x8 *>((uintptr_t)(fb ^ (uint32_t)(j * 16)));
  };
  // max(v1, pk) as med3(v1, pk, +inf): fmaxf() costs a canonicalising v_max per operand on top
  float pinf = INFINITY;
  asm volatile("" : "+s"(pinf));
  auto book = [&](float v, int r, float &v1, float &v2) {
    const float pk = __uint_as_float((__float_as_uint(v) & 0xFFFFFFF0u) | (uint32_t)r);
    v2 = __builtin_amdgcn_fmed3f(v1, v2, pk);
    v1 = __builtin_amdgcn_fmed3f(v1, pk, pinf);
  };
  // One tile: 2 x KS MFMAs (each A fragment feeds both row sets), then the top-2 bookkeeping of its
  // 2 x 16 scores on the VALU.  The two waves a SIMD holds belong to DIFFERENT blocks (4 waves per
  // block, one per SIMD), so they are not in step: one's bookkeeping runs under the other's MFMAs.
  // (Double-buffered accumulators with the bookkeeping interleaved in-wave need ~230 registers: the
  // B operands spill, measured slower.)
  auto tile_pass = [&](uint32_t ldsbase, uint32_t biasaddr, uint32_t t) {
    f32x16 accA, accB;
    {
#pragma unroll
      for (int g = 0; g < 4; g++) {
        const f32x4 b4 = *reinterpret_cast<const __attribute__((address_space(3))) f32x4 *>((uintptr_t)(biasaddr + (8 * g + 4 * h) * 4));
        accA[4 * g + 0] = b4.x; accA[4 * g + 1] = b4.y; accA[4 * g + 2] = b4.z; accA[4 * g + 3] = b4.w;
      }
      accB = accA;
    }
    // (the 1-KB aligned tile base adds into bits the XOR never touches.)  Opaque on purpose: left
    // visible, the KS addresses are hoisted out of the tile loop and the B operands spill instead
    uint32_t fb = fragbase + ldsbase;
    asm volatile("" : "+v"(fb));
    f16x8 f0 = frag(fb, 0), f1 = frag(fb, KS > 1 ? 1 : 0);
#pragma unroll
    for (int j = 0; j < KS; j++) {
      f16x8 f2 = f1;
      if (j + 2 < KS) f2 = frag(fb, j + 2);
      __builtin_amdgcn_sched_barrier(0);
      accA = __builtin_amdgcn_mfma_f32_32x32x16_f16(f0, xa[j], accA, 0, 0, 0);
      accB = __builtin_amdgcn_mfma_f32_32x32x16_f16(f0, xb[j], accB, 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      f0 = f1;
      f1 = f2;
    }
    const float v1a_in = v1a, v1b_in = v1b;
#pragma unroll
    for (int r = 0; r < 16; r++) {
      book(accA[r], r, v1a, v2a);
      book(accB[r], r, v1b, v2b);
    }
    tba = (v1a != v1a_in) ? t : tba;
    tbb = (v1b != v1b_in) ? t : tbb;
  };

  for (uint32_t sp = 0; sp < nsuper; sp++) {
    const int buf = sp & 1;
    if (sp + 1 < nsuper) stage_issue(sp + 1, buf ^ 1);
    const uint32_t base = buf * SUPB, bb = bias0 + buf * 256;
    tile_pass(base, bb, 2 * sp);
    tile_pass(base + 32 * ROWB, bb + 128, 2 * sp + 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }

  // |coarse score - reference score| <= E_c as in lloyd_coarse_kernel, plus the 4 index bits packed
  // into each score (<= 16 ulp of a score of magnitude <= ||x'|| C'max (1 + 2^-10) + B'max).  Rows or
  // panels with a centred norm near the half range could hold inf halves: never decided here.
  const float cmaxc = sqrtf(__uint_as_float(stats[0])) * 1.000001f;
  const float bmaxc = __uint_as_float(stats[1]);
  const float cmaxo = sqrtf(__uint_as_float(stats[2])) * 1.000001f;
  const float u = 5.9604645e-8f;
  uint32_t und_count = 0;
  unsigned long long uma = 0, umb = 0;
  bool unda = false, undb = false;
  auto finish = [&](uint32_t s, bool live, float v1, float v2, uint32_t tb, float xn2, float x0, bool &und,
                    unsigned long long &um) {
    const bool insane = (x0 != x0);  // kmeans.cu:312
    const uint32_t r = __float_as_uint(v1) & 15u;
    uint32_t i1 = tb * 32u + (r & 3u) + 8u * (r >> 2) + 4u * h;
    {
      const float pv1 = __shfl_xor(v1, 32), pv2 = __shfl_xor(v2, 32);
      const uint32_t pi1 = __shfl_xor(i1, 32);
      const bool g = pv1 > v1;
      const float second = fmaxf(g ? v1 : pv1, fmaxf(v2, pv2));
      i1 = g ? pi1 : i1;
      v1 = g ? pv1 : v1;
      v2 = second;
    }
    // ||x|| <= ||x'|| + ||mu|| <= ||x'|| + Cmax (mu is a mean of centroids): saves a second norm
    const float xn = sqrtf(xn2) * 1.0001f, xo = (xn + cmaxo) * 1.0001f;
    const float e_c = 2.0f * eps * (xn * cmaxc + bmaxc) + 9.8e-4f * xn * cmaxc +
                      6e-8f * sqrtf((float)DP) * (xn + cmaxc) + 2.0e-6f * (1.001f * xn * cmaxc + bmaxc);
    const float e_ref = u * (12.0f * xo * cmaxo + 4.0f * cmaxo * cmaxo);
    const float thr = 2.0f * (e_c + e_ref) * 1.001f + tie_slack;
    const bool in_range = (xn < 6.0e4f) && (cmaxc < 6.0e4f) && (v1 > -1.0e38f) && (i1 < K);
    const bool certain = insane || (in_range && ((v1 - v2) > thr));  // NaN gap / thr => not certain
    const bool mine = (h == 0) && live;
    bool changed = false;
    if (mine && certain) changed = commit_row(s, insane ? K : i1, assignments, assignments_prev);
    und = mine && !certain;
    const unsigned long long cm = __ballot(changed);
    um = __ballot(und);
    if (lane == 0 && cm) atomicAdd(&counters[0], (uint32_t)__popcll(cm));
    und_count += (uint32_t)__popcll(um);
  };
  finish(sA, liveA, v1a, v2a, tba, xn2a, x0a, unda, uma);
  finish(sB, liveB, v1b, v2b, tbb, xn2b, x0b, undb, umb);
  if (und_count) {
    uint32_t base = 0;
    if (lane == 0) base = atomicAdd(&counters[4], und_count);
    base = __shfl(base, 0);
    const unsigned long long below = (1ull << lane) - 1ull;
    if (unda) undecided[base + (uint32_t)__popcll(uma & below)] = sA;
    if (undb) undecided[base + (uint32_t)__popcll(uma) + (uint32_t)__popcll(umb & below)] = sB;
  }
}

// c' = c - mu, hi halves only: panelhi[c] = hi(c'_0..DP-1)
__global__ void centroid_panelhi_kernel(const _Float16 *__restrict__ panel16, const float *__restrict__ bias,
                                        uint32_t K_pad, uint32_t DP, _Float16 *__restrict__ panelhi) {
  // grid = K_pad rounded up to 64 rows (the second-generation kernel stages whole 64-row super-tiles);
  // behind the rows: the biases with -inf (padding / non-finite centroids) clamped to a finite floor
  const uint32_t c = blockIdx.x;
  for (uint32_t f = threadIdx.x; f < DP; f += blockDim.x)
    panelhi[(size_t)c * DP + f] = c < K_pad ? panel16[(size_t)c * 2 * DP + f] : (_Float16)0.f;
  if (threadIdx.x == 0) {
    float *biashi = reinterpret_cast<float *>(panelhi + (size_t)gridDim.x * DP);
    biashi[c] = c < K_pad ? fmaxf(bias[c], -3.0e38f) : -3.0e38f;
  }
}

hipError_t launch_centroid_panel16(const float *centroids, uint32_t K, uint32_t D, uint32_t K_pad, uint32_t DP,
                                   const uint32_t *finite, const float *mu, void *panel16, hipStream_t st) {
  hipLaunchKernelGGL(centroid_panel16_kernel, dim3(K_pad), dim3(DP >= 256 ? 256 : 64), 0, st, centroids, K, D, K_pad,
                     DP, finite, mu, reinterpret_cast<_Float16 *>(panel16));
  return hipGetLastError();
}

template <int DP>
static hipError_t launch_f16_dp(const LloydArgs &a, const void *rows, bool half_rows, const void *panel16,
                                const uint32_t *row_list, const uint32_t *n_list, hipStream_t st) {
  const size_t lds_bytes = (2 * 32 * (DP + 4) + 64) * sizeof(float);
  uint32_t grid = (a.N + 127) / 128;
  if (row_list && grid > 4096) grid = 4096;  // the list kernel strides over the device-side count
  const bool fast = a.D == (uint32_t)DP;
#define KMX_F16_LAUNCH(H, F)                                                                                       \
  hipLaunchKernelGGL((lloyd_filter_f16_kernel<DP, H, F>), dim3(grid), dim3(256), lds_bytes, st, rows, a.N, a.D,     \
                     reinterpret_cast<const float *>(panel16), a.bias, a.mu, a.K_pad, a.K, a.stats, a.eps,           \
                     a.tie_slack, a.assignments, a.assignments_prev, a.flagged, a.pairs, a.counters, row_list, n_list)
  if (half_rows) {
    if (fast) KMX_F16_LAUNCH(true, true); else KMX_F16_LAUNCH(true, false);
  } else {
    if (fast) KMX_F16_LAUNCH(false, true); else KMX_F16_LAUNCH(false, false);
  }
#undef KMX_F16_LAUNCH
  return hipGetLastError();
}

template <int DP>
static hipError_t launch_coarse_dp(const LloydArgs &a, const void *rows, bool half_rows, const void *panelhi,
                                   uint32_t *undecided, hipStream_t st) {
  const size_t lds_bytes = (4 * 32 * (DP / 2 + 4) + 128) * sizeof(float);
  const uint32_t rows_per_block = kCoarseThreads / 2;
  const uint32_t grid = (a.N + rows_per_block - 1) / rows_per_block;
  const bool fast = a.D == (uint32_t)DP;
#define KMX_CRS_LAUNCH(H, F)                                                                                       \
  hipLaunchKernelGGL((lloyd_coarse_kernel<DP, H, F>), dim3(grid), dim3(kCoarseThreads), lds_bytes, st, rows, a.N, a.D, \
                     reinterpret_cast<const float *>(panelhi), a.bias, a.mu, a.K_pad, a.K, a.stats, a.eps,           \
                     a.tie_slack, a.assignments, a.assignments_prev, undecided, a.counters)
  if (half_rows) {
    if (fast) KMX_CRS_LAUNCH(true, true); else KMX_CRS_LAUNCH(true, false);
  } else {
    if (fast) KMX_CRS_LAUNCH(false, true); else KMX_CRS_LAUNCH(false, false);
  }
#undef KMX_CRS_LAUNCH
  return hipGetLastError();
}

// one MFMA consumes 8 features per half-wave: the padded width must be at least 16
bool lloyd_filter_f16_supported(uint32_t D, uint32_t DP) { return DP >= 16 && D <= DP; }

hipError_t launch_lloyd_filter_f16(const LloydArgs &a, const void *rows, bool half_rows, const void *panel16,
                                   const uint32_t *row_list, const uint32_t *n_list, hipStream_t st) {
  switch (a.DP) {
    case 16: return launch_f16_dp<16>(a, rows, half_rows, panel16, row_list, n_list, st);
    case 32: return launch_f16_dp<32>(a, rows, half_rows, panel16, row_list, n_list, st);
    case 64: return launch_f16_dp<64>(a, rows, half_rows, panel16, row_list, n_list, st);
    case 128: return launch_f16_dp<128>(a, rows, half_rows, panel16, row_list, n_list, st);
    case 256: return launch_f16_dp<256>(a, rows, half_rows, panel16, row_list, n_list, st);
    default: return hipErrorInvalidValue;
  }
}

template <int DP>
static hipError_t launch_coarse2_dp(const LloydArgs &a, const void *rows, bool half_rows, const void *panelhi,
                                    uint32_t *undecided, hipStream_t st) {
  const size_t lds_bytes = 2 * 64 * (size_t)(DP * 2) + 512;
  const uint32_t grid = (a.N + 255u) / 256u;
  const bool fast = a.D == (uint32_t)DP;
#define KMX_CRS2_LAUNCH(H, F)                                                                                      \
  hipLaunchKernelGGL((lloyd_coarse2_kernel<DP, H, F>), dim3(grid), dim3(kCoarse2Threads), lds_bytes, st, rows, a.N, \
                     a.D, reinterpret_cast<const float *>(panelhi), a.bias, a.mu, a.K_pad, a.K, a.stats, a.eps,      \
                     a.tie_slack, a.assignments, a.assignments_prev, undecided, a.counters)
  if (half_rows) {
    if (fast) KMX_CRS2_LAUNCH(true, true); else KMX_CRS2_LAUNCH(true, false);
  } else {
    if (fast) KMX_CRS2_LAUNCH(false, true); else KMX_CRS2_LAUNCH(false, false);
  }
#undef KMX_CRS2_LAUNCH
  return hipGetLastError();
}

hipError_t launch_lloyd_coarse(const LloydArgs &a, const void *rows, bool half_rows, const void *panel16,
                               void *panelhi, uint32_t *undecided, int generation, hipStream_t st) {
  hipLaunchKernelGGL(centroid_panelhi_kernel, dim3((a.K_pad + 63u) / 64u * 64u), dim3(a.DP >= 256 ? 256 : 64), 0, st,
                     reinterpret_cast<const _Float16 *>(panel16), a.bias, a.K_pad, a.DP, reinterpret_cast<_Float16 *>(panelhi));
  if (generation >= 2) {
    switch (a.DP) {
      case 16: return launch_coarse2_dp<16>(a, rows, half_rows, panelhi, undecided, st);
      case 32: return launch_coarse2_dp<32>(a, rows, half_rows, panelhi, undecided, st);
      case 64: return launch_coarse2_dp<64>(a, rows, half_rows, panelhi, undecided, st);
      case 128: return launch_coarse2_dp<128>(a, rows, half_rows, panelhi, undecided, st);
      case 256: return launch_coarse2_dp<256>(a, rows, half_rows, panelhi, undecided, st);
      default: return hipErrorInvalidValue;
    }
  }
  switch (a.DP) {
    case 16: return launch_coarse_dp<16>(a, rows, half_rows, panelhi, undecided, st);
    case 32: return launch_coarse_dp<32>(a, rows, half_rows, panelhi, undecided, st);
    case 64: return launch_coarse_dp<64>(a, rows, half_rows, panelhi, undecided, st);
    case 128: return launch_coarse_dp<128>(a, rows, half_rows, panelhi, undecided, st);
    case 256: return launch_coarse_dp<256>(a, rows, half_rows, panelhi, undecided, st);
    default: return hipErrorInvalidValue;
  }
}

}  // namespace kmx
