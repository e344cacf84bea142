// lloyd_f16.hip -- the Lloyd assignment filter (reference: src/kmeans.cu:293-364) on the f16 matrix
// cores, for fp32 rows AND for the fp16x2 path's half rows (src/fp_abstraction.h:100-182).
//
// The filter only has to produce scores with a RIGOROUS error bound (lloyd.hip: rows it cannot
// decide go to the exact kernels), so nothing forces it onto the f32 MFMA (64 FLOP/clk/SIMD, 1/16
// of the f16 rate).  Two stages:
//   1. lloyd_coarse2_kernel: operands CENTRED in fp32 (x' = x - mu, c' = c - mu) and rounded to halves;
//      hi(x').hi(c') with ONE v_mfma_f32_32x32x16_f16 per 16 features, products exact in the fp32
//      accumulator, the operand rounding carried explicitly in the bound.  Decides the
//      rows whose best / second-best gap exceeds the bound (the large majority).
//   2. lloyd_refine_kernel: the others -- contenders above the row's cut-off, scored in fp32.
// (Round 1 also had a single-stage three-product pass, x_hi.c_hi + x_hi.c_lo + x_lo.c_hi; it was a third
// copy of the filter kept only as a cross-check and is gone: the f32 matrix-core filter of lloyd.hip and
// the exact kernels are the two cross-checks.)
#include <hip/hip_fp16.h>

#include "exact.hpp"
#include "filter_common.hpp"
#include "kernels.hpp"
#include "lloyd_coarse.hpp"
#include "lloyd_refine.hpp"

namespace kmx {

// hi(c - mu) as halves for the coarse stage: panelhi = K_pad rounded up to whole 64-row super-tiles
// (zero rows for non-finite / padding centroids) and, behind them, the biases with -inf clamped to a
// finite floor; stats[5] = max ||c' - hi(c')||^2, the rounding residual the coarse bound needs.
// One wave per centroid.
__global__ __launch_bounds__(256) void centroid_panelhi_kernel(
    const float *__restrict__ centroids, uint32_t K, uint32_t D, uint32_t K_pad, uint32_t DP,
    const uint32_t *__restrict__ finite, const float *__restrict__ mu, const float *__restrict__ bias,
    _Float16 *__restrict__ panelhi, uint32_t K_pad64, uint32_t *__restrict__ stats) {
  const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint32_t c = blockIdx.x * 4 + wave;
  uint32_t res_bits = 0;
  if (c < K_pad64) {
    const bool real = c < K_pad;
    const bool ok = c < K && finite[c];
    float res2 = 0.f;  // ||c' - hi(c')||^2: what the coarse stage's hi.hi products drop on this side
    for (uint32_t f = lane; f < DP; f += 64) {
      float v = 0.f;
      if (ok && f < D) v = centroids[(size_t)c * D + f] - mu[f];
      const _Float16 hi = (_Float16)v;
      const float r = v - (float)hi;  // exact: hi keeps the leading 11 bits of v
      panelhi[(size_t)c * DP + f] = hi;
      res2 = fmaf(r, r, res2);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) res2 += __shfl_xor(res2, o);
    // an overflowed half leaves inf - inf = NaN: "no bound", the coarse stage decides nothing
    res_bits = ((res2 - res2) == 0.f) ? __float_as_uint(res2 * 1.0001f) : 0x7F800000u;
    if (lane == 0)
      reinterpret_cast<float *>(panelhi + (size_t)K_pad64 * DP)[c] = real ? fmaxf(bias[c], -3.0e38f) : -3.0e38f;
  }
  __shared__ uint32_t part[4];
  if (lane == 0) part[wave] = res_bits;
  __syncthreads();
  if (threadIdx.x == 0) atomicMax(&stats[5], max(max(part[0], part[1]), max(part[2], part[3])));
}

// The whole centroid preparation of a pass in ONE kernel, for the steady state of the two-stage filter
// (mean frozen by the row cache): finite flags, the centred fp32 panel + biases (centroid_panel_kernel),
// the hi halves + clamped biases + residual maximum (centroid_panelhi_kernel) and the uncentred norm
// maximum (centroid_rows_kernel) -- what stage 1 waits for.  The reference's exact sum_squares chain
// (serial, 256 steps) and the transposed panel are only read by the pair / exact kernels: they run
// beside stage 1 on the side stream.  One wave per padded row.  Also zeroes the per-pass list counters
// and the OTHER half of the double-buffered stats (the next pass's), saving the memset launches.
// APPLY (L2 only): the centroid update of update.hip's apply_delta_kernel -- the same fp64 formula element by
// element, the same device-side stop rule (StopCtl) -- runs first, on the row the wave is about to prepare: the
// update and the next pass's preparation are ONE launch (Engine::apply_prepare).
template <int METRIC, bool APPLY>
__global__ __launch_bounds__(256) void centroid_prep_frozen_kernel(
    float *__restrict__ centroids, uint32_t K, uint32_t D, uint32_t K_pad, uint32_t DP, uint32_t K_pad64,
    const float *__restrict__ mu, uint32_t *__restrict__ finite, float *__restrict__ bias, float *__restrict__ bias2,
    float *__restrict__ cfil, _Float16 *__restrict__ panelhi, uint32_t *__restrict__ stats,
    uint32_t *__restrict__ stats_next, uint32_t *__restrict__ zero_a, uint32_t *__restrict__ zero_b,
    uint32_t *__restrict__ zero_c, const double *__restrict__ delta, const double *__restrict__ dcount_d,
    uint32_t *__restrict__ ccounts, StopCtl ctl, float *__restrict__ drift, uint32_t *__restrict__ zero_d) {
  const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (blockIdx.x == 0 && threadIdx.x < 8) {
    stats_next[threadIdx.x] = 0u;
    if (threadIdx.x == 0) { *zero_a = 0u; *zero_b = 0u; *zero_c = 0u; zero_c[kDuoCount - 4] = 0u; if (zero_d) *zero_d = 0u; }   // (zero_c = counters + 4)
  }
  bool update = APPLY;
  if (APPLY && ctl.counters) {   // as apply_delta_kernel
    bool stop = ctl.counters[kStopFlag] != 0u;
    if (ctl.threshold >= 0.f) stop = stop || (float)(uint32_t)dcount_d[K] <= ctl.threshold;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
      if (stop) ctl.counters[kStopFlag] = 1u;
      else if (ctl.threshold >= 0.f) ctl.counters[0] = 0u;
      if (ctl.host_tail) {
        volatile uint32_t *ht = ctl.host_tail;
        for (uint32_t i = 0; i < 4; i++) ht[i] = (uint32_t)dcount_d[K + i];
        ht[4] = stop ? 1u : 0u;
        __threadfence_system();
        ht[5] = ctl.seq;
      }
    }
    update = !stop;
  }
  uint32_t s0 = 0, s1 = 0, s2 = 0, s3 = 0, s4 = 0, s5 = 0, s6 = 0, s7 = 0;
  for (uint32_t i = 0; i < 4; i++) {
    const uint32_t c = blockIdx.x * 16 + wave * 4 + i;
    if (c >= K_pad64) break;
    const bool real = c < K_pad;
    float plain = 0.f;
    if (APPLY && update && c < K) {
      // c = (c * count + delta) / new count in fp64, rounded once (apply_delta_kernel<0>); the row is read back
      // below by the lanes that wrote it
      const uint32_t cnt_old = ccounts[c];
      const uint32_t cnt_new = cnt_old + (uint32_t)(int32_t)dcount_d[c];
      const double w = (double)cnt_old, cn = (double)cnt_new;
      for (uint32_t f = lane; f < D; f += 64) {
        const float v = cnt_new == 0 ? __builtin_nanf("")
                                     : (float)(((double)centroids[(size_t)c * D + f] * w + delta[(size_t)c * D + f]) / cn);
        centroids[(size_t)c * D + f] = v;
        plain = fmaf(v, v, plain);
      }
      if (lane == 0) ccounts[c] = cnt_new;
    } else if (c < K)
      for (uint32_t f = lane; f < D; f += 64) {
        const float v = centroids[(size_t)c * D + f];
        plain = fmaf(v, v, plain);
      }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) plain += __shfl_xor(plain, off);
    const bool ok = c < K && (plain - plain) == 0.f;   // false for a row holding a NaN or an inf
    if (c < K && lane == 0) finite[c] = ok ? 1u : 0u;
    float n2 = 0.f, mc = 0.f, m2 = 0.f, res2 = 0.f, dr2 = 0.f, o2 = 0.f;
    for (uint32_t f = lane; f < DP; f += 64) {
      float v = 0.f, m = 0.f;
      if (ok && f < D) {
        m = mu[f];
        v = centroids[(size_t)c * D + f] - m;
      }
      const _Float16 hi = (_Float16)v;
      const float r = v - (float)hi;
      if (drift && real) {   // the panel still holds the previous pass's centred centroid (the mean is frozen)
        const float old = cfil[(size_t)c * DP + f], d = v - old;
        dr2 = fmaf(d, d, dr2);
        o2 = fmaf(old, old, o2);
      }
      if (real) cfil[(size_t)c * DP + f] = v;
      panelhi[(size_t)c * DP + f] = hi;
      n2 = fmaf(v, v, n2);
      mc = fmaf(m, v, mc);
      m2 = fmaf(m, m, m2);
      res2 = fmaf(r, r, res2);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      n2 += __shfl_xor(n2, off);
      mc += __shfl_xor(mc, off);
      m2 += __shfl_xor(m2, off);
      res2 += __shfl_xor(res2, off);
      dr2 += __shfl_xor(dr2, off);
      o2 += __shfl_xor(o2, off);
    }
    if (drift && c < K) {
      // ||c_new - c_old|| <= ||fl(c_new - mu) - fl(c_old - mu)|| + u (||c_new'|| + ||c_old'||): an upper bound, fp32
      // sums of DP terms rounded up; anything not finite = "no bound" (+inf: no row is spared)
      float dr = sqrtf(dr2) * 1.0002f + 2.4e-7f * (sqrtf(n2) + sqrtf(o2)) + 1e-37f;
      if (!((dr - dr) == 0.f)) dr = INFINITY;
      if (lane == 0) drift[c] = dr;
      s6 = max(s6, __float_as_uint(dr));
    }
    float b = -INFINITY, b2 = -INFINITY;
    if (ok) {
      float bmag;
      if (METRIC == 0) {
        b = -0.5f * n2;
        bmag = 0.5f * n2;
      } else {
        b = mc;
        bmag = sqrtf(m2) * sqrtf(n2) * 1.0001f;
      }
      const float mag2 = (METRIC == 0) ? sqrtf(m2) * sqrtf(n2) * 1.0001f + 0.5f * n2 : 0.f;
      b2 = (METRIC == 0) ? -mc - 0.5f * n2 : 0.f;
      s0 = max(s0, __float_as_uint(n2 * 1.0001f));
      s1 = max(s1, __float_as_uint(bmag * 1.0001f));
      s2 = max(s2, __float_as_uint(plain * 1.0001f));
      s3 = max(s3, __float_as_uint(m2 * 1.0001f));
      s4 = max(s4, __float_as_uint(mag2 * 1.0001f));
    }
    s5 = max(s5, ((res2 - res2) == 0.f) ? __float_as_uint(res2 * 1.0001f) : 0x7F800000u);
    if (lane == 0) {
      if (METRIC != 0 && drift && c < K) {
        // angular: the score x'.c' + mu.c' moves by at most ||x'|| drift(c) PLUS this, known per centroid (the bias
        // is mu.c', the mean being frozen): carry_skip_kernel charges a row max_c(db) - db(its centroid)
        const float db = b - bias[c];
        drift[K + c] = db;
        if (ok && (db - db) == 0.f && db > 0.f) s7 = max(s7, __float_as_uint(db * 1.000001f));
        else if (ok && !((db - db) == 0.f)) s7 = 0x7F800000u;
      }
      if (real) { bias[c] = b; bias2[c] = b2; }
      reinterpret_cast<float *>(panelhi + (size_t)K_pad64 * DP)[c] = real ? fmaxf(b, -3.0e38f) : -3.0e38f;
    }
  }
  __shared__ uint32_t red[4][8];
  if (lane == 0) { red[wave][0] = s0; red[wave][1] = s1; red[wave][2] = s2; red[wave][3] = s3; red[wave][4] = s4; red[wave][5] = s5; red[wave][6] = s6; red[wave][7] = s7; }
  __syncthreads();
  if (threadIdx.x < 8) {
    const uint32_t m = max(max(red[0][threadIdx.x], red[1][threadIdx.x]), max(red[2][threadIdx.x], red[3][threadIdx.x]));
    if (m) atomicMax(&stats[threadIdx.x], m);
  }
}

hipError_t launch_centroid_prep_frozen(int metric, const float *centroids, uint32_t K, uint32_t D, uint32_t K_pad,
                                       uint32_t DP, const float *mu, uint32_t *finite, float *bias, float *bias2,
                                       float *cfil, void *panelhi, uint32_t *stats, uint32_t *stats_next,
                                       uint32_t *zero_a, uint32_t *zero_b, uint32_t *zero_c, float *drift,
                                       uint32_t *zero_d, hipStream_t st) {
  const uint32_t K_pad64 = (K_pad + 63u) / 64u * 64u;
  float *cen = const_cast<float *>(centroids);   // (only the APPLY instantiation writes)
  if (metric == 0)
    hipLaunchKernelGGL((centroid_prep_frozen_kernel<0, false>), dim3(K_pad64 / 16), dim3(256), 0, st, cen, K, D, K_pad,
                       DP, K_pad64, mu, finite, bias, bias2, cfil, reinterpret_cast<_Float16 *>(panelhi), stats,
                       stats_next, zero_a, zero_b, zero_c, (const double *)nullptr, (const double *)nullptr,
                       (uint32_t *)nullptr, StopCtl(), drift, zero_d);
  else
    hipLaunchKernelGGL((centroid_prep_frozen_kernel<1, false>), dim3(K_pad64 / 16), dim3(256), 0, st, cen, K, D, K_pad,
                       DP, K_pad64, mu, finite, bias, bias2, cfil, reinterpret_cast<_Float16 *>(panelhi), stats,
                       stats_next, zero_a, zero_b, zero_c, (const double *)nullptr, (const double *)nullptr,
                       (uint32_t *)nullptr, StopCtl(), drift, zero_d);
  return hipGetLastError();
}

// L2: the centroid update (delta / dcount_d: the fused reduce buffer, StopCtl as launch_apply_delta) and the next
// pass's preparation in one launch
hipError_t launch_apply_prep_frozen(const double *delta, const double *dcount_d, float *centroids, uint32_t *ccounts,
                                    const StopCtl &stop, uint32_t K, uint32_t D, uint32_t K_pad, uint32_t DP,
                                    const float *mu, uint32_t *finite, float *bias, float *bias2, float *cfil,
                                    void *panelhi, uint32_t *stats, uint32_t *stats_next, uint32_t *zero_a,
                                    uint32_t *zero_b, uint32_t *zero_c, float *drift, uint32_t *zero_d, hipStream_t st) {
  const uint32_t K_pad64 = (K_pad + 63u) / 64u * 64u;
  hipLaunchKernelGGL((centroid_prep_frozen_kernel<0, true>), dim3(K_pad64 / 16), dim3(256), 0, st, centroids, K, D, K_pad,
                     DP, K_pad64, mu, finite, bias, bias2, cfil, reinterpret_cast<_Float16 *>(panelhi), stats, stats_next,
                     zero_a, zero_b, zero_c, delta, dcount_d, ccounts, stop, drift, zero_d);
  return hipGetLastError();
}


hipError_t launch_centroid_panelhi(const float *centroids, uint32_t K, uint32_t D, uint32_t K_pad, uint32_t DP,
                                   const uint32_t *finite, const float *mu, const float *bias, void *panelhi,
                                   uint32_t *stats, hipStream_t st) {
  const uint32_t K_pad64 = (K_pad + 63u) / 64u * 64u;
  hipLaunchKernelGGL(centroid_panelhi_kernel, dim3(K_pad64 / 4), dim3(256), 0, st, centroids, K, D, K_pad, DP, finite,
                     mu, bias, reinterpret_cast<_Float16 *>(panelhi), K_pad64, stats);
  return hipGetLastError();
}

// one MFMA consumes 8 features per half-wave: the padded width must be at least 16
bool lloyd_filter_f16_supported(uint32_t D, uint32_t DP) { return DP >= 16 && D <= DP; }

template <int DP>
static hipError_t launch_coarse2_dp(const LloydArgs &a, const void *rows, bool half_rows, const void *xcache,
                                    const float *xmeta, const void *panelhi, uint32_t *undecided, float *und_thr,
                                    uint32_t *duo, hipStream_t st) {
  constexpr int NSET = DP <= 256 ? 2 : 1;
  const size_t lds_bytes = 2 * 64 * (size_t)(DP * 2) + 512 + 64 + (size_t)DP * 4;
  const uint32_t rows_per_block = 128u * NSET;
  const uint32_t grid = (a.N + rows_per_block - 1) / rows_per_block;
  const bool fast = a.D == (uint32_t)DP;
#define KMX_CRS2_LAUNCH(H, F, C, SRC)                                                                              \
  hipLaunchKernelGGL((lloyd_coarse2_kernel<DP, H, F, C, NSET>), dim3(grid), dim3(256), lds_bytes, st, SRC, xmeta,   \
                     a.N, a.D, reinterpret_cast<const float *>(panelhi), a.bias, a.mu, a.K_pad, a.K, a.stats,       \
                     a.eps, a.tie_slack, a.assignments, a.assignments_prev, undecided, und_thr, a.counters, CarryArgs(), duo)
  if (xcache) {
    KMX_CRS2_LAUNCH(false, true, true, xcache);
  } else if (half_rows) {
    if (fast) KMX_CRS2_LAUNCH(true, true, false, rows); else KMX_CRS2_LAUNCH(true, false, false, rows);
  } else {
    if (fast) KMX_CRS2_LAUNCH(false, true, false, rows); else KMX_CRS2_LAUNCH(false, false, false, rows);
  }
#undef KMX_CRS2_LAUNCH
  return hipGetLastError();
}

hipError_t launch_lloyd_coarse(const LloydArgs &a, const void *rows, bool half_rows, const void *xcache,
                               const float *xmeta, const void *panelhi, uint32_t *undecided, float *und_thr,
                               uint32_t *duo, hipStream_t st) {
  switch (a.DP) {
    case 16: return launch_coarse2_dp<16>(a, rows, half_rows, xcache, xmeta, panelhi, undecided, und_thr, duo, st);
    case 32: return launch_coarse2_dp<32>(a, rows, half_rows, xcache, xmeta, panelhi, undecided, und_thr, duo, st);
    case 64: return launch_coarse2_dp<64>(a, rows, half_rows, xcache, xmeta, panelhi, undecided, und_thr, duo, st);
    case 128: return launch_coarse2_dp<128>(a, rows, half_rows, xcache, xmeta, panelhi, undecided, und_thr, duo, st);
    case 256: return launch_coarse2_dp<256>(a, rows, half_rows, xcache, xmeta, panelhi, undecided, und_thr, duo, st);
    case 512: return launch_coarse2_dp<512>(a, rows, half_rows, xcache, xmeta, panelhi, undecided, und_thr, duo, st);
    default: return hipErrorInvalidValue;
  }
}

hipError_t launch_lloyd_refine(const LloydArgs &a, const void *rows, bool half_rows, const void *panelhi,
                               const uint32_t *row_list, const float *thr_list, const uint32_t *n_list,
                               uint32_t rows_hint, hipStream_t st) {
  return launch_lloyd_refine_t<false>(a, rows, half_rows, panelhi, row_list, thr_list, n_list, rows_hint, CarryArgs(), st);
}

// ---------------------------------------------------------------------------------------
// Row cache: the coarse stage's B operands, x' = x - mu rounded to halves, in the operand order of
// lloyd_coarse2_kernel -- per 32-row block, DP/16 pieces of 64 lanes x 16 bytes (lane = (row, half
// of the features)) -- plus (||x'||^2, x_0) per row.  Built once per engine while mu stays frozen
// (engine.cpp): afterwards an iteration streams 2 DP bytes per row in whole 1-KB bursts instead of
// 4 DP bytes as 16-byte pieces of 64 different lines per load, and converts nothing.
// ---------------------------------------------------------------------------------------
template <int DP, bool HALF_ROWS, bool FAST>
__global__ __launch_bounds__(256) void row_cache_kernel(const void *__restrict__ rows, uint32_t N, uint32_t D,
                                                        const float *__restrict__ mu, f16x8 *__restrict__ xcache,
                                                        float2 *__restrict__ xmeta, uint32_t nblocks32) {
  // (x.mu per row behind the records and ||mu||: score + x.mu = the product, which the angular metric clamps at 1 / -1,
  //  filter_common.hpp)
  float *__restrict__ xdot = reinterpret_cast<float *>(xmeta) + 2 * ((size_t)nblocks32 * 32) + 2;
  constexpr int NKH = DP / 2, KS = NKH / 8;
  const int lane = threadIdx.x & 63, col = lane & 31, h = lane >> 5;
  // (a wave per 32-row block, strided from a bounded grid: kernels.hpp, wave_row_grid)
  for (uint32_t b = blockIdx.x * 4u + (threadIdx.x >> 6); b < nblocks32; b += gridDim.x * 4u) {
  if (b == 0) {  // ||mu|| behind the per-row records: ||x|| <= ||x'|| + ||mu|| in the coarse kernel's bound
    float m2 = 0.f;
    for (uint32_t f = lane; f < (uint32_t)DP; f += 64) m2 = fmaf(mu[f], mu[f], m2);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m2 += __shfl_xor(m2, o);
    if (lane == 0) xmeta[(size_t)nblocks32 * 32] = make_float2(sqrtf(m2) * 1.00001f, 0.f);
  }
  const uint32_t s = b * 32u + col;
  const bool live = s < N;
  const size_t row = (size_t)(live ? s : 0);
  float xn2 = 0.f, dx2 = 0.f, x0 = 0.f, xdm = 0.f;
#pragma unroll
  for (int j = 0; j < KS; j++) {
    float xv[8];
    if (FAST && HALF_ROWS) {
      const f16x8 raw = reinterpret_cast<const f16x8 *>(reinterpret_cast<const _Float16 *>(rows) + row * DP + h * NKH)[j];
#pragma unroll
      for (int q = 0; q < 8; q++) xv[q] = (float)raw[q];
    } else if (FAST) {
      const f32x4 *src = reinterpret_cast<const f32x4 *>(reinterpret_cast<const float *>(rows) + row * DP + h * NKH);
      const f32x4 a = src[2 * j], c = src[2 * j + 1];
      xv[0] = a.x; xv[1] = a.y; xv[2] = a.z; xv[3] = a.w;
      xv[4] = c.x; xv[5] = c.y; xv[6] = c.z; xv[7] = c.w;
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
      const float mq = mu[h * NKH + 8 * j + q];                        // mu: DP floats, zero beyond D
      const float xc = live ? xv[q] - mq : 0.f;
      xdm = fmaf(live ? xv[q] : 0.f, mq, xdm);
      const _Float16 a = (_Float16)xc;
      const float r = xc - (float)a;   // exact; what the hi.hi products drop on this side
      hi[q] = a;
      xn2 = fmaf(xc, xc, xn2);
      dx2 = fmaf(r, r, dx2);
    }
    if (j == 0) x0 = live ? xv[0] : 0.f;
    xcache[((size_t)b * KS + j) * 64 + lane] = hi;
  }
  xn2 += __shfl_xor(xn2, 32);
  dx2 += __shfl_xor(dx2, 32);
  xdm += __shfl_xor(xdm, 32);
  if (h == 0) xdot[s] = xdm;
  // record = (||x'||^2, ||x' - hi(x')||^2); a NaN first feature (kmeans.cu:312) is flagged by -1
  if (h == 0) xmeta[s] = make_float2(xn2, (x0 != x0) ? -1.f : dx2 * 1.0001f);   // xmeta covers the padded row count
  }
}

template <int DP>
static hipError_t launch_row_cache_dp(const void *rows, bool half_rows, uint32_t N, uint32_t D, const float *mu,
                                      void *xcache, float *xmeta, hipStream_t st) {
  const uint32_t nblocks32 = (N + 255u) / 256u * 8u;
  const bool fast = D == (uint32_t)DP;
#define KMX_RC_LAUNCH(H, F)                                                                                       \
  hipLaunchKernelGGL((row_cache_kernel<DP, H, F>), dim3(wave_row_grid(nblocks32)), dim3(256), 0, st, rows, N, D, mu, \
                     reinterpret_cast<f16x8 *>(xcache), reinterpret_cast<float2 *>(xmeta), nblocks32)
  if (half_rows) {
    if (fast) KMX_RC_LAUNCH(true, true); else KMX_RC_LAUNCH(true, false);
  } else {
    if (fast) KMX_RC_LAUNCH(false, true); else KMX_RC_LAUNCH(false, false);
  }
#undef KMX_RC_LAUNCH
  return hipGetLastError();
}

hipError_t launch_row_cache(const void *rows, bool half_rows, uint32_t N, uint32_t D, uint32_t DP, const float *mu,
                            void *xcache, float *xmeta, hipStream_t st) {
  switch (DP) {
    case 16: return launch_row_cache_dp<16>(rows, half_rows, N, D, mu, xcache, xmeta, st);
    case 32: return launch_row_cache_dp<32>(rows, half_rows, N, D, mu, xcache, xmeta, st);
    case 64: return launch_row_cache_dp<64>(rows, half_rows, N, D, mu, xcache, xmeta, st);
    case 128: return launch_row_cache_dp<128>(rows, half_rows, N, D, mu, xcache, xmeta, st);
    case 256: return launch_row_cache_dp<256>(rows, half_rows, N, D, mu, xcache, xmeta, st);
    case 512: return launch_row_cache_dp<512>(rows, half_rows, N, D, mu, xcache, xmeta, st);
    default: return hipErrorInvalidValue;
  }
}

hipError_t preload_lloyd_f16_code() {   // (kernels.hpp: preload_code_objects)
  hipFuncAttributes at;
  return hipFuncGetAttributes(&at, reinterpret_cast<const void *>(&centroid_panelhi_kernel));
}

}  // namespace kmx
