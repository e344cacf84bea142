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
    if (threadIdx.x == 0) { *zero_a = 0u; *zero_b = 0u; *zero_c = 0u; if (zero_d) *zero_d = 0u; }
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
                                    hipStream_t st) {
  constexpr int NSET = DP <= 256 ? 2 : 1;
  const size_t lds_bytes = 2 * 64 * (size_t)(DP * 2) + 512 + 64 + (size_t)DP * 4;
  const uint32_t rows_per_block = 128u * NSET;
  const uint32_t grid = (a.N + rows_per_block - 1) / rows_per_block;
  const bool fast = a.D == (uint32_t)DP;
#define KMX_CRS2_LAUNCH(H, F, C, SRC)                                                                              \
  hipLaunchKernelGGL((lloyd_coarse2_kernel<DP, H, F, C, NSET>), dim3(grid), dim3(256), lds_bytes, st, SRC, xmeta,   \
                     a.N, a.D, reinterpret_cast<const float *>(panelhi), a.bias, a.mu, a.K_pad, a.K, a.stats,       \
                     a.eps, a.tie_slack, a.assignments, a.assignments_prev, undecided, und_thr, a.counters, CarryArgs())
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
                               hipStream_t st) {
  switch (a.DP) {
    case 16: return launch_coarse2_dp<16>(a, rows, half_rows, xcache, xmeta, panelhi, undecided, und_thr, st);
    case 32: return launch_coarse2_dp<32>(a, rows, half_rows, xcache, xmeta, panelhi, undecided, und_thr, st);
    case 64: return launch_coarse2_dp<64>(a, rows, half_rows, xcache, xmeta, panelhi, undecided, und_thr, st);
    case 128: return launch_coarse2_dp<128>(a, rows, half_rows, xcache, xmeta, panelhi, undecided, und_thr, st);
    case 256: return launch_coarse2_dp<256>(a, rows, half_rows, xcache, xmeta, panelhi, undecided, und_thr, st);
    case 512: return launch_coarse2_dp<512>(a, rows, half_rows, xcache, xmeta, panelhi, undecided, und_thr, st);
    default: return hipErrorInvalidValue;
  }
}

// ---------------------------------------------------------------------------------------
// Stage 2 of the default filter: the rows stage 1 could not decide.  For such a row every centroid
// whose coarse score lies below (best coarse score - thr) is already ruled out (und_thr, written by
// stage 1), and what is left are a handful of CONTENDERS.  So instead of the three-product pass over
// all K centroids (lloyd_filter_f16_kernel, 3 MFMAs per 16 features) this kernel
//   1. recomputes the coarse scores exactly as stage 1 did (same operands, same MFMA order: 1 MFMA
//      per 16 features, 64 rows per wave, LDS-DMA tiles) and, instead of any top-k bookkeeping,
//      compares each tile's maximum with the row's cut-off (8 v_max3 + 1 compare per 16 scores);
//      the rare hits append the centroid to the row's contender list in LDS;
//   2. scores the contenders in fp32 on the VALU: x'.c' + bias as an FMA dot product of the centred
//      fp32 row with the centred fp32 centroid (the f32 matrix-core filter's operands, so its bound
//      E = 2 eps (||x'|| C'max + B'max) + E_ref applies: gamma_{D+1} of a recursive sum);
//   3. decides like the other filters: best - second > 2E commits, best - third > 2E hands the two
//      contenders to the pair kernel, anything else (also: more than kCap contenders, no usable
//      cut-off, operands near the half range) goes to the full exact scan.
// ---------------------------------------------------------------------------------------
constexpr int kRefineCap = 8;   // contenders kept per row
template <int DP, bool HALF_ROWS, bool FAST, int NSET>
__global__ __launch_bounds__(256, 2) void lloyd_refine_kernel(
    const void *__restrict__ rows, const float *__restrict__ samples, uint32_t N, uint32_t D,
    const float *__restrict__ panelhi, const float *__restrict__ cfil, const float *__restrict__ bias,
    const float *__restrict__ mu, uint32_t K_pad, uint32_t K, const uint32_t *__restrict__ stats, float eps,
    float tie_slack, uint32_t *__restrict__ assignments, uint32_t *__restrict__ assignments_prev,
    const uint32_t *__restrict__ row_list, const float *__restrict__ thr_list, const uint32_t *__restrict__ n_list,
    uint32_t *__restrict__ flagged, uint32_t *__restrict__ pairs, uint32_t *__restrict__ counters) {
  constexpr int NKH = DP / 2;
  constexpr int KS = NKH / 8;
  constexpr int ROWB = DP * 2;
  constexpr int SUPB = 64 * ROWB;
  constexpr int NP = SUPB / 1024;
  constexpr int SWM = (KS < 16 ? KS : 16) - 1;
  const uint32_t total = *n_list;
  constexpr bool TWO = NSET == 2;
  typedef __attribute__((address_space(3))) unsigned char lds_byte;
  extern __shared__ __attribute__((aligned(1024))) unsigned char lds2[];
  const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_byte *)lds2;
  if (lds0 & 1023u) __builtin_trap();
  const uint32_t bias0 = lds0 + 2 * SUPB;            // 2 x 64 floats
  const uint32_t mu_lds = bias0 + 512 + 64;          // DP floats
  const uint32_t cnt_lds = mu_lds + DP * 4;          // 256 contender counts
  const uint32_t list_lds = cnt_lds + 1024;          // 256 x kRefineCap centroid indices
  auto lds_u32 = [](uint32_t addr) { return reinterpret_cast<__attribute__((address_space(3))) uint32_t *>((uintptr_t)addr); };
  // the grid follows the previous pass's list length (engine.cpp); a longer list is strided over
  for (uint32_t blk = blockIdx.x; (size_t)blk * (128u * NSET) < total; blk += gridDim.x) {

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), col = lane & 31, h = lane >> 5;
  const uint32_t posA = blk * (128u * NSET) + wave * (32u * NSET) + col, posB = posA + 32u;
  const bool liveA = posA < total, liveB = TWO && posB < total;
  const uint32_t sA = liveA ? row_list[posA] : 0u, sB = liveB ? row_list[posB] : 0u;
  // a lane without a row gets a cut-off nothing reaches; NaN (no usable cut-off) behaves the same and is
  // caught below by "no contender"
  const float cutA = liveA ? thr_list[posA] : INFINITY, cutB = liveB ? thr_list[posB] : INFINITY;
  const uint32_t rlA = wave * (32u * NSET) + col, rlB = rlA + 32u;   // row slots of the block's contender lists
  *lds_u32(cnt_lds + tid * 4) = 0u;

  // ---- operands: as lloyd_coarse2_kernel without the row cache (gathered rows) ----
  f16x8 xa[KS], xb[TWO ? KS : 1];
  auto load_chunk = [&](uint32_t s, int j, float (&xv)[8]) {
    const size_t row = (size_t)s;
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
  const uint32_t nsuper = (K_pad + 63) / 64;
  const float *biashi = reinterpret_cast<const float *>(reinterpret_cast<const unsigned char *>(panelhi) + (size_t)nsuper * SUPB);
  auto stage_piece = [&](uint32_t sp, int buf, int p) {
    uint32_t P0 = (uint32_t)lane * 16u;
    asm volatile("" : "+v"(P0));
    const unsigned char *src = reinterpret_cast<const unsigned char *>(panelhi) + (size_t)sp * SUPB;
    const uint32_t P = (uint32_t)p * 1024u + P0;
    const uint32_t from = P ^ (((P / ROWB) & SWM) << 4);
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + from),
                                     (__attribute__((address_space(3))) void *)(uintptr_t)(lds0 + buf * SUPB + p * 1024), 16, 0, 0);
  };
  auto stage_bias = [&](uint32_t sp, int buf) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(biashi + sp * 64u + lane),
                                     (__attribute__((address_space(3))) void *)(uintptr_t)(bias0 + buf * 256), 4, 0, 0);
  };
  {
    constexpr int MUP = (DP * 4 + 1023) / 1024;
    if (wave < MUP && lane * 16 < DP * 4 - wave * 1024)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(reinterpret_cast<const unsigned char *>(mu) + wave * 1024 + lane * 16),
                                       (__attribute__((address_space(3))) void *)(uintptr_t)(mu_lds + wave * 1024), 16, 0, 0);
  }
  for (int p = wave; p < NP; p += 4) stage_piece(0, 0, p);
  if (wave == 0) stage_bias(0, 0);
  {
    constexpr int BJ = KS > 8 ? 8 : KS;
#pragma unroll
    for (int j0 = 0; j0 < KS; j0 += BJ) {
      float va[BJ][8], vb[TWO ? BJ : 1][8];
#pragma unroll
      for (int jj = 0; jj < BJ; jj++) load_chunk(sA, j0 + jj, va[jj]);
#pragma unroll
      for (int jj = 0; jj < (TWO ? BJ : 0); jj++) load_chunk(sB, j0 + jj, vb[jj]);
      __builtin_amdgcn_sched_barrier(0);
      if (j0 == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
      }
#pragma unroll
      for (int jj = 0; jj < BJ; jj++) {
        const int j = j0 + jj;
        const f32x4 m0 = *reinterpret_cast<const __attribute__((address_space(3))) f32x4 *>((uintptr_t)(mu_lds + (h * NKH + 8 * j) * 4));
        const f32x4 m1 = *reinterpret_cast<const __attribute__((address_space(3))) f32x4 *>((uintptr_t)(mu_lds + (h * NKH + 8 * j + 4) * 4));
        const float mm[8] = {m0.x, m0.y, m0.z, m0.w, m1.x, m1.y, m1.z, m1.w};
        f16x8 ha;
#pragma unroll
        for (int q = 0; q < 8; q++) ha[q] = (_Float16)(va[jj][q] - mm[q]);
        xa[j] = ha;
        asm volatile("" : "+v"(xa[j]));
        if constexpr (TWO) {
          f16x8 hb;
#pragma unroll
          for (int q = 0; q < 8; q++) hb[q] = (_Float16)(vb[jj][q] - mm[q]);
          xb[j] = hb;
          asm volatile("" : "+v"(xb[j]));
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }

  const uint32_t fragbase = lds0 + (uint32_t)col * ROWB + (uint32_t)h * (KS * 16) + (uint32_t)((col & SWM) * 16);
  float pinf = INFINITY;
  asm volatile("" : "+s"(pinf));
  auto append = [&](uint32_t rl, uint32_t c) {
    const uint32_t at = __hip_atomic_fetch_add(lds_u32(cnt_lds + rl * 4), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (at < (uint32_t)kRefineCap) *lds_u32(list_lds + (rl * kRefineCap + at) * 4) = c;
  };
  auto tile_pass = [&](uint32_t ldsbase, uint32_t biasaddr, uint32_t t, bool stage, uint32_t sp_next, int buf_next) {
    f32x16 accA, accB;
#pragma unroll
    for (int g = 0; g < 4; g++) {
      const f32x4 b4 = *reinterpret_cast<const __attribute__((address_space(3))) f32x4 *>((uintptr_t)(biasaddr + (8 * g + 4 * h) * 4));
      accA[4 * g + 0] = b4.x; accA[4 * g + 1] = b4.y; accA[4 * g + 2] = b4.z; accA[4 * g + 3] = b4.w;
    }
    accB = accA;
    uint32_t fb = fragbase + ldsbase;
    asm volatile("" : "+v"(fb));
    constexpr int PD = KS <= 3 ? KS - 1 : 3;
    f16x8 fr[PD + 1];
#pragma unroll
    for (int j = 0; j < PD; j++) fr[j] = lds_frag_issue(fb ^ (uint32_t)(j * 16));
#pragma unroll
    for (int j = 0; j < KS; j++) {
      if (j + PD < KS) fr[(j + PD) % (PD + 1)] = lds_frag_issue(fb ^ (uint32_t)((j + PD) * 16));
      const int behind = (KS - 1 - j) < PD ? (KS - 1 - j) : PD;
      f16x8 &f = fr[j % (PD + 1)];
      if (behind == 3) lds_frag_wait<3>(f);
      else if (behind == 2) lds_frag_wait<2>(f);
      else if (behind == 1) lds_frag_wait<1>(f);
      else lds_frag_wait<0>(f);
      accA = __builtin_amdgcn_mfma_f32_32x32x16_f16(f, xa[j], accA, 0, 0, 0);
      if constexpr (TWO) accB = __builtin_amdgcn_mfma_f32_32x32x16_f16(f, xb[j], accB, 0, 0, 0);
      constexpr int SPREAD = KS >= 8 ? KS / 8 : 1;
      if (stage && (j % SPREAD) == SPREAD / 2 && j / SPREAD < 8) {
        const int slot = j / SPREAD;
        for (int p = slot * 4 + wave; p < NP; p += 32) stage_piece(sp_next, buf_next, p);
        if (slot == 0 && wave == 0) stage_bias(sp_next, buf_next);
      }
    }
    // tile maximum against the row's cut-off (max as med3(a, b, +inf)); the padding tiles' floor (-3e38)
    // is below any cut-off; operands are finite here or the row is never settled from its list
    // (real instructions, not inline asm: the compiler must see the MFMA -> VALU read hazard)
    auto max16 = [&](const f32x16 &a) {
      float m0 = __builtin_amdgcn_fmed3f(a[0], a[1], pinf), m1 = __builtin_amdgcn_fmed3f(a[2], a[3], pinf);
      float m2 = __builtin_amdgcn_fmed3f(a[4], a[5], pinf), m3 = __builtin_amdgcn_fmed3f(a[6], a[7], pinf);
      m0 = __builtin_amdgcn_fmed3f(m0, a[8], pinf);  m1 = __builtin_amdgcn_fmed3f(m1, a[9], pinf);
      m2 = __builtin_amdgcn_fmed3f(m2, a[10], pinf); m3 = __builtin_amdgcn_fmed3f(m3, a[11], pinf);
      m0 = __builtin_amdgcn_fmed3f(m0, a[12], pinf); m1 = __builtin_amdgcn_fmed3f(m1, a[13], pinf);
      m2 = __builtin_amdgcn_fmed3f(m2, a[14], pinf); m3 = __builtin_amdgcn_fmed3f(m3, a[15], pinf);
      m0 = __builtin_amdgcn_fmed3f(m0, m1, pinf);
      m2 = __builtin_amdgcn_fmed3f(m2, m3, pinf);
      return __builtin_amdgcn_fmed3f(m0, m2, pinf);
    };
    const bool hitA = max16(accA) >= cutA, hitB = TWO && max16(accB) >= cutB;
    if (__ballot(hitA || hitB)) {
#pragma unroll
      for (int r = 0; r < 16; r++) {
        const uint32_t c = t * 32u + (uint32_t)((r & 3) + 8 * (r >> 2)) + 4u * h;
        if (accA[r] >= cutA) append(rlA, c);
        if (TWO && accB[r] >= cutB) append(rlB, c);
      }
    }
  };
  for (uint32_t sp = 0; sp < nsuper; sp++) {
    const int buf = sp & 1;
    const bool stage = sp + 1 < nsuper;
    const uint32_t base = buf * SUPB, bb = bias0 + buf * 256;
    tile_pass(base, bb, 2 * sp, stage, sp + 1, buf ^ 1);
    tile_pass(base + 32 * ROWB, bb + 128, 2 * sp + 1, false, sp + 1, buf ^ 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();   // also: every contender list is complete after the last one
  }

  // ---- contenders in fp32 + the decision ----
  const float cmaxc = sqrtf(__uint_as_float(stats[0])) * 1.000001f;
  const float bmaxc = __uint_as_float(stats[1]);
  const float cmaxo = sqrtf(__uint_as_float(stats[2])) * 1.000001f;
  const float u = 5.9604645e-8f;
  auto settle = [&](uint32_t s, bool live, uint32_t rl) {
    // my half of the centred fp32 row, 64 features at a time, against every contender of the row; a contender's
    // partial dot products add up in a register of its own (at most kRefineCap of them).  (128 features at a time
    // had the row chunk AND a contender's 32 sixteen-byte loads in flight: 256 registers, 650 bytes of scratch
    // per lane -- a third of a gigabyte of spill traffic per 8M-row pass.  Requesting contender i + 1's chunk and the
    // row's next chunk one step ahead -- 32-feature chunks, two buffers -- measured no better: 481 against 464 us per
    // 8M-row pass, profiles/r3j_*.)
    constexpr int FC = NKH < 64 ? NKH : 64, NCHUNK = NKH / FC;
    float xn2 = 0.f, xo2 = 0.f, x0 = 0.f;
    float v1 = -INFINITY, v2 = -INFINITY, v3 = -INFINITY;
    uint32_t i1 = 0xFFFFFFFFu, i2 = 0xFFFFFFFFu;
    float part[kRefineCap];
#pragma unroll
    for (int i = 0; i < kRefineCap; i++) part[i] = 0.f;
    const uint32_t n = *lds_u32(cnt_lds + rl * 4);
    const bool usable = n >= 1 && n <= (uint32_t)kRefineCap;
    uint32_t nmax = usable ? n : 0u;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) nmax = max(nmax, (uint32_t)__shfl_xor((int)nmax, off));
    // (a rolled loop: unrolled, hipcc starts the next chunk's loads before this chunk's contenders are done and
    // spills what it cannot hold; the mean comes from its copy in LDS)
#pragma unroll 1
    for (int ch = 0; ch < NCHUNK; ch++) {
      const int f0 = ch * FC;
      float xv[FC];
      const float *xr = samples + (size_t)s * D + h * NKH + f0;
      const float *m = mu + h * NKH + f0;
      const uint32_t m_lds = mu_lds + (uint32_t)(h * NKH + f0) * 4u;
#pragma unroll
      for (int f = 0; f < FC; f += 4) {
        float x4[4], m4[4];
        if (FAST) {
          const f32x4 a = *reinterpret_cast<const f32x4 *>(xr + f);
          const f32x4 b = *reinterpret_cast<const __attribute__((address_space(3))) f32x4 *>((uintptr_t)(m_lds + f * 4));
          x4[0] = a.x; x4[1] = a.y; x4[2] = a.z; x4[3] = a.w;
          m4[0] = b.x; m4[1] = b.y; m4[2] = b.z; m4[3] = b.w;
        } else {
#pragma unroll
          for (int q = 0; q < 4; q++) {
            const uint32_t ff = h * NKH + f0 + f + q;
            x4[q] = ff < D ? xr[f + q] : 0.f;
            m4[q] = m[f + q];   // DP floats, zero beyond D
          }
        }
#pragma unroll
        for (int q = 0; q < 4; q++) {
          const float xc = x4[q] - m4[q];
          xv[f + q] = xc;
          xn2 = fmaf(xc, xc, xn2);
          xo2 = fmaf(x4[q], x4[q], xo2);
        }
        if (f == 0 && ch == 0) x0 = x4[0];
      }
#pragma unroll
      for (int i = 0; i < kRefineCap; i++) {
        if ((uint32_t)i < nmax) {   // wave-uniform
          const bool on = usable && (uint32_t)i < n;
          const uint32_t c = on ? *lds_u32(list_lds + (rl * kRefineCap + i) * 4) : 0u;
          const float *cr = cfil + (size_t)c * DP + h * NKH + f0;
          float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
          for (int f = 0; f < FC; f += 4) {
            const f32x4 c4 = *reinterpret_cast<const f32x4 *>(cr + f);
            a0 = fmaf(xv[f + 0], c4.x, a0);
            a1 = fmaf(xv[f + 1], c4.y, a1);
            a2 = fmaf(xv[f + 2], c4.z, a2);
            a3 = fmaf(xv[f + 3], c4.w, a3);
          }
          part[i] += (a0 + a1) + (a2 + a3);
        }
      }
    }
#pragma unroll
    for (int i = 0; i < kRefineCap; i++) {
      if ((uint32_t)i < nmax) {
        const bool on = usable && (uint32_t)i < n;
        const uint32_t c = on ? *lds_u32(list_lds + (rl * kRefineCap + i) * 4) : 0u;
        const float v = (part[i] + __shfl_xor(part[i], 32)) + bias[c];
        if (on) {
          const bool g1 = v > v1, g2 = v > v2, g3 = v > v3;
          v3 = g2 ? v2 : (g3 ? v : v3);
          i2 = g1 ? i1 : (g2 ? c : i2);
          v2 = g1 ? v1 : (g2 ? v : v2);
          i1 = g1 ? c : i1;
          v1 = g1 ? v : v1;
        }
      }
    }
    xn2 += __shfl_xor(xn2, 32);
    xo2 += __shfl_xor(xo2, 32);
    x0 = __shfl(x0, col);   // feature 0 lives in the lower half-wave; only its NaN-ness matters (kmeans.cu:312)
    const bool insane = (x0 != x0);
    const float xn = sqrtf(xn2) * 1.0001f, xo = sqrtf(xo2) * 1.0001f;
    const float e_mfma = 2.0f * eps * (xn * cmaxc + bmaxc);
    const float e_ref = u * (12.0f * xo * cmaxo + 4.0f * cmaxo * cmaxo);
    const float thr = 2.0f * (e_mfma + e_ref) * 1.001f + tie_slack;
    // (ranges: the contender lists came out of half operands)
    const bool in_range = usable && (xn < 6.0e4f) && (cmaxc < 6.0e4f) && i1 < K;
    const bool certain = insane || (in_range && ((v1 - v2) > thr));
    const bool two = !certain && in_range && ((v1 - v3) > thr) && i2 < K;
    const bool mine = (h == 0) && live;
    const bool pair_now = mine && two, flag_now = mine && !certain && !two;
    bool changed = false;
    if (mine && certain) changed = commit_row(s, insane ? K : i1, assignments, assignments_prev);
    const unsigned long long cm = __ballot(changed), pm = __ballot(pair_now), fm = __ballot(flag_now);
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
  };
  // (Round 3 also scored (row, contender) PAIRS with 8 lanes each -- rows staged coalesced through the idle tile
  // buffers 16 at a time, 128 contiguous bytes of a contender's row per load instruction instead of 64 cache
  // lines.  Same decisions, stage 2 0.70 ms against 0.61 on the same box (profiles/r3g_*): the phase is a chain of
  // dependent round trips, not address-path throughput, and the batching added more of them.  Removed.)
  settle(sA, liveA, rlA);
  if constexpr (TWO) settle(sB, liveB, rlB);
  __syncthreads();   // the lists and tile buffers are reused by the next group
  }
}

template <int DP, int NSET>
static hipError_t launch_refine_dp_n(const LloydArgs &a, const void *rows, bool half_rows, const void *panelhi,
                                     const uint32_t *row_list, const float *thr_list, const uint32_t *n_list,
                                     uint32_t rows_hint, hipStream_t st) {
  const size_t lds_bytes = 2 * 64 * (size_t)(DP * 2) + 512 + 64 + (size_t)DP * 4 + 1024 + 256 * kRefineCap * 4;
  const uint32_t rows_per_block = 128u * NSET;
  // blocks beyond the device-side list length leave at once, but dispatching 31k of them for a list
  // 2k long is not free: the grid follows the caller's estimate of the list (the kernel strides)
  uint32_t grid = (a.N + rows_per_block - 1) / rows_per_block;
  if (rows_hint != 0xFFFFFFFFu) {
    const uint32_t want = rows_hint / rows_per_block + rows_hint / (4 * rows_per_block) + 64;
    if (want < grid) grid = want;
  }
  const bool fast = a.D == (uint32_t)DP;
#define KMX_RFN_LAUNCH(H, F)                                                                                       \
  hipLaunchKernelGGL((lloyd_refine_kernel<DP, H, F, NSET>), dim3(grid), dim3(256), lds_bytes, st, rows, a.samples,  \
                     a.N, a.D, reinterpret_cast<const float *>(panelhi), a.cfil, a.bias, a.mu, a.K_pad, a.K,        \
                     a.stats, a.eps, a.tie_slack, a.assignments, a.assignments_prev, row_list, thr_list, n_list,    \
                     a.flagged, a.pairs, a.counters)
  if (half_rows) {
    if (fast) KMX_RFN_LAUNCH(true, true); else KMX_RFN_LAUNCH(true, false);
  } else {
    if (fast) KMX_RFN_LAUNCH(false, true); else KMX_RFN_LAUNCH(false, false);
  }
#undef KMX_RFN_LAUNCH
  return hipGetLastError();
}

template <int DP>
static hipError_t launch_refine_dp(const LloydArgs &a, const void *rows, bool half_rows, const void *panelhi,
                                   const uint32_t *row_list, const float *thr_list, const uint32_t *n_list,
                                   uint32_t rows_hint, hipStream_t st) {
  if constexpr (DP > 256) {
    return launch_refine_dp_n<DP, 1>(a, rows, half_rows, panelhi, row_list, thr_list, n_list, rows_hint, st);
  } else if constexpr (DP >= 64) {
    // a short list in 256-row blocks is one block per CU, one wave per SIMD, and the kernel's gather /
    // contender phases are latency: 128-row blocks put two on every CU -- while they all fit in ONE round
    // of 512 resident blocks.  Beyond that (an 8-GPU shard's ~70k rows: 547 blocks, the second round
    // nearly empty, 0.10 ms) a single round of 256-row blocks is faster again
    if (rows_hint != 0xFFFFFFFFu && rows_hint <= 128u * 500u)
      return launch_refine_dp_n<DP, 1>(a, rows, half_rows, panelhi, row_list, thr_list, n_list, rows_hint, st);
    return launch_refine_dp_n<DP, 2>(a, rows, half_rows, panelhi, row_list, thr_list, n_list, rows_hint, st);
  } else {
    return launch_refine_dp_n<DP, 2>(a, rows, half_rows, panelhi, row_list, thr_list, n_list, rows_hint, st);
  }
}

hipError_t launch_lloyd_refine(const LloydArgs &a, const void *rows, bool half_rows, const void *panelhi,
                               const uint32_t *row_list, const float *thr_list, const uint32_t *n_list,
                               uint32_t rows_hint, hipStream_t st) {
  switch (a.DP) {
    case 16: return launch_refine_dp<16>(a, rows, half_rows, panelhi, row_list, thr_list, n_list, rows_hint, st);
    case 32: return launch_refine_dp<32>(a, rows, half_rows, panelhi, row_list, thr_list, n_list, rows_hint, st);
    case 64: return launch_refine_dp<64>(a, rows, half_rows, panelhi, row_list, thr_list, n_list, rows_hint, st);
    case 128: return launch_refine_dp<128>(a, rows, half_rows, panelhi, row_list, thr_list, n_list, rows_hint, st);
    case 256: return launch_refine_dp<256>(a, rows, half_rows, panelhi, row_list, thr_list, n_list, rows_hint, st);
    case 512: return launch_refine_dp<512>(a, rows, half_rows, panelhi, row_list, thr_list, n_list, rows_hint, st);
    default: return hipErrorInvalidValue;
  }
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
  float xn2 = 0.f, dx2 = 0.f, x0 = 0.f;
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
      const float xc = live ? xv[q] - mu[h * NKH + 8 * j + q] : 0.f;   // mu: DP floats, zero beyond D
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
