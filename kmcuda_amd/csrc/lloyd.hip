// lloyd.hip -- the Lloyd assignment step (reference: src/kmeans.cu:293-364 kmeans_assign_lloyd,
// :214-291 kmeans_assign_lloyd_smallc) re-designed for MI355X / gfx950.
//
// The reference decides  nearest[s] = first c minimising  RD(-2*prod(s,c) + csqr(c))  with
// prod / csqr Kahan-compensated round-down-FMA chains (4 dependent VALU ops per MAC).  That
// arithmetic can never approach the matrix-core roofline, so the step is split:
//
//   1. centroid_prep     K*D work: exact csqr[c] (the reference's sum_squares), a sanitised,
//                        padded centroid panel for the filter, its transpose for the refine
//                        kernel, and the two magnitudes the error bound needs.
//   2. lloyd_filter      samples x centroids^T on the f32 MFMA (v_mfma_f32_32x32x2_f32).  Each
//                        wave keeps 32 samples resident in VGPRs as the B operand and streams
//                        32-centroid A tiles through LDS; the accumulator is seeded with
//                        -csqr/2 so the running per-sample (max, argmax, second max) of
//                        score = x.c - csqr/2 falls out of the accumulator registers.  A row
//                        whose best/second-best gap exceeds a rigorous bound on
//                        |score_mfma - score_reference| is decided (and committed) here.
//   3. lloyd_exact       the (rare) undecided rows are recomputed with the reference's exact
//                        arithmetic, one wave per row, and committed.
//
// Assignments are therefore bit-identical to the reference's for ANY input; only the split of
// work between 2 and 3 depends on the data.  See DESIGN.md for the bound's derivation.
#include "exact.hpp"
#include "filter_common.hpp"
#include "kernels.hpp"

namespace kmx {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// ---------------------------------------------------------------------------------------
// 1. centroid_prep
// ---------------------------------------------------------------------------------------
// rows: one thread per centroid row.  csqr follows metric_abstraction.h:21-36 exactly (L2) or is
// the constant 1 (:149-158, angular); finite[c] = 0 for a row holding a NaN or an inf (never
// chosen by the reference: its distance is NaN or +inf and "dist < min_dist" is false).
template <int METRIC>
__global__ void centroid_rows_kernel(const float *__restrict__ centroids, uint32_t K, uint32_t D, uint32_t Kt,
                                     float *__restrict__ csqr, float *__restrict__ ct,
                                     uint32_t *__restrict__ finite, uint32_t *__restrict__ stats) {
  const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
  float plain_max = 0.f;  // this lane's contribution to stats[2]
  if (c < Kt && c < K) {
    const float *row = centroids + (size_t)c * D;
    float ssqr = 0.f, corr = 0.f, plain = 0.f;
    uint32_t f = 0;
    if ((D & 3u) == 0 && (((uintptr_t)row) & 15u) == 0) {
      for (; f < D; f += 4) {  // 16-byte loads; the chain order is unchanged
        const float4 v4 = *reinterpret_cast<const float4 *>(row + f);
        const float vs[4] = {v4.x, v4.y, v4.z, v4.w};
#pragma unroll
        for (int q = 0; q < 4; q++) {
          if (METRIC == 0) kahan_fold(fma_rd(vs[q], vs[q], corr), ssqr, corr);
          plain = fmaf(vs[q], vs[q], plain);
          if (ct) ct[(size_t)(f + q) * Kt + c] = vs[q];
        }
      }
    }
    for (; f < D; f++) {
      const float v = row[f];
      if (METRIC == 0) kahan_fold(fma_rd(v, v, corr), ssqr, corr);
      plain = fmaf(v, v, plain);
      if (ct) ct[(size_t)f * Kt + c] = v;
    }
    const bool fin = (plain - plain) == 0.f;  // false for NaN and inf
    if (finite) finite[c] = fin ? 1u : 0u;
    if (csqr) csqr[c] = (METRIC == 0) ? ssqr : 1.f;
    // uncentered max ||c||^2 for the bound on the REFERENCE's own rounding error
    if (fin) plain_max = plain * 1.0001f;
  } else if (c < Kt && ct) {
    for (uint32_t f = 0; f < D; f++) ct[(size_t)f * Kt + c] = 0.f;
  }
  // one atomic per wave, not per centroid (K same-address atomics cost more than the kernel)
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) plain_max = fmaxf(plain_max, __shfl_xor(plain_max, off));
  if (stats && (threadIdx.x & 63) == 0 && plain_max > 0.f) atomicMax(&stats[2], __float_as_uint(plain_max));
}

// mean of the finite centroid rows, one thread per (padded) feature.  Any vector would do: the
// argmin is translation invariant, the mean just makes the centred norms (and so the filter's
// error bound) small.
__global__ __launch_bounds__(1024) void centroid_mean_kernel(const float *__restrict__ centroids, uint32_t K,
                                                             uint32_t D, uint32_t DP,
                                                             const uint32_t *__restrict__ finite,
                                                             float *__restrict__ mu, bool freeze_mu,
                                                             uint32_t *__restrict__ zero_a,
                                                             uint32_t *__restrict__ zero_b,
                                                             uint32_t *__restrict__ zero_c) {
  // block = 64 features x 16 row-slices: coalesced row reads, fixed summation order
  constexpr int S = 16;
  __shared__ float part[S][64];
  __shared__ uint32_t cnt[S];
  const uint32_t fl = threadIdx.x & 63, sl = threadIdx.x >> 6;
  const uint32_t f = blockIdx.x * 64 + fl;
  if (blockIdx.x == 0 && threadIdx.x == 0) {  // the filter's per-pass list counters (saves two memset launches)
    if (zero_a) *zero_a = 0u;
    if (zero_b) *zero_b = 0u;
    if (zero_c) { *zero_c = 0u; zero_c[kDuoCount - 4] = 0u; }   // (zero_c = counters + 4: the undecided list, and the duo list)
  }
  if (freeze_mu) return;  // the engine's row cache holds x - mu: mu stays what it was (any mu is valid)
  float sum = 0.f;
  uint32_t n = 0;
  for (uint32_t c = sl; c < K; c += S) {
    if (finite[c]) {
      if (f < D) sum += centroids[(size_t)c * D + f];
      n++;
    }
  }
  part[sl][fl] = sum;
  if (fl == 0) cnt[sl] = n;
  __syncthreads();
  if (sl == 0 && f < DP) {
    float tot = 0.f;
    uint32_t nt = 0;
#pragma unroll
    for (int i = 0; i < S; i++) {
      tot += part[i][fl];
      nt += cnt[i];
    }
    const float m = (nt && f < D) ? tot / (float)nt : 0.f;
    mu[f] = ((m - m) == 0.f) ? m : 0.f;
  }
}

// panel: one thread per padded centroid row: cfil = c - mu (zero row + bias -inf if not finite),
// bias = -||c'||^2/2 (L2) or mu.c' (angular: x.c = x'.c' + mu.c' + terms constant in c), and the
// two magnitudes of the centred bound.
template <int METRIC>
__global__ __launch_bounds__(256) void centroid_panel_kernel(const float *__restrict__ centroids, uint32_t K,
                                                             uint32_t D, uint32_t K_pad, uint32_t DP,
                                                             const uint32_t *__restrict__ finite,
                                                             const float *__restrict__ mu, float *__restrict__ bias,
                                                             float *__restrict__ bias2, float *__restrict__ cfil,
                                                             uint32_t *__restrict__ stats) {
  // one WAVE per padded centroid row (4 rows per wave, 16 per block): coalesced reads / writes, plain
  // (order-free) sums by shuffles; the four maxima go out once per block -- per-row same-address
  // atomics used to be most of this kernel's 49 us
  const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // maxima as BITS: the values are non-negative, so bits order like values, and a NaN (0 * inf in a
  // bound term) sorts above inf and poisons the bound -- as the per-row atomicMax on bits did
  uint32_t s0 = 0, s1 = 0, s3 = 0, s4 = 0;
  for (uint32_t i = 0; i < 4; i++) {
    const uint32_t c = blockIdx.x * 16 + wave * 4 + i;
    if (c >= K_pad) break;
    float *dst = cfil + (size_t)c * DP;
    const bool ok = c < K && finite[c];
    float n2 = 0.f, mc = 0.f, m2 = 0.f;
    for (uint32_t f = lane; f < DP; f += 64) {
      float v = 0.f, m = 0.f;
      if (ok && f < D) {
        m = mu[f];
        v = centroids[(size_t)c * D + f] - m;
      }
      dst[f] = v;
      n2 = fmaf(v, v, n2);
      mc = fmaf(m, v, mc);
      m2 = fmaf(m, m, m2);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      n2 += __shfl_xor(n2, off);
      mc += __shfl_xor(mc, off);
      m2 += __shfl_xor(m2, off);
    }
    if (ok) {
      float b, bmag;
      if (METRIC == 0) {
        b = -0.5f * n2;
        bmag = 0.5f * n2;
      } else {
        b = mc;
        bmag = sqrtf(m2) * sqrtf(n2) * 1.0001f;  // >= sum |mu_f c'_f|
      }
      // the variant for kernels that keep the ORIGINAL row resident:
      // ||x - c||^2 = ||x - mu||^2 - 2 (x.c' - mu.c' - ||c'||^2/2)   /   x.c = x.c' + x.mu
      const float mag2 = (METRIC == 0) ? sqrtf(m2) * sqrtf(n2) * 1.0001f + 0.5f * n2 : 0.f;
      if (lane == 0) {
        bias[c] = b;
        bias2[c] = (METRIC == 0) ? -mc - 0.5f * n2 : 0.f;
      }
      s0 = max(s0, __float_as_uint(n2 * 1.0001f));
      s1 = max(s1, __float_as_uint(bmag * 1.0001f));
      s3 = max(s3, __float_as_uint(m2 * 1.0001f));
      s4 = max(s4, __float_as_uint(mag2 * 1.0001f));
    } else if (lane == 0) {
      bias[c] = -INFINITY;
      bias2[c] = -INFINITY;
    }
  }
  __shared__ uint32_t red[4][4];
  if (lane == 0) { red[wave][0] = s0; red[wave][1] = s1; red[wave][2] = s3; red[wave][3] = s4; }
  __syncthreads();
  if (threadIdx.x < 4) {
    const uint32_t m = max(max(red[0][threadIdx.x], red[1][threadIdx.x]), max(red[2][threadIdx.x], red[3][threadIdx.x]));
    const uint32_t slot = threadIdx.x < 2 ? threadIdx.x : threadIdx.x + 1;  // stats[0], [1], [3], [4]
    atomicMax(&stats[slot], m);
  }
}

// ---------------------------------------------------------------------------------------
// 2. lloyd_filter (MFMA)
// ---------------------------------------------------------------------------------------
// Block = 256 threads = 4 waves, each wave owns 32 consecutive samples.
// MFMA orientation: A = centroids (32 rows), B = samples (32 columns), so that lane l holds,
// for ITS sample (column l&31), the scores of 16 centroids (rows (r&3)+8*(r>>2)+4*(l>>5)).
// The contraction index is permuted (legal: both operands use the same permutation): at
// k-step kk the lower half-wave supplies feature kk, the upper half feature DP/2 + kk, so
// every lane loads one contiguous half row of its sample / reads contiguous LDS words.
template <int DP, bool FAST>
__global__ __launch_bounds__(256, 2) void lloyd_filter_kernel(
    const float *__restrict__ samples, uint32_t N, uint32_t D, const float *__restrict__ cfil,
    const float *__restrict__ bias, const float *__restrict__ mu, uint32_t K_pad, uint32_t K,
    const uint32_t *__restrict__ stats, float eps, float tie_slack, uint32_t *__restrict__ assignments,
    uint32_t *__restrict__ assignments_prev, uint32_t *__restrict__ flagged, uint32_t *__restrict__ pairs,
    uint32_t *__restrict__ counters) {
  constexpr int NK = DP / 2;           // k-steps (each MFMA consumes 2 features)
  constexpr int LDW = DP + 4;          // padded LDS row (floats): conflict-free ds_read_b128
  constexpr int TILE = 32 * LDW;       // one centroid tile
  constexpr int NST = (8 * DP + 255) / 256;  // float4 staging registers per thread
  extern __shared__ __attribute__((aligned(16))) float lds[];
  if (counters[kStopFlag] != 0u) return;   // the run has stopped on the device (apply_delta_kernel): touch nothing
  auto tile_ptr = [&](int buf) { return lds + buf * TILE; };
  auto bias_ptr = [&](int buf) { return lds + 2 * TILE + buf * 32; };

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int col = lane & 31;  // my sample within the wave
  const int h = lane >> 5;    // which half of the features / which centroid rows
  const uint32_t s = blockIdx.x * 128u + wave * 32u + col;

  // ---- B operand: my half row of my sample MINUS the centroid mean, resident for the whole
  // kernel (the argmin is translation invariant; centring shrinks the norms in the error bound)
  float xb[NK];
  float xo2 = 0.f;  // squared norm of the ORIGINAL half row (bound on the reference's own error)
  float xm = 0.f, xma = 0.f;  // x.mu and sum |x_f mu_f|: score + x.mu = the product (angular clamp, filter_common.hpp)
  {
    const bool live = s < N;
    if (FAST) {
      const f32x4 *src = reinterpret_cast<const f32x4 *>(samples + (size_t)(live ? s : 0) * D + h * NK);
      const f32x4 *msrc = reinterpret_cast<const f32x4 *>(mu + h * NK);
#pragma unroll
      for (int j = 0; j < NK / 4; j++) {
        const f32x4 v = src[j], m = msrc[j];
        xo2 = fmaf(v.x, v.x, xo2); xo2 = fmaf(v.y, v.y, xo2); xo2 = fmaf(v.z, v.z, xo2); xo2 = fmaf(v.w, v.w, xo2);
        xm = fmaf(v.x, m.x, xm); xm = fmaf(v.y, m.y, xm); xm = fmaf(v.z, m.z, xm); xm = fmaf(v.w, m.w, xm);
        xma = fmaf(fabsf(v.x), fabsf(m.x), xma); xma = fmaf(fabsf(v.y), fabsf(m.y), xma);
        xma = fmaf(fabsf(v.z), fabsf(m.z), xma); xma = fmaf(fabsf(v.w), fabsf(m.w), xma);
        xb[4 * j + 0] = live ? v.x - m.x : 0.f;
        xb[4 * j + 1] = live ? v.y - m.y : 0.f;
        xb[4 * j + 2] = live ? v.z - m.z : 0.f;
        xb[4 * j + 3] = live ? v.w - m.w : 0.f;
      }
    } else {
      const float *src = samples + (size_t)(live ? s : 0) * D;
#pragma unroll
      for (int j = 0; j < NK; j++) {
        const uint32_t f = h * NK + j;
        const float v = (live && f < D) ? src[f] : 0.f;
        const float mf = (f < D) ? mu[f] : 0.f;
        xo2 = fmaf(v, v, xo2);
        xm = fmaf(v, mf, xm);
        xma = fmaf(fabsf(v), fabsf(mf), xma);
        xb[j] = (live && f < D) ? v - mf : 0.f;
      }
    }
    if (!live) xo2 = 0.f;
  }
  // squared norm of the centred sample (for the error bound) and the "insane" test of kmeans.cu:312
  float xn2 = 0.f;
#pragma unroll
  for (int j = 0; j < NK; j++) xn2 = fmaf(xb[j], xb[j], xn2);
  xn2 += __shfl_xor(xn2, 32);
  xo2 += __shfl_xor(xo2, 32);
  xm += __shfl_xor(xm, 32);
  xma += __shfl_xor(xma, 32);
  const float x0 = __shfl(xb[0], col);  // feature 0 lives in the lower half-wave (NaN - mu = NaN)
  const bool insane = (x0 != x0);

  // ---- staging of centroid tiles: global -> registers -> LDS (double buffered) ----
  f32x4 stage[NST];
  float bstage = 0.f;
  auto stage_load = [&](uint32_t tile) {
    const float *src = cfil + (size_t)tile * 32 * DP;
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

  // running top-3 of my centroid rows: (v1,c1) best, (v2,c2) second, v3 third value
  float v1 = -INFINITY, v2 = -INFINITY, v3 = -INFINITY;
  uint32_t c1 = 0xFFFFFFFFu, c2 = 0xFFFFFFFFu;  // codes: tile*16 + accumulator register

  for (uint32_t t = 0; t < ntiles; t++) {
    const int buf = t & 1;
    if (t + 1 < ntiles) stage_load(t + 1);

    // accumulator seeded with the bias of my 16 centroid rows
    f32x16 acc;
    {
      const float *bb = bias_ptr(buf) + 4 * h;
#pragma unroll
      for (int g = 0; g < 4; g++) {
        const f32x4 b4 = *reinterpret_cast<const f32x4 *>(bb + 8 * g);
        acc[4 * g + 0] = b4.x;
        acc[4 * g + 1] = b4.y;
        acc[4 * g + 2] = b4.z;
        acc[4 * g + 3] = b4.w;
      }
    }
    const float *arow = tile_ptr(buf) + col * LDW + h * NK;
    // A fragments are read two 16-byte pieces (8 MFMAs = 512 cycles) AHEAD of their use and the
    // order is pinned with sched_barrier, so the LDS latency never lands on the dependent MFMA chain
    // (left alone, hipcc sinks each ds_read_b128 pair right in front of its first MFMA)
    constexpr int NG = NK / 8 > 0 ? NK / 8 : 1;   // groups of 8 k-steps
    if constexpr (NK >= 16) {
      f32x4 c0 = *reinterpret_cast<const f32x4 *>(arow), c1 = *reinterpret_cast<const f32x4 *>(arow + 4);
#pragma unroll
      for (int g = 0; g < NG; g++) {
        f32x4 n0 = c0, n1 = c1;
        if (g + 1 < NG) {
          n0 = *reinterpret_cast<const f32x4 *>(arow + 8 * (g + 1));
          n1 = *reinterpret_cast<const f32x4 *>(arow + 8 * (g + 1) + 4);
        }
        __builtin_amdgcn_sched_barrier(0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(c0.x, xb[8 * g + 0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(c0.y, xb[8 * g + 1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(c0.z, xb[8 * g + 2], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(c0.w, xb[8 * g + 3], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(c1.x, xb[8 * g + 4], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(c1.y, xb[8 * g + 5], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(c1.z, xb[8 * g + 6], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(c1.w, xb[8 * g + 7], acc, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        c0 = n0;
        c1 = n1;
      }
    } else {
#pragma unroll
      for (int j = 0; j < NK / 4; j++) {
        const f32x4 a4 = *reinterpret_cast<const f32x4 *>(arow + 4 * j);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.x, xb[4 * j + 0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.y, xb[4 * j + 1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.z, xb[4 * j + 2], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.w, xb[4 * j + 3], acc, 0, 0, 0);
      }
    }
    // strict '>' keeps the earlier index on equal scores; NaN scores compare false everywhere
    // and are ignored (as in the reference, where a NaN distance is never "less").
#pragma unroll
    for (int r = 0; r < 16; r++) {
      const float v = acc[r];
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

  // ---- merge the half-waves, decide, commit / hand over (filter_common.hpp) ----
  const float cmaxc = sqrtf(__uint_as_float(stats[0])) * 1.000001f;
  const float bmaxc = __uint_as_float(stats[1]);
  const float cmaxo = sqrtf(__uint_as_float(stats[2])) * 1.000001f;
  const float xn = sqrtf(xn2) * 1.0001f, xo = sqrtf(xo2) * 1.0001f;
  const float e_mfma = 2.0f * eps * (xn * cmaxc + bmaxc);
  const float e_ref = 5.9604645e-8f * (12.0f * xo * cmaxo + 4.0f * cmaxo * cmaxo);
  const float thr = 2.0f * (e_mfma + e_ref) * 1.001f + tie_slack;
  // (tie_slack != 0 <=> the angular metric, engine.cpp)
  const ClampLimits lim = clamp_limits(tie_slack > 0.f, xm, dot_error(DP, xma), 0.5f * thr);
  filter_finish(v1, v2, v3, c1, c2, h, lane, s, N, K, insane, thr, lim, assignments, assignments_prev, flagged, pairs,
                counters);
}

// ---------------------------------------------------------------------------------------
// 2b. lloyd_pair: rows whose minimum is one of two centroids -- two exact Kahan chains per
// thread (the reference's arithmetic), the smaller distance wins, the smaller index on a tie
// (== the reference's ascending scan with strict '<').
// ---------------------------------------------------------------------------------------
template <int METRIC>
__global__ __launch_bounds__(128) void lloyd_pair_kernel(
    const float *__restrict__ samples, uint32_t D, const float *__restrict__ centroids,
    const float *__restrict__ csqr, const uint32_t *__restrict__ pairs, const uint32_t *__restrict__ npairs,
    uint32_t *__restrict__ assignments, uint32_t *__restrict__ assignments_prev, uint32_t *__restrict__ counters) {
  const uint32_t total = *npairs;
  for (uint32_t pi = blockIdx.x * blockDim.x + threadIdx.x; pi < total; pi += gridDim.x * blockDim.x) {
    const uint32_t s = pairs[3 * (size_t)pi], ia = pairs[3 * (size_t)pi + 1], ib = pairs[3 * (size_t)pi + 2];
    const uint32_t lo = ia < ib ? ia : ib, hi = ia < ib ? ib : ia;
    const float *x = samples + (size_t)s * D;
    const float *ca = centroids + (size_t)lo * D, *cb = centroids + (size_t)hi * D;
    float acc[4] = {0.f, 0.f, 0.f, 0.f}, corr[4] = {0.f, 0.f, 0.f, 0.f};
    uint32_t f = 0;
    const bool aligned16 = ((((uintptr_t)x) | ((uintptr_t)ca) | ((uintptr_t)cb)) & 15u) == 0;
    if ((D & 7u) == 0 && aligned16) {
      // 8 features per group, the next group's six 16-byte loads issued before the current group's
      // dependent steps (the three rows are scattered: every group is a round trip to L2 / HBM)
      auto load8 = [&](uint32_t f0, float4 (&xv)[2], float4 (&av)[2], float4 (&bv)[2]) {
#pragma unroll
        for (int t = 0; t < 2; t++) {
          xv[t] = *reinterpret_cast<const float4 *>(x + f0 + 4 * t);
          av[t] = *reinterpret_cast<const float4 *>(ca + f0 + 4 * t);
          bv[t] = *reinterpret_cast<const float4 *>(cb + f0 + 4 * t);
        }
      };
      auto fold8 = [&](const float4 (&xv)[2], const float4 (&av)[2], const float4 (&bv)[2]) {
#pragma unroll
        for (int t = 0; t < 2; t++) {
          const float xs[4] = {xv[t].x, xv[t].y, xv[t].z, xv[t].w}, as[4] = {av[t].x, av[t].y, av[t].z, av[t].w},
                      bs[4] = {bv[t].x, bv[t].y, bv[t].z, bv[t].w};
#pragma unroll
          for (int q = 0; q < 4; q++) {
            const float cv[4] = {as[q], bs[q], 0.f, 0.f};
            float y[4];
            fma_rd4(xs[q], cv, corr, y);
            kahan_fold(y[0], acc[0], corr[0]);
            kahan_fold(y[1], acc[1], corr[1]);
          }
        }
      };
      float4 x0[2], a0[2], b0[2], x1[2], a1[2], b1[2];
      load8(0, x0, a0, b0);
      for (; f + 16 <= D; f += 16) {
        load8(f + 8, x1, a1, b1);
        fold8(x0, a0, b0);
        if (f + 24 <= D) load8(f + 16, x0, a0, b0);
        fold8(x1, a1, b1);
      }
      if (f + 8 <= D) {  // an odd number of groups: the last one is already loaded
        fold8(x0, a0, b0);
        f += 8;
      }
    } else if ((D & 3u) == 0 && aligned16) {
      for (; f < D; f += 4) {  // 16-byte loads; the chain order is unchanged
        const float4 xv = *reinterpret_cast<const float4 *>(x + f);
        const float4 av = *reinterpret_cast<const float4 *>(ca + f), bv = *reinterpret_cast<const float4 *>(cb + f);
        const float xs[4] = {xv.x, xv.y, xv.z, xv.w}, as[4] = {av.x, av.y, av.z, av.w}, bs[4] = {bv.x, bv.y, bv.z, bv.w};
#pragma unroll
        for (int q = 0; q < 4; q++) {
          const float cv[4] = {as[q], bs[q], 0.f, 0.f};
          float y[4];
          fma_rd4(xs[q], cv, corr, y);
          kahan_fold(y[0], acc[0], corr[0]);
          kahan_fold(y[1], acc[1], corr[1]);
        }
      }
    }
    for (; f < D; f++) {
      const float cv[4] = {ca[f], cb[f], 0.f, 0.f};
      float y[4];
      fma_rd4(x[f], cv, corr, y);
      kahan_fold(y[0], acc[0], corr[0]);
      kahan_fold(y[1], acc[1], corr[1]);
    }
    const float da = lloyd_distance<METRIC>(csqr[lo], acc[0]);
    const float db = lloyd_distance<METRIC>(csqr[hi], acc[1]);
    // ascending scan, strict '<', starting from FLT_MAX
    float min_dist = 3.402823466e+38f;
    uint32_t nearest = 0xFFFFFFFFu;
    if (da < min_dist) { min_dist = da; nearest = lo; }
    if (db < min_dist) { min_dist = db; nearest = hi; }
    if (nearest != 0xFFFFFFFFu && commit_row(s, nearest, assignments, assignments_prev)) atomicAdd(&counters[0], 1u);
  }
}

// ---------------------------------------------------------------------------------------
// 3. lloyd_exact: the reference's arithmetic, one wave per row.
// ---------------------------------------------------------------------------------------
// rows == nullptr: every row of [0, N); else rows[0 .. *nrows).  Lane l runs CH independent
// Kahan chains, centroid (p*CH + j)*64 + l, reading the transposed panel ct[f*Kt + c]
// (coalesced).  Lane-local strict '<' in ascending c, then a wave argmin that prefers the
// smaller index on equal distances == the reference's sequential first-minimum.
template <int METRIC>
__global__ __launch_bounds__(256) void lloyd_exact_kernel(
    const float *__restrict__ samples, uint32_t N, uint32_t D, const float *__restrict__ ct,
    const float *__restrict__ csqr, uint32_t K, uint32_t Kt, const uint32_t *__restrict__ rows,
    const uint32_t *__restrict__ nrows, uint32_t *__restrict__ assignments,
    uint32_t *__restrict__ assignments_prev, uint32_t *__restrict__ counters) {
  // One BLOCK of 4 waves per row (the flagged rows are few, so the latency of a row matters, not the
  // throughput): wave w takes the centroid chunks w, w+4, ... of 64*CH centroids, lane l runs CH
  // independent chains per chunk; lane-local strict '<' in ascending c, then wave and block argmin
  // preferring the smaller index on equal distances == the reference's sequential first minimum.
  constexpr int CH = 4;
  __shared__ float sh_dist[4];
  __shared__ uint32_t sh_idx[4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (counters[kStopFlag] != 0u) return;   // stopped on the device: touch nothing
  const uint32_t total = rows ? *nrows : N;
  for (uint32_t ri = blockIdx.x; ri < total; ri += gridDim.x) {
    const uint32_t s = rows ? rows[ri] : ri;
    const float *x = samples + (size_t)s * D;  // block-uniform address: scalar loads
    const bool insane = (x[0] != x[0]);
    float min_dist = 3.402823466e+38f;
    uint32_t nearest = 0xFFFFFFFFu;
    if (!insane) {
      for (uint32_t cbase = wave * 64 * CH; cbase < K; cbase += 4 * 64 * CH) {
        float acc[CH], corr[CH];
#pragma unroll
        for (int j = 0; j < CH; j++) { acc[j] = 0.f; corr[j] = 0.f; }
        // chain j reads column cbase + 64 j + lane of the panel; a column beyond the panel reads column
        // `lane` instead (unconditional loads: a predicated load costs a branch each) and its result is
        // dropped below (c < K)
        const float *cp[CH];
#pragma unroll
        for (int j = 0; j < CH; j++) cp[j] = ct + ((cbase + 64 * j + lane < Kt) ? cbase + 64 * j + lane : (uint32_t)lane);
        uint32_t f = 0;
        // groups of GF features, the NEXT group's GF * (CH + 1) loads issued before the current group's
        // dependent steps: the panel is L2 resident (~1 us away) and a flagged row is latency, not
        // throughput -- one group in flight covers GF steps of arithmetic (GF = 4 measured 68 us per row
        // at D = 256: 64 round trips)
        constexpr int GF = 8;
        auto load_group = [&](uint32_t f0, float (&xf)[GF], float (&cv)[GF][CH]) {
#pragma unroll
          for (int q = 0; q < GF; q++) {
            xf[q] = x[f0 + q];
#pragma unroll
            for (int j = 0; j < CH; j++) cv[q][j] = cp[j][(size_t)(f0 + q) * Kt];
          }
        };
        auto fold_group = [&](const float (&xf)[GF], float (&cv)[GF][CH]) {
#pragma unroll
          for (int q = 0; q < GF; q++) {
            float y[CH];
            fma_rd4(xf[q], cv[q], corr, y);
#pragma unroll
            for (int j = 0; j < CH; j++) kahan_fold(y[j], acc[j], corr[j]);
          }
        };
        if (D >= GF) {
          float xa[GF], ca[GF][CH], xb[GF], cb[GF][CH];
          load_group(0, xa, ca);
          for (; f + 2 * GF <= D; f += 2 * GF) {
            load_group(f + GF, xb, cb);
            fold_group(xa, ca);
            if (f + 3 * GF <= D) load_group(f + 2 * GF, xa, ca);
            fold_group(xb, cb);
          }
          if (f + GF <= D) {  // an odd number of groups: the last one is already loaded
            fold_group(xa, ca);
            f += GF;
          }
        }
        for (; f < D; f++) {
          const float xf = x[f];
          float cv[CH], y[CH];
#pragma unroll
          for (int j = 0; j < CH; j++) cv[j] = cp[j][(size_t)f * Kt];
          fma_rd4(xf, cv, corr, y);
#pragma unroll
          for (int j = 0; j < CH; j++) kahan_fold(y[j], acc[j], corr[j]);
        }
#pragma unroll
        for (int j = 0; j < CH; j++) {
          const uint32_t c = cbase + 64 * j + lane;
          if (c < K) {
            const float dist = lloyd_distance<METRIC>(csqr[c], acc[j]);
            if (dist < min_dist) { min_dist = dist; nearest = c; }
          }
        }
      }
      // wave argmin, lowest index among equal minima; lanes without a candidate never win
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) {
        const float od = __shfl_xor(min_dist, off);
        const uint32_t oi = __shfl_xor(nearest, off);
        const bool take = (oi != 0xFFFFFFFFu) &&
                          (nearest == 0xFFFFFFFFu || od < min_dist || (od == min_dist && oi < nearest));
        if (take) { min_dist = od; nearest = oi; }
      }
    }
    if (lane == 0) {
      sh_dist[wave] = min_dist;
      sh_idx[wave] = nearest;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      for (int w = 1; w < 4; w++) {
        const float od = sh_dist[w];
        const uint32_t oi = sh_idx[w];
        const bool take = (oi != 0xFFFFFFFFu) &&
                          (nearest == 0xFFFFFFFFu || od < min_dist || (od == min_dist && oi < nearest));
        if (take) { min_dist = od; nearest = oi; }
      }
      if (nearest == 0xFFFFFFFFu && insane) nearest = K;  // kmeans.cu:349-356
      if (nearest != 0xFFFFFFFFu) {                       // else: "search failed", row left untouched
        if (commit_row(s, nearest, assignments, assignments_prev)) atomicAdd(&counters[0], 1u);
      }
    }
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------------------
// 4. lloyd_settle: everything the filter could not decide, in ONE launch.  Blocks [0, kSettlePairBlocks) take the
// two-contender rows, the other blocks the rows that need the full exact scan.  Both are latency problems -- one
// serial D-step chain per contender whatever the row count -- and both used to pay a round trip to L2 / HBM per 8
// features (32 of them at D = 256: ~45 us for any list length, with two launches and a fork / join of streams):
//   pairs   a wave takes 32 rows at a time, ONE chain per lane (lane l: the lower-index contender of row l,
//           lane l + 32: the other).  The 32 sample rows and the 64 centroid rows come in through LDS in chunks
//           of 64 features -- 16 lanes fetch a row's 256 bytes, coalesced; the next chunk's loads fly during the
//           chain; rows are stored 68 floats apart (conflict-free 16-byte reads by 16 consecutive rows).
//   scans   one block of 4 waves per row as lloyd_exact_kernel, with three 8-feature groups of the transposed
//           panel in flight instead of one.
// The arithmetic and the tie rules are lloyd_pair_kernel's / lloyd_exact_kernel's (the reference's).  Needs
// D % 4 == 0 and 16-byte aligned rows (else the host launches the two old kernels).
// ---------------------------------------------------------------------------------------
constexpr int kSettleFC = 64;                  // features per staged chunk
constexpr int kSettleLD = kSettleFC + 4;       // LDS row stride in floats
constexpr int kSettleRows = 96;                // per wave: 32 sample rows + 64 centroid rows
constexpr uint32_t kSettlePairBlocks = 256, kSettleScanBlocks = 256;
constexpr size_t kSettleLds = (size_t)4 * (kSettleRows * kSettleLD + kSettleRows) * sizeof(float);

template <int METRIC>
__global__ __launch_bounds__(256) void lloyd_settle_kernel(
    const float *__restrict__ samples, uint32_t N, uint32_t D, const float *__restrict__ centroids,
    const float *__restrict__ ct, const float *__restrict__ csqr, uint32_t K, uint32_t Kt,
    const uint32_t *__restrict__ pairs, const uint32_t *__restrict__ npairs, const uint32_t *__restrict__ flagged,
    const uint32_t *__restrict__ nflagged, uint32_t *__restrict__ assignments,
    uint32_t *__restrict__ assignments_prev, uint32_t *__restrict__ counters) {
  extern __shared__ __attribute__((aligned(16))) float settle_lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (blockIdx.x < kSettlePairBlocks) {
    // ---------------- pairs ----------------
    const uint32_t total = *npairs;
    float *tile = settle_lds + (size_t)wave * (kSettleRows * kSettleLD + kSettleRows);
    uint32_t *rowid = reinterpret_cast<uint32_t *>(tile + kSettleRows * kSettleLD);
    const uint32_t nch = (D + kSettleFC - 1) / kSettleFC;
    // (every wave of the block makes the same number of trips: the barriers below are block barriers)
    for (uint32_t base0 = blockIdx.x * 128u; base0 < total; base0 += kSettlePairBlocks * 128u) {
      const uint32_t p = base0 + wave * 32u + (lane & 31);
      const bool live = p < total;
      uint32_t s = 0, lo = 0, hi = 0;
      if (live) {
        s = pairs[3 * (size_t)p];
        const uint32_t ia = pairs[3 * (size_t)p + 1], ib = pairs[3 * (size_t)p + 2];
        lo = ia < ib ? ia : ib;
        hi = ia < ib ? ib : ia;
      }
      const uint32_t mine = lane < 32 ? lo : hi;
      __syncthreads();   // the previous trip's tile and row table are no longer read
      if (lane < 32) rowid[lane] = s;
      rowid[32 + lane] = mine;
      __syncthreads();
      float4 stage[kSettleRows / 4];
      auto fetch = [&](uint32_t ch) {
#pragma unroll
        for (int it = 0; it < kSettleRows / 4; it++) {
          const uint32_t r = it * 4 + (lane >> 4), f = ch * kSettleFC + (lane & 15) * 4;
          const float *src = (r < 32 ? samples : centroids) + (size_t)rowid[r] * D + f;
          stage[it] = f < D ? *reinterpret_cast<const float4 *>(src) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
      };
      float acc = 0.f, corr = 0.f;
      fetch(0);
      for (uint32_t ch = 0; ch < nch; ch++) {
        __syncthreads();   // the previous chunk has been consumed
#pragma unroll
        for (int it = 0; it < kSettleRows / 4; it++) {
          const uint32_t r = it * 4 + (lane >> 4);
          *reinterpret_cast<float4 *>(&tile[r * kSettleLD + (lane & 15) * 4]) = stage[it];
        }
        __syncthreads();
        if (ch + 1 < nch) fetch(ch + 1);
        const uint32_t fmax = D - ch * kSettleFC < (uint32_t)kSettleFC ? D - ch * kSettleFC : (uint32_t)kSettleFC;   // multiple of 4
        const float *xr = tile + (lane & 31) * kSettleLD, *cr = tile + (32 + lane) * kSettleLD;
        for (uint32_t c4 = 0; c4 * 4 < fmax; c4++) {
          const float4 xv = *reinterpret_cast<const float4 *>(xr + c4 * 4);
          const float4 cv = *reinterpret_cast<const float4 *>(cr + c4 * 4);
          kahan_fold(fma_rd(xv.x, cv.x, corr), acc, corr);
          kahan_fold(fma_rd(xv.y, cv.y, corr), acc, corr);
          kahan_fold(fma_rd(xv.z, cv.z, corr), acc, corr);
          kahan_fold(fma_rd(xv.w, cv.w, corr), acc, corr);
        }
      }
      const float dist = lloyd_distance<METRIC>(csqr[mine], acc);
      const float db = __shfl(dist, (lane & 31) + 32);
      bool changed = false;
      if (live && lane < 32) {
        // ascending scan, strict '<', starting from FLT_MAX
        float min_dist = 3.402823466e+38f;
        uint32_t nearest = 0xFFFFFFFFu;
        if (dist < min_dist) { min_dist = dist; nearest = lo; }
        if (db < min_dist) { min_dist = db; nearest = hi; }
        if (nearest != 0xFFFFFFFFu) changed = commit_row(s, nearest, assignments, assignments_prev);
      }
      const unsigned long long cm = __ballot(changed);
      if (lane == 0 && cm) atomicAdd(&counters[0], (uint32_t)__popcll(cm));
    }
    return;
  }
  // ---------------- full scans: one block per row ----------------
  constexpr int CH = 4, GF = 8, NB = 4;
  float *sh_dist = settle_lds;
  uint32_t *sh_idx = reinterpret_cast<uint32_t *>(settle_lds + 4);
  const uint32_t total = *nflagged;
  for (uint32_t ri = blockIdx.x - kSettlePairBlocks; ri < total; ri += gridDim.x - kSettlePairBlocks) {
    const uint32_t s = flagged[ri];
    const float *x = samples + (size_t)s * D;  // block-uniform address: scalar loads
    const bool insane = (x[0] != x[0]);
    float min_dist = 3.402823466e+38f;
    uint32_t nearest = 0xFFFFFFFFu;
    if (!insane) {
      for (uint32_t cbase = wave * 64 * CH; cbase < K; cbase += 4 * 64 * CH) {
        float acc[CH], corr[CH];
#pragma unroll
        for (int j = 0; j < CH; j++) { acc[j] = 0.f; corr[j] = 0.f; }
        const float *cp[CH];   // a column beyond the panel reads column `lane`; its result is dropped below
#pragma unroll
        for (int j = 0; j < CH; j++) cp[j] = ct + ((cbase + 64 * j + lane < Kt) ? cbase + 64 * j + lane : (uint32_t)lane);
        float xg[NB][GF], cg[NB][GF][CH];
        const uint32_t ngroups = D / GF;   // D % 4 == 0: a 4-feature tail is handled below
        auto load_group = [&](uint32_t g, float (&xf)[GF], float (&cv)[GF][CH]) {
#pragma unroll
          for (int q = 0; q < GF; q++) {
            xf[q] = x[g * GF + q];
#pragma unroll
            for (int j = 0; j < CH; j++) cv[q][j] = cp[j][(size_t)(g * GF + q) * Kt];
          }
        };
        auto fold_group = [&](const float (&xf)[GF], float (&cv)[GF][CH]) {
#pragma unroll
          for (int q = 0; q < GF; q++) {
            float y[CH];
            fma_rd4(xf[q], cv[q], corr, y);
#pragma unroll
            for (int j = 0; j < CH; j++) kahan_fold(y[j], acc[j], corr[j]);
          }
        };
#pragma unroll
        for (int b = 0; b < NB - 1; b++)
          if ((uint32_t)b < ngroups) load_group(b, xg[b], cg[b]);
        for (uint32_t g0 = 0; g0 < ngroups; g0 += NB) {
#pragma unroll
          for (int b = 0; b < NB; b++) {
            const uint32_t g = g0 + b;
            if (g < ngroups) {
              if (g + NB - 1 < ngroups) load_group(g + NB - 1, xg[(b + NB - 1) % NB], cg[(b + NB - 1) % NB]);
              fold_group(xg[b], cg[b]);
            }
          }
        }
        for (uint32_t f = ngroups * GF; f < D; f++) {
          const float xf = x[f];
          float cv[CH], y[CH];
#pragma unroll
          for (int j = 0; j < CH; j++) cv[j] = cp[j][(size_t)f * Kt];
          fma_rd4(xf, cv, corr, y);
#pragma unroll
          for (int j = 0; j < CH; j++) kahan_fold(y[j], acc[j], corr[j]);
        }
#pragma unroll
        for (int j = 0; j < CH; j++) {
          const uint32_t c = cbase + 64 * j + lane;
          if (c < K) {
            const float dist = lloyd_distance<METRIC>(csqr[c], acc[j]);
            if (dist < min_dist) { min_dist = dist; nearest = c; }
          }
        }
      }
      // wave argmin, lowest index among equal minima; lanes without a candidate never win
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) {
        const float od = __shfl_xor(min_dist, off);
        const uint32_t oi = __shfl_xor(nearest, off);
        const bool take = (oi != 0xFFFFFFFFu) &&
                          (nearest == 0xFFFFFFFFu || od < min_dist || (od == min_dist && oi < nearest));
        if (take) { min_dist = od; nearest = oi; }
      }
    }
    if (lane == 0) {
      sh_dist[wave] = min_dist;
      sh_idx[wave] = nearest;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      for (int w = 1; w < 4; w++) {
        const float od = sh_dist[w];
        const uint32_t oi = sh_idx[w];
        const bool take = (oi != 0xFFFFFFFFu) &&
                          (nearest == 0xFFFFFFFFu || od < min_dist || (od == min_dist && oi < nearest));
        if (take) { min_dist = od; nearest = oi; }
      }
      if (nearest == 0xFFFFFFFFu && insane) nearest = K;  // kmeans.cu:349-356
      if (nearest != 0xFFFFFFFFu) {                       // else: "search failed", row left untouched
        if (commit_row(s, nearest, assignments, assignments_prev)) atomicAdd(&counters[0], 1u);
      }
    }
    __syncthreads();
  }
}

bool lloyd_settle_supported(const LloydArgs &a, const float *centroids) {
  return (a.D & 3u) == 0 && (((uintptr_t)a.samples | (uintptr_t)centroids) & 15u) == 0;
}

hipError_t launch_lloyd_settle(int metric, const LloydArgs &a, const float *centroids, hipStream_t st) {
  static bool attr_set[2] = {false, false};
  const void *fn = metric == 0 ? (const void *)lloyd_settle_kernel<0> : (const void *)lloyd_settle_kernel<1>;
  // (per device: the attribute belongs to the current device's copy of the kernel; setting it again is cheap)
  hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kSettleLds);
  if (e != hipSuccess) return e;
  (void)attr_set;
  const dim3 grid(kSettlePairBlocks + kSettleScanBlocks);
  if (metric == 0)
    hipLaunchKernelGGL((lloyd_settle_kernel<0>), grid, dim3(256), kSettleLds, st, a.samples, a.N, a.D, centroids, a.ct,
                       a.csqr, a.K, a.Kt, a.pairs, a.counters + 3, a.flagged, a.counters + 1, a.assignments,
                       a.assignments_prev, a.counters);
  else
    hipLaunchKernelGGL((lloyd_settle_kernel<1>), grid, dim3(256), kSettleLds, st, a.samples, a.N, a.D, centroids, a.ct,
                       a.csqr, a.K, a.Kt, a.pairs, a.counters + 3, a.flagged, a.counters + 1, a.assignments,
                       a.assignments_prev, a.counters);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------
// host-side launchers
// ---------------------------------------------------------------------------------------
// the Lloyd two-stage f16 filter also has a 512-wide instantiation (one operand set per wave)
uint32_t filter_dp_for(uint32_t D);
uint32_t lloyd_dp_for(uint32_t D) { return (D > 256 && D <= 512) ? 512u : filter_dp_for(D); }

uint32_t filter_dp_for(uint32_t D) {
  static const uint32_t sizes[] = {8, 16, 32, 64, 128, 256};
  for (uint32_t v : sizes)
    if (D <= v) return v;
  return 0;  // no MFMA filter instantiation: exact kernel handles everything
}

template <int DP>
static hipError_t launch_filter_dp(const LloydArgs &a, hipStream_t st) {
  const bool fast = (a.D == (uint32_t)DP);
  const size_t lds_bytes = (2 * 32 * (DP + 4) + 64) * sizeof(float);
  const uint32_t grid = (a.N + 127) / 128;
  if (fast) {
    hipLaunchKernelGGL((lloyd_filter_kernel<DP, true>), dim3(grid), dim3(256), lds_bytes, st, a.samples, a.N,
                       a.D, a.cfil, a.bias, a.mu, a.K_pad, a.K, a.stats, a.eps, a.tie_slack, a.assignments,
                       a.assignments_prev, a.flagged, a.pairs, a.counters);
  } else {
    hipLaunchKernelGGL((lloyd_filter_kernel<DP, false>), dim3(grid), dim3(256), lds_bytes, st, a.samples, a.N,
                       a.D, a.cfil, a.bias, a.mu, a.K_pad, a.K, a.stats, a.eps, a.tie_slack, a.assignments,
                       a.assignments_prev, a.flagged, a.pairs, a.counters);
  }
  return hipGetLastError();
}

hipError_t launch_lloyd_filter(const LloydArgs &a, hipStream_t st) {
  switch (a.DP) {
    case 8: return launch_filter_dp<8>(a, st);
    case 16: return launch_filter_dp<16>(a, st);
    case 32: return launch_filter_dp<32>(a, st);
    case 64: return launch_filter_dp<64>(a, st);
    case 128: return launch_filter_dp<128>(a, st);
    case 256: return launch_filter_dp<256>(a, st);
    default: return hipErrorInvalidValue;
  }
}

hipError_t launch_centroid_prep(int metric, const float *centroids, uint32_t K, uint32_t D, uint32_t K_pad,
                                uint32_t DP, uint32_t Kt, float *csqr, float *bias, float *bias2, float *cfil,
                                float *ct, float *mu, bool freeze_mu, uint32_t *finite, uint32_t *stats,
                                uint32_t *zero_a, uint32_t *zero_b, uint32_t *zero_c, hipStream_t st) {
  // (stats: zeroed by the caller)
  const dim3 block(64);
  if (metric == 0)
    hipLaunchKernelGGL((centroid_rows_kernel<0>), dim3((Kt + 63) / 64), block, 0, st, centroids, K, D, Kt, csqr, ct,
                       finite, stats);
  else
    hipLaunchKernelGGL((centroid_rows_kernel<1>), dim3((Kt + 63) / 64), block, 0, st, centroids, K, D, Kt, csqr, ct,
                       finite, stats);
  hipLaunchKernelGGL(centroid_mean_kernel, dim3((DP + 63) / 64), dim3(1024), 0, st, centroids, K, D, DP, finite, mu,
                     freeze_mu, zero_a, zero_b, zero_c);
  if (metric == 0)
    hipLaunchKernelGGL((centroid_panel_kernel<0>), dim3((K_pad + 15) / 16), dim3(256), 0, st, centroids, K, D, K_pad, DP,
                       finite, mu, bias, bias2, cfil, stats);
  else
    hipLaunchKernelGGL((centroid_panel_kernel<1>), dim3((K_pad + 15) / 16), dim3(256), 0, st, centroids, K, D, K_pad, DP,
                       finite, mu, bias, bias2, cfil, stats);
  return hipGetLastError();
}

hipError_t launch_centroid_rows(int metric, const float *centroids, uint32_t K, uint32_t D, uint32_t Kt, float *csqr,
                                float *ct, hipStream_t st) {
  if (metric == 0)
    hipLaunchKernelGGL((centroid_rows_kernel<0>), dim3((Kt + 63) / 64), dim3(64), 0, st, centroids, K, D, Kt, csqr, ct,
                       (uint32_t *)nullptr, (uint32_t *)nullptr);
  else
    hipLaunchKernelGGL((centroid_rows_kernel<1>), dim3((Kt + 63) / 64), dim3(64), 0, st, centroids, K, D, Kt, csqr, ct,
                       (uint32_t *)nullptr, (uint32_t *)nullptr);
  return hipGetLastError();
}

hipError_t launch_lloyd_pair(int metric, const LloydArgs &a, const float *centroids, uint32_t grid, hipStream_t st) {
  if (grid == 0) return hipSuccess;
  if (metric == 0)
    hipLaunchKernelGGL((lloyd_pair_kernel<0>), dim3(grid), dim3(128), 0, st, a.samples, a.D, centroids, a.csqr,
                       a.pairs, a.counters + 3, a.assignments, a.assignments_prev, a.counters);
  else
    hipLaunchKernelGGL((lloyd_pair_kernel<1>), dim3(grid), dim3(128), 0, st, a.samples, a.D, centroids, a.csqr,
                       a.pairs, a.counters + 3, a.assignments, a.assignments_prev, a.counters);
  return hipGetLastError();
}

hipError_t launch_lloyd_exact(int metric, const LloydArgs &a, const uint32_t *rows, const uint32_t *nrows,
                              uint32_t grid, hipStream_t st) {
  if (grid == 0) return hipSuccess;
  if (metric == 0)
    hipLaunchKernelGGL((lloyd_exact_kernel<0>), dim3(grid), dim3(256), 0, st, a.samples, a.N, a.D, a.ct, a.csqr,
                       a.K, a.Kt, rows, nrows, a.assignments, a.assignments_prev, a.counters);
  else
    hipLaunchKernelGGL((lloyd_exact_kernel<1>), dim3(grid), dim3(256), 0, st, a.samples, a.N, a.D, a.ct, a.csqr,
                       a.K, a.Kt, rows, nrows, a.assignments, a.assignments_prev, a.counters);
  return hipGetLastError();
}

hipError_t preload_lloyd_code() {   // (kernels.hpp: preload_code_objects)
  hipFuncAttributes at;
  return hipFuncGetAttributes(&at, reinterpret_cast<const void *>(&centroid_rows_kernel<0>));
}

}  // namespace kmx
