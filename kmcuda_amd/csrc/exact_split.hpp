// exact_split.hpp -- the reference's exact distance chain evaluated by a (lane, lane+32) pair of the
// MFMA kernels' waves, four candidates at a time (yinyang_mfma.hip, knn_f16.hip).
#pragma once
#include "exact.hpp"

namespace kmx {

typedef float f32x4_es __attribute__((ext_vector_type(4)));

// Up to four exact chains at once over the D features of (original sample row, four centroid rows), all
// from global memory: the lower half-wave runs features [0, NK), hands (acc, corr) to the upper
// half which continues with [NK, D).  Every lane of a (col, col+32) pair gets the results.  The
// round-down FMAs of a feature share one rounding-mode window (exact.hpp).  Rolled loops: the
// function is instantiated at several call sites and must stay small.
// metric_abstraction.h:73-86 (L2 distance_t) / :193-205 (angular).
// nq (wave-uniform, 1..4): only the first nq candidate rows are real -- the others are neither loaded nor
// meaningful (their dist[] is garbage); with nq <= 2 a TWO-chain body runs (half the arithmetic: round 2
// counters put a four-chain flush at 100 K cycles whatever its loads did -- 7000 instructions per wave, most of
// them for chains nobody had queued).
// lane_q (per lane, <= nq): this row has only lane_q real candidates -- its other loads are skipped, and all of
// them (its x values too) when it has none.
// DEEP: the gathers of eight steps are issued as one batch before their arithmetic (callers whose registers are
// otherwise idle: a wave's last flush).
template <int NK, int METRIC, bool FAST, bool DEEP, int W>
__device__ __forceinline__ void exact_distance_w(const float *__restrict__ xrow, const float *const (&crow)[4],
                                                 uint32_t D, int h, int col, float (&dist)[4], int nq, int lane_q) {
  float acc[W], corr[W];
#pragma unroll
  for (int i = 0; i < W; i++) acc[i] = corr[i] = 0.f;
  const int nvalid = (int)D - h * NK < 0 ? 0 : ((int)D - h * NK > NK ? NK : (int)D - h * NK);
  for (int pass = 0; pass < 2; pass++) {
    if (pass == 1) {
#pragma unroll
      for (int i = 0; i < W; i++) {
        acc[i] = __shfl(acc[i], col);
        corr[i] = __shfl(corr[i], col);
      }
    }
    // only the half-wave whose features this pass covers loads and computes
    if (h == pass) {
      // one step: four features of the W chains
      auto step = [&](const float (&xv)[4], const float (&cv)[W][4], int j) {
#pragma unroll
        for (int q = 0; q < 4; q++) {
          float y[W];
          if constexpr (W == 4) {
            if (METRIC == 0) {
              float d[4];
#pragma unroll
              for (int i = 0; i < 4; i++) d[i] = xv[q] - cv[i][q];
              sqfma_rd4(d, corr, y);
            } else {
              const float b[4] = {cv[0][q], cv[1][q], cv[2][q], cv[3][q]};
              fma_rd4(xv[q], b, corr, y);
            }
          } else {
            if (METRIC == 0) {
              const float d[2] = {xv[q] - cv[0][q], xv[q] - cv[1][q]};
              sqfma_rd2(d, corr, y);
            } else {
              const float b[2] = {cv[0][q], cv[1][q]};
              fma_rd2(xv[q], b, corr, y);
            }
          }
          const bool on = FAST || (j + q < nvalid);
#pragma unroll
          for (int i = 0; i < W; i++) {
            const float t = acc[i] + y[i];
            const float nc = y[i] - (t - acc[i]);
            acc[i] = on ? t : acc[i];
            corr[i] = on ? nc : corr[i];
          }
        }
      };
      if constexpr (DEEP && FAST) {
        constexpr int NB = NK / 4 < 8 ? NK / 4 : 8;
#pragma unroll 1
        for (int j0 = 0; j0 < NK; j0 += 4 * NB) {
          f32x4_es xs[NB], cs[NB][W];
#pragma unroll
          for (int b = 0; b < NB; b++) {
            xs[b] = f32x4_es{0.f, 0.f, 0.f, 0.f};
            if (lane_q > 0) xs[b] = *reinterpret_cast<const f32x4_es *>(xrow + h * NK + j0 + 4 * b);
#pragma unroll
            for (int i = 0; i < W; i++) {
              cs[b][i] = f32x4_es{0.f, 0.f, 0.f, 0.f};
              if (i < nq && i < lane_q) cs[b][i] = *reinterpret_cast<const f32x4_es *>(crow[i] + h * NK + j0 + 4 * b);
            }
          }
#pragma unroll
          for (int b = 0; b < NB; b++) {
            const float xv[4] = {xs[b].x, xs[b].y, xs[b].z, xs[b].w};
            float cv[W][4];
#pragma unroll
            for (int i = 0; i < W; i++) {
              cv[i][0] = cs[b][i].x; cv[i][1] = cs[b][i].y; cv[i][2] = cs[b][i].z; cv[i][3] = cs[b][i].w;
            }
            step(xv, cv, j0 + 4 * b);
          }
        }
      } else {
#pragma unroll 2
        for (int j = 0; j < NK; j += 4) {
          float xv[4], cv[W][4];
          if (FAST) {
            f32x4_es x4 = {0.f, 0.f, 0.f, 0.f};
            if (lane_q > 0) x4 = *reinterpret_cast<const f32x4_es *>(xrow + h * NK + j);
            xv[0] = x4.x; xv[1] = x4.y; xv[2] = x4.z; xv[3] = x4.w;
#pragma unroll
            for (int i = 0; i < W; i++) {
              f32x4_es v = {0.f, 0.f, 0.f, 0.f};
              if (i < nq && i < lane_q) v = *reinterpret_cast<const f32x4_es *>(crow[i] + h * NK + j);   // nq: scalar
              cv[i][0] = v.x; cv[i][1] = v.y; cv[i][2] = v.z; cv[i][3] = v.w;
            }
          } else {
#pragma unroll
            for (int q = 0; q < 4; q++) {
              const bool in = j + q < nvalid;
              xv[q] = in ? xrow[h * NK + j + q] : 0.f;
#pragma unroll
              for (int i = 0; i < W; i++) cv[i][q] = in ? crow[i][h * NK + j + q] : 0.f;
            }
          }
          step(xv, cv, j);
        }
      }
    }
  }
#pragma unroll
  for (int i = 0; i < W; i++) {
    const float total = __shfl(acc[i], col + 32);
    dist[i] = METRIC == 0 ? sqrtf(total) : angular_from_prod(total);
  }
}

template <int NK, int METRIC, bool FAST, bool DEEP = false>
__device__ __forceinline__ void exact_distance4(const float *__restrict__ xrow, const float *const (&crow)[4],
                                                uint32_t D, int h, int col, float (&dist)[4], int nq = 4, int lane_q = 4) {
  if (nq <= 2) exact_distance_w<NK, METRIC, FAST, DEEP, 2>(xrow, crow, D, h, col, dist, nq, lane_q);
  else exact_distance_w<NK, METRIC, FAST, DEEP, 4>(xrow, crow, D, h, col, dist, nq, lane_q);
}


// State of the same chains run as a two-stage pipeline over consecutive batches (yinyang_init.hip:
// exact_chain4_lds_pipe): in one call the lower half-wave runs features [0, NK) of the NEW batch while the
// upper half-wave finishes features [NK, D) of the PREVIOUS one, whose (acc, corr) it received at the end of
// the previous call -- every lane works in every call (exact_distance4 leaves one half idle per pass), at the
// price of results arriving one call late.  The chains themselves are unchanged: same operations, same order.
struct ExactPipe4 {
  float acc[4] = {0.f, 0.f, 0.f, 0.f}, corr[4] = {0.f, 0.f, 0.f, 0.f};  // upper half: the pending batch
};

}  // namespace kmx
