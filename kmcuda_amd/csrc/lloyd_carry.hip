// lloyd_carry.hip -- Lloyd passes that CARRY distance bounds from one iteration to the next, the Yinyang phase of
// the default schedule (reference: kmeans_cuda_yy, src/kmeans.cu:1028-1263, whose purpose -- not recomputing
// distances the triangle inequality already decides -- this serves; its own kernels, kmeans.cu:431-672, stay
// available bit for bit as KMCUDA_AMD_YY=reference).
//
// The reference refreshes its G + 1 bounds per row with >= G exact distance chains per row (kmeans_yy_init): 38
// assignment passes' worth of time on this hardware, which no run of the benchmark data sets ever earns back.  The
// two-stage assignment filter, on the other hand, ends every pass knowing each row's best and second-best coarse
// score with a rigorous error bound.  That is one upper bound u(x) >= d(x, c_a) and one lower bound
// l(x) <= min_{c != a} d(x, c) per row (Hamerly's pair) AT NO COST.  After the update the centroids have moved by
// drift(c) = ||c_new - c_old||; by the triangle inequality u + drift(a) and l - max_c drift(c) bound the same two
// distances for the new centroids, and a row with
//     (l' - u') (l' + u') > 4 E_ref        (E_ref: the rounding of the reference's own score)
// keeps its centroid in the reference's arithmetic, strictly: the pass does not look at it.  The other rows go
// through the filter as a LIST (lloyd_coarse2_kernel<..., CARRY = 2>), stage 2 and the settle kernel as ever, and
// come out with fresh bounds.  Assignments, previous assignments and the reassignment counter are therefore what a
// plain pass produces, row for row -- tests/test_gpu_carry.py runs the two side by side; the inequality itself is
// checked in float64 in tests/test_carry_bound_model.py.
//
// drift(c) comes out of the preparation kernel (centroid_prep_frozen_kernel: the centred panel it overwrites IS the
// previous pass's centroid, the mean being frozen), max drift in stats[6].
//
// Angular metric: the reference decides on PRODUCTS there (acos is monotone; its plateaus are the filter's tie slack),
// and the score s(c) = x'.c' + mu.c' (primes: centred by the frozen mean mu) moves by at most ||x'|| ||c_new - c_old||
// plus the change db(c) of its second term -- which is the filter's bias, known per centroid -- whatever the norms are:
// no angles, no unit length assumed.  One number per row then: the certified gap (v1 - e_c) - (v2 + e_c) between the
// row's centroid and the best of the others, shrunk every pass by ||x'|| (drift(a) + max drift) + max_c db(c) - db(a);
// the row is spared while it exceeds 4 E_ref + the tie slack.  (Charging ||x|| ||c_new - c_old|| instead -- the rows
// as they are -- spares nothing on rows that share a direction, the usual case for unit rows with positive entries:
// ||x'|| is a fraction of ||x|| there.)
//
// Pair certificates: on clustered data the listed rows are mostly the same ones every pass -- rows of a blob that two
// centroids share, too close to the border between them for stage 1's operand-rounding bound -- and stage 2 settles
// them between the same two contenders again and again.  It leaves them (lloyd_refine.hpp, PAIRS) the pair, an upper
// bound of BOTH distances and a lower bound l3 of every other centroid's (angular: the gap by which both scores exceed
// every other centroid's); carry_skip_kernel moves those like the other bounds and, while l3 stays above the upper
// bound by the same margin, queues (row, p1, p2) for the two-contender kernel (lloyd_settle_kernel: the reference's
// arithmetic and tie rule over the two).  The row is on neither the spared side nor the list.
#include <cstdlib>

#include "lloyd_coarse.hpp"
#include "lloyd_refine.hpp"

namespace kmx {

// One pass over (assignment, u, l): moves the bounds by the drifts, keeps the rows they still decide (whose previous
// assignment becomes the assignment, kmeans.cu:285-286 / :358-359: "assignments_prev[sample] = ass" for every row of
// a pass) and lists the others.  probe: count only (the host wants to know what a listed pass would cover; a whole
// pass follows and rewrites every bound).  The rows a listed pass spares are N - the list: the listed coarse kernel
// adds them up (one thread), not 2000 blocks on one address.
constexpr int kSkipRowsPerThread = 8, kSkipBlock = 256;
// WIDE: the rows' records are the streamed filter's (lloyd_wide.hip: float4 = ||x'||^2, residual^2, x[0], ||x||^2 per row,
// ||mu|| behind the last one); terms: the length of the fp32 sums behind a bias (padded features + 8)
template <bool WIDE>
__global__ __launch_bounds__(kSkipBlock) void carry_skip_kernel(
    uint32_t N, uint32_t K, const uint32_t *__restrict__ assignments, uint32_t *__restrict__ assignments_prev,
    CarryArgs cy, const float2 *__restrict__ xmeta, const float *__restrict__ drift, const uint32_t *__restrict__ stats,
    float tie_slack, uint32_t *__restrict__ row_list, const uint32_t *__restrict__ finite, uint32_t *__restrict__ pairs,
    uint32_t *__restrict__ counters, int probe, float terms) {
  if (counters[kStopFlag] != 0u) return;   // the run has stopped on the device: touch nothing
  float *__restrict__ ub = cy.ub, *__restrict__ lb = cy.lb, *__restrict__ l3 = cy.l3;
  const int angular = cy.angular;
  const float maxdrift = __uint_as_float(stats[6]);                       // +inf if any drift is not finite
  const float cmaxo = sqrtf(__uint_as_float(stats[2])) * 1.000001f;       // max ||c|| of THIS pass's centroids
  const float4 *__restrict__ wmeta = reinterpret_cast<const float4 *>(xmeta);
  const float mu_norm = WIDE ? wmeta[N].x : reinterpret_cast<const float *>(xmeta)[2 * (((size_t)N + 255) / 256 * 256)];
  const float u = 5.9604645e-8f;
  // angular: max(0, max_c db(c)) and the rounding of the four fp32 sums mu.c' behind two differences of them (each
  // within (DP + 2) u ||mu|| ||c'||, terms >= DP + 8; ||c'_old|| <= ||c'_new|| + drift)
  const float maxdb = __uint_as_float(stats[7]);
  const float cmaxc = sqrtf(__uint_as_float(stats[0])) * 1.000001f;
  const float eb = 4.0f * terms * u * mu_norm * (cmaxc + maxdrift);
  const uint32_t chunk = kSkipBlock * kSkipRowsPerThread;
  const uint32_t base = blockIdx.x * chunk;
  // ONE cursor atomic per block and list: same-address atomics are served one at a time by L2 (~11 ns each), and a
  // cursor advanced once per 256-row round made this kernel 0.19 ms per 4M rows where its 100 MB of traffic take 0.03
  __shared__ uint32_t wave_cnt[kSkipRowsPerThread][kSkipBlock / 64], pair_cnt[kSkipRowsPerThread][kSkipBlock / 64];
  __shared__ uint32_t blk_base, blk_pair_base;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint32_t listbits = 0, pairbits = 0;
#pragma unroll
  for (int it = 0; it < kSkipRowsPerThread; it++) {
    const uint32_t s = base + it * kSkipBlock + threadIdx.x;
    bool keep = false, pair = false, live = s < N;
    if (live) {
      const uint32_t a = assignments[s];
      // (a centroid that has turned non-finite while it still has members -- an overflow, inf features -- never wins in
      //  the reference (a NaN distance is never "<"): its rows are listed, whatever the drifts of its zeroed panel row say)
      if (a < K && finite[a] != 0u) {
        float xn2, xo;
        if (WIDE) {
          const float4 m = wmeta[s];
          xn2 = m.x;
          xo = sqrtf(m.w) * 1.0001f;                                // ||x||, measured
        } else {
          xn2 = xmeta[s].x;
          xo = (sqrtf(xn2) * 1.0001f + mu_norm) * 1.0001f;          // ||x|| <= ||x - mu|| + ||mu||
        }
        const float e_ref = u * (12.0f * xo * cmaxo + 4.0f * cmaxo * cmaxo);
        if (angular) {
          // score space: s(c) = x'.c' + mu.c' moves by at most ||x'|| ||c_new - c_old|| plus the change db(c) of its
          // second term, known per centroid: the certified gap shrinks by the first for the row's centroid and for the
          // best of the others, and by max_c db(c) - db(a)
          // The stored gap is also the room of the other centroids' products below the clamp at 1 and of the row's own
          // above the one at -1 (the reference's angular distance is flat beyond either, metric_abstraction.h:171-177:
          // ties, lowest index wins -- filter_common.hpp).  Those two move by ONE side each -- the others' scores rise
          // by at most B, the row's own falls by at most A -- so each side is charged at least 0: the sum bounds the
          // gap's loss and either room's.
          const float xn = sqrtf(xn2) * 1.0001f;
          const float sideA = fmaxf(xn * drift[a] - drift[K + a] + 0.5f * eb, 0.f);
          const float sideB = fmaxf(xn * maxdrift + maxdb + 0.5f * eb, 0.f);
          const float g = ub[s] - (sideA + sideB) * 1.000001f;
          keep = g > 4.1f * e_ref + 2.0f * tie_slack;   // (-inf, NaN: false)
          if (keep && !probe) {
            ub[s] = g * 0.999999f;
            if (assignments_prev[s] != a) assignments_prev[s] = a;
          } else if (!keep && l3) {
            // the pair certificate in score space: l3 = the gap by which both p1's and p2's scores exceed every other
            // centroid's; it shrinks by ||x'|| (the larger of the two drifts + max drift) + max_c db(c) - the smaller db
            const float l3s = l3[s];
            if (l3s > 0.f) {
              const uint32_t q1 = cy.p1[s], q2 = cy.p2[s];
              if ((a == q1 || a == q2) && q1 < K && q2 < K && q1 != q2 && finite[q1] != 0u && finite[q2] != 0u) {
                const float sideP = fmaxf(xn * fmaxf(drift[q1], drift[q2]) - fminf(drift[K + q1], drift[K + q2]) + 0.5f * eb, 0.f);
                const float g2 = l3s - (sideP + sideB) * 1.000001f;
                pair = g2 > 4.1f * e_ref + 2.0f * tie_slack;
                if (pair && !probe) l3[s] = g2 * 0.999999f;
              }
            }
          }
        } else {
          // (rounded away from the certificate: the sums up, the differences down)
          const float ubs = ub[s];
          const float un = (ubs + drift[a]) * 1.0000005f, ln = (lb[s] - maxdrift) * 0.9999995f;
          keep = (ln > un) && ((ln - un) * (ln + un) > 4.1f * e_ref + 2.0f * tie_slack);   // a NaN anywhere: false
          if (keep && !probe) {
            ub[s] = un;
            lb[s] = ln;
            if (assignments_prev[s] != a) assignments_prev[s] = a;
          } else if (!keep && l3) {
            // the pair certificate (stage 2 left it: ub bounds the distances to BOTH p1 and p2, l3 every other
            // centroid's; lb is void for such a row, so the test above has failed): while l3 - max drift stays above
            // ub + the larger of the two drifts by the same margin, the reference's nearest is p1 or p2, and which
            // of the two the pair kernel finds out in the reference's arithmetic
            const float l3s = l3[s];
            if (l3s > 0.f) {
              const uint32_t q1 = cy.p1[s], q2 = cy.p2[s];
              if ((a == q1 || a == q2) && q1 < K && q2 < K && q1 != q2 && finite[q1] != 0u && finite[q2] != 0u) {
                const float up = (ubs + fmaxf(drift[q1], drift[q2])) * 1.0000005f, lp = (l3s - maxdrift) * 0.9999995f;
                pair = (lp > up) && ((lp - up) * (lp + up) > 4.1f * e_ref + 2.0f * tie_slack);
                if (pair && !probe) {
                  ub[s] = up;
                  l3[s] = lp;
                }
              }
            }
          }
        }
      }
    }
    const bool list = live && !keep && !pair;
    const unsigned long long lm = __ballot(list), pm = __ballot(pair);
    if (list) listbits |= 1u << it;
    if (pair) pairbits |= 1u << it;
    if (lane == 0) {
      wave_cnt[it][wave] = (uint32_t)__popcll(lm);
      pair_cnt[it][wave] = (uint32_t)__popcll(pm);
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {   // the counts become offsets, round by round, wave by wave: the list keeps the rows' order
    uint32_t t = 0, tp = 0;
#pragma unroll
    for (int it = 0; it < kSkipRowsPerThread; it++)
#pragma unroll
      for (int w = 0; w < kSkipBlock / 64; w++) {
        const uint32_t c = wave_cnt[it][w], cp = pair_cnt[it][w];
        wave_cnt[it][w] = t;
        pair_cnt[it][w] = tp;
        t += c;
        tp += cp;
      }
    // (probe: the would-be list is only counted; the whole pass that follows looks at every row and rewrites every
    // bound and pair)
    blk_base = t ? atomicAdd(&counters[kCarryCursor], t) : 0u;
    blk_pair_base = (tp && !probe) ? atomicAdd(&counters[3], tp) : 0u;
  }
  if (probe) return;
  __syncthreads();
#pragma unroll
  for (int it = 0; it < kSkipRowsPerThread; it++) {
    const bool list = (listbits >> it) & 1u, pair = (pairbits >> it) & 1u;
    const unsigned long long lm = __ballot(list), pm = __ballot(pair);
    const unsigned long long below = (1ull << lane) - 1ull;
    const uint32_t s = base + it * kSkipBlock + threadIdx.x;
    if (list) row_list[blk_base + wave_cnt[it][wave] + (uint32_t)__popcll(lm & below)] = s;
    if (pair) {
      const size_t at = blk_pair_base + pair_cnt[it][wave] + (uint32_t)__popcll(pm & below);
      pairs[3 * at + 0] = s;
      pairs[3 * at + 1] = cy.p1[s];
      pairs[3 * at + 2] = cy.p2[s];
    }
  }
}

hipError_t launch_carry_skip(uint32_t N, uint32_t K, const uint32_t *assignments, uint32_t *assignments_prev,
                             const CarryArgs &cy, const float *xmeta, const float *drift, const uint32_t *stats,
                             float tie_slack, uint32_t *row_list, const uint32_t *finite, uint32_t *pairs,
                             uint32_t *counters, bool probe, hipStream_t st, uint32_t wide_dg) {
  const uint32_t chunk = kSkipBlock * kSkipRowsPerThread;
  if (wide_dg)
    hipLaunchKernelGGL(carry_skip_kernel<true>, dim3((N + chunk - 1) / chunk), dim3(kSkipBlock), 0, st, N, K, assignments,
                       assignments_prev, cy, reinterpret_cast<const float2 *>(xmeta), drift, stats, tie_slack, row_list,
                       finite, pairs, counters, probe ? 1 : 0, (float)(wide_dg + 8u));
  else
    hipLaunchKernelGGL(carry_skip_kernel<false>, dim3((N + chunk - 1) / chunk), dim3(kSkipBlock), 0, st, N, K, assignments,
                       assignments_prev, cy, reinterpret_cast<const float2 *>(xmeta), drift, stats, tie_slack, row_list,
                       finite, pairs, counters, probe ? 1 : 0, 520.0f);
  return hipGetLastError();
}

template <int DP>
static hipError_t launch_coarse_carry_dp(const LloydArgs &a, const void *rows, bool half_rows, const void *xcache,
                                         const float *xmeta, const void *panelhi, uint32_t *undecided, float *und_thr,
                                         const CarryArgs &cy, uint32_t rows_hint, uint32_t *duo, hipStream_t st) {
  constexpr int NSET = DP <= 256 ? 2 : 1;
  const size_t lds_bytes = 2 * 64 * (size_t)(DP * 2) + 512 + 64 + (size_t)DP * 4;
  const uint32_t rows_per_block = 128u * NSET;
  uint32_t grid = (a.N + rows_per_block - 1) / rows_per_block;
  const bool fast = a.D == (uint32_t)DP;
#define KMX_CARRY_LAUNCH(H, F, C, MODE, SRC)                                                                          \
  hipLaunchKernelGGL((lloyd_coarse2_kernel<DP, H, F, C, NSET, MODE>), dim3(grid), dim3(256), lds_bytes, st, SRC,      \
                     xmeta, a.N, a.D, reinterpret_cast<const float *>(panelhi), a.bias, a.mu, a.K_pad, a.K, a.stats,   \
                     a.eps, a.tie_slack, a.assignments, a.assignments_prev, undecided, und_thr, a.counters, cy, duo)
  if (!cy.row_list) {   // every row, streamed from the row cache
    KMX_CARRY_LAUNCH(false, true, true, 1, xcache);
    return hipGetLastError();
  }
  // the listed rows: blocks past the (device-side) end of the list leave at once; the grid follows the host's
  // estimate of the list so that a short list does not dispatch N / 256 of them
  // (the kernel strides: a list longer than the estimate costs time, never rows.  KMCUDA_AMD_CARRY_GRID caps the
  // grid -- the tests' way of making every block take several trips)
  if (rows_hint != 0xFFFFFFFFu) {
    const uint32_t want = rows_hint / rows_per_block + rows_hint / (4 * rows_per_block) + 64;
    if (want < grid) grid = want;
  }
  if (const char *v = getenv("KMCUDA_AMD_CARRY_GRID")) {
    const long cap = atol(v);
    if (cap > 0 && (uint32_t)cap < grid) grid = (uint32_t)cap;
  }
  if (half_rows) {
    if (fast) KMX_CARRY_LAUNCH(true, true, false, 2, rows); else KMX_CARRY_LAUNCH(true, false, false, 2, rows);
  } else {
    if (fast) KMX_CARRY_LAUNCH(false, true, false, 2, rows); else KMX_CARRY_LAUNCH(false, false, false, 2, rows);
  }
#undef KMX_CARRY_LAUNCH
  return hipGetLastError();
}

// cy.row_list == nullptr: a whole pass from the row cache (xcache) that leaves bounds; else the listed rows
hipError_t launch_lloyd_coarse_carry(const LloydArgs &a, const void *rows, bool half_rows, const void *xcache,
                                     const float *xmeta, const void *panelhi, uint32_t *undecided, float *und_thr,
                                     const CarryArgs &cy, uint32_t rows_hint, uint32_t *duo, hipStream_t st) {
  switch (a.DP) {
#define KMX_CARRY_CASE(dp) \
    case dp: return launch_coarse_carry_dp<dp>(a, rows, half_rows, xcache, xmeta, panelhi, undecided, und_thr, cy, rows_hint, duo, st)
    KMX_CARRY_CASE(16); KMX_CARRY_CASE(32); KMX_CARRY_CASE(64); KMX_CARRY_CASE(128); KMX_CARRY_CASE(256); KMX_CARRY_CASE(512);
#undef KMX_CARRY_CASE
    default: return hipErrorInvalidValue;
  }
}

// stage 2 of a carried L2 pass: as launch_lloyd_refine, and the rows leave with bounds / pair certificates (cy)
hipError_t launch_lloyd_refine_carry(const LloydArgs &a, const void *rows, bool half_rows, const void *panelhi,
                                     const uint32_t *row_list, const float *thr_list, const uint32_t *n_list,
                                     uint32_t rows_hint, const CarryArgs &cy, hipStream_t st) {
  return launch_lloyd_refine_t<true>(a, rows, half_rows, panelhi, row_list, thr_list, n_list, rows_hint, cy, st);
}

hipError_t preload_lloyd_carry_code() {   // (kernels.hpp: preload_code_objects)
  hipFuncAttributes at;
  return hipFuncGetAttributes(&at, reinterpret_cast<const void *>(&carry_skip_kernel<false>));
}

}  // namespace kmx
