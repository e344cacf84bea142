// lloyd_duo.hip -- stage 2 of the default Lloyd filter for the rows whose contenders stage 1 already knows BY INDEX
// (reference: src/kmeans.cu:293-364, the assignment these kernels reproduce; lloyd_f16.hip has the filter's story).
//
// lloyd_coarse2_kernel keeps four top-2 trackers per row (two half-waves x even / odd tiles).  A row it cannot decide
// whose two best QUARTER bests are the only scores at or above its cut-off needs no second sweep over the centroids
// to find its contenders: it leaves on the duo list as (row, contender, contender, best score of all the others).
// This kernel does what lloyd_refine_kernel does behind its sweep -- the two contenders scored in fp32 against the
// centred row, then the same three-way decision (commit / pair kernel / full exact scan) -- with the memory access
// shaped for it: 16 lanes per row, so every load instruction of a wave covers four rows' 256 contiguous bytes
// (lloyd_refine_kernel's settle phase walks 64 rows per wave 16 bytes at a time: 64 cache lines per instruction),
// reductions inside a DPP row (no LDS), two groups of four rows in flight per wave, one atomic per block and list.
#include <hip/hip_runtime.h>

#include "filter_common.hpp"
#include "kernels.hpp"

namespace kmx {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// the sum over the 16 lanes of a DPP row, left in every one of them (the same bits: each step adds a lane's value
// to its mirror's, a + b == b + a)
__device__ __forceinline__ float row16_sum(float v) {
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, true));    // quad_perm [1,0,3,2]
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, true));    // quad_perm [2,3,0,1]
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xF, 0xF, true));   // row_half_mirror
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xF, 0xF, true));   // row_mirror
  return v;
}

constexpr uint32_t kDuoPend = 64;   // pair / full-scan records a wave holds back before it asks for list slots

// NI: 64-feature slices per row (the padded width / 64, at least 1).  FAST: D == DP and 16-byte aligned rows.
// BOUNDS: a carried pass (lloyd_carry.hip) -- the rows leave with what lloyd_refine_kernel<..., PAIRS> leaves for a row
// with two contenders: an upper bound of BOTH distances, a lower bound for every other centroid (from the record's
// best score of the others), the pair itself (angular: the gap by which both scores exceed every other centroid's).
template <int NI, bool FAST, bool BOUNDS = false>
__global__ __launch_bounds__(256, NI >= 4 ? 2 : 4) void lloyd_duo_kernel(
    const float *__restrict__ samples, uint32_t D, uint32_t DP, uint32_t K, const float *__restrict__ cfil,
    const float *__restrict__ bias, const float *__restrict__ mu, const uint32_t *__restrict__ stats, float eps,
    float tie_slack, const uint4 *__restrict__ duo, uint32_t *__restrict__ assignments,
    uint32_t *__restrict__ assignments_prev, uint32_t *__restrict__ flagged, uint32_t *__restrict__ pairs,
    uint32_t *__restrict__ counters, CarryArgs cy) {
  if (counters[kStopFlag] != 0u) return;   // (block-uniform: the run has stopped on the device)
  constexpr int SLOTS = NI >= 8 ? 1 : 2;   // groups of four rows in flight per wave
  const uint32_t total = counters[kDuoCount];
  const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4, l = lane & 15;
  const float cmaxc = sqrtf(__uint_as_float(stats[0])) * 1.000001f;
  const float bmaxc = __uint_as_float(stats[1]);
  const float cmaxo = sqrtf(__uint_as_float(stats[2])) * 1.000001f;
  const float u = 5.9604645e-8f;
  const bool angular = tie_slack > 0.f;
  // my four features of every slice of the mean (DP floats, zero beyond D)
  float mreg[NI][4];
#pragma unroll
  for (int i = 0; i < NI; i++) {
    const uint32_t f = 64u * i + 4u * l;
#pragma unroll
    for (int e = 0; e < 4; e++) mreg[i][e] = f + e < DP ? mu[f + e] : 0.f;
  }
  __shared__ uint32_t pbuf[4][kDuoPend * 3], fbuf[4][kDuoPend], blk_n[4][3], blk_base[2];
  uint32_t *mypairs = pbuf[wave], *myflags = fbuf[wave];
  uint32_t np = 0, nf = 0, nchg = 0;   // (np, nf wave-uniform; nchg per lane)
  auto flush = [&](uint32_t *list_counter, uint32_t *list, const uint32_t *buf, uint32_t words_each, uint32_t &n) {
    if (n == 0) return;
    uint32_t base = 0;
    if (lane == 0) base = atomicAdd(list_counter, n);
    base = __shfl(base, 0);
    for (uint32_t i = lane; i < n * words_each; i += 64) list[words_each * (size_t)base + i] = buf[i];
    n = 0;
  };
    const uint32_t per_iter = 4u * SLOTS;
  // (a trip's records are requested one trip ahead: record -> row is a chain of two round trips otherwise)
  uint4 rec_next[SLOTS];
  auto request = [&](uint32_t base) {
#pragma unroll
    for (int k = 0; k < SLOTS; k++) {
      const uint32_t p = base + 4u * k + g;
      rec_next[k] = p < total ? duo[p] : make_uint4(0u, 0u, 0u, 0u);
    }
  };
  request((blockIdx.x * 4u + wave) * per_iter);
  for (uint32_t base = (blockIdx.x * 4u + wave) * per_iter; base < total; base += gridDim.x * 4u * per_iter) {   // wave-uniform
    uint4 rec[SLOTS];
    bool live[SLOTS];
    f32x4 xv[SLOTS][NI], c1v[SLOTS][NI], c2v[SLOTS][NI];
    uint32_t old_asg[SLOTS];
    float bias_a[SLOTS], bias_b[SLOTS];
#pragma unroll
    for (int k = 0; k < SLOTS; k++) {
      live[k] = base + 4u * k + g < total;
      rec[k] = rec_next[k];
    }
    request(base + gridDim.x * 4u * per_iter);
    // everything the decision will need leaves with the rows: the previous assignment (commit_row's read) and the two
    // biases behind the products would each be a round trip of their own at the end of the trip
#pragma unroll
    for (int k = 0; k < SLOTS; k++) {
      old_asg[k] = assignments[rec[k].x];
      bias_a[k] = bias[rec[k].y];
      bias_b[k] = bias[rec[k].z];
    }
#pragma unroll
    for (int k = 0; k < SLOTS; k++) {
      const float *xr = samples + (size_t)rec[k].x * D;
      const float *r1 = cfil + (size_t)rec[k].y * DP, *r2 = cfil + (size_t)rec[k].z * DP;
#pragma unroll
      for (int i = 0; i < NI; i++) {
        const uint32_t f = 64u * i + 4u * l;
        f32x4 x = {0.f, 0.f, 0.f, 0.f}, a = x, b = x;
        if (f < DP) {   // (narrower rows than 64 features: the upper lanes of a group idle)
          if (FAST) {
            x = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(xr + f));   // read once: past the caches
          } else {
            x.x = f + 0 < D ? xr[f + 0] : 0.f; x.y = f + 1 < D ? xr[f + 1] : 0.f;
            x.z = f + 2 < D ? xr[f + 2] : 0.f; x.w = f + 3 < D ? xr[f + 3] : 0.f;
          }
          a = *reinterpret_cast<const f32x4 *>(r1 + f);   // the centred fp32 panel: zero beyond D
          b = *reinterpret_cast<const f32x4 *>(r2 + f);
        }
        xv[k][i] = x; c1v[k][i] = a; c2v[k][i] = b;
      }
    }
#pragma unroll
    for (int k = 0; k < SLOTS; k++) {
      float xn2 = 0.f, xo2 = 0.f, p1 = 0.f, p2 = 0.f, xdm = 0.f, xab = 0.f, nan0 = 0.f;
#pragma unroll
      for (int i = 0; i < NI; i++) {
        const float x4[4] = {xv[k][i].x, xv[k][i].y, xv[k][i].z, xv[k][i].w};
        const float a4[4] = {c1v[k][i].x, c1v[k][i].y, c1v[k][i].z, c1v[k][i].w};
        const float b4[4] = {c2v[k][i].x, c2v[k][i].y, c2v[k][i].z, c2v[k][i].w};
#pragma unroll
        for (int e = 0; e < 4; e++) {
          const float xc = x4[e] - mreg[i][e];
          xn2 = fmaf(xc, xc, xn2);
          xo2 = fmaf(x4[e], x4[e], xo2);
          p1 = fmaf(xc, a4[e], p1);
          p2 = fmaf(xc, b4[e], p2);
          if (angular) {   // x.mu and sum |x_f mu_f|: the clamp's limits (filter_common.hpp)
            xdm = fmaf(x4[e], mreg[i][e], xdm);
            xab = fmaf(fabsf(x4[e]), fabsf(mreg[i][e]), xab);
          }
        }
        if (i == 0 && l == 0 && x4[0] != x4[0]) nan0 = 1.f;   // kmeans.cu:312: a NaN first feature
      }
      xn2 = row16_sum(xn2); xo2 = row16_sum(xo2); p1 = row16_sum(p1); p2 = row16_sum(p2); nan0 = row16_sum(nan0);
      if (angular) { xdm = row16_sum(xdm); xab = row16_sum(xab); }
      // ---- the decision: lloyd_refine_kernel's, for a list of two ----
      const uint32_t s = rec[k].x, ca = rec[k].y, cb = rec[k].z;
      const float va = p1 + bias_a[k], vb = p2 + bias_b[k];
      float v1 = va, v2 = -INFINITY;
      const float v3 = -INFINITY;
      uint32_t i1 = ca, i2 = 0xFFFFFFFFu;
      if (!(va > -INFINITY)) { v1 = -INFINITY; i1 = 0xFFFFFFFFu; }   // (the list scan's strict '>' from -inf: NaN / -inf never enter)
      {
        const bool g1 = vb > v1, g2 = vb > v2;
        i2 = g1 ? i1 : (g2 ? cb : i2);
        v2 = g1 ? v1 : (g2 ? vb : v2);
        i1 = g1 ? cb : i1;
        v1 = g1 ? vb : v1;
      }
      const bool insane = nan0 > 0.f;
      const float xn = sqrtf(xn2) * 1.0001f, xo = sqrtf(xo2) * 1.0001f;
      const float e_mfma = 2.0f * eps * (xn * cmaxc + bmaxc);   // an fp32 FMA sum of D + 1 terms, any order
      const float e_ref = u * (12.0f * xo * cmaxo + 4.0f * cmaxo * cmaxo);
      const float thr = 2.0f * (e_mfma + e_ref) * 1.001f + tie_slack;
      const bool in_range = (xn < 6.0e4f) && (cmaxc < 6.0e4f) && i1 < K;
      // angular: what a decision rules out stays below the clamp at product 1, its winner above the one at -1; the
      // centroids off the list scored below stage 1's cut-off, which is at most stage 1's own limit
      const ClampLimits lim = clamp_limits(angular, xdm, dot_error((int)DP, xab), 0.5f * thr);
      const bool certain = insane || (in_range && ((v1 - v2) > thr) && (v2 < lim.hi) && (v1 > lim.lo));
      const bool two = !certain && in_range && ((v1 - v3) > thr) && (v3 < lim.hi) && (v1 > lim.lo) && i2 < K;
      const bool mine = live[k] && l == 0;
      const bool pair_now = mine && two, flag_now = mine && !certain && !two;
      if (mine && certain) {   // commit_row() with the previous assignment already here
        const uint32_t nearest = insane ? K : i1;
        assignments_prev[s] = old_asg[k];
        if (old_asg[k] != nearest) {
          assignments[s] = nearest;
          nchg++;
        }
      }
      if constexpr (BOUNDS) {   // lloyd_refine_kernel's statements for n = 2 contenders (its comments hold the derivation)
        if (mine) {
          const float rest = __uint_as_float(rec[k].w);
          const bool ok = !insane && in_range && (certain || two);
          const float e = e_mfma * 1.001f;
          const float dcmax = sqrtf(__uint_as_float(stats[5])) * 1.000001f, dxw = 4.8829e-4f * xn;
          const float e_c = 2.0f * eps * (xn * cmaxc + bmaxc) + (xn * dcmax + dxw * cmaxc + dxw * dcmax) * 1.001f +
                            6e-8f * sqrtf((float)DP) * (xn + cmaxc) + 2.0e-6f * (1.001f * xn * cmaxc + bmaxc);
          const float w = fmaxf(rest + e_c * 1.001f, v3 + e);
          if (cy.angular) {
            float pairg = 0.f;
            if (ok && i2 < K) pairg = fminf(fminf((v2 - e) - w, lim.hi - w), v2 - lim.lo) * 0.999999f;
            if (!(pairg > 0.f)) pairg = 0.f;   // (NaN too)
            cy.ub[s] = -INFINITY;              // (no single-contender gap: the row has two)
            cy.l3[s] = pairg;
            if (pairg > 0.f) { cy.p1[s] = i1; cy.p2[s] = i2; }
          } else {
            float ubv = INFINITY, l3v = 0.f;
            if (ok) {
              const float geo = 2.4e-7f * (xn + cmaxc);
              const float d2l = xn2 * (1.0f - 2.0f * eps) - 2.0f * w;
              float low = d2l > 0.f ? fmaxf(sqrtf(d2l) * 0.999999f - geo, 0.f) : 0.f;
              if (!(low == low)) low = 0.f;
              ubv = sqrtf(fmaxf(xn2 * (1.0f + 2.0f * eps) - 2.0f * (v2 - e), 0.f)) * 1.000001f + geo;
              if (!(ubv == ubv)) ubv = INFINITY;
              if (i2 < K) l3v = low;
            }
            cy.ub[s] = ubv;
            cy.lb[s] = 0.f;
            cy.l3[s] = l3v;
            if (l3v > 0.f) { cy.p1[s] = i1; cy.p2[s] = i2; }
          }
        }
      }
      const unsigned long long pm = __ballot(pair_now), fm = __ballot(flag_now);
      if (pm) {   // wave-uniform
        if (np + 4 > kDuoPend) flush(&counters[3], pairs, mypairs, 3, np);
        if (pair_now) {
          const uint32_t at = np + (uint32_t)__popcll(pm & ((1ull << lane) - 1ull));
          mypairs[3 * at + 0] = s; mypairs[3 * at + 1] = i1; mypairs[3 * at + 2] = i2;
        }
        np += (uint32_t)__popcll(pm);
      }
      if (fm) {
        if (nf + 4 > kDuoPend) flush(&counters[1], flagged, myflags, 1, nf);
        if (flag_now) myflags[nf + (uint32_t)__popcll(fm & ((1ull << lane) - 1ull))] = s;
        nf += (uint32_t)__popcll(fm);
      }
    }
  }
  // what is left: one atomic per block and list
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) nchg += __shfl_xor(nchg, off);
  if (lane == 0) { blk_n[wave][0] = np; blk_n[wave][1] = nf; blk_n[wave][2] = nchg; }
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t tp = 0, tf = 0, tc = 0;
    for (int w = 0; w < 4; w++) { tp += blk_n[w][0]; tf += blk_n[w][1]; tc += blk_n[w][2]; }
    blk_base[0] = tp ? atomicAdd(&counters[3], tp) : 0u;
    blk_base[1] = tf ? atomicAdd(&counters[1], tf) : 0u;
    if (tc) atomicAdd(&counters[0], tc);
  }
  __syncthreads();
  uint32_t bp = blk_base[0], bf = blk_base[1];
  for (uint32_t w = 0; w < wave; w++) { bp += blk_n[w][0]; bf += blk_n[w][1]; }
  for (uint32_t i = lane; i < np * 3; i += 64) pairs[3 * (size_t)bp + i] = mypairs[i];
  for (uint32_t i = lane; i < nf; i += 64) flagged[bf + i] = myflags[i];
}

template <int NI>
static hipError_t launch_duo_ni(const LloydArgs &a, const uint32_t *duo, hipStream_t st, const CarryArgs *cy) {
  const bool fast = a.D == a.DP && (((uintptr_t)a.samples) & 15u) == 0 && a.DP % 4 == 0;
  // the waves that are resident at once at this kernel's register count (2 or 4 blocks per CU of 256); every wave
  // strides over the device-side count
  const uint32_t want = (a.N + 31u) / 32u, full = NI >= 4 ? 512u : 1024u;
  const dim3 grid(want < full ? (want ? want : 1u) : full);
#define KMX_DUO_LAUNCH(F, B, CY)                                                                                      \
  hipLaunchKernelGGL((lloyd_duo_kernel<NI, F, B>), grid, dim3(256), 0, st, a.samples, a.D, a.DP, a.K, a.cfil, a.bias,    \
                     a.mu, a.stats, a.eps, a.tie_slack, reinterpret_cast<const uint4 *>(duo), a.assignments,          \
                     a.assignments_prev, a.flagged, a.pairs, a.counters, CY)
  if (cy && cy->l3) {
    if (fast) KMX_DUO_LAUNCH(true, true, *cy); else KMX_DUO_LAUNCH(false, true, *cy);
  } else {
    if (fast) KMX_DUO_LAUNCH(true, false, CarryArgs()); else KMX_DUO_LAUNCH(false, false, CarryArgs());
  }
#undef KMX_DUO_LAUNCH
  return hipGetLastError();
}

hipError_t launch_lloyd_duo(const LloydArgs &a, const uint32_t *duo, hipStream_t st, const CarryArgs *cy) {
  if (a.N == 0) return hipSuccess;
  switch (a.DP) {
    case 16: case 32: case 64: return launch_duo_ni<1>(a, duo, st, cy);
    case 128: return launch_duo_ni<2>(a, duo, st, cy);
    case 256: return launch_duo_ni<4>(a, duo, st, cy);
    case 512: return launch_duo_ni<8>(a, duo, st, cy);
    default: return hipErrorInvalidValue;
  }
}

hipError_t preload_lloyd_duo_code() {   // (kernels.hpp: preload_code_objects)
  hipFuncAttributes at;
  return hipFuncGetAttributes(&at, reinterpret_cast<const void *>(&lloyd_duo_kernel<4, true>));
}

}  // namespace kmx
