// lloyd_refine.hpp -- stage 2 of the default assignment filter (lloyd_f16.hip's launcher; lloyd_carry.hip instantiates
// it with PAIRS = true, where it also leaves the carried bounds and pair certificates of the rows it settles).
#pragma once
#include "lloyd_coarse.hpp"

namespace kmx {

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
// PAIRS: a carried pass (lloyd_carry.hip, L2) -- the rows leave with bounds, CarryArgs::l3 / p1 / p2; nothing of it
// exists in the plain instantiation.
template <int DP, bool HALF_ROWS, bool FAST, int NSET, bool PAIRS = false>
__global__ __launch_bounds__(256, 2) void lloyd_refine_kernel(
    const void *__restrict__ rows, const float *__restrict__ samples, uint32_t N, uint32_t D,
    const float *__restrict__ panelhi, const float *__restrict__ cfil, const float *__restrict__ bias,
    const float *__restrict__ mu, uint32_t K_pad, uint32_t K, const uint32_t *__restrict__ stats, float eps,
    float tie_slack, uint32_t *__restrict__ assignments, uint32_t *__restrict__ assignments_prev,
    const uint32_t *__restrict__ row_list, const float *__restrict__ thr_list, const uint32_t *__restrict__ n_list,
    uint32_t *__restrict__ flagged, uint32_t *__restrict__ pairs, uint32_t *__restrict__ counters, CarryArgs cy) {
  constexpr int NKH = DP / 2;
  constexpr int KS = NKH / 8;
  constexpr int ROWB = DP * 2;
  constexpr int SUPB = 64 * ROWB;
  constexpr int NP = SUPB / 1024;
  constexpr int SWM = (KS < 16 ? KS : 16) - 1;
  const uint32_t total = *n_list;
  // carried bounds (lloyd_carry.hip, L2): the best coarse score among the centroids that are NOT contenders, per row
  constexpr bool want_rest = PAIRS;
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
  float restA = -INFINITY, restB = -INFINITY;
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
    const float mA = max16(accA), mB = TWO ? max16(accB) : -INFINITY;
    const bool hitA = mA >= cutA, hitB = TWO && mB >= cutB;
    if constexpr (want_rest) {   // a tile without a contender: all of it belongs to the rest
      restA = hitA ? restA : fmaxf(restA, mA);
      if constexpr (TWO) restB = hitB ? restB : fmaxf(restB, mB);
    }
    if (__ballot(hitA || hitB)) {
#pragma unroll
      for (int r = 0; r < 16; r++) {
        const uint32_t c = t * 32u + (uint32_t)((r & 3) + 8 * (r >> 2)) + 4u * h;
        if (accA[r] >= cutA) append(rlA, c);
        else if (want_rest && hitA) restA = fmaxf(restA, accA[r]);
        if constexpr (TWO) {
          if (accB[r] >= cutB) append(rlB, c);
          else if (want_rest && hitB) restB = fmaxf(restB, accB[r]);
        }
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
  auto settle = [&](uint32_t s, bool live, uint32_t rl, float rest) {
    // my half of the centred fp32 row, 64 features at a time, against every contender of the row; a contender's
    // partial dot products add up in a register of its own (at most kRefineCap of them).  (128 features at a time
    // had the row chunk AND a contender's 32 sixteen-byte loads in flight: 256 registers, 650 bytes of scratch
    // per lane -- a third of a gigabyte of spill traffic per 8M-row pass.  Requesting contender i + 1's chunk and the
    // row's next chunk one step ahead -- 32-feature chunks, two buffers -- measured no better: 481 against 464 us per
    // 8M-row pass, profiles/r3j_*.)
    constexpr int FC = NKH < 64 ? NKH : 64, NCHUNK = NKH / FC;
    float xn2 = 0.f, xo2 = 0.f, x0 = 0.f;
    float xdm = 0.f, xab = 0.f;   // x.mu, sum |x_f mu_f|: the angular clamp (filter_common.hpp)
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
    if (tie_slack > 0.f) {   // angular (block-uniform): the row once more, from the caches
      const uint32_t fb = (uint32_t)(h * NKH), fe = min(fb + (uint32_t)NKH, D);
      if (fb < fe) row_dot_mu(samples + (size_t)s * D, mu, fb, fe, xdm, xab);
      xdm += __shfl_xor(xdm, 32);
      xab += __shfl_xor(xab, 32);
    }
    x0 = __shfl(x0, col);   // feature 0 lives in the lower half-wave; only its NaN-ness matters (kmeans.cu:312)
    const bool insane = (x0 != x0);
    const float xn = sqrtf(xn2) * 1.0001f, xo = sqrtf(xo2) * 1.0001f;
    const float e_mfma = 2.0f * eps * (xn * cmaxc + bmaxc);
    const float e_ref = u * (12.0f * xo * cmaxo + 4.0f * cmaxo * cmaxo);
    const float thr = 2.0f * (e_mfma + e_ref) * 1.001f + tie_slack;
    // (ranges: the contender lists came out of half operands)
    const bool in_range = usable && (xn < 6.0e4f) && (cmaxc < 6.0e4f) && i1 < K;
    // Angular: the centroids a decision rules out must stay below the clamp at product 1, its winner above the one at
    // -1 (filter_common.hpp).  The centroids that are NOT on the list scored below stage 1's cut-off, which is at most
    // stage 1's own limit: their products are below 1 already.
    const ClampLimits lim = clamp_limits(tie_slack > 0.f, xdm, dot_error(DP, xab), 0.5f * thr);
    const bool certain = insane || (in_range && ((v1 - v2) > thr) && (v2 < lim.hi) && (v1 > lim.lo));
    const bool two = !certain && in_range && ((v1 - v3) > thr) && (v3 < lim.hi) && (v1 > lim.lo) && i2 < K;
    const bool mine = (h == 0) && live;
    const bool pair_now = mine && two, flag_now = mine && !certain && !two;
    bool changed = false;
    if (mine && certain) changed = commit_row(s, insane ? K : i1, assignments, assignments_prev);
    if constexpr (want_rest) {
      // What this row carries into the next pass (lloyd_carry.hip).  d(x, c)^2 = ||x'||^2 - 2 s(c); the contenders'
      // scores v are within e_mfma of s, every other centroid's coarse score is <= rest, i.e. its s <= rest + e_c
      // (stage 1's bound with the worst-case operand rounding of the row), xn2 within 2 eps of ||x'||^2.
      //   one contender:   the row's centroid; u from v1, l from the rest (Hamerly's pair)
      //   two or more:     the row ends on i1 or i2 (here, or in the pair kernel): u = an upper bound of BOTH
      //                    distances (from v2 <= v1), l3 = a lower bound for every centroid but the two, l void
      rest = fmaxf(rest, __shfl_xor(rest, 32));
      if (mine && cy.angular) {
        // angular: the same statements in score space (lloyd_carry.hip).  One contender: the certified gap between its
        // score and every other centroid's (ub[], Hamerly's test there); two or more: l3[] = the gap by which BOTH
        // i1's and i2's scores exceed every other centroid's
        float gapv = -INFINITY, pairg = 0.f;
        if (!insane && in_range && (certain || two)) {
          const float e = e_mfma * 1.001f;
          const float dcmax = sqrtf(__uint_as_float(stats[5])) * 1.000001f, dxw = 4.8829e-4f * xn;
          const float e_c = 2.0f * eps * (xn * cmaxc + bmaxc) + (xn * dcmax + dxw * cmaxc + dxw * dcmax) * 1.001f +
                            6e-8f * sqrtf((float)DP) * (xn + cmaxc) + 2.0e-6f * (1.001f * xn * cmaxc + bmaxc);
          const float w = fmaxf(rest + e_c * 1.001f, v3 + e);
          // (and the rooms below the clamp at product 1 / above the one at -1, which the same drifts use up: the
          //  others' upper bound w against lim.hi, the winner's / the weaker contender's score against lim.lo;
          //  carry_skip_kernel charges the two sides' moves separately, each at least 0)
          const float room_hi = lim.hi - w;
          if (n == 1) gapv = fminf(fminf((v1 - e) - w, room_hi), v1 - lim.lo) * 0.999999f;
          else if (i2 < K) pairg = fminf(fminf((v2 - e) - w, room_hi), v2 - lim.lo) * 0.999999f;
          if (!(gapv == gapv)) gapv = -INFINITY;
          if (!(pairg > 0.f)) pairg = 0.f;   // (NaN too)
        }
        cy.ub[s] = gapv;
        cy.l3[s] = pairg;
        if (pairg > 0.f) {
          cy.p1[s] = i1;
          cy.p2[s] = i2;
        }
      } else if (mine) {
        float ubv = INFINITY, lbv = 0.f, l3v = 0.f;
        if (!insane && in_range && (certain || two)) {
          const float e = e_mfma * 1.001f;
          const float dcmax = sqrtf(__uint_as_float(stats[5])) * 1.000001f, dxw = 4.8829e-4f * xn;
          const float e_c = 2.0f * eps * (xn * cmaxc + bmaxc) + (xn * dcmax + dxw * cmaxc + dxw * dcmax) * 1.001f +
                            6e-8f * sqrtf((float)DP) * (xn + cmaxc) + 2.0e-6f * (1.001f * xn * cmaxc + bmaxc);
          const float geo = 2.4e-7f * (xn + cmaxc);
          const float w = fmaxf(rest + e_c * 1.001f, v3 + e);   // (v3 = -inf below three contenders)
          const float d2l = xn2 * (1.0f - 2.0f * eps) - 2.0f * w;
          float low = d2l > 0.f ? fmaxf(sqrtf(d2l) * 0.999999f - geo, 0.f) : 0.f;
          if (!(low == low)) low = 0.f;
          const bool single = n == 1;
          const float vu = single ? v1 : v2;
          ubv = sqrtf(fmaxf(xn2 * (1.0f + 2.0f * eps) - 2.0f * (vu - e), 0.f)) * 1.000001f + geo;
          if (!(ubv == ubv)) ubv = INFINITY;
          if (single) lbv = low;
          else if (i2 < K) l3v = low;
        }
        cy.ub[s] = ubv;
        cy.lb[s] = lbv;
        cy.l3[s] = l3v;
        if (l3v > 0.f) {
          cy.p1[s] = i1;
          cy.p2[s] = i2;
        }
      }
    }
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
  settle(sA, liveA, rlA, restA);
  if constexpr (TWO) settle(sB, liveB, rlB, restB);
  __syncthreads();   // the lists and tile buffers are reused by the next group
  }
}


template <int DP, int NSET, bool PAIRS>
static hipError_t launch_refine_dp_n(const LloydArgs &a, const void *rows, bool half_rows, const void *panelhi,
                                     const uint32_t *row_list, const float *thr_list, const uint32_t *n_list,
                                     uint32_t rows_hint, const CarryArgs &cy, hipStream_t st) {
  const size_t lds_bytes = 2 * 64 * (size_t)(DP * 2) + 512 + 64 + (size_t)DP * 4 + 1024 + 256 * kRefineCap * 4;
  const uint32_t rows_per_block = 128u * NSET;
  // blocks beyond the device-side list length leave at once, but dispatching 31k of them for a list
  // 2k long is not free: the grid follows the caller's estimate of the list (the kernel strides)
  uint32_t grid = (a.N + rows_per_block - 1) / rows_per_block;
  if (rows_hint != 0xFFFFFFFFu) {
    // (never less than one round of resident blocks, 2 per CU: a report that undercounts -- a pass that listed few rows
    //  followed by one that lists many -- then costs strides of a full machine, not of 64 blocks)
    uint32_t want = rows_hint / rows_per_block + rows_hint / (4 * rows_per_block) + 64;
    if (want < 512u) want = 512u;
    if (want < grid) grid = want;
  }
  const bool fast = a.D == (uint32_t)DP;
#define KMX_RFN_LAUNCH(H, F)                                                                                       \
  hipLaunchKernelGGL((lloyd_refine_kernel<DP, H, F, NSET, PAIRS>), dim3(grid), dim3(256), lds_bytes, st, rows, a.samples,  \
                     a.N, a.D, reinterpret_cast<const float *>(panelhi), a.cfil, a.bias, a.mu, a.K_pad, a.K,        \
                     a.stats, a.eps, a.tie_slack, a.assignments, a.assignments_prev, row_list, thr_list, n_list,    \
                     a.flagged, a.pairs, a.counters, cy)
  if (half_rows) {
    if (fast) KMX_RFN_LAUNCH(true, true); else KMX_RFN_LAUNCH(true, false);
  } else {
    if (fast) KMX_RFN_LAUNCH(false, true); else KMX_RFN_LAUNCH(false, false);
  }
#undef KMX_RFN_LAUNCH
  return hipGetLastError();
}

template <int DP, bool PAIRS>
static hipError_t launch_refine_dp(const LloydArgs &a, const void *rows, bool half_rows, const void *panelhi,
                                   const uint32_t *row_list, const float *thr_list, const uint32_t *n_list,
                                   uint32_t rows_hint, const CarryArgs &cy, hipStream_t st) {
  if constexpr (DP > 256) {
    return launch_refine_dp_n<DP, 1, PAIRS>(a, rows, half_rows, panelhi, row_list, thr_list, n_list, rows_hint, cy, st);
  } else if constexpr (DP >= 64) {
    // a short list in 256-row blocks is one block per CU, one wave per SIMD, and the kernel's gather /
    // contender phases are latency: 128-row blocks put two on every CU -- while they all fit in ONE round
    // of 512 resident blocks.  Beyond that (an 8-GPU shard's ~70k rows: 547 blocks, the second round
    // nearly empty, 0.10 ms) a single round of 256-row blocks is faster again
    if (rows_hint != 0xFFFFFFFFu && rows_hint <= 128u * 500u)
      return launch_refine_dp_n<DP, 1, PAIRS>(a, rows, half_rows, panelhi, row_list, thr_list, n_list, rows_hint, cy, st);
    return launch_refine_dp_n<DP, 2, PAIRS>(a, rows, half_rows, panelhi, row_list, thr_list, n_list, rows_hint, cy, st);
  } else {
    return launch_refine_dp_n<DP, 2, PAIRS>(a, rows, half_rows, panelhi, row_list, thr_list, n_list, rows_hint, cy, st);
  }
}

template <bool PAIRS>
static hipError_t launch_lloyd_refine_t(const LloydArgs &a, const void *rows, bool half_rows, const void *panelhi,
                               const uint32_t *row_list, const float *thr_list, const uint32_t *n_list,
                               uint32_t rows_hint, const CarryArgs &cy, hipStream_t st) {
  switch (a.DP) {
    case 16: return launch_refine_dp<16, PAIRS>(a, rows, half_rows, panelhi, row_list, thr_list, n_list, rows_hint, cy, st);
    case 32: return launch_refine_dp<32, PAIRS>(a, rows, half_rows, panelhi, row_list, thr_list, n_list, rows_hint, cy, st);
    case 64: return launch_refine_dp<64, PAIRS>(a, rows, half_rows, panelhi, row_list, thr_list, n_list, rows_hint, cy, st);
    case 128: return launch_refine_dp<128, PAIRS>(a, rows, half_rows, panelhi, row_list, thr_list, n_list, rows_hint, cy, st);
    case 256: return launch_refine_dp<256, PAIRS>(a, rows, half_rows, panelhi, row_list, thr_list, n_list, rows_hint, cy, st);
    case 512: return launch_refine_dp<512, PAIRS>(a, rows, half_rows, panelhi, row_list, thr_list, n_list, rows_hint, cy, st);
    default: return hipErrorInvalidValue;
  }
}


}  // namespace kmx
