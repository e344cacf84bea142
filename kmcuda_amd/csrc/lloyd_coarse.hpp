// lloyd_coarse.hpp -- stage 1 of the default Lloyd assignment filter (lloyd_f16.hip has the story; reference:
// src/kmeans.cu:293-364), as a template shared by two translation units: lloyd_f16.hip instantiates the plain
// pass, lloyd_carry.hip the passes that carry per-row distance bounds from one iteration to the next.
#pragma once
#include <hip/hip_fp16.h>

#include "exact.hpp"
#include "filter_common.hpp"
#include "kernels.hpp"

// KMX_ABL: timing-only ablations for scripts/coarse_harness.hip (WRONG results; never defined in the library build;
// what each part of the kernel costs on one box: DESIGN.md 4.5, profiles/r6c_*).
//   1 no LDS-DMA behind super-tile 0   2 rows from 64 blocks' worth of the cache (L2 resident)   4 no bookkeeping
//   8 no decision epilogue   16 no barrier per super-tile
#ifndef KMX_ABL
#define KMX_ABL 0
#endif

namespace kmx {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// ---------------------------------------------------------------------------------------
// Stage 1: ONE f16 MFMA per 16 features (hi.hi only).  What shaped the kernel (DESIGN.md 4.5,
// profiles/r1e..r1k): a wave owns 64 rows (two B-operand sets) so each A fragment read from LDS feeds
// two MFMAs; blocks are 4 waves, one per SIMD, two blocks per CU, so one block's bookkeeping runs under
// the other's MFMAs; the accumulator register number travels in the low 4 mantissa bits of the score
// (<= 16 ulp, part of the bound), the tile index is noted once per tile; tiles arrive by LDS-DMA
// (global_load_lds_dwordx4: no staging registers, no ds_write issue slots) with the bank swizzle
// (16-byte chunk j of row r sits in slot j ^ (r & 15) of its half row) applied to the SOURCE address
// and again by the fragment reads.
// ---------------------------------------------------------------------------------------
template <int P>
struct TileParity { static constexpr int value = P; };   // (which of a lane's two trackers a tile's scores go into)
// hand-issued LDS fragment read + counted wait (see lloyd_coarse2_kernel)
__device__ __forceinline__ f16x8 lds_frag_issue(uint32_t addr) {
  f16x8 f;
  asm volatile("ds_read_b128 %0, %1" : "=v"(f) : "v"(addr) : "memory");
  return f;
}
template <int N>
__device__ __forceinline__ void lds_frag_wait(f16x8 &f) {
  asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(f) : "n"(N));
}

// 4 waves x 64 rows per block, 2 independent blocks per CU (2 x 67 KB of LDS).  (Tried and dropped:
// one 8-wave block per CU run as a ping-pong -- the waves sharing a SIMD, read from HW_ID, alternate
// MFMA and bookkeeping phases between workgroup barriers; 65 barriers per block made it 20 % slower.)
// CACHED: the B operands come from the engine's row cache (row_cache_kernel below) instead of the rows.
// NSET: 32-row operand sets per wave -- 2 up to 256 features; 1 for 512 (the halves of 64 rows x 512
// features would be the whole register file), with each A fragment feeding one MFMA again.
// CARRY (lloyd_carry.hip; 0 = the plain pass, nothing below exists in it): 1 = every row, and the pass leaves per-row
// distance bounds behind -- an upper bound of the distance to the row's centroid, a lower bound of the distance to
// every other finite centroid, both read off the best and second-best coarse scores it has anyway; 2 = the same over
// the rows of cy.row_list only (the rows whose bounds, moved by the centroids' drifts, no longer certify their
// assignment: carry_skip_kernel), gathered from the rows like stage 2 does, with the row cache's measured norms.
template <int DP, bool HALF_ROWS, bool FAST, bool CACHED, int NSET, int CARRY = 0>
__global__ __launch_bounds__(256, 2) void lloyd_coarse2_kernel(
    const void *__restrict__ rows, const float *__restrict__ xmeta, uint32_t N, uint32_t D, const float *__restrict__ panelhi,
    const float *__restrict__ bias, const float *__restrict__ mu, uint32_t K_pad, uint32_t K,
    const uint32_t *__restrict__ stats, float eps, float tie_slack, uint32_t *__restrict__ assignments,
    uint32_t *__restrict__ assignments_prev, uint32_t *__restrict__ undecided, float *__restrict__ und_thr,
    uint32_t *__restrict__ counters, CarryArgs cy, uint32_t *__restrict__ duo = nullptr) {
  static_assert(CARRY == 0 || CARRY == 1 || CARRY == 2, "CARRY");
  static_assert(CARRY != 2 || !CACHED, "listed rows are gathered from the rows, not streamed from the row cache");
  constexpr int NKH = DP / 2;
  constexpr int KS = NKH / 8;               // k-steps = 16-byte chunks per half row
  constexpr int ROWB = DP * 2;              // bytes of one LDS row (DP hi halves)
  constexpr int SUPB = 64 * ROWB;           // one super-tile: 64 centroids
  constexpr int NP = SUPB / 1024;           // 1-KB LDS-DMA pieces per super-tile
  constexpr int SWM = (KS < 16 ? KS : 16) - 1;
  typedef __attribute__((address_space(3))) unsigned char lds_byte;
  extern __shared__ __attribute__((aligned(1024))) unsigned char lds2[];
  if (counters[kStopFlag] != 0u) return;   // the run has stopped on the device (apply_delta_kernel): touch nothing
  // raw LDS byte addresses (the fragment address is built with XOR: needs the 1-KB aligned base)
  const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_byte *)lds2;
  if (lds0 & 1023u) __builtin_trap();
  const uint32_t bias0 = lds0 + 2 * SUPB;   // 2 x 64 floats
  constexpr uint32_t BIASB = 512u;

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), col = lane & 31, h = lane >> 5;
  constexpr int WV = 4;
  constexpr bool TWO = NSET == 2;
  uint32_t total = N;
  if constexpr (CARRY == 2) total = __builtin_amdgcn_readfirstlane(*cy.n_list);
  if constexpr (CARRY != 0) {
    if (blockIdx.x == 0 && threadIdx.x == 0) {
      if (cy.host_report) {   // how long this pass's list was: the host sizes later passes by it
        volatile uint32_t *hr = cy.host_report;
        hr[0] = cy.n_list ? *cy.n_list : 0xFFFFFFFFu;   // (a pass without bounds to move has no list to report)
        hr[1] = cy.seq;
      }
      if constexpr (CARRY == 2) {   // the rows this pass does not look at (statistics: kmamd_carry_stats)
        // counters[3]: the pairs carry_skip_kernel has queued (stage 2 has not added its own yet)
        const uint32_t paired = cy.l3 ? counters[3] : 0u;
        *reinterpret_cast<unsigned long long *>(counters + kCarrySkipped) += (unsigned long long)(N - total - paired);
        *reinterpret_cast<unsigned long long *>(counters + kCarryPaired) += (unsigned long long)paired;
      }
    }
  }
  if constexpr (CARRY == 2) {
    if (blockIdx.x * (128u * NSET) >= total) return;   // (block-uniform, in front of every barrier and DMA)
  }
  // The listed pass's grid follows the host's ESTIMATE of the list (an earlier pass's length): the blocks stride over
  // the list, whose length only the device knows -- a cluster that dies, say, "drifts" by its whole centred norm (its
  // zeroed panel row against the old one: the maximum drift every row is charged; a drift that is not finite counts as
  // +inf) and can list every row in one pass, its own rows always (carry_skip_kernel: finite[a]).
  // Every other instantiation makes one trip (its block index is its 128 NSET rows).
  for (uint32_t blk = blockIdx.x;;) {
  blk = __builtin_amdgcn_readfirstlane(blk);
  const uint32_t posA = blk * (128u * NSET) + wave * (32u * NSET) + col, posB = posA + 32u;
  const bool liveA = posA < total, liveB = TWO && posB < total;
  uint32_t sA = posA, sB = posB;
  if constexpr (CARRY == 2) {
    sA = liveA ? cy.row_list[posA] : 0u;
    sB = liveB ? cy.row_list[posB] : 0u;
  }

  f16x8 xa[KS], xb[TWO ? KS : 1];
  float xn2a = 0.f, x0a = 0.f, xn2b = 0.f, x0b = 0.f;
  // x.mu and sum |x_f mu_f| per row: score + x.mu = the product, which the angular metric clamps (filter_common.hpp)
  float xdma = 0.f, xdmb = 0.f, xaba = 0.f, xabb = 0.f;
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
  // Rows arrive in at most two batches of 8 k-steps, every load of a batch issued before the first
  // use: issued k-step by k-step the prologue is 8 dependent HBM round trips (~20 us of an 80-us
  // block), and the other block's LDS-DMA pieces queue behind those misses in the in-order texture path.
  // The mean comes from LDS (staged by DMA with super-tile 0): an ordinary global load would drain vmcnt.
  constexpr int BJ = KS > 8 ? 8 : KS;
  const uint32_t mu_lds = bias0 + BIASB + 64;
  auto load_rows = [&]() {
    if constexpr (CACHED) {
      // rows = the row cache: per 32-row block KS pieces of 64 lanes x 16 bytes, already centred halves
      // in operand order -> 2 KS fully coalesced 1-KB loads per wave straight into the operand
      // registers, no conversion; the norms wait in xmeta until the decision
      const f16x8 *c = reinterpret_cast<const f16x8 *>(rows) + ((size_t)((KMX_ABL & 2) ? (blockIdx.x & 63u) : blockIdx.x) * (4 * NSET) + wave * NSET) * (KS * 64) + lane;
#pragma unroll
      for (int j = 0; j < KS; j++) xa[j] = c[j * 64];
#pragma unroll
      for (int j = 0; j < (TWO ? KS : 0); j++) xb[j] = c[(KS + j) * 64];
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      return;
    }
#pragma unroll
    for (int j0 = 0; j0 < KS; j0 += BJ) {
      float va[BJ][8], vb[TWO ? BJ : 1][8];
#pragma unroll
      for (int jj = 0; jj < BJ; jj++) load_chunk(sA, liveA, j0 + jj, va[jj]);
#pragma unroll
      for (int jj = 0; jj < (TWO ? BJ : 0); jj++) load_chunk(sB, liveB, j0 + jj, vb[jj]);
      __builtin_amdgcn_sched_barrier(0);
      if (j0 == 0) {  // the mean (and super-tile 0) landed, visible to every wave
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
      }
#pragma unroll
      for (int jj = 0; jj < BJ; jj++) {
        const int j = j0 + jj;
        float mm[8];
        const f32x4 m0 = *reinterpret_cast<const __attribute__((address_space(3))) f32x4 *>((uintptr_t)(mu_lds + (h * NKH + 8 * j) * 4));
        const f32x4 m1 = *reinterpret_cast<const __attribute__((address_space(3))) f32x4 *>((uintptr_t)(mu_lds + (h * NKH + 8 * j + 4) * 4));
        mm[0] = m0.x; mm[1] = m0.y; mm[2] = m0.z; mm[3] = m0.w;
        mm[4] = m1.x; mm[5] = m1.y; mm[6] = m1.z; mm[7] = m1.w;
        centre(va[jj], mm, xa[j], xn2a);
        if constexpr (TWO) centre(vb[jj], mm, xb[j], xn2b);
        if (j == 0) { x0a = va[0][0]; if constexpr (TWO) x0b = vb[0][0]; }
        asm volatile("" : "+v"(xa[j]));  // convert NOW: hipcc parks the fp32 values in scratch otherwise
        if constexpr (TWO) asm volatile("" : "+v"(xb[j]));
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    xn2a += __shfl_xor(xn2a, 32); x0a = __shfl(x0a, col);
    xn2b += __shfl_xor(xn2b, 32); x0b = __shfl(x0b, col);
  };

  // ---- LDS-DMA staging of super-tile sp into buffer buf ----
  const uint32_t nsuper = (K_pad + 63) / 64;
  const float *biashi = reinterpret_cast<const float *>(reinterpret_cast<const unsigned char *>(panelhi) + (size_t)nsuper * SUPB);
  // linear byte P of the super-tile image lands in LDS at P; it is fetched from source byte
  // P ^ (((P / ROWB) & SWM) << 4): the 16-byte chunk index XORed with the row's low bits (inside a
  // half row, SWM < KS).  Recomputed per piece from one opaque register -- as loop invariants the
  // per-piece addresses cost 30 VGPRs the MFMA loop needs.
  auto stage_piece = [&](uint32_t sp, int buf, int p) {
    uint32_t P0 = (uint32_t)lane * 16u;
    asm volatile("" : "+v"(P0));
    const unsigned char *src = reinterpret_cast<const unsigned char *>(panelhi) + (size_t)sp * SUPB;
    const uint32_t P = (uint32_t)p * 1024u + P0;
    const uint32_t from = P ^ (((P / ROWB) & SWM) << 4);
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + from),
                                     (__attribute__((address_space(3))) void *)(uintptr_t)(lds0 + buf * SUPB + p * 1024), 16, 0, 0);
  };
  // the 64 biases of the super-tile (clamped copy behind the panel): one 4-byte DMA.  No ordinary
  // global load lives in the loop: hipcc waits vmcnt(0) at its first use, draining the DMA
  const uint32_t mybias = bias0;
  auto stage_bias = [&](uint32_t sp, int buf) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(biashi + sp * 64u + lane),
                                     (__attribute__((address_space(3))) void *)(uintptr_t)(mybias + buf * 256), 4, 0, 0);
  };
  auto stage_issue = [&](uint32_t sp, int buf, int nw, int me) {   // nw waves share the pieces, I am number me
    for (int p = me; p < NP; p += nw) stage_piece(sp, buf, p);
    if (me == 0) stage_bias(sp, buf);
  };

  {  // the mean -> LDS: DP floats = DP / 4 sixteen-byte lanes
    constexpr int MUP = (DP * 4 + 1023) / 1024;   // 1-KB pieces
    if (wave < MUP && lane * 16 < DP * 4 - wave * 1024)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(reinterpret_cast<const unsigned char *>(mu) + wave * 1024 + lane * 16),
                                       (__attribute__((address_space(3))) void *)(uintptr_t)(mu_lds + wave * 1024), 16, 0, 0);
  }
  stage_issue(0, 0, WV, wave);
  load_rows();   // waits for the DMA above and closes with a barrier after its first batch

  // FOUR top-2 trackers per row, not two: a lane keeps one for the even tiles and one for the odd tiles (the two
  // halves of the wave see different centroids of a tile anyway).  Same work per score -- a tile's scores go into
  // its parity's tracker -- and at the end the best of every quarter is known WITH its index: when the row's
  // contenders sit in different quarters (3 cases of 4), stage 2 need not sweep the centroids again to find them.
  float v1a[2] = {-INFINITY, -INFINITY}, v2a[2] = {-INFINITY, -INFINITY}, v1b[2] = {-INFINITY, -INFINITY}, v2b[2] = {-INFINITY, -INFINITY};
  uint32_t tba[2] = {0, 0}, tbb[2] = {0, 0};
  // fragment address of k-step j: rowbase ^ swizzle ^ (16 j); (row, half) part fixed per lane
  const uint32_t fragbase = lds0 + (uint32_t)col * ROWB + (uint32_t)h * (KS * 16) + (uint32_t)((col & SWM) * 16);
  // max(v1, pk) as med3(v1, pk, +inf): fmaxf() costs a canonicalising v_max per operand on top
  float pinf = INFINITY;
  asm volatile("" : "+s"(pinf));
  auto pack = [&](float v, int r) { return __uint_as_float((__float_as_uint(v) & 0xFFFFFFF0u) | (uint32_t)r); };
  // two scores at once: the new second = max(second, median(best, a, b)), the new best = max3(best, a, b):
  // 3 ops for the pair + 2 packs.  v_max3 only sees PACKED values (results of VALU ops the compiler
  // scheduled itself), never an MFMA result: the MFMA -> VALU read hazard stays the compiler's business
  auto book2 = [&](float a, float b, int r, float &v1, float &v2) {
    const float pa = pack(a, r), pb = pack(b, r + 1);
    const float m = __builtin_amdgcn_fmed3f(v1, pa, pb);
    v2 = __builtin_amdgcn_fmed3f(v2, m, pinf);
    float t;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(t) : "v"(v1), "v"(pa), "v"(pb));
    v1 = t;
  };
  auto lds_f4 = [](uint32_t addr) {
    return *reinterpret_cast<const __attribute__((address_space(3))) f32x4 *>((uintptr_t)addr);
  };
  auto load_bias = [&](uint32_t biasaddr, f32x16 &b) {
#pragma unroll
    for (int g = 0; g < 4; g++) {
      const f32x4 b4 = lds_f4(biasaddr + (8 * g + 4 * h) * 4);
      b[4 * g + 0] = b4.x; b[4 * g + 1] = b4.y; b[4 * g + 2] = b4.z; b[4 * g + 3] = b4.w;
    }
  };
  // One tile: 2 x KS MFMAs (each A fragment feeds both row sets), then the top-2 bookkeeping of its
  // 2 x 16 scores on the VALU.  The two waves a SIMD holds belong to DIFFERENT blocks (4 waves per
  // block, one per SIMD), so they are not in step: one's bookkeeping runs under the other's MFMAs.
  // (Double-buffered accumulators with the bookkeeping interleaved in-wave need ~230 registers: the
  // B operands spill, measured slower.)
  auto tile_pass = [&](uint32_t ldsbase, uint32_t biasaddr, uint32_t t, bool stage, uint32_t sp_next, int buf_next, auto parity) {
    constexpr int PAR = decltype(parity)::value;
    f32x16 accA, accB;
    load_bias(biasaddr, accA);
    accB = accA;
    // (the 1-KB aligned tile base adds into bits the XOR never touches.)  Opaque on purpose: left
    // visible, the KS addresses are hoisted out of the tile loop and the B operands spill instead
    uint32_t fb = fragbase + ldsbase;
    asm volatile("" : "+v"(fb));
    // Fragment reads are issued by hand, PD k-steps ahead, with counted waits: while an LDS-DMA is in
    // flight hipcc turns every wait on a fragment into lgkmcnt(0), i.e. it waits for the read it
    // has just issued.  (LDS returns in order: lgkmcnt(n) = all but the youngest n reads landed.)
    constexpr int PD = KS <= 3 ? KS - 1 : 3;   // (every other depth spills the B operands at D = 256)
    f16x8 fr[PD + 1];
#pragma unroll
    for (int j = 0; j < PD; j++) fr[j] = lds_frag_issue(fb ^ (uint32_t)(j * 16));
#pragma unroll
    for (int j = 0; j < KS; j++) {
      if (j + PD < KS) fr[(j + PD) % (PD + 1)] = lds_frag_issue(fb ^ (uint32_t)((j + PD) * 16));
      constexpr int kMaxBehind = PD;
      const int behind = (KS - 1 - j) < kMaxBehind ? (KS - 1 - j) : kMaxBehind;  // younger reads in flight
      f16x8 &f = fr[j % (PD + 1)];
      if (behind == 7) lds_frag_wait<7>(f);
      else if (behind == 6) lds_frag_wait<6>(f);
      else if (behind == 5) lds_frag_wait<5>(f);
      else if (behind == 4) lds_frag_wait<4>(f);
      else if (behind == 3) lds_frag_wait<3>(f);
      else if (behind == 2) lds_frag_wait<2>(f);
      else if (behind == 1) lds_frag_wait<1>(f);
      else lds_frag_wait<0>(f);
      accA = __builtin_amdgcn_mfma_f32_32x32x16_f16(f, xa[j], accA, 0, 0, 0);
      if constexpr (TWO) accB = __builtin_amdgcn_mfma_f32_32x32x16_f16(f, xb[j], accB, 0, 0, 0);
      // the next super-tile's LDS-DMA pieces, one at a time in the shadow of the MFMAs: issued
      // back to back the four waves' 32 pieces queue up in the texture path and hold up the wave
      // (all of them during the super-tile's FIRST tile: the second one's 32 MFMAs cover the flight.  Issued from
      // the bookkeeping phase instead -- the other block's MFMAs would cover the issue -- the kernel is 2 % slower:
      // 3.51 against 3.44 ms on the same box, profiles/r3d_*)
      constexpr int SPREAD = KS >= 8 ? KS / 8 : 1;            // a piece every SPREAD k-steps
      if (!(KMX_ABL & 1) && stage && (j % SPREAD) == SPREAD / 2 && j / SPREAD < 8) {
        const int slot = j / SPREAD;                           // 0..7
        for (int p = slot * 4 + wave; p < NP; p += 32) stage_piece(sp_next, buf_next, p);
        if (slot == 0 && wave == 0) stage_bias(sp_next, buf_next);
      }
    }
    const float v1a_in = v1a[PAR], v1b_in = v1b[PAR];
    if (KMX_ABL & 4) {
      v1a[PAR] = fmaxf(v1a[PAR], accA[0]); v2a[PAR] = fmaxf(v2a[PAR], accA[15]);
      if constexpr (TWO) { v1b[PAR] = fmaxf(v1b[PAR], accB[0]); v2b[PAR] = fmaxf(v2b[PAR], accB[15]); }
    } else {
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        book2(accA[r], accA[r + 1], r, v1a[PAR], v2a[PAR]);
        if constexpr (TWO) book2(accB[r], accB[r + 1], r, v1b[PAR], v2b[PAR]);
      }
    }
    tba[PAR] = (v1a[PAR] != v1a_in) ? t : tba[PAR];
    tbb[PAR] = (v1b[PAR] != v1b_in) ? t : tbb[PAR];
  };

  for (uint32_t sp = 0; sp < nsuper; sp++) {
    const int buf = sp & 1;
    const bool stage = sp + 1 < nsuper;
    const uint32_t base = buf * SUPB, bb = mybias + buf * 256;
    tile_pass(base, bb, 2 * sp, stage, sp + 1, buf ^ 1, TileParity<0>());
    tile_pass(base + 32 * ROWB, bb + 128, 2 * sp + 1, false, sp + 1, buf ^ 1, TileParity<1>());
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (!(KMX_ABL & 16)) __syncthreads();
  }
  if (KMX_ABL & 8) {   // (keep the sweep alive, skip the epilogue)
    if (v1a[0] + v2a[0] + v1b[0] + v2b[0] + v1a[1] + v2a[1] + v1b[1] + v2b[1] == 1.2345f) assignments[0] = tba[0] + tbb[0] + tba[1] + tbb[1];
    return;
  }

  // |coarse score - reference score| <= E_c: the f32-accumulated hi.hi products (gamma_{DP+1}), the
  // dropped lo terms (|a_lo| <= 2^-11 |a|: (2^-10 + 2^-22) ||x'|| C'max), half underflow, the 4 index
  // bits packed into each score (<= 16 ulp of a score of magnitude <= ||x'|| C'max (1 + 2^-10) +
  // B'max), + E_ref.  Rows or panels with a centred norm near the half range could hold inf halves:
  // never decided here.
  const float cmaxc = sqrtf(__uint_as_float(stats[0])) * 1.000001f;
  const float bmaxc = __uint_as_float(stats[1]);
  const float cmaxo = sqrtf(__uint_as_float(stats[2])) * 1.000001f;
  const float mu_norm = (CACHED || CARRY == 2) ? xmeta[2 * (((size_t)N + 255) / 256 * 256)] : cmaxo;
  const float dcmax = sqrtf(__uint_as_float(stats[5])) * 1.000001f;   // max ||c' - hi(c')||; inf = no bound
  const float u = 5.9604645e-8f;
  uint32_t und_count = 0, changed_count = 0, duo_count = 0;
  unsigned long long uma = 0, umb = 0, dma = 0, dmb = 0;
  bool unda = false, undb = false, duoa = false, duob = false;
  uint32_t d1a = 0, d2a = 0, d1b = 0, d2b = 0;
  float dra = 0.f, drb = 0.f;
  // rows whose contenders are known by index (DUO) leave on a list of their own (lloyd_duo.hip; in a carried pass the
  // duo kernel also leaves the bounds / pair certificates stage 2 would have: the record carries the best score of
  // all the other centroids)
  const bool want_duo = duo != nullptr;
  const bool angular = tie_slack > 0.f;   // (engine.cpp: the angular metric's plateau slack; 0 under L2)
  // xdm = x.mu, xab >= sum |x_f mu_f| (< 0: not summed here -- the row cache's record: ||x|| ||mu|| bounds it)
  auto finish = [&](uint32_t s, bool live, const float (&q1)[2], const float (&q2)[2], const uint32_t (&qt)[2], float xn2,
                    float x0, float dx2, float xdm, float xab, bool &und, unsigned long long &um, float &cut, bool &is_duo,
                    unsigned long long &dm, uint32_t &dc1, uint32_t &dc2, float &drest) {
    const bool insane = (x0 != x0);  // kmeans.cu:312
    // the four quarters' bests with their centroids (mine: even / odd tiles; the other half-wave's), sorted; everything
    // else the row has seen scored at most `others`
    auto index_of = [&](float v, uint32_t tb, uint32_t hh) {
      const uint32_t r = __float_as_uint(v) & 15u;
      return tb * 32u + (r & 3u) + 8u * (r >> 2) + 4u * hh;
    };
    float k0 = q1[0], k1 = q1[1], k2 = __shfl_xor(q1[0], 32), k3 = __shfl_xor(q1[1], 32);
    uint32_t j0 = index_of(k0, qt[0], h), j1 = index_of(k1, qt[1], h);
    uint32_t j2 = __shfl_xor(j0, 32), j3 = __shfl_xor(j1, 32);
    const float mine2 = fmaxf(q2[0], q2[1]);
    const float others = fmaxf(mine2, __shfl_xor(mine2, 32));
    auto order = [](float &a, uint32_t &ia, float &b, uint32_t &ib) {   // a >= b afterwards (a stays in front on ties)
      const bool g = b > a;
      const float ta = g ? b : a, tb2 = g ? a : b;
      const uint32_t ja = g ? ib : ia, jb = g ? ia : ib;
      a = ta; b = tb2; ia = ja; ib = jb;
    };
    order(k0, j0, k1, j1); order(k2, j2, k3, j3); order(k0, j0, k2, j2); order(k1, j1, k3, j3); order(k1, j1, k2, j2);
    float v1 = k0;
    const uint32_t i1 = j0;
    const float v2 = fmaxf(k1, others);   // the row's second-best coarse score
    // ||x|| <= ||x'|| + ||mu||, and ||mu|| <= Cmax while mu is the mean of the current centroids
    // (with the row cache mu is frozen: its norm is stored behind the per-row records)
    const float xn = sqrtf(xn2) * 1.0001f, xo = (xn + mu_norm) * 1.0001f;
    // operand rounding: x'.c' - hi(x').hi(c') = x'.dc + dx.c' - dx.dc with dx = x' - hi(x'), dc likewise,
    // bounded by Cauchy-Schwarz on MEASURED residual norms (row cache / centroid_panelhi_kernel; about
    // half the worst case 2^-11 ||.||, which the uncached path uses for its rows)
    const float dx = dx2 >= 0.f ? sqrtf(dx2) * 1.0001f : 4.8829e-4f * xn;
    const float e_c = 2.0f * eps * (xn * cmaxc + bmaxc) + (xn * dcmax + dx * cmaxc + dx * dcmax) * 1.001f +
                      6e-8f * sqrtf((float)DP) * (xn + cmaxc) + 2.0e-6f * (1.001f * xn * cmaxc + bmaxc);
    const float e_ref = u * (12.0f * xo * cmaxo + 4.0f * cmaxo * cmaxo);
    const float thr = 2.0f * (e_c + e_ref) * 1.001f + tie_slack;
    const bool in_range = (xn < 6.0e4f) && (cmaxc < 6.0e4f) && (v1 > -1.0e38f) && (i1 < K);
    // angular: no centroid but the best may reach the clamp at product 1, the best not the one at -1 (filter_common.hpp)
    const ClampLimits lim = clamp_limits(angular, xdm, dot_error(DP, xab >= 0.f ? xab : xo * mu_norm), 0.5f * thr);
    const bool certain = insane || (in_range && ((v1 - v2) > thr) && (v2 < lim.hi) && (v1 > lim.lo));  // NaN anywhere => not certain
    const bool mine = (h == 0) && live;
    bool changed = false;
    if (mine && certain) changed = commit_row(s, insane ? K : i1, assignments, assignments_prev);
    und = mine && !certain;
    is_duo = false;
    {
      // undecided, a usable cut-off (below), and only the two best quarters' bests reach it: stage 2's answer -- the
      // contenders are j0 and j1 -- without its sweep.  (The scores compared are the packed ones stage 1 decides on
      // itself: e_c covers the index bits.)
      float c0 = in_range ? v1 - thr : __builtin_nanf("");
      if (angular) c0 = (lim.hi == lim.hi) ? fminf(c0, lim.hi) : __builtin_nanf("");
      is_duo = want_duo && und && (k1 >= c0) && (k2 < c0) && (others < c0) && (j1 < K);   // (NaN cut-off: no)
      if (is_duo) und = false;
      dc1 = j0; dc2 = j1; drest = fmaxf(k2, others);
    }
    if constexpr (CARRY != 0) {
      // d(x, c)^2 = ||x - mu||^2 - 2 s(c) with s(c) the exact score; |v - s(c)| <= e_c for every centroid (packed
      // index bits included), xn2 within 2 eps of ||x - mu||^2 (an fp32 sum of DP squares of rounded differences).
      // Every centroid other than i1 scored <= v2.  A row that stage 2 / the settle kernels decide may end on another
      // contender than i1: its lower bound is void (0: the next pass looks at it again); ub stays valid, the final
      // centroid being the reference's nearest (carried bounds).
      if (mine && cy.angular) {
        // the certified gap of the scores (= products up to a term constant in c): what is left of it after the
        // centroids' moves is what carry_skip_kernel tests
        // (and the room of the others' products below 1 / of the best one's above -1, which the same drifts use up:
        //  carry_skip_kernel charges the two sides' moves separately, each at least 0)
        const float e = e_c * 1.001f;
        float gapv = (certain && !insane && in_range)
                         ? fminf(fminf((v1 - e) - (v2 + e), lim.hi - v2), v1 - lim.lo) * 0.999999f : -INFINITY;
        if (!(gapv == gapv)) gapv = -INFINITY;
        cy.ub[s] = gapv;
        if (cy.l3) cy.l3[s] = 0.f;   // (no pair statement)
      } else if (mine) {
        float ubv = INFINITY, lbv = 0.f;
        if (!insane && in_range) {
          const float e = e_c * 1.001f;
          const float d2u = fmaxf(xn2 * (1.0f + 2.0f * eps) - 2.0f * (v1 - e), 0.f);
          const float d2l = xn2 * (1.0f - 2.0f * eps) - 2.0f * (v2 + e);
          const float geo = 2.4e-7f * (xn + cmaxc);
          ubv = sqrtf(d2u) * 1.000001f + geo;
          if (certain && d2l > 0.f) lbv = fmaxf(sqrtf(d2l) * 0.999999f - geo, 0.f);
          if (!(ubv == ubv)) ubv = INFINITY;
          if (!(lbv == lbv)) lbv = 0.f;
        }
        cy.ub[s] = ubv;
        cy.lb[s] = lbv;
        if (cy.l3) cy.l3[s] = 0.f;   // (no pair statement; stage 2 writes one for the rows it settles between two)
      }
    }
    // what the refine stage may drop: a centroid whose coarse score is below best - thr cannot be the
    // reference's nearest (both scores are within thr / 2 of the reference's); NaN = no such statement
    // (angular: nor one whose product may reach 1 -- every such centroid ties with the best at distance 0)
    cut = in_range ? v1 - thr : __builtin_nanf("");
    if (angular) cut = (lim.hi == lim.hi) ? fminf(cut, lim.hi) : __builtin_nanf("");
    const unsigned long long cm = __ballot(changed);
    um = __ballot(und);
    dm = __ballot(is_duo);
    changed_count += (uint32_t)__popcll(cm);
    und_count += (uint32_t)__popcll(um);
    duo_count += (uint32_t)__popcll(dm);
  };
  float dx2a = -2.f, dx2b = -2.f;   // -2: not measured (no row cache) -> the worst case 2^-11 ||x'||
  if constexpr (CACHED || CARRY == 2) {   // (the listed pass runs beside a valid row cache: its records hold for these rows)
    const float2 ma = reinterpret_cast<const float2 *>(xmeta)[sA], mb = reinterpret_cast<const float2 *>(xmeta)[TWO ? sB : sA];
    xn2a = ma.x; dx2a = ma.y; xn2b = mb.x; dx2b = mb.y;
    x0a = (ma.y == -1.f) ? __builtin_nanf("") : 0.f;
    x0b = (mb.y == -1.f) ? __builtin_nanf("") : 0.f;
    xaba = xabb = -1.f;
    if (angular) {   // x.mu: the third table of the row cache (row_cache_kernel)
      const float *xdot = xmeta + 2 * (((size_t)N + 255) / 256 * 256) + 2;
      xdma = xdot[sA];
      xdmb = xdot[TWO ? sB : sA];
    }
  } else if (angular) {   // no record: the rows once more (a loop of its own, the L2 passes keep their registers)
    const uint32_t fb = (uint32_t)(h * NKH), fe = min(fb + (uint32_t)NKH, D);
    if (fb < fe) {
      if constexpr (HALF_ROWS) {
        row_dot_mu(reinterpret_cast<const _Float16 *>(rows) + (size_t)(liveA ? sA : 0u) * D, mu, fb, fe, xdma, xaba);
        if constexpr (TWO) row_dot_mu(reinterpret_cast<const _Float16 *>(rows) + (size_t)(liveB ? sB : 0u) * D, mu, fb, fe, xdmb, xabb);
      } else {
        row_dot_mu(reinterpret_cast<const float *>(rows) + (size_t)(liveA ? sA : 0u) * D, mu, fb, fe, xdma, xaba);
        if constexpr (TWO) row_dot_mu(reinterpret_cast<const float *>(rows) + (size_t)(liveB ? sB : 0u) * D, mu, fb, fe, xdmb, xabb);
      }
    }
    xdma += __shfl_xor(xdma, 32); xaba += __shfl_xor(xaba, 32);
    xdmb += __shfl_xor(xdmb, 32); xabb += __shfl_xor(xabb, 32);
  }
  float cuta, cutb;
  finish(sA, liveA, v1a, v2a, tba, xn2a, x0a, dx2a, xdma, xaba, unda, uma, cuta, duoa, dma, d1a, d2a, dra);
  cutb = 0.f;
  if constexpr (TWO) finish(sB, liveB, v1b, v2b, tbb, xn2b, x0b, dx2b, xdmb, xabb, undb, umb, cutb, duob, dmb, d1b, d2b, drb);
  // ONE pair of global atomics per block, not three per wave: the counters share a cache line, same-address
  // atomics are served one at a time by L2 (measured round 2: 11 ns each in a kernel that did nothing else), and
  // 125 K waves per launch all arrive with theirs at the end of the same scheduling round
  __shared__ uint32_t blk_und[WV], blk_changed[WV], blk_duo[WV], blk_base, blk_duo_base;
  if (lane == 0) {
    blk_und[wave] = und_count;
    blk_changed[wave] = changed_count;
    blk_duo[wave] = duo_count;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t tu = 0, tc = 0, td = 0;
#pragma unroll
    for (int w = 0; w < WV; w++) {
      tu += blk_und[w];
      tc += blk_changed[w];
      td += blk_duo[w];
    }
    if (tc) atomicAdd(&counters[0], tc);
    blk_base = tu ? atomicAdd(&counters[4], tu) : 0u;
    blk_duo_base = td ? atomicAdd(&counters[kDuoCount], td) : 0u;
  }
  __syncthreads();
  if (duo_count) {   // (row, contender, contender, the best score of all the others)
    uint32_t base = blk_duo_base;
    for (int w = 0; w < wave; w++) base += blk_duo[w];
    const unsigned long long below = (1ull << lane) - 1ull;
    if (duoa) {
      const uint32_t at = base + (uint32_t)__popcll(dma & below);
      reinterpret_cast<uint4 *>(duo)[at] = make_uint4(sA, d1a, d2a, __float_as_uint(dra));
    }
    if (duob) {
      const uint32_t at = base + (uint32_t)__popcll(dma) + (uint32_t)__popcll(dmb & below);
      reinterpret_cast<uint4 *>(duo)[at] = make_uint4(sB, d1b, d2b, __float_as_uint(drb));
    }
  }
  if (und_count) {
    uint32_t base = blk_base;
    for (int w = 0; w < wave; w++) base += blk_und[w];
    const unsigned long long below = (1ull << lane) - 1ull;
    if (unda) {
      const uint32_t at = base + (uint32_t)__popcll(uma & below);
      undecided[at] = sA;
      und_thr[at] = cuta;
    }
    if (undb) {
      const uint32_t at = base + (uint32_t)__popcll(uma) + (uint32_t)__popcll(umb & below);
      undecided[at] = sB;
      und_thr[at] = cutb;
    }
  }
  if constexpr (CARRY != 2) {
    break;
  } else {
    blk += gridDim.x;
    if ((uint64_t)blk * (128u * NSET) >= total) break;   // (block-uniform)
    __syncthreads();   // the tiles, the mean and the block sums in LDS belong to the next trip from here on
  }
  }
}

}  // namespace kmx
