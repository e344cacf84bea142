// knn_f16.hip -- the cluster-pruned k-NN search (reference: src/knn.cu:177-243) with the candidate
// filter on the f16 matrix cores.  Same contract as knn.hip's knn_filter_kernel (heap evolution,
// prune decisions, neighbour indices and order identical to the reference's); what changes:
//
//   * the corpus is additionally stored CENTRED and rounded to halves, xs16[p] = hi(x_p - mu) (half the
//     bytes of an fp32 row; mu = mean of the centroids: distances are translation invariant, centring
//     shrinks the norms in the error bound), so ||x - y||^2 ~= ||x'||^2 + ||y'||^2 - 2 x_hi.y_hi runs as
//     ONE v_mfma_f32_32x32x16_f16 per 16 features: products of halves are exact in the fp32 accumulator,
//     the operand rounding |x'.y' - hi(x').hi(y')| <= (2^-10 + 2^-22) ||x'|| ||y'|| widens the acceptance
//     band (the candidate test only has to never drop a candidate the reference would accept);
//     angular: x.y = x'.y' + mu.y' + (mu.x' + ||mu||^2): bias mu.y' per candidate, the rest per query;
//   * survivors are QUEUED per query, four deep, and a flush evaluates the exact distances as four
//     interleaved chains (exact_split.hpp; original fp32 rows from the cluster-sorted copy), then
//     replays "distance <= kth => push" in the reference's visiting order.  Evaluating a candidate the
//     reference would have rejected has no effect, and the filter threshold as of the last flush is
//     a superset of the live one (kth only decreases).
#include <hip/hip_fp16.h>
#include <stdlib.h>

#include "exact.hpp"
#include "exact_split.hpp"
#include "kernels.hpp"

namespace kmx {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

constexpr float kFltMaxK = 3.402823466e+38f;

// hand-issued LDS reads (raw LDS byte address) with counted waits: the compiler neither knows these reads nor
// drains the tile DMA in front of them
__device__ __forceinline__ f16x8 knn_frag_issue(uint32_t addr) {
  f16x8 f;
  asm volatile("ds_read_b128 %0, %1" : "=v"(f) : "v"(addr) : "memory");
  return f;
}
__device__ __forceinline__ f32x4 knn_lds_read4(uint32_t addr) {
  f32x4 f;
  asm volatile("ds_read_b128 %0, %1" : "=v"(f) : "v"(addr) : "memory");
  return f;
}
template <int N>
__device__ __forceinline__ void knn_frag_wait(f16x8 &f) {
  asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(f) : "n"(N));
}

#ifndef KMX_KNN_ABL
#define KMX_KNN_ABL 0   // timing ablations (WRONG results): 1 flushes drop their queue, 2 every tile DMA reads the cluster's first tile, 3 no mask phase
#endif
#ifdef KMX_KNN_TRACE
// s_memtime stamps of ONE block's tile iterations KMX_KNN_TRACE_FROM .. +400 (scratch/knn_trace.py):
// per wave and iteration: 0 top | 1 own DMA landed | 2 barrier passed | 3 next DMA issued | 4 chains done |
// 5 mask / queue done | 6 flushes so far | 7 cluster << 32 | tile
__device__ unsigned long long kmx_knn_trace[8 * 400 * 8];
#ifndef KMX_KNN_TRACE_FROM
#define KMX_KNN_TRACE_FROM 3000
#endif
#endif
#ifdef KMX_KNN_DBG
// instrumented build (scratch/knn_dbg.py): [0] flushes [1] queued candidates [2] cycles in flushes [3] wave cycles
// [4] wave-tiles computed [5] cycles waiting for a tile (DMA + barrier) [6] wave-tiles passed (computed or not)
// [7] live unpruned queries summed over the computed wave-tiles [8] cycles in bias + MFMA [9] cycles in mask / queue
__device__ unsigned long long kmx_knn_dbg[12];
#endif

// one wave per sorted row: xs16 (centred halves), centred squared norm, mu.x', max norm
template <int METRIC>
__global__ __launch_bounds__(256) void knn_split_kernel(const float *__restrict__ xs, uint32_t N, uint32_t D,
                                                        uint32_t DP, const float *__restrict__ mu,
                                                        _Float16 *__restrict__ xs16, float *__restrict__ n2c,
                                                        float *__restrict__ mux, float *__restrict__ kbias,
                                                        uint32_t *__restrict__ stats) {
  const uint32_t p = blockIdx.x * 4 + (threadIdx.x >> 6);
  const uint32_t lane = threadIdx.x & 63;
  if (p >= N) return;
  const float *src = xs + (size_t)p * DP;
  _Float16 *dst = xs16 + (size_t)p * DP;
  float a = 0.f, b = 0.f;
  for (uint32_t f = lane; f < DP; f += 64) {
    const float m = f < D ? mu[f] : 0.f;
    const float v = f < D ? src[f] - m : 0.f;
    dst[f] = (_Float16)v;
    a = fmaf(v, v, a);
    b = fmaf(m, v, b);
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    a += __shfl_xor(a, off);
    b += __shfl_xor(b, off);
  }
  if (lane == 0) {
    n2c[p] = a;
    mux[p] = b;
    kbias[p] = METRIC == 0 ? -0.5f * a : b;   // what a candidate adds to the matrix-core score
    if ((a - a) == 0.f) atomicMax(&stats[0], __float_as_uint(a));
  }
  (void)METRIC;
}

__device__ __forceinline__ void knn_push_sample(uint32_t k, float dist, uint32_t index, float *heap) {
  // knn.cu:133-175 (same as knn.hip's push_sample)
  uint32_t pos = 0;
  uint32_t *heapi = reinterpret_cast<uint32_t *>(heap);
  while (true) {
    float left = 0.f, right = 0.f;
    bool left_le, right_le;
    if ((2 * pos + 1) < k) { left = heap[4 * pos + 2]; left_le = dist >= left; } else left_le = true;
    if ((2 * pos + 2) < k) { right = heap[4 * pos + 4]; right_le = dist >= right; } else right_le = true;
    if (left_le && right_le) {
      heap[2 * pos] = dist;
      heapi[2 * pos + 1] = index;
      break;
    }
    bool go_right;
    if (!left_le && !right_le) go_right = left <= right;
    else go_right = left_le;
    if (go_right) {
      heap[2 * pos] = right;
      heapi[2 * pos + 1] = heapi[4 * pos + 5];
      pos = 2 * pos + 2;
    } else {
      heap[2 * pos] = left;
      heapi[2 * pos + 1] = heapi[4 * pos + 3];
      pos = 2 * pos + 1;
    }
  }
}

// hi.hi products only, ONE MFMA per 16 features: the operand rounding widens the acceptance band by
// ~0.1 % of a typical squared distance, i.e. lets through about one more candidate per query for the
// exact chain.  (Round 1 kept a three-product hi/lo variant as a second cross-check; the f32 matrix-core
// filter of knn.hip and the unfiltered exact search are the two that remain.)
// A block is KNN16_WAVES waves = KNN16_QPB queries of ONE cluster sharing every candidate tile it visits.
// Round 2 measured the 4-wave version (128 queries per tile fetch) against the counters: 3.5 TB/s of fetch
// traffic, waves parked 65 % of their cycles, the matrix pipe busy 26 % -- the kernel was bound by the
// candidate fetches, not by the MFMAs, and every compute-side change (deeper prefetch, hand-issued LDS
// reads, split accumulators) left the time unchanged.  Eight waves halve the bytes fetched per query-candidate
// pair at the same occupancy (one 8-wave block per CU instead of two 4-wave ones).
template <int DP, int METRIC, bool FASTX>
__global__ __launch_bounds__(KNN16_WAVES * 64, KNN16_BLOCKS_PER_CU) void knn_filter_f16_kernel(KnnArgs a) {
  constexpr int WV = KNN16_WAVES;
  constexpr int NKH = DP / 2;   // features per half-wave
  constexpr int KS = NKH / 8;   // k-steps = 16-byte chunks per half row
  constexpr int ROWB = DP * 2;  // bytes of one candidate row (DP halves)
  constexpr int SUB = KNN16_SUB;            // 32-candidate sub-tiles per staged tile (= per barrier)
  constexpr int TILEB = 32 * SUB * ROWB;
  constexpr int NP = (TILEB + 1023) / 1024;   // 1-KB LDS-DMA pieces per tile
  constexpr int SWM = (KS < 16 ? KS : 16) - 1;
  constexpr int NBUF = KNN16_NBUF;            // ring of tile buffers: NBUF - 1 tiles in flight
  typedef __attribute__((address_space(3))) unsigned char lds_byte;
  extern __shared__ __attribute__((aligned(1024))) unsigned char lds2[];
  const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_byte *)lds2;
  if (lds0 & 1023u) __builtin_trap();
  constexpr uint32_t TILES = (uint32_t)(NBUF * TILEB);
  const uint32_t bias0 = lds0 + TILES;                       // NBUF x 64 floats
  uint32_t *flags = reinterpret_cast<uint32_t *>(lds2 + TILES + NBUF * 256);  // 2 x WV words

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), col = lane & 31, h = lane >> 5;
  const uint32_t K = a.K, k = a.k, D = a.D;

  const uint32_t cls0 = a.blocks[2 * (size_t)blockIdx.x], p0 = a.blocks[2 * (size_t)blockIdx.x + 1];
  const uint32_t own_end = a.offsets[cls0 + 1];
  const uint32_t qp = p0 + wave * 32 + col;
  const bool live = qp < own_end;

  // B operand: my half of my query's split row
  f16x8 xhi[KS];
  {
    const _Float16 *src = reinterpret_cast<const _Float16 *>(a.xs16) + (size_t)(live ? qp : p0) * DP + h * NKH;
#pragma unroll
    for (int j = 0; j < KS; j++) {
      xhi[j] = reinterpret_cast<const f16x8 *>(src)[j];
      if (!live) {
#pragma unroll
        for (int q = 0; q < 8; q++) xhi[j][q] = (_Float16)0.f;
      }
    }
  }
  const float qn2 = live ? a.n2s[qp] : 0.f;      // centred squared norm
  const float md = live ? a.mydist[qp] : 0.f;
  const float *xrow = a.xs + (size_t)(live ? qp : p0) * DP;  // original values (exact chains)
  float *heap = a.heaps + (size_t)((live ? qp : p0) - a.p_base) * 2 * k;
  if (live && h == 0) {
    for (uint32_t i = 0; i < k; i++) {
      heap[2 * i] = kFltMaxK;
      reinterpret_cast<uint32_t *>(heap)[2 * i + 1] = 0;
    }
  }
  float mndist = kFltMaxK;

  // a candidate can only be accepted by the reference if acc >= amin
  const float nmax2 = __uint_as_float(a.stats[0]);
  const float u = 5.9604645e-8f;
  const float qn = sqrtf(qn2) * 1.0001f, nmx = sqrtf(nmax2) * 1.0001f;
  float E, kq = 0.f;
  // operand rounding of the hi.hi-only score, in the units of the respective test
  const float e_round = 9.78e-4f * qn * nmx;
  if (METRIC == 0) {
    E = 4.04f * (3.0f * a.eps + 16.0f * u) * (qn2 + nmax2) + 6e-8f * sqrtf((float)DP) * (qn + nmx) + 2.0f * e_round;
  } else {
    const float mun = sqrtf(a.mu2) * 1.0001f;
    kq = (live ? a.mux[qp] : 0.f) + a.mu2;      // x.y = acc + mu.x' + ||mu||^2
    E = 2.02f * (3.0f * a.eps + 16.0f * u) * (qn * nmx + mun * nmx) + 3e-8f * sqrtf((float)DP) * (qn + nmx) +
        a.eps * (mun * qn + a.mu2) + 1e-6f + e_round;
  }
  auto amin_of = [&](float mnd) -> float {
    if (METRIC == 0) {
      const float T2 = mnd * mnd * 1.000001f;  // inf when the heap is not full yet
      return 0.5f * (qn2 - T2 - E) - 1e-6f * (qn2 + T2);
    }
    if (mnd >= 3.1415925f) return -INFINITY;
    return cosf(mnd) - kq - E;
  };
  float amin = amin_of(mndist);

  // A tile = 32 consecutive sorted rows = TILEB contiguous bytes of xs16, copied by LDS-DMA
  // (global_load_lds_dwordx4: no staging registers, no ds_write issue slots, no address arithmetic per row).
  // Linear byte P of the tile lands in LDS at P and is fetched from source byte P ^ (((P / ROWB) & SWM) << 4):
  // the 16-byte chunk index XORed with the row's low bits (inside a half row), which makes the 16 rows that one
  // ds_read_b128 pass touches fall into 16 different bank groups.  Rows past the cluster's end (other clusters'
  // rows, or the zero padding behind the corpus) are scored like any other and dropped when queued.
  // The biases of the tile: one 4-byte DMA by wave 0.
  const int my_dma = (NP > wave ? (NP - wave + WV - 1) / WV : 0) + (wave == 0 ? 1 : 0);   // DMAs I issue per tile
  constexpr int PPW = (NP + WV - 1) / WV;   // pieces per wave and tile (waves >= NP % WV may have one less)
  // my i-th piece of the tile that starts at sorted row tile_base (i = PPW: the biases, wave 0 only)
  auto issue_piece = [&](uint32_t tile_base, int buf, int i) {
    if (i == PPW) {
      if (wave == 0)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(a.kbias + tile_base + lane),
                                         (__attribute__((address_space(3))) void *)(uintptr_t)(bias0 + buf * 256), 4, 0, 0);
      return;
    }
    const int p = wave + WV * i;
    if (p >= NP) return;   // wave-uniform
    const unsigned char *src = reinterpret_cast<const unsigned char *>(a.xs16) + (size_t)tile_base * ROWB;
    uint32_t P0 = (uint32_t)lane * 16u;
    asm volatile("" : "+v"(P0));
    const uint32_t P = (uint32_t)p * 1024u + P0;
    const uint32_t from = P ^ (((P / ROWB) & SWM) << 4);
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + from),
                                     (__attribute__((address_space(3))) void *)(uintptr_t)(lds0 + buf * TILEB + p * 1024), 16, 0, 0);
  };
  auto issue_tile = [&](uint32_t tile_base, int buf) {
#pragma unroll
    for (int i = 0; i <= PPW; i++) issue_piece(tile_base, buf, i);
  };
  // waits until at most `newer` tiles' worth of my DMAs are still in flight (they complete in order)
  // steady state: the NBUF - 2 tiles behind tile t may still be in flight; the last tiles of a cluster drain
  auto wait_tiles = [&](bool steady) {
#define KMX_VM_CASE(v) case v: asm volatile("s_waitcnt vmcnt(%0)" :: "n"((NBUF - 2) * v) : "memory"); break
    switch (steady ? my_dma : 0) {
      KMX_VM_CASE(1); KMX_VM_CASE(2); KMX_VM_CASE(3); KMX_VM_CASE(4); KMX_VM_CASE(5);
      default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
#undef KMX_VM_CASE
  };
  static_assert((NP + KNN16_WAVES - 1) / KNN16_WAVES + 1 <= 5 && (NBUF - 2) * 5 < 64, "wait_tiles: DMAs per wave and tile");
  static_assert(SUB == 1 || SUB == 2, "one 64-lane bias DMA per tile");
  static_assert(NBUF >= 2, "ring");
  const uint32_t fragbase = lds0 + (uint32_t)col * ROWB + (uint32_t)h * (KS * 16);
  const uint32_t fragswz = (uint32_t)(col & SWM) * 16u;

  // queue of survivors (sorted positions, in visiting order)
  uint32_t qc[4] = {0, 0, 0, 0};
  int qn_ = 0;
#ifdef KMX_KNN_DBG
  unsigned long long dbg[12] = {0};
  const unsigned long long dbg_t0 = __builtin_amdgcn_s_memtime();
#endif
#ifdef KMX_KNN_TRACE
  unsigned long long tr_iter = 0, tr_flushes = 0;
  const bool tr_on = blockIdx.x == KMX_KNN_TRACE;
#endif
  auto flush = [&]() {  // wave-uniform call
#ifdef KMX_KNN_TRACE
    tr_flushes++;
#endif
#ifdef KMX_KNN_DBG
    const unsigned long long f0 = __builtin_amdgcn_s_memtime();
    dbg[0]++;
    dbg[1] += (unsigned long long)__popcll(__ballot(qn_ >= 1)) + __popcll(__ballot(qn_ >= 2)) + __popcll(__ballot(qn_ >= 3)) + __popcll(__ballot(qn_ >= 4));
#endif
    if (KMX_KNN_ABL == 1) { qn_ = 0; return; }
    const float *crow[4];
#pragma unroll
    for (int i = 0; i < 4; i++) crow[i] = a.xs + (size_t)(i < qn_ ? qc[i] : 0) * DP;
    float dist[4];
    const int nq = __ballot(qn_ >= 4) ? 4 : (__ballot(qn_ >= 3) ? 3 : (__ballot(qn_ >= 2) ? 2 : 1));
    exact_distance4<NKH, METRIC, FASTX>(xrow, crow, D, h, col, dist, nq);
#pragma unroll
    for (int i = 0; i < 4; i++) {
      if (h == 0 && i < qn_ && dist[i] <= mndist) {  // knn.cu:209-212
        knn_push_sample(k, dist[i], a.inv[qc[i]], heap);
        mndist = heap[0];
      }
    }
    mndist = __shfl(mndist, col);
    amin = amin_of(mndist);
    qn_ = 0;
#ifdef KMX_KNN_DBG
    asm volatile("" :: "v"(mndist), "v"(amin));
    dbg[2] += __builtin_amdgcn_s_memtime() - f0;
#endif
  };

  unsigned long long calced = 0;
  int ph = 0;
  for (uint32_t step = 0; step <= K; step++) {
    const uint32_t cls = step == 0 ? cls0 : step - 1;
    if (step > 0 && cls == cls0) continue;
    const uint32_t beg = a.offsets[cls], end = a.offsets[cls + 1];
    bool pruned = !live;
    if (step > 0) {
      const float cd = a.C[(size_t)cls * K + cls0];
      if (cd != cd) continue;                     // knn.cu:219-221 (block-uniform)
      // the prune test needs the LIVE kth distance: settle the queue first (wave-uniform)
      if (__ballot(qn_ > 0) != 0ull) flush();
      const float lim = cd - md - a.R[cls];
      pruned = pruned || (lim > mndist);          // knn.cu:222-225
    }
    if (beg == end) continue;                     // nothing to visit (block-uniform)
    const unsigned long long visiting = __ballot(!pruned);
    const bool wave_need = visiting != 0ull;
    if (lane == 0) flags[ph * WV + wave] = wave_need ? 1u : 0u;
    __syncthreads();
    uint32_t any_need = 0;
#pragma unroll
    for (int w = 0; w < WV; w++) any_need |= flags[ph * WV + w];
    const bool need = any_need != 0u;
    ph ^= 1;
    if (!need) continue;
    calced += (unsigned long long)__popcll(visiting & 0xFFFFFFFFull) * (end - beg);  // knn.cu:228 per query

    const uint32_t ntiles = (end - beg + 32 * SUB - 1) / (32 * SUB);
    // (the barrier above ordered every wave's reads of the previous cluster's tiles before these writes)
#pragma unroll
    for (int i = 0; i < NBUF - 1; i++)
      if ((uint32_t)i < ntiles) issue_tile(beg + 32u * SUB * i, i);
    f32x16 acc;
    // scores of one tile: the biases seed the accumulator, KS matrix-core steps on hand-issued fragment reads
    // The DMAs of the tile NBUF - 1 ahead are issued BETWEEN the matrix-core steps, a piece every DSTR steps:
    // the CU's load path takes 64 bytes a cycle, i.e. 512 cycles for a 32-KB tile, and eight waves that issue
    // their pieces together right after the barrier all sit in that queue with the matrix pipe idle (traced
    // round 2: 560 of 4200 cycles per iteration).  A wave blocked on a full queue now leaves the pipe to its
    // SIMD partner, which is inside its own chain.
    constexpr int DSTR = (SUB * KS) / (PPW + 1) > 0 ? (SUB * KS) / (PPW + 1) : 1;
    auto mfma_tile = [&](int buf, int sub, bool dma, uint32_t dma_base, int dma_buf) {
#ifdef KMX_KNN_DBG
      const unsigned long long w1 = __builtin_amdgcn_s_memtime();
      dbg[4]++;
      dbg[7] += (unsigned long long)__popcll(visiting & 0xFFFFFFFFull);
#endif
      const uint32_t tb = fragbase + (uint32_t)buf * TILEB + (uint32_t)sub * (32 * ROWB);
      const uint32_t bb = bias0 + (uint32_t)buf * 256u + (uint32_t)sub * 128u + 16u * h;
      f32x4 b4[4];
#pragma unroll
      for (int g = 0; g < 4; g++) b4[g] = knn_lds_read4(bb + 32u * g);
      constexpr int PD = KS < 4 ? KS : 4;   // fragments in flight
      f16x8 fr[PD + 1];
#pragma unroll
      for (int j = 0; j < PD; j++) fr[j] = knn_frag_issue(tb + ((16u * j) ^ fragswz));
      asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(b4[0]), "+v"(b4[1]), "+v"(b4[2]), "+v"(b4[3]) : "n"(PD));
#pragma unroll
      for (int g = 0; g < 4; g++) {
        acc[4 * g + 0] = b4[g].x; acc[4 * g + 1] = b4[g].y; acc[4 * g + 2] = b4[g].z; acc[4 * g + 3] = b4[g].w;
      }
#pragma unroll
      for (int j = 0; j < KS; j++) {
        if (j + PD < KS) fr[(j + PD) % (PD + 1)] = knn_frag_issue(tb + ((16u * (j + PD)) ^ fragswz));
        const int behind = (KS - 1 - j) < PD ? (KS - 1 - j) : PD;
        f16x8 &f = fr[j % (PD + 1)];
        if (behind == 4) knn_frag_wait<4>(f);
        else if (behind == 3) knn_frag_wait<3>(f);
        else if (behind == 2) knn_frag_wait<2>(f);
        else if (behind == 1) knn_frag_wait<1>(f);
        else knn_frag_wait<0>(f);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(f, xhi[j], acc, 0, 0, 0);
        {
          const int slot = sub * KS + j;   // compile-time after unrolling
          if (dma && slot % DSTR == 0 && slot / DSTR <= PPW) issue_piece(dma_base, dma_buf, slot / DSTR);
        }
      }
#ifdef KMX_KNN_DBG
      asm volatile("" :: "v"(acc[0]), "v"(acc[15]));
      dbg[8] += __builtin_amdgcn_s_memtime() - w1;
#endif
    };
    // which of the tile's 32 candidates can still be accepted by which query: queue them, settle full queues
    auto mask_tile = [&](uint32_t tile_base) {
#ifdef KMX_KNN_DBG
      const unsigned long long w2 = __builtin_amdgcn_s_memtime();
#endif
      uint32_t m16 = 0;
      if (KMX_KNN_ABL == 3) asm volatile("" :: "v"(acc[0]), "v"(acc[5]), "v"(acc[10]), "v"(acc[15]));
      // the tile's best score first: most tiles hold no candidate for any query of the wave.  v_max3 ignores a
      // NaN operand (NaN scores never pass); amin = -inf while the heap is not full (every finite score passes)
      bool some = false;
      if (!pruned && KMX_KNN_ABL != 3) {
        // (builtins, not inline asm: the MFMA -> VALU read hazard of acc[] stays the compiler's business)
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
        if (some) {
#pragma unroll
          for (int r = 0; r < 16; r++) m16 |= (acc[r] >= amin ? 1u : 0u) << r;
        }
        const uint32_t pm = __shfl_xor(m16, 32);
        const uint32_t m0 = h ? pm : m16, m1 = h ? m16 : pm;
        uint32_t rowmask = 0;
#pragma unroll
        for (int g = 0; g < 4; g++)
          rowmask |= (((m0 >> (4 * g)) & 0xFu) << (8 * g)) | (((m1 >> (4 * g)) & 0xFu) << (8 * g + 4));
        while (__ballot(rowmask != 0u) != 0ull) {
          if (__ballot(qn_ == 4) != 0ull) flush();
          bool active = rowmask != 0u;
          const uint32_t rho = active ? (uint32_t)__ffs((int)rowmask) - 1u : 0u;
          rowmask &= rowmask - 1u;
          const uint32_t cp = tile_base + rho;
          if (step == 0 && cp == qp) active = false;  // knn.cu:204-206: not its own neighbour
          if (cp >= end) active = false;              // tile padding passes while the heap is not full
          if (active) {
#pragma unroll
            for (int i = 0; i < 4; i++)
              if (i == qn_) qc[i] = cp;
            qn_++;
          }
        }
      }
#ifdef KMX_KNN_DBG
      asm volatile("" :: "v"(qn_));
      dbg[9] += __builtin_amdgcn_s_memtime() - w2;
#endif
    };
    // (Tried round 2: the upper half of the block booking tile t - 1 after the barrier of tile t, so that the
    //  two waves of a SIMD alternate matrix-core and mask phases -- no change, 3.03 vs 3.10 s on config D: the
    //  time between barriers is set by the slowest of the eight waves, the one that found candidates.)
    for (uint32_t t = 0; t < ntiles; t++) {
      const int buf = (int)(t % NBUF);
      const uint32_t tile_base = beg + t * (32 * SUB);
#ifdef KMX_KNN_TRACE
      const bool tr = tr_on && tr_iter >= KMX_KNN_TRACE_FROM && tr_iter < KMX_KNN_TRACE_FROM + 400;
      unsigned long long *trp = kmx_knn_trace + ((size_t)wave * 400 + (tr ? tr_iter - KMX_KNN_TRACE_FROM : 0)) * 8;
      tr_iter++;
      if (tr && lane == 0) { trp[0] = __builtin_amdgcn_s_memtime(); trp[7] = ((unsigned long long)cls << 32) | t; }
#define KMX_TR(k) do { if (tr && lane == 0) trp[k] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define KMX_TR(k) do { } while (0)
#endif
      {
        // tile t has landed (my pieces: counted wait; everybody's: the barrier), and every wave is done with
        // tile t - 1, whose buffer the tile NBUF - 1 ahead goes into
        const uint32_t ahead = ntiles - 1 - t;
#ifdef KMX_KNN_DBG
        const unsigned long long w0 = __builtin_amdgcn_s_memtime();
#endif
        wait_tiles(ahead >= (uint32_t)(NBUF - 2));
        KMX_TR(1);
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        KMX_TR(2);
#ifdef KMX_KNN_DBG
        dbg[5] += __builtin_amdgcn_s_memtime() - w0;
        dbg[6]++;
#endif
      }
      const bool dma = t + (NBUF - 1) < ntiles;
      const uint32_t dma_base = KMX_KNN_ABL == 2 ? beg : tile_base + 32u * SUB * (NBUF - 1);
      const int dma_buf = (int)((t + NBUF - 1) % NBUF);
      if (dma && !wave_need) issue_tile(dma_base, dma_buf);
      KMX_TR(3);
      if (wave_need) {
#pragma unroll
        for (int sub = 0; sub < SUB; sub++) {
          // (a last tile of one sub-tile only has no tile NBUF - 1 ahead: no piece is lost by the break)
          if (sub > 0 && tile_base + 32u * sub >= end) break;   // block-uniform
          mfma_tile(buf, sub, dma, dma_base, dma_buf);
          if (sub == SUB - 1 && dma) {   // pieces the slots did not cover (very short rows)
#pragma unroll
            for (int i = (SUB * KS - 1) / DSTR + 1; i <= PPW; i++) issue_piece(dma_base, dma_buf, i);
          }
          if (sub == SUB - 1) { asm volatile("" :: "v"(acc[0]), "v"(acc[15])); KMX_TR(4); }
          mask_tile(tile_base + 32u * sub);
        }
      }
#ifdef KMX_KNN_TRACE
      asm volatile("" :: "v"(qn_));
      KMX_TR(5);
      if (tr && lane == 0) trp[6] = tr_flushes;
#endif
#undef KMX_TR
    }
  }
  if (__ballot(qn_ > 0) != 0ull) flush();
  if (live && h == 0) {  // knn.cu:239-242
    uint32_t *out = a.out + (size_t)(qp - a.p_base) * k;
    for (int i = (int)k - 1; i >= 0; i--) {
      out[i] = reinterpret_cast<uint32_t *>(heap)[1];
      knn_push_sample(k, -1.f, 0xFFFFFFFFu, heap);
    }
  }
  if (lane == 0 && calced) atomicAdd(a.calced, calced);
#ifdef KMX_KNN_DBG
  dbg[3] = __builtin_amdgcn_s_memtime() - dbg_t0;
  if (lane == 0)
    for (int i = 0; i < 10; i++) atomicAdd(&kmx_knn_dbg[i], dbg[i]);
#endif
}

#ifdef KMX_KNN_TRACE
extern "C" int kmamd_knn_trace(unsigned long long *host) {
  return hipMemcpyFromSymbol(host, HIP_SYMBOL(kmx_knn_trace), sizeof(unsigned long long) * 8 * 400 * 8) == hipSuccess ? 0 : 4;
}
#endif
#ifdef KMX_KNN_DBG
extern "C" int kmamd_knn_debug(unsigned long long *host12) {
  unsigned long long z[12] = {0};
  if (hipMemcpyFromSymbol(host12, HIP_SYMBOL(kmx_knn_dbg), sizeof(z)) != hipSuccess) return 4;
  return hipMemcpyToSymbol(HIP_SYMBOL(kmx_knn_dbg), z, sizeof(z)) == hipSuccess ? 0 : 4;
}
#endif

hipError_t launch_knn_split(int metric, const float *xs, uint32_t N, uint32_t D, uint32_t DP, const float *mu,
                            void *xs16, float *n2c, float *mux, float *kbias, uint32_t *stats, hipStream_t st) {
  hipError_t e = hipMemsetAsync(stats, 0, sizeof(uint32_t), st);
  if (e != hipSuccess) return e;
  // the tile DMA of the filter reads whole 32-row tiles: KNN16_PAD_ROWS rows / biases past the last one
  e = hipMemsetAsync(reinterpret_cast<_Float16 *>(xs16) + (size_t)N * DP, 0, (size_t)KNN16_PAD_ROWS * DP * 2, st);
  if (e != hipSuccess) return e;
  e = hipMemsetAsync(kbias + N, 0, (size_t)KNN16_PAD_ROWS * sizeof(float), st);
  if (e != hipSuccess) return e;
  if (metric == 0)
    hipLaunchKernelGGL((knn_split_kernel<0>), dim3((N + 3) / 4), dim3(256), 0, st, xs, N, D, DP, mu,
                       reinterpret_cast<_Float16 *>(xs16), n2c, mux, kbias, stats);
  else
    hipLaunchKernelGGL((knn_split_kernel<1>), dim3((N + 3) / 4), dim3(256), 0, st, xs, N, D, DP, mu,
                       reinterpret_cast<_Float16 *>(xs16), n2c, mux, kbias, stats);
  return hipGetLastError();
}

template <int DP, int METRIC>
static hipError_t launch_knn_f16_t(const KnnArgs &a, uint32_t nblocks, hipStream_t st) {
  const size_t lds_bytes = (size_t)KNN16_NBUF * (32 * KNN16_SUB * DP * 2) + KNN16_NBUF * 256 + 2 * KNN16_WAVES * 4;
  if (lds_bytes > 65536) {
    static bool raised = false;   // per instantiation
    if (!raised) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&knn_filter_f16_kernel<DP, METRIC, true>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
      if (e == hipSuccess)
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(&knn_filter_f16_kernel<DP, METRIC, false>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
      if (e != hipSuccess) return e;
      raised = true;
    }
  }
  if (a.D == (uint32_t)DP)
    hipLaunchKernelGGL((knn_filter_f16_kernel<DP, METRIC, true>), dim3(nblocks), dim3(KNN16_WAVES * 64), lds_bytes, st, a);
  else
    hipLaunchKernelGGL((knn_filter_f16_kernel<DP, METRIC, false>), dim3(nblocks), dim3(KNN16_WAVES * 64), lds_bytes, st, a);
  return hipGetLastError();
}

hipError_t launch_knn_filter_f16(int metric, const KnnArgs &a, uint32_t nblocks, hipStream_t st) {
  if (nblocks == 0) return hipSuccess;
#define KMX_KNN16_CASE(dp)                                                           \
  case dp:                                                                           \
    return metric == 0 ? launch_knn_f16_t<dp, 0>(a, nblocks, st) : launch_knn_f16_t<dp, 1>(a, nblocks, st)
  switch (a.DP) {
    KMX_KNN16_CASE(16);
    KMX_KNN16_CASE(32);
    KMX_KNN16_CASE(64);
    KMX_KNN16_CASE(128);
    KMX_KNN16_CASE(256);
    default: return hipErrorInvalidValue;
  }
#undef KMX_KNN16_CASE
}

}  // namespace kmx
