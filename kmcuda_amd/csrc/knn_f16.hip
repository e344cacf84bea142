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
#include "knn_heap.hpp"

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

// one wave per sorted row: xs16 (centred halves), centred squared norm, mu.x', max norm
template <int METRIC>
__global__ __launch_bounds__(256) void knn_split_kernel(const float *__restrict__ xs, uint32_t N, uint32_t D,
                                                        uint32_t DP, const float *__restrict__ mu,
                                                        _Float16 *__restrict__ xs16, float *__restrict__ n2c,
                                                        float *__restrict__ mux, float *__restrict__ kbias,
                                                        uint32_t *__restrict__ stats) {
  const uint32_t lane = threadIdx.x & 63;
  for (uint32_t p = blockIdx.x * 4 + (threadIdx.x >> 6); p < N; p += gridDim.x * 4) {   // (kernels.hpp: wave_row_grid)
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
      // finite, non-negative: bits order like values.  Look before the atomic: N of them on one address serialise
      // in L2 (8M rows: 90 ms for a kernel that moves 16 GB), and the running maximum rarely moves
      if ((a - a) == 0.f && __float_as_uint(a) > *reinterpret_cast<volatile uint32_t *>(&stats[0]))
        atomicMax(&stats[0], __float_as_uint(a));
    }
  }
  (void)METRIC;
}

// hi.hi products only, ONE MFMA per 16 features: the operand rounding widens the acceptance band by
// ~0.1 % of a typical squared distance, i.e. lets through about one more candidate per query for the
// exact chain.  (Round 1 kept a three-product hi/lo variant as a second cross-check; the f32 matrix-core
// filter of knn.hip and the unfiltered exact search are the two that remain.)
//
// Shape (round 2, each step measured on BASELINE config D, 8M x 256 corpus, 1M queries: 3.72 s -> see
// profiles/README.md): a block is knn16_waves(DP) waves x knn16_nset(DP) operand sets of 32 queries of ONE cluster
// sharing every candidate tile it visits.
//   * 256 queries per tile fetch (was 128): the 4-wave / one-set kernel pulled 3.5 TB/s through L2 with the
//     matrix pipe busy 26 %.
//   * Tiles arrive by LDS-DMA into a ring (no staging registers, no ds_write), bank-swizzled by source address;
//     the DMA pieces are issued BETWEEN the matrix-core steps: the CU's load path takes 64 bytes a cycle, and
//     waves that issue a 32-KB tile's pieces together right after the barrier all sit in that queue.
//   * Two 32-candidate sub-tiles per barrier; best-score-first mask test (most tiles hold no candidate).
//   * Two operand sets per wave: every candidate fragment read from LDS feeds two matrix products -- with one
//     set a ds_read_b128 per MFMA is the whole LDS bandwidth at full matrix rate -- and the two blocks a CU
//     holds are not in step, so one's mask phases and barriers run under the other's products.
// Instrumented builds of the one-set kernel (per-phase s_memtime counters, per-wave timeline of one block):
// profiles/r2e_knn_filter_phase_counters.log, r2e_knn_filter_block_timeline.log.
template <int DP, int METRIC, bool FASTX>
__global__ __launch_bounds__(knn16_waves(DP) * 64, knn16_blocks_per_cu(DP)) void knn_filter_f16_kernel(KnnArgs a) {
  constexpr int WV = knn16_waves(DP), NSET = knn16_nset(DP);
  constexpr int NKH = DP / 2;   // features per half-wave
  constexpr int KS = NKH / 8;   // k-steps = 16-byte chunks per half row
  constexpr int ROWB = DP * 2;  // bytes of one candidate row (DP halves)
  constexpr int SUB = knn16_sub(DP);        // 32-candidate sub-tiles per staged tile (= per barrier)
  constexpr int TILEB = 32 * SUB * ROWB;
  constexpr int NP = (TILEB + 1023) / 1024;   // 1-KB LDS-DMA pieces per tile
  constexpr int SWM = (KS < 16 ? KS : 16) - 1;
  constexpr int NBUF = knn16_nbuf(DP);        // ring of tile buffers: NBUF - 1 tiles in flight
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
  if (cls0 == 0xFFFFFFFFu) return;   // an empty slot of the dispatch plan (kmcuda_api.cpp: one cluster's blocks per XCD)
  const uint32_t own_end = a.offsets[cls0 + 1];
  uint32_t qp[NSET];
  bool live[NSET];
  // B operands: my half of my queries' rows (centred halves)
  f16x8 xhi[NSET][KS];
  float qn2[NSET], md[NSET], mndist[NSET], amin[NSET], E[NSET], kq[NSET];
  const float nmax2 = __uint_as_float(a.stats[0]);
  const float u = 5.9604645e-8f;
  const float nmx = sqrtf(nmax2) * 1.0001f;
  // a candidate can only be accepted by the reference if acc >= amin
  auto amin_of = [&](int e, float mnd) -> float {
    if (METRIC == 0) {
      const float T2 = mnd * mnd * 1.000001f;  // inf when the heap is not full yet
      return 0.5f * (qn2[e] - T2 - E[e]) - 1e-6f * (qn2[e] + T2);
    }
    if (mnd >= 3.1415925f) return -INFINITY;
    return cosf(mnd) - kq[e] - E[e];
  };
#pragma unroll
  for (int e = 0; e < NSET; e++) {
    qp[e] = p0 + (uint32_t)wave * (32u * NSET) + 32u * e + col;   // my slot of the block plan ...
    live[e] = qp[e] < own_end;
    if (a.qperm && live[e]) qp[e] = a.qperm[qp[e] - a.p_base];     // ... and the query of this cluster it stands for
    const uint32_t qq = live[e] ? qp[e] : p0;
    const _Float16 *src = reinterpret_cast<const _Float16 *>(a.xs16) + (size_t)qq * DP + h * NKH;
#pragma unroll
    for (int j = 0; j < KS; j++) {
      xhi[e][j] = reinterpret_cast<const f16x8 *>(src)[j];
      if (!live[e]) {
#pragma unroll
        for (int q = 0; q < 8; q++) xhi[e][j][q] = (_Float16)0.f;
      }
    }
    qn2[e] = live[e] ? a.n2s[qp[e]] : 0.f;      // centred squared norm
    md[e] = live[e] ? a.mydist[qp[e]] : 0.f;
    float *heap = a.heaps + (size_t)(qq - a.p_base) * 2 * k;
    if (live[e] && h == 0) {
      for (uint32_t i = 0; i < k; i++) {
        heap[2 * i] = kFltMaxK;
        reinterpret_cast<uint32_t *>(heap)[2 * i + 1] = 0;
      }
    }
    mndist[e] = kFltMaxK;
    const float qn = sqrtf(qn2[e]) * 1.0001f;
    // operand rounding of the hi.hi-only score, in the units of the respective test
    const float e_round = 9.78e-4f * qn * nmx;
    kq[e] = 0.f;
    if (METRIC == 0) {
      E[e] = 4.04f * (3.0f * a.eps + 16.0f * u) * (qn2[e] + nmax2) + 6e-8f * sqrtf((float)DP) * (qn + nmx) + 2.0f * e_round;
    } else {
      const float mun = sqrtf(a.mu2) * 1.0001f;
      kq[e] = (live[e] ? a.mux[qp[e]] : 0.f) + a.mu2;      // x.y = acc + mu.x' + ||mu||^2
      E[e] = 2.02f * (3.0f * a.eps + 16.0f * u) * (qn * nmx + mun * nmx) + 3e-8f * sqrtf((float)DP) * (qn + nmx) +
             a.eps * (mun * qn + a.mu2) + 1e-6f + e_round;
    }
    amin[e] = amin_of(e, mndist[e]);
  }

  // A tile = 32 SUB consecutive sorted rows = TILEB contiguous bytes of xs16, copied by LDS-DMA
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
  // steady state: the NBUF - 2 tiles behind tile t may still be in flight (my DMAs complete in order); the last
  // tiles of a cluster drain
  auto wait_tiles = [&](bool steady) {
#define KMX_VM_CASE(v) case v: asm volatile("s_waitcnt vmcnt(%0)" :: "n"((NBUF - 2) * v) : "memory"); break
    switch (steady && NBUF > 2 ? my_dma : 0) {
      KMX_VM_CASE(1); KMX_VM_CASE(2); KMX_VM_CASE(3); KMX_VM_CASE(4); KMX_VM_CASE(5);
      KMX_VM_CASE(6); KMX_VM_CASE(7); KMX_VM_CASE(8); KMX_VM_CASE(9);
      default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
#undef KMX_VM_CASE
  };
  static_assert(NBUF == 2 || (PPW + 1 <= 9 && (NBUF - 2) * 9 < 64), "wait_tiles: DMAs per wave and tile");
  static_assert(SUB == 1 || SUB == 2, "one 64-lane bias DMA per tile");
  static_assert(NBUF >= 2, "ring");
  const uint32_t fragbase = lds0 + (uint32_t)col * ROWB + (uint32_t)h * (KS * 16);
  const uint32_t fragswz = (uint32_t)(col & SWM) * 16u;

  // queues of survivors (sorted positions, in visiting order), one per operand set
  uint32_t qc[NSET][4];
  int qn_[NSET];
#pragma unroll
  for (int e = 0; e < NSET; e++) {
    qn_[e] = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) qc[e][i] = 0;
  }
  // (The heaps stay in global memory, in the reference's interleaved layout.  Round 3 moved them into LDS for
  // k <= 12 -- distances plus one-byte slots, the indices parked in global memory, so a push was LDS-only but
  // for one store -- with identical lists and NO gain: 2.72 s against 2.59 s for the config D share
  // (profiles/r3f_*).  The pushes were not what the other waves wait for at the barrier.)
  uint32_t chains = 0;   // exact chains this lane's queries have paid for (statistics)
  auto flush = [&](int e) {  // wave-uniform call
    const uint32_t qq = live[e] ? qp[e] : p0;
    const float *xrow = a.xs + (size_t)qq * DP;   // original values (exact chains)
    float *heap = a.heaps + (size_t)(qq - a.p_base) * 2 * k;
    const float *crow[4];
#pragma unroll
    for (int i = 0; i < 4; i++) crow[i] = a.xs + (size_t)(i < qn_[e] ? qc[e][i] : 0) * DP;
    float dist[4];
    const int nq = __ballot(qn_[e] >= 4) ? 4 : (__ballot(qn_[e] >= 3) ? 3 : (__ballot(qn_[e] >= 2) ? 2 : 1));
    chains += h == 0 ? (uint32_t)qn_[e] : 0u;
    exact_distance4<NKH, METRIC, FASTX>(xrow, crow, D, h, col, dist, nq, qn_[e]);
    float mnd = mndist[e];
#pragma unroll
    for (int i = 0; i < 4; i++) {
      if (h == 0 && i < qn_[e] && dist[i] <= mnd) {  // knn.cu:209-212
        knn_push_sample(k, dist[i], a.inv[qc[e][i]], heap);
        mnd = heap[0];
      }
    }
    mndist[e] = __shfl(mnd, col);
    amin[e] = amin_of(e, mndist[e]);
    qn_[e] = 0;
  };

  unsigned long long calced = 0, scored = 0, useful = 0, scored_sets = 0;   // wave-uniform (scalar registers)
  int ph = 0;
  for (uint32_t step = 0; step <= K; step++) {
    const uint32_t cls = step == 0 ? cls0 : step - 1;
    if (step > 0 && cls == cls0) continue;
    const uint32_t beg = a.offsets[cls], end = a.offsets[cls + 1];
    bool pruned[NSET];
#pragma unroll
    for (int e = 0; e < NSET; e++) pruned[e] = !live[e];
    if (step > 0) {
      const float cd = a.C[(size_t)cls * K + cls0];
      if (cd != cd) continue;                     // knn.cu:219-221 (block-uniform)
      const float rr = a.R[cls];
#pragma unroll
      for (int e = 0; e < NSET; e++) {
        // the prune test needs the LIVE kth distance: settle the queue first (wave-uniform)
        if (__ballot(qn_[e] > 0) != 0ull) flush(e);
        const float lim = cd - md[e] - rr;
        pruned[e] = pruned[e] || (lim > mndist[e]);          // knn.cu:222-225
      }
    }
    if (beg == end) continue;                     // nothing to visit (block-uniform)
    unsigned long long visiting = 0;
    uint32_t nvisit = 0;   // queries that visit the cluster BY THE REFERENCE'S RULE: what knn.cu:228 counts
#pragma unroll
    for (int e = 0; e < NSET; e++) nvisit += (uint32_t)__popcll(__ballot(!pruned[e]) & 0xFFFFFFFFull);
    calced += (unsigned long long)nvisit * (end - beg);
    // A second, tighter test of the same kind (a.lb, knn_centroid_bounds_kernel): every member x of the cluster has
    // d(q, x) >= d(q, c) - d(x, c) >= d(q, c) - R[c], with the query's OWN distance to the centroid instead of the
    // reference's bound for it, C[c][mine] - d(q, c_mine).  A cluster it rules out holds no candidate the reference
    // would accept (its "distance <= kth" fails for every member), so skipping the visit changes no heap -- only the
    // work: on k-means clusters of a Gaussian mixture the reference's rule visits 57 % of all pairs, this one 23 %.
    if (step > 0 && a.lb) {
#pragma unroll
      for (int e = 0; e < NSET; e++)
        if (!pruned[e]) pruned[e] = a.lb[(size_t)cls * a.lb_stride + (qp[e] - a.p_base)] > mndist[e];
    }
    uint32_t nvis_tight = 0, nsets_live = 0;   // queries / operand sets of the wave that visit the cluster after both tests
#pragma unroll
    for (int e = 0; e < NSET; e++) {
      const unsigned long long b = __ballot(!pruned[e]);
      visiting |= b;
      nvis_tight += (uint32_t)__popcll(b & 0xFFFFFFFFull);
      nsets_live += b ? 1u : 0u;
    }
    const bool wave_need = visiting != 0ull;
    if (lane == 0) flags[ph * WV + wave] = wave_need ? 1u : 0u;
    __syncthreads();
    uint32_t any_need = 0;
#pragma unroll
    for (int w = 0; w < WV; w++) any_need |= flags[ph * WV + w];
    const bool need = any_need != 0u;
    ph ^= 1;
    if (!need) continue;

    const uint32_t ntiles = (end - beg + 32 * SUB - 1) / (32 * SUB);
    // (the barrier above ordered every wave's reads of the previous cluster's tiles before these writes)
#pragma unroll
    for (int i = 0; i < NBUF - 1; i++)
      if ((uint32_t)i < ntiles) issue_tile(beg + 32u * SUB * i, i);
    f32x16 acc[NSET];
    // The DMAs of the tile NBUF - 1 ahead are issued BETWEEN the matrix-core steps, a piece every DSTR steps.
    // (every k-step, every second, every third instead of spread over the tile: 0.985 / 0.995 / 0.988 s against 1.000 for
    //  config D's share -- the filter does not wait for its tiles, profiles/r5ab_*)
    constexpr int DSTR = (SUB * KS) / (PPW + 1) > 0 ? (SUB * KS) / (PPW + 1) : 1;
    // scores of one sub-tile: the biases seed the accumulators, KS k-steps on hand-issued fragment reads, every
    // fragment feeding the NSET operand sets
    // (an operand set none of whose 32 queries visits the cluster is not multiplied: 3.4 % of the scored pairs of
    //  config D's share, KnnArgs::calced[4]; wave-uniform)
    bool set_live[NSET];
#pragma unroll
    for (int e = 0; e < NSET; e++) set_live[e] = __ballot(!pruned[e]) != 0ull;
    auto mfma_tile = [&](int buf, int sub, bool dma, uint32_t dma_base, int dma_buf) {
      const uint32_t tb = fragbase + (uint32_t)buf * TILEB + (uint32_t)sub * (32 * ROWB);
      const uint32_t bb = bias0 + (uint32_t)buf * 256u + (uint32_t)sub * 128u + 16u * h;
      f32x4 b4[4];
#pragma unroll
      for (int g = 0; g < 4; g++) b4[g] = knn_lds_read4(bb + 32u * g);
      constexpr int PD = KS < KNN16_PD ? KS : KNN16_PD;   // fragments in flight
      f16x8 fr[PD + 1];
#pragma unroll
      for (int j = 0; j < PD; j++) fr[j] = knn_frag_issue(tb + ((16u * j) ^ fragswz));
      asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(b4[0]), "+v"(b4[1]), "+v"(b4[2]), "+v"(b4[3]) : "n"(PD));
#pragma unroll
      for (int e = 0; e < NSET; e++) {
#pragma unroll
        for (int g = 0; g < 4; g++) {
          acc[e][4 * g + 0] = b4[g].x; acc[e][4 * g + 1] = b4[g].y; acc[e][4 * g + 2] = b4[g].z; acc[e][4 * g + 3] = b4[g].w;
        }
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
#pragma unroll
        for (int e = 0; e < NSET; e++)
          if (NSET == 1 || set_live[e]) acc[e] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f, xhi[e][j], acc[e], 0, 0, 0);
        {
          const int slot = sub * KS + j;   // compile-time after unrolling
          if (dma && slot % DSTR == 0 && slot / DSTR <= PPW) issue_piece(dma_base, dma_buf, slot / DSTR);
        }
      }
    };
    // which of the sub-tile's 32 candidates can still be accepted by which query of set e: queue them, settle
    // full queues
    auto mask_tile = [&](int e, uint32_t tile_base) {
      uint32_t m16 = 0;
      // the tile's best score first: most tiles hold no candidate for any query of the wave.  fmaxf ignores a
      // NaN operand (NaN scores never pass); amin = -inf while the heap is not full (every finite score passes)
      bool some = false;
      if (!pruned[e]) {
        // (builtins, not inline asm: the MFMA -> VALU read hazard of acc[] stays the compiler's business)
        const f32x16 &c = acc[e];
        const float m0 = __builtin_fmaxf(__builtin_fmaxf(c[0], c[1]), c[2]);
        const float m1 = __builtin_fmaxf(__builtin_fmaxf(c[3], c[4]), c[5]);
        const float m2 = __builtin_fmaxf(__builtin_fmaxf(c[6], c[7]), c[8]);
        const float m3 = __builtin_fmaxf(__builtin_fmaxf(c[9], c[10]), c[11]);
        const float m4 = __builtin_fmaxf(__builtin_fmaxf(c[12], c[13]), c[14]);
        const float m5 = __builtin_fmaxf(__builtin_fmaxf(m0, m1), c[15]);
        const float m6 = __builtin_fmaxf(__builtin_fmaxf(m2, m3), m4);
        some = __builtin_fmaxf(m5, m6) >= amin[e];
      }
      if (__ballot(some) != 0ull) {
        if (some) {
#pragma unroll
          for (int r = 0; r < 16; r++) m16 |= (acc[e][r] >= amin[e] ? 1u : 0u) << r;
        }
        const uint32_t pm = __shfl_xor(m16, 32);
        const uint32_t m0 = h ? pm : m16, m1 = h ? m16 : pm;
        uint32_t rowmask = 0;
#pragma unroll
        for (int g = 0; g < 4; g++)
          rowmask |= (((m0 >> (4 * g)) & 0xFu) << (8 * g)) | (((m1 >> (4 * g)) & 0xFu) << (8 * g + 4));
        while (__ballot(rowmask != 0u) != 0ull) {
          if (__ballot(qn_[e] == 4) != 0ull) flush(e);
          bool active = rowmask != 0u;
          const uint32_t rho = active ? (uint32_t)__ffs((int)rowmask) - 1u : 0u;
          rowmask &= rowmask - 1u;
          const uint32_t cp = tile_base + rho;
          if (step == 0 && cp == qp[e]) active = false;  // knn.cu:204-206: not its own neighbour
          if (cp >= end) active = false;                 // tile padding passes while the heap is not full
          if (active) {
#pragma unroll
            for (int i = 0; i < 4; i++)
              if (i == qn_[e]) qc[e][i] = cp;
            qn_[e]++;
          }
        }
      }
    };
    for (uint32_t t = 0; t < ntiles; t++) {
      const int buf = (int)(t % NBUF);
      const uint32_t tile_base = beg + t * (32 * SUB);
      {
        // tile t has landed (my pieces: counted wait; everybody's: the barrier), and every wave is done with
        // tile t - 1, whose buffer the tile NBUF - 1 ahead goes into
        const uint32_t ahead = ntiles - 1 - t;
        wait_tiles(ahead >= (uint32_t)(NBUF - 2));
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
      }
      const bool dma = t + (NBUF - 1) < ntiles;
      const uint32_t dma_base = tile_base + 32u * SUB * (NBUF - 1);
      const int dma_buf = (int)((t + NBUF - 1) % NBUF);
      if (dma && !wave_need) issue_tile(dma_base, dma_buf);
      if (wave_need) {
#pragma unroll
        for (int sub = 0; sub < SUB; sub++) {
          // (a last tile of one sub-tile only has no tile NBUF - 1 ahead: no piece is lost by the break)
          if (sub > 0 && tile_base + 32u * sub >= end) break;   // block-uniform
          mfma_tile(buf, sub, dma, dma_base, dma_buf);
          {   // statistics (wave-uniform)
            scored += 1024ull * (NSET == 1 ? 1u : nsets_live);
            scored_sets += 1024ull * nsets_live;
            const uint32_t left = end - (tile_base + 32u * sub);
            useful += (unsigned long long)nvis_tight * (left < 32u ? left : 32u);
          }
          if (sub == SUB - 1 && dma) {   // pieces the slots did not cover (very short rows)
#pragma unroll
            for (int i = (SUB * KS - 1) / DSTR + 1; i <= PPW; i++) issue_piece(dma_base, dma_buf, i);
          }
#pragma unroll
          for (int e = 0; e < NSET; e++) mask_tile(e, tile_base + 32u * sub);
        }
      }
    }
  }
#pragma unroll
  for (int e = 0; e < NSET; e++) {
    if (__ballot(qn_[e] > 0) != 0ull) flush(e);
    if (live[e] && h == 0) {  // knn.cu:239-242
      float *heap = a.heaps + (size_t)(qp[e] - a.p_base) * 2 * k;
      uint32_t *out = a.out + (size_t)(qp[e] - a.p_base) * k;
      for (int i = (int)k - 1; i >= 0; i--) {
        out[i] = reinterpret_cast<uint32_t *>(heap)[1];
        knn_push_sample(k, -1.f, 0xFFFFFFFFu, heap);
      }
    }
  }
  if (lane == 0 && calced) atomicAdd(a.calced, calced);
  if (lane == 0 && scored) {
    atomicAdd(a.calced + 1, scored);
    atomicAdd(a.calced + 2, useful);
    atomicAdd(a.calced + 4, scored_sets);
  }
#pragma unroll
  for (int off = 16; off > 0; off >>= 1) chains += __shfl_xor(chains, off);   // (the upper half-wave holds zeros)
  if (lane == 0 && chains) atomicAdd(a.calced + 3, (unsigned long long)chains);
}

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
    hipLaunchKernelGGL((knn_split_kernel<0>), dim3(wave_row_grid(N)), dim3(256), 0, st, xs, N, D, DP, mu,
                       reinterpret_cast<_Float16 *>(xs16), n2c, mux, kbias, stats);
  else
    hipLaunchKernelGGL((knn_split_kernel<1>), dim3(wave_row_grid(N)), dim3(256), 0, st, xs, N, D, DP, mu,
                       reinterpret_cast<_Float16 *>(xs16), n2c, mux, kbias, stats);
  return hipGetLastError();
}

template <int DP, int METRIC>
static hipError_t launch_knn_f16_t(const KnnArgs &a, uint32_t nblocks, hipStream_t st) {
  const size_t lds_bytes = (size_t)knn16_nbuf(DP) * (32 * knn16_sub(DP) * DP * 2) + knn16_nbuf(DP) * 256 + 2 * knn16_waves(DP) * 4;
  if (lds_bytes > 65536) {   // (per launch: the attribute belongs to the current device's copy of the kernel)
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&knn_filter_f16_kernel<DP, METRIC, true>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e == hipSuccess)
      e = hipFuncSetAttribute(reinterpret_cast<const void *>(&knn_filter_f16_kernel<DP, METRIC, false>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e != hipSuccess) return e;
  }
  if (a.D == (uint32_t)DP)
    hipLaunchKernelGGL((knn_filter_f16_kernel<DP, METRIC, true>), dim3(nblocks), dim3(knn16_waves(DP) * 64), lds_bytes, st, a);
  else
    hipLaunchKernelGGL((knn_filter_f16_kernel<DP, METRIC, false>), dim3(nblocks), dim3(knn16_waves(DP) * 64), lds_bytes, st, a);
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
    KMX_KNN16_CASE(512);
    KMX_KNN16_CASE(768);
    KMX_KNN16_CASE(1024);
    default: return hipErrorInvalidValue;
  }
#undef KMX_KNN16_CASE
}

}  // namespace kmx
