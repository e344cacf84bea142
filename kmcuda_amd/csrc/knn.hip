// knn.hip -- cluster-pruned k nearest neighbours (reference: src/knn.cu) re-designed for
// MI355X / gfx950.
//
// Reference semantics (kept, knn.cu:177-243): per sample a max-heap of k (distance, index)
// pairs; candidates are visited as "own cluster members in ascending index (skipping the
// sample itself), then every other cluster in ascending id unless the triangle-inequality
// test  C[cls][mine] - d(s, c_mine) - R[cls] > kth  prunes it"; a candidate is pushed iff
// distance <= kth (push_sample, knn.cu:133-175); the output is the heap popped from the back.
// All distances are the reference's exact arithmetic (distance_tt: Kahan / round-down-FMA sum of
// squared differences, correctly rounded sqrt; metric_abstraction.h:88-101).
//
// Reference mechanics (discarded): one thread per sample gathering candidate rows from a
// feature-major matrix (fully uncoalesced) and evaluating EVERY candidate with the 4-ops-per-MAC
// exact chain.  Here:
//   gather        samples are copied once into CLUSTER-SORTED order (position p = rank in the
//                 inverse assignments, kmcuda.cc:648-691), rows padded to the filter width, so a
//                 candidate cluster is one contiguous slab that streams through LDS.
//   filter        queries x candidates^T on the f32 MFMA, exactly as the Lloyd filter: a wave
//                 keeps 32 queries of ONE cluster resident in VGPRs, candidate tiles of 32 rows
//                 stream through LDS (shared by the block's 4 waves), the accumulator is seeded
//                 with -||y||^2/2 so  ||x-y||^2 ~= ||x||^2 - 2*acc.  A candidate whose approximate
//                 distance exceeds the query's current kth distance by more than a rigorous
//                 error bound would be REJECTED by the reference (distance > kth), and a rejected
//                 candidate never changes the heap -- so it is dropped without evaluation.
//   refine        the survivors (the first k candidates, then ~k*ln(n/k) more per query) are
//                 evaluated with the exact arithmetic IN THE REFERENCE'S VISITING ORDER and pushed
//                 through the same heap, so heap evolution, pruning decisions, neighbour indices
//                 and their order are identical to the reference's.
// knn_exact_kernel is the same search with every candidate evaluated exactly (no filter): the
// in-library cross-check, and the path for feature counts the filter is not instantiated for.
#include "exact.hpp"
#include "half2_ops.hpp"
#include "kernels.hpp"
#include "knn_heap.hpp"

namespace kmx {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// ---------------------------------------------------------------------------------------
// prep 1: cluster-sorted padded copy, plain squared norms (filter only), max norm
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void knn_gather_kernel(const float *__restrict__ samples, uint32_t N, uint32_t D,
                                                         uint32_t DP, const uint32_t *__restrict__ inv,
                                                         float *__restrict__ xs, float *__restrict__ n2s,
                                                         uint32_t *__restrict__ stats) {
  const uint32_t lane = threadIdx.x & 63;
  for (uint32_t p = blockIdx.x * 4 + (threadIdx.x >> 6); p < N; p += gridDim.x * 4) {   // (kernels.hpp: wave_row_grid)
    const float *src = samples + (size_t)inv[p] * D;
    float *dst = xs + (size_t)p * DP;
    float a = 0.f;
    for (uint32_t f = lane; f < DP; f += 64) {
      const float v = f < D ? src[f] : 0.f;
      dst[f] = v;
      a = fmaf(v, v, a);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) a += __shfl_xor(a, off);
    if (lane == 0) {
      n2s[p] = a;
      // finite, non-negative: bits order like values.  Look before the atomic: N of them on one address serialise
      // in L2 (8M rows: 90 ms for a kernel that moves 16 GB), and the running maximum rarely moves
      if ((a - a) == 0.f && __float_as_uint(a) > *reinterpret_cast<volatile uint32_t *>(&stats[0]))
        atomicMax(&stats[0], __float_as_uint(a));
    }
  }
}

__device__ __forceinline__ uint32_t cluster_of(const uint32_t *__restrict__ offsets, uint32_t K, uint32_t p) {
  // largest c with offsets[c] <= p, c in [0, K]; K means "no cluster" (p >= offsets[K])
  uint32_t lo = 0, hi = K + 1;
  while (hi - lo > 1) {
    const uint32_t mid = (lo + hi) / 2;
    if (offsets[mid] <= p) lo = mid; else hi = mid;
  }
  return lo;
}

// metric_abstraction.h:103-118 partial (L2: Kahan sum of squared differences; angular: Kahan dot)
// H2 (KMCUDA_AMD_FP16_STRICT): the same with F = half2 (half2_ops.hpp; the rows hold half values, n is even)
template <int METRIC, bool H2 = false>
__device__ __forceinline__ float partial_vv(const float *__restrict__ a, const float *__restrict__ b, uint32_t n) {
  if constexpr (H2) return h2_partial<METRIC>(a, b, n);
  float acc = 0.f, corr = 0.f;
  for (uint32_t f = 0; f < n; f++) {
    if (METRIC == 0) {
      const float d = a[f] - b[f];
      kahan_fold(fma_rd(d, d, corr), acc, corr);
    } else {
      kahan_fold(fma_rd(a[f], b[f], corr), acc, corr);
    }
  }
  return acc;
}
template <int METRIC>
__device__ __forceinline__ float finalize(float p) {  // metric_abstraction.h:134-136 / :248-253
  return METRIC == 0 ? sqrtf(p) : angular_from_prod(p);
}

// ---------------------------------------------------------------------------------------
// prep 2: per member, distance to its own centroid (knn.cu:190-191, distance_t) and the CHUNKED
// distance the radius is built from (knn.cu:31-45: 16-feature partials added with plain '+')
// ---------------------------------------------------------------------------------------
template <int METRIC, bool H2>
__global__ void knn_member_kernel(const float *__restrict__ xs, uint32_t N, uint32_t D, uint32_t DP,
                                  const uint32_t *__restrict__ offsets, uint32_t K,
                                  const float *__restrict__ centroids, float *__restrict__ mydist,
                                  float *__restrict__ rdist) {
  const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= N) return;
  const uint32_t c = cluster_of(offsets, K, p);
  if (c >= K) {
    mydist[p] = NAN;
    rdist[p] = NAN;
    return;
  }
  const float *x = xs + (size_t)p * DP, *cen = centroids + (size_t)c * D;
  // (half2: distance_t rounds the angle to half, finalize does not -- metric_abstraction.h:171-177 vs :248-253)
  mydist[p] = H2 ? h2_distance<METRIC>(x, cen, D) : finalize<METRIC>(partial_vv<METRIC>(x, cen, D));
  // CLUSTER_RADIUSES_SHMEM / blockDim = 8192 / 512 = 16 elements of F: 16 features, or 16 half2 = 32 halves
  const uint32_t step = H2 ? (D < 32 ? D : 32) : (D < 16 ? D : 16);
  float sd = 0.f;
  for (uint32_t cfi = 0; cfi < D; cfi += step) {
    const uint32_t fsize = (D - cfi) < step ? (D - cfi) : step;
    sd += partial_vv<METRIC, H2>(x + cfi, cen + cfi, fsize);
  }
  rdist[p] = finalize<METRIC>(sd);
}

// The same two chains per member with the rows coming in through LDS (D % 32 == 0, 16-byte aligned rows): a thread
// walking its own 1-KB row makes every load of a wave touch 64 different lines (27 ms for 8M x 256: 0.3 TB/s); 8 lanes
// fetch a row's 128-byte line per 32-feature chunk, the tile has a 36-float row stride (conflict-free 16-byte reads),
// the next chunk's loads fly during the chains -- seeding.hip's kmpp_step2_kernel, with a centroid per row (the rows
// are cluster-sorted: a block's threads share one or two centroid rows, served by L1).  A 32-feature chunk is two of
// the reference's 16-feature partials.
constexpr int kMemberBlock = 256;
template <int METRIC>
__global__ __launch_bounds__(kMemberBlock) void knn_member_tiled_kernel(const float *__restrict__ xs, uint32_t N, uint32_t D,
                                                                        uint32_t DP, const uint32_t *__restrict__ offsets,
                                                                        uint32_t K, const float *__restrict__ centroids,
                                                                        float *__restrict__ mydist, float *__restrict__ rdist) {
  __shared__ __attribute__((aligned(16))) float tile[kMemberBlock * 36];
  const uint32_t p = blockIdx.x * kMemberBlock + threadIdx.x;
  const uint32_t c = p < N ? cluster_of(offsets, K, p) : K;
  const float *cen = centroids + (size_t)(c < K ? c : 0) * D;
  const uint32_t nchunk = D / 32;
  float4 stage[8];
  auto fetch = [&](uint32_t ch) {
#pragma unroll
    for (int q = 0; q < 8; q++) {
      const uint32_t i = threadIdx.x + kMemberBlock * q, r = i >> 3, c4 = i & 7u;
      const uint32_t row = blockIdx.x * kMemberBlock + r;
      stage[q] = row < N ? *reinterpret_cast<const float4 *>(xs + (size_t)row * DP + ch * 32 + c4 * 4)
                         : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  float acc = 0.f, corr = 0.f, sd = 0.f;
  fetch(0);
  for (uint32_t ch = 0; ch < nchunk; ch++) {
    __syncthreads();   // the previous chunk has been consumed
#pragma unroll
    for (int q = 0; q < 8; q++) {
      const uint32_t i = threadIdx.x + kMemberBlock * q, r = i >> 3, c4 = i & 7u;
      *reinterpret_cast<float4 *>(&tile[r * 36 + c4 * 4]) = stage[q];
    }
    __syncthreads();
    if (ch + 1 < nchunk) fetch(ch + 1);
#pragma unroll
    for (int half = 0; half < 2; half++) {   // one 16-feature partial of the radius distance (knn.cu:31-45)
      float pacc = 0.f, pcorr = 0.f;
#pragma unroll
      for (int c4 = 0; c4 < 4; c4++) {
        const float4 xv = *reinterpret_cast<const float4 *>(&tile[threadIdx.x * 36 + half * 16 + c4 * 4]);
        const float4 cv = *reinterpret_cast<const float4 *>(cen + ch * 32 + half * 16 + c4 * 4);
        const float aa[4] = {xv.x, xv.y, xv.z, xv.w}, bb[4] = {cv.x, cv.y, cv.z, cv.w};
#pragma unroll
        for (int q = 0; q < 4; q++) {
          if (METRIC == 0) {
            const float d = aa[q] - bb[q];
            kahan_fold(fma_rd(d, d, corr), acc, corr);      // the one chain over all features (distance_t)
            kahan_fold(fma_rd(d, d, pcorr), pacc, pcorr);   // the partial's own chain
          } else {
            kahan_fold(fma_rd(aa[q], bb[q], corr), acc, corr);
            kahan_fold(fma_rd(aa[q], bb[q], pcorr), pacc, pcorr);
          }
        }
      }
      sd += pacc;
    }
  }
  if (p < N) {
    mydist[p] = c < K ? finalize<METRIC>(acc) : NAN;
    rdist[p] = c < K ? finalize<METRIC>(sd) : NAN;
  }
}

// knn.cu:46-57: radius = max member distance, NaN for an empty cluster.  One wave per cluster.
__global__ __launch_bounds__(64) void knn_radii_kernel(const float *__restrict__ rdist,
                                                       const uint32_t *__restrict__ offsets, uint32_t K,
                                                       float *__restrict__ R) {
  const uint32_t c = blockIdx.x;
  float mx = -1.f;
  for (uint32_t p = offsets[c] + threadIdx.x; p < offsets[c + 1]; p += 64) {
    const float d = rdist[p];
    if (d > mx) mx = d;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const float o = __shfl_xor(mx, off);
    if (o > mx) mx = o;
  }
  if (threadIdx.x == 0) R[c] = mx > -1.f ? mx : NAN;
}

// knn.cu:61-131: K x K centroid distances from 24-feature partials (12288 / 512), finalized.
template <int METRIC, bool H2>
__global__ void knn_cdist_kernel(const float *__restrict__ centroids, uint32_t K, uint32_t D, float *__restrict__ C) {
  const uint32_t jb = (K + blockDim.x - 1) / blockDim.x;  // 1-D grid: K may exceed the grid.y limit
  const uint32_t i = blockIdx.x / jb, j = (blockIdx.x % jb) * blockDim.x + threadIdx.x;
  if (j >= K) return;
  const float *a = centroids + (size_t)i * D, *b = centroids + (size_t)j * D;
  float acc = 0.f;
  constexpr uint32_t kStep = H2 ? 48 : 24;   // 24 elements of F (half2: 48 halves)
  for (uint32_t fpos = 0; fpos < D; fpos += kStep) {
    const uint32_t fsize = (D - fpos) < kStep ? (D - fpos) : kStep;
    acc += partial_vv<METRIC, H2>(a + fpos, b + fpos, fsize);
  }
  C[(size_t)i * K + j] = finalize<METRIC>(acc);
}

// ---------------------------------------------------------------------------------------
// the filtered search
// ---------------------------------------------------------------------------------------
// Block = 256 threads = 4 waves; each wave owns 32 consecutive sorted positions of ONE cluster
// (blocks[] = (cluster, first position); a block never straddles clusters, so the visiting
// order "own cluster, then 0..K-1" is block-uniform and the candidate tiles are shared).
// MFMA orientation as in lloyd.hip: A = candidate rows (32), B = queries (32 columns); lane l
// holds, for query l&31, the scores of rows (r&3) + 8*(r>>2) + 4*(l>>5), r = 0..15; the lower
// half-wave contracts features [0, DP/2), the upper half [DP/2, DP).
template <int DP, int METRIC>
__global__ __launch_bounds__(256, 2) void knn_filter_kernel(KnnArgs a) {
  constexpr int NK = DP / 2;
  constexpr int LDW = DP + 4;
  constexpr int TILE = 32 * LDW;
  constexpr int NST = (8 * DP + 255) / 256;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  auto tile_ptr = [&](int buf) { return lds + buf * TILE; };
  auto bias_ptr = [&](int buf) { return lds + 2 * TILE + buf * 32; };
  uint32_t *flags = reinterpret_cast<uint32_t *>(lds + 2 * TILE + 64);  // 2 x 4 words

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int col = lane & 31;
  const int h = lane >> 5;
  const uint32_t K = a.K, k = a.k, D = a.D;

  const uint32_t cls0 = a.blocks[2 * (size_t)blockIdx.x], p0 = a.blocks[2 * (size_t)blockIdx.x + 1];
  const uint32_t own_end = a.offsets[cls0 + 1];
  const uint32_t qp = p0 + wave * 32 + col;
  const bool live = qp < own_end;

  float xb[NK];
  {
    const f32x4 *src = reinterpret_cast<const f32x4 *>(a.xs + (size_t)(live ? qp : p0) * DP + h * NK);
#pragma unroll
    for (int j = 0; j < NK / 4; j++) {
      const f32x4 v = src[j];
      xb[4 * j + 0] = live ? v.x : 0.f;
      xb[4 * j + 1] = live ? v.y : 0.f;
      xb[4 * j + 2] = live ? v.z : 0.f;
      xb[4 * j + 3] = live ? v.w : 0.f;
    }
  }
  const int nvalid = (int)D - h * NK < 0 ? 0 : ((int)D - h * NK > NK ? NK : (int)D - h * NK);  // real features here
  const float qn2 = live ? a.n2s[qp] : 0.f;
  const float md = live ? a.mydist[qp] : 0.f;
  float *heap = a.heaps + (size_t)((live ? qp : p0) - a.p_base) * 2 * k;
  if (live && h == 0) {
    for (uint32_t i = 0; i < k; i++) {
      heap[2 * i] = 3.402823466e+38f;
      reinterpret_cast<uint32_t *>(heap)[2 * i + 1] = 0;
    }
  }
  float mndist = 3.402823466e+38f;

  // Filter threshold in accumulator space ("k-NN filter bound"): a candidate can only
  // be accepted by the reference if  acc >= amin.
  const float nmax2 = __uint_as_float(a.stats[0]);
  float E;
  if (METRIC == 0) E = 4.04f * a.eps * (qn2 + nmax2);
  else E = 1.01f * a.eps * (sqrtf(qn2) * sqrtf(nmax2) * 1.0001f) + 1e-6f;
  auto amin_of = [&](float mnd) -> float {
    if (METRIC == 0) {
      const float T2 = mnd * mnd * 1.000001f;  // inf when the heap is not full yet
      return 0.5f * (qn2 - T2 - E) - 1e-6f * (qn2 + T2);
    }
    if (mnd >= 3.1415925f) return -INFINITY;
    return cosf(mnd) - E;
  };
  float amin = amin_of(mndist);

  f32x4 stage[NST];
  float bstage = 0.f;
  auto stage_load = [&](uint32_t base, uint32_t end) {  // 32 sorted rows from `base`
    const uint32_t last = a.N - 1;
#pragma unroll
    for (int i = 0; i < NST; i++) {
      const int q = tid + i * 256;
      if (q < 8 * DP) {
        const uint32_t row = base + q / (DP / 4);
        const uint32_t rr = row <= last ? row : last;
        stage[i] = reinterpret_cast<const f32x4 *>(a.xs + (size_t)rr * DP)[q % (DP / 4)];
      }
    }
    if (tid < 32) {
      const uint32_t row = base + tid;
      if (row < end) bstage = METRIC == 0 ? -0.5f * a.n2s[row] : 0.f;
      else bstage = -INFINITY;  // rows past the cluster never pass the filter
    }
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
      const float lim = cd - md - a.R[cls];
      pruned = pruned || (lim > mndist);          // knn.cu:222-225
    }
    if (beg == end) continue;                     // nothing to visit (block-uniform)
    const unsigned long long visiting = __ballot(!pruned);
    const bool wave_need = visiting != 0ull;
    if (lane == 0) flags[ph * 4 + wave] = wave_need ? 1u : 0u;
    __syncthreads();
    const bool need = (flags[ph * 4] | flags[ph * 4 + 1] | flags[ph * 4 + 2] | flags[ph * 4 + 3]) != 0u;
    ph ^= 1;
    if (!need) continue;
    calced += (unsigned long long)__popcll(visiting & 0xFFFFFFFFull) * (end - beg);  // knn.cu:228 per query

    const uint32_t ntiles = (end - beg + 31) / 32;
    stage_load(beg, end);
    stage_store(0);
    __syncthreads();
    for (uint32_t t = 0; t < ntiles; t++) {
      const int buf = t & 1;
      const uint32_t tile_base = beg + t * 32;
      if (t + 1 < ntiles) stage_load(tile_base + 32, end);
      if (wave_need) {
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
#pragma unroll
        for (int j = 0; j < NK / 4; j++) {
          const f32x4 a4 = *reinterpret_cast<const f32x4 *>(arow + 4 * j);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.x, xb[4 * j + 0], acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.y, xb[4 * j + 1], acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.z, xb[4 * j + 2], acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.w, xb[4 * j + 3], acc, 0, 0, 0);
        }
        uint32_t m16 = 0;
        if (!pruned) {
#pragma unroll
          for (int r = 0; r < 16; r++) m16 |= (acc[r] >= amin ? 1u : 0u) << r;  // NaN scores never pass
        }
        if (__ballot(m16 != 0u) != 0ull) {
          // ---- refine: survivors in ascending row order, exact arithmetic, reference heap ----
          const uint32_t pm = __shfl_xor(m16, 32);
          const uint32_t m0 = h ? pm : m16, m1 = h ? m16 : pm;
          uint32_t rowmask = 0;
#pragma unroll
          for (int g = 0; g < 4; g++)
            rowmask |= (((m0 >> (4 * g)) & 0xFu) << (8 * g)) | (((m1 >> (4 * g)) & 0xFu) << (8 * g + 4));
          while (__ballot(rowmask != 0u) != 0ull) {
            bool active = rowmask != 0u;
            const uint32_t rho = active ? (uint32_t)__ffs((int)rowmask) - 1u : 0u;
            rowmask &= rowmask - 1u;
            const uint32_t cp = tile_base + rho;
            if (step == 0 && cp == qp) active = false;  // knn.cu:204-206: not its own neighbour
            if (cp >= end) active = false;  // tile padding (-inf score) passes while the heap is not full (amin = -inf)
            const float *crow = tile_ptr(buf) + rho * LDW + h * NK;
            // one serial Kahan chain over the D features: the lower half-wave runs features
            // [0, NK), hands (acc, corr) to the upper half which continues with [NK, D)
            float dacc = 0.f, dcorr = 0.f;
#pragma unroll
            for (int pass = 0; pass < 2; pass++) {
              if (pass == 1) {
                dacc = __shfl(dacc, col);
                dcorr = __shfl(dcorr, col);
              }
#pragma unroll
              for (int j = 0; j < NK; j++) {
                float y;
                if (METRIC == 0) {
                  const float d = xb[j] - crow[j];
                  y = fma_rd(d, d, dcorr);
                } else {
                  y = fma_rd(xb[j], crow[j], dcorr);
                }
                const float tt = dacc + y;
                const float nc = y - (tt - dacc);
                const bool on = j < nvalid;
                dacc = on ? tt : dacc;
                dcorr = on ? nc : dcorr;
              }
            }
            const float dist = finalize<METRIC>(__shfl(dacc, col + 32));
            if (h == 0 && active && dist <= mndist) {  // knn.cu:209-212
              knn_push_sample(k, dist, a.inv[cp], heap);
              mndist = heap[0];
            }
            mndist = __shfl(mndist, col);
            amin = amin_of(mndist);
          }
        }
      }
      if (t + 1 < ntiles) stage_store(buf ^ 1);
      __syncthreads();
    }
  }
  if (live && h == 0) {  // knn.cu:239-242
    uint32_t *out = a.out + (size_t)(qp - a.p_base) * k;
    for (int i = (int)k - 1; i >= 0; i--) {
      out[i] = reinterpret_cast<uint32_t *>(heap)[1];
      knn_push_sample(k, -1.f, 0xFFFFFFFFu, heap);
    }
  }
  if (lane == 0 && calced) atomicAdd(a.calced, calced);
}

// ---------------------------------------------------------------------------------------
// the unfiltered search: one thread per sorted position, every candidate evaluated exactly
// ---------------------------------------------------------------------------------------
template <int METRIC, bool H2>
__global__ __launch_bounds__(64) void knn_exact_kernel(KnnArgs a) {
  const uint32_t qp = a.p_base + blockIdx.x * blockDim.x + threadIdx.x;
  if (qp >= a.p_end) return;
  const uint32_t K = a.K, k = a.k, D = a.D, DP = a.DP;
  const uint32_t mycls = cluster_of(a.offsets, K, qp);
  uint32_t *out = a.out + (size_t)(qp - a.p_base) * k;
  if (mycls >= K) {  // a row without a cluster (NaN sample): the reference reads out of bounds here
    for (uint32_t i = 0; i < k; i++) out[i] = 0xFFFFFFFFu;
    return;
  }
  const float *x = a.xs + (size_t)qp * DP;
  const float md = a.mydist[qp];
  float *heap = a.heaps + (size_t)(qp - a.p_base) * 2 * k;
  for (uint32_t i = 0; i < k; i++) {
    heap[2 * i] = 3.402823466e+38f;
    reinterpret_cast<uint32_t *>(heap)[2 * i + 1] = 0;
  }
  float mndist = 3.402823466e+38f;
  unsigned long long calced = 0;
  for (uint32_t step = 0; step <= K; step++) {
    const uint32_t cls = step == 0 ? mycls : step - 1;
    if (step > 0) {
      if (cls == mycls) continue;
      const float cd = a.C[(size_t)cls * K + mycls];
      if (cd != cd) continue;
      if (cd - md - a.R[cls] > mndist) continue;
    }
    const uint32_t beg = a.offsets[cls], end = a.offsets[cls + 1];
    calced += end - beg;
    for (uint32_t cp = beg; cp < end; cp++) {
      if (cp == qp) continue;
      const float dist = H2 ? h2_distance<METRIC>(x, a.xs + (size_t)cp * DP, D)   // distance_tt, F = half2
                            : finalize<METRIC>(partial_vv<METRIC>(x, a.xs + (size_t)cp * DP, D));
      if (dist <= mndist) {
        knn_push_sample(k, dist, a.inv[cp], heap);
        mndist = heap[0];
      }
    }
  }
  for (int i = (int)k - 1; i >= 0; i--) {
    out[i] = reinterpret_cast<uint32_t *>(heap)[1];
    knn_push_sample(k, -1.f, 0xFFFFFFFFu, heap);
  }
  atomicAdd(a.calced, calced);
}

// ---------------------------------------------------------------------------------------
// A lower bound of "distance to any member of cluster c" per (query, cluster): d(q, c) - R[c], from the query's own
// distance to the centroid (the reference bounds that one by the triangle C[c][mine] - d(q, c_mine): knn.cu:222-225).
// Plain fp32 sums of squared differences, a wave per four consecutive sorted rows, 4 features per lane and 256-feature
// chunk; rounded DOWN past every error in sight: the fp32 sum ((D + 3) u relative on the square), the reference's own
// candidate distances (Kahan + round-down products: ~2e-7) and its radii (round-down products, chunked float sums:
// ~1e-6 up to 256 features, 4e-6 at 1024) -- 2e-5 or (D + 16) u, 3e-5 and 3e-5 relative are taken.
// 1M queries x 1024 centroids x 256 features: ~20 ms beside a search of seconds.
// ---------------------------------------------------------------------------------------
// (Round 4: 64 queries per block -- a lane per query, a wave per quarter of the features, the queries' values in
// registers, the centroid's values broadcast from LDS -- instead of a wave per four rows with a 6-step butterfly per
// (row, centroid) and 16-byte writes: 63 -> see profiles/README.md for the 1M x 1024 x 256 case of config D.)
constexpr int kCbQ = 64, kCbTile = 16;   // queries per block; centroids staged per round
__global__ __launch_bounds__(256) void knn_centroid_bounds_kernel(const float *__restrict__ xs, uint32_t D, uint32_t DP,
                                                                  uint32_t p_base, uint32_t p_end,
                                                                  const float *__restrict__ centroids, uint32_t K,
                                                                  const float *__restrict__ R, float *__restrict__ lb,
                                                                  size_t stride) {
  // features in chunks of 256: wave w owns features [64 w, 64 w + 64) of the chunk
  __shared__ __attribute__((aligned(16))) float ctile[kCbTile][256];
  __shared__ float part[4][kCbTile][kCbQ];
  const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint32_t q0 = p_base + blockIdx.x * kCbQ;
  const uint32_t p = q0 + lane < p_end ? q0 + lane : q0;   // (lanes past the end repeat a row; their results are not written)
  const int nch = (int)((D + 255) / 256);
  const float sd = fminf(0.99998f, 1.0f - ((float)D + 16.0f) * 6.0e-8f);
  for (uint32_t c0 = 0; c0 < K; c0 += kCbTile) {
    float acc[kCbTile];
#pragma unroll
    for (int t = 0; t < kCbTile; t++) acc[t] = 0.f;
    for (int ch = 0; ch < nch; ch++) {
      __syncthreads();   // the previous chunk's tile has been consumed
      // stage 16 centroids x 256 features of this chunk (zeros beyond D / beyond K)
      for (uint32_t i = threadIdx.x; i < kCbTile * 256; i += 256) {
        const uint32_t t = i >> 8, f = ch * 256 + (i & 255u);
        ctile[t][i & 255u] = (c0 + t < K && f < D) ? centroids[(size_t)(c0 + t) * D + f] : 0.f;
      }
      // my 64 features of my query (xs rows are zero padded to DP; beyond DP: zeros against the tile's zeros)
      float xv[64];
#pragma unroll
      for (int f4 = 0; f4 < 16; f4++) {
        const uint32_t f = ch * 256 + wave * 64 + f4 * 4;
        const float4 v = f + 3 < DP ? *reinterpret_cast<const float4 *>(xs + (size_t)p * DP + f) : make_float4(0.f, 0.f, 0.f, 0.f);
        xv[4 * f4 + 0] = v.x; xv[4 * f4 + 1] = v.y; xv[4 * f4 + 2] = v.z; xv[4 * f4 + 3] = v.w;
      }
      __syncthreads();
#pragma unroll
      for (int t = 0; t < kCbTile; t++) {
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
        for (int f4 = 0; f4 < 16; f4++) {
          const float4 cv = *reinterpret_cast<const float4 *>(&ctile[t][wave * 64 + f4 * 4]);   // same address in every lane: a broadcast
          const float d0 = xv[4 * f4 + 0] - cv.x, d1 = xv[4 * f4 + 1] - cv.y, d2 = xv[4 * f4 + 2] - cv.z, d3 = xv[4 * f4 + 3] - cv.w;
          a0 = fmaf(d0, d0, a0); a1 = fmaf(d1, d1, a1); a2 = fmaf(d2, d2, a2); a3 = fmaf(d3, d3, a3);
        }
        acc[t] += (a0 + a1) + (a2 + a3);
      }
    }
#pragma unroll
    for (int t = 0; t < kCbTile; t++) part[wave][t][lane] = acc[t];
    __syncthreads();
    // 16 centroids x 64 queries = 1024 results, four per thread: thread -> (centroid t, query)
    for (uint32_t i = threadIdx.x; i < kCbTile * kCbQ; i += 256) {
      const uint32_t t = i >> 6, q = i & 63u;
      const uint32_t c = c0 + t;
      if (c < K && q0 + q < p_end) {
        const float d = sqrtf((part[0][t][q] + part[1][t][q]) + (part[2][t][q] + part[3][t][q]));
        // NaN (a NaN row or centroid, an empty cluster's radius): compares false in the search, nothing is skipped
        lb[(size_t)c * stride + (q0 + q - p_base)] = (d * sd - R[c] * 1.00003f) * 0.99997f;
      }
    }
    __syncthreads();   // part[] is reused by the next round
  }
}

hipError_t launch_knn_centroid_bounds(const float *xs, uint32_t D, uint32_t DP, uint32_t p_base, uint32_t p_end,
                                      const float *centroids, uint32_t K, const float *R, float *lb, size_t stride,
                                      hipStream_t st) {
  if (p_end <= p_base) return hipSuccess;
  if (D > 1024 || (DP & 3u)) return hipErrorInvalidValue;
  const uint32_t nrows = p_end - p_base;
  hipLaunchKernelGGL(knn_centroid_bounds_kernel, dim3((nrows + kCbQ - 1) / kCbQ), dim3(256), 0, st, xs, D, DP, p_base,
                     p_end, centroids, K, R, lb, stride);
  return hipGetLastError();
}

// neighbors[inv[p]][:] = sorted_out[p - p_base][:]
__global__ void knn_scatter_kernel(const uint32_t *__restrict__ sorted_out, const uint32_t *__restrict__ inv,
                                   uint32_t p_base, uint32_t p_end, uint32_t k, uint32_t *__restrict__ neighbors) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t total = (size_t)(p_end - p_base) * k;
  if (i >= total) return;
  const uint32_t p = p_base + (uint32_t)(i / k), j = (uint32_t)(i % k);
  neighbors[(size_t)inv[p] * k + j] = sorted_out[i];
}

// ---------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------
hipError_t launch_knn_gather(const float *samples, uint32_t N, uint32_t D, uint32_t DP, const uint32_t *inv,
                             float *xs, float *n2s, uint32_t *stats, hipStream_t st) {
  hipError_t e = hipMemsetAsync(stats, 0, sizeof(uint32_t), st);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(knn_gather_kernel, dim3(wave_row_grid(N)), dim3(256), 0, st, samples, N, D, DP, inv, xs, n2s, stats);
  return hipGetLastError();
}

hipError_t launch_knn_prep(int metric, const float *xs, uint32_t N, uint32_t D, uint32_t DP, const uint32_t *offsets,
                           uint32_t K, const float *centroids, float *mydist, float *rdist, float *R, float *C,
                           bool strict_h2, hipStream_t st) {
  // (the tiled member kernel: whole 32-feature chunks, 16-byte aligned rows and centroids; else, and for the half2
  //  arithmetic, a thread per row)
  const bool tiled = !strict_h2 && D >= 32 && (D & 31u) == 0 && (DP & 3u) == 0 && (((uintptr_t)xs | (uintptr_t)centroids) & 15u) == 0;
#define KMX_KNN_PREP(M, H)                                                                                          \
  do {                                                                                                              \
    if (tiled)                                                                                                      \
      hipLaunchKernelGGL((knn_member_tiled_kernel<M>), dim3((N + kMemberBlock - 1) / kMemberBlock), dim3(kMemberBlock), 0, \
                         st, xs, N, D, DP, offsets, K, centroids, mydist, rdist);                                   \
    else                                                                                                            \
      hipLaunchKernelGGL((knn_member_kernel<M, H>), dim3((N + 127) / 128), dim3(128), 0, st, xs, N, D, DP, offsets, K, \
                         centroids, mydist, rdist);                                                                 \
    hipLaunchKernelGGL((knn_cdist_kernel<M, H>), dim3(((K + 127) / 128) * K), dim3(128), 0, st, centroids, K, D, C); \
  } while (0)
  if (metric == 0) {
    if (strict_h2) KMX_KNN_PREP(0, true); else KMX_KNN_PREP(0, false);
  } else {
    if (strict_h2) KMX_KNN_PREP(1, true); else KMX_KNN_PREP(1, false);
  }
#undef KMX_KNN_PREP
  hipLaunchKernelGGL(knn_radii_kernel, dim3(K), dim3(64), 0, st, rdist, offsets, K, R);
  return hipGetLastError();
}

template <int DP, int METRIC>
static hipError_t launch_knn_filter_t(const KnnArgs &a, uint32_t nblocks, hipStream_t st) {
  const size_t lds_bytes = (2 * 32 * (DP + 4) + 64 + 8) * sizeof(float);
  hipLaunchKernelGGL((knn_filter_kernel<DP, METRIC>), dim3(nblocks), dim3(256), lds_bytes, st, a);
  return hipGetLastError();
}

hipError_t launch_knn_filter(int metric, const KnnArgs &a, uint32_t nblocks, hipStream_t st) {
  if (nblocks == 0) return hipSuccess;
#define KMX_KNN_CASE(dp)                                                             \
  case dp:                                                                           \
    return metric == 0 ? launch_knn_filter_t<dp, 0>(a, nblocks, st) : launch_knn_filter_t<dp, 1>(a, nblocks, st)
  switch (a.DP) {
    KMX_KNN_CASE(8);
    KMX_KNN_CASE(16);
    KMX_KNN_CASE(32);
    KMX_KNN_CASE(64);
    KMX_KNN_CASE(128);
    KMX_KNN_CASE(256);
    default: return hipErrorInvalidValue;
  }
#undef KMX_KNN_CASE
}

hipError_t launch_knn_exact(int metric, const KnnArgs &a, bool strict_h2, hipStream_t st) {
  if (a.p_end <= a.p_base) return hipSuccess;
  const uint32_t grid = (a.p_end - a.p_base + 63) / 64;
  if (metric == 0) {
    if (strict_h2) hipLaunchKernelGGL((knn_exact_kernel<0, true>), dim3(grid), dim3(64), 0, st, a);
    else hipLaunchKernelGGL((knn_exact_kernel<0, false>), dim3(grid), dim3(64), 0, st, a);
  } else {
    if (strict_h2) hipLaunchKernelGGL((knn_exact_kernel<1, true>), dim3(grid), dim3(64), 0, st, a);
    else hipLaunchKernelGGL((knn_exact_kernel<1, false>), dim3(grid), dim3(64), 0, st, a);
  }
  return hipGetLastError();
}

hipError_t launch_knn_scatter(const uint32_t *sorted_out, const uint32_t *inv, uint32_t p_base, uint32_t p_end,
                              uint32_t k, uint32_t *neighbors, hipStream_t st) {
  const size_t total = (size_t)(p_end - p_base) * k;
  if (total == 0) return hipSuccess;
  hipLaunchKernelGGL(knn_scatter_kernel, dim3((uint32_t)((total + 255) / 256)), dim3(256), 0, st, sorted_out, inv,
                     p_base, p_end, k, neighbors);
  return hipGetLastError();
}

}  // namespace kmx
