// update.hip -- centroid update (reference: src/kmeans.cu:366-429 kmeans_adjust + normalize,
// metric_abstraction.h:138-144, :255-302) re-designed for MI355X.
//
// Reference semantics (kept): the update is INCREMENTAL --
//     c_new = normalize( c_old * count_old  +  sum(samples that moved in)  -  sum(moved out) )
// with normalize = divide by the new count (L2) or scale to unit length (angular).  For the
// angular metric this is not the same thing as "normalise the sum of the members" (c_old has
// unit length, so c_old*count_old over-weights the old direction); the reference's iteration
// pins (test.py:426-466) depend on it, so the incremental form is what we implement.
//
// Reference mechanics (discarded): one thread per centroid scanning all N (assignment, prev)
// pairs with a serial fp32 Kahan chain.  Here:
//   move events      key 2*cur if the row moved in, 2*prev+1 if it moved out.  Late iterations move a
//                    few percent of the rows, so nothing here is sized by N except one read of
//                    (prev, cur):
//     direct         every event's row dropped into its key's bucket through an atomic cursor
//                    (move_scatter), then ONE kernel with a block per centroid (cluster_sums): both
//                    buckets sorted ascending in LDS -- the set is exact, the order fixed again --
//                    summed, folded.  No host read; an overflowing bucket is rebuilt in order from
//                    (prev, cur) by that block
//     radix sort     (the first iterations: most rows move) events compacted in row order (count per
//                    1024-row block, scan, write) and sorted stably by key (rocprim); the same
//                    cluster_sums kernel reads its segments
//                    -- both give per (cluster, sign) lists with rows ascending, bit-identical sums
//   cluster_sums     fp64 column sums of each list in a fixed association (only MOVED rows are touched),
//                    delta[c] = sum_in - sum_out, dcount[c] = n_in - n_out, the reduce buffer's tail
//   [row-sharded multi-GPU: all-reduce of delta / dcount happens here]
//   apply_delta      the formula above in fp64, rounded once to fp32; the stop rule on the device
// Centroids agree with the reference's fp32 Kahan chain to a few ulp (tests: 2e-6 relative);
// they are bit-reproducible run to run and across the two paths.  An empty cluster yields
// a non-finite row that is never chosen again, as in the reference (kmeans.cu:425-426).
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>

#include <rocprim/device/device_radix_sort.hpp>

#include "exact.hpp"
#include "kernels.hpp"

namespace kmx {

static unsigned bits_for(uint64_t maxval) {
  unsigned b = 1;
  while (b < 32 && (1ull << b) <= maxval) b++;
  return b;
}

// ---- sorting helpers --------------------------------------------------------------------
size_t sort_temp_bytes(size_t n, uint32_t max_key) {
  size_t bytes = 0;
  (void)rocprim::radix_sort_pairs(nullptr, bytes, (const uint32_t *)nullptr, (uint32_t *)nullptr,
                                  (const uint32_t *)nullptr, (uint32_t *)nullptr, n, 0u, bits_for(max_key),
                                  (hipStream_t)0);
  return bytes;
}

// offsets[k] = first index with key >= k, k = 0..nkeys  (nkeys+1 entries)
__global__ void offsets_kernel(const uint32_t *__restrict__ keys_sorted, uint32_t n, uint32_t nkeys,
                               uint32_t *__restrict__ offsets) {
  const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k > nkeys) return;
  uint32_t lo = 0, hi = n;
  while (lo < hi) {
    const uint32_t mid = lo + (hi - lo) / 2;
    if (keys_sorted[mid] < k) lo = mid + 1; else hi = mid;
  }
  offsets[k] = lo;
}

// ---- inverse assignments (k-NN's CSR, kmcuda.cc:648-691) ----------------------------------
__global__ void cluster_keys_kernel(const uint32_t *__restrict__ assignments, uint32_t N, uint32_t K,
                                    uint32_t *__restrict__ keys, uint32_t *__restrict__ vals) {
  const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= N) return;
  const uint32_t a = assignments[s];
  keys[s] = a < K ? a : K;  // K = "no cluster" (NaN sample or never assigned)
  vals[s] = s;
}

hipError_t launch_inverse_assignments(const uint32_t *assignments, uint32_t N, uint32_t K, uint32_t *keys_tmp,
                                      uint32_t *vals_tmp, uint32_t *keys_sorted, uint32_t *inv,
                                      uint32_t *offsets, void *temp, size_t temp_bytes, hipStream_t st) {
  hipLaunchKernelGGL(cluster_keys_kernel, dim3((N + 255) / 256), dim3(256), 0, st, assignments, N, K, keys_tmp,
                     vals_tmp);
  hipError_t e = rocprim::radix_sort_pairs(temp, temp_bytes, (const uint32_t *)keys_tmp, keys_sorted,
                                           (const uint32_t *)vals_tmp, inv, (size_t)N, 0u, bits_for(K), st);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(offsets_kernel, dim3((K + 1 + 255) / 256), dim3(256), 0, st, keys_sorted, N, K, offsets);
  return hipGetLastError();
}

// ---- k-NN: the order in which a cluster's queries are grouped into blocks -------------------------------------
// Every query of the f16 search (knn_f16.hip) decides per cluster whether to visit it, but a tile of candidates is
// scored by a whole wave (64 queries) and staged for a whole block (256) if ANY of them visits: queries taken in
// sorted-position order -- sample order inside a cluster, i.e. spatially random -- make the union of 64 visit sets a
// third larger than one set (profiles/r4a_*: 1.98e12 pairs scored for 1.44e12 wanted).  Queries that lie on the same
// side of their cluster want the same clusters: key = (own cluster, the OTHER cluster whose members can come closest,
// argmin_c lb[c][q]); sorting the positions by it groups them.  Only the grouping of queries into waves changes:
// every query still scans all clusters in the reference's order against its own heap.
// mode 1: the key above.  mode 2: key = (own cluster, the query's distance to its own centroid, 15 bits of the float) --
// under the reference's prune rule (C[c][mine] - d(q, c_mine) - R[c] > kth) a cluster's queries visit NESTED sets of
// clusters, the farther from their centroid the more.  mode 3: (own cluster, argmin as mode 1, 6 bits of that distance
// relative to the cluster's radius).  mydist: per sorted position; R: per cluster.
__global__ void knn_query_keys_kernel(const float *__restrict__ lb, size_t stride, const uint32_t *__restrict__ offsets,
                                      uint32_t K, uint32_t p_base, uint32_t p_end, uint32_t *__restrict__ keys,
                                      uint32_t *__restrict__ vals, int mode, const float *__restrict__ mydist,
                                      const float *__restrict__ R) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= p_end - p_base) return;
  const uint32_t p = p_base + i;
  uint32_t lo = 0, hi = K;   // the cluster of sorted position p: offsets[cls] <= p < offsets[cls + 1]
  while (hi - lo > 1) {
    const uint32_t mid = (lo + hi) / 2;
    if (offsets[mid] <= p) lo = mid; else hi = mid;
  }
  const uint32_t cls = lo;
  vals[i] = p;
  if (mode == 2) {
    const float md = mydist[p];
    const uint32_t q = (md == md && md > 0.f) ? ((__float_as_uint(md) >> 16) & 0x7FFFu) : 0u;
    keys[i] = (cls << 15) | q;
    return;
  }
  float best = INFINITY;
  uint32_t arg = 0;
  for (uint32_t c = 0; c < K; c++) {
    const float v = lb[(size_t)c * stride + i];   // coalesced across the block's queries
    if (c != cls && v < best) { best = v; arg = c; }
  }
  if (mode == 3) {
    const float r = R[cls], md = mydist[p];
    uint32_t q = 0;
    if (r > 0.f && md == md) q = (uint32_t)fminf(63.f, fmaxf(0.f, md / r * 63.f));
    keys[i] = ((cls * K + arg) << 6) | q;
    return;
  }
  keys[i] = cls * K + arg;
}

// qperm[i] = the sorted position of the query that slot p_base + i of the block plan handles.  false: not possible
// here (the key does not fit 32 bits, or the sort's scratch is too small): the caller keeps the identity order
bool launch_knn_query_order(const float *lb, size_t stride, const uint32_t *offsets, uint32_t K, uint32_t p_base,
                            uint32_t p_end, uint32_t *keys_tmp, uint32_t *vals_tmp, uint32_t *keys_sorted,
                            uint32_t *qperm, void *temp, size_t temp_bytes, hipStream_t st, int mode,
                            const float *mydist, const float *R) {
  if (p_end <= p_base || K > 65535u) return false;
  const uint64_t key_max = mode == 2 ? ((uint64_t)K << 15) : (mode == 3 ? ((uint64_t)K * K) << 6 : (uint64_t)K * K);
  if (key_max > 0xFFFFFFFFull) return false;
  const uint32_t n = p_end - p_base;
  size_t need = 0;
  if (rocprim::radix_sort_pairs(nullptr, need, (const uint32_t *)nullptr, (uint32_t *)nullptr, (const uint32_t *)nullptr,
                                (uint32_t *)nullptr, (size_t)n, 0u, bits_for(key_max), st) != hipSuccess ||
      need > temp_bytes)
    return false;
  hipLaunchKernelGGL(knn_query_keys_kernel, dim3((n + 255) / 256), dim3(256), 0, st, lb, stride, offsets, K, p_base,
                     p_end, keys_tmp, vals_tmp, mode, mydist, R);
  size_t bytes = temp_bytes;
  return rocprim::radix_sort_pairs(temp, bytes, (const uint32_t *)keys_tmp, keys_sorted, (const uint32_t *)vals_tmp,
                                   qperm, (size_t)n, 0u, bits_for(key_max), st) == hipSuccess &&
         hipGetLastError() == hipSuccess;
}

// ---- move events --------------------------------------------------------------------------
// uncompacted form (two slots per row, sentinel keys): the strict-parity update sorts all of them
__global__ void move_events_kernel(const uint32_t *__restrict__ prev, const uint32_t *__restrict__ cur, uint32_t N,
                                   uint32_t K, uint32_t *__restrict__ keys, uint32_t *__restrict__ vals) {
  const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= N) return;
  const uint32_t p = prev[s], a = cur[s];
  const bool moved = p != a;
  const uint32_t sentinel = 2u * K;
  keys[2 * (size_t)s] = (moved && a < K) ? 2u * a : sentinel;
  keys[2 * (size_t)s + 1] = (moved && p < K) ? 2u * p + 1u : sentinel;
  vals[2 * (size_t)s] = s;
  vals[2 * (size_t)s + 1] = s;
}

constexpr uint32_t kMoveRows = 1024;  // rows per block of the event compaction (256 threads x 4)

__device__ __forceinline__ void move_flags(const uint32_t *__restrict__ prev, const uint32_t *__restrict__ cur,
                                           uint32_t s, uint32_t N, uint32_t K, uint32_t &p, uint32_t &a, bool &ein,
                                           bool &eout) {
  p = a = 0;
  ein = eout = false;
  if (s < N) {
    p = prev[s];
    a = cur[s];
    const bool moved = p != a;
    ein = moved && a < K;    // joins cluster a
    eout = moved && p < K;   // leaves cluster p (p >= K: it had no cluster, kmeans.cu:395-403)
  }
}

__global__ __launch_bounds__(256) void move_count_kernel(const uint32_t *__restrict__ prev,
                                                         const uint32_t *__restrict__ cur, uint32_t N, uint32_t K,
                                                         uint32_t *__restrict__ blockcnt) {
  uint32_t cnt = 0;
#pragma unroll
  for (uint32_t i = 0; i < kMoveRows / 256; i++) {
    uint32_t p, a;
    bool ein, eout;
    move_flags(prev, cur, blockIdx.x * kMoveRows + i * 256 + threadIdx.x, N, K, p, a, ein, eout);
    cnt += (uint32_t)ein + (uint32_t)eout;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o);
  __shared__ uint32_t w[4];
  if ((threadIdx.x & 63) == 0) w[threadIdx.x >> 6] = cnt;
  __syncthreads();
  if (threadIdx.x == 0) blockcnt[blockIdx.x] = w[0] + w[1] + w[2] + w[3];
}

// counts -> exclusive offsets in place, total -> total[0]; one block (nb = N / 1024 entries)
__global__ __launch_bounds__(1024) void move_scan_kernel(uint32_t *__restrict__ blockcnt, uint32_t nb,
                                                         uint32_t *__restrict__ total) {
  __shared__ uint32_t wsum[16];
  __shared__ uint32_t carry_s;
  const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (threadIdx.x == 0) carry_s = 0;
  __syncthreads();
  for (uint32_t base = 0; base < nb; base += 1024) {
    const uint32_t i = base + threadIdx.x;
    const uint32_t v = i < nb ? blockcnt[i] : 0u;
    uint32_t inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const uint32_t t = __shfl_up(inc, o);
      if ((int)lane >= o) inc += t;
    }
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    uint32_t wbase = 0, all = 0;
#pragma unroll
    for (uint32_t k = 0; k < 16; k++) {
      const uint32_t t = wsum[k];
      if (k < wave) wbase += t;
      all += t;
    }
    const uint32_t carry = carry_s;
    if (i < nb) blockcnt[i] = carry + wbase + inc - v;
    __syncthreads();
    if (threadIdx.x == 0) carry_s = carry + all;
    __syncthreads();
  }
  if (threadIdx.x == 0) total[0] = carry_s;
}

// events in row order, per row (in, out): exactly the order the stable sort needs
__global__ __launch_bounds__(256) void move_write_kernel(const uint32_t *__restrict__ prev,
                                                         const uint32_t *__restrict__ cur, uint32_t N, uint32_t K,
                                                         const uint32_t *__restrict__ blockoff,
                                                         uint32_t *__restrict__ keys, uint32_t *__restrict__ vals) {
  __shared__ uint32_t wcnt[4];
  const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint32_t off = blockoff[blockIdx.x];
#pragma unroll
  for (uint32_t i = 0; i < kMoveRows / 256; i++) {
    const uint32_t s = blockIdx.x * kMoveRows + i * 256 + threadIdx.x;
    uint32_t p, a;
    bool ein, eout;
    move_flags(prev, cur, s, N, K, p, a, ein, eout);
    const unsigned long long min = __ballot(ein), mout = __ballot(eout);
    const unsigned long long lower = (1ull << lane) - 1ull;
    const uint32_t rank = (uint32_t)__popcll(min & lower) + (uint32_t)__popcll(mout & lower);
    if (lane == 0) wcnt[wave] = (uint32_t)__popcll(min) + (uint32_t)__popcll(mout);
    __syncthreads();
    uint32_t wbase = 0, all = 0;
#pragma unroll
    for (uint32_t k = 0; k < 4; k++) {
      if (k < wave) wbase += wcnt[k];
      all += wcnt[k];
    }
    const size_t at = (size_t)off + wbase + rank;
    if (ein) { keys[at] = 2u * a; vals[at] = s; }
    if (eout) { keys[at + (ein ? 1 : 0)] = 2u * p + 1u; vals[at + (ein ? 1 : 0)] = s; }
    off += all;
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------------------
// cluster_sums: ONE kernel per update, one block per centroid -- the ordered move lists of the centroid (rows
// that joined it, rows that left it), their fp64 column sums, delta = in - out, the count change, and the
// fused reduce buffer's tail.  The lists come from one of two places:
//   DIRECT   move_scatter_kernel has dropped every event's row into its key's bucket (key = 2 c + sign, `cap`
//            slots) through an atomic cursor: arbitrary positions, exact set.  The block sorts the bucket
//            ascending in LDS -- the order is fixed again.  A bucket that overflowed (the cursor counts every
//            attempt) is rebuilt from (prev, cur) themselves by an ordered compaction over all N rows: O(N) per
//            such bucket, correct for any input, so the host never has to look at the counts first.
//   SORTED   the radix path's stably sorted events (first iterations: most rows move).
// Either way a list is ascending by row and the summation below depends on nothing else: the update is
// bit-reproducible across the two paths, the host's choice of path, and run to run.
//
// Summation of a list of n rows (threads = groups x fl, fl lanes across the features): group g takes the
// contiguous rows [g * ceil(n / groups), ...), eight independent accumulators by row position, folded as
// ((a0 + a1) + (a2 + a3)) + ((a4 + a5) + (a6 + a7)); the groups' sums are added in ascending g.
// ---------------------------------------------------------------------------------------
constexpr uint32_t kSumThreads = 1024;
constexpr uint32_t kBucketCapMax = 4096;   // rows an LDS sort holds (two lists: 32 KB)

uint32_t move_bucket_cap(uint32_t N, uint32_t K) {
  // about the rows per centroid, as a power of two in [32, 4096]: steady iterations move a few percent of them
  uint32_t cap = 32;
  while (cap < kBucketCapMax && (uint64_t)cap * 2 * K <= (uint64_t)N) cap <<= 1;
  return cap;
}

struct SumArgs {
  const float *samples;
  uint32_t N, D, K;
  const uint32_t *prev, *cur;           // DIRECT: the overflow fallback re-reads them
  uint32_t *cursors;                    // DIRECT: 2 K event counts, `stride` words apart; left zero
  uint32_t stride;
  const uint32_t *bucket_rows;          // DIRECT: 2 K x cap
  uint32_t cap;
  uint32_t *overflow;                   // DIRECT: 2 N words for rebuilt lists
  const uint32_t *rows_sorted, *offsets2;   // SORTED
  double *delta;
  int32_t *dcount;                      // may be null
  double *tail;                         // may be null: [dcount K | counters 4] behind delta
  const uint32_t *counters;
  uint32_t *res;                        // device: [0] events, [1] largest list, [2] ticket, [3] overflow cursor
  uint32_t *host;                       // pinned, device-visible: [0] events, [1] largest list, [2] counters[4], [4] counters[kDuoCount]
};

__device__ __forceinline__ double list_sum(const float *__restrict__ samples, uint32_t D, const uint32_t *rows,
                                           uint32_t r0, uint32_t r1, uint32_t f) {
  double a[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  uint32_t r = r0;
  for (; r + 8 <= r1; r += 8) {
    float x[8];
#pragma unroll
    for (int j = 0; j < 8; j++) x[j] = samples[(size_t)rows[r + j] * D + f];
#pragma unroll
    for (int j = 0; j < 8; j++) a[j] += (double)x[j];
  }
  for (uint32_t j = 0; r < r1; r++, j++) a[j] += (double)samples[(size_t)rows[r] * D + f];
  return ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
}

// the same for four consecutive features per lane (16-byte loads; D % 4 == 0, aligned rows): four accumulator
// sets by row position, folded as (a0 + a1) + (a2 + a3)
__device__ __forceinline__ void list_sum4(const float *__restrict__ samples, uint32_t D, const uint32_t *rows,
                                          uint32_t r0, uint32_t r1, uint32_t f, double (&out)[4]) {
  double a[4][4];
#pragma unroll
  for (int j = 0; j < 4; j++)
#pragma unroll
    for (int e = 0; e < 4; e++) a[j][e] = 0.0;
  uint32_t r = r0;
  for (; r + 4 <= r1; r += 4) {
    float4 x[4];
#pragma unroll
    for (int j = 0; j < 4; j++) x[j] = *reinterpret_cast<const float4 *>(samples + (size_t)rows[r + j] * D + f);
#pragma unroll
    for (int j = 0; j < 4; j++) {
      a[j][0] += (double)x[j].x; a[j][1] += (double)x[j].y; a[j][2] += (double)x[j].z; a[j][3] += (double)x[j].w;
    }
  }
  for (uint32_t j = 0; r < r1; r++, j++) {
    const float4 x = *reinterpret_cast<const float4 *>(samples + (size_t)rows[r] * D + f);
    // (j < 4: static indexing after unrolling)
    if (j == 0) { a[0][0] += (double)x.x; a[0][1] += (double)x.y; a[0][2] += (double)x.z; a[0][3] += (double)x.w; }
    else if (j == 1) { a[1][0] += (double)x.x; a[1][1] += (double)x.y; a[1][2] += (double)x.z; a[1][3] += (double)x.w; }
    else { a[2][0] += (double)x.x; a[2][1] += (double)x.y; a[2][2] += (double)x.z; a[2][3] += (double)x.w; }
  }
#pragma unroll
  for (int e = 0; e < 4; e++) out[e] = (a[0][e] + a[1][e]) + (a[2][e] + a[3][e]);
}

// VEC4: lanes own four consecutive features each (16-byte loads), so a 1-KB row takes 64 lanes and the block's
// 1024 threads are 16 row groups instead of 4: the lists of a steady iteration (tens to hundreds of rows) are
// read in one or two trips.  D % 4 == 0 and 16-byte aligned rows; anything else takes the scalar mapping.
template <bool DIRECT, bool VEC4>
__global__ __launch_bounds__(kSumThreads) void cluster_sums_kernel(SumArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char sums_lds[];
  double *part = reinterpret_cast<double *>(sums_lds);                               // VEC4: 4 per thread, else 1
  uint32_t (*srt)[kBucketCapMax] = reinterpret_cast<uint32_t (*)[kBucketCapMax]>(sums_lds + kSumThreads * 4 * sizeof(double));
  uint32_t *part_u = reinterpret_cast<uint32_t *>(part);   // scratch of the rank sort (before the sums use `part`)
  __shared__ uint32_t wcnt[16];
  __shared__ uint32_t sh_base;
  const uint32_t c = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const uint32_t D = a.D;
  const uint32_t *rows[2];
  uint32_t n[2];
  if (DIRECT) {
    n[0] = a.cursors[(size_t)(2 * c) * a.stride];
    n[1] = a.cursors[(size_t)(2 * c + 1) * a.stride];
    __syncthreads();   // everybody has read the counts
    if (tid < 2) a.cursors[(size_t)(2 * c + tid) * a.stride] = 0u;   // zero again for the next update
#pragma unroll
    for (int sg = 0; sg < 2; sg++) {
      const uint32_t cnt = n[sg], key = 2 * c + sg;
      if (cnt <= kSumThreads && cnt <= a.cap) {
        // the usual case, a short list: every thread ranks its own row (rows are distinct) -- two barriers
        // instead of the bitonic network's log^2
        uint32_t *v = srt[sg];
        const uint32_t mine = tid < cnt ? a.bucket_rows[(size_t)key * a.cap + tid] : 0xFFFFFFFFu;
        if (tid < cnt) part_u[tid] = mine;
        __syncthreads();
        if (tid < cnt) {
          uint32_t rank = 0;
          for (uint32_t j = 0; j < cnt; j++) rank += part_u[j] < mine ? 1u : 0u;
          v[rank] = mine;
        }
        __syncthreads();
        rows[sg] = v;
      } else if (cnt <= a.cap) {
        // bitonic sort of the bucket, padded to a power of two with 0xFFFFFFFF
        uint32_t m = 2;
        while (m < cnt) m <<= 1;
        uint32_t *v = srt[sg];
        for (uint32_t i = tid; i < m; i += kSumThreads) v[i] = i < cnt ? a.bucket_rows[(size_t)key * a.cap + i] : 0xFFFFFFFFu;
        __syncthreads();
        if (cnt > 1) {
          for (uint32_t k = 2; k <= m; k <<= 1) {
            for (uint32_t j = k >> 1; j > 0; j >>= 1) {
              for (uint32_t i = tid; i < m; i += kSumThreads) {
                const uint32_t l = i ^ j;
                if (l > i) {
                  const uint32_t x = v[i], y = v[l];
                  const bool up = (i & k) == 0;
                  if ((x > y) == up) { v[i] = y; v[l] = x; }
                }
              }
              __syncthreads();
            }
          }
        }
        rows[sg] = v;
      } else {
        // the bucket overflowed: rebuild the list, ascending, from (prev, cur) -- 4 consecutive rows per thread
        if (tid == 0) sh_base = atomicAdd(&a.res[3], cnt);
        __syncthreads();
        uint32_t *dst = a.overflow + sh_base;
        uint32_t done = 0;
        for (uint32_t base = 0; base < a.N; base += 4 * kSumThreads) {
          uint32_t hit[4], cntme = 0;
#pragma unroll
          for (uint32_t q = 0; q < 4; q++) {
            const uint32_t s = base + tid * 4 + q;
            uint32_t p, cu;
            bool ein, eout;
            move_flags(a.prev, a.cur, s, a.N, a.K, p, cu, ein, eout);
            const bool mine = sg == 0 ? (ein && cu == c) : (eout && p == c);
            hit[q] = mine ? s : 0xFFFFFFFFu;
            cntme += mine ? 1u : 0u;
          }
          uint32_t inc = cntme;   // inclusive scan over the wave
#pragma unroll
          for (int o = 1; o < 64; o <<= 1) {
            const uint32_t t = __shfl_up(inc, o);
            if ((int)lane >= o) inc += t;
          }
          if (lane == 63) wcnt[wave] = inc;
          __syncthreads();
          uint32_t wbase = 0, all = 0;
#pragma unroll
          for (uint32_t k = 0; k < 16; k++) {
            const uint32_t t = wcnt[k];
            if (k < wave) wbase += t;
            all += t;
          }
          uint32_t at = done + wbase + inc - cntme;
#pragma unroll
          for (uint32_t q = 0; q < 4; q++)
            if (hit[q] != 0xFFFFFFFFu) dst[at++] = hit[q];
          done += all;
          __syncthreads();
        }
        __threadfence_block();
        rows[sg] = dst;
      }
    }
    __syncthreads();
  } else {
    const uint32_t o0 = a.offsets2[2 * c], o1 = a.offsets2[2 * c + 1], o2 = a.offsets2[2 * c + 2];
    n[0] = o1 - o0; n[1] = o2 - o1;
    rows[0] = a.rows_sorted + o0;
    rows[1] = a.rows_sorted + o1;
  }

  // ---- sums: fl lanes across the features, groups of rows ----
  // Two mappings, chosen by the centroid's longer list (a function of the lists alone: the same on every path).
  // Long lists (the first iterations; steady iterations of a many-row shard): a lane owns four consecutive
  // features, 16 row groups, 16-byte loads -- 2.2 against 3.5 ms for the first updates of an 8M-row run.
  // Short lists: a lane owns one feature, 4 row groups -- 7 us less at 1M rows, where the sixteen-group fold
  // costs more than the loads it saves (profiles/r3i_*).
  if (VEC4 && max(n[0], n[1]) >= 128u) {
    uint32_t fl = 16;   // lanes per row: a power of two covering D / 4, at most 256
    while (fl < 256u && fl * 4u < D) fl <<= 1;
    const uint32_t groups = kSumThreads / fl, g = tid / fl, lf = tid % fl;
    for (uint32_t f0 = 0; f0 < D; f0 += 4u * fl) {
      const uint32_t f = f0 + 4u * lf;
      const bool fv = f < D;
      double tot[2][4];
#pragma unroll
      for (int sg = 0; sg < 2; sg++) {
        const uint32_t cnt = n[sg], chunk = (cnt + groups - 1) / groups;
        const uint32_t r0 = min(cnt, g * chunk), r1 = min(cnt, r0 + chunk);
        double mine[4] = {0.0, 0.0, 0.0, 0.0};
        if (fv && r1 > r0) list_sum4(a.samples, D, rows[sg], r0, r1, f, mine);
#pragma unroll
        for (int e = 0; e < 4; e++) part[(size_t)tid * 4 + e] = mine[e];
        __syncthreads();
        if (g == 0) {
#pragma unroll
          for (int e = 0; e < 4; e++) {
            double t = part[(size_t)lf * 4 + e];
            for (uint32_t gg = 1; gg < groups; gg++) t += part[(size_t)(gg * fl + lf) * 4 + e];
            tot[sg][e] = t;
          }
        }
        __syncthreads();
      }
      if (g == 0 && fv) {
#pragma unroll
        for (int e = 0; e < 4; e++) a.delta[(size_t)c * D + f + e] = tot[0][e] - tot[1][e];
      }
    }
  } else {
  const uint32_t fl = D > 128 ? 256u : (D > 64 ? 128u : 64u), groups = kSumThreads / fl;
  const uint32_t g = tid / fl, lf = tid % fl;
  for (uint32_t f0 = 0; f0 < D; f0 += fl) {
    const uint32_t f = f0 + lf;
    const bool fv = f < D;
    double tot[2] = {0, 0};
#pragma unroll
    for (int sg = 0; sg < 2; sg++) {
      const uint32_t cnt = n[sg], chunk = (cnt + groups - 1) / groups;
      const uint32_t r0 = min(cnt, g * chunk), r1 = min(cnt, r0 + chunk);
      part[tid] = (fv && r1 > r0) ? list_sum(a.samples, D, rows[sg], r0, r1, f) : 0.0;
      __syncthreads();
      if (g == 0) {
        double t = part[lf];
        for (uint32_t gg = 1; gg < groups; gg++) t += part[gg * fl + lf];
        tot[sg] = t;
      }
      __syncthreads();
    }
    if (g == 0 && fv) a.delta[(size_t)c * D + f] = tot[0] - tot[1];
  }
  }
  if (tid == 0) {
    const int32_t dc = (int32_t)n[0] - (int32_t)n[1];
    if (a.dcount) a.dcount[c] = dc;
    if (a.tail) a.tail[c] = (double)dc;
    atomicAdd(&a.res[0], n[0] + n[1]);
    atomicMax(&a.res[1], max(n[0], n[1]));
    __threadfence();
    const uint32_t ticket = atomicAdd(&a.res[2], 1u);
    if (ticket == gridDim.x - 1) {   // the last block: report to the host's pinned words, reset for the next update
      // (not from a pass behind the device-side stop: it touched nothing, and its zeros would size the NEXT run's
      //  first update and stage 2 -- the default schedule's first pass behind the hand-over point ran stage 2 on a
      //  64-block grid, 2.16 ms instead of 0.47 at 4M rows, profiles/r5af_*)
      if (a.counters[kStopFlag] == 0u) {
        volatile uint32_t *h = a.host;
        h[0] = atomicAdd(&a.res[0], 0u);
        h[1] = atomicAdd(&a.res[1], 0u);
        h[2] = a.counters[4];
        h[4] = a.counters[kDuoCount];   // (engine.cpp: whether the duo list pays is judged by both lengths)
      }
      a.res[0] = 0u; a.res[1] = 0u; a.res[2] = 0u; a.res[3] = 0u;
    }
  }
  if (a.tail && c == 0 && tid < 4) a.tail[a.K + tid] = (double)a.counters[tid];
}

// counters one cache line apart while that stays under ~1 MB per array
uint32_t move_bucket_stride(uint32_t K) {
  uint32_t s = 32;
  while (s > 1 && 2ull * K * s > 262144ull) s >>= 1;
  return s;
}
// cursors (2 K, `stride` apart) + 4 result words; zero-initialised by the owner, left zero by every update
size_t move_bucket_words(uint32_t K) { return 2 * (size_t)K * move_bucket_stride(K) + 4; }

// every event's row into its key's bucket; the cursor counts every attempt, also beyond the capacity
// (counters `stride` words apart: packed, the 2K counters of K = 1024 share 64 cache lines and the L2
// serialises the atomics per line -- 118 us for 6e5 events, one line each ~20 us)
__global__ __launch_bounds__(256) void move_scatter_kernel(const uint32_t *__restrict__ prev,
                                                           const uint32_t *__restrict__ cur, uint32_t N, uint32_t K,
                                                           uint32_t *__restrict__ cursors, uint32_t stride,
                                                           uint32_t *__restrict__ bucket_rows, uint32_t cap) {
  const uint32_t s = blockIdx.x * 256u + threadIdx.x;
  uint32_t p, a;
  bool ein, eout;
  move_flags(prev, cur, s, N, K, p, a, ein, eout);
  if (ein) {
    const uint32_t at = atomicAdd(&cursors[(size_t)(2u * a) * stride], 1u);
    if (at < cap) bucket_rows[(size_t)(2u * a) * cap + at] = s;
  }
  if (eout) {
    const uint32_t at = atomicAdd(&cursors[(size_t)(2u * p + 1u) * stride], 1u);
    if (at < cap) bucket_rows[(size_t)(2u * p + 1u) * cap + at] = s;
  }
}

// The host side of the update.  Two ways to the ordered move lists, bit-identical sums either way:
//   radix    the first iterations (most rows move: the buckets would overflow): compaction + rocprim sort
//            sized by the event count -> ONE host read
//   direct   everything else: scatter + cluster_sums, NOTHING read.  The kernel reports (events, largest
//            list) to pinned words; the host looks at whatever has landed -- an earlier update's figures --
//            only to pick the path of LATER updates.  Nothing but speed depends on that choice.
hipError_t launch_move_deltas(const float *samples, uint32_t N, uint32_t D, uint32_t K, const uint32_t *prev,
                              const uint32_t *cur, uint32_t *keys_tmp, uint32_t *vals_tmp, uint32_t *keys_sorted,
                              uint32_t *rows_sorted, uint32_t *offsets2, void *temp, size_t temp_bytes,
                              uint32_t *bucket_rows, uint32_t cap, double *delta, int32_t *dcount, double *tail,
                              const uint32_t *counters, uint32_t *blockoff, uint32_t *bucket_work, MoveState *ms,
                              hipStream_t st) {
  // blockoff: N / 1024 + 2 words; bucket_work: move_bucket_words(K) words, zero on entry and on exit;
  // ms->host / host_dev: 4 pinned words
  const uint32_t stride = move_bucket_stride(K);
  SumArgs a;
  a.samples = samples; a.N = N; a.D = D; a.K = K; a.prev = prev; a.cur = cur;
  a.cursors = bucket_work; a.stride = stride; a.bucket_rows = bucket_rows; a.cap = cap;
  a.overflow = rows_sorted; a.rows_sorted = rows_sorted; a.offsets2 = offsets2;
  a.delta = delta; a.dcount = dcount; a.tail = tail; a.counters = counters;
  a.res = bucket_work + 2 * (size_t)K * stride; a.host = ms->host_dev;
  hipError_t e = hipSuccess;
  // (the mapping depends on D and the rows' alignment only, never on the path: the two paths stay bit-identical)
  const bool vec4 = (D & 3u) == 0 && (((uintptr_t)samples) & 15u) == 0;
  const size_t sums_lds = kSumThreads * 4 * sizeof(double) + 2 * kBucketCapMax * sizeof(uint32_t);   // 64 KB
  auto launch_sums = [&](bool direct) {
    const void *fn = direct ? (vec4 ? (const void *)cluster_sums_kernel<true, true> : (const void *)cluster_sums_kernel<true, false>)
                            : (vec4 ? (const void *)cluster_sums_kernel<false, true> : (const void *)cluster_sums_kernel<false, false>);
    // (per launch: the attribute belongs to the current device's copy of the kernel)
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sums_lds) != hipSuccess) return;
    if (direct) {
      if (vec4) hipLaunchKernelGGL((cluster_sums_kernel<true, true>), dim3(K), dim3(kSumThreads), sums_lds, st, a);
      else hipLaunchKernelGGL((cluster_sums_kernel<true, false>), dim3(K), dim3(kSumThreads), sums_lds, st, a);
    } else {
      if (vec4) hipLaunchKernelGGL((cluster_sums_kernel<false, true>), dim3(K), dim3(kSumThreads), sums_lds, st, a);
      else hipLaunchKernelGGL((cluster_sums_kernel<false, false>), dim3(K), dim3(kSumThreads), sums_lds, st, a);
    }
  };
  const bool force_radix = ms->force == 1, force_sync = ms->force == 2, force_direct = ms->force == 3;
  // last_events / host[1]: the newest event count and largest list the host knows (2 N before the first call).
  // Lists shrink from iteration to iteration, so an older figure errs towards the radix path, which takes any
  // size; after a radix call only the count is known yet and the largest list is taken as twice the average.
  // A list beyond the bucket capacity would still be summed correctly (rebuilt from (prev, cur)), only slowly.
  const uint32_t est_max = std::max((uint32_t)ms->host[1] /* whatever has landed */, ms->last_events / K);
  const bool radix = N != 0 && !force_direct &&
                     (force_radix || ms->last_events > N / 2 || est_max > cap - cap / 4);
  if (N == 0) {
    e = hipMemsetAsync(offsets2, 0, (2 * (size_t)K + 1) * sizeof(uint32_t), st);
    if (e != hipSuccess) return e;
    launch_sums(false);
  } else if (radix) {
    ms->n_radix++;
    const uint32_t nb = (N + kMoveRows - 1) / kMoveRows;
    hipLaunchKernelGGL(move_count_kernel, dim3(nb), dim3(256), 0, st, prev, cur, N, K, blockoff);
    hipLaunchKernelGGL(move_scan_kernel, dim3(1), dim3(1024), 0, st, blockoff, nb, blockoff + nb + 1);
    hipLaunchKernelGGL(move_write_kernel, dim3(nb), dim3(256), 0, st, prev, cur, N, K, blockoff, keys_tmp, vals_tmp);
    e = hipGetLastError();
    if (e != hipSuccess) return e;
    e = hipMemcpyAsync(ms->host + 3, blockoff + nb + 1, sizeof(uint32_t), hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) return e;
    const uint32_t m = ms->host[3];
    if (m) {
      e = rocprim::radix_sort_pairs(temp, temp_bytes, (const uint32_t *)keys_tmp, keys_sorted,
                                    (const uint32_t *)vals_tmp, rows_sorted, (size_t)m, 0u, bits_for(2ull * K), st);
      if (e != hipSuccess) return e;
    }
    // segment starts by binary search in the sorted keys
    hipLaunchKernelGGL(offsets_kernel, dim3((2 * K + 1 + 255) / 256), dim3(256), 0, st, keys_sorted, m, 2 * K, offsets2);
    ms->last_events = m;
    // (written BEFORE the kernel that reports is launched -- the stream is idle here, nothing else writes these words --
    //  so that its report, whenever it lands, is the last word)
    ms->host[0] = m;    // (the kernel's report will say the same)
    // (the largest list: whatever an EARLIER update has reported stays in host[1] until this one's report lands --
    //  lists shrink from iteration to iteration, so the older figure errs towards this path.  Round 4 zeroed the word
    //  here and took twice the average: the pass enqueued behind the stop at the hand-over point repeats that
    //  iteration's events, and on skewed clusters -- a 4M-row mixture after three iterations -- the direct path then
    //  rebuilt dozens of overflowing buckets from all N rows: 5.3 ms of a 27-ms call, profiles/r5g_handover_*.  A
    //  marker "not known yet" written here instead kept every later update on this path -- the host runs one pass ahead
    //  of the reports: kmeans_cuda's 46-iteration loop 0.28-0.40 s instead of 0.21, profiles/r5k_bench_api_*.)
    launch_sums(false);
  } else {
    ms->n_direct++;
    hipLaunchKernelGGL(move_scatter_kernel, dim3((N + 255) / 256), dim3(256), 0, st, prev, cur, N, K, bucket_work,
                       stride, bucket_rows, cap);
    launch_sums(true);
    e = hipGetLastError();
    if (e != hipSuccess) return e;
    if (force_sync) {   // KMCUDA_AMD_UPDATE=sync: the next call decides on THIS call's figures
      e = hipStreamSynchronize(st);
      if (e != hipSuccess) return e;
    }
    ms->last_events = ms->host[0];   // whatever has landed: an earlier update's figures
  }
  return hipGetLastError();
}

// StopCtl (kernels.hpp): the reference's stop rule -- check_changed, kmeans.cu:697-717, evaluated BEFORE the
// update -- decided here, on the device, from the reduced reassignment count behind dcount_d: every block reads
// the same word and takes the same branch.  Stopping: nothing is modified, the flag is raised (the kernels that
// assign every row return at once from then on, so whatever the host has enqueued past this point leaves the
// state as the reference returns it: assignments of this iteration, centroids one update behind).  Going on:
// counters[0] is zeroed for the next pass (kmeans.cu:710-714).  The host learns the outcome from the pinned
// words, without a stream synchronisation in front of the next pass.
template <int METRIC>
__global__ void apply_delta_kernel(const double *__restrict__ delta, const int32_t *__restrict__ dcount,
                                   const double *__restrict__ dcount_d, uint32_t D, float *__restrict__ centroids,
                                   uint32_t *__restrict__ ccounts, StopCtl ctl) {
  const uint32_t c = blockIdx.x;
  if (ctl.counters) {
    bool stop = ctl.counters[kStopFlag] != 0u;   // once stopped, stay stopped
    if (ctl.threshold >= 0.f && dcount_d) stop = stop || (float)(uint32_t)dcount_d[gridDim.x] <= ctl.threshold;
    if (c == 0 && threadIdx.x == 0) {
      if (stop) ctl.counters[kStopFlag] = 1u;
      else if (ctl.threshold >= 0.f) ctl.counters[0] = 0u;
      if (ctl.host_tail) {
        volatile uint32_t *ht = ctl.host_tail;
        for (uint32_t i = 0; i < 4; i++) ht[i] = dcount_d ? (uint32_t)dcount_d[gridDim.x + i] : 0u;
        ht[4] = stop ? 1u : 0u;
        __threadfence_system();
        ht[5] = ctl.seq;
      }
    }
    if (stop) return;
  }
  const double *d = delta + (size_t)c * D;
  float *cen = centroids + (size_t)c * D;
  const uint32_t cnt_old = ccounts[c];
  // (dcount_d: the fused reduce buffer's tail -- sums of int32 counts, exact in fp64)
  const uint32_t cnt_new = cnt_old + (uint32_t)(dcount_d ? (int32_t)dcount_d[c] : dcount[c]);
  const double w = (double)cnt_old;
  if (cnt_new == 0) {
    // empty cluster => NaN centroid row, never chosen again (kmeans.cu:425-426, README "NaN
    // centroid").  The reference gets there through 0 * (1/0); its fp32 residual c*count - sum is
    // usually exactly 0, ours (fp64) usually is not and would give +-inf, which poisons the
    // k-means++ of the Yinyang group clustering -- so the contract value is written directly.
    for (uint32_t f = threadIdx.x; f < D; f += blockDim.x) cen[f] = __builtin_nanf("");
  } else if (METRIC == 0) {
    const double cn = (double)cnt_new;  // 0 -> 0/0 = NaN or x/0 = inf: never chosen again
    for (uint32_t f = threadIdx.x; f < D; f += blockDim.x) cen[f] = (float)(((double)cen[f] * w + d[f]) / cn);
  } else {
    __shared__ double red[256];
    double a = 0;
    for (uint32_t f = threadIdx.x; f < D; f += blockDim.x) {
      const double v = (double)cen[f] * w + d[f];
      a += v * v;
    }
    red[threadIdx.x] = a;
    __syncthreads();
    for (uint32_t s = blockDim.x / 2; s > 0; s >>= 1) {
      if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
      __syncthreads();
    }
    const double nrm = sqrt(red[0]);  // empty cluster: 0/0 = NaN
    for (uint32_t f = threadIdx.x; f < D; f += blockDim.x) cen[f] = (float)(((double)cen[f] * w + d[f]) / nrm);
  }
  __syncthreads();
  if (threadIdx.x == 0) ccounts[c] = cnt_new;
}

hipError_t launch_apply_delta(int metric, const double *delta, const int32_t *dcount, const double *dcount_d,
                              uint32_t K, uint32_t D, float *centroids, uint32_t *ccounts, const StopCtl &stop,
                              hipStream_t st) {
  const uint32_t bs = D >= 256 ? 256 : (D > 64 ? 128 : 64);
  if (metric == 0)
    hipLaunchKernelGGL((apply_delta_kernel<0>), dim3(K), dim3(bs), 0, st, delta, dcount, dcount_d, D, centroids, ccounts,
                       stop);
  else
    hipLaunchKernelGGL((apply_delta_kernel<1>), dim3(K), dim3(bs), 0, st, delta, dcount, dcount_d, D, centroids, ccounts,
                       stop);
  return hipGetLastError();
}

// several row shards on ONE device (KMCUDA_AMD_VIRTUAL_SHARDS): the all-reduce's stand-in.  Ascending buffer
// order, the same sum in every buffer
__global__ __launch_bounds__(256) void sum_buffers_kernel(double *const *__restrict__ bufs, uint32_t nbuf, size_t len) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < len; i += (size_t)gridDim.x * 256) {
    double a = bufs[0][i];
    for (uint32_t b = 1; b < nbuf; b++) a += bufs[b][i];
    for (uint32_t b = 0; b < nbuf; b++) bufs[b][i] = a;
  }
}
hipError_t launch_sum_buffers(double *const *bufs_dev, uint32_t nbuf, size_t len, hipStream_t st) {
  const uint32_t grid = (uint32_t)std::min<size_t>((len + 255) / 256, 2048);
  hipLaunchKernelGGL(sum_buffers_kernel, dim3(grid ? grid : 1), dim3(256), 0, st, bufs_dev, nbuf, len);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------
// adjust_exact: the reference's kmeans_adjust (kmeans.cu:366-429) restated operation for
// operation -- c*count, then ONE serial Kahan chain per centroid over its move events in
// ascending sample order with a single compensation term shared by all features and samples
// (kmeans.cu:388, :410-419), then normalize (metric_abstraction.h:138-144, :255-272).  Centroids
// come out BIT-IDENTICAL to the reference's.  The chain is inherently serial (one lane per
// centroid, ~4 dependent VALU ops + a rounding-mode window per element), so this is the
// verification / strict-parity mode (KMCUDA_AMD_EXACT_UPDATE=1, single shard); the default
// update above is the fast one.
//   rows / offsets : the (cluster, sign)-sorted move events of launch_move_events(): segment 2c =
//                    rows that moved INTO c, 2c+1 = rows that moved OUT, each ascending; the lane
//                    merges its two segments on the fly to recover the reference's scan order.
//   work           : the 64 centroid rows of a wave, feature-major [f][lane] (LDS when it fits).
// ---------------------------------------------------------------------------------------
template <int METRIC, bool USE_LDS>
__global__ __launch_bounds__(64) void adjust_exact_kernel(const float *__restrict__ samples, uint32_t D, uint32_t K,
                                                          const uint32_t *__restrict__ rows,
                                                          const uint32_t *__restrict__ offsets,
                                                          float *__restrict__ centroids,
                                                          uint32_t *__restrict__ ccounts, float *__restrict__ work_g) {
  extern __shared__ float work_l[];
  float *work = USE_LDS ? work_l : work_g + (size_t)blockIdx.x * D * 64;
  const uint32_t lane = threadIdx.x;
  const uint32_t c = blockIdx.x * 64 + lane;
  const bool live = c < K;
  uint32_t my_count = live ? ccounts[c] : 0;
  float *cen = centroids + (size_t)(live ? c : 0) * D;
  {
    const float fmy = (float)my_count;  // _const<F>(my_count), kmeans.cu:380
    for (uint32_t f = 0; f < D; f++) work[f * 64 + lane] = live ? cen[f] * fmy : 0.f;
  }
  uint32_t ia = live ? offsets[2 * c] : 0, ea = live ? offsets[2 * c + 1] : 0;
  uint32_t ib = ea, eb = live ? offsets[2 * c + 2] : 0;
  float corr = 0.f;
  while (ia < ea || ib < eb) {  // divergent per lane: lanes with fewer events idle
    const uint32_t ra = ia < ea ? rows[ia] : 0xFFFFFFFFu;
    const uint32_t rb = ib < eb ? rows[ib] : 0xFFFFFFFFu;
    const bool in = ra < rb;  // a row is never in both lists of one centroid
    const uint32_t row = in ? ra : rb;
    const float fsign = in ? 1.f : -1.f;
    if (in) { ia++; my_count++; } else { ib++; my_count--; }
    const float *x = samples + (size_t)row * D;
    uint32_t f = 0;
    if ((D & 3u) == 0) {
      for (; f < D; f += 4) {
        const float4 xv = *reinterpret_cast<const float4 *>(x + f);
        float c0 = work[(f + 0) * 64 + lane], c1 = work[(f + 1) * 64 + lane];
        float c2 = work[(f + 2) * 64 + lane], c3 = work[(f + 3) * 64 + lane];
        float y, t;
        y = fma_rd(xv.x, fsign, corr); t = c0 + y; corr = y - (t - c0); c0 = t;
        y = fma_rd(xv.y, fsign, corr); t = c1 + y; corr = y - (t - c1); c1 = t;
        y = fma_rd(xv.z, fsign, corr); t = c2 + y; corr = y - (t - c2); c2 = t;
        y = fma_rd(xv.w, fsign, corr); t = c3 + y; corr = y - (t - c3); c3 = t;
        work[(f + 0) * 64 + lane] = c0; work[(f + 1) * 64 + lane] = c1;
        work[(f + 2) * 64 + lane] = c2; work[(f + 3) * 64 + lane] = c3;
      }
    }
    for (; f < D; f++) {
      const float cv = work[f * 64 + lane];
      const float y = fma_rd(x[f], fsign, corr);
      const float t = cv + y;
      corr = y - (t - cv);
      work[f * 64 + lane] = t;
    }
  }
  if (!live) return;
  if (METRIC == 0) {  // metric_abstraction.h:138-144: count 0 => NaN/inf row, never chosen again
    const float rc = 1.0f / (float)my_count;
    for (uint32_t f = 0; f < D; f++) cen[f] = work[f * 64 + lane] * rc;
  } else {            // metric_abstraction.h:255-272
    float norm = 0.f, ncorr = 0.f;
    for (uint32_t f = 0; f < D; f++) {
      const float v = work[f * 64 + lane];
      kahan_fold(fma_rd(v, v, ncorr), norm, ncorr);
    }
    norm = 1.0f / sqrtf(norm);
    for (uint32_t f = 0; f < D; f++) cen[f] = work[f * 64 + lane] * norm;
  }
  ccounts[c] = my_count;
}

hipError_t launch_adjust_exact(int metric, const float *samples, uint32_t N, uint32_t D, uint32_t K,
                               const uint32_t *prev, const uint32_t *cur, uint32_t *keys_tmp, uint32_t *vals_tmp,
                               uint32_t *keys_sorted, uint32_t *rows_sorted, uint32_t *offsets2, void *temp,
                               size_t temp_bytes, float *work, float *centroids, uint32_t *ccounts, hipStream_t st) {
  hipLaunchKernelGGL(move_events_kernel, dim3((N + 255) / 256), dim3(256), 0, st, prev, cur, N, K, keys_tmp, vals_tmp);
  hipError_t e = rocprim::radix_sort_pairs(temp, temp_bytes, (const uint32_t *)keys_tmp, keys_sorted,
                                           (const uint32_t *)vals_tmp, rows_sorted, 2 * (size_t)N, 0u,
                                           bits_for(2ull * K), st);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(offsets_kernel, dim3((2 * K + 1 + 255) / 256), dim3(256), 0, st, keys_sorted, 2 * N, 2 * K,
                     offsets2);
  const uint32_t grid = (K + 63) / 64;
  const size_t lds = (size_t)D * 64 * sizeof(float);
  const bool use_lds = lds <= 128 * 1024;
#define KMX_ADJ(M, L)                                                                                        \
  hipLaunchKernelGGL((adjust_exact_kernel<M, L>), dim3(grid), dim3(64), (L) ? lds : 0, st, samples, D, K,    \
                     rows_sorted, offsets2, centroids, ccounts, work)
  if (use_lds) {
    e = hipFuncSetAttribute(metric == 0 ? (const void *)adjust_exact_kernel<0, true>
                                        : (const void *)adjust_exact_kernel<1, true>,
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    if (metric == 0) KMX_ADJ(0, true); else KMX_ADJ(1, true);
  } else {
    if (metric == 0) KMX_ADJ(0, false); else KMX_ADJ(1, false);
  }
#undef KMX_ADJ
  return hipGetLastError();
}

// (kernels.hpp: preload_code_objects) touches one kernel of this translation unit: its code object -- with the radix
// sort's instantiations the largest of the library -- is loaded now instead of at the first update
hipError_t preload_update_code() {
  hipFuncAttributes at;
  return hipFuncGetAttributes(&at, reinterpret_cast<const void *>(&move_count_kernel));
}

}  // namespace kmx
