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
//                    (prev, cur): histogram of the keys (integer atomics: exact) -> scan -> the host
//                    reads (event count, largest bucket), then either
//     buckets        (largest bucket <= 8192) rows scattered to their key's bucket through atomic
//                    cursors, each bucket sorted ascending in LDS: the set is exact, the order fixed
//     radix sort     (else) events compacted in row order (count per 1024-row block, scan, write)
//                    and sorted stably by key (rocprim)
//                    -- both give per (cluster, sign) segments with rows ascending, bit-identical sums
//   segment_sums     grid (2K, kSumSplit): fp64 column sums of each segment slice, rows read as
//                    whole coalesced rows (only MOVED rows are touched: late iterations are cheap)
//   fold_delta       delta[c] = sum_in - sum_out (fixed order), dcount[c] = n_in - n_out
//   [row-sharded multi-GPU: all-reduce of delta / dcount happens here]
//   apply_delta      the formula above in fp64, rounded once to fp32
// Centroids agree with the reference's fp32 Kahan chain to a few ulp (tests: 2e-6 relative);
// they are bit-reproducible run to run and independent of kSumSplit.  An empty cluster yields
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

// grid (2K, kSumSplit); thread t owns features t, t+blockDim, ...
__global__ void segment_sums_kernel(const float *__restrict__ samples, uint32_t D, const uint32_t *__restrict__ rows,
                                    const uint32_t *__restrict__ offsets, double *__restrict__ partial) {
  const uint32_t seg = blockIdx.x, j = blockIdx.y;
  const uint32_t beg = offsets[seg], end = offsets[seg + 1];
  const uint32_t n = end - beg;
  const uint32_t chunk = (n + kSumSplit - 1) / kSumSplit;
  uint32_t r0 = beg + j * chunk, r1 = r0 + chunk;
  if (r0 > end) r0 = end;
  if (r1 > end) r1 = end;
  for (uint32_t f = threadIdx.x; f < D; f += blockDim.x) {
    double a0 = 0, a1 = 0, a2 = 0, a3 = 0;
    uint32_t r = r0;
    for (; r + 4 <= r1; r += 4) {
      const float x0 = samples[(size_t)rows[r + 0] * D + f];
      const float x1 = samples[(size_t)rows[r + 1] * D + f];
      const float x2 = samples[(size_t)rows[r + 2] * D + f];
      const float x3 = samples[(size_t)rows[r + 3] * D + f];
      a0 += x0; a1 += x1; a2 += x2; a3 += x3;
    }
    for (; r < r1; r++) a0 += samples[(size_t)rows[r] * D + f];
    partial[((size_t)seg * kSumSplit + j) * D + f] = (a0 + a1) + (a2 + a3);
  }
}

// tail (may be null): the fused reduce buffer's end, [dcount as doubles (K) | counters 0..3 as doubles]
// right behind delta -- everything one all-reduce carries (exact in fp64: |values| < 2^32)
__global__ void fold_delta_kernel(const double *__restrict__ partial, const uint32_t *__restrict__ offsets,
                                  uint32_t K, uint32_t D, double *__restrict__ delta, int32_t *__restrict__ dcount,
                                  double *__restrict__ tail, const uint32_t *__restrict__ counters) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (tail && i < 4) tail[K + i] = (double)counters[i];
  if (i >= (size_t)K * D) return;
  const uint32_t c = i / D, f = i % D;
  double in = 0, out = 0;
  for (uint32_t j = 0; j < kSumSplit; j++) in += partial[((size_t)(2 * c) * kSumSplit + j) * D + f];
  for (uint32_t j = 0; j < kSumSplit; j++) out += partial[((size_t)(2 * c + 1) * kSumSplit + j) * D + f];
  delta[i] = in - out;
  if (f == 0) {
    const int32_t dc = (int32_t)(offsets[2 * c + 1] - offsets[2 * c]) - (int32_t)(offsets[2 * c + 2] - offsets[2 * c + 1]);
    if (dcount) dcount[c] = dc;
    if (tail) tail[c] = (double)dc;
  }
}

// ---- bucket path: histogram / scan / scatter / per-bucket LDS sort ----------------------------
// counters one cache line apart while that stays under ~1 MB per array
uint32_t move_bucket_stride(uint32_t K) {
  uint32_t s = 32;
  while (s > 1 && 2ull * K * s > 262144ull) s >>= 1;
  return s;
}
size_t move_bucket_words(uint32_t K) { return 4 * (size_t)K * move_bucket_stride(K) + 4; }

__global__ __launch_bounds__(256) void move_hist_kernel(const uint32_t *__restrict__ prev,
                                                        const uint32_t *__restrict__ cur, uint32_t N, uint32_t K,
                                                        uint32_t *__restrict__ hist, uint32_t stride) {
  // counters `stride` words apart: packed, the 2K counters of K = 1024 share 64 cache lines and the
  // L2 serialises the atomics per line (measured 118 us for 6e5 events; one line each: ~20 us)
  const uint32_t s = blockIdx.x * 256u + threadIdx.x;
  uint32_t p, a;
  bool ein, eout;
  move_flags(prev, cur, s, N, K, p, a, ein, eout);
  if (ein) atomicAdd(&hist[(size_t)(2u * a) * stride], 1u);
  if (eout) atomicAdd(&hist[(size_t)(2u * p + 1u) * stride], 1u);
}

// hist[0..nkeys) -> offsets[0..nkeys] (exclusive), cursors = offsets, hist zeroed for the next call;
// out[0] = events, out[1] = largest bucket.  One block.
__global__ __launch_bounds__(1024) void move_bucket_scan_kernel(uint32_t *__restrict__ hist, uint32_t nkeys,
                                                                uint32_t stride, uint32_t *__restrict__ offsets,
                                                                uint32_t *__restrict__ cursors,
                                                                uint32_t *__restrict__ out) {
  __shared__ uint32_t wsum[16], wmax[16];
  __shared__ uint32_t carry_s, max_s;
  const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (threadIdx.x == 0) { carry_s = 0; max_s = 0; }
  __syncthreads();
  for (uint32_t base = 0; base < nkeys; base += 1024) {
    const uint32_t i = base + threadIdx.x;
    const uint32_t v = i < nkeys ? hist[(size_t)i * stride] : 0u;
    if (i < nkeys) hist[(size_t)i * stride] = 0u;
    uint32_t inc = v, mx = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const uint32_t t = __shfl_up(inc, o);
      if ((int)lane >= o) inc += t;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = max(mx, __shfl_xor(mx, o));
    if (lane == 63) wsum[wave] = inc;
    if (lane == 0) wmax[wave] = mx;
    __syncthreads();
    uint32_t wbase = 0, all = 0, m = 0;
#pragma unroll
    for (uint32_t k = 0; k < 16; k++) {
      const uint32_t t = wsum[k];
      if (k < wave) wbase += t;
      all += t;
      m = max(m, wmax[k]);
    }
    const uint32_t carry = carry_s;
    if (i < nkeys) {
      const uint32_t ex = carry + wbase + inc - v;
      offsets[i] = ex;
      cursors[(size_t)i * stride] = ex;
    }
    __syncthreads();
    if (threadIdx.x == 0) { carry_s = carry + all; max_s = max(max_s, m); }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    offsets[nkeys] = carry_s;
    out[0] = carry_s;
    out[1] = max_s;
  }
}

__global__ __launch_bounds__(256) void move_scatter_kernel(const uint32_t *__restrict__ prev,
                                                           const uint32_t *__restrict__ cur, uint32_t N, uint32_t K,
                                                           uint32_t *__restrict__ cursors, uint32_t stride,
                                                           uint32_t *__restrict__ rows_out) {
  const uint32_t s = blockIdx.x * 256u + threadIdx.x;
  uint32_t p, a;
  bool ein, eout;
  move_flags(prev, cur, s, N, K, p, a, ein, eout);
  if (ein) rows_out[atomicAdd(&cursors[(size_t)(2u * a) * stride], 1u)] = s;
  if (eout) rows_out[atomicAdd(&cursors[(size_t)(2u * p + 1u) * stride], 1u)] = s;
}

// one block per bucket: ascending row order (bitonic in LDS; the scatter order above is arbitrary).
// A bucket beyond the LDS capacity is ranked by counting through `scratch` (same offsets): O(n^2), only
// there so that the bucket path is CORRECT for any input without the host having looked at the counts
// first -- the host steers such iterations to the radix path as soon as it sees them (launch_move_deltas).
constexpr uint32_t kBucketCap = 4096;
__global__ __launch_bounds__(256) void bucket_sort_kernel(const uint32_t *__restrict__ offsets,
                                                          uint32_t *__restrict__ rows,
                                                          uint32_t *__restrict__ scratch) {
  __shared__ uint32_t v[kBucketCap];
  const uint32_t beg = offsets[blockIdx.x], n = offsets[blockIdx.x + 1] - beg;
  if (n < 2) return;
  if (n > kBucketCap) {
    for (uint32_t i = threadIdx.x; i < n; i += 256) {
      const uint32_t mine = rows[beg + i];
      uint32_t rank = 0;
      for (uint32_t j = 0; j < n; j++) rank += rows[beg + j] < mine ? 1u : 0u;   // rows are distinct
      scratch[beg + rank] = mine;
    }
    __threadfence_block();
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < n; i += 256) rows[beg + i] = scratch[beg + i];
    return;
  }
  uint32_t m = 2;
  while (m < n) m <<= 1;
  for (uint32_t i = threadIdx.x; i < m; i += 256) v[i] = i < n ? rows[beg + i] : 0xFFFFFFFFu;
  __syncthreads();
  for (uint32_t k = 2; k <= m; k <<= 1) {
    for (uint32_t j = k >> 1; j > 0; j >>= 1) {
      for (uint32_t i = threadIdx.x; i < m; i += 256) {
        const uint32_t l = i ^ j;
        if (l > i) {
          const uint32_t a = v[i], b = v[l];
          const bool up = (i & k) == 0;
          if ((a > b) == up) { v[i] = b; v[l] = a; }
        }
      }
      __syncthreads();
    }
  }
  for (uint32_t i = threadIdx.x; i < n; i += 256) rows[beg + i] = v[i];
}

// The host side of the update.  Three ways through it, all leaving per (cluster, sign) segments with
// rows ascending (bit-identical sums):
//   radix   the first iterations (most rows move: the histogram's atomics alone would cost more):
//           compaction + rocprim sort sized by the event count -> ONE host read
//   bucket, checked   histogram, scan, the host reads (events, largest bucket) and picks bucket / radix
//   bucket, unchecked (steady state; MoveState::async_ok) histogram, scan, scatter, LDS sort -- NO host
//           read: the counts go to the pinned words by an async copy nobody waits for; the host looks at
//           whatever has landed (an EARLIER call's counts) only to steer later calls.  A bucket beyond the
//           LDS sort's capacity is still sorted correctly by the kernel's fallback, so nothing depends on
//           the prediction but speed.
hipError_t launch_move_deltas(const float *samples, uint32_t N, uint32_t D, uint32_t K, const uint32_t *prev,
                              const uint32_t *cur, uint32_t *keys_tmp, uint32_t *vals_tmp, uint32_t *keys_sorted,
                              uint32_t *rows_sorted, uint32_t *offsets2, void *temp, size_t temp_bytes,
                              double *partial, double *delta, int32_t *dcount, double *tail,
                              const uint32_t *counters, uint32_t *blockoff, uint32_t *bucket_work, MoveState *ms,
                              hipEvent_t copied, hipStream_t st) {
  // blockoff: N / 1024 + 2 words; bucket_work: move_bucket_words(K) words = histogram | cursors (both
  // 2 K counters, move_bucket_stride(K) words apart) | 2 results; the histogram is zero on entry (engine:
  // zeroed at creation, left zero by the scan); ms->host: 4 pinned words
  const uint32_t stride = move_bucket_stride(K);
  uint32_t *hist = bucket_work, *cursors = bucket_work + 2 * (size_t)K * stride,
           *res = bucket_work + 4 * (size_t)K * stride;
  uint32_t *host_count = ms->host;
  hipError_t e = hipSuccess;
  uint32_t m = 0, maxb = 0;
  const bool force_radix = ms->force == 1, force_sync = ms->force == 2, force_bucket = ms->force == 3;
  // last_events: the newest event count the host knows (2 N before the first call)
  const bool direct_radix = !force_bucket && (force_radix || ms->last_events > N / 2);
  auto bucket_kernels = [&]() {
    hipLaunchKernelGGL(move_scatter_kernel, dim3((N + 255) / 256), dim3(256), 0, st, prev, cur, N, K, cursors,
                       stride, rows_sorted);
    hipLaunchKernelGGL(bucket_sort_kernel, dim3(2 * K), dim3(256), 0, st, offsets2, rows_sorted, keys_sorted);
  };
  if (N == 0) {
    e = hipMemsetAsync(offsets2, 0, (2 * (size_t)K + 1) * sizeof(uint32_t), st);
    if (e != hipSuccess) return e;
  } else if (direct_radix) {
    ms->async_ok = false;
    const uint32_t nb = (N + kMoveRows - 1) / kMoveRows;
    hipLaunchKernelGGL(move_count_kernel, dim3(nb), dim3(256), 0, st, prev, cur, N, K, blockoff);
    hipLaunchKernelGGL(move_scan_kernel, dim3(1), dim3(1024), 0, st, blockoff, nb, blockoff + nb + 1);
    hipLaunchKernelGGL(move_write_kernel, dim3(nb), dim3(256), 0, st, prev, cur, N, K, blockoff, keys_tmp, vals_tmp);
    e = hipGetLastError();
    if (e != hipSuccess) return e;
    e = hipMemcpyAsync(host_count, blockoff + nb + 1, sizeof(uint32_t), hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) return e;
    m = host_count[0];
    if (m) {
      e = rocprim::radix_sort_pairs(temp, temp_bytes, (const uint32_t *)keys_tmp, keys_sorted,
                                    (const uint32_t *)vals_tmp, rows_sorted, (size_t)m, 0u, bits_for(2ull * K), st);
      if (e != hipSuccess) return e;
    }
    // no histogram ran: segment starts by binary search in the sorted keys
    hipLaunchKernelGGL(offsets_kernel, dim3((2 * K + 1 + 255) / 256), dim3(256), 0, st, keys_sorted, m, 2 * K, offsets2);
    e = hipGetLastError();
    if (e != hipSuccess) return e;
    ms->last_events = m;
  } else {
    hipLaunchKernelGGL(move_hist_kernel, dim3((N + 255) / 256), dim3(256), 0, st, prev, cur, N, K, hist, stride);
    hipLaunchKernelGGL(move_bucket_scan_kernel, dim3(1), dim3(1024), 0, st, hist, 2 * K, stride, offsets2, cursors, res);
    e = hipGetLastError();
    if (e != hipSuccess) return e;
    e = hipMemcpyAsync(host_count, res, 2 * sizeof(uint32_t), hipMemcpyDeviceToHost, st);
    if (e != hipSuccess) return e;
    if (force_bucket || (ms->async_ok && !force_sync)) {
      bucket_kernels();
      e = hipGetLastError();
      if (e != hipSuccess) return e;
      // whatever has landed: an earlier call's counts (this call's copy is still queued)
      m = host_count[0];
      maxb = host_count[1];
      if (maxb > kBucketCap - kBucketCap / 4 || m > N / 2) ms->async_ok = false;   // look before leaping next time
      ms->last_events = m;
    } else {
      if (copied) e = hipEventRecord(copied, st);
      if (e != hipSuccess) return e;
      // while the previous call took the bucket path its two kernels are launched BEFORE the host waits for
      // the counts: the GPU runs them during the round trip; if the counts say otherwise the radix path
      // below simply redoes rows_sorted
      const bool speculate = copied && ms->bucket_last;
      if (speculate) bucket_kernels();
      e = copied ? hipEventSynchronize(copied) : hipStreamSynchronize(st);   // the copy, not what was queued behind it
      if (e != hipSuccess) return e;
      m = host_count[0];
      maxb = host_count[1];
      ms->last_events = m;
      ms->bucket_last = false;
      if (m && maxb <= kBucketCap) {
        ms->bucket_last = true;
        ms->async_ok = maxb <= kBucketCap - kBucketCap / 4;   // comfortably inside: stop looking first
        if (!speculate) bucket_kernels();
        e = hipGetLastError();
        if (e != hipSuccess) return e;
      } else if (m) {
        const uint32_t nb = (N + kMoveRows - 1) / kMoveRows;
        hipLaunchKernelGGL(move_count_kernel, dim3(nb), dim3(256), 0, st, prev, cur, N, K, blockoff);
        hipLaunchKernelGGL(move_scan_kernel, dim3(1), dim3(1024), 0, st, blockoff, nb, blockoff + nb + 1);
        hipLaunchKernelGGL(move_write_kernel, dim3(nb), dim3(256), 0, st, prev, cur, N, K, blockoff, keys_tmp, vals_tmp);
        e = rocprim::radix_sort_pairs(temp, temp_bytes, (const uint32_t *)keys_tmp, keys_sorted,
                                      (const uint32_t *)vals_tmp, rows_sorted, (size_t)m, 0u, bits_for(2ull * K), st);
        if (e != hipSuccess) return e;
      }
    }
  }
  // offsets2 (segment starts per key) came out of the histogram scan or the binary search
  const uint32_t bs = D >= 256 ? 256 : (D > 64 ? 128 : 64);
  hipLaunchKernelGGL(segment_sums_kernel, dim3(2 * K, kSumSplit), dim3(bs), 0, st, samples, D, rows_sorted, offsets2,
                     partial);
  const size_t n = (size_t)K * D;
  hipLaunchKernelGGL(fold_delta_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, st, partial, offsets2, K, D,
                     delta, dcount, tail, counters);
  return hipGetLastError();
}

// one block per centroid: c = normalize(c*count + delta), count += dcount
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

}  // namespace kmx
