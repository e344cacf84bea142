// lloyd_gemm.hip -- the Lloyd assignment filter for feature counts beyond the register-resident kernels
// (D > 512: lloyd_f16.hip keeps a wave's rows in registers as the matrix-core B operand, which stops at 512
// features), reference: kmeans_assign_lloyd, src/kmeans.cu:293-364.
//
// Same decision chain as the two-stage filter, with stage 1's scores coming out of ONE plain
// library GEMM -- rocBLAS, f16 operands, f32 accumulation: S = hi(X - mu) . hi(C - mu)^T, the product the coarse
// kernel forms tile by tile -- into a chunk-sized score matrix:
//   row_halves      x' = x - mu as halves, row-major (the GEMM's operand; the engine's row cache for this path)
//                   + per row (||x'||^2, ||x' - hi(x')||^2, NaN-first-feature flag)
//   gemm_decide     one wave per row over its K scores (+ bias): best, second, the coarse bound E_c (the same
//                   formula as lloyd_coarse2_kernel, any summation order of the f32 accumulation is covered by
//                   gamma_D); certain rows are committed, the others are listed with their CONTENDERS -- every
//                   centroid whose score is within 2 E_c of the best: read off the score matrix, no second sweep
//   gemm_contenders one wave per listed row: the contenders scored in fp32 (x' . c' + bias, coalesced row reads),
//                   decided with the f32 bound; what is left goes to the pair / full-scan kernels (lloyd_settle)
// Assignments are therefore bit-identical to the reference's for any input, as on every other path; only how many
// rows each stage settles depends on the data.
#include <hip/hip_fp16.h>

#include "exact.hpp"
#include "filter_common.hpp"
#include "kernels.hpp"

namespace kmx {

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int kGemmCap = 16;       // contenders kept per row
constexpr int kGemmLists = 64;     // undecided rows are appended to 64 lists, cursors one cache line apart

// x' = x - mu as halves (DG per row, zero padded) + the row's record
template <bool HALF_ROWS>
__global__ __launch_bounds__(256) void row_halves_kernel(const void *__restrict__ rows, uint32_t N, uint32_t D,
                                                         uint32_t DG, const float *__restrict__ mu,
                                                         _Float16 *__restrict__ xg, float4 *__restrict__ meta) {
  const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (uint32_t r = blockIdx.x * 4 + wave; r < N; r += gridDim.x * 4) {   // (kernels.hpp: wave_row_grid)
    float n2 = 0.f, d2 = 0.f, o2 = 0.f, x0 = 0.f;
    for (uint32_t f = lane; f < DG; f += 64) {
      float x = 0.f;
      if (f < D) x = HALF_ROWS ? (float)reinterpret_cast<const _Float16 *>(rows)[(size_t)r * D + f]
                               : reinterpret_cast<const float *>(rows)[(size_t)r * D + f];
      const float xc = f < D ? x - mu[f] : 0.f;
      const _Float16 hi = (_Float16)xc;
      const float res = xc - (float)hi;
      xg[(size_t)r * DG + f] = hi;
      n2 = fmaf(xc, xc, n2);
      d2 = fmaf(res, res, d2);
      o2 = fmaf(x, x, o2);
      if (f == 0) x0 = x;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      n2 += __shfl_xor(n2, off);
      d2 += __shfl_xor(d2, off);
      o2 += __shfl_xor(o2, off);
    }
    x0 = __shfl(x0, 0);
    if (lane == 0) meta[r] = make_float4(n2, d2, x0, o2);
  }
}

// One wave per row, 4 rows per wave, 16 rows per block.  scores: rows [row0, row0 + nrows) x ld floats.
__global__ __launch_bounds__(256) void gemm_decide_kernel(
    const float *__restrict__ scores, uint32_t ld, uint32_t row0, uint32_t nrows, uint32_t K, uint32_t K_pad,
    uint32_t DG, const float *__restrict__ bias, const float4 *__restrict__ meta, const uint32_t *__restrict__ stats,
    float eps, float tie_slack, uint32_t *__restrict__ assignments, uint32_t *__restrict__ assignments_prev,
    uint32_t *__restrict__ und_rows, uint32_t *__restrict__ und_cont, uint32_t list_cap,
    uint32_t *__restrict__ cursors, uint32_t *__restrict__ counters) {
  if (counters[kStopFlag] != 0u) return;   // stopped on the device: touch nothing
  const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float cmaxc = sqrtf(__uint_as_float(stats[0])) * 1.000001f;
  const float bmaxc = __uint_as_float(stats[1]);
  const float cmaxo = sqrtf(__uint_as_float(stats[2])) * 1.000001f;
  const float dcmax = sqrtf(__uint_as_float(stats[5])) * 1.000001f;
  const float u = 5.9604645e-8f;
  __shared__ uint32_t sh_und[16], sh_base, sh_changed[4];
  uint32_t my_changed = 0;
  bool und[4];
  float cut[4];
  uint32_t rows_s[4];
#pragma unroll
  for (int q = 0; q < 4; q++) {
    const uint32_t lr = blockIdx.x * 16 + wave * 4 + q;   // row inside the chunk
    const bool live = lr < nrows;
    const uint32_t s = row0 + (live ? lr : 0);
    rows_s[q] = s;
    const float *sc = scores + (size_t)(live ? lr : 0) * ld;
    float v1 = -INFINITY, v2 = -INFINITY;
    uint32_t i1 = 0xFFFFFFFFu;
    for (uint32_t c = lane * 4; c < K_pad; c += 256) {
      const f32x4 s4 = *reinterpret_cast<const f32x4 *>(sc + c);
      const f32x4 b4 = *reinterpret_cast<const f32x4 *>(bias + c);
      const float v[4] = {s4.x + b4.x, s4.y + b4.y, s4.z + b4.z, s4.w + b4.w};
#pragma unroll
      for (int e = 0; e < 4; e++) {
        const bool g1 = v[e] > v1, g2 = v[e] > v2;   // NaN: neither
        v2 = g1 ? v1 : (g2 ? v[e] : v2);
        i1 = g1 ? c + e : i1;
        v1 = g1 ? v[e] : v1;
      }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      const float pv1 = __shfl_xor(v1, off), pv2 = __shfl_xor(v2, off);
      const uint32_t pi1 = __shfl_xor(i1, off);
      const bool g = pv1 > v1 || (pv1 == v1 && pi1 < i1);
      const float second = fmaxf(g ? v1 : pv1, fmaxf(v2, pv2));
      i1 = g ? pi1 : i1;
      v1 = g ? pv1 : v1;
      v2 = second;
    }
    const float4 m = meta[s];
    const bool insane = (m.z != m.z);   // kmeans.cu:312
    const float xn = sqrtf(m.x) * 1.0001f, xo = sqrtf(m.w) * 1.0001f;
    const float dx = sqrtf(m.y) * 1.0001f;
    // |score - reference score| <= E_c: lloyd_coarse2_kernel's bound (f32 accumulation of D exact half products in
    // any order, the operands' measured rounding residuals, half underflow, a 2e-6 relative slack), + E_ref
    const float e_c = 2.0f * eps * (xn * cmaxc + bmaxc) + (xn * dcmax + dx * cmaxc + dx * dcmax) * 1.001f +
                      6e-8f * sqrtf((float)DG) * (xn + cmaxc) + 2.0e-6f * (1.001f * xn * cmaxc + bmaxc);
    const float e_ref = u * (12.0f * xo * cmaxo + 4.0f * cmaxo * cmaxo);
    const float thr = 2.0f * (e_c + e_ref) * 1.001f + tie_slack;
    const bool in_range = (xn < 6.0e4f) && (cmaxc < 6.0e4f) && (v1 > -1.0e38f) && (i1 < K);
    const bool certain = insane || (in_range && ((v1 - v2) > thr));   // NaN gap / thr => not certain
    if (live && certain && lane == 0 && commit_row(s, insane ? K : i1, assignments, assignments_prev)) my_changed++;
    und[q] = live && !certain;
    cut[q] = in_range ? v1 - thr : __builtin_nanf("");
    if (lane == 0) sh_und[wave * 4 + q] = und[q] ? 1u : 0u;
  }
  if (lane == 0) sh_changed[wave] = my_changed;
  __syncthreads();
  // one cursor bump per block, on the block's list (64 lists: the cursors sit on 64 cache lines)
  const uint32_t list = blockIdx.x % kGemmLists;
  if (threadIdx.x == 0) {
    uint32_t n = 0;
    for (int i = 0; i < 16; i++) n += sh_und[i];
    sh_base = n ? atomicAdd(&cursors[list * 32], n) : 0u;
    const uint32_t ch = sh_changed[0] + sh_changed[1] + sh_changed[2] + sh_changed[3];
    if (ch) atomicAdd(&counters[0], ch);
    if (n) atomicAdd(&counters[4], n);
  }
  __syncthreads();
#pragma unroll
  for (int q = 0; q < 4; q++) {
    if (!und[q]) continue;   // wave-uniform
    uint32_t before = 0;
    for (int i = 0; i < wave * 4 + q; i++) before += sh_und[i];
    const size_t slot = (size_t)list * list_cap + sh_base + before;
    // the contenders: every centroid whose score reaches the cut-off (NaN cut-off: none -- the row goes to the
    // full scan); the row's scores are still in L2
    const uint32_t lr = blockIdx.x * 16 + wave * 4 + q;
    const float *sc = scores + (size_t)lr * ld;
    uint32_t n = 0;
    for (uint32_t c0 = 0; c0 < K_pad; c0 += 256) {
      const uint32_t c = c0 + lane * 4;
      bool hit[4] = {false, false, false, false};
      if (c < K_pad) {
        const f32x4 s4 = *reinterpret_cast<const f32x4 *>(sc + c);
        const f32x4 b4 = *reinterpret_cast<const f32x4 *>(bias + c);
        hit[0] = s4.x + b4.x >= cut[q]; hit[1] = s4.y + b4.y >= cut[q];
        hit[2] = s4.z + b4.z >= cut[q]; hit[3] = s4.w + b4.w >= cut[q];
      }
      const uint32_t mine = (uint32_t)hit[0] + (uint32_t)hit[1] + (uint32_t)hit[2] + (uint32_t)hit[3];
      uint32_t incl = mine;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const uint32_t t = (uint32_t)__shfl_up((int)incl, o);
        if ((int)lane >= o) incl += t;
      }
      uint32_t at = n + incl - mine;
#pragma unroll
      for (int e = 0; e < 4; e++)
        if (hit[e]) {
          if (at < (uint32_t)kGemmCap) und_cont[slot * (kGemmCap + 1) + 1 + at] = c + e;
          at++;
        }
      n += (uint32_t)__shfl((int)incl, 63);
    }
    if (lane == 0) {
      und_rows[slot] = rows_s[q];
      und_cont[slot * (kGemmCap + 1)] = n;
    }
  }
}

// One wave per listed row: the contenders in fp32, the decision (as lloyd_refine_kernel's contender phase).  A row
// whose best three are still within the f32 bound is settled HERE with the reference's exact chains over its
// contenders only -- every other centroid is already ruled out by the cut-off -- one chain per lane (lanes beyond
// the row's contenders idle: such rows are rare, and the alternative is a full scan of all K).  Only rows without a
// usable list (more than kGemmCap contenders, operands out of the half range, NaN scores) go to the full scan.
template <int METRIC, bool FAST>
__global__ __launch_bounds__(256) void gemm_contenders_kernel(
    const float *__restrict__ samples, uint32_t D, uint32_t DG, uint32_t K, const float *__restrict__ centroids,
    const float *__restrict__ csqr, const float *__restrict__ cfil,
    const float *__restrict__ bias, const float *__restrict__ mu, const uint32_t *__restrict__ stats, float eps,
    float tie_slack, const uint32_t *__restrict__ und_rows, const uint32_t *__restrict__ und_cont, uint32_t list_cap,
    const uint32_t *__restrict__ cursors, uint32_t *__restrict__ assignments, uint32_t *__restrict__ assignments_prev,
    uint32_t *__restrict__ flagged, uint32_t *__restrict__ pairs, uint32_t *__restrict__ counters) {
  const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint32_t list = blockIdx.y;
  const uint32_t total = cursors[list * 32];
  const float cmaxc = sqrtf(__uint_as_float(stats[0])) * 1.000001f;
  const float bmaxc = __uint_as_float(stats[1]);
  const float cmaxo = sqrtf(__uint_as_float(stats[2])) * 1.000001f;
  const float u = 5.9604645e-8f;
  __shared__ uint32_t sh_pair[4], sh_flag[4], sh_chg[4], sh_pbase, sh_fbase;
  for (uint32_t p0 = blockIdx.x * 4; p0 < total; p0 += gridDim.x * 4) {   // block-uniform trip count
    const uint32_t p = p0 + wave;
    const bool live = p < total;
    const size_t slot = (size_t)list * list_cap + (live ? p : 0);
    const uint32_t s = und_rows[slot];
    const uint32_t n = live ? und_cont[slot * (kGemmCap + 1)] : 0u;
    const bool usable = n >= 1 && n <= (uint32_t)kGemmCap;
    uint32_t cid[kGemmCap];
#pragma unroll
    for (int i = 0; i < kGemmCap; i++) cid[i] = (usable && (uint32_t)i < n) ? und_cont[slot * (kGemmCap + 1) + 1 + i] : 0u;
    float acc[kGemmCap];
#pragma unroll
    for (int i = 0; i < kGemmCap; i++) acc[i] = 0.f;
    float xn2 = 0.f, xo2 = 0.f, x0 = 0.f;
    const float *xr = samples + (size_t)s * D;
    for (uint32_t f = lane * 4; f < DG; f += 256) {
      float x4[4], m4[4];
      if (FAST) {
        const f32x4 a = *reinterpret_cast<const f32x4 *>(xr + f), b = *reinterpret_cast<const f32x4 *>(mu + f);
        x4[0] = a.x; x4[1] = a.y; x4[2] = a.z; x4[3] = a.w;
        m4[0] = b.x; m4[1] = b.y; m4[2] = b.z; m4[3] = b.w;
      } else {
#pragma unroll
        for (int e = 0; e < 4; e++) {
          x4[e] = (f + e) < D ? xr[f + e] : 0.f;
          m4[e] = mu[f + e];   // DG floats, zero beyond D
        }
      }
      float xc[4];
#pragma unroll
      for (int e = 0; e < 4; e++) {
        xc[e] = x4[e] - m4[e];
        xn2 = fmaf(xc[e], xc[e], xn2);
        xo2 = fmaf(x4[e], x4[e], xo2);
      }
      if (f == 0) x0 = x4[0];
#pragma unroll
      for (int i = 0; i < kGemmCap; i++) {
        if ((uint32_t)i < n && usable) {   // wave-uniform
          const f32x4 c4 = *reinterpret_cast<const f32x4 *>(cfil + (size_t)cid[i] * DG + f);
          acc[i] = fmaf(xc[0], c4.x, fmaf(xc[1], c4.y, fmaf(xc[2], c4.z, fmaf(xc[3], c4.w, acc[i]))));
        }
      }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      xn2 += __shfl_xor(xn2, off);
      xo2 += __shfl_xor(xo2, off);
#pragma unroll
      for (int i = 0; i < kGemmCap; i++) acc[i] += __shfl_xor(acc[i], off);
    }
    x0 = __shfl(x0, 0);
    float v1 = -INFINITY, v2 = -INFINITY, v3 = -INFINITY;
    uint32_t i1 = 0xFFFFFFFFu, i2 = 0xFFFFFFFFu;
#pragma unroll
    for (int i = 0; i < kGemmCap; i++) {
      if (usable && (uint32_t)i < n) {
        const uint32_t c = cid[i];
        const float v = acc[i] + bias[c];
        const bool g1 = v > v1, g2 = v > v2, g3 = v > v3;
        v3 = g2 ? v2 : (g3 ? v : v3);
        i2 = g1 ? i1 : (g2 ? c : i2);
        v2 = g1 ? v1 : (g2 ? v : v2);
        i1 = g1 ? c : i1;
        v1 = g1 ? v : v1;
      }
    }
    const bool insane = (x0 != x0);
    const float xn = sqrtf(xn2) * 1.0001f, xo = sqrtf(xo2) * 1.0001f;
    const float e_mfma = 2.0f * eps * (xn * cmaxc + bmaxc);   // a recursive fp32 sum of D + 1 terms, any order
    const float e_ref = u * (12.0f * xo * cmaxo + 4.0f * cmaxo * cmaxo);
    const float thr = 2.0f * (e_mfma + e_ref) * 1.001f + tie_slack;
    const bool in_range = usable && (xn < 6.0e4f) && (cmaxc < 6.0e4f) && i1 < K;
    const bool certain = insane || (in_range && ((v1 - v2) > thr));
    const bool two = !certain && in_range && ((v1 - v3) > thr) && i2 < K;
    const bool multi = live && !certain && !two && usable;
    const bool pair_now = live && two, flag_now = live && !certain && !two && !usable;
    bool changed = false;
    if (live && certain && lane == 0) changed = commit_row(s, insane ? K : i1, assignments, assignments_prev);
    if (multi) {   // wave-uniform
      uint32_t mine = 0;
#pragma unroll
      for (int i = 0; i < kGemmCap; i++) mine = ((int)lane == i) ? cid[i] : mine;
      const bool on = lane < n;
      const float *cr = centroids + (size_t)(on ? mine : 0) * D;
      const float *xs = samples + (size_t)__builtin_amdgcn_readfirstlane(s) * D;   // wave-uniform: scalar loads
      float ac = 0.f, co = 0.f;
      uint32_t f = 0;
      for (; f + 8 <= D; f += 8) {   // (the chain order is the reference's: features ascending)
        float xv[8], cv[8];
#pragma unroll
        for (int e = 0; e < 8; e++) { xv[e] = xs[f + e]; cv[e] = cr[f + e]; }
#pragma unroll
        for (int e = 0; e < 8; e++) kahan_fold(fma_rd(xv[e], cv[e], co), ac, co);
      }
      for (; f < D; f++) kahan_fold(fma_rd(xs[f], cr[f], co), ac, co);
      float dist = on ? lloyd_distance<METRIC>(csqr[mine], ac) : 0.f;
      // the reference's ascending scan with strict '<' over the contenders = the smallest distance, the smallest
      // index among equals; a NaN distance never wins
      bool has = on && (dist < 3.402823466e+38f);
      uint32_t best = has ? mine : 0xFFFFFFFFu;
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) {
        const float od = __shfl_xor(dist, off);
        const uint32_t ob = __shfl_xor(best, off);
        const bool take = (ob != 0xFFFFFFFFu) && (best == 0xFFFFFFFFu || od < dist || (od == dist && ob < best));
        if (take) { dist = od; best = ob; }
      }
      if (lane == 0 && best != 0xFFFFFFFFu) changed = commit_row(s, best, assignments, assignments_prev);
    }
    if (lane == 0) { sh_pair[wave] = pair_now ? 1u : 0u; sh_flag[wave] = flag_now ? 1u : 0u; sh_chg[wave] = changed ? 1u : 0u; }
    __syncthreads();
    if (threadIdx.x == 0) {
      const uint32_t np = sh_pair[0] + sh_pair[1] + sh_pair[2] + sh_pair[3];
      const uint32_t nf = sh_flag[0] + sh_flag[1] + sh_flag[2] + sh_flag[3];
      const uint32_t nc = sh_chg[0] + sh_chg[1] + sh_chg[2] + sh_chg[3];
      sh_pbase = np ? atomicAdd(&counters[3], np) : 0u;
      sh_fbase = nf ? atomicAdd(&counters[1], nf) : 0u;
      if (nc) atomicAdd(&counters[0], nc);
    }
    __syncthreads();
    if (lane == 0) {
      uint32_t pb = 0, fb = 0;
      for (uint32_t w = 0; w < wave; w++) { pb += sh_pair[w]; fb += sh_flag[w]; }
      if (pair_now) {
        const size_t at = (size_t)sh_pbase + pb;
        pairs[3 * at + 0] = s; pairs[3 * at + 1] = i1; pairs[3 * at + 2] = i2;
      }
      if (flag_now) flagged[sh_fbase + fb] = s;
    }
    __syncthreads();
  }
}

hipError_t launch_row_halves(const void *rows, bool half_rows, uint32_t N, uint32_t D, uint32_t DG, const float *mu,
                             void *xg, float *meta, hipStream_t st) {
  if (N == 0) return hipSuccess;
  if (half_rows)
    hipLaunchKernelGGL((row_halves_kernel<true>), dim3(wave_row_grid(N)), dim3(256), 0, st, rows, N, D, DG, mu,
                       reinterpret_cast<_Float16 *>(xg), reinterpret_cast<float4 *>(meta));
  else
    hipLaunchKernelGGL((row_halves_kernel<false>), dim3(wave_row_grid(N)), dim3(256), 0, st, rows, N, D, DG, mu,
                       reinterpret_cast<_Float16 *>(xg), reinterpret_cast<float4 *>(meta));
  return hipGetLastError();
}

// rows one list can receive: chunks are whole multiples of 16 x 64 rows (gemm_chunk_rows), so every chunk but the
// last spreads its blocks evenly over the lists
uint32_t gemm_list_cap(uint32_t N) { return ((N + 15) / 16 + kGemmLists - 1) / kGemmLists * 16 + 32; }
uint32_t gemm_chunk_rows(uint32_t N, uint32_t K_pad) {
  // scores of a chunk: at most 1 GiB, whole multiples of 1024 rows (measured at 2M x 1024, K = 1024: 9.9 ms per pass
  // with 256K-row chunks, 10.7 with 64K, 12.0 with 16K: the GEMM wants the large n, the score matrix does not stay
  // in cache either way)
  uint64_t r = (1ull << 28) / (K_pad ? K_pad : 1);
  r = r / 1024 * 1024;
  if (r < 1024) r = 1024;
  const uint64_t all = ((uint64_t)N + 1023) / 1024 * 1024;
  return (uint32_t)(r < all ? r : all);
}
size_t gemm_cont_words(uint32_t N) { return (size_t)kGemmLists * gemm_list_cap(N) * (kGemmCap + 1); }
size_t gemm_rows_words(uint32_t N) { return (size_t)kGemmLists * gemm_list_cap(N); }

hipError_t launch_gemm_decide(const LloydArgs &a, const float *scores, uint32_t ld, uint32_t row0, uint32_t nrows,
                              uint32_t DG, const float *meta, uint32_t *und_rows, uint32_t *und_cont,
                              uint32_t *cursors, hipStream_t st) {
  if (nrows == 0) return hipSuccess;
  // (row0 is a multiple of 16: whole blocks map to whole lists)
  hipLaunchKernelGGL(gemm_decide_kernel, dim3((nrows + 15) / 16), dim3(256), 0, st, scores, ld, row0, nrows, a.K, a.K_pad,
                     DG, a.bias, reinterpret_cast<const float4 *>(meta), a.stats, a.eps, a.tie_slack, a.assignments,
                     a.assignments_prev, und_rows, und_cont, gemm_list_cap(a.N), cursors, a.counters);
  return hipGetLastError();
}

hipError_t launch_gemm_contenders(int metric, const LloydArgs &a, const float *centroids, uint32_t DG,
                                  const uint32_t *und_rows, const uint32_t *und_cont, const uint32_t *cursors,
                                  hipStream_t st) {
  if (a.N == 0) return hipSuccess;
  const bool fast = a.D == DG && (((uintptr_t)a.samples) & 15u) == 0;
  const dim3 grid(64, kGemmLists);
#define KMX_GC_LAUNCH(M, F)                                                                                          \
  hipLaunchKernelGGL((gemm_contenders_kernel<M, F>), grid, dim3(256), 0, st, a.samples, a.D, DG, a.K, centroids, a.csqr,  \
                     a.cfil, a.bias, a.mu, a.stats, a.eps, a.tie_slack, und_rows, und_cont, gemm_list_cap(a.N), cursors, \
                     a.assignments, a.assignments_prev, a.flagged, a.pairs, a.counters)
  if (metric == 0) { if (fast) KMX_GC_LAUNCH(0, true); else KMX_GC_LAUNCH(0, false); }
  else { if (fast) KMX_GC_LAUNCH(1, true); else KMX_GC_LAUNCH(1, false); }
#undef KMX_GC_LAUNCH
  return hipGetLastError();
}

}  // namespace kmx
