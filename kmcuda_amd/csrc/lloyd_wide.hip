// lloyd_wide.hip -- the Lloyd assignment filter for rows wider than 256 features (lloyd_f16.hip keeps a wave's rows
// in registers as the matrix-core B operand: that stops at 512 features, and from 257 on this kernel is the faster
// one -- engine.cpp, Engine::init), reference: kmeans_assign_lloyd, src/kmeans.cu:293-364 (any D).
//
// Same decision chain as the two-stage filter: coarse hi.hi scores on the f16 matrix cores with a
// rigorous bound -> the contenders of the undecided rows in fp32 -> two exact chains / full exact scan.  Until round 4
// stage 1 was a library GEMM (rocBLAS) into an 8-GB fp32 score matrix that a second kernel read back (16 GB of HBM
// traffic against 4 GB of operands, 4.7 + 2.5 ms at 2M x 1024 @ 1024).  Since round 5 it is ONE hand-written kernel
// that never writes a score:
//   row_halves       x' = x - mu as halves, row-major (the operand; this path's row cache) + per row
//                    (||x'||^2, ||x' - hi(x')||^2, x_0, ||x||^2)
//   lloyd_wide<0>    S = hi(X') . hi(C')^T tile by tile: 256 rows x 256 centroids per block, BOTH operands streamed
//                    through LDS in 64-feature chunks (LDS-DMA, double buffered), 8 waves of 64 rows x 128 centroids
//                    (8 accumulator tiles: six fragment reads feed eight v_mfma_f32_32x32x16_f16), the running
//                    best / second-best per row kept in registers across the K / 256 centroid blocks exactly as
//                    lloyd_coarse2_kernel keeps them (index bits packed into the score); at the end the same bound
//                    E_c decides: commit, or list the row with its CUT-OFF (best - 2 E_c)
//   lloyd_wide<2/3>  <0> with carried bounds (lloyd_carry.hip): over every row / over the rows the moved bounds no
//                    longer decide (gathered by index), each row leaving with fresh bounds from its best two scores
//   lloyd_wide<1>    the listed rows only, the same sweep (same operands, same products, same order: the same
//                    scores): every centroid whose score reaches the row's cut-off is a CONTENDER (<= 16 kept)
//   wide_contenders  one wave per listed row: the contenders scored in fp32 (x' . c' + bias), decided with the f32
//                    bound; rows still within it are settled on the spot by the reference's exact chains over
//                    their contenders only; what is left goes to the pair / full-scan kernels (lloyd_settle)
// Assignments are therefore bit-identical to the reference's for any input, as on every other path; only how many
// rows each stage settles depends on the data.
#include <hip/hip_fp16.h>

#include "exact.hpp"
#include "filter_common.hpp"
#include "kernels.hpp"
#include "lloyd_coarse.hpp"   // f16x8 / f32x16, lds_frag_issue, lds_frag_wait

namespace kmx {

constexpr int kWideCap = 16;       // contenders kept per row

// x' = x - mu as halves (DG per row, zero padded) + the row's record
template <bool HALF_ROWS>
__global__ __launch_bounds__(256) void row_halves_kernel(const void *__restrict__ rows, uint32_t N, uint32_t D,
                                                         uint32_t DG, const float *__restrict__ mu,
                                                         _Float16 *__restrict__ xg, float4 *__restrict__ meta) {
  const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (blockIdx.x == 0 && wave == 0) {   // ||mu|| behind the rows' records (carry_skip_kernel: the angular bias sums)
    float m2 = 0.f;
    for (uint32_t f = lane; f < DG; f += 64) m2 = fmaf(mu[f], mu[f], m2);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m2 += __shfl_xor(m2, off);
    // (.y: ||mu||^2 as summed -- x.mu = (||x||^2 + ||mu||^2 - ||x - mu||^2) / 2, the angular clamp, filter_common.hpp)
    if (lane == 0) meta[N] = make_float4(sqrtf(m2) * 1.0001f, m2, 0.f, 0.f);
  }
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

// ---------------------------------------------------------------------------------------
// Stage 1.  Block = 8 waves = 256 rows x 256 centroids; wave (wm, wn) owns rows [64 wm, +64) (two 32-row operand
// sets) and centroids [128 wn, +128) (four 32-centroid tiles): 8 accumulator tiles.  Per 16 features a wave reads
// four centroid fragments and two row fragments from LDS and issues eight matrix products (0.75 reads per product;
// a fragment read takes 4 LDS cycles per wave, a product 32 SIMD cycles: 3/8 of the LDS rate at full matrix rate).
// LDS tiles: 256 rows of 64 features (128 bytes), the 16-byte chunk j of tile row r in slot j ^ ((r >> 1) & 7): the
// 16 lanes ds_read_b128 serves per cycle ({0-3, 12-15, 20-27}, ...) then hit 16 different bank groups (a row's
// 128 bytes cover half the banks, (r & 1) picks the half, (r >> 1) & 7 the slot).  The LDS-DMA writes lane-linear, so
// the swizzle is applied to the SOURCE address (as lloyd_coarse2_kernel does).  One barrier per chunk: chunk i + 1
// is in flight while chunk i is multiplied.  A lane (col, h) holds, for row `col` of a set, the scores of centroids
// (r & 3) + 8 (r >> 2) + 4 h of a tile in accumulator register r: the bookkeeping is lane-local.
// ---------------------------------------------------------------------------------------
constexpr int kWideRows = 256, kWideCents = 256, kWideBK = 64;
constexpr int kWideRowB = kWideBK * 2;              // bytes of an LDS tile row
constexpr int kWideTileB = 256 * kWideRowB;         // one operand tile: 32 KB
constexpr int kWideOffA = 0, kWideOffB = 2 * kWideTileB, kWideOffBias = 4 * kWideTileB;
constexpr int kWideOffMerge = kWideOffBias + 1024;  // 4 x 64 rows x (v1, v2, i1)
constexpr int kWideOffCnt = kWideOffMerge + 4 * 64 * 12;
constexpr int kWideOffCont = kWideOffCnt + 1024;    // MODE 1: 256 rows x kWideCap contenders
constexpr size_t kWideLds0 = kWideOffCnt, kWideLds1 = kWideOffCont + 256 * kWideCap * 4;

template <int OFF>
__device__ __forceinline__ f16x8 lds_frag_issue_at(uint32_t addr) {
  f16x8 f;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(f) : "v"(addr), "n"(OFF) : "memory");
  return f;
}
template <int N>
__device__ __forceinline__ void lds_frag_wait6(f16x8 &a0, f16x8 &a1, f16x8 &a2, f16x8 &a3, f16x8 &b0, f16x8 &b1) {
  asm volatile("s_waitcnt lgkmcnt(%6)" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(b0), "+v"(b1) : "n"(N));
}

// MODE 0: every row; commits the rows the bound decides, lists the others (undecided / und_thr, counters[4]).
// MODE 1: the listed rows; writes their contenders (und_cont: kWideCap + 1 words per listed row: the count, the ids).
// MODE 2 / 3 (carried bounds, lloyd_carry.hip): as MODE 0 over every row / over the rows of cy.row_list (the rows
// whose bounds, moved by the centroids' drifts, no longer certify their assignment), and every row the pass looks at
// leaves with fresh bounds read off its best two scores -- lloyd_coarse2_kernel<..., CARRY>'s statements word for word.
template <int MODE>
__global__ __launch_bounds__(512) void lloyd_wide_kernel(
    const _Float16 *__restrict__ xg, const float4 *__restrict__ meta, uint32_t N, uint32_t DG,
    const _Float16 *__restrict__ panelhi, uint32_t K_pad64, uint32_t K, const uint32_t *__restrict__ stats, float eps,
    float tie_slack, uint32_t *__restrict__ assignments, uint32_t *__restrict__ assignments_prev,
    uint32_t *__restrict__ undecided, float *__restrict__ und_thr, uint32_t *__restrict__ und_cont,
    uint32_t *__restrict__ counters, CarryArgs cy) {
  constexpr bool BOUNDS = MODE >= 2, LISTED = MODE == 3;
  typedef __attribute__((address_space(3))) unsigned char lds_byte;
  extern __shared__ __attribute__((aligned(1024))) unsigned char ldsw[];
  if (counters[kStopFlag] != 0u) return;   // the run has stopped on the device: touch nothing
  uint32_t total = N;
  if (MODE == 1) total = __builtin_amdgcn_readfirstlane(counters[4]);
  if (LISTED) total = __builtin_amdgcn_readfirstlane(*cy.n_list);
  if (BOUNDS && blockIdx.x == 0 && threadIdx.x == 0) {
    if (cy.host_report) {   // how long this pass's list was: the host sizes later passes by it
      volatile uint32_t *hr = cy.host_report;
      hr[0] = cy.n_list ? *cy.n_list : 0xFFFFFFFFu;   // (a pass without bounds to move has no list to report)
      hr[1] = cy.seq;
    }
    if (LISTED)   // the rows this pass does not look at (statistics: kmamd_carry_stats)
      *reinterpret_cast<unsigned long long *>(counters + kCarrySkipped) += (unsigned long long)(N - total);
  }
  const uint32_t blk = blockIdx.x;
  if ((uint64_t)blk * kWideRows >= total) return;   // (block-uniform, in front of every barrier and DMA)
  const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_byte *)ldsw;
  if (lds0 & 1023u) __builtin_trap();

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int col = lane & 31, h = lane >> 5, wm = wave >> 1, wn = wave & 1;

  // ---- the rows of this block ----
  // a lane's two rows (one per operand set): position in the pass (MODE 0: the row itself; MODE 1: in the list)
  uint32_t pos[2], srow[2];
  bool live[2];
#pragma unroll
  for (int s = 0; s < 2; s++) {
    pos[s] = blk * kWideRows + wm * 64 + s * 32 + col;
    live[s] = pos[s] < total;
    srow[s] = live[s] ? (MODE == 1 ? undecided[pos[s]] : (LISTED ? cy.row_list[pos[s]] : pos[s])) : 0u;
  }
  float cutv[2] = {0.f, 0.f};
  if (MODE == 1) {
#pragma unroll
    for (int s = 0; s < 2; s++) cutv[s] = live[s] ? und_thr[pos[s]] : __builtin_nanf("");   // NaN: no contender
  }

  // ---- LDS-DMA staging: wave w moves pieces w, w + 8, w + 16, w + 24 (1 KB = 8 tile rows each) of both tiles ----
  // linear byte P = 1024 p + 16 L of a tile lands at P: tile row R = 8 p + (L >> 3), slot L & 7, which must hold
  // chunk (L & 7) ^ ((R >> 1) & 7) of that row -- and ((R >> 1) & 7) = (4 w + (L >> 4)) & 7 for all four pieces
  const uint32_t chunk16 = (uint32_t)(((lane & 7) ^ ((4 * wave + (lane >> 4)) & 7)) * 16);
  const unsigned char *xg_b = reinterpret_cast<const unsigned char *>(xg);
  const unsigned char *pan_b = reinterpret_cast<const unsigned char *>(panelhi);
  const size_t rowbytes = (size_t)DG * 2;
  size_t boff[4], aoff[4];
#pragma unroll
  for (int q = 0; q < 4; q++) {
    const uint32_t R = (uint32_t)(wave + 8 * q) * 8u + (uint32_t)(lane >> 3);
    const uint32_t p = blk * kWideRows + R;
    uint32_t r = 0u;   // rows past the end read row 0: their scores are never used
    if (p < total) r = MODE == 1 ? undecided[p] : (LISTED ? cy.row_list[p] : p);
    boff[q] = (size_t)r * rowbytes + chunk16;
    aoff[q] = 0;
  }
  auto set_aoff = [&](uint32_t pass) {
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const uint32_t R = (uint32_t)(wave + 8 * q) * 8u + (uint32_t)(lane >> 3);
      uint32_t c = pass * kWideCents + R;
      c = c < K_pad64 ? c : K_pad64 - 1u;   // (past the panel: any row; its bias is the floor)
      aoff[q] = (size_t)c * rowbytes + chunk16;
    }
  };
  auto stage_piece = [&](bool rows_tile, int q, uint32_t kc, int buf) {
    const unsigned char *src = (rows_tile ? xg_b + boff[q] : pan_b + aoff[q]) + (size_t)kc * kWideRowB;
    const uint32_t dst = lds0 + (rows_tile ? kWideOffB : kWideOffA) + (uint32_t)buf * kWideTileB + (uint32_t)(wave + 8 * q) * 1024u;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                     (__attribute__((address_space(3))) void *)(uintptr_t)dst, 16, 0, 0);
  };

  const uint32_t npass = (K_pad64 + kWideCents - 1) / kWideCents, NC = DG / kWideBK, nit = npass * NC;
  const float *biashi = reinterpret_cast<const float *>(pan_b + (size_t)K_pad64 * rowbytes);

  // fragment addresses: (row, half) part fixed per lane; k-step j flips the slot's low bits, tile / set / buffer add
  const uint32_t slot0 = (uint32_t)((((h * 4) ^ ((col >> 1) & 7))) * 16);
  const uint32_t fbA = lds0 + kWideOffA + (uint32_t)(wn * 128 + col) * kWideRowB + slot0;
  const uint32_t fbB = lds0 + kWideOffB + (uint32_t)(wm * 64 + col) * kWideRowB + slot0;
  const uint32_t bias_lds = lds0 + kWideOffBias;

  float pinf = INFINITY;
  asm volatile("" : "+s"(pinf));
  auto pack = [&](float v, int r) { return __uint_as_float((__float_as_uint(v) & 0xFFFFFFF0u) | (uint32_t)r); };
  // (lloyd_coarse2_kernel's bookkeeping: two scores at once, 3 ops + 2 packs; v_max3 only sees packed values)
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
  // (plain accesses go through pointers derived from the array: the compiler knows they are LDS)
  float *biasw = reinterpret_cast<float *>(ldsw + kWideOffBias);
  uint32_t *mergew = reinterpret_cast<uint32_t *>(ldsw + kWideOffMerge);
  uint32_t *cnt = reinterpret_cast<uint32_t *>(ldsw + kWideOffCnt);
  uint32_t *cont = reinterpret_cast<uint32_t *>(ldsw + kWideOffCont);

  if (MODE == 1) {   // the rows' contender counters
    if (tid < 256) cnt[tid] = 0u;
  }

  f32x16 acc[2][4];
  float v1[2] = {-INFINITY, -INFINITY}, v2[2] = {-INFINITY, -INFINITY};
  uint32_t tb[2] = {0u, 0u};

  set_aoff(0);
#pragma unroll
  for (int q = 0; q < 4; q++) stage_piece(false, q, 0, 0);
#pragma unroll
  for (int q = 0; q < 4; q++) stage_piece(true, q, 0, 0);

  for (uint32_t pass = 0; pass < npass; pass++) {
    // this pass's 256 biases (the floor past the panel); with two or more chunks per pass the last pass's readers have
    // passed a barrier since -- with ONE (rows of up to 64 features: KMCUDA_AMD_WIDE_MIN_D, the tests' way to this filter
    // at every width) a fast wave would overwrite the biases a slow one has not read yet
    if (NC == 1 && pass != 0) __syncthreads();
    if (tid < 256) {
      const uint32_t c = pass * kWideCents + (uint32_t)tid;
      biasw[tid] = c < K_pad64 ? biashi[c] : -3.0e38f;
    }
    for (uint32_t kc = 0; kc < NC; kc++) {
      const uint32_t it = pass * NC + kc;
      const int buf = (int)(it & 1u);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // my pieces of this chunk have landed ...
      __syncthreads();                                    // ... everybody's have, and the other buffer is free
      if (kc == 0) {
#pragma unroll
        for (int t = 0; t < 4; t++) {
          const uint32_t ba = bias_lds + (uint32_t)(wn * 128 + t * 32 + 4 * h) * 4u;
#pragma unroll
          for (int g = 0; g < 4; g++) {
            const f32x4 b4 = lds_f4(ba + (uint32_t)g * 32u);
            acc[0][t][4 * g + 0] = b4.x; acc[0][t][4 * g + 1] = b4.y; acc[0][t][4 * g + 2] = b4.z; acc[0][t][4 * g + 3] = b4.w;
          }
          acc[1][t] = acc[0][t];
        }
      }
      // The next chunk goes into the other buffer between the products below, and EARLY: the rows' four pieces behind
      // the first k-step's products, the panel's behind the second's.  (Two pieces per k-step, the rows' last, left the
      // late ones ~0.2 us to land before the wait at the top: 5.55 -> 5.03 ms for both sweeps at 2M x 1024 @ 1024,
      // profiles/r5x_*.)  No branch inside the products -- with one, the compiler spills 140 registers once four
      // pieces share a k-step: behind the last chunk the first one is staged once more, into the free buffer.
      uint32_t nkc = kc + 1;
      if (nkc == NC) {
        nkc = 0;
        set_aoff(it + 1 < nit ? pass + 1 : 0);
      }
      uint32_t fa = fbA + (uint32_t)buf * kWideTileB, fb = fbB + (uint32_t)buf * kWideTileB;
      asm volatile("" : "+v"(fa), "+v"(fb));
      f16x8 a[2][4], b[2][2];
      auto issue = [&](int set, int j) {
        const uint32_t xa = fa ^ (uint32_t)(j * 16), xb = fb ^ (uint32_t)(j * 16);
        a[set][0] = lds_frag_issue_at<0>(xa);
        a[set][1] = lds_frag_issue_at<32 * kWideRowB>(xa);
        a[set][2] = lds_frag_issue_at<64 * kWideRowB>(xa);
        a[set][3] = lds_frag_issue_at<96 * kWideRowB>(xa);
        b[set][0] = lds_frag_issue_at<0>(xb);
        b[set][1] = lds_frag_issue_at<32 * kWideRowB>(xb);
      };
      issue(0, 0);
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const int cur = j & 1;
        if (j + 1 < 4) {
          issue(cur ^ 1, j + 1);
          lds_frag_wait6<6>(a[cur][0], a[cur][1], a[cur][2], a[cur][3], b[cur][0], b[cur][1]);
        } else {
          lds_frag_wait6<0>(a[cur][0], a[cur][1], a[cur][2], a[cur][3], b[cur][0], b[cur][1]);
        }
#pragma unroll
        for (int t = 0; t < 4; t++) {
          acc[0][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[cur][t], b[cur][0], acc[0][t], 0, 0, 0);
          acc[1][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[cur][t], b[cur][1], acc[1][t], 0, 0, 0);
          if (j < 2) {
            stage_piece(j == 0, t, nkc, buf ^ 1);
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      }
    }
    // ---- the pass's 256 centroids are scored ----
    if (MODE != 1) {
#pragma unroll
      for (int t = 0; t < 4; t++) {
        const uint32_t gt = pass * 8u + (uint32_t)(wn * 4 + t);   // global 32-centroid tile number
#pragma unroll
        for (int s = 0; s < 2; s++) {
          const float in = v1[s];
#pragma unroll
          for (int r = 0; r < 16; r += 2) book2(acc[s][t][r], acc[s][t][r + 1], r, v1[s], v2[s]);
          tb[s] = (v1[s] != in) ? gt : tb[s];
        }
      }
    } else {
#pragma unroll
      for (int t = 0; t < 4; t++) {
        const uint32_t gt = pass * 8u + (uint32_t)(wn * 4 + t);
#pragma unroll
        for (int s = 0; s < 2; s++) {
          float m = acc[s][t][0];
#pragma unroll
          for (int r = 1; r < 16; r++) m = fmaxf(m, acc[s][t][r]);
          if (m >= cutv[s]) {   // rare: a handful of centroids per listed row
            const uint32_t rl = (uint32_t)(wm * 64 + s * 32 + col);
#pragma unroll
            for (int r = 0; r < 16; r++) {
              if (acc[s][t][r] >= cutv[s]) {
                const uint32_t c = gt * 32u + (uint32_t)((r & 3) + 8 * (r >> 2) + 4 * h);
                const uint32_t at = atomicAdd(&cnt[rl], 1u);
                if (at < (uint32_t)kWideCap) cont[rl * kWideCap + at] = c;
              }
            }
          }
        }
      }
    }
  }

  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (the spare pieces staged behind the last chunk)
  if (MODE == 1) {
    __syncthreads();
    if (tid < 256) {
      const uint32_t p = blk * kWideRows + (uint32_t)tid;
      if (p < total) {
        const uint32_t n = cnt[tid];
        uint32_t *out = und_cont + (size_t)p * (kWideCap + 1);
        out[0] = n;
        for (uint32_t i = 0; i < n && i < (uint32_t)kWideCap; i++) out[1 + i] = cont[(uint32_t)tid * kWideCap + i];
      }
    }
    return;
  }

  // ---- MODE 0 / 2 / 3: the two half-waves, then the two waves that share the rows, then the decision ----
  float cut[2] = {0.f, 0.f};
  uint32_t i1v[2];
#pragma unroll
  for (int s = 0; s < 2; s++) {
    const uint32_t r = __float_as_uint(v1[s]) & 15u;
    uint32_t i1 = tb[s] * 32u + (r & 3u) + 8u * (r >> 2) + 4u * (uint32_t)h;
    const float pv1 = __shfl_xor(v1[s], 32), pv2 = __shfl_xor(v2[s], 32);
    const uint32_t pi1 = __shfl_xor(i1, 32);
    const bool g = pv1 > v1[s];
    const float second = fmaxf(g ? v1[s] : pv1, fmaxf(v2[s], pv2));
    i1v[s] = g ? pi1 : i1;
    v1[s] = g ? pv1 : v1[s];
    v2[s] = second;
  }
  uint32_t *mg = mergew + (uint32_t)(wm * 64) * 3u;
  if (wn == 1 && h == 0) {
#pragma unroll
    for (int s = 0; s < 2; s++) {
      uint32_t *at = mg + (uint32_t)(s * 32 + col) * 3u;
      at[0] = __float_as_uint(v1[s]);
      at[1] = __float_as_uint(v2[s]);
      at[2] = i1v[s];
    }
  }
  __syncthreads();
  const float cmaxc = sqrtf(__uint_as_float(stats[0])) * 1.000001f;
  const float bmaxc = __uint_as_float(stats[1]);
  const float cmaxo = sqrtf(__uint_as_float(stats[2])) * 1.000001f;
  const float dcmax = sqrtf(__uint_as_float(stats[5])) * 1.000001f;   // max ||c' - hi(c')||; inf = no bound
  const float u = 5.9604645e-8f;
  uint32_t und_count = 0, changed_count = 0;
  unsigned long long um[2] = {0ull, 0ull};
  bool und[2] = {false, false};
  if (wn == 0) {
#pragma unroll
    for (int s = 0; s < 2; s++) {
      const uint32_t *at = mg + (uint32_t)(s * 32 + col) * 3u;
      const float pv1 = __uint_as_float(at[0]), pv2 = __uint_as_float(at[1]);
      const uint32_t pi1 = at[2];
      const bool g = pv1 > v1[s];
      const float second = fmaxf(g ? v1[s] : pv1, fmaxf(v2[s], pv2));
      const uint32_t i1 = g ? pi1 : i1v[s];
      const float b1 = g ? pv1 : v1[s], b2 = second;
      const float4 m = meta[srow[s]];
      const bool insane = (m.z != m.z);   // kmeans.cu:312
      const float xn = sqrtf(m.x) * 1.0001f, xo = sqrtf(m.w) * 1.0001f;
      const float dx = sqrtf(m.y) * 1.0001f;
      // |score - reference score| <= E_c: lloyd_coarse2_kernel's bound (f32 accumulation of DG exact half products, the
      // operands' measured rounding residuals, half underflow, the index bits packed into the score), + E_ref
      const float e_c = 2.0f * eps * (xn * cmaxc + bmaxc) + (xn * dcmax + dx * cmaxc + dx * dcmax) * 1.001f +
                        6e-8f * sqrtf((float)DG) * (xn + cmaxc) + 2.0e-6f * (1.001f * xn * cmaxc + bmaxc);
      const float e_ref = u * (12.0f * xo * cmaxo + 4.0f * cmaxo * cmaxo);
      const float thr = 2.0f * (e_c + e_ref) * 1.001f + tie_slack;
      const bool in_range = (xn < 6.0e4f) && (cmaxc < 6.0e4f) && (b1 > -1.0e38f) && (i1 < K);
      // angular: no centroid but the best may reach the clamp at product 1, the best not the one at -1
      // (filter_common.hpp); x.mu from the three squared norms on record, each an fp32 sum of DG terms
      const bool angular = tie_slack > 0.f;
      const float mun2 = meta[N].y;
      const ClampLimits lim = clamp_limits(angular, 0.5f * ((m.w + mun2) - m.x), dot_error((int)DG, (m.w + mun2) + m.x), 0.5f * thr);
      const bool certain = insane || (in_range && ((b1 - b2) > thr) && (b2 < lim.hi) && (b1 > lim.lo));   // NaN anywhere => not certain
      const bool mine = (h == 0) && live[s];
      bool changed = false;
      if (mine && certain) changed = commit_row(srow[s], insane ? K : i1, assignments, assignments_prev);
      und[s] = mine && !certain;
      if (BOUNDS && mine) {
        // lloyd_coarse2_kernel's statements (lloyd_coarse.hpp, CARRY): d(x, c)^2 = ||x'||^2 - 2 s(c), every coarse score
        // within e_c of s(c), m.x within 2 eps of ||x'||^2, every centroid other than i1 scored <= b2.  A row the later
        // stages decide may end on another contender than i1: its lower bound is void (0), its upper bound holds.
        const float e = e_c * 1.001f;
        if (cy.angular) {
          // (with the rooms below the clamp at product 1 / above the one at -1: lloyd_coarse.hpp)
          float gapv = (certain && !insane && in_range)
                           ? fminf(fminf((b1 - e) - (b2 + e), lim.hi - b2), b1 - lim.lo) * 0.999999f : -INFINITY;
          if (!(gapv == gapv)) gapv = -INFINITY;
          cy.ub[srow[s]] = gapv;
        } else {
          float ubv = INFINITY, lbv = 0.f;
          if (!insane && in_range) {
            const float d2u = fmaxf(m.x * (1.0f + 2.0f * eps) - 2.0f * (b1 - e), 0.f);
            const float d2l = m.x * (1.0f - 2.0f * eps) - 2.0f * (b2 + e);
            const float geo = 2.4e-7f * (xn + cmaxc);
            ubv = sqrtf(d2u) * 1.000001f + geo;
            if (certain && d2l > 0.f) lbv = fmaxf(sqrtf(d2l) * 0.999999f - geo, 0.f);
            if (!(ubv == ubv)) ubv = INFINITY;
            if (!(lbv == lbv)) lbv = 0.f;
          }
          cy.ub[srow[s]] = ubv;
          cy.lb[srow[s]] = lbv;
        }
      }
      cut[s] = in_range ? b1 - thr : __builtin_nanf("");
      // (angular: a centroid whose product may reach 1 ties with the best at distance 0 -- a contender)
      if (angular) cut[s] = (lim.hi == lim.hi) ? fminf(cut[s], lim.hi) : __builtin_nanf("");
      um[s] = __ballot(und[s]);
      changed_count += (uint32_t)__popcll(__ballot(changed));
      und_count += (uint32_t)__popcll(um[s]);
    }
  }
  // ONE pair of global atomics per block (same-address atomics are served one at a time by L2)
  __shared__ uint32_t blk_und[8], blk_changed[8], blk_base;
  if (lane == 0) {
    blk_und[wave] = und_count;
    blk_changed[wave] = changed_count;
  }
  __syncthreads();
  if (tid == 0) {
    uint32_t tu = 0, tc = 0;
#pragma unroll
    for (int w = 0; w < 8; w++) {
      tu += blk_und[w];
      tc += blk_changed[w];
    }
    if (tc) atomicAdd(&counters[0], tc);
    blk_base = tu ? atomicAdd(&counters[4], tu) : 0u;
  }
  __syncthreads();
  if (und_count) {
    uint32_t base = blk_base;
    for (int w = 0; w < wave; w++) base += blk_und[w];
    const unsigned long long below = (1ull << lane) - 1ull;
    if (und[0]) {
      const uint32_t at = base + (uint32_t)__popcll(um[0] & below);
      undecided[at] = srow[0];
      und_thr[at] = cut[0];
    }
    if (und[1]) {
      const uint32_t at = base + (uint32_t)__popcll(um[0]) + (uint32_t)__popcll(um[1] & below);
      undecided[at] = srow[1];
      und_thr[at] = cut[1];
    }
  }
}

// One wave per listed row: the contenders in fp32, the decision (as lloyd_refine_kernel's contender phase).  A row
// whose best three are still within the f32 bound is settled HERE with the reference's exact chains over its
// contenders only -- every other centroid is already ruled out by the cut-off -- one chain per lane (lanes beyond
// the row's contenders idle: such rows are rare, and the alternative is a full scan of all K).  Only rows without a
// usable list (more than kWideCap contenders, operands out of the half range, NaN scores) go to the full scan.
template <int METRIC, bool FAST>
__global__ __launch_bounds__(256, 4) void wide_contenders_kernel(
    const float *__restrict__ samples, uint32_t D, uint32_t DG, uint32_t K, const float *__restrict__ centroids,
    const float *__restrict__ csqr, const float *__restrict__ cfil,
    const float *__restrict__ bias, const float *__restrict__ mu, const uint32_t *__restrict__ stats, float eps,
    float tie_slack, const uint32_t *__restrict__ und_rows, const uint32_t *__restrict__ und_cont,
    uint32_t *__restrict__ assignments, uint32_t *__restrict__ assignments_prev,
    uint32_t *__restrict__ flagged, uint32_t *__restrict__ pairs, uint32_t *__restrict__ counters) {
  if (counters[kStopFlag] != 0u) return;
  const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint32_t total = counters[4];   // the rows stage 1 listed
  const float cmaxc = sqrtf(__uint_as_float(stats[0])) * 1.000001f;
  const float bmaxc = __uint_as_float(stats[1]);
  const float cmaxo = sqrtf(__uint_as_float(stats[2])) * 1.000001f;
  const float u = 5.9604645e-8f;
  // Every wave works through its rows on its own: no block barrier per row (round 5: three of them per row made the
  // four waves of a block wait for the slowest -- a row settled by exact chains takes microseconds).  The pair / full-
  // scan records wait in the wave's LDS buffer, one global atomic per kPend of them; the reassignments are counted in
  // lane 0 and added once.
  constexpr uint32_t kPend = 16;
  __shared__ uint32_t pbuf[4][kPend * 3], fbuf[4][kPend];
  volatile uint32_t *mypairs = pbuf[wave], *myflags = fbuf[wave];
  uint32_t np = 0, nf = 0, nchg = 0;   // (np, nf wave-uniform; nchg lane 0's)
  auto flush_pairs = [&]() {
    if (np == 0) return;
    uint32_t base = 0;
    if (lane == 0) base = atomicAdd(&counters[3], np);
    base = __shfl(base, 0);
    for (uint32_t i = lane; i < np * 3; i += 64) pairs[3 * (size_t)base + i] = mypairs[i];
    np = 0;
  };
  auto flush_flags = [&]() {
    if (nf == 0) return;
    uint32_t base = 0;
    if (lane == 0) base = atomicAdd(&counters[1], nf);
    base = __shfl(base, 0);
    for (uint32_t i = lane; i < nf; i += 64) flagged[base + i] = myflags[i];
    nf = 0;
  };
  for (uint32_t p = blockIdx.x * 4 + wave; p < total; p += gridDim.x * 4) {
    const bool live = true;
    const size_t slot = p;
    const uint32_t s = und_rows[slot];
    const uint32_t n = live ? und_cont[slot * (kWideCap + 1)] : 0u;
    const bool usable = n >= 1 && n <= (uint32_t)kWideCap;
    uint32_t cid[kWideCap];
#pragma unroll
    for (int i = 0; i < kWideCap; i++) cid[i] = (usable && (uint32_t)i < n) ? und_cont[slot * (kWideCap + 1) + 1 + i] : 0u;
    float acc[kWideCap];
#pragma unroll
    for (int i = 0; i < kWideCap; i++) acc[i] = 0.f;
    float xn2 = 0.f, xo2 = 0.f, x0 = 0.f;
    float xdm = 0.f, xab = 0.f;   // x.mu, sum |x_f mu_f|: the angular clamp (filter_common.hpp)
    const float *xr = samples + (size_t)s * D;
    // two 256-feature slices per trip, both slices' loads in flight before the first product: the stage is a chain
    // of round trips to L2 / the Infinity Cache (a 4-MB fp32 panel does not stay in L2), not arithmetic
    for (uint32_t f0 = lane * 4; f0 < DG; f0 += 512) {
      float xc[2][4];
      const bool second = f0 + 256 < DG;   // (DG is a multiple of 64: a lane's slice is whole or absent)
#pragma unroll
      for (int h2 = 0; h2 < 2; h2++) {
        const uint32_t f = f0 + 256u * h2;
        float x4[4] = {0.f, 0.f, 0.f, 0.f}, m4[4] = {0.f, 0.f, 0.f, 0.f};
        if (h2 == 0 || second) {
          if (FAST) {
            // (the row is read once: streamed past the caches, which the contenders' centroid rows need)
            const f32x4 a = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(xr + f)), b = *reinterpret_cast<const f32x4 *>(mu + f);
            x4[0] = a.x; x4[1] = a.y; x4[2] = a.z; x4[3] = a.w;
            m4[0] = b.x; m4[1] = b.y; m4[2] = b.z; m4[3] = b.w;
          } else {
#pragma unroll
            for (int e = 0; e < 4; e++) {
              x4[e] = (f + e) < D ? xr[f + e] : 0.f;
              m4[e] = mu[f + e];   // DG floats, zero beyond D
            }
          }
        }
#pragma unroll
        for (int e = 0; e < 4; e++) {
          xc[h2][e] = x4[e] - m4[e];
          xn2 = fmaf(xc[h2][e], xc[h2][e], xn2);
          xo2 = fmaf(x4[e], x4[e], xo2);
        }
        if (f == 0) x0 = x4[0];
      }
#pragma unroll
      for (int i = 0; i < kWideCap; i++) {
        if ((uint32_t)i < n && usable) {   // wave-uniform
          const float *cr = cfil + (size_t)cid[i] * DG + f0;
          const f32x4 ca = *reinterpret_cast<const f32x4 *>(cr);
          f32x4 cb = {0.f, 0.f, 0.f, 0.f};
          if (second) cb = *reinterpret_cast<const f32x4 *>(cr + 256);
          acc[i] = fmaf(xc[0][0], ca.x, fmaf(xc[0][1], ca.y, fmaf(xc[0][2], ca.z, fmaf(xc[0][3], ca.w, acc[i]))));
          acc[i] = fmaf(xc[1][0], cb.x, fmaf(xc[1][1], cb.y, fmaf(xc[1][2], cb.z, fmaf(xc[1][3], cb.w, acc[i]))));
        }
      }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      xn2 += __shfl_xor(xn2, off);
      xo2 += __shfl_xor(xo2, off);
#pragma unroll
      for (int i = 0; i < kWideCap; i++) acc[i] += __shfl_xor(acc[i], off);
    }
    if (tie_slack > 0.f) {   // angular: the row once more, from the caches (a loop of its own: the L2 passes keep their registers)
#pragma unroll 2
      for (uint32_t f = lane; f < D; f += 64) {
        const float x = xr[f], m = mu[f];
        xdm = fmaf(x, m, xdm);
        xab = fmaf(fabsf(x), fabsf(m), xab);
      }
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) {
        xdm += __shfl_xor(xdm, off);
        xab += __shfl_xor(xab, off);
      }
    }
    x0 = __shfl(x0, 0);
    float v1 = -INFINITY, v2 = -INFINITY, v3 = -INFINITY;
    uint32_t i1 = 0xFFFFFFFFu, i2 = 0xFFFFFFFFu;
#pragma unroll
    for (int i = 0; i < kWideCap; i++) {
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
    // Angular: what a decision rules out must stay below the clamp at product 1, its winner above the one at -1
    // (filter_common.hpp); the centroids that are not on the list scored below stage 1's cut-off <= its own limit.
    // The exact chains over a longer list follow the clamp themselves -- unless every product may sit at -1, where the
    // lowest index of ALL centroids wins: the full scan's.
    const bool angular = tie_slack > 0.f;
    const ClampLimits lim = clamp_limits(angular, xdm, dot_error((int)DG, xab), 0.5f * thr);
    const bool certain = insane || (in_range && ((v1 - v2) > thr) && (v2 < lim.hi) && (v1 > lim.lo));
    const bool two = !certain && in_range && ((v1 - v3) > thr) && (v3 < lim.hi) && (v1 > lim.lo) && i2 < K;
    const bool multi = live && !certain && !two && usable && (!angular || v1 > lim.lo);
    const bool pair_now = live && two, flag_now = live && !certain && !two && !multi;
    bool changed = false;
    if (live && certain && lane == 0) changed = commit_row(s, insane ? K : i1, assignments, assignments_prev);
    if (multi) {   // wave-uniform
      uint32_t mine = 0;
#pragma unroll
      for (int i = 0; i < kWideCap; i++) mine = ((int)lane == i) ? cid[i] : mine;
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
    if (lane == 0 && changed) nchg++;
    if (pair_now) {   // wave-uniform
      if (lane == 0) { mypairs[3 * np + 0] = s; mypairs[3 * np + 1] = i1; mypairs[3 * np + 2] = i2; }
      if (++np == kPend) flush_pairs();
    }
    if (flag_now) {   // wave-uniform
      if (lane == 0) myflags[nf] = s;
      if (++nf == kPend) flush_flags();
    }
  }
  flush_pairs();
  flush_flags();
  if (lane == 0 && nchg) atomicAdd(&counters[0], nchg);
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

size_t wide_cont_words(uint32_t N) { return (size_t)N * (kWideCap + 1); }

// stage 1 over every row, then over the rows it listed (their contenders); DG: a multiple of 64
hipError_t launch_lloyd_wide(const LloydArgs &a, const void *xg, const float *meta, uint32_t DG, const void *panelhi,
                             uint32_t *undecided, float *und_thr, uint32_t *und_cont, hipStream_t st, const CarryArgs *cy) {
  if (a.N == 0) return hipSuccess;
  if (DG % kWideBK != 0) return hipErrorInvalidValue;
  const uint32_t k_pad64 = (a.K_pad + 63u) / 64u * 64u;
  const uint32_t grid = (a.N + kWideRows - 1) / kWideRows;
  // Dynamic LDS beyond 64 KB is an opt-in that belongs to the CURRENT DEVICE's copy of the kernel (and shard workers
  // launch from their own threads): set for the instantiations this call launches, on every launch, like the other
  // launchers (lloyd.hip, update.hip, knn_f16.hip) -- and a refusal is the caller's error, not a launch failure later.
  {
    const int mode = !cy ? 0 : (!cy->row_list ? 2 : 3);
    hipError_t e = hipSuccess;
    if (mode == 0) e = hipFuncSetAttribute(reinterpret_cast<const void *>(&lloyd_wide_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kWideLds0);
    else if (mode == 2) e = hipFuncSetAttribute(reinterpret_cast<const void *>(&lloyd_wide_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kWideLds0);
    else e = hipFuncSetAttribute(reinterpret_cast<const void *>(&lloyd_wide_kernel<3>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kWideLds0);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void *>(&lloyd_wide_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kWideLds1);
    if (e != hipSuccess) return e;
  }
#define KMX_WIDE_LAUNCH(MODE, LDS, CY)                                                                                \
  hipLaunchKernelGGL((lloyd_wide_kernel<MODE>), dim3(grid), dim3(512), LDS, st, reinterpret_cast<const _Float16 *>(xg), \
                     reinterpret_cast<const float4 *>(meta), a.N, DG, reinterpret_cast<const _Float16 *>(panelhi),      \
                     k_pad64, a.K, a.stats, a.eps, a.tie_slack, a.assignments, a.assignments_prev, undecided, und_thr,  \
                     und_cont, a.counters, CY)
  if (!cy) {
    KMX_WIDE_LAUNCH(0, kWideLds0, CarryArgs());
  } else if (!cy->row_list) {   // every row, leaving bounds
    KMX_WIDE_LAUNCH(2, kWideLds0, *cy);
  } else {   // the rows carry_skip_kernel listed: blocks past the (device-side) end of the list leave at once
    KMX_WIDE_LAUNCH(3, kWideLds0, *cy);
  }
  // the undecided rows: the grid covers the worst case, blocks past the (device-side) end of the list leave at once
  KMX_WIDE_LAUNCH(1, kWideLds1, CarryArgs());
#undef KMX_WIDE_LAUNCH
  return hipGetLastError();
}

hipError_t launch_wide_contenders(int metric, const LloydArgs &a, const float *centroids, uint32_t DG,
                                  const uint32_t *und_rows, const uint32_t *und_cont, hipStream_t st) {
  if (a.N == 0) return hipSuccess;
  const bool fast = a.D == DG && (((uintptr_t)a.samples) & 15u) == 0;
  const uint32_t want = (a.N + 3) / 4;
  const dim3 grid(want < 4096u ? want : 4096u);
#define KMX_WC_LAUNCH(M, F)                                                                                           \
  hipLaunchKernelGGL((wide_contenders_kernel<M, F>), grid, dim3(256), 0, st, a.samples, a.D, DG, a.K, centroids, a.csqr, \
                     a.cfil, a.bias, a.mu, a.stats, a.eps, a.tie_slack, und_rows, und_cont, a.assignments,            \
                     a.assignments_prev, a.flagged, a.pairs, a.counters)
  if (metric == 0) { if (fast) KMX_WC_LAUNCH(0, true); else KMX_WC_LAUNCH(0, false); }
  else { if (fast) KMX_WC_LAUNCH(1, true); else KMX_WC_LAUNCH(1, false); }
#undef KMX_WC_LAUNCH
  return hipGetLastError();
}

}  // namespace kmx
