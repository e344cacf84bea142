// yinyang_init.hip -- kmeans_yy_init (reference: src/kmeans.cu:431-485) with the matrix cores in front of
// the reference's exact arithmetic, the exact chains fed from registers and LDS.
//
//   bounds[0][s]     = exact distance to the row's own centroid                         (kmeans.cu:474-476)
//   bounds[1 + g][s] = min over the group's centroids other than the row's own of the exact distance
//                                                                                       (kmeans.cu:477-481)
// A minimum does not depend on the visiting order, so the panel is streamed GROUP-SORTED (yy_configure:
// groups padded to whole 4-slot chunks, a group never crosses a 32-slot tile boundary without a new start
// flag), each half-wave keeps a running top-3 of the f32 matrix-core scores of the current group, and at the
// group boundary the 1-2 contenders are queued for the exact chain (all members, when three or more are
// within the error bound).  Every stored number is the reference's exact arithmetic.
//
// Round 1's kernel (202 ms per 8M x 256 rows, K = 1024, G = 102; counters: one L1 access per cycle per CU)
// gathered the sample row and four centroid rows of every chain 16 bytes per lane from global memory -- five
// tag look-ups per lane and chain step -- and, unrolled over the tile's chunks, came to 90 KB of code.  Here
//   * the tiles hold the ORIGINAL centroid values (an f32 product needs no centring: its error bound is
//     ~1e-5 of a squared distance either way), so a contender's row is read from the tile it was scored in;
//   * the wave's matrix-core B operand IS the original sample row, 128 values per lane: the chain takes its
//     x values from those registers (dynamic index into 32-wide register vectors, the loops stay rolled).
// A flush therefore touches no global memory.  A queue entry names an LDS row; it is settled before the
// buffer it points into is overwritten (end of the iteration after the one that read the tile).
#include "exact_split.hpp"
#include "kernels.hpp"

namespace kmx {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr float kFltMaxI = 3.402823466e+38f;

// Four exact chains at once, x from the register vectors xv (this lane's NK features), the four centroid
// rows from LDS (float index of the row's first value), as a two-stage pipeline (exact_split.hpp, ExactPipe4):
// the lower half-wave runs features [0, NK) of the NEW batch, the upper one finishes [NK, D) of the
// PREVIOUS batch.  Same operations in the same order per candidate as metric_abstraction.h:73-86 / :193-205.
template <int NK, int VN, int METRIC, bool FAST, typename XV>
__device__ __forceinline__ void exact_chain4_lds_pipe(const XV (&xv)[NK / VN], const float *lds, const uint32_t (&crow)[4],
                                                      uint32_t D, int h, int col, ExactPipe4 &st, float (&dist_old)[4]) {
  float acc[4], corr[4];
#pragma unroll
  for (int i = 0; i < 4; i++) {
    acc[i] = h ? st.acc[i] : 0.f;
    corr[i] = h ? st.corr[i] : 0.f;
  }
  const int nvalid = (int)D - h * NK < 0 ? 0 : ((int)D - h * NK > NK ? NK : (int)D - h * NK);
#pragma unroll
  for (int b = 0; b < NK / VN; b++) {
#pragma unroll 1
    for (int jj = 0; jj < VN; jj += 4) {
      const int j = b * VN + jj;
      float x[4], cv[4][4];
#pragma unroll
      for (int q = 0; q < 4; q++) x[q] = xv[b][jj + q];   // jj is wave-uniform: indexed register read
#pragma unroll
      for (int i = 0; i < 4; i++) {
        const f32x4 v = *reinterpret_cast<const f32x4 *>(lds + crow[i] + h * NK + j);
        cv[i][0] = v.x; cv[i][1] = v.y; cv[i][2] = v.z; cv[i][3] = v.w;
      }
#pragma unroll
      for (int q = 0; q < 4; q++) {
        float y[4];
        if (METRIC == 0) {
          float d[4];
#pragma unroll
          for (int i = 0; i < 4; i++) d[i] = x[q] - cv[i][q];
          sqfma_rd4(d, corr, y);
        } else {
          const float bb[4] = {cv[0][q], cv[1][q], cv[2][q], cv[3][q]};
          fma_rd4(x[q], bb, corr, y);
        }
        const bool on = FAST || (j + q < nvalid);
#pragma unroll
        for (int i = 0; i < 4; i++) {
          const float t = acc[i] + y[i];
          const float nc = y[i] - (t - acc[i]);
          acc[i] = on ? t : acc[i];
          corr[i] = on ? nc : corr[i];
        }
      }
    }
  }
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const float total = __shfl(acc[i], col + 32);   // the previous batch, finished by the upper half
    dist_old[i] = METRIC == 0 ? sqrtf(total) : angular_from_prod(total);
    st.acc[i] = __shfl(acc[i], col);                // the new batch after its first half
    st.corr[i] = __shfl(corr[i], col);
  }
}

// pids[slot]  centroid id of the slot or 0xFFFFFFFF (padding)
// pmeta[8*tile + ch]  (group << 1) | starts_new_group, for the 4-slot chunk ch of the tile
template <int DP, int METRIC, bool FAST>
__global__ __launch_bounds__(256, 2) void yy_init_lds_kernel(YyArgs a) {
  constexpr int NK = DP / 2, LDW = DP + 4, TILE = 32 * LDW, NST = (8 * DP + 255) / 256;
  constexpr int VN = NK < 32 ? NK : 32, NV = NK / VN;
  typedef float xvec __attribute__((ext_vector_type(VN)));
  extern __shared__ __attribute__((aligned(16))) float lds[];
  auto tile_ptr = [&](int buf) { return lds + buf * TILE; };
  auto bias_ptr = [&](int buf) { return lds + 2 * TILE + buf * 32; };
  auto id_ptr = [&](int buf) { return reinterpret_cast<uint32_t *>(lds + 2 * TILE + 64) + buf * 32; };
  auto meta_ptr = [&](int buf) { return reinterpret_cast<uint32_t *>(lds + 2 * TILE + 128) + buf * 8; };

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, col = lane & 31, h = lane >> 5;
  const uint32_t D = a.D, K = a.K, G = a.G, len = a.len;
  const uint32_t s = blockIdx.x * 128u + wave * 32u + col;
  const bool live = s < len;
  const float *xrow = a.samples + (size_t)(live ? s : 0) * D;

  // B operand = my half of the ORIGINAL row
  xvec xv[NV];
  float xo2 = 0.f;
#pragma unroll
  for (int j = 0; j < NK; j += 4) {
    float v[4];
    if (FAST) {
      const f32x4 vv = *reinterpret_cast<const f32x4 *>(xrow + h * NK + j);
      v[0] = vv.x; v[1] = vv.y; v[2] = vv.z; v[3] = vv.w;
    } else {
#pragma unroll
      for (int q = 0; q < 4; q++) v[q] = ((uint32_t)(h * NK + j + q) < D) ? xrow[h * NK + j + q] : 0.f;
    }
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const float x = live ? v[q] : 0.f;
      xv[(j + q) / VN][(j + q) % VN] = x;
      xo2 = fmaf(x, x, xo2);
    }
  }
  xo2 += __shfl_xor(xo2, 32);

  const uint32_t nearest = live ? a.assignments[s] : 0xFFFFFFFFu;

  // two scores closer than thr cannot be ordered by the filter (with mu = 0: the score of
  // centroid c is x.c - 0.5 ||c||^2 (L2) or x.c (angular), off by at most 2 eps (||x|| Cmax + Bmax))
  const float cmaxo = sqrtf(__uint_as_float(a.stats[2])) * 1.000001f;
  const float bmaxo = METRIC == 0 ? 0.5f * cmaxo * cmaxo : 0.f;
  const float xo = sqrtf(xo2) * 1.0001f;
  const float u = 5.9604645e-8f;
  float thr = 2.0f * (2.0f * a.eps * (xo * cmaxo + bmaxo)) * 1.01f;
  if (METRIC == 0) thr += 16.0f * u * (xo + cmaxo) * (xo + cmaxo);
  else thr += 16.0f * u * xo * cmaxo + 2e-6f;

  f32x4 stage[NST];
  float bstage = 0.f;
  uint32_t istage = 0xFFFFFFFFu, mstage = 0;
  auto stage_load = [&](uint32_t tile) {
    const float *src = a.pfil + (size_t)tile * 32 * DP;
#pragma unroll
    for (int i = 0; i < NST; i++) {
      const int q = tid + i * 256;
      if (q < 8 * DP) stage[i] = reinterpret_cast<const f32x4 *>(src)[q];
    }
    if (tid < 32) {
      bstage = a.pbias[tile * 32 + tid];
      istage = a.pids[tile * 32 + tid];
    }
    if (tid < 8) mstage = a.pmeta[tile * 8 + tid];
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
    if (tid < 32) {
      bias_ptr(buf)[tid] = bstage;
      id_ptr(buf)[tid] = istage;
    }
    if (tid < 8) meta_ptr(buf)[tid] = mstage;
  };

  // queue of (group, LDS row) distance evaluations; a group's entries are adjacent, its minimum is carried
  // across flushes and stored when the next group's first entry is replayed (or at the end)
  uint32_t qa[4] = {0, 0, 0, 0}, qg[4] = {0, 0, 0, 0};
  int qn = 0;
  uint32_t carry_g = 0xFFFFFFFFu;
  float carry_min = kFltMaxI;
  auto store_carry = [&]() __attribute__((always_inline)) {
    if (carry_g != 0xFFFFFFFFu && live && h == 0) a.bounds[(size_t)len * (1 + carry_g) + s] = carry_min;
  };
  // two-batch pipeline: a flush starts the queued batch and completes -- and replays -- the one before it
  uint32_t pqa[4] = {0, 0, 0, 0}, pqg[4] = {0, 0, 0, 0};
  int pqn = 0;
  ExactPipe4 pipe;
  auto flush = [&]() __attribute__((always_inline)) {  // wave-uniform call
    uint32_t crow[4];
#pragma unroll
    for (int i = 0; i < 4; i++) crow[i] = h ? (i < pqn ? pqa[i] : 0u) : (i < qn ? qa[i] : 0u);
    float dist[4];
    exact_chain4_lds_pipe<NK, VN, METRIC, FAST>(xv, lds, crow, D, h, col, pipe, dist);
#pragma unroll
    for (int i = 0; i < 4; i++) {
      if (i < pqn) {
        const uint32_t g = pqg[i] & 0x7FFFFFFFu;
        if (g != carry_g) {
          store_carry();
          carry_g = g;
          carry_min = kFltMaxI;
        }
        if (!(pqg[i] >> 31) && dist[i] < carry_min) carry_min = dist[i];   // kmeans.cu:477-481 (NaN never "less")
      }
      pqa[i] = qa[i];
      pqg[i] = qg[i];
    }
    pqn = qn;
    qn = 0;
  };
  auto drain = [&]() __attribute__((always_inline)) {  // wave-uniform call: nothing queued, nothing pending afterwards
#pragma unroll 1
    for (int r = 0; r < 2; r++)
      if (__ballot(qn > 0 || pqn > 0) != 0ull) flush();
  };
  auto enqueue = [&](uint32_t g, uint32_t addr, bool on) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 4; i++)
      if (on && i == qn) {
        qa[i] = addr;
        qg[i] = g;
      }
    if (on) qn++;
  };

  // upper bound: exact distance to the row's own centroid (kmeans.cu:474-476); stays FLT_MAX if the
  // row has none (NaN row) or its centroid is in no group (NaN centroid)
  float upper = kFltMaxI;
  {
    const bool has = live && nearest < K && a.groups[nearest] < G;
    if (__ballot(has) != 0ull) {
      const float *crow[4];
#pragma unroll
      for (int i = 0; i < 4; i++) crow[i] = a.centroids + (size_t)(has ? nearest : 0) * D;
      float dist[4];
      exact_distance4<NK, METRIC, FAST>(xrow, crow, D, h, col, dist, 1, has ? 1 : 0);
      if (has) upper = dist[0];
    }
  }

  // running top-3 (by score = smallest distance first) of the CURRENT group in this half-wave; r1, r2: the
  // LDS rows (float index) of the best two
  float v1 = -INFINITY, v2 = -INFINITY, v3 = -INFINITY;
  uint32_t r1 = 0xFFFFFFFFu, r2 = 0xFFFFFFFFu;
  uint32_t cur_group = 0xFFFFFFFFu;
  auto insert = [&](float v, uint32_t idx) {
    const bool g1 = v > v1, g2 = v > v2, g3 = v > v3;
    v3 = g2 ? v2 : (g3 ? v : v3);
    r2 = g1 ? r1 : (g2 ? idx : r2);
    v2 = g1 ? v1 : (g2 ? v : v2);
    r1 = g1 ? idx : r1;
    v1 = g1 ? v : v1;
  };
  uint32_t cur_row0 = 0;   // first tile row of the current (part of a) group
  auto finalize_group = [&](int buf, uint32_t end_row) __attribute__((always_inline)) {  // wave-uniform call
    if (cur_group == 0xFFFFFFFFu) return;
    {  // merge the partner half-wave's top-3; the lower half-wave's view wins (both lanes of a pair must
       // hold the SAME contenders: they evaluate one chain between them)
      const float pv1 = __shfl_xor(v1, 32), pv2 = __shfl_xor(v2, 32), pv3 = __shfl_xor(v3, 32);
      const uint32_t pr1 = __shfl_xor(r1, 32), pr2 = __shfl_xor(r2, 32);
      insert(pv1, pr1);
      insert(pv2, pr2);
      insert(pv3, 0xFFFFFFFFu);
      v1 = __shfl(v1, col); v2 = __shfl(v2, col); v3 = __shfl(v3, col);
      r1 = __shfl(r1, col); r2 = __shfl(r2, col);
    }
    const bool has1 = live && r1 != 0xFFFFFFFFu;
    const bool sure1 = has1 && ((v1 - v2) > thr);                      // NaN gap => not sure
    const bool sure2 = has1 && !sure1 && r2 != 0xFFFFFFFFu && ((v1 - v3) > thr);
    const bool scan = has1 && !sure1 && !sure2;
    // one loop, one flush site: the best, the second best, then -- for the rows with three or more contenders
    // within the error bound -- every member of this (part of the) group, which lies in the current tile
    const uint32_t nscan = __ballot(scan) != 0ull ? end_row - cur_row0 : 0u;
#pragma unroll 1
    for (uint32_t kk = 0; kk < 2u + nscan; kk++) {
      uint32_t addr;
      bool on;
      uint32_t tag = cur_group;
      if (kk == 0) {
        // (no member other than the row's own centroid, or an empty group: the bound stays FLT_MAX -- a NULL
        //  entry, replayed in order like the others so that a group's carry is never interrupted)
        addr = has1 ? r1 : 0u;
        on = live && !scan;
        if (!has1) tag |= 0x80000000u;
      } else if (kk == 1) {
        addr = r2;
        on = sure2;
      } else {
        const uint32_t row = cur_row0 + (kk - 2u);
        const uint32_t id = id_ptr(buf)[row];
        addr = (uint32_t)(buf * TILE) + row * LDW;
        on = scan && id != 0xFFFFFFFFu && id != nearest;
      }
      if (__ballot(on && qn > 3) != 0ull) flush();
      enqueue(tag, addr, on);
    }
    v1 = v2 = v3 = -INFINITY;
    r1 = r2 = 0xFFFFFFFFu;
  };

  const uint32_t ntiles = a.nslots / 32;
  stage_load(0);
  stage_store(0);
  __syncthreads();
  for (uint32_t t = 0; t < ntiles; t++) {
    const int buf = t & 1;
    f32x16 acc;
    {
      const float *bb = bias_ptr(buf) + 4 * h;
#pragma unroll
      for (int g4 = 0; g4 < 4; g4++) {
        const f32x4 b4 = *reinterpret_cast<const f32x4 *>(bb + 8 * g4);
        acc[4 * g4 + 0] = b4.x; acc[4 * g4 + 1] = b4.y; acc[4 * g4 + 2] = b4.z; acc[4 * g4 + 3] = b4.w;
      }
      const float *arow = tile_ptr(buf) + col * LDW + h * NK;
#pragma unroll
      for (int j = 0; j < NK / 4; j++) {
        const f32x4 a4 = *reinterpret_cast<const f32x4 *>(arow + 4 * j);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.x, xv[(4 * j + 0) / VN][(4 * j + 0) % VN], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.y, xv[(4 * j + 1) / VN][(4 * j + 1) % VN], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.z, xv[(4 * j + 2) / VN][(4 * j + 2) % VN], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.w, xv[(4 * j + 3) / VN][(4 * j + 3) % VN], acc, 0, 0, 0);
      }
    }
    // The eight 4-slot chunks as a ROLLED loop (the scores are read with a wave-uniform register index):
    // unrolled, the group-closing code below -- two flush sites, each a whole exact-chain loop -- was
    // instantiated nine times and the kernel came to 90 KB of code, more than the instruction cache.
#pragma unroll 1
    for (int ch = 0; ch <= 8; ch++) {
      // ch == 8: the end of the tile closes its last group (yy_configure: the next tile starts a new group or
      // a new part of this one), so that every queue entry names a row of the tile it was queued in
      const uint32_t meta = ch < 8 ? meta_ptr(buf)[ch] : 0xFFFFFFFFu;
      if (meta & 1u) {  // this chunk starts a new group (or a new tile inside one): close the previous one
        finalize_group(buf, 4u * ch);
        cur_group = ch < 8 ? meta >> 1 : 0xFFFFFFFFu;
        cur_row0 = 4u * ch;
      }
      if (ch == 8) break;
      const int rbase = 4 * (ch >> 1);
      float sc[4];
#pragma unroll
      for (int q = 0; q < 4; q++) sc[q] = acc[rbase + q];   // ch is wave-uniform: indexed register read
      if ((ch & 1) == h) {
#pragma unroll
        for (int q = 0; q < 4; q++) {
          const uint32_t row = 8 * (ch >> 1) + q + 4 * h;
          const uint32_t id = id_ptr(buf)[row];
          const float v = (id != 0xFFFFFFFFu && id != nearest) ? sc[q] : -INFINITY;
          insert(v, (uint32_t)(buf * TILE) + row * LDW);
        }
      }
    }
    if (t + 1 < ntiles) {
      // tile t + 1 goes into the buffer tile t - 1 was read from.  One flush per tile, here: it completes the
      // batch started at the end of the previous tile (rows of tile t - 1) and starts the rows queued during
      // this one (tile t, alive for another iteration).  The same work in every wave of the block at the same
      // point: none waits for another's chains at the barrier below.  (A flush inside the tile only happens when
      // a row collects more than four entries; whatever it leaves pending names rows of tile t - 1 or t, and the
      // flush here settles it.)
      // (The tile is fetched here, not a tile ahead through registers: 32 staging registers live across
      // the matrix-core loop and the chains spilled; the block's second resident partner covers the trip.)
      if (__ballot(qn > 0 || pqn > 0) != 0ull) {
        flush();
      }
      stage_load(t + 1);
      __syncthreads();   // the other waves' chains may still be reading that buffer
      stage_store(buf ^ 1);
    }
    __syncthreads();
  }
  drain();
  store_carry();
  if (live && h == 0) a.bounds[s] = upper;
}

// group-sorted padded panel of the ORIGINAL centroid values, zero padded to DP; bias = -0.5 ||c||^2 (L2) or 0
// (angular), -inf for padding slots
__global__ void yy_orig_panel_kernel(int metric, const float *__restrict__ centroids, uint32_t D, uint32_t DP,
                                     const uint32_t *__restrict__ pids, float *__restrict__ pfil,
                                     float *__restrict__ pbias) {
  const uint32_t slot = blockIdx.x;
  const uint32_t id = pids[slot];
  double n2 = 0.0;
  for (uint32_t f = threadIdx.x; f < DP; f += 64) {
    const float v = (id != 0xFFFFFFFFu && f < D) ? centroids[(size_t)id * D + f] : 0.f;
    pfil[(size_t)slot * DP + f] = v;
    n2 += (double)v * (double)v;
  }
  for (int o = 32; o > 0; o >>= 1) n2 += __shfl_xor(n2, o);
  if (threadIdx.x == 0) pbias[slot] = id != 0xFFFFFFFFu ? (metric == 0 ? (float)(-0.5 * n2) : 0.f) : -INFINITY;
}

template <int DP, int METRIC>
static hipError_t launch_init_lds_t(const YyArgs &a, hipStream_t st) {
  const size_t lds_bytes = (2 * 32 * (DP + 4) + 64 + 64 + 16) * sizeof(float);
  const uint32_t grid = (a.len + 127) / 128;
  if (a.D == (uint32_t)DP)
    hipLaunchKernelGGL((yy_init_lds_kernel<DP, METRIC, true>), dim3(grid), dim3(256), lds_bytes, st, a);
  else
    hipLaunchKernelGGL((yy_init_lds_kernel<DP, METRIC, false>), dim3(grid), dim3(256), lds_bytes, st, a);
  return hipGetLastError();
}

hipError_t launch_yy_init_lds(int metric, const YyArgs &a, hipStream_t st) {
  if (a.len == 0) return hipSuccess;
  switch (a.DP) {
    case 8: return metric == 0 ? launch_init_lds_t<8, 0>(a, st) : launch_init_lds_t<8, 1>(a, st);
    case 16: return metric == 0 ? launch_init_lds_t<16, 0>(a, st) : launch_init_lds_t<16, 1>(a, st);
    case 32: return metric == 0 ? launch_init_lds_t<32, 0>(a, st) : launch_init_lds_t<32, 1>(a, st);
    case 64: return metric == 0 ? launch_init_lds_t<64, 0>(a, st) : launch_init_lds_t<64, 1>(a, st);
    case 128: return metric == 0 ? launch_init_lds_t<128, 0>(a, st) : launch_init_lds_t<128, 1>(a, st);
    case 256: return metric == 0 ? launch_init_lds_t<256, 0>(a, st) : launch_init_lds_t<256, 1>(a, st);
    default: return hipErrorInvalidValue;
  }
}

hipError_t launch_yy_orig_panel(int metric, const float *centroids, uint32_t D, uint32_t DP, const uint32_t *pids,
                                uint32_t nslots, float *pfil, float *pbias, hipStream_t st) {
  if (nslots == 0) return hipSuccess;
  hipLaunchKernelGGL(yy_orig_panel_kernel, dim3(nslots), dim3(64), 0, st, metric, centroids, D, DP, pids, pfil, pbias);
  return hipGetLastError();
}

}  // namespace kmx
