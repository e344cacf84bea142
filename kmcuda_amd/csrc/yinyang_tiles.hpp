// yinyang_tiles.hpp -- row loading and the f32 matrix-core tile product shared by the Yinyang
// kernels of yinyang_mfma.hip and yinyang_hint.hip (see yinyang_mfma.hip for the scheme).
#pragma once
#include "exact.hpp"
#include "exact_split.hpp"
#include "kernels.hpp"

namespace kmx {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr float kFltMax = 3.402823466e+38f;

// rows of the wave, centred: xb = x - mu; squared norms of x and of x - mu; x.mu
#define KMX_YY_LOAD_ROWS(samples_, row_, live_)                                                      \
  float xb[NK];                                                                                      \
  float xo2 = 0.f, xc2 = 0.f, xmu = 0.f;                                                             \
  {                                                                                                  \
    const float *src = (samples_) + (size_t)((live_) ? (row_) : 0) * D;                              \
    _Pragma("unroll") for (int j = 0; j < NK; j += 4) {                                              \
      float v[4], m[4];                                                                              \
      if (FAST) {                                                                                    \
        const f32x4 vv = *reinterpret_cast<const f32x4 *>(src + h * NK + j);                         \
        const f32x4 mm = *reinterpret_cast<const f32x4 *>(a.mu + h * NK + j);                        \
        v[0] = vv.x; v[1] = vv.y; v[2] = vv.z; v[3] = vv.w;                                          \
        m[0] = mm.x; m[1] = mm.y; m[2] = mm.z; m[3] = mm.w;                                          \
      } else {                                                                                       \
        _Pragma("unroll") for (int q = 0; q < 4; q++) {                                              \
          const uint32_t f = h * NK + j + q;                                                         \
          v[q] = (f < D) ? src[f] : 0.f;                                                             \
          m[q] = (f < D) ? a.mu[f] : 0.f;                                                            \
        }                                                                                            \
      }                                                                                              \
      _Pragma("unroll") for (int q = 0; q < 4; q++) {                                                \
        if (j + q >= NK) break;                                                                      \
        const float x = (live_) ? v[q] : 0.f;                                                        \
        const float xc = (live_) ? v[q] - m[q] : 0.f;                                                \
        xb[j + q] = xc;                                                                              \
        xo2 = fmaf(x, x, xo2);                                                                       \
        xc2 = fmaf(xc, xc, xc2);                                                                     \
        xmu = fmaf(x, m[q], xmu);                                                                    \
      }                                                                                              \
    }                                                                                                \
  }                                                                                                  \
  xo2 += __shfl_xor(xo2, 32);                                                                        \
  xc2 += __shfl_xor(xc2, 32);                                                                        \
  xmu += __shfl_xor(xmu, 32);                                                                        \
  const float *xrow = (samples_) + (size_t)((live_) ? (row_) : 0) * D;

// the same into variables the caller has declared (xb is declared here)
#define KMX_YY_LOAD_ROWS_INTO(samples_, row_, live_, xb, xo2, xc2, xmu)                              \
  float xb[NK];                                                                                      \
  {                                                                                                  \
    const float *src = (samples_) + (size_t)((live_) ? (row_) : 0) * D;                              \
    _Pragma("unroll") for (int j = 0; j < NK; j += 4) {                                              \
      float v[4], m[4];                                                                              \
      if (FAST) {                                                                                    \
        const f32x4 vv = *reinterpret_cast<const f32x4 *>(src + h * NK + j);                         \
        const f32x4 mm = *reinterpret_cast<const f32x4 *>(a.mu + h * NK + j);                        \
        v[0] = vv.x; v[1] = vv.y; v[2] = vv.z; v[3] = vv.w;                                          \
        m[0] = mm.x; m[1] = mm.y; m[2] = mm.z; m[3] = mm.w;                                          \
      } else {                                                                                       \
        _Pragma("unroll") for (int q = 0; q < 4; q++) {                                              \
          const uint32_t f = h * NK + j + q;                                                         \
          v[q] = (f < D) ? src[f] : 0.f;                                                             \
          m[q] = (f < D) ? a.mu[f] : 0.f;                                                            \
        }                                                                                            \
      }                                                                                              \
      _Pragma("unroll") for (int q = 0; q < 4; q++) {                                                \
        if (j + q >= NK) break;                                                                      \
        const float x = (live_) ? v[q] : 0.f;                                                        \
        const float xc = (live_) ? v[q] - m[q] : 0.f;                                                \
        xb[j + q] = xc;                                                                              \
        xo2 = fmaf(x, x, xo2);                                                                       \
        xc2 = fmaf(xc, xc, xc2);                                                                     \
        xmu = fmaf(x, m[q], xmu);                                                                    \
      }                                                                                              \
    }                                                                                                \
  }                                                                                                  \
  xo2 += __shfl_xor(xo2, 32);                                                                        \
  xc2 += __shfl_xor(xc2, 32);                                                                        \
  xmu += __shfl_xor(xmu, 32);



// Rows of a wave (32 rows, lane (col, h) owns features [h NK, h NK + NK) of row col) brought in COALESCED and
// handed to their owners through a per-wave LDS scratch, four chunks of NK / 4 features per half:
// KMX_YY_LOAD_ROWS above reads 16 bytes per lane from 64 different rows per instruction -- 64 cache lines, of
// which the next seven instructions want the rest, from a 32-KB L1 that eight waves share (round 2 counters:
// the prologue of yy_local_hint_kernel took 27 % of its cycles).  Here DP / 16 neighbouring lanes read one
// row's DP bytes (both halves' CH floats), every line is fetched once.
// scratch: LDS float index of this wave's 32 x (DP / 4 + 4) floats; row_of(r): global row number of the wave's
// row r (valid in every lane); f(c, k, v): called by the owner lane with the four values of features
// h NK + c CH + 4 k .. + 3.
template <int DP, typename RowOf, typename F>
__device__ __forceinline__ void yy_rows_staged(const float *__restrict__ samples, float *lds, uint32_t scratch, int lane,
                                               RowOf row_of, F f) {
  constexpr int NK = DP / 2, CH = NK / 4, LPR = DP / 16, RPI = 64 / LPR, NI = 32 / RPI, ROWS = 2 * CH + 4;
  typedef float v4 __attribute__((ext_vector_type(4)));
  const int col = lane & 31, h = lane >> 5;
  const int q = lane % LPR, rr = lane / LPR;
  const int seg = q < LPR / 2 ? 0 : 1, qq = q - seg * (LPR / 2);
  // every chunk's loads are issued before the first is used: one memory round trip, not four
  v4 st[4][NI];
#pragma unroll
  for (int c = 0; c < 4; c++) {
#pragma unroll
    for (int i = 0; i < NI; i++) {
      const int r = i * RPI + rr;
      const uint32_t row = row_of(r);
      st[c][i] = *reinterpret_cast<const v4 *>(samples + (size_t)row * DP + seg * NK + c * CH + qq * 4);
    }
  }
#pragma unroll
  for (int c = 0; c < 4; c++) {   // unrolled: the owners' callbacks index registers with c and k
#pragma unroll
    for (int i = 0; i < NI; i++) {
      const int r = i * RPI + rr;
      *reinterpret_cast<v4 *>(lds + scratch + r * ROWS + seg * CH + qq * 4) = st[c][i];
    }
    // (one wave: its LDS operations complete in order, the compiler waits for the reads' data)
#pragma unroll
    for (int k = 0; k < CH / 4; k++) {
      const v4 v = *reinterpret_cast<const v4 *>(lds + scratch + col * ROWS + h * CH + 4 * k);
      f(c, k, v);
    }
  }
}

#define KMX_YY_MFMA_TILE(acc_, buf_)                                                                 \
  f32x16 acc_;                                                                                       \
  {                                                                                                  \
    const float *bb = bias_ptr(buf_) + 4 * h;                                                        \
    _Pragma("unroll") for (int g4 = 0; g4 < 4; g4++) {                                               \
      const f32x4 b4 = *reinterpret_cast<const f32x4 *>(bb + 8 * g4);                                \
      acc_[4 * g4 + 0] = b4.x; acc_[4 * g4 + 1] = b4.y; acc_[4 * g4 + 2] = b4.z; acc_[4 * g4 + 3] = b4.w; \
    }                                                                                                \
    const float *arow = tile_ptr(buf_) + col * LDW + h * NK;                                         \
    _Pragma("unroll") for (int j = 0; j < NK / 4; j++) {                                             \
      const f32x4 a4 = *reinterpret_cast<const f32x4 *>(arow + 4 * j);                               \
      acc_ = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.x, xb[4 * j + 0], acc_, 0, 0, 0);               \
      acc_ = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.y, xb[4 * j + 1], acc_, 0, 0, 0);               \
      acc_ = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.z, xb[4 * j + 2], acc_, 0, 0, 0);               \
      acc_ = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.w, xb[4 * j + 3], acc_, 0, 0, 0);               \
    }                                                                                                \
  }

}  // namespace kmx
