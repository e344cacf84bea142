// transpose.hip -- out[c][r] = in[r][c] for 4-byte elements (fp32 or packed half2).
// Reference: src/transpose.cu:16-54 (32x32 tile, 32x8 threads, managed-memory staging).
// gfx950 version: 64x64 tile per 256-thread block through LDS (rows padded to 65 words).  When both
// extents are multiples of 4 and both buffers 16-byte aligned (every shape the library itself
// transposes: N is padded by the callers, D is a multiple of 4 on the filtered paths) each thread
// moves 16 bytes per global access on BOTH sides -- a float4 along the input row in, a float4 along
// the output row out -- and the LDS accesses in between are 4-byte ones whose bank is
// (row + column) mod 64: 64 distinct banks per wave in both directions.  Any other shape takes the
// 4-byte kernel.  HBM-bound: 2 * rows * cols * 4 bytes.
#include "kernels.hpp"

namespace kmx {

typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void transpose_kernel(const float *__restrict__ in, uint32_t rows, uint32_t cols,
                                                        float *__restrict__ out) {
  __shared__ float tile[64][65];
  const uint32_t tiles_c = (cols + 63) / 64;
  const uint32_t bx = (blockIdx.x % tiles_c) * 64, by = (blockIdx.x / tiles_c) * 64;  // column / row base
  const uint32_t tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
#pragma unroll 4
  for (uint32_t j = ty; j < 64; j += 4) {
    const uint32_t r = by + j, c = bx + tx;
    if (r < rows && c < cols) tile[j][tx] = in[(size_t)r * cols + c];
  }
  __syncthreads();
#pragma unroll 4
  for (uint32_t j = ty; j < 64; j += 4) {
    const uint32_t c = bx + j, r = by + tx;
    if (r < rows && c < cols) out[(size_t)c * rows + r] = tile[tx][j];
  }
}

// rows % 4 == 0, cols % 4 == 0, 16-byte aligned buffers
__global__ __launch_bounds__(256) void transpose_vec4_kernel(const float *__restrict__ in, uint32_t rows, uint32_t cols,
                                                             float *__restrict__ out) {
  __shared__ float tile[64][65];
  const uint32_t tiles_c = (cols + 63) / 64;
  const uint32_t bx = (blockIdx.x % tiles_c) * 64, by = (blockIdx.x / tiles_c) * 64;
  const uint32_t q4 = threadIdx.x & 15, line = threadIdx.x >> 4;   // 16 float4 per 64-element line, 16 lines per pass
#pragma unroll
  for (uint32_t p = 0; p < 4; p++) {
    const uint32_t j = line + 16 * p, r = by + j, c = bx + 4 * q4;
    if (r < rows && c < cols) {
      const f32x4 v = *reinterpret_cast<const f32x4 *>(in + (size_t)r * cols + c);
      tile[j][4 * q4 + 0] = v.x; tile[j][4 * q4 + 1] = v.y; tile[j][4 * q4 + 2] = v.z; tile[j][4 * q4 + 3] = v.w;
    }
  }
  __syncthreads();
#pragma unroll
  for (uint32_t p = 0; p < 4; p++) {
    const uint32_t j = line + 16 * p, c = bx + j, r = by + 4 * q4;   // output row c, four input rows r .. r + 3
    if (c < cols && r < rows) {
      f32x4 v;
      v.x = tile[4 * q4 + 0][j]; v.y = tile[4 * q4 + 1][j]; v.z = tile[4 * q4 + 2][j]; v.w = tile[4 * q4 + 3][j];
      *reinterpret_cast<f32x4 *>(out + (size_t)c * rows + r) = v;
    }
  }
}

hipError_t launch_transpose(const float *in, uint32_t rows, uint32_t cols, float *out, hipStream_t st) {
  if (rows == 0 || cols == 0) return hipSuccess;
  // 1-D grid over tiles: either dimension may exceed the 65535 limit of grid.y
  const uint64_t tiles = (uint64_t)((cols + 63) / 64) * ((rows + 63) / 64);
  if (tiles > 0x7FFFFFFFull) return hipErrorInvalidValue;
  const bool vec = rows % 4 == 0 && cols % 4 == 0 && (reinterpret_cast<uintptr_t>(in) & 15) == 0 &&
                   (reinterpret_cast<uintptr_t>(out) & 15) == 0;
  if (vec) hipLaunchKernelGGL(transpose_vec4_kernel, dim3((uint32_t)tiles), dim3(256), 0, st, in, rows, cols, out);
  else hipLaunchKernelGGL(transpose_kernel, dim3((uint32_t)tiles), dim3(256), 0, st, in, rows, cols, out);
  return hipGetLastError();
}

}  // namespace kmx
