// seeding.hip -- exact-arithmetic helper kernels around the hot path:
//   kmpp_step          (reference: kmeans.cu:42-67 kmeans_plus_plus): d[s] = min(d[s], dist(s, newest c))
//   member_distances   (reference: kmeans.cu:674-691 kmeans_calc_average_distance, per-sample part)
// Distances use the reference's exact arithmetic (exact.hpp) so that the host-side chooser sees
// the very same floats as the reference's and picks the same seeds.
#include <hip/hip_fp16.h>

#include "exact.hpp"
#include "kernels.hpp"

namespace kmx {

template <int METRIC>
__global__ void kmpp_step_kernel(const float *__restrict__ samples, uint32_t N, uint32_t D,
                                 const float *__restrict__ centroid, uint32_t cc, float *__restrict__ dists) {
  const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= N) return;
  const float *x = samples + (size_t)s * D;
  float dist = 0.f;
  if (x[0] == x[0]) dist = distance_vv<METRIC>(x, centroid, D);  // kmeans.cu:53-56
  if (cc == 1 || dist < dists[s]) dists[s] = dist;                // :57-62
}

hipError_t launch_kmpp_step(int metric, const float *samples, uint32_t N, uint32_t D, const float *centroid,
                            uint32_t cc, float *dists, hipStream_t st) {
  const dim3 grid((N + 127) / 128), block(128);
  if (metric == 0)
    hipLaunchKernelGGL((kmpp_step_kernel<0>), grid, block, 0, st, samples, N, D, centroid, cc, dists);
  else
    hipLaunchKernelGGL((kmpp_step_kernel<1>), grid, block, 0, st, samples, N, D, centroid, cc, dists);
  return hipGetLastError();
}

template <int METRIC>
__global__ void member_distances_kernel(const float *__restrict__ samples, uint32_t N, uint32_t D,
                                        const float *__restrict__ centroids, const uint32_t *__restrict__ assignments,
                                        uint32_t K, float *__restrict__ dists) {
  const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= N) return;
  const uint32_t a = assignments[s];
  dists[s] = a < K ? distance_vv<METRIC>(samples + (size_t)s * D, centroids + (size_t)a * D, D) : NAN;
}

hipError_t launch_member_distances(int metric, const float *samples, uint32_t N, uint32_t D,
                                   const float *centroids, const uint32_t *assignments, uint32_t K,
                                   float *dists, hipStream_t st) {
  const dim3 grid((N + 127) / 128), block(128);
  if (metric == 0)
    hipLaunchKernelGGL((member_distances_kernel<0>), grid, block, 0, st, samples, N, D, centroids, assignments, K,
                       dists);
  else
    hipLaunchKernelGGL((member_distances_kernel<1>), grid, block, 0, st, samples, N, D, centroids, assignments, K,
                       dists);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------
// fp16x2 boundary (reference: fp_abstraction.h:100-182, kmcuda.h:107-108): the public buffers
// hold IEEE halves (two per 32-bit "feature").  This implementation computes the fp16 path as
// "the fp32 arithmetic applied to the half values" -- halves are widened exactly on the way in,
// centroids are rounded to half (RN) after every update, exactly where the reference stores them
// as half2 -- instead of accumulating in half2 like the reference (tolerance).
// ---------------------------------------------------------------------------------------
__global__ void half_to_float_kernel(const __half *__restrict__ src, size_t n, float *__restrict__ dst) {
  const size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 8;
  if (i + 8 <= n) {
    const uint4 raw = *reinterpret_cast<const uint4 *>(src + i);
    const __half2 *h2 = reinterpret_cast<const __half2 *>(&raw);
    float4 lo, hi;
    float2 a = __half22float2(h2[0]), b = __half22float2(h2[1]), c = __half22float2(h2[2]), d = __half22float2(h2[3]);
    lo.x = a.x; lo.y = a.y; lo.z = b.x; lo.w = b.y;
    hi.x = c.x; hi.y = c.y; hi.z = d.x; hi.w = d.y;
    *reinterpret_cast<float4 *>(dst + i) = lo;
    *reinterpret_cast<float4 *>(dst + i + 4) = hi;
  } else {
    for (size_t j = i; j < n; j++) dst[j] = __half2float(src[j]);
  }
}

__global__ void float_to_half_kernel(const float *__restrict__ src, size_t n, __half *__restrict__ dst) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = __float2half_rn(src[i]);
}

// v = float(half_rn(v)) in place: the value a half2 centroid buffer would hold
__global__ void quantize_half_kernel(float *__restrict__ v, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) v[i] = __half2float(__float2half_rn(v[i]));
}

hipError_t launch_half_to_float(const void *src, size_t n, float *dst, hipStream_t st) {
  if (n == 0) return hipSuccess;
  const size_t threads = (n + 7) / 8;
  hipLaunchKernelGGL(half_to_float_kernel, dim3((uint32_t)((threads + 255) / 256)), dim3(256), 0, st,
                     reinterpret_cast<const __half *>(src), n, dst);
  return hipGetLastError();
}

hipError_t launch_float_to_half(const float *src, size_t n, void *dst, hipStream_t st) {
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(float_to_half_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, st, src, n,
                     reinterpret_cast<__half *>(dst));
  return hipGetLastError();
}

hipError_t launch_quantize_half(float *v, size_t n, hipStream_t st) {
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(quantize_half_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, st, v, n);
  return hipGetLastError();
}

}  // namespace kmx
