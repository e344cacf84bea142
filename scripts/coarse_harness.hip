// coarse_harness.hip -- lloyd_coarse2_kernel<256, fp32 rows, row cache> itself (the header is included, not copied) on
// synthetic operands of the headline shape: N x 256 centred halves in the row cache's layout, a 1024 x 256 half panel.
// Built several times with -DKMX_ABL=<mask> (lloyd_coarse.hpp: timing-only ablations) it prices the kernel's parts
// on ONE box; -DHARNESS_KERNEL=3 runs lloyd_coarse3_kernel (lloyd_coarse3.hpp) on the same inputs and checks that its
// assignments / undecided lists equal the stage it replaces.
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -Ikmcuda_amd/csrc -Iinclude scripts/coarse_harness.hip -o h && ./h [rows] [reps]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <vector>

#include "lloyd_coarse.hpp"
#if HARNESS_KERNEL == 3
#include "lloyd_coarse3.hpp"
#endif

using namespace kmx;

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ uint32_t mix(uint32_t a) {
  a ^= a >> 16; a *= 0x7feb352du; a ^= a >> 15; a *= 0x846ca68bu; a ^= a >> 16;
  return a;
}
// uniform(-amp, amp) halves, 8 per thread
__global__ void fill_halves(f16x8 *dst, size_t n8, float amp, uint32_t seed) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (size_t)gridDim.x * blockDim.x) {
    f16x8 v;
#pragma unroll
    for (int q = 0; q < 8; q++) {
      const uint32_t r = mix((uint32_t)(i * 8 + q) ^ seed ^ (uint32_t)((i * 8 + q) >> 32) * 0x9e3779b9u);
      v[q] = (_Float16)(((float)(r >> 8) * (1.0f / 16777216.0f) * 2.f - 1.f) * amp);
    }
    dst[i] = v;
  }
}
// (||x'||^2, ||dx||^2) per row from the cache itself: row s = block32 b, column col: pieces j of lanes col, col + 32
__global__ void fill_meta(const f16x8 *xc, float2 *meta, float *xdot, uint32_t npad) {
  const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= npad) return;
  const uint32_t b = s / 32, col = s % 32;
  float n2 = 0.f;
  for (int j = 0; j < 16; j++)
    for (int hh = 0; hh < 2; hh++) {
      const f16x8 v = xc[((size_t)b * 16 + j) * 64 + col + 32 * hh];
      for (int q = 0; q < 8; q++) n2 = fmaf((float)v[q], (float)v[q], n2);
    }
  meta[s] = make_float2(n2, n2 * 6e-8f);   // residual norm: ~2^-12 ||x'||
  xdot[s] = 0.f;
}

int main(int argc, char **argv) {
  const uint32_t N = argc > 1 ? (uint32_t)atoll(argv[1]) : 8000000u;
  const int reps = argc > 2 ? atoi(argv[2]) : 10;
  constexpr int DP = 256;
  const uint32_t K = 1024, K_pad = 1024, nsuper = K_pad / 64;
  const size_t npad = ((size_t)N + 255) / 256 * 256;
  f16x8 *xcache;
  float *xmeta, *mu, *und_thr, *biasf;
  unsigned char *panel;
  uint32_t *stats, *asg, *prev, *und, *counters;
  CHECK(hipMalloc(&xcache, npad * DP * 2));
  CHECK(hipMalloc(&xmeta, (npad * 3 + 2) * 4));
  CHECK(hipMalloc(&mu, DP * 4));
  CHECK(hipMalloc(&biasf, K_pad * 4));
  const size_t panel_bytes = (size_t)nsuper * 64 * DP * 2 + (size_t)nsuper * 64 * 4;
  CHECK(hipMalloc(&panel, panel_bytes));
  CHECK(hipMalloc(&stats, 64));
  CHECK(hipMalloc(&asg, (size_t)N * 4));
  CHECK(hipMalloc(&prev, (size_t)N * 4));
  CHECK(hipMalloc(&und, (size_t)N * 4));
  CHECK(hipMalloc(&und_thr, (size_t)N * 4));
  CHECK(hipMalloc(&counters, 256));
  CHECK(hipMemset(mu, 0, DP * 4));
  CHECK(hipMemset(asg, 0xFF, (size_t)N * 4));
  CHECK(hipMemset(counters, 0, 256));
  hipLaunchKernelGGL(fill_halves, dim3(4096), dim3(256), 0, 0, xcache, npad * DP / 8, 0.5f, 1u);          // uniform rows, centred
  hipLaunchKernelGGL(fill_halves, dim3(64), dim3(256), 0, 0, reinterpret_cast<f16x8 *>(panel), (size_t)K_pad * DP / 8, 0.08f, 2u);
  hipLaunchKernelGGL(fill_meta, dim3((uint32_t)(npad / 256)), dim3(256), 0, 0, xcache, reinterpret_cast<float2 *>(xmeta), xmeta + 2 * npad + 2, (uint32_t)npad);
  // biases -||c'||^2 / 2 ~ -0.27; ||c'|| ~ 0.74; the rows' ||x'|| ~ 4.6: scores spread ~ +-0.2, gaps as k-means leaves them
  std::vector<float> hb(K_pad);
  srand(3);
  for (auto &v : hb) v = -0.27f - 0.02f * (rand() / (float)RAND_MAX);
  CHECK(hipMemcpy(panel + (size_t)nsuper * 64 * DP * 2, hb.data(), K_pad * 4, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(biasf, hb.data(), K_pad * 4, hipMemcpyHostToDevice));
  float st[16] = {0};
  st[0] = 0.56f;    // max ||c'||^2
  st[1] = 0.30f;    // max |bias|
  st[2] = 86.0f;    // max ||c||^2  (uniform [0,1)^256 centroids: ~ 256/3)
  st[5] = 2.0e-8f;  // max ||c' - hi(c')||^2
  CHECK(hipMemcpy(stats, st, 64, hipMemcpyHostToDevice));
  const float mun[2] = {9.24f, 0.f};
  CHECK(hipMemcpy(xmeta + 2 * npad, mun, 8, hipMemcpyHostToDevice));
  CHECK(hipDeviceSynchronize());

  const float eps = 1.02f * (DP + 12) * 5.9604645e-8f;
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  auto launch2 = [&]() {
    const size_t lds_bytes = 2 * 64 * (size_t)(DP * 2) + 512 + 64 + (size_t)DP * 4;
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&lloyd_coarse2_kernel<DP, false, true, true, 2, 0>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    hipLaunchKernelGGL((lloyd_coarse2_kernel<DP, false, true, true, 2, 0>), dim3((N + 255) / 256), dim3(256), lds_bytes, 0,
                       (const void *)xcache, xmeta, N, (uint32_t)DP, reinterpret_cast<const float *>(panel), biasf, mu, K_pad, K, stats, eps, 0.f,
                       asg, prev, und, und_thr, counters, CarryArgs());
  };
  auto time_it = [&](auto &&launch, const char *what) {
    launch();
    CHECK(hipDeviceSynchronize());
    CHECK(hipMemset(counters, 0, 256));
    std::vector<float> ms(reps);
    for (int r = 0; r < reps; r++) {
      CHECK(hipMemsetAsync(counters, 0, 256, 0));
      CHECK(hipEventRecord(e0, 0));
      launch();
      CHECK(hipEventRecord(e1, 0));
      CHECK(hipEventSynchronize(e1));
      CHECK(hipEventElapsedTime(&ms[r], e0, e1));
    }
    CHECK(hipGetLastError());
    std::sort(ms.begin(), ms.end());
    double avg = 0;
    for (float m : ms) avg += m;
    avg /= reps;
    uint32_t hc[16];
    CHECK(hipMemcpy(hc, counters, 64, hipMemcpyDeviceToHost));
    const double flop = 2.0 * DP * K * (double)N;
    // what the launch left behind, as one number per table (the variants must agree with variant 0)
    std::vector<uint32_t> ha(N), hu(hc[4]);
    std::vector<float> ht(hc[4]);
    CHECK(hipMemcpy(ha.data(), asg, (size_t)N * 4, hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(hu.data(), und, (size_t)hc[4] * 4, hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(ht.data(), und_thr, (size_t)hc[4] * 4, hipMemcpyDeviceToHost));
    uint64_t sa = 0, su = 0;
    for (uint32_t i = 0; i < N; i++) sa += (uint64_t)ha[i] * (i % 1000003u + 1u);
    for (uint32_t i = 0; i < hc[4]; i++) { uint32_t b; memcpy(&b, &ht[i], 4); su += (uint64_t)(hu[i] + 1u) * (uint64_t)(b % 65521u + 1u); }
    printf("sums %016llx %016llx  ", (unsigned long long)sa, (unsigned long long)su);
    printf("%-34s abl %2d  N %u : avg %.3f ms  min %.3f  median %.3f  -> %.1f TFLOP/s = %.3f of 2500   changed %u undecided %u\n", what, KMX_ABL, N,
           avg, ms[0], ms[reps / 2], flop / (avg * 1e-3) / 1e12, flop / (avg * 1e-3) / 1e12 / 2500.0, hc[0], hc[4]);
    fflush(stdout);
  };
  time_it(launch2, "lloyd_coarse2_kernel");
#if HARNESS_KERNEL == 3
  // the reference result of the stage it replaces
  std::vector<uint32_t> a2(N), a3(N);
  CHECK(hipMemset(asg, 0xFF, (size_t)N * 4));
  CHECK(hipMemset(counters, 0, 256));
  launch2();
  CHECK(hipDeviceSynchronize());
  CHECK(hipMemcpy(a2.data(), asg, (size_t)N * 4, hipMemcpyDeviceToHost));
  uint32_t c2[16];
  CHECK(hipMemcpy(c2, counters, 64, hipMemcpyDeviceToHost));
  std::vector<uint32_t> u2(c2[4]);
  std::vector<float> t2(c2[4]);
  CHECK(hipMemcpy(u2.data(), und, (size_t)c2[4] * 4, hipMemcpyDeviceToHost));
  CHECK(hipMemcpy(t2.data(), und_thr, (size_t)c2[4] * 4, hipMemcpyDeviceToHost));
  auto launch3 = [&]() {
    CHECK(launch_lloyd_coarse3(xcache, xmeta, N, reinterpret_cast<const float *>(panel), mu, K_pad, K, stats, eps, 0.f, asg, prev, und,
                               und_thr, counters, CarryArgs(), 0));
  };
  CHECK(hipMemset(asg, 0xFF, (size_t)N * 4));
  CHECK(hipMemset(counters, 0, 256));
  launch3();
  CHECK(hipDeviceSynchronize());
  CHECK(hipMemcpy(a3.data(), asg, (size_t)N * 4, hipMemcpyDeviceToHost));
  uint32_t c3[16];
  CHECK(hipMemcpy(c3, counters, 64, hipMemcpyDeviceToHost));
  size_t diff = 0;
  for (uint32_t i = 0; i < N; i++) diff += a2[i] != a3[i];
  std::vector<uint32_t> u3(c3[4]);
  std::vector<float> t3(c3[4]);
  CHECK(hipMemcpy(u3.data(), und, (size_t)c3[4] * 4, hipMemcpyDeviceToHost));
  CHECK(hipMemcpy(t3.data(), und_thr, (size_t)c3[4] * 4, hipMemcpyDeviceToHost));
  // lists as sets of (row, cut-off)
  std::vector<std::pair<uint32_t, uint32_t>> l2(c2[4]), l3(c3[4]);
  for (size_t i = 0; i < l2.size(); i++) { uint32_t b; memcpy(&b, &t2[i], 4); l2[i] = {u2[i], b}; }
  for (size_t i = 0; i < l3.size(); i++) { uint32_t b; memcpy(&b, &t3[i], 4); l3[i] = {u3[i], b}; }
  std::sort(l2.begin(), l2.end());
  std::sort(l3.begin(), l3.end());
  printf("coarse3 against coarse2: %zu assignments differ, changed %u / %u, undecided %u / %u, lists %s\n", diff, c3[0], c2[0], c3[4],
         c2[4], l2 == l3 ? "equal" : "DIFFER");
  time_it(launch3, "lloyd_coarse3_kernel");
#endif
  return 0;
}
