// mfma_probe.hip -- what the f16 matrix pipe of THIS chip sustains under its power budget.
// A register-only loop of v_mfma_f32_32x32x16_f16 (no LDS, no HBM): every wave keeps NB B-operand
// fragments and streams NA A fragments over them, NACC independent accumulators.  Operands are
//   zero    all-zero halves            (no toggling: the clock stays at its ceiling)
//   const   one repeated finite value  (multiplier busy, operand buses quiet)
//   random  uniform(-1, 1) halves, a fresh pair of fragments per MFMA (what a real filter feeds it)
// at 1 and 2 waves per SIMD, optionally with VALU filler ops per MFMA (the bookkeeping load of
// lloyd_coarse2_kernel: 3 per score = 1.5 VALU per ... see DESIGN.md 4.6).  Prints TFLOP/s and the
// effective clock implied by the instruction count.  Built by scripts/gpu_round2.sh into scratch/bin.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int FILL>
__global__ __launch_bounds__(256) void probe(const f16x8 *__restrict__ ops, int iters, float *__restrict__ out) {
  constexpr int NF = 8;
  f16x8 a[NF], b[NF];
  const int lane = threadIdx.x & 63;
  const size_t base = ((size_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * (2 * NF) * 64 + lane;
#pragma unroll
  for (int i = 0; i < NF; i++) { a[i] = ops[base + (2 * i) * 64]; b[i] = ops[base + (2 * i + 1) * 64]; }
  f32x16 acc0 = {0}, acc1 = {0}, acc2 = {0}, acc3 = {0};
  float v1 = -1e30f, v2 = -1e30f;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < NF; i++) {
      acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i], b[i], acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[(i + 1) % NF], b[i], acc1, 0, 0, 0);
      acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[(i + 2) % NF], b[i], acc2, 0, 0, 0);
      acc3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[(i + 3) % NF], b[i], acc3, 0, 0, 0);
      if (FILL) {   // FILL VALU ops per MFMA, independent of the accumulators in flight
#pragma unroll
        for (int q = 0; q < 4 * FILL; q++) {
          v2 = __builtin_amdgcn_fmed3f(v1, v2, (float)(it + q));
          v1 = __builtin_amdgcn_fmed3f(v1, v2, 1e30f);
        }
      }
    }
  }
  float s = 0;
#pragma unroll
  for (int r = 0; r < 16; r++) s += acc0[r] + acc1[r] + acc2[r] + acc3[r];
  if (s == 1.2345f || v1 + v2 == 3.3f) out[0] = s;   // keep everything alive
}

static uint16_t f2h(float f) {
  _Float16 h = (_Float16)f;
  uint16_t u;
  memcpy(&u, &h, 2);
  return u;
}

int main(int argc, char **argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 4000;
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  const int cus = p.multiProcessorCount;
  printf("device %s, %d CUs, clock %d MHz\n", p.name, cus, p.clockRate / 1000);
  const size_t maxblocks = (size_t)cus * 2;
  const size_t n = maxblocks * 4 * 16 * 64;   // f16x8 elements
  std::vector<uint16_t> host(n * 8);
  f16x8 *dev;
  float *out;
  hipMalloc(&dev, n * 16);
  hipMalloc(&out, 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const char *names[3] = {"zero", "const", "random"};
  for (int data = 0; data < 3; data++) {
    srand(1);
    for (size_t i = 0; i < host.size(); i++)
      host[i] = data == 0 ? 0 : (data == 1 ? f2h(0.5f) : f2h((rand() / (float)RAND_MAX) * 2.f - 1.f));
    hipMemcpy(dev, host.data(), n * 16, hipMemcpyHostToDevice);
    for (int wps = 1; wps <= 2; wps++) {
      for (int fill = 0; fill <= 2; fill++) {
        const int blocks = cus * wps;
        auto launch = [&](int its) {
          if (fill == 0) hipLaunchKernelGGL(probe<0>, dim3(blocks), dim3(256), 0, 0, dev, its, out);
          else if (fill == 1) hipLaunchKernelGGL(probe<1>, dim3(blocks), dim3(256), 0, 0, dev, its, out);
          else hipLaunchKernelGGL(probe<2>, dim3(blocks), dim3(256), 0, 0, dev, its, out);
        };
        launch(iters / 4);   // warm up / ramp the clocks
        hipDeviceSynchronize();
        float best = 0, sum = 0;
        const int reps = 3;
        for (int r = 0; r < reps; r++) {
          hipEventRecord(e0, 0);
          launch(iters);
          hipEventRecord(e1, 0);
          hipEventSynchronize(e1);
          float ms;
          hipEventElapsedTime(&ms, e0, e1);
          sum += ms;
          if (r == 0 || ms < best) best = ms;
        }
        const double mfmas_per_wave = (double)iters * 8 * 4;
        const double flop = mfmas_per_wave * blocks * 4 * 32768.0;
        const double ms = sum / reps;
        // one SIMD issues an MFMA every 32 cycles at best: implied clock if the pipe were saturated
        const double cycles_per_simd = mfmas_per_wave * wps * 32.0;
        printf("%-6s waves/SIMD %d  VALU per MFMA %d : %8.3f ms  %7.1f TFLOP/s  (%.3f of 2500)  pipe-saturated clock >= %.2f GHz\n",
               names[data], wps, 4 * fill * 2 / 4, ms, flop / (ms * 1e-3) / 1e12, flop / (ms * 1e-3) / 1e12 / 2500.0,
               cycles_per_simd / (ms * 1e-3) / 1e9);
      }
    }
  }
  return 0;
}
