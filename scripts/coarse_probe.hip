// coarse_probe.hip -- the ceiling of stage 1's INSTRUCTION MIX on this chip, without HBM or LDS-DMA in the picture:
// every wave keeps NSET 32-row operand sets (random halves) in registers and sweeps 32-centroid tiles that are resident
// in LDS (random halves, lloyd_coarse2_kernel's swizzle and fragment reads: one ds_read_b128 per k-step feeds NSET
// products), with the top-2 bookkeeping of the real kernel on the accumulators.  What varies:
//   WPS   waves per SIMD: 2 = today's kernel (4-wave blocks, two per CU, <= 256 registers), 1 = 512 registers
//   NSET  operand sets per wave (2 today; 4: half the fragment reads per product)
//   PIPE  0 = today's order (a tile's products, then its bookkeeping), 1 = the bookkeeping of tile t - 1 between the
//         products of tile t (double accumulators)
//   BOOK  0 none, 1 = 2.5 VALU per score (pack + med3 + med3 + max3 per pair), 2 = 1.5 (no index bits: value-only)
//   SYNC  a block barrier every second tile (the super-tile hand-over of the real kernel); 2: every fourth
//   BW    waves per block: 4 (two blocks per CU at WPS 2) or 8 (one block: both waves of a SIMD meet at its barriers)
// Prints TFLOP/s, the fraction of 2500 and the clock implied if the matrix pipe never idled.
//   hipcc -O3 --offload-arch=gfx950 scripts/coarse_probe.hip -o scratch/bin/coarse_probe && scratch/bin/coarse_probe [tiles]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ f16x8 lds_frag_issue(uint32_t addr) {
  f16x8 f;
  asm volatile("ds_read_b128 %0, %1" : "=v"(f) : "v"(addr) : "memory");
  return f;
}
template <int N>
__device__ __forceinline__ void lds_frag_wait(f16x8 &f) {
  asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(f) : "n"(N));
}

constexpr int KS = 16, ROWB = 512, TILEB = 32 * ROWB;

template <int WPS, int NSET, int PIPE, int BOOK, int SYNC, int BW = 4>
__global__ __launch_bounds__(BW * 64, BW == 8 ? 1 : WPS) void probe(const f16x8 *__restrict__ ops, const f16x8 *__restrict__ panel, int tiles,
                                                  float *__restrict__ out) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
  typedef __attribute__((address_space(3))) unsigned char lds_byte;
  const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_byte *)lds;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, col = lane & 31, h = lane >> 5;
  // four tiles = 64 KB of random halves
  for (int i = tid; i < 4 * TILEB / 16; i += BW * 64) reinterpret_cast<f16x8 *>(lds)[i] = panel[i];
  __syncthreads();
  f16x8 x[NSET][KS];
  const size_t base = ((size_t)blockIdx.x * BW + wave) * (NSET * KS) * 64 + lane;
#pragma unroll
  for (int s = 0; s < NSET; s++)
#pragma unroll
    for (int j = 0; j < KS; j++) x[s][j] = ops[base + (size_t)(s * KS + j) * 64];

  const uint32_t fragbase = lds0 + (uint32_t)col * ROWB + (uint32_t)h * (KS * 16) + (uint32_t)((col & 15) * 16);
  float pinf = INFINITY;
  asm volatile("" : "+s"(pinf));
  float v1[NSET], v2[NSET];
#pragma unroll
  for (int s = 0; s < NSET; s++) v1[s] = v2[s] = -INFINITY;
  auto pack = [&](float v, int r) { return __uint_as_float((__float_as_uint(v) & 0xFFFFFFF0u) | (uint32_t)r); };
  auto book2 = [&](float a, float b, int r, float &b1, float &b2) {
    if (BOOK == 1) {
      const float pa = pack(a, r), pb = pack(b, r + 1);
      const float m = __builtin_amdgcn_fmed3f(b1, pa, pb);
      b2 = __builtin_amdgcn_fmed3f(b2, m, pinf);
      float t;
      asm("v_max3_f32 %0, %1, %2, %3" : "=v"(t) : "v"(b1), "v"(pa), "v"(pb));
      b1 = t;
    } else if (BOOK == 2) {
      const float m = __builtin_amdgcn_fmed3f(b1, a, b);
      b2 = __builtin_amdgcn_fmed3f(b2, m, pinf);
      b1 = __builtin_amdgcn_fmed3f(__builtin_amdgcn_fmed3f(b1, a, pinf), b, pinf);   // (max3 through the compiler's eyes)
    }
  };
  f32x16 acc[2][NSET];
#pragma unroll
  for (int s = 0; s < NSET; s++) acc[0][s] = acc[1][s] = f32x16{0};
  constexpr int PD = 3;

  // one tile: products into acc[cur]; PIPE: the bookkeeping of acc[cur ^ 1] between them
  auto tile = [&](int t, auto curc, bool have_prev) {
    constexpr int cur = decltype(curc)::value;
    uint32_t fb = fragbase + (uint32_t)(t & 3) * TILEB;
    asm volatile("" : "+v"(fb));
#pragma unroll
    for (int s = 0; s < NSET; s++) acc[cur][s] = f32x16{(float)t};   // (the real kernel loads 16 biases from LDS here)
    f16x8 fr[PD + 1];
#pragma unroll
    for (int j = 0; j < PD; j++) fr[j] = lds_frag_issue(fb ^ (uint32_t)(j * 16));
#pragma unroll
    for (int j = 0; j < KS; j++) {
      if (j + PD < KS) fr[(j + PD) % (PD + 1)] = lds_frag_issue(fb ^ (uint32_t)((j + PD) * 16));
      const int behind = (KS - 1 - j) < PD ? (KS - 1 - j) : PD;
      f16x8 &f = fr[j % (PD + 1)];
      if (behind == 3) lds_frag_wait<3>(f);
      else if (behind == 2) lds_frag_wait<2>(f);
      else if (behind == 1) lds_frag_wait<1>(f);
      else lds_frag_wait<0>(f);
      // PIPE: the NSET * 8 score pairs of the previous tile are spread over the 16 k-steps, one pair behind a product
      const int p0 = j * (NSET * 8) / KS, p1 = (j + 1) * (NSET * 8) / KS;
#pragma unroll
      for (int s = 0; s < NSET; s++) {
        acc[cur][s] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f, x[s][j], acc[cur][s], 0, 0, 0);
        if (PIPE && BOOK) {
          __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);
          if (p0 + s < p1) {
            if (have_prev) {
              const int q = p0 + s, ss = q / 8, r = (q % 8) * 2;
              book2(acc[cur ^ 1][ss][r], acc[cur ^ 1][ss][r + 1], r, v1[ss], v2[ss]);
            }
            if constexpr (BOOK == 1) __builtin_amdgcn_sched_group_barrier(0x2, 5, 0);
            else __builtin_amdgcn_sched_group_barrier(0x2, 4, 0);
          }
        }
      }
    }
    if (!PIPE && BOOK) {
#pragma unroll
      for (int r = 0; r < 16; r += 2)
#pragma unroll
        for (int s = 0; s < NSET; s++) book2(acc[cur][s][r], acc[cur][s][r + 1], r, v1[s], v2[s]);
    }
  };
  for (int t = 0; t < tiles; t += 2) {
    tile(t, std::integral_constant<int, 0>(), t > 0);
    tile(t + 1, std::integral_constant<int, 1>(), true);
    if (SYNC == 1 || (SYNC == 2 && (t & 2))) __syncthreads();   // (SYNC 2: a barrier every 4 tiles)
  }
  float sum = 0;
#pragma unroll
  for (int s = 0; s < NSET; s++) {
    sum += v1[s] + v2[s];
    if (!BOOK || PIPE)
#pragma unroll
      for (int r = 0; r < 16; r++) sum += acc[0][s][r] + acc[1][s][r];
  }
  if (sum == 1.2345f) out[0] = sum;
}

static uint16_t f2h(float f) {
  _Float16 hh = (_Float16)f;
  uint16_t u;
  memcpy(&u, &hh, 2);
  return u;
}

template <int WPS, int NSET, int PIPE, int BOOK, int SYNC, int BW = 4>
static void run(const char *what, int cus, const f16x8 *ops, const f16x8 *panel, int tiles, float *out) {
  const int blocks = cus * WPS * 4 / BW;
  const size_t ldsb = WPS == 1 ? 96 * 1024 : 4 * TILEB;   // (one block per CU when a wave is to have its SIMD to itself)
  (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&probe<WPS, NSET, PIPE, BOOK, SYNC, BW>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL((probe<WPS, NSET, PIPE, BOOK, SYNC, BW>), dim3(blocks), dim3(BW * 64), ldsb, 0, ops, panel, tiles / 4, out);
  hipDeviceSynchronize();
  float sum = 0, best = 1e30f;
  const int reps = 3;
  for (int r = 0; r < reps; r++) {
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((probe<WPS, NSET, PIPE, BOOK, SYNC, BW>), dim3(blocks), dim3(BW * 64), ldsb, 0, ops, panel, tiles, out);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    sum += ms;
    if (ms < best) best = ms;
  }
  hipError_t err = hipGetLastError();
  const double mfmas_per_wave = (double)tiles * KS * NSET;
  const double flop = mfmas_per_wave * blocks * BW * 32768.0;
  const double ms = sum / reps;
  const double cyc = mfmas_per_wave * WPS * 32.0;
  printf("%-46s wps %d nset %d pipe %d book %d sync %d : %8.3f ms  %7.1f TFLOP/s  (%.3f of 2500)  pipe-saturated clock >= %.2f GHz %s\n", what,
         WPS, NSET, PIPE, BOOK, SYNC, ms, flop / (ms * 1e-3) / 1e12, flop / (ms * 1e-3) / 1e12 / 2500.0, cyc / (ms * 1e-3) / 1e9,
         err == hipSuccess ? "" : hipGetErrorString(err));
  fflush(stdout);
}

int main(int argc, char **argv) {
  const int tiles = argc > 1 ? atoi(argv[1]) : 4096;
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  const int cus = p.multiProcessorCount;
  printf("device %s, %d CUs, clock %d MHz, %d tiles per wave\n", p.name, cus, p.clockRate / 1000, tiles);
  const size_t nops = (size_t)cus * 2 * 4 * 4 * KS * 64;   // f16x8 elements, enough for NSET = 4
  std::vector<uint16_t> host(nops * 8), hpanel(4 * TILEB / 2);
  srand(1);
  for (auto &v : host) v = f2h((rand() / (float)RAND_MAX) * 2.f - 1.f);
  for (auto &v : hpanel) v = f2h((rand() / (float)RAND_MAX) * 2.f - 1.f);
  f16x8 *ops, *panel;
  float *out;
  hipMalloc(&ops, nops * 16);
  hipMalloc(&panel, 4 * TILEB);
  hipMalloc(&out, 4);
  hipMemcpy(ops, host.data(), nops * 16, hipMemcpyHostToDevice);
  hipMemcpy(panel, hpanel.data(), 4 * TILEB, hipMemcpyHostToDevice);
  //   WPS NSET PIPE BOOK SYNC
  run<2, 2, 0, 0, 0>("2 waves/SIMD, products + fragment reads only", cus, ops, panel, tiles, out);
  run<2, 2, 0, 1, 0>("  + bookkeeping (today's mix), no barrier", cus, ops, panel, tiles, out);
  run<2, 2, 0, 1, 1>("  today's structure: + barrier per 2 tiles", cus, ops, panel, tiles, out);
  run<2, 2, 0, 2, 1>("  value-only bookkeeping (1.5 per score)", cus, ops, panel, tiles, out);
  run<2, 2, 1, 1, 1>("  2 waves/SIMD, bookkeeping of tile t-1 between products", cus, ops, panel, tiles, out);
  run<2, 2, 1, 2, 1>("  the same, value-only bookkeeping", cus, ops, panel, tiles, out);
  run<1, 2, 0, 0, 0>("1 wave/SIMD, products + fragment reads only", cus, ops, panel, tiles, out);
  run<1, 2, 0, 1, 1>("  bookkeeping behind the tile (not overlapped)", cus, ops, panel, tiles, out);
  run<1, 2, 1, 1, 0>("  bookkeeping of tile t-1 between products", cus, ops, panel, tiles, out);
  run<1, 2, 1, 1, 1>("  + barrier per 2 tiles", cus, ops, panel, tiles, out);
  run<1, 2, 1, 2, 1>("  value-only bookkeeping, pipelined", cus, ops, panel, tiles, out);
  run<1, 4, 0, 0, 0>("1 wave/SIMD, 4 sets, products + reads only", cus, ops, panel, tiles, out);
  run<1, 4, 1, 1, 0>("  4 sets, pipelined bookkeeping", cus, ops, panel, tiles, out);
  run<1, 4, 1, 1, 1>("  + barrier per 2 tiles", cus, ops, panel, tiles, out);
  run<1, 4, 1, 2, 1>("  4 sets, value-only bookkeeping, pipelined", cus, ops, panel, tiles, out);
  run<1, 3, 1, 1, 1>("  3 sets, pipelined bookkeeping, barrier", cus, ops, panel, tiles, out);
  // ONE 8-wave block per CU (both waves of a SIMD in the same block: half the staging traffic per row in the real kernel)
  run<2, 2, 0, 1, 1, 8>("8-wave block, today's mix, barrier per 2 tiles", cus, ops, panel, tiles, out);
  run<2, 2, 0, 1, 2, 8>("8-wave block, today's mix, barrier per 4 tiles", cus, ops, panel, tiles, out);
  run<2, 2, 0, 1, 0, 8>("8-wave block, today's mix, no barrier", cus, ops, panel, tiles, out);
  return 0;
}
