// Issue-rate details of v_mfma_f64_4x4x4_4b_f64 on MI355X: unroll depth, co-issue with VALU fp64
// FMAs, with LDS reads, and dependent-accumulator latency.
#include <hip/hip_runtime.h>
#include <cstdio>
#define MFMA4(a, b, c) __builtin_amdgcn_mfma_f64_4x4x4f64((a), (b), (c), 0, 0, 0)
struct Clk { unsigned long long cyc, real; };

template <int NACC, int NV, int NLDS>
__global__ void __launch_bounds__(256) k(int iters, double* sink, Clk* clk) {
  __shared__ double sh[64 * 64];
  double acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = 0;
  double va[NV > 0 ? NV : 1];
  for (int i = 0; i < NV; ++i) va[i] = threadIdx.x * 1e-3 + i;
  for (int i = threadIdx.x; i < 64 * 64; i += 256) sh[i] = i * 1e-6;
  __syncthreads();
  double x = 1.0 + threadIdx.x * 1e-9, y = 1e-7 * threadIdx.x;
  int li = threadIdx.x & 63;
  unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int i = 0; i < iters; ++i) {
    asm volatile("" : "+v"(li));
#pragma unroll
    for (int j = 0; j < NACC; ++j) {
      double a = x;
      if (NLDS > 0 && (j % (NACC / (NLDS > 0 ? NLDS : 1))) == 0) a = sh[li + 64 * (j % 64)];
      acc[j] = MFMA4(a, y, acc[j]);
      if (NV > 0 && j < NV) va[j] = __builtin_fma(va[j], x, y);
    }
  }
  unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  double s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i];
  for (int i = 0; i < NV; ++i) s += va[i];
  if (s == 12345.678) sink[0] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) { clk->cyc = c1 - c0; clk->real = r1 - r0; }
}

template <typename F>
static void run(const char* name, F launch, int nacc, int nv, int iters, int blocks, Clk* dclk) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  launch(iters / 10 + 1); hipDeviceSynchronize();
  hipEventRecord(a); launch(iters); hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  Clk h; hipMemcpy(&h, dclk, sizeof h, hipMemcpyDeviceToHost);
  double flops = blocks * 4.0 * iters * (nacc * 512.0 + nv * 128.0);
  printf("%-44s %7.3f ms %7.2f TF  clk %.2f GHz  %.2f cyc/MFMA/wave\n", name, ms, flops / (ms * 1e-3) / 1e12,
         (double)h.cyc / ((double)h.real / 100e6) / 1e9, (double)h.cyc / iters / nacc);
}
#define RUN(NACC, NV, NLDS, WPS, label)                                                              \
  run(label, [&](int n) { hipLaunchKernelGGL((k<NACC, NV, NLDS>), dim3(cu * WPS), dim3(256), 0, 0, n, sink, clk); }, \
      NACC, NV, 4000, cu * WPS, clk)
int main() {
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
  int cu = p.multiProcessorCount;
  double* sink; Clk* clk; hipMalloc(&sink, 8); hipMalloc(&clk, sizeof(Clk));
  RUN(8, 0, 0, 1, "8 acc, 1 w/SIMD");
  RUN(16, 0, 0, 1, "16 acc, 1 w/SIMD");
  RUN(64, 0, 0, 1, "64 acc, 1 w/SIMD");
  RUN(64, 0, 0, 2, "64 acc, 2 w/SIMD");
  RUN(4, 0, 0, 1, "4 acc, 1 w/SIMD (dependent every 4)");
  RUN(2, 0, 0, 1, "2 acc, 1 w/SIMD (dependent every 2)");
  RUN(1, 0, 0, 1, "1 acc, 1 w/SIMD (dependent chain)");
  RUN(64, 8, 0, 1, "64 acc + 8 fma, 1 w/SIMD");
  RUN(64, 16, 0, 1, "64 acc + 16 fma, 1 w/SIMD");
  RUN(64, 32, 0, 1, "64 acc + 32 fma, 1 w/SIMD");
  RUN(64, 64, 0, 1, "64 acc + 64 fma, 1 w/SIMD");
  RUN(64, 32, 0, 2, "64 acc + 32 fma, 2 w/SIMD");
  RUN(64, 0, 16, 1, "64 acc, A from LDS every 4th, 1 w/SIMD");
  RUN(64, 0, 64, 1, "64 acc, A from LDS every MFMA, 1 w/SIMD");
  RUN(64, 0, 16, 2, "64 acc, A from LDS every 4th, 2 w/SIMD");
  return 0;
}
