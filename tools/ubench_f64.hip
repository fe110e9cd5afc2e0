// fp64 issue-rate micro-benchmarks for MI355X (gfx950): what bounds the sweeps?
//   hipcc -O3 --offload-arch=gfx950 -o /tmp/ubench_f64 tools/ubench_f64.hip && /tmp/ubench_f64
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double v4d __attribute__((ext_vector_type(4)));
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f64_16x16x4f64((a), (b), (c), 0, 0, 0)
#define MFMA4(a, b, c) __builtin_amdgcn_mfma_f64_4x4x4f64((a), (b), (c), 0, 0, 0)

struct Clk { unsigned long long cyc, real; };

template <int NACC>
__global__ void __launch_bounds__(256) k_mfma16(int iters, double* sink, Clk* clk) {
  v4d acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = v4d{0, 0, 0, 0};
  double x = 1.0 + threadIdx.x * 1e-3, y = 0.5 - threadIdx.x * 1e-4;
  unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int j = 0; j < NACC; ++j) acc[j] = MFMA(x, y, acc[j]);
  }
  unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  double s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  if (s == 12345.678) sink[0] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) { clk->cyc = c1 - c0; clk->real = r1 - r0; }
}

template <int NACC>
__global__ void __launch_bounds__(256) k_mfma4(int iters, double* sink, Clk* clk) {
  double acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = 0;
  double x = 1.0 + threadIdx.x * 1e-3, y = 0.5 - threadIdx.x * 1e-4;
  unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int j = 0; j < NACC; ++j) acc[j] = MFMA4(x, y, acc[j]);
  }
  unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  double s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i];
  if (s == 12345.678) sink[0] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) { clk->cyc = c1 - c0; clk->real = r1 - r0; }
}

template <int NACC>
__global__ void __launch_bounds__(256) k_valu(int iters, double* sink, Clk* clk) {
  double acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = threadIdx.x * 1e-3 + i;
  double x = 1.0 + threadIdx.x * 1e-9, y = 1e-7 * threadIdx.x;
  unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int j = 0; j < NACC; ++j) acc[j] = __builtin_fma(acc[j], x, y);
  }
  unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  double s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i];
  if (s == 12345.678) sink[0] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) { clk->cyc = c1 - c0; clk->real = r1 - r0; }
}

// MFMA + independent VALU fp64 FMAs in the same wave: do the pipes overlap?
template <int NV>
__global__ void __launch_bounds__(256) k_mix(int iters, double* sink, Clk* clk) {
  v4d acc[4];
  for (int i = 0; i < 4; ++i) acc[i] = v4d{0, 0, 0, 0};
  double va[NV > 0 ? NV : 1];
  for (int i = 0; i < NV; ++i) va[i] = threadIdx.x * 1e-3 + i;
  double x = 1.0 + threadIdx.x * 1e-9, y = 1e-7 * threadIdx.x;
  unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      acc[j] = MFMA(x, y, acc[j]);
#pragma unroll
      for (int q = 0; q < NV / 4; ++q) va[j * (NV / 4) + q] = __builtin_fma(va[j * (NV / 4) + q], x, y);
    }
  }
  unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  double s = 0;
  for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  for (int i = 0; i < NV; ++i) s += va[i];
  if (s == 12345.678) sink[0] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) { clk->cyc = c1 - c0; clk->real = r1 - r0; }
}

template <typename F>
static void run(const char* name, F launch, double flops_per_thread_iter_x64, int iters, int blocks, Clk* dclk) {
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  launch(iters / 10 + 1);
  hipDeviceSynchronize();
  hipEventRecord(a);
  launch(iters);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  Clk h; hipMemcpy(&h, dclk, sizeof h, hipMemcpyDeviceToHost);
  double waves = blocks * 4.0;
  double flops = waves * iters * flops_per_thread_iter_x64;
  double ghz = (double)h.cyc / ((double)h.real / 100e6) / 1e9;
  double cyc_per_iter = (double)h.cyc / iters;
  printf("%-34s blocks=%5d  %8.3f ms  %7.2f TFLOP/s  clk %.2f GHz  %.1f cyc/iter/wave\n", name, blocks, ms,
         flops / (ms * 1e-3) / 1e12, ghz, cyc_per_iter);
}

int main() {
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
  int cu = p.multiProcessorCount;
  printf("device %s, %d CUs, clock %d MHz\n", p.gcnArchName, cu, p.clockRate / 1000);
  double* sink; Clk* clk;
  hipMalloc(&sink, 8); hipMalloc(&clk, sizeof(Clk));
  const int it = 20000;
  for (int wpc : {1, 2, 4}) {  // blocks per CU -> waves per SIMD
    int blocks = cu * wpc;
    char nm[64];
    snprintf(nm, 64, "mfma16x16x4 f64 4acc  %dw/SIMD", wpc);
    run(nm, [&](int n) { hipLaunchKernelGGL(k_mfma16<4>, dim3(blocks), dim3(256), 0, 0, n, sink, clk); }, 4 * 2048.0, it, blocks, clk);
    snprintf(nm, 64, "mfma16x16x4 f64 8acc  %dw/SIMD", wpc);
    run(nm, [&](int n) { hipLaunchKernelGGL(k_mfma16<8>, dim3(blocks), dim3(256), 0, 0, n, sink, clk); }, 8 * 2048.0, it, blocks, clk);
    snprintf(nm, 64, "mfma16x16x4 f64 1acc  %dw/SIMD", wpc);
    run(nm, [&](int n) { hipLaunchKernelGGL(k_mfma16<1>, dim3(blocks), dim3(256), 0, 0, n, sink, clk); }, 1 * 2048.0, it, blocks, clk);
    snprintf(nm, 64, "mfma4x4x4  f64 8acc  %dw/SIMD", wpc);
    run(nm, [&](int n) { hipLaunchKernelGGL(k_mfma4<8>, dim3(blocks), dim3(256), 0, 0, n, sink, clk); }, 8 * 512.0, it, blocks, clk);
    snprintf(nm, 64, "valu fma f64 16acc   %dw/SIMD", wpc);
    run(nm, [&](int n) { hipLaunchKernelGGL(k_valu<16>, dim3(blocks), dim3(256), 0, 0, n, sink, clk); }, 16 * 128.0, it, blocks, clk);
    snprintf(nm, 64, "mix 4 mfma + 8 fma   %dw/SIMD", wpc);
    run(nm, [&](int n) { hipLaunchKernelGGL(k_mix<8>, dim3(blocks), dim3(256), 0, 0, n, sink, clk); }, 4 * 2048.0 + 8 * 128.0, it, blocks, clk);
    snprintf(nm, 64, "mix 4 mfma + 32 fma  %dw/SIMD", wpc);
    run(nm, [&](int n) { hipLaunchKernelGGL(k_mix<32>, dim3(blocks), dim3(256), 0, 0, n, sink, clk); }, 4 * 2048.0 + 32 * 128.0, it, blocks, clk);
  }
  // one CU only (no chip-level power effects)
  run("mfma16x16x4 f64 4acc 1 block", [&](int n) { hipLaunchKernelGGL(k_mfma16<4>, dim3(1), dim3(256), 0, 0, n, sink, clk); }, 4 * 2048.0, it, 1, clk);
  run("valu fma f64 16acc 1 block", [&](int n) { hipLaunchKernelGGL(k_valu<16>, dim3(1), dim3(256), 0, 0, n, sink, clk); }, 16 * 128.0, it, 1, clk);
  return 0;
}
