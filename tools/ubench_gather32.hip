// ubench_gather32.hip -- what limits the class sweep on fp32 inputs?  Rows of D floats gathered in
// shuffled order, 4 fields, every byte read once, summed in fp64 (as the sweep does).
//   W   floats per lane per load (1: 4 rows x 64 B per instruction; 2: 4 rows x 128 B; 4: 4 rows x 256 B)
//   NI  load instructions per field issued back to back (NI * 4 loads in flight per wave)
//   OCC workgroups of 256 per CU (1 = one wave per SIMD, like the one-pass sweep; 2 = two)
// build: hipcc -O3 --offload-arch=gfx950 -o ubench_gather32 ubench_gather32.hip ; run: ./ubench_gather32
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <random>

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s\n", hipGetErrorString(e_)); exit(1); } } while (0)

template <int W, int NI, int OCC, int RPI = 4>
__global__ void __launch_bounds__(256, OCC) gather(const float* const* f, const int* rows, int nrows, int D,
                                                   int colgroups, double* sink) {
  extern __shared__ double hog[];            // sized by the host so that exactly OCC workgroups fit a CU
  constexpr int LPR = 64 / RPI;              // lanes per row
  constexpr int CPW = LPR * W;               // columns per wave
  const int wave = (blockIdx.x * 4 + (threadIdx.x >> 6));
  const int lane = threadIdx.x & 63;
  const int cg = wave % colgroups, rw = wave / colgroups, nrw = (gridDim.x * 4) / colgroups;
  const int g = lane / LPR, c = (lane % LPR) * W;
  const int col = cg * CPW + c;
  if (col >= D) return;
  double s[4][W];
  for (int i = 0; i < 4; ++i) for (int w = 0; w < W; ++w) s[i][w] = 0.0;
  const int r0 = (int)((long)nrows * rw / nrw), r1 = (int)((long)nrows * (rw + 1) / nrw);
  for (int r = r0; r + NI * RPI <= r1; r += NI * RPI) {
    int rr[NI];
    for (int j = 0; j < NI; ++j) rr[j] = rows[r + j * RPI + g];
    float v[4][NI][W];
    for (int i = 0; i < 4; ++i)
      for (int j = 0; j < NI; ++j) {
        const float* p = f[i] + (long)rr[j] * D + col;
        if (W == 1) v[i][j][0] = *p;
        else if (W == 2) { float2 t = *reinterpret_cast<const float2*>(p); v[i][j][0] = t.x; v[i][j][1] = t.y; }
        else { float4 t = *reinterpret_cast<const float4*>(p); v[i][j][0] = t.x; v[i][j][1] = t.y; v[i][j][2 % W] = t.z; v[i][j][3 % W] = t.w; }
      }
    for (int i = 0; i < 4; ++i)
      for (int j = 0; j < NI; ++j)
        for (int w = 0; w < W; ++w) s[i][w] += (double)v[i][j][w];
  }
  double t = 0.0;
  for (int i = 0; i < 4; ++i) for (int w = 0; w < W; ++w) t += s[i][w];
  if (t == 1.2345e300) sink[0] = t + hog[0];
}

int main() {
  const int N = 3110402, D = 128;
  const size_t bytes = (size_t)N * D * 4;
  float* fd[4];
  for (int i = 0; i < 4; ++i) { CHK(hipMalloc(&fd[i], bytes)); CHK(hipMemset(fd[i], 0, bytes)); }
  const float** fdev; CHK(hipMalloc(&fdev, 4 * sizeof(float*)));
  CHK(hipMemcpy(fdev, fd, 4 * sizeof(float*), hipMemcpyHostToDevice));
  std::vector<int> perm(N);
  for (int i = 0; i < N; ++i) perm[i] = i;
  std::mt19937 gen(1);
  std::shuffle(perm.begin(), perm.end(), gen);
  int* rows; CHK(hipMalloc(&rows, (N + 256) * sizeof(int)));
  CHK(hipMemcpy(rows, perm.data(), N * sizeof(int), hipMemcpyHostToDevice));
  double* sink; CHK(hipMalloc(&sink, 8));
  hipEvent_t a, b; CHK(hipEventCreate(&a)); CHK(hipEventCreate(&b));
  auto run = [&](const char* name, auto kern, int W, int occ, int rpi = 4) {
    const int colgroups = D / ((64 / rpi) * W);
    const int waves = ((256 * 4 * occ * 4) / colgroups) * colgroups;   // 4 rounds of the resident waves
    const size_t lds = occ == 1 ? 100 * 1024 : (occ == 2 ? 60 * 1024 : 0);
    CHK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    float best = 1e9f;
    for (int rep = 0; rep < 4; ++rep) {
      CHK(hipEventRecord(a));
      hipLaunchKernelGGL(kern, dim3(waves / 4), dim3(256), lds, 0, (const float* const*)fdev, rows, N, D, colgroups, sink);
      CHK(hipEventRecord(b)); CHK(hipEventSynchronize(b));
      float ms; CHK(hipEventElapsedTime(&ms, a, b)); best = std::min(best, ms);
    }
    printf("%-66s %7.3f ms  %6.2f TB/s\n", name, best, 4.0 * bytes / best / 1e9);
  };
  run("1 wave/SIMD, 4 B/lane (4 rows x 64 B), 16 loads in flight", gather<1, 4, 1>, 1, 1);
  run("1 wave/SIMD, 4 B/lane, 32 loads in flight", gather<1, 8, 1>, 1, 1);
  run("1 wave/SIMD, 4 B/lane, 60 loads in flight", gather<1, 15, 1>, 1, 1);
  run("2 waves/SIMD, 4 B/lane, 32 loads in flight each", gather<1, 8, 2>, 1, 2);
  run("2 waves/SIMD, 4 B/lane, 60 loads in flight each", gather<1, 15, 2>, 1, 2);
  run("1 wave/SIMD, 8 B/lane (4 rows x 128 B), 16 loads in flight", gather<2, 4, 1>, 2, 1);
  run("1 wave/SIMD, 8 B/lane, 32 loads in flight", gather<2, 8, 1>, 2, 1);
  run("1 wave/SIMD, 8 B/lane, 60 loads in flight", gather<2, 15, 1>, 2, 1);
  run("2 waves/SIMD, 8 B/lane, 32 loads in flight each", gather<2, 8, 2>, 2, 2);
  run("1 wave/SIMD, 16 B/lane (4 rows x 256 B), 32 loads in flight", gather<4, 8, 1>, 4, 1);
  run("2 waves/SIMD, 16 B/lane, 32 loads in flight each", gather<4, 8, 2>, 4, 2);
  // the same bytes per instruction, more rows of narrower pieces: the d-tile stays 16 columns
  run("1 wave/SIMD, 16 B/lane as 8 rows x 128 B, 16 loads in flight", gather<4, 4, 1, 8>, 4, 1, 8);
  run("1 wave/SIMD, 16 B/lane as 8 rows x 128 B, 32 loads in flight", gather<4, 8, 1, 8>, 4, 1, 8);
  run("1 wave/SIMD, 16 B/lane as 16 rows x 64 B, 16 loads in flight", gather<4, 4, 1, 16>, 4, 1, 16);
  run("1 wave/SIMD, 16 B/lane as 16 rows x 64 B, 32 loads in flight", gather<4, 8, 1, 16>, 4, 1, 16);
  run("1 wave/SIMD, 8 B/lane as 8 rows x 64 B, 32 loads in flight", gather<2, 8, 1, 8>, 2, 1, 8);
  run("2 waves/SIMD, 16 B/lane as 16 rows x 64 B, 16 loads in flight each", gather<4, 4, 2, 16>, 4, 2, 16);
  return 0;
}
