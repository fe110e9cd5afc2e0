// Development aid: what does the shape of the sweep's loads cost?  Four [N][D] fp64 (or fp32) arrays are read once,
// rows in class order (a random permutation here), by workgroups of four waves that each own 64 columns x a slice of
// the rows -- the structure of the latitude-class sweeps -- with two lane maps:
//   A  "tile":  a wave owns 16 columns; one load instruction = 4 rows x 128 B (the sweeps as they are)
//   B  "row":   a wave owns one row slot of the batch and all 64 columns; one load instruction = 1 row x 512 B
// Same bytes, same rows per batch (16 per workgroup), same ring depth.  hipcc -O3 --offload-arch=gfx950
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <random>
#include <algorithm>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <typename T> struct F4 { const T* p[4]; };

template <typename T, int MAP, int PD>
__global__ void __launch_bounds__(256, 1)
gather_kernel(F4<T> fp, int64_t D, const int* __restrict__ rows, int nbatch, int nsplit, double* __restrict__ out) {
  const int ndq = (int)((D + 63) / 64);
  const int wg = blockIdx.x;
  if (wg >= ndq * nsplit) return;
  const int dq = wg % ndq, split = wg / ndq;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int b0 = (int)((int64_t)nbatch * split / nsplit), b1 = (int)((int64_t)nbatch * (split + 1) / nsplit);
  // a batch = 16 rows: rows[b * 16 + k * 4 + j], k = class slot, j = member
  int64_t col;
  if (MAP == 0) col = (int64_t)dq * 64 + wave * 16 + (lane & 15);
  else col = (int64_t)dq * 64 + lane;
  if (col >= D) col = D - 1;
  T xb[PD][4][4];
  double acc[4] = {0, 0, 0, 0};
  auto issue = [&](int slot, int b) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      int r;
      if (MAP == 0) r = rows[b * 16 + (lane >> 4) * 4 + j];
      else r = __builtin_amdgcn_readfirstlane(rows[b * 16 + wave * 4 + j]);
      const int64_t off = (int64_t)r * D + col;
#pragma unroll
      for (int f = 0; f < 4; ++f) xb[slot][j][f] = __builtin_nontemporal_load(fp.p[f] + off);
    }
  };
#pragma unroll
  for (int k = 0; k < PD - 1; ++k) if (b0 + k < b1) issue(k, b0 + k);
  for (int b = b0; b < b1; b += PD) {
#pragma unroll
    for (int k = 0; k < PD; ++k) {
      if (b + k < b1) {
        if (b + k + PD - 1 < b1) issue((k + PD - 1) % PD, b + k + PD - 1);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int f = 0; f < 4; ++f) acc[f] += (double)xb[k][j][f];
      }
    }
  }
  out[(int64_t)blockIdx.x * 256 + threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3];
}

template <typename T> __global__ void fill(T* p, int64_t n) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) p[i] = (T)(i & 1023);
}

template <typename T> int run(int64_t N, int64_t D, int reps) {
  F4<T> fp;
  for (int f = 0; f < 4; ++f) { T* p; CHK(hipMalloc(&p, (size_t)N * D * sizeof(T))); hipLaunchKernelGGL(fill<T>, dim3(4096), dim3(256), 0, 0, p, N * D); fp.p[f] = p; }
  std::vector<int> rows((size_t)((N + 15) / 16) * 16);
  for (size_t i = 0; i < rows.size(); ++i) rows[i] = (int)(i % N);
  if (!getenv("UB_SEQUENTIAL")) { std::mt19937 g(3); std::shuffle(rows.begin(), rows.begin() + N, g); }
  int* d_rows; CHK(hipMalloc(&d_rows, rows.size() * 4)); CHK(hipMemcpy(d_rows, rows.data(), rows.size() * 4, hipMemcpyHostToDevice));
  const int nbatch = (int)(rows.size() / 16);
  const int ndq = (int)((D + 63) / 64);
  const int nsplit = std::max(1, (getenv("UB_WGS") ? atoi(getenv("UB_WGS")) : 512) / ndq);   // ~2 workgroups per CU over the run, one resident
  const int grid = ndq * nsplit;
  double* out; CHK(hipMalloc(&out, (size_t)grid * 256 * 8));
  hipEvent_t a, b; CHK(hipEventCreate(&a)); CHK(hipEventCreate(&b));
  const double gb = 4.0 * N * D * sizeof(T) / 1e9;
  const size_t lds = 100 << 10;                     // one workgroup per CU, as the sweeps' LDS use makes it
  auto bench = [&](const char* name, auto kern) {
    CHK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, 0, fp, D, d_rows, nbatch, nsplit, out);
    CHK(hipEventRecord(a));
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, 0, fp, D, d_rows, nbatch, nsplit, out);
    CHK(hipEventRecord(b)); CHK(hipEventSynchronize(b));
    float ms; CHK(hipEventElapsedTime(&ms, a, b)); ms /= reps;
    printf("%-58s %8.3f ms  %6.2f TB/s\n", name, ms, gb / ms);
  };
  printf("N = %lld rows, D = %lld columns, %zu-byte elements, %.2f GB per pass, grid %d (nsplit %d)\n", (long long)N, (long long)D, sizeof(T), gb, grid, nsplit);
  bench("A tile map (4 rows x 16 columns per instruction), ring 2", gather_kernel<T, 0, 2>);
  bench("A tile map, ring 3", gather_kernel<T, 0, 3>);
  bench("A tile map, ring 4", gather_kernel<T, 0, 4>);
  bench("B row map (1 row x 64 columns per instruction), ring 2", gather_kernel<T, 1, 2>);
  bench("B row map, ring 3", gather_kernel<T, 1, 3>);
  bench("B row map, ring 4", gather_kernel<T, 1, 4>);
  return 0;
}

int main(int argc, char** argv) {
  const int64_t N = argc > 1 ? atoll(argv[1]) : 777602;
  const int64_t D = argc > 2 ? atoll(argv[2]) : 2160;
  const int reps = argc > 3 ? atoi(argv[3]) : 5;
  if (argc > 4 && !strcmp(argv[4], "f32")) return run<float>(N, D, reps);
  return run<double>(N, D, reps);
}
