// ubench_gather.hip -- how fast can rows be gathered and summed, as the class sweeps do it?
//   A: 8-byte loads, lane = (row slot g = lane>>4, column c = lane&15): 4 rows x 128 B per instruction
//   B: 16-byte loads, lane = (row slot g = lane>>4, column pair): 4 rows x 256 B per instruction
//   C: 16-byte loads, lane = (row slot g = lane>>5, column pair): 2 rows x 512 B per instruction
// Rows of D doubles in random order (a permutation), 4 "fields", every byte read once.
// build: hipcc -O3 --offload-arch=gfx950 -o ubench_gather ubench_gather.hip ; run: ./ubench_gather
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <random>

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s\n", hipGetErrorString(e_)); exit(1); } } while (0)

// ST > 0: after every 64 rows the wave also stores ST x 512 B (the class-sum record of the one-pass sweep)
template <int W, int RPI, int ST = 0, int NI = 4>   // W = doubles per lane per load (1 or 2), RPI = rows per load instruction, NI = load instructions per field in flight
__global__ void __launch_bounds__(256, 2) gather(const double* const* f, const int* rows, int nrows, int D,
                                                 int colgroups, double* sink, double* out = nullptr) {
  constexpr int LPR = 64 / RPI;            // lanes per row
  constexpr int CPW = LPR * W;             // columns per wave
  const int wave = (blockIdx.x * 4 + (threadIdx.x >> 6));
  const int lane = threadIdx.x & 63;
  const int cg = wave % colgroups, rw = wave / colgroups, nrw = (gridDim.x * 4) / colgroups;
  const int g = lane / LPR, c = (lane % LPR) * W;
  const int col = cg * CPW + c;
  if (col >= D) return;
  double s[4][W];
  for (int i = 0; i < 4; ++i) for (int w = 0; w < W; ++w) s[i][w] = 0.0;
  const int r0 = (int)((long)nrows * rw / nrw), r1 = (int)((long)nrows * (rw + 1) / nrw);
  long rec = ((long)rw * colgroups + cg) * ((nrows / nrw) / 64 + 2);
  int since = 0;
  for (int r = r0; r + NI * RPI <= r1; r += NI * RPI) {      // NI instructions per field in flight
    int rr[NI];
    for (int j = 0; j < NI; ++j) rr[j] = rows[r + j * RPI + g];
    for (int i = 0; i < 4; ++i)
      for (int j = 0; j < NI; ++j) {
        const double* p = f[i] + (long)rr[j] * D + col;
        if (W == 1) s[i][0] += *p;
        else { double2 v = *reinterpret_cast<const double2*>(p); s[i][0] += v.x; s[i][1] += v.y; }
      }
    if (ST > 0) {
      since += NI * RPI;
      if (since >= 64) {
        since = 0;
        if (ST == 7) {          // the same bytes as 7 x 16-byte pairs (what the one-pass sweep stores)
          double2* o2 = reinterpret_cast<double2*>(out + rec * (14 * 64)) + lane;
          for (int k = 0; k < 7; ++k) o2[k * 64] = make_double2(s[k & 3][0] + k, s[k & 3][0] - k);
        } else {
          double* o = out + rec * (ST * 64) + lane;
          for (int k = 0; k < ST; ++k) o[k * 64] = s[k & 3][0] + k;
        }
        ++rec;
      }
    }
  }
  double t = 0.0;
  for (int i = 0; i < 4; ++i) for (int w = 0; w < W; ++w) t += s[i][w];
  if (t == 1.2345e300) sink[0] = t;
}

int main() {
  const int N = 777602, D = 2160;
  const size_t bytes = (size_t)N * D * 8;
  double* fd[4];
  for (int i = 0; i < 4; ++i) { CHK(hipMalloc(&fd[i], bytes)); CHK(hipMemset(fd[i], 0, bytes)); }
  const double** fdev; CHK(hipMalloc(&fdev, 4 * sizeof(double*)));
  CHK(hipMemcpy(fdev, fd, 4 * sizeof(double*), hipMemcpyHostToDevice));
  std::vector<int> perm(N);
  for (int i = 0; i < N; ++i) perm[i] = i;
  std::mt19937 gen(1);
  for (int ordered = 1; ordered >= 0; --ordered) {
    if (!ordered) std::shuffle(perm.begin(), perm.end(), gen);
    int* rows; CHK(hipMalloc(&rows, (N + 64) * sizeof(int)));
    CHK(hipMemcpy(rows, perm.data(), N * sizeof(int), hipMemcpyHostToDevice));
    double* sink; CHK(hipMalloc(&sink, 8));
    hipEvent_t a, b; CHK(hipEventCreate(&a)); CHK(hipEventCreate(&b));
    auto run = [&](const char* name, auto kern, int cpw) {
      const int colgroups = (D + cpw - 1) / cpw;
      const int waves = ((256 * 8 * 4) / colgroups) * colgroups;   // ~8 waves per CU, several rounds
      const int blocks = waves / 4;
      float best = 1e9f;
      for (int rep = 0; rep < 4; ++rep) {
        CHK(hipEventRecord(a));
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, (const double* const*)fdev, rows, N, D, colgroups, sink, (double*)nullptr);
        CHK(hipEventRecord(b)); CHK(hipEventSynchronize(b));
        float ms; CHK(hipEventElapsedTime(&ms, a, b)); best = std::min(best, ms);
      }
      printf("%-8s rows %-8s  %-44s %7.3f ms  %6.2f TB/s\n", ordered ? "ordered" : "shuffled", "", name, best,
             4.0 * bytes / best / 1e9);
    };
    double* outbuf; CHK(hipMalloc(&outbuf, (size_t)16 << 30));
    auto run_st = [&](const char* name, auto kern, int cpw) {
      const int colgroups = (D + cpw - 1) / cpw;
      const int waves = ((256 * 8 * 4) / colgroups) * colgroups;
      const int blocks = waves / 4;
      float best = 1e9f;
      for (int rep = 0; rep < 4; ++rep) {
        CHK(hipEventRecord(a));
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, (const double* const*)fdev, rows, N, D, colgroups, sink, outbuf);
        CHK(hipEventRecord(b)); CHK(hipEventSynchronize(b));
        float ms; CHK(hipEventElapsedTime(&ms, a, b)); best = std::min(best, ms);
      }
      const double wr = (double)N / 64 * colgroups * 14 * 512;
      printf("%-8s rows %-8s  %-44s %7.3f ms  %6.2f TB/s read, %6.2f TB/s read+write\n", ordered ? "ordered" : "shuffled", "",
             name, best, 4.0 * bytes / best / 1e9, (4.0 * bytes + wr) / best / 1e9);
    };
    run("A: 8 B/lane, 4 rows x 128 B per instruction", gather<1, 4>, 16);
    run("B: 16 B/lane, 4 rows x 256 B per instruction", gather<2, 4>, 32);
    run("C: 16 B/lane, 2 rows x 512 B per instruction", gather<2, 2>, 64);
    run("D: 8 B/lane, 1 row x 512 B per instruction", gather<1, 1>, 64);
    run_st("E: as A + 14 x 512 B stored per 64 rows", gather<1, 4, 14>, 16);
    run_st("F: as A + 7 x 1 KB (16 B/lane) per 64 rows", gather<1, 4, 7>, 16);
    run_st("G: as F, 8 load instructions per field in flight", gather<1, 4, 7, 8>, 16);
    run_st("H: as F, 16 load instructions per field in flight", gather<1, 4, 7, 16>, 16);
    CHK(hipFree(rows));
    CHK(hipFree(outbuf));
  }
  return 0;
}
