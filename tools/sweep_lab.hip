// sweep_lab.hip -- bench + cross-check of the forms of sweep 1 of the one-pass class path on a synthetic
// cubed-sphere-like class structure (classes of 8 + 8 member rows, rows in random order), outside the
// library: each form is timed alone with HIP events and its outputs (class-sum records, reduced
// projections) are compared with those of sweep_op_kernel.
//   build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -o sweep_lab tools/sweep_lab.hip
//   run:   ./sweep_lab [N=777602] [D=2160] [reps=5] [f64|f32] [only=<substring>]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <random>
#include <string>
#include <vector>

#include "lab_kernels.hpp"
#include "../pytemdiags_amd/csrc/side_tables.hpp"

using namespace temx;

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

struct HCls { std::vector<int> n, s; };

struct Split { int ndt = 0, nsplit = 0, grid = 0; };
static Split choose_split(int64_t D, int64_t nchunk, int slots, int dpw, int minchunk) {
  Split s;
  s.ndt = (int)((D + 15) / 16);
  const int ndq = (s.ndt + dpw - 1) / dpw;
  int64_t maxsplit = std::max<int64_t>(1, nchunk / minchunk);
  maxsplit = std::min<int64_t>(maxsplit, std::max<int64_t>(1, (int64_t)4 * slots / ndq + 1));
  double best = -1.0; int bestn = 1;
  for (int n = 1; n <= maxsplit; ++n) {
    const int64_t nwg = (int64_t)ndq * n, rounds = (nwg + slots - 1) / slots;
    const double eff = (double)nwg / (double)(rounds * slots);
    if (eff > best * 1.02) { best = eff; bestn = n; }
  }
  s.nsplit = bestn;
  s.grid = (int)((((int64_t)ndq * s.nsplit + 7) / 8) * 8);
  return s;
}

// row table of kernels_cls.hpp (MB member rows per class and batch, northern batches then southern ones)
static void build_crow(const std::vector<HCls>& cls, std::vector<int>& crow, std::vector<int>& gbatch0) {
  constexpr int MB = CLS_MB;
  const int64_t ncls = cls.size(), ng = (ncls + 3) / 4;
  auto nb = [](size_t m) { return (int)((m + MB - 1) / MB); };
  crow.clear(); gbatch0.assign(ng + 1, 0);
  for (int64_t gi = 0; gi < ng; ++gi) {
    int bN = 0, bS = 0;
    for (int k = 0; k < 4; ++k) if (gi * 4 + k < ncls) { bN = std::max(bN, nb(cls[gi * 4 + k].n.size())); bS = std::max(bS, nb(cls[gi * 4 + k].s.size())); }
    gbatch0[gi] = (int)(crow.size() / (4 * MB));
    for (int side = 0; side < 2; ++side) {
      const int nbat = side ? bS : bN;
      for (int bi = 0; bi < nbat; ++bi) {
        int flags = side ? CLS_SOUTH : 0;
        if (bi == 0 && (side == 0 || bN == 0)) flags |= CLS_FIRST;
        if (bi == nbat - 1 && (side == 1 || bS == 0)) flags |= CLS_LAST;
        int batch[4 * MB]; bool haspad = false;
        for (int k = 0; k < 4; ++k)
          for (int j = 0; j < MB; ++j) {
            const size_t m = (size_t)bi * MB + j;
            int ent = (int)0x80000000 | (flags << 28);
            if (gi * 4 + k < ncls) { const auto& mem = side ? cls[gi * 4 + k].s : cls[gi * 4 + k].n; if (m < mem.size()) ent = mem[m] | (flags << 28); }
            haspad = haspad || ent < 0;
            batch[k * MB + j] = ent;
          }
        for (int e = 0; e < 4 * MB; ++e) crow.push_back(batch[e] | (haspad ? CLS_HASPAD_BIT : 0));
      }
    }
  }
  gbatch0[ng] = (int)(crow.size() / (4 * MB));
  crow.resize(crow.size() + (size_t)CLS_PADB * 4 * MB, (int)0x80000000);
}

// row-pair table: [batch][k][h][MBV]; both sides of a class-group are walked together
static void build_crow16(const std::vector<HCls>& cls, int MBV, std::vector<int>& crow, std::vector<int>& gbatch0) {
  const int64_t ncls = cls.size(), ng = (ncls + 3) / 4;
  crow.clear(); gbatch0.assign(ng + 1, 0);
  for (int64_t gi = 0; gi < ng; ++gi) {
    size_t mx = 1;
    for (int k = 0; k < 4; ++k) if (gi * 4 + k < ncls) mx = std::max({mx, cls[gi * 4 + k].n.size(), cls[gi * 4 + k].s.size()});
    const int nbat = (int)((mx + MBV - 1) / MBV);
    gbatch0[gi] = (int)(crow.size() / (8 * MBV));
    for (int bi = 0; bi < nbat; ++bi) {
      int flags = 0;
      if (bi == 0) flags |= CLS_FIRST;
      if (bi == nbat - 1) flags |= CLS_LAST;
      std::vector<int> batch(8 * MBV); bool haspad = false;
      for (int k = 0; k < 4; ++k)
        for (int h = 0; h < 2; ++h)
          for (int j = 0; j < MBV; ++j) {
            const size_t m = (size_t)bi * MBV + j;
            int ent = (int)0x80000000 | (flags << 28);
            if (gi * 4 + k < ncls) { const auto& mem = h ? cls[gi * 4 + k].s : cls[gi * 4 + k].n; if (m < mem.size()) ent = mem[m] | (flags << 28); }
            haspad = haspad || ent < 0;
            batch[(k * 2 + h) * MBV + j] = ent;
          }
      for (int e : batch) crow.push_back(e | (haspad ? CLS_HASPAD_BIT : 0));
    }
  }
  gbatch0[ng] = (int)(crow.size() / (8 * MBV));
  crow.resize(crow.size() + (size_t)CLS_PADB * 8 * MBV, (int)0x80000000);
}

static std::vector<int> group_cuts(const std::vector<int>& gbatch0, int nsub) {
  const int64_t ng = (int64_t)gbatch0.size() - 1, nbatch = gbatch0[ng];
  std::vector<int> cut(2 * (nsub + 1));
  int g = 0;
  const int64_t gc = getenv("LAB_GROUP_COST") ? atoi(getenv("LAB_GROUP_COST")) : 2;   // batches a group end is worth (temx.hip: class_cuts)
  const int64_t total = nbatch + gc * ng;
  for (int k = 0; k <= nsub; ++k) {
    const int64_t want = total * k / nsub;
    while (g < ng && gbatch0[g] + gc * g < want) ++g;
    if (k == nsub) g = (int)ng;
    cut[2 * k] = gbatch0[g]; cut[2 * k + 1] = g;
  }
  return cut;
}

template <typename V> static V* to_dev(const std::vector<V>& v) {
  V* p; CHK(hipMalloc(&p, std::max<size_t>(v.size(), 1) * sizeof(V)));
  CHK(hipMemcpy(p, v.data(), v.size() * sizeof(V), hipMemcpyHostToDevice));
  return p;
}

template <typename T>
__global__ void fill_kernel(T* p, int64_t n, uint32_t seed, double base, double amp) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    uint32_t x = (uint32_t)i * 2654435761u ^ (uint32_t)(i >> 32) * 40503u ^ seed;
    x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;
    p[i] = (T)(base + amp * ((double)(x & 0xFFFFF) / 1048576.0 - 0.5));
  }
}
__global__ void reduce_kernel(const double* partial, int nsplit, int64_t n, double* out) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= n) return;
  double a = 0.0;
  for (int s = 0; s < nsplit; ++s) a += partial[(int64_t)s * n + i];
  out[i] = a;
}
__global__ void maxdiff_kernel(const double* a, const double* b, int64_t n, double* out /* [2]: max |a-b|, max |a| */) {
  double m = 0.0, r = 0.0;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    m = fmax(m, fabs(a[i] - b[i])); r = fmax(r, fabs(a[i]));
  }
  for (int o = 32; o; o >>= 1) { m = fmax(m, __shfl_xor(m, o)); r = fmax(r, __shfl_xor(r, o)); }
  if ((threadIdx.x & 63) == 0) {
    atomicMax(reinterpret_cast<unsigned long long*>(out), (unsigned long long)__double_as_longlong(m));
    atomicMax(reinterpret_cast<unsigned long long*>(out) + 1, (unsigned long long)__double_as_longlong(r));
  }
}

__global__ void reader_kernel(const double2* in, int64_t nrec, double* sink) {
  const int64_t wave = blockIdx.x * (int64_t)(blockDim.x >> 6) + (threadIdx.x >> 6), nw = (int64_t)gridDim.x * (blockDim.x >> 6);
  const int lane = threadIdx.x & 63;
  double a = 0.0;
  for (int64_t r = wave; r < nrec; r += nw) { const double2 v = in[r * 64 + lane]; a += v.x + v.y; }
  if (a == 1.2345e300) sink[0] = a;
}
// LAB_WRITER: a second kernel on its own stream writes `bytes` in 1 KB store instructions, `burst` stores in
// flight per wave, while a sweep built without its class-sum stores (-DTEMX_CSTORE_NONE) runs
__global__ void writer_kernel(double2* out, int64_t nrec /* 1 KB records */, int burst) {
  const int64_t wave = blockIdx.x * (int64_t)(blockDim.x >> 6) + (threadIdx.x >> 6), nw = (int64_t)gridDim.x * (blockDim.x >> 6);
  const int lane = threadIdx.x & 63;
  const double2 v = make_double2((double)wave, (double)lane);
  for (int64_t r = wave * burst; r < nrec; r += nw * burst) {
    for (int k = 0; k < burst; ++k)
      if (r + k < nrec) out[(r + k) * 64 + lane] = v;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
}

template <typename T>
int run(int64_t N, int64_t D, int reps, const char* only) {
  constexpr int TBS = 7, K = 51, NA = 7;
  const int ndt = (int)((D + 15) / 16);
  printf("N = %lld rows, D = %lld columns (%d d-tiles), %s inputs, %.2f GB per sweep\n", (long long)N, (long long)D, ndt,
         sizeof(T) == 8 ? "fp64" : "fp32", 4.0 * N * D * sizeof(T) / 1e9);
  // classes of 8 + 8 rows in random order; what is left over makes one small class
  std::vector<int> perm(N);
  for (int64_t i = 0; i < N; ++i) perm[i] = (int)i;
  std::mt19937 gen(7);
  std::shuffle(perm.begin(), perm.end(), gen);
  std::vector<HCls> cls;
  if (const char* cf = getenv("LAB_CLASSES")) {     // int32 file: ncls, then per class nN, nS, rows (tools/dump_classes.py)
    FILE* fh = fopen(cf, "rb");
    if (!fh) { printf("cannot open %s\n", cf); return 1; }
    int32_t nc = 0;
    if (fread(&nc, 4, 1, fh) != 1) return 1;
    for (int i = 0; i < nc; ++i) {
      int32_t hdr[2];
      if (fread(hdr, 4, 2, fh) != 2) return 1;
      HCls c; c.n.resize(hdr[0]); c.s.resize(hdr[1]);
      if (hdr[0] && fread(c.n.data(), 4, hdr[0], fh) != (size_t)hdr[0]) return 1;
      if (hdr[1] && fread(c.s.data(), 4, hdr[1], fh) != (size_t)hdr[1]) return 1;
      cls.push_back(c);
    }
    fclose(fh);
    printf("classes from %s: %zu\n", cf, cls.size());
  } else
  for (int64_t i = 0; i < N; i += 16) {
    HCls c;
    const int64_t m = std::min<int64_t>(16, N - i);
    for (int64_t j = 0; j < m; ++j) (j < (m + 1) / 2 ? c.n : c.s).push_back(perm[i + j]);
    std::sort(c.n.begin(), c.n.end()); std::sort(c.s.begin(), c.s.end());
    cls.push_back(c);
  }
  int64_t ng = ((int64_t)cls.size() + 3) / 4;
  std::vector<int> crow_file, gb0_file;
  if (const char* cf = getenv("LAB_CROW")) {        // the library's own row table (TEMX_DUMP_CROW): ngroups, entries, gbatch0, crow
    FILE* fh = fopen(cf, "rb");
    if (!fh) { printf("cannot open %s\n", cf); return 1; }
    int32_t hdr[2];
    if (fread(hdr, 4, 2, fh) != 2) return 1;
    gb0_file.resize((size_t)hdr[0] + 1); crow_file.resize((size_t)hdr[1]);
    if (fread(gb0_file.data(), 4, gb0_file.size(), fh) != gb0_file.size()) return 1;
    if (fread(crow_file.data(), 4, crow_file.size(), fh) != crow_file.size()) return 1;
    fclose(fh);
    ng = hdr[0];
    printf("row table from %s: %d class-groups, %d batches\n", cf, hdr[0], gb0_file.back());
  }
#ifdef TEMX_LAB
  CHK(hipMemcpyToSymbol(HIP_SYMBOL(temx_lab_ngr), &ng, sizeof(ng)));
#endif
  std::vector<int> crow, gb0, crow16_2, gb16_2, crow16_4, gb16_4;
  build_crow(cls, crow, gb0);
  if (!crow_file.empty()) { crow = crow_file; gb0 = gb0_file; }
  if (getenv("LAB_PERMUTE")) {      // the same classes, but every row index sent through a random permutation of the rows:
    for (auto& e : crow)            // is it the class structure or the addresses of the members that costs?
      if (e >= 0) e = (e & ~CLS_ROWMASK) | perm[e & CLS_ROWMASK];
    printf("LAB_PERMUTE: row indices of the table permuted\n");
  }
  build_crow16(cls, 2, crow16_2, gb16_2);
  build_crow16(cls, 4, crow16_4, gb16_4);
  int* d_crow = to_dev(crow); int* d_crow16_2 = to_dev(crow16_2); int* d_crow16_4 = to_dev(crow16_4);
  std::vector<double> ycls((size_t)(ng + 1) * 2 * TBS * 16);
  for (auto& v : ycls) v = std::generate_canonical<double, 53>(gen) - 0.5;
  double* d_ycls = to_dev(ycls);
  std::vector<double> cs(D);
  for (int64_t i = 0; i < D; ++i) cs[i] = 1.0 + 0.001 * (i % 72);
  double* d_cs = to_dev(cs);
#ifdef LAB_OFFSETS
  double* early = nullptr;
  CHK(hipMalloc(&early, (size_t)((N + 63) / 64) * ndt * 8 * 64 * 8));
  printf("early buffer %p\n", (void*)early);
#endif
  FieldPtrs<4> fp;
  const double base[4] = {10.0, -3.0, 250.0, 0.05}, amp[4] = {20.0, 10.0, 30.0, 0.2};
  for (int f = 0; f < 4; ++f) {
    T* p; CHK(hipMalloc(&p, (size_t)N * D * sizeof(T)));
    hipLaunchKernelGGL(fill_kernel<T>, dim3(4096), dim3(256), 0, 0, p, N * D, 1234u + f, base[f], amp[f]);
    fp.p[f] = p;
  }
  if (getenv("LAB_SAMEFIELD")) {                    // one array four times: a quarter of the HBM bytes, the same instructions
    for (int f = 1; f < 4; ++f) fp.p[f] = fp.p[0];
    printf("LAB_SAMEFIELD: the four field pointers are one array\n");
  }
  CHK(hipDeviceSynchronize());
  const size_t csum_n = (size_t)ng * ndt * 8 * 64;
  double *csum_ref, *csum_t, *B_ref, *B_t, *partial, *dm;
  CHK(hipMalloc(&csum_ref, csum_n * 8)); CHK(hipMalloc(&csum_t, csum_n * 8));
  const int64_t nB = (int64_t)NA * K * D;
  CHK(hipMalloc(&B_ref, nB * 8)); CHK(hipMalloc(&B_t, nB * 8));
  CHK(hipMalloc(&dm, 16));
  const size_t partial_max = (size_t)520 * nB;
  CHK(hipMalloc(&partial, partial_max * 8));
  hipEvent_t ea, eb; CHK(hipEventCreate(&ea)); CHK(hipEventCreate(&eb));
  bool have_ref = false;
  int bad = 0;

  struct Variant { std::string name; int nsplit; std::function<void(double*)> launch; };
  double *px = nullptr, *pp = nullptr; size_t os_px_n = 0, os_pp_n = 0, os_nsplit = 0;   // outputs of the single-sweep variants
  auto os_fits = [&](int nsplit) {              // a single-sweep variant writes nsplit slabs of px and of pp
    if ((size_t)nsplit <= os_nsplit) return;
    printf("single-sweep variant cuts the work into %d splits, px / pp hold %zu: not launched\n", nsplit, os_nsplit);
    exit(2);
  };
  std::vector<double> os_ref;
  std::vector<Variant> vars;
  const int64_t cunits = std::max<int64_t>(1, gb0[ng] / 4);
  {   // reference: one wave per SIMD, quads of d-tiles
    Split sp = choose_split(D, cunits, getenv("LAB_SLOTS") ? atoi(getenv("LAB_SLOTS")) : 256, 4, 8);
    int2* cuts = reinterpret_cast<int2*>(to_dev(group_cuts(gb0, sp.nsplit)));
    vars.push_back({"op    8B loads, 98 acc, 1 wave/SIMD (library)", sp.nsplit, [=](double* cs_out) {
      hipLaunchKernelGGL((sweep_op_kernel<T, TBS, sizeof(T) == 8 ? 3 : 6, 0>), dim3(sp.grid), dim3(256), 0, 0, fp, D, K, d_ycls,
                         reinterpret_cast<const int4*>(d_crow), cuts, d_cs, partial, sp.nsplit, sp.ndt, cs_out); }});
  }
  auto add_w = [&](auto pdc) {
    constexpr int PD = decltype(pdc)::value;
    Split sp = choose_split(D, cunits / 4, 256 * TEMX_OPW_WPS, 1, 8);
    int2* cuts = reinterpret_cast<int2*>(to_dev(group_cuts(gb0, sp.nsplit * 4)));
    vars.push_back({"opw   8B loads, shared d-tile, " + std::to_string(TEMX_OPW_WPS) + " waves/SIMD, PD=" + std::to_string(PD), sp.nsplit, [=](double* cs_out) {
      hipLaunchKernelGGL((sweep_opw_kernel<T, TBS, PD, 0>), dim3(sp.grid), dim3(256), 0, 0, fp, D, K, d_ycls,
                         reinterpret_cast<const int4*>(d_crow), cuts, d_cs, partial, sp.nsplit, sp.ndt, cs_out, (double*)nullptr); }});
  };
  add_w(std::integral_constant<int, sizeof(T) == 8 ? 2 : 4>{});
  add_w(std::integral_constant<int, sizeof(T) == 8 ? 3 : 6>{});
#ifdef LAB_OS
  // the store-free single sweep (timing only here: random basis and reference tables)
  {
    constexpr int TBX = 13, KX = 101, K4r = 52;
    static double *ycx = nullptr, *rho = nullptr;
    if (!ycx) {
      std::vector<double> h((size_t)(ng + 1) * 2 * TBX * 16);
      for (auto& v : h) v = std::generate_canonical<double, 53>(gen) - 0.5;
      ycx = to_dev(h);
      std::vector<double> r((size_t)4 * K4r * D);
      for (auto& v : r) v = std::generate_canonical<double, 53>(gen) - 0.5;
      rho = to_dev(r);
      // px / pp hold one slab per split of the variant that is launched: every single-sweep variant below cuts the work
      // with this same choose_split (os_fits() checks it before each launch).  Round 3's first version sized them for 16
      // splits; ne240 x 128 (D = 128) runs 127, and the stores of the first osr variant ran 8x past the buffers -- the
      // "write access to a read-only page" fault of that round's lab15 log.  The library sizes `partial` from sp_os.nsplit.
      os_nsplit = (size_t)choose_split(D, cunits, getenv("LAB_SLOTS") ? atoi(getenv("LAB_SLOTS")) : 256, 4, 8).nsplit;
      os_px_n = os_nsplit * 4 * KX * D; os_pp_n = os_nsplit * 3 * K * D;
      CHK(hipMalloc(&px, os_px_n * 8));
      CHK(hipMalloc(&pp, os_pp_n * 8));
      CHK(hipMemset(px, 0, os_px_n * 8)); CHK(hipMemset(pp, 0, os_pp_n * 8));
    }
    auto add_os = [&](auto nbrc, auto pdc, auto dfc) {
      constexpr int NBR = decltype(nbrc)::value, PD = decltype(pdc)::value;
      constexpr int DF = decltype(dfc)::value;
      Split sp = choose_split(D, cunits, getenv("LAB_SLOTS") ? atoi(getenv("LAB_SLOTS")) : 256, 4, 8);
      int2* cuts = reinterpret_cast<int2*>(to_dev(group_cuts(gb0, sp.nsplit)));
      os_fits(sp.nsplit);
      const size_t ldsb = ((size_t)4 * (DF ? 2 : 1) * 2 * TBX * 16 + (size_t)4 * 4 * 2 * NBR * 64 + (size_t)4 * 3 * 2 * TBS * 64) * 8;
      auto kern = sweep_os_kernel<T, TBS, TBX, NBR, PD, 0, DF>;
      CHK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb));
      double *ycx_ = ycx, *rho_ = rho, *px_ = px, *pp_ = pp;
      vars.push_back({"os    ONE sweep, no class-sum stream: 4 x 26 + 3 x 14 accumulators, reference blocks " + std::to_string(NBR) +
                      ", PD=" + std::to_string(PD) + (DF ? ", deferred projection " + std::to_string(DF) : std::string()), 0, [=](double*) {
        hipLaunchKernelGGL(kern, dim3(sp.grid), dim3(256), ldsb, 0, fp, D, K, KX, ycx_, reinterpret_cast<const int4*>(d_crow), cuts,
                           d_cs, rho_, K4r, px_, pp_, sp.nsplit, sp.ndt); }});
    };
    auto add_osr = [&](auto nbrc, auto pdc) {
      constexpr int NBR = decltype(nbrc)::value, PD = decltype(pdc)::value;
      Split sp = choose_split(D, cunits, getenv("LAB_SLOTS") ? atoi(getenv("LAB_SLOTS")) : 256, 4, 8);
      int2* cuts = reinterpret_cast<int2*>(to_dev(group_cuts(gb0, sp.nsplit)));
      os_fits(sp.nsplit);
      const size_t ldsb = ((size_t)2 * 2 * TBX * 16 + 16 + (size_t)4 * 4 * 2 * NBR * 64 + (size_t)4 * 3 * 2 * TBS * 64 + (size_t)14 * 256) * 8;
      auto kern = sweep_osr_kernel<T, TBS, TBX, NBR, PD, 0>;
      CHK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb));
      double *ycx_ = ycx, *rho_ = rho, *px_ = px, *pp_ = pp;
      vars.push_back({"osr   ONE sweep, loads of 1 row x 64 columns, exchange at the group's end, PD=" + std::to_string(PD), 0, [=](double*) {
        hipLaunchKernelGGL(kern, dim3(sp.grid), dim3(256), ldsb, 0, fp, D, K, KX, ycx_, reinterpret_cast<const int4*>(d_crow), cuts,
                           d_cs, rho_, K4r, px_, pp_, sp.nsplit, sp.ndt); }});
    };
    static int *d_crowN = nullptr, *d_crowS = nullptr, *d_gfN = nullptr, *d_gfS = nullptr;
    if (!d_crowN) {
      SideTables stb;
      build_side_tables(crow, gb0, ng, CLS_MB, CLS_PADB, CLS_SOUTH, CLS_FIRST, CLS_LAST, CLS_HASPAD_BIT, stb);
      d_crowN = to_dev(stb.crow[0]); d_crowS = to_dev(stb.crow[1]); d_gfN = to_dev(stb.gfirst[0]); d_gfS = to_dev(stb.gfirst[1]);
      printf("side tables: %d northern and %d southern batches\n", stb.gfirst[0].back(), stb.gfirst[1].back());
    }
    auto add_os2 = [&](auto nbrc, auto pdc) {
      constexpr int NBR = decltype(nbrc)::value, PD = decltype(pdc)::value;
      Split sp = choose_split(D, cunits, getenv("LAB_SLOTS") ? atoi(getenv("LAB_SLOTS")) : 256, 4, 8);
      int2* cuts = reinterpret_cast<int2*>(to_dev(group_cuts(gb0, sp.nsplit)));
      os_fits(sp.nsplit);
      const size_t ldsb = ((size_t)2 * 2 * TBX * 16 + 16 + (size_t)4 * 4 * 2 * NBR * 64 + (size_t)8 * 3 * TBS * 64 + (size_t)7 * 512) * 8;
      auto kern = sweep_os2_kernel<T, TBS, TBX, NBR, PD, 0>;
      CHK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb));
      double *ycx_ = ycx, *rho_ = rho, *px_ = px, *pp_ = pp;
      vars.push_back({"os2   ONE sweep, 2 waves per SIMD: a wave per class side reads, a wave per d-tile and parity projects, PD=" + std::to_string(PD), 0, [=](double*) {
        hipLaunchKernelGGL(kern, dim3(sp.grid), dim3(512), ldsb, 0, fp, D, K, KX, ycx_, reinterpret_cast<const int4*>(d_crowN),
                           reinterpret_cast<const int4*>(d_crowS), d_gfN, d_gfS, cuts, d_cs, rho_, K4r, px_, pp_, sp.nsplit, sp.ndt); }});
    };
    add_os2(std::integral_constant<int, 2>{}, std::integral_constant<int, 2>{});
    add_os2(std::integral_constant<int, 2>{}, std::integral_constant<int, 3>{});
    add_osr(std::integral_constant<int, 2>{}, std::integral_constant<int, 2>{});
    add_osr(std::integral_constant<int, 2>{}, std::integral_constant<int, 3>{});
    add_osr(std::integral_constant<int, 2>{}, std::integral_constant<int, 4>{});     // (vmcnt counts to 63: 3 x 16 + 12 loads)
    using N0 = std::integral_constant<int, 0>; using N1 = std::integral_constant<int, 1>;
    add_os(std::integral_constant<int, 2>{}, std::integral_constant<int, sizeof(T) == 8 ? 2 : 4>{}, N0{});
    add_os(std::integral_constant<int, 2>{}, std::integral_constant<int, sizeof(T) == 8 ? 2 : 4>{}, N1{});
#ifndef LAB_OS_FEW
    add_os(std::integral_constant<int, 2>{}, std::integral_constant<int, sizeof(T) == 8 ? 3 : 6>{}, N0{});
#endif
  }
#endif
#ifdef LAB_FUSED
  // TEM + one tracer in one sweep: five fields, ten projections (its outputs are not compared here: the
  // library's tests do that; slabs differ from the reference kernel's seven)
  static T* qf = nullptr;
  if (!qf) {
    CHK(hipMalloc(&qf, (size_t)N * D * sizeof(T)));
    hipLaunchKernelGGL(fill_kernel<T>, dim3(4096), dim3(256), 0, 0, qf, N * D, 99u, 1e-3, 1e-3);
    CHK(hipDeviceSynchronize());
  }
  static double* csq = nullptr;
  if (!csq) CHK(hipMalloc(&csq, csum_n * 8 / 4));
  auto add_f = [&](auto pdc) {
    constexpr int PD = decltype(pdc)::value;
    Split sp = choose_split(D, cunits / 4, getenv("LAB_SLOTS") ? atoi(getenv("LAB_SLOTS")) : 256, 1, 8);
    int2* cuts = reinterpret_cast<int2*>(to_dev(group_cuts(gb0, sp.nsplit * 4)));
    FieldPtrs<5> f5;
    for (int i = 0; i < 4; ++i) f5.p[i] = fp.p[i];
    f5.p[4] = qf;
    double* csq_ = csq;
    vars.push_back({"opw2  TEM + tracer, 5 fields, shared d-tile, 1 wave/SIMD, PD=" + std::to_string(PD) + " [53.7 GB counted, 67.2 read]", sp.nsplit * 10 / 7 + 1, [=](double* cs_out) {
      hipLaunchKernelGGL((sweep_opw_kernel<T, TBS, PD, 2>), dim3(sp.grid), dim3(256), 0, 0, f5, D, K, d_ycls,
                         reinterpret_cast<const int4*>(d_crow), cuts, d_cs, partial, sp.nsplit, sp.ndt, cs_out, csq_); }});
  };
  add_f(std::integral_constant<int, sizeof(T) == 8 ? 2 : 4>{});
  add_f(std::integral_constant<int, sizeof(T) == 8 ? 3 : 6>{});
#endif
  {   // parity pair with redundant loads: 2 d-tiles per workgroup
    Split sp = choose_split(D, cunits, 512, 2, 8);
    int2* cuts = reinterpret_cast<int2*>(to_dev(group_cuts(gb0, sp.nsplit)));
    vars.push_back({"opp   8B loads, parity pair + redundant loads, 2 waves/SIMD, PD=2", sp.nsplit, [=](double* cs_out) {
      hipLaunchKernelGGL((sweep_opp_kernel<T, TBS, sizeof(T) == 8 ? 2 : 4, 0>), dim3(sp.grid), dim3(256), 0, 0, fp, D, K, d_ycls,
                         reinterpret_cast<const int4*>(d_crow), cuts, d_cs, partial, sp.nsplit, sp.ndt, cs_out); }});
  }
  auto add_16 = [&](auto mbc, auto pdc, auto wsc, auto rbc) {
    constexpr int MBV = decltype(mbc)::value, PD = decltype(pdc)::value, WS = decltype(wsc)::value, RB = decltype(rbc)::value;
    const std::vector<int>& gb = MBV == 2 ? gb16_2 : gb16_4;
    const int* tab = MBV == 2 ? d_crow16_2 : d_crow16_4;
    const int64_t units = std::max<int64_t>(1, (int64_t)gb[ng] * MBV / 8);
    Split sp = WS == 1 ? choose_split(D, units, 256, 4, 8) : choose_split(D, units / 4, 512, 1, 8);
    int2* cuts = reinterpret_cast<int2*>(to_dev(group_cuts(gb, sp.nsplit * WS)));
    vars.push_back({std::string("op16  row pairs, ") + (WS == 1 ? "98 acc, 1 wave/SIMD" : "shared d-tile, 2 waves/SIMD") + ", MB=" +
                        std::to_string(MBV) + " PD=" + std::to_string(PD) + (RB ? " RB=" + std::to_string(RB) : std::string()), sp.nsplit, [=](double* cs_out) {
      hipLaunchKernelGGL((sweep_op16_kernel<T, TBS, MBV, PD, 0, WS, RB>), dim3(sp.grid), dim3(256), 0, 0, fp, D, K, d_ycls, tab, cuts,
                         d_cs, partial, sp.nsplit, sp.ndt, cs_out); }});
  };
  using I1 = std::integral_constant<int, 1>; using I2 = std::integral_constant<int, 2>; using I3 = std::integral_constant<int, 3>;
  using I4 = std::integral_constant<int, 4>; using I5 = std::integral_constant<int, 5>; using I6 = std::integral_constant<int, 6>;
  using I0 = std::integral_constant<int, 0>; using I8 = std::integral_constant<int, 8>;
  add_16(I2{}, I4{}, I1{}, I0{});
  add_16(I2{}, I4{}, I1{}, I4{});
  add_16(I2{}, I4{}, I1{}, I8{});
  add_16(I2{}, I5{}, I1{}, I0{});
  add_16(I2{}, I3{}, I4{}, I0{});

  for (auto& v : vars) {
    if (have_ref && only && !strstr(v.name.c_str(), only)) continue;
    if ((size_t)v.nsplit * nB > partial_max) { printf("%s: nsplit %d too large\n", v.name.c_str(), v.nsplit); continue; }
    double* cso = have_ref ? csum_t : csum_ref;
    CHK(hipMemset(cso, 0, csum_n * 8));
    v.launch(cso);
    CHK(hipGetLastError());
    CHK(hipDeviceSynchronize());
    if (v.nsplit > 0)
      hipLaunchKernelGGL(reduce_kernel, dim3((unsigned)((nB + 255) / 256)), dim3(256), 0, 0, partial, v.nsplit, nB, have_ref ? B_t : B_ref);
    float best = 1e9f, sum = 0.f;
    for (int r = 0; r < reps; ++r) {
      CHK(hipEventRecord(ea));
      v.launch(cso);
      CHK(hipEventRecord(eb)); CHK(hipEventSynchronize(eb));
      float ms; CHK(hipEventElapsedTime(&ms, ea, eb));
      best = std::min(best, ms); sum += ms;
    }
    CHK(hipGetLastError());
    if (px && !strncmp(v.name.c_str(), "os", 2)) {     // the single-sweep variants against the first of them
      std::vector<double> h(os_px_n + os_pp_n);
      CHK(hipMemcpy(h.data(), px, os_px_n * 8, hipMemcpyDeviceToHost));
      CHK(hipMemcpy(h.data() + os_px_n, pp, os_pp_n * 8, hipMemcpyDeviceToHost));
      if (os_ref.empty()) os_ref = h;
      double dm_ = 0.0, rm_ = 0.0;
      for (size_t i = 0; i < h.size(); ++i) { dm_ = std::max(dm_, std::fabs(h[i] - os_ref[i])); rm_ = std::max(rm_, std::fabs(os_ref[i])); }
      printf("    outputs vs the first single-sweep variant: max |diff| / max |ref| = %.2e (max |ref| %.3e)\n", dm_ / rm_, rm_);
      if (!(dm_ <= 1e-11 * rm_)) ++bad;
      CHK(hipMemset(px, 0, os_px_n * 8)); CHK(hipMemset(pp, 0, os_pp_n * 8));
    }
#ifdef LAB_OFFSETS
    if (!have_ref) {   // does the placement of the class-sum buffer relative to the fields matter?
      static double* big = nullptr;
      if (!big) CHK(hipMalloc(&big, csum_n * 8 + ((size_t)80 << 20)));
      printf("fields %p %p %p %p  csum_ref %p csum_t %p late %p partial %p\n", fp.p[0], fp.p[1], fp.p[2], fp.p[3], (void*)csum_ref, (void*)csum_t, (void*)big, (void*)partial);
      auto tm = [&](const char* what, double* buf) {
        float sum2 = 0.f;
        for (int r = 0; r < 3; ++r) {
          CHK(hipEventRecord(ea));
          v.launch(buf);
          CHK(hipEventRecord(eb)); CHK(hipEventSynchronize(eb));
          float ms; CHK(hipEventElapsedTime(&ms, ea, eb)); sum2 += ms;
        }
        const int64_t nrec = (int64_t)csum_n * 8 / 1024;
        float wms = 0.f, rms = 0.f, ms;
        for (int r = 0; r < 2; ++r) {
          CHK(hipEventRecord(ea));
          hipLaunchKernelGGL(writer_kernel, dim3(2048), dim3(256), 0, 0, reinterpret_cast<double2*>(buf), nrec, 4);
          CHK(hipEventRecord(eb)); CHK(hipEventSynchronize(eb)); CHK(hipEventElapsedTime(&ms, ea, eb)); wms = ms;
          CHK(hipEventRecord(ea));
          hipLaunchKernelGGL(reader_kernel, dim3(2048), dim3(256), 0, 0, reinterpret_cast<const double2*>(buf), nrec, dm);
          CHK(hipEventRecord(eb)); CHK(hipEventSynchronize(eb)); CHK(hipEventElapsedTime(&ms, ea, eb)); rms = ms;
        }
        printf("    class sums in %-40s %7.3f ms   (buffer alone: written in %.3f ms = %.2f TB/s, read in %.3f ms = %.2f TB/s)\n", what, sum2 / 3,
               wms, csum_n * 8.0 / wms / 1e9, rms, csum_n * 8.0 / rms / 1e9);
      };
      tm("the buffer allocated before the fields", early);
      tm("csum_ref (after the fields, memset)", csum_ref);
      tm("csum_t (after the fields, never touched)", csum_t);
      tm("a late buffer (after everything)", big);
      tm("the late buffer + 33 MB", big + ((size_t)33 << 20) / 8);
      CHK(hipMemset(big, 0, csum_n * 8));
      tm("the late buffer after a memset", big);
      tm("the early buffer again", early);
    }
#endif
#ifdef LAB_WRITER
    {
      static hipStream_t s2 = nullptr; static hipEvent_t wa, wb;
      if (!s2) { CHK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking)); CHK(hipEventCreate(&wa)); CHK(hipEventCreate(&wb)); }
      const int64_t nrec = (int64_t)csum_n * 8 / 1024;
      for (int burst : {4, 8, 16}) for (int wpc : {1, 2}) {
        CHK(hipDeviceSynchronize());
        CHK(hipEventRecord(wa, s2));
        hipLaunchKernelGGL(writer_kernel, dim3(256 * wpc), dim3(64), 0, s2, reinterpret_cast<double2*>(csum_t), nrec, burst);
        CHK(hipEventRecord(wb, s2));
        CHK(hipEventRecord(ea));
        v.launch(cso);
        CHK(hipEventRecord(eb)); CHK(hipEventSynchronize(eb)); CHK(hipEventSynchronize(wb));
        float ms, wms; CHK(hipEventElapsedTime(&ms, ea, eb)); CHK(hipEventElapsedTime(&wms, wa, wb));
        printf("    with a concurrent writer (%d waves/CU, %2d x 1 KB in flight each): sweep %7.3f ms, writer %7.3f ms (%.2f TB/s of writes)\n",
               wpc, burst, ms, wms, csum_n * 8.0 / wms / 1e9);
      }
    }
#endif
    double eB = 0.0, eC = 0.0;
    const bool compared = have_ref && v.nsplit > 0 && strncmp(v.name.c_str(), "opw2", 4) != 0;   // (single-sweep variants: among themselves, above; opw2: ten slabs, the library's tests)
    if (compared) {
      double h[2];
      CHK(hipMemset(dm, 0, 16));
      hipLaunchKernelGGL(maxdiff_kernel, dim3(1024), dim3(256), 0, 0, B_ref, B_t, nB, dm);
      CHK(hipMemcpy(h, dm, 16, hipMemcpyDeviceToHost)); eB = h[0] / h[1];
      CHK(hipMemset(dm, 0, 16));
      hipLaunchKernelGGL(maxdiff_kernel, dim3(4096), dim3(256), 0, 0, csum_ref, csum_t, (int64_t)csum_n, dm);
      CHK(hipMemcpy(h, dm, 16, hipMemcpyDeviceToHost)); eC = h[0] / h[1];
      if (!(eB < 1e-11) || !(eC < 1e-12)) ++bad;
    }
    const double bytes = 4.0 * N * D * sizeof(T);
    printf("%-70s nsplit %3d  avg %7.3f ms  min %7.3f ms  %5.2f TB/s  (%.3f of 8)", v.name.c_str(), v.nsplit,
           sum / reps, best, bytes / (sum / reps) / 1e9, bytes / (sum / reps) / 1e9 / 8.0);
    if (compared) printf("  dB %.1e dcsum %.1e\n", eB, eC);
    else printf(have_ref ? "  (not compared with the library kernel)\n" : "  (the reference)\n");
    fflush(stdout);
    have_ref = true;
  }
  printf(bad ? "MISMATCH in %d compared variant(s)\n" : "all compared variants agree\n", bad);
  return bad;
}

int main(int argc, char** argv) {
  const int64_t N = argc > 1 ? atoll(argv[1]) : 777602;
  const int64_t D = argc > 2 ? atoll(argv[2]) : 2160;
  const int reps = argc > 3 ? atoi(argv[3]) : 5;
  const bool f32 = argc > 4 && !strcmp(argv[4], "f32");
  const char* only = argc > 5 ? argv[5] : nullptr;
  return f32 ? run<float>(N, D, reps, only) : run<double>(N, D, reps, only);
}
