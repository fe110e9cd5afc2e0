// temx.hip -- host side of libtemx.so: plan, workspace, launch logic and the C ABI of
// include/temx.h.  gfx950 only.  Build: see csrc/Makefile (hipcc --offload-arch=gfx950).
#include "../../include/temx.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <new>
#include <stdexcept>
#include <string>
#include <thread>
#include <atomic>
#include <vector>

#include "kernels.hpp"
#include "kernels_sym.hpp"
#include "kernels_cls.hpp"
#include "kernels_op.hpp"
#include "kernels_op2.hpp"
#include "kernels_osc.hpp"
#include "side_tables.hpp"

using namespace temx;

// ------------------------------------------------------------------------------------------------
// errors
// ------------------------------------------------------------------------------------------------
static thread_local std::string g_err;

static int fail(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_err = buf;
  return code;
}

#define HIPCHK(expr)                                                                     \
  do {                                                                                   \
    hipError_t e_ = (expr);                                                              \
    if (e_ != hipSuccess)                                                                \
      return fail(e_ == hipErrorOutOfMemory ? TEMX_ENOMEM : TEMX_EHIP, "%s failed: %s",  \
                  #expr, hipGetErrorString(e_));                                         \
  } while (0)

// physical constants of the reference (PyTEMDiags/constants.py:6-14)
static const double kR = 287.058, kCp = 1004.64, kOm = 7.29212e-5;

// ------------------------------------------------------------------------------------------------
// plan
// ------------------------------------------------------------------------------------------------
struct DevBuf {
  void* p = nullptr;
  size_t bytes = 0;
  int ensure(size_t need) {
    if (need <= bytes) return TEMX_OK;
    if (p) (void)hipFree(p);
    p = nullptr;
    bytes = 0;
    hipError_t e = hipMalloc(&p, need);
    if (e != hipSuccess) return fail(TEMX_ENOMEM, "hipMalloc(%zu) failed: %s", need, hipGetErrorString(e));
    bytes = need;
    return TEMX_OK;
  }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    bytes = 0;
  }
  double* d() const { return static_cast<double*>(p); }
};

struct Split {
  int ndt = 0, nsplit = 0, grid = 0, dpw = 4;   // dpw: d-tiles per workgroup
};

struct TimedLaunch {
  hipEvent_t a, b;
};

struct temx_plan {
  int device = 0, num_cu = 256;
  int64_t N = 0, nchunk = 0;
  int L = 0, K = 0, TB = 0, K4 = 0, M = 0;
  // harmonics beyond one fused sweep (K > 64): slices of 16 blocks, `stride` blocks stored per group
  int stride = 0, nslice = 1;
  bool large = false;
  DevBuf Bs, XB, P3;             // large-L path: slice sums, native means [4][N][D], products [3][N][D]
  bool finalized = false;
  bool weighted = false;      // weights mode (temx_plan_set_weights): projection rows are scaled, reconstruction rows are not
  int rank = 0;               // numerical rank of Y0 (== K unless the pseudo-inverse fallback ran)
  DevBuf x, Y0, yblk, yblk_w, Y0p, G, Ginv, norm, flag;
  DevBuf gblk, ypblk;            // Ginv / Qp as 4x4 MFMA A-operand blocks (solve_mfma_kernel, K <= 64)
  // Projection basis (temx_plan_finalize): qbasis = the sweeps project on Q = Y0 R^-1, R the Cholesky factor of
  // the Gram matrix, T = R^-1 on the device.  Qp = Y0p T is the operand that takes coefficients to the output
  // latitudes (Y0p itself stays the attribute), Ginv the inverse of the second Gram matrix Q^T Q (the identity up
  // to rounding), GinvA = T T^T the inverse of the Gram matrix of Y0 (attributes Y0inv / sanity numbers only).
  bool qbasis = false;
  bool g_checker = false;      // the Gram matrix the plan was finalised with is a checkerboard (odd entries cleared)
  bool ext_G = false;          // finalised with a Gram matrix from outside (ncol-sharded: the all-reduced one)
  bool os_need_global = false; // single sweep: this rank's own subsample cannot be fitted, the job's matrices are awaited
  DevBuf T, Qp, GinvA, G2, xo, xc;
  int64_t cls_npad = 0;
  const double* yproj_ptr() const { return yblk_w.p ? yblk_w.d() : yblk.d(); }
  std::vector<double> lat_out_deg;
  // TEM configuration
  bool tem = false;
  int nlev = 0;
  int64_t nt = 0, D = 0;
  double p0 = 101325.0;
  DevBuf p, pg, lg, coslat, fcor, colscale;
  DevBuf B4, B3, C4, zb;
  DevBuf Bq, Bq2, Ct, tz;          // tracer workspace: sums, coefficients (q, v, w), zonal means
  Split sp_proj1;
  // mirror-paired path (equatorially symmetric grids), see kernels_sym.hpp
  bool sym = false;
  int TBS = 0;
  int64_t npair = 0, npg = 0, npg_alloc = 0;
  DevBuf rows, ysym;
  Split sp_sproj4, sp_sproj1, sp_seddy;
  Split sp_proj4, sp_eddy;
  // latitude-class path (columns sharing a latitude share a basis row), see kernels_cls.hpp
  bool cls = false;
  int64_t cgroups = 0, cbatches = 0, ncls = 0;
  std::vector<int> gbatch0;            // first batch of every class-group (+ total)
  DevBuf crow, ycls;
  std::map<int, DevBuf> csplits;       // work cuts per number of pieces
  Split sp_cproj4, sp_cproj1, sp_ceddy, sp_cflux;
  // single-sweep form (kernels_op2.hpp, sweep_os_kernel): no class-sum stream; see build_os_tables / tem_run_os
  bool os_built = false, os_on = false;
  bool os_valid = false;               // Ax / rho / C4 are those of the latest single-sweep temx_tem_run (its v, omega serve the tracer)
  DevBuf Axq, rho_t;                   // tracer in the single-sweep form: [KX + 2 K][D] projections (+ [KR][D] pre-pass sums), references of (q, v, omega)
  int TBX = 0, KX = 0, KR = 0, NQ = 0;
  std::vector<double> h_xc, h_cnt;     // host copies of the class latitudes (cos colat) and member counts
  std::vector<int> h_crow;             // host copy of the row table
  std::vector<int> sgbatch0;           // subsample of class-groups (reference pre-pass): first batch of each (+ total)
  int64_t sgroups = 0, sbatches = 0;
  DevBuf ycx, ycx_s, crow_s, rho, rho0, gaunt /* Yq[NQ][KX] */, wq2, Gx, Gsinv, Ax /* [4 KX + 3 K][D]: projections of the fields, then of the products */, Axs /* [4][KR][D] */;
  std::vector<double> h_Gx, h_Gs;      // this plan's rows: Y0^T Y0ext [K][KX] and the subsample's Gram matrix [KR][KR] (all-reduced when ncol-sharded)
  // contraction of the single sweep on the matrix cores (kernels_osc.hpp): host copies of its matrices, their 4x4
  // blocks on the device (one buffer), the synthesised fields At / ab [4][NQ][D] (kept: the tracer pairs with v, omega)
  std::vector<double> h_T, h_Ginv, h_G, h_Yq;
  DevBuf oscblk, osAt, osAb, osAtq, osAbq, Bqp;
  OscMats osc{};
  bool osc_lds = false;                // TEMX_OPT_OS_CONTRACT = 1 / TEMX_OS_CONTRACT=lds: the round-3 LDS form (A/B)
  int opt_os_contract = -1;
  int os_keep = 32;                    // class-groups of the reference subsample (TEMX_OPT_OS_SUBSAMPLE): 128 latitudes for 16 coefficients per column
  // what the tail of the pipeline (solve, contraction, scan, epilogue) currently describes: the snapshots
  // [tt0, tt0 + tnt) of the run, tD = nlev * tnt columns.  The whole run unless a time-sliced tail ran last.
  int64_t tD = 0, tnt = 0, tt0 = 0;
  // path selection (temx_plan_configure; the TEMX_* environment variables override): -1 = automatic
  int opt_form = -1, opt_os_map = -1, opt_op_map = -1, opt_tracer_one_pass = -1, opt_single_sweep_min_groups = -1;
  bool os_tile = false, op_tile = false;   // effective lane map of the loads (fixed in temx_plan_set_tem)
  bool no_qr = false;                      // TEMX_NO_QR (flag of temx_plan_create or environment)
  DevBuf side_crow[2][2], side_gfirst[2][2];   // [full table, subsample][north, south]: side_tables.hpp (sweep_os2_kernel)
  std::map<int, DevBuf> csplits_s;
  Split sp_os, sp_os_s;
  Split sp_copw;                 // TEM + tracer in one sweep (sweep_opw_kernel<.., 2>): one d-tile per workgroup
  // one-pass form of the class path: sweep 1 also stores per-class sums of products (csum), the
  // flux kernel replaces sweep 2 (kernels_cls.hpp)
  // large-L class path (64 < K <= 256 with latitude classes): class sums first, then sliced work on the sums
  bool lcls = false, lone = false, xb_valid = false;
  DevBuf ycls_l, pbuf;
  int64_t ycls_lstride = 0;
  Split sp_lflux;
  // op_valid: csum (and Pq) hold the class sums of the latest temx_tem_stage1 on this plan
  bool onepass = false, op_valid = false;
  // c4_valid: C4 holds the coefficients of the fields of the latest temx_tem_stage1 (a stage-2 solve has run since)
  bool c4_valid = false;
  DevBuf csum, ccnt;
  DevBuf Pq;                     // [3][K][D] projections of the co-moments of u v, u omega, v theta (sweep 1)
  // one-pass tracer: class sums of q ([groups][d-tiles][64] {north, south} pairs) and the projected
  // co-moments of q v, q omega; tq_valid: they describe the latest temx_tracer_stage1_sums
  DevBuf csq, Pq2;
  bool tq_valid = false;
  // shared workspaces
  DevBuf partial;
  // operator-API workspace (any D)
  DevBuf opB, opC;
  // timing hooks
  bool timing = false;
  std::vector<TimedLaunch> timed[2];
};

static int upload(DevBuf& b, const void* src, size_t bytes) {
  int rc = b.ensure(bytes);
  if (rc) return rc;
  HIPCHK(hipMemcpy(b.p, src, bytes, hipMemcpyHostToDevice));
  return TEMX_OK;
}

// A buffer that a sweep streams into while it reads the fields.  On MI355X the write bandwidth of a large
// allocation is bimodal with its physical placement -- a zero fill of 6.7 GB runs at 5.3-5.6 or at 6.3-6.5
// TB/s, reads do not care -- and a sweep whose class sums go to a slow one loses ~5 % (tools/sweep_lab.hip
// built with -DLAB_OFFSETS, profiles/r03_lab_csum_placement.log).  So: allocate up to TEMX_PLACE_TRIES
// candidates (held together, so that each comes from other memory), time a zero fill of each, keep the
// fastest.  The fill is also the initialisation the callers need.  Small buffers are not probed.
__global__ void fill_zero_kernel(double2* __restrict__ p, int64_t n) {
  const double2 z = make_double2(0.0, 0.0);
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) p[i] = z;
}

static int alloc_write_stream(DevBuf& b, size_t need) {
  need = (need + 15) & ~(size_t)15;
  if (b.bytes >= need) return TEMX_OK;          // (kept from an earlier configuration: already chosen)
  int tries = 4;
  if (const char* e = getenv("TEMX_PLACE_TRIES")) tries = std::max(1, std::min(8, atoi(e)));
  if (need < ((size_t)1 << 30)) tries = 1;
  b.release();
  std::vector<DevBuf> cand;
  std::vector<float> ms;
  hipEvent_t ea = nullptr, eb = nullptr;
  if (tries > 1 && (hipEventCreate(&ea) != hipSuccess || hipEventCreate(&eb) != hipSuccess)) tries = 1;
  int best = -1;
  float slowest = 0.f;
  for (int t = 0; t < tries; ++t) {
    size_t fr = 0, tot = 0;
    if (t > 0 && (hipMemGetInfo(&fr, &tot) != hipSuccess || need > fr / 2)) break;
    DevBuf c;
    if (c.ensure(need) != TEMX_OK) {
      if (t == 0) return TEMX_ENOMEM;
      break;
    }
    float best_ms = 1e30f;
    for (int rep = 0; rep < (tries > 1 ? 2 : 1); ++rep) {
      if (ea) (void)hipEventRecord(ea, nullptr);
      hipLaunchKernelGGL(fill_zero_kernel, dim3(2048), dim3(256), 0, nullptr, static_cast<double2*>(c.p), (int64_t)(need / 16));
      if (ea) {
        float m = 0.f;
        (void)hipEventRecord(eb, nullptr);
        if (hipEventSynchronize(eb) == hipSuccess && hipEventElapsedTime(&m, ea, eb) == hipSuccess) best_ms = std::min(best_ms, m);
      }
    }
    cand.push_back(c);
    ms.push_back(best_ms);
    if (best < 0 || best_ms < ms[(size_t)best]) best = t;
    slowest = std::max(slowest, best_ms);
    // stop once both modes have been seen (or the first candidate is plainly a fast one)
    if ((double)need / (best_ms * 1e-3) >= 6.1e12 || ms[(size_t)best] < 0.92f * slowest) break;
  }
  if (ea) (void)hipEventDestroy(ea);
  if (eb) (void)hipEventDestroy(eb);
  HIPCHK(hipDeviceSynchronize());
  HIPCHK(hipGetLastError());
  for (size_t i = 0; i < cand.size(); ++i)
    if ((int)i == best) b = cand[i]; else cand[i].release();
  return TEMX_OK;
}

// How to cut (d-tiles x chunk range) into wave-sized work so that `slots` workgroup slots
// (CUs x resident workgroups) are evenly filled.  Smaller nsplit is preferred on near-ties
// (fewer partial slabs to write and re-read).
static Split choose_split(int64_t D, int64_t nchunk, int slots, int dpw = 4, int minchunk = 4) {
  Split s;
  s.dpw = dpw;
  s.ndt = (int)((D + 15) / 16);
  const int ndq = (s.ndt + dpw - 1) / dpw;    // a workgroup owns dpw consecutive d-tiles
  int64_t maxsplit = std::max<int64_t>(1, nchunk / minchunk);
  maxsplit = std::min<int64_t>(maxsplit, std::max<int64_t>(1, (int64_t)4 * slots / ndq + 1));
  maxsplit = std::min<int64_t>(maxsplit, 4096);
  double best = -1.0;
  int bestn = 1;
  for (int n = 1; n <= maxsplit; ++n) {
    const int64_t nwg = (int64_t)ndq * n;
    const int64_t rounds = (nwg + slots - 1) / slots;
    const double eff = (double)nwg / (double)(rounds * slots);
    if (eff > best * 1.02) {
      best = eff;
      bestn = n;
    }
  }
  s.nsplit = bestn;
  const int64_t nwg = (int64_t)ndq * s.nsplit;
  s.grid = (int)(((nwg + 7) / 8) * 8);
  return s;
}


// hipFuncSetAttribute is per device: remember, per kernel instantiation, which devices have the
// dynamic-LDS limit raised (a process may own plans on several GPUs).
static int lds_attr_once(std::atomic<uint64_t>& done, int device, const void* fn, int bytes) {
  const uint64_t bit = 1ull << (device & 63);
  if (done.load(std::memory_order_acquire) & bit) return TEMX_OK;
  HIPCHK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
  done.fetch_or(bit, std::memory_order_release);
  return TEMX_OK;
}


// 4x4 MFMA A-operand blocks of a row-major R x K matrix: blk[rb][t][k*4+i] = A[4rb+i][4t+k], zero padded
static int upload_blocks(DevBuf& dst, const double* A, int R, int K, int TB) {
  const int nrb = (R + 3) / 4;
  std::vector<double> blk((size_t)nrb * TB * 16, 0.0);
  for (int rb = 0; rb < nrb; ++rb)
    for (int t = 0; t < TB; ++t)
      for (int k = 0; k < 4; ++k)
        for (int i = 0; i < 4; ++i) {
          const int r = 4 * rb + i, col = 4 * t + k;
          if (r < R && col < K) blk[((size_t)rb * TB + t) * 16 + k * 4 + i] = A[(size_t)r * K + col];
        }
  return upload(dst, blk.data(), blk.size() * 8);
}

static int os_upload_blocks(temx_plan* pl);

static int set_ginv(temx_plan* pl, const double* Gi_host) {
  HIPCHK(hipMemcpy(pl->Ginv.p, Gi_host, (size_t)pl->K * pl->K * 8, hipMemcpyHostToDevice));
  pl->h_Ginv.assign(Gi_host, Gi_host + (size_t)pl->K * pl->K);
  if (pl->os_built)
    if (int rc = os_upload_blocks(pl)) return rc;
  if (pl->K > 64) return TEMX_OK;
  // Ginv padded to 4*TB rows: upload_blocks wants TB row-blocks
  std::vector<double> Gp((size_t)4 * pl->TB * pl->K, 0.0);
  std::copy(Gi_host, Gi_host + (size_t)pl->K * pl->K, Gp.begin());
  return upload_blocks(pl->gblk, Gp.data(), 4 * pl->TB, pl->K, pl->TB);
}

constexpr size_t kMaxTimedLaunches = 256;   // enough for an average; bounds the events a long run creates

static void time_begin(temx_plan* pl, int which, hipStream_t st, TimedLaunch& tl) {
  tl.a = tl.b = nullptr;
  if (!pl->timing || pl->timed[which].size() >= kMaxTimedLaunches) return;
  (void)hipEventCreate(&tl.a);
  (void)hipEventCreate(&tl.b);
  (void)hipEventRecord(tl.a, st);
}
static void time_end(temx_plan* pl, int which, hipStream_t st, TimedLaunch& tl) {
  if (!pl->timing || tl.a == nullptr) return;
  (void)hipEventRecord(tl.b, st);
  pl->timed[which].push_back(tl);
}

// ------------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------------
// sweep-1 configurations (fields per wave, X ring depth, waves per SIMD).  Measured on
// ne120x72x30: 2 fields/wave with 2- or 3-deep rings at 3 waves/SIMD were within +-3 % of this.
struct ProjCfgC { static constexpr int NFW = 4, PD = 1, WPS = 2; };   // default: quads of d-tiles
struct ProjCfgE { static constexpr int NFW = 1, PD = 2, WPS = 3; };   // small ragged D: one d-tile per workgroup
struct ProjCfg1 { static constexpr int NFW = 1, PD = 2, WPS = 2; };   // single-field operator API / Gram

// d-tiles per workgroup that waste the fewest wave slots on a ragged last workgroup
static int pick_dpw(int ndt, int maxdpw) {
  int best = maxdpw;
  double bw = 1e9;
  for (int d = maxdpw; d >= 1; d >>= 1) {
    const double w = (double)(((ndt + d - 1) / d) * d) / ndt;
    if (w < bw - 0.05) {
      bw = w;
      best = d;
    }
  }
  return best;
}
static int proj_dpw(int NF, int ndt) {
  if (NF == 1) return 4;
  return pick_dpw(ndt, 4) == 1 ? 1 : 4;     // small ragged D: config E (1 field per wave)
}
static int proj_wps(int NF, int dpw) {
  if (NF == 1) return ProjCfg1::WPS;
  return dpw == 1 ? ProjCfgE::WPS : ProjCfgC::WPS;
}

template <typename T, int NF, typename Cfg>
static int launch_project_c(temx_plan* pl, const FieldPtrs<NF>& fp, int64_t D, const double* colscale,
                            int sfield, double* partial, const Split& sp, hipStream_t st, int tb_off,
                            int Kloc) {
  dim3 grid(sp.grid), block(256);
#define TEMX_LP(TBv)                                                                                  \
  hipLaunchKernelGGL((project_kernel<T, NF, Cfg::NFW, TBv, Cfg::PD, Cfg::WPS>), grid, block, 0, st, fp, \
                     pl->N, D, Kloc, pl->yproj_ptr(), pl->stride, tb_off, pl->nchunk, colscale,       \
                     sfield, partial, sp.nsplit, sp.ndt)
  switch (pl->TB) {
    case 4: TEMX_LP(4); break;
    case 8: TEMX_LP(8); break;
    case 13: TEMX_LP(13); break;
    default: TEMX_LP(16); break;
  }
#undef TEMX_LP
  HIPCHK(hipGetLastError());
  return TEMX_OK;
}

constexpr int SYM_PROJ_E_WPS = 3;
struct ProjCfg3 { static constexpr int NFW = 3, PD = 1, WPS = 2; };   // the 3 eddy products (large-L path)

template <typename T, int NF>
static int launch_project_t(temx_plan* pl, const FieldPtrs<NF>& fp, int64_t D, const double* colscale,
                            int sfield, double* partial, const Split& sp, hipStream_t st, int tb_off,
                            int Kloc) {
  if constexpr (NF == 1) {
    return launch_project_c<T, NF, ProjCfg1>(pl, fp, D, colscale, sfield, partial, sp, st, tb_off, Kloc);
  } else if constexpr (NF == 3) {
    return launch_project_c<T, NF, ProjCfg3>(pl, fp, D, colscale, sfield, partial, sp, st, tb_off, Kloc);
  } else {
    if (sp.dpw == 1)
      return launch_project_c<T, NF, ProjCfgE>(pl, fp, D, colscale, sfield, partial, sp, st, tb_off, Kloc);
    return launch_project_c<T, NF, ProjCfgC>(pl, fp, D, colscale, sfield, partial, sp, st, tb_off, Kloc);
  }
}

template <int NF>
static int launch_project(temx_plan* pl, const FieldPtrs<NF>& fp, int dtype, int64_t D,
                          const double* colscale, int sfield, double* partial, const Split& sp,
                          hipStream_t st, int tb_off = 0, int Kloc = -1) {
  if (Kloc < 0) Kloc = pl->K;
  if (dtype == TEMX_F64)
    return launch_project_t<double, NF>(pl, fp, D, colscale, sfield, partial, sp, st, tb_off, Kloc);
  if (dtype == TEMX_F32)
    return launch_project_t<float, NF>(pl, fp, D, colscale, sfield, partial, sp, st, tb_off, Kloc);
  return fail(TEMX_EINVAL, "dtype must be TEMX_F64 or TEMX_F32");
}

// B[n] = (addend ? addend : 0) + sum over the nsplit slabs; slab sp starts at partial + sp * stride
// (stride < 0: the slabs are dense, stride = n)
// map: where entry idx of the sum goes (default: B[idx]; time slices for a reduce-scatter: kernels.hpp, SliceMap)
static int launch_reduce(temx_plan* pl, const double* partial, int nsplit, int64_t n, double* B,
                         hipStream_t st, int64_t stride = -1, const double* addend = nullptr,
                         const SliceMap& map = SliceMap()) {
  if (stride < 0) stride = n;
  if (nsplit <= 16 && n >= 32768) {   // many entries, few slabs: one thread per entry (same bits)
    const dim3 grid((unsigned)((n + 255) / 256));
    int* flag = static_cast<int*>(pl->flag.p);
    if (nsplit <= 2)
      hipLaunchKernelGGL(reduce_partials_flat_kernel<2>, grid, dim3(256), 0, st, partial, nsplit, stride, n, addend, B, flag, map);
    else if (nsplit <= 4)
      hipLaunchKernelGGL(reduce_partials_flat_kernel<4>, grid, dim3(256), 0, st, partial, nsplit, stride, n, addend, B, flag, map);
    else if (nsplit <= 8)
      hipLaunchKernelGGL(reduce_partials_flat_kernel<8>, grid, dim3(256), 0, st, partial, nsplit, stride, n, addend, B, flag, map);
    else
      hipLaunchKernelGGL(reduce_partials_flat_kernel<16>, grid, dim3(256), 0, st, partial, nsplit, stride, n, addend, B, flag, map);
    HIPCHK(hipGetLastError());
    return TEMX_OK;
  }
  const int64_t blocks = (n + 15) / 16;
  hipLaunchKernelGGL(reduce_partials_kernel, dim3((unsigned)blocks), dim3(256), 0, st, partial, nsplit, stride, n,
                     addend, B, static_cast<int*>(pl->flag.p), map);
  HIPCHK(hipGetLastError());
  return TEMX_OK;
}

// out[f][r][d] = sum over the slabs of the first `rows` of each field's `rows_in` rows (kernels.hpp)
static int launch_reduce_rows(temx_plan* pl, const double* partial, int nsplit, int64_t stride, int nf, int rows_in,
                              int rows, int64_t D, double* out, hipStream_t st) {
  const int64_t n = (int64_t)nf * rows * D;
  if (nsplit > 16)
    hipLaunchKernelGGL(reduce_rows_wide_kernel, dim3((unsigned)((n + 15) / 16)), dim3(256), 0, st, partial, nsplit, stride,
                       nf, rows_in, rows, D, out, static_cast<int*>(pl->flag.p));
  else
    hipLaunchKernelGGL(reduce_rows_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, partial, nsplit, stride,
                       nf, rows_in, rows, D, out, static_cast<int*>(pl->flag.p));
  HIPCHK(hipGetLastError());
  return TEMX_OK;
}

static int launch_solve(temx_plan* pl, const double* B, int NF, int64_t D, double* C, double* Xb,
                        hipStream_t st) {
  if (pl->K <= 64) {   // two small MFMA GEMMs per d-tile
    const int nmb = (pl->M + 3) / 4;
    // slices of the output latitudes: every slice recomputes the coefficients (TB^2 MFMAs), so slices
    // are as tall as the LDS image allows (SOLVE_MB) unless the grid would leave most of the chip idle
    const int gx = (int)(((D + 15) / 16 + 3) / 4);
    int mbs = SOLVE_MB;
    while (mbs > 4 && (int64_t)gx * NF * ((nmb + mbs - 1) / mbs) < pl->num_cu / 2) mbs -= 4;
    dim3 grid((unsigned)gx, NF, Xb ? (nmb + mbs - 1) / mbs : 1);
#define TEMX_LS(TBv)                                                                                  \
  do {                                                                                                \
    const size_t slds = ((size_t)TBv * TBv + (size_t)SOLVE_MB * TBv) * 16 * sizeof(double);           \
    static std::atomic<uint64_t> attr{0};                                                             \
    if (int rc_ = lds_attr_once(attr, pl->device, reinterpret_cast<const void*>(solve_mfma_kernel<TBv>), (int)slds)) \
      return rc_;                                                                                     \
    hipLaunchKernelGGL(solve_mfma_kernel<TBv>, grid, dim3(256), slds, st, B, pl->K, pl->M, D,         \
                       pl->gblk.d(), pl->ypblk.d(), C, Xb, mbs);                                      \
  } while (0)
    switch (pl->TB) {
      case 4: TEMX_LS(4); break;
      case 8: TEMX_LS(8); break;
      case 13: TEMX_LS(13); break;
      default: TEMX_LS(16); break;
    }
#undef TEMX_LS
    HIPCHK(hipGetLastError());
    return TEMX_OK;
  }
  // 16 waves per block (the solve is latency bound: one round of dot products per phase); slices of
  // 64 output latitudes = one zonal-mean output per thread
  const int ms = Xb ? (pl->M + 63) / 64 : 1;
  dim3 grid((unsigned)((D + 15) / 16), NF, ms);
  {
    const size_t slds = (size_t)2 * pl->K4 * 17 * sizeof(double);
    static std::atomic<uint64_t> attr{0};
    if (int rc = lds_attr_once(attr, pl->device, reinterpret_cast<const void*>(solve_kernel), 160 * 1024)) return rc;
    hipLaunchKernelGGL(solve_kernel, grid, dim3(1024), slds, st, B, pl->K, pl->K4, pl->M, D, pl->Ginv.d(),
                       pl->Qp.d(), C, Xb);
  }
  HIPCHK(hipGetLastError());
  return TEMX_OK;
}

template <typename T, int MODE, int DPW, int KIND>
static int launch_eddy_d(temx_plan* pl, const FieldPtrs<4>& fp, const double* C, double* partial,
                         const Split& sp, const EddyOut& eo, hipStream_t st) {
  dim3 grid(sp.grid), block(512);
  constexpr int NFR = KIND == 0 ? 4 : 3;
#define TEMX_LE(TBv)                                                                                  \
  do {                                                                                                \
    auto kern = eddy_kernel<T, TBv, MODE, DPW, KIND>;                                                 \
    const size_t lds = ((size_t)DPW * NFR * TBv * 64 + 8 * EDDY_GR * TBv * 16) * sizeof(double);      \
    static std::atomic<uint64_t> attr_set{0};   /* per instantiation, one bit per device */           \
    if (int rc_ = lds_attr_once(attr_set, pl->device, reinterpret_cast<const void*>(kern), (int)lds)) \
      return rc_;                                                                                     \
    hipLaunchKernelGGL(kern, grid, block, lds, st, fp, pl->N, pl->D, pl->K, pl->yblk.d(),             \
                       pl->nchunk, pl->colscale.d(), C, partial, sp.nsplit, sp.ndt, eo);              \
  } while (0)
  switch (pl->TB) {
    case 4: TEMX_LE(4); break;
    case 8: TEMX_LE(8); break;
    case 13: TEMX_LE(13); break;
    default: TEMX_LE(16); break;
  }
#undef TEMX_LE
  HIPCHK(hipGetLastError());
  return TEMX_OK;
}

template <typename T, int MODE, int KIND>
static int launch_eddy_t(temx_plan* pl, const FieldPtrs<4>& fp, const double* C, double* partial,
                         const Split& sp, const EddyOut& eo, hipStream_t st) {
  switch (sp.dpw) {
    case 1: return launch_eddy_d<T, MODE, 1, KIND>(pl, fp, C, partial, sp, eo, st);
    case 2: return launch_eddy_d<T, MODE, 2, KIND>(pl, fp, C, partial, sp, eo, st);
    default: return launch_eddy_d<T, MODE, 4, KIND>(pl, fp, C, partial, sp, eo, st);
  }
}

// native-grid reconstruction out[row][d] = sum_l Y0[row0 + row][l] C[l][d] for rows [row0, row0 + nrows),
// row0 a multiple of 16 (whole chunks of the blocked Y0 copy); out is compact [nrows][D]
static int launch_recon(temx_plan* pl, int64_t D, const double* C, double* out, hipStream_t st,
                        int64_t row0 = 0, int64_t nrows = -1) {
  if (nrows < 0) nrows = pl->N - row0;
  const int64_t nch = (nrows + 15) / 16;
  const double* yb0 = pl->yblk.d() + (row0 / 4) * pl->stride * 16;
  Split sp = choose_split(D, nch, 2 * pl->num_cu);
  dim3 grid(sp.grid), block(256);
#define TEMX_LR(TBv, tboff, Cs, acc)                                                                  \
  do {                                                                                                \
    auto kern = recon_kernel<TBv>;                                                                    \
    const size_t lds = (size_t)4 * TBv * 64 * sizeof(double);                                         \
    hipLaunchKernelGGL(kern, grid, block, lds, st, nrows, D, yb0, pl->stride, tboff,                  \
                       nch, Cs, out, acc, sp.nsplit, sp.ndt);                                         \
  } while (0)
  if (pl->large) {   // one pass per slice of 64 harmonics, accumulating into out
    for (int sl = 0; sl < pl->nslice; ++sl) TEMX_LR(16, 16 * sl, C + (int64_t)64 * sl * D, sl > 0 ? 1 : 0);
  } else {
    switch (pl->TB) {
      case 4: TEMX_LR(4, 0, C, 0); break;
      case 8: TEMX_LR(8, 0, C, 0); break;
      case 13: TEMX_LR(13, 0, C, 0); break;
      default: TEMX_LR(16, 0, C, 0); break;
    }
  }
#undef TEMX_LR
  HIPCHK(hipGetLastError());
  return TEMX_OK;
}

// B[NF][K][D] = Y0^T {fields}: one projection sweep, or one per slice of 64 harmonics (large L)
template <int NF>
static int project_all(temx_plan* pl, const FieldPtrs<NF>& fp, int dtype, int64_t D, const double* colscale,
                       int sfield, const Split& sp, double* B, hipStream_t st) {
  int rc;
  if (!pl->large) {
    if ((rc = launch_project<NF>(pl, fp, dtype, D, colscale, sfield, pl->partial.d(), sp, st))) return rc;
    return launch_reduce(pl, pl->partial.d(), sp.nsplit, (int64_t)NF * pl->K * D, B, st);
  }
  if ((rc = pl->Bs.ensure((size_t)NF * 64 * D * 8))) return rc;
  for (int sl = 0; sl < pl->nslice; ++sl) {
    const int Ks = std::min(64, pl->K - 64 * sl);
    if ((rc = launch_project<NF>(pl, fp, dtype, D, colscale, sfield, pl->partial.d(), sp, st, 16 * sl, Ks))) return rc;
    if ((rc = launch_reduce(pl, pl->partial.d(), sp.nsplit, (int64_t)NF * Ks * D, pl->Bs.d(), st))) return rc;
    for (int f = 0; f < NF; ++f)
      HIPCHK(hipMemcpyAsync(B + ((int64_t)f * pl->K + 64 * sl) * D, pl->Bs.d() + (int64_t)f * Ks * D,
                            (size_t)Ks * D * 8, hipMemcpyDeviceToDevice, st));
  }
  return TEMX_OK;
}


// ------------------------------------------------------------------------------------------------
// mirror pairing of an equatorially symmetric grid (kernels_sym.hpp)
// ------------------------------------------------------------------------------------------------
static double sym_tol_deg(double dflt);

// Every column with lat > tol must have a partner with the opposite latitude (any longitude);
// |lat| <= tol are equator columns (pairs without a southern partner).  Returns false if the grid is
// not symmetric.  Pairs are ordered by their northern row so one operand still streams.
static bool find_mirror_pairs(const double* lat, int64_t N, std::vector<int>& rowN, std::vector<int>& rowS, double tol_dflt) {
  if (N >= ((int64_t)1 << 31)) return false;
  // Two columns pair up when their latitudes are opposite to within `tol` degrees; the pair is then
  // treated as sitting exactly at +-(northern latitude), which perturbs the operator by
  // O(L^2 tol).  The default keeps that below the fp64 parity tolerance; TEMX_SYM_TOL_DEG widens
  // it for grids whose files carry noisier latitudes.
  const double tol = sym_tol_deg(tol_dflt);
  std::vector<int> north, south, eq;
  for (int64_t i = 0; i < N; ++i) {
    if (!(std::fabs(lat[i]) <= 90.0 + 1e-9)) return false;
    if (lat[i] > tol) north.push_back((int)i);
    else if (lat[i] < -tol) south.push_back((int)i);
    else eq.push_back((int)i);
  }
  if (north.size() != south.size()) return false;
  std::stable_sort(north.begin(), north.end(), [&](int a, int b) { return lat[a] < lat[b]; });
  std::stable_sort(south.begin(), south.end(), [&](int a, int b) { return -lat[a] < -lat[b]; });
  std::vector<std::pair<int, int>> pairs;
  pairs.reserve(north.size() + eq.size());
  for (size_t k = 0; k < north.size(); ++k) {
    if (std::fabs(lat[north[k]] + lat[south[k]]) > tol) return false;
    pairs.emplace_back(north[k], south[k]);
  }
  for (int e : eq) pairs.emplace_back(e, -1);
  std::sort(pairs.begin(), pairs.end());
  rowN.resize(pairs.size());
  rowS.resize(pairs.size());
  for (size_t k = 0; k < pairs.size(); ++k) {
    rowN[k] = pairs[k].first;
    rowS[k] = pairs[k].second;
  }
  return true;
}


static double sym_tol_deg(double dflt) {
  // Two columns share a latitude class (or pair up) when their |lat| agree to within `tol` degrees; the members are
  // then treated as sitting exactly at the class latitude (the mean of its members), which perturbs a basis row by
  // about L tol (in radians) of its size.  The default, 1e-11 degrees, keeps that below 1e-11 -- a tenth of the fp64
  // parity tolerance -- and covers what asin / round-off leave on grids whose latitudes are equal in exact
  // arithmetic (the natural construction of the cubed sphere is off by ~1e-12 degrees at ne240; round 3 used 1e-12
  // and a grid generator that mirrored the hemispheres bit for bit).  TEMX_LAT_TOL_F32 (fp32 fields: results are
  // compared to 2e-5) widens it to 1e-8 degrees; TEMX_SYM_TOL_DEG in the environment sets it outright, for grids
  // whose files carry noisier latitudes.
  double tol = dflt;
  if (const char* e = getenv("TEMX_SYM_TOL_DEG")) {
    const double t = atof(e);
    if (t > 0.0 && t < 1e-3) tol = t;
  }
  return tol;
}

// Latitude classes (kernels_cls.hpp).  Returns false when the grid has too few columns per class
// for the class sweeps to pay (the paired or generic sweeps are used instead).
struct ClassTables {
  std::vector<int> crow;        // [nbatch + 2][4][CLS_MB]
  std::vector<double> xc;       // [4 * (ngroups + 1)] cos(colat) of the class latitude
  std::vector<int> gbatch0;     // [ngroups + 1]
  std::vector<double> cnt;      // [ngroups][2 sides][4 classes] member counts
  int64_t ncls = 0, ngroups = 0, nbatch = 0;
};

static bool build_classes(const double* lat, int64_t N, ClassTables& ct, double tol_dflt) {
  if (N >= ((int64_t)1 << 27) || N < 64) return false;     // row indices live in 27 bits of a table entry
  const double tol = sym_tol_deg(tol_dflt);
  std::vector<int> order((size_t)N);
  for (int64_t i = 0; i < N; ++i) {
    if (!(std::fabs(lat[i]) <= 90.0 + 1e-9)) return false;
    order[(size_t)i] = (int)i;
  }
  std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return std::fabs(lat[a]) < std::fabs(lat[b]); });
  struct Cls {
    double alat;
    std::vector<int> n, s;
  };
  std::vector<Cls> cls;
  for (size_t i = 0; i < order.size();) {
    Cls c;
    c.alat = std::fabs(lat[order[i]]);
    size_t j = i;
    for (; j < order.size() && std::fabs(lat[order[j]]) - c.alat <= tol; ++j) {
      const int r = order[j];
      (lat[r] < -tol ? c.s : c.n).push_back(r);      // equator columns count as northern
    }
    // the class sits at the mean |lat| of its members (they agree to within tol): deviations of either
    // sign, half the size of those from the smallest member
    long double sum = 0.0L;
    for (size_t m = i; m < j; ++m) sum += (long double)std::fabs(lat[order[m]]) - (long double)c.alat;
    c.alat += (double)(sum / (long double)(j - i));
    std::sort(c.n.begin(), c.n.end());
    std::sort(c.s.begin(), c.s.end());
    cls.push_back(std::move(c));
    i = j;
  }
  if ((double)N < 3.0 * (double)cls.size()) return false;   // < 3 columns per class: not worth it
  constexpr int MB = CLS_MB;
  auto nb = [](size_t m) { return (int)((m + MB - 1) / MB); };
  // Outsized classes are cut into several classes at the same latitude.  Any subset of the columns of a
  // latitude is a class (the algebra of kernels_cls.hpp holds per class side), and a class-group is the unit of
  // work between two hand-overs of the shared-d-tile sweep and of the work cuts: the cubed sphere has ONE
  // class of 1440 equator columns (360 batches) among 48 000 of 8 + 8 (4 batches), and every workgroup that
  // met it ran 20 % longer -- all of them on one XCD (ne120 x 72 x 30, TEM + tracer sweep: 17.6 -> 14 ms).
  {
    std::map<std::pair<size_t, size_t>, size_t> hist;
    for (const Cls& c : cls) ++hist[{c.n.size(), c.s.size()}];
    std::pair<size_t, size_t> typ{0, 0};
    size_t best = 0;
    for (const auto& kv : hist)
      if (kv.second > best) {
        best = kv.second;
        typ = kv.first;
      }
    const size_t cap_n = std::max<size_t>(typ.first, MB), cap_s = std::max<size_t>(typ.second, MB);
    const int typ_b = std::max(1, nb(typ.first) + nb(typ.second));
    std::vector<Cls> out;
    out.reserve(cls.size());
    for (Cls& c : cls) {
      if (nb(c.n.size()) + nb(c.s.size()) <= 4 * typ_b) {
        out.push_back(std::move(c));
        continue;
      }
      const size_t parts = std::max((c.n.size() + cap_n - 1) / cap_n, (c.s.size() + cap_s - 1) / cap_s);
      for (size_t k = 0; k < parts; ++k) {
        Cls d;
        d.alat = c.alat;
        for (size_t m = k * cap_n; m < std::min((k + 1) * cap_n, c.n.size()); ++m) d.n.push_back(c.n[m]);
        for (size_t m = k * cap_s; m < std::min((k + 1) * cap_s, c.s.size()); ++m) d.s.push_back(c.s[m]);
        if (!d.n.empty() || !d.s.empty()) out.push_back(std::move(d));
      }
    }
    cls.swap(out);
  }
  // equal member counts inside a class-group; then by first row (some streaming order)
  std::stable_sort(cls.begin(), cls.end(), [&](const Cls& a, const Cls& b) {
    const int an = nb(a.n.size()), as = nb(a.s.size()), bn = nb(b.n.size()), bs = nb(b.s.size());
    if (an != bn) return an > bn;
    if (as != bs) return as > bs;
    const int ar = a.n.empty() ? a.s[0] : a.n[0], br = b.n.empty() ? b.s[0] : b.n[0];
    return ar < br;
  });
  // The class-groups (4 consecutive classes) of the few small strata -- ne240: 361 groups of 4 + 4 members and 90 of
  // 8 + 0 behind 48 374 of 8 + 8 -- are spread evenly among the others: a work cut is a run of consecutive groups
  // balanced by batch count, and the end of a group costs about as much as two batches, so the workgroup that got
  // the tail of a size-sorted table ran 30 % longer than the rest (ne240 x 128 x 1: 1.9 instead of 1.5 ms;
  // profiles/r03_lab20_class_order_d128_f32.log).  Inside a stratum the order stays.  A last, partial group stays last.
  {
    const size_t ngr = (cls.size() + 3) / 4;
    std::vector<std::pair<int, int>> shape(ngr);
    std::map<std::pair<int, int>, size_t> count, seen;
    for (size_t gi = 0; gi < ngr; ++gi) {
      int bN = 0, bS = 0;
      for (size_t ci = gi * 4; ci < std::min(gi * 4 + 4, cls.size()); ++ci) {
        bN = std::max(bN, nb(cls[ci].n.size()));
        bS = std::max(bS, nb(cls[ci].s.size()));
      }
      shape[gi] = {bN, bS};
      ++count[shape[gi]];
    }
    std::vector<std::pair<double, size_t>> key(ngr);
    for (size_t gi = 0; gi < ngr; ++gi) {
      const size_t i = seen[shape[gi]]++;
      key[gi] = {((double)i + 0.5) / (double)count[shape[gi]], gi};
      if (gi + 1 == ngr && cls.size() % 4 != 0) key[gi].first = 2.0;
    }
    std::stable_sort(key.begin(), key.end(), [](const std::pair<double, size_t>& a, const std::pair<double, size_t>& b) { return a.first < b.first; });
    std::vector<Cls> out;
    out.reserve(cls.size());
    for (const auto& kv : key)
      for (size_t ci = kv.second * 4; ci < std::min(kv.second * 4 + 4, cls.size()); ++ci) out.push_back(std::move(cls[ci]));
    cls.swap(out);
  }
  ct.ncls = (int64_t)cls.size();
  ct.ngroups = (ct.ncls + 3) / 4;
  ct.xc.assign((size_t)(ct.ngroups + 1) * 4, 0.0);
  ct.gbatch0.assign((size_t)ct.ngroups + 1, 0);
  ct.cnt.assign((size_t)ct.ngroups * 8, 0.0);
  ct.crow.clear();
  const double d2r = M_PI / 180.0;
  for (int64_t gi = 0; gi < ct.ngroups; ++gi) {
    int bN = 0, bS = 0;
    for (int k = 0; k < 4; ++k) {
      const int64_t ci = gi * 4 + k;
      if (ci >= ct.ncls) continue;
      bN = std::max(bN, nb(cls[(size_t)ci].n.size()));
      bS = std::max(bS, nb(cls[(size_t)ci].s.size()));
      ct.xc[(size_t)ci] = std::cos((90.0 - cls[(size_t)ci].alat) * d2r);
      ct.cnt[(size_t)gi * 8 + k] = (double)cls[(size_t)ci].n.size();
      ct.cnt[(size_t)gi * 8 + 4 + k] = (double)cls[(size_t)ci].s.size();
    }
    ct.gbatch0[(size_t)gi] = (int)(ct.crow.size() / (4 * MB));
    for (int side = 0; side < 2; ++side) {
      const int nbat = side ? bS : bN;
      for (int bi = 0; bi < nbat; ++bi) {
        int flags = side ? CLS_SOUTH : 0;
        if (bi == 0 && (side == 0 || bN == 0)) flags |= CLS_FIRST;
        if (bi == nbat - 1 && (side == 1 || bS == 0)) flags |= CLS_LAST;
        int batch[4 * MB];
        bool haspad = false;
        for (int k = 0; k < 4; ++k) {
          const int64_t ci = gi * 4 + k;
          for (int j = 0; j < MB; ++j) {
            const size_t m = (size_t)bi * MB + j;
            int ent = (int)0x80000000 | (flags << 28);
            if (ci < ct.ncls) {
              const std::vector<int>& mem = side ? cls[(size_t)ci].s : cls[(size_t)ci].n;
              if (m < mem.size()) ent = mem[m] | (flags << 28);
            }
            haspad = haspad || ent < 0;
            batch[k * MB + j] = ent;
          }
        }
        for (int e = 0; e < 4 * MB; ++e) ct.crow.push_back(batch[e] | (haspad ? CLS_HASPAD_BIT : 0));
      }
    }
  }
  ct.nbatch = (int64_t)(ct.crow.size() / (4 * MB));
  ct.gbatch0[(size_t)ct.ngroups] = (int)ct.nbatch;
  ct.crow.resize(ct.crow.size() + (size_t)CLS_PADB * 4 * MB, (int)0x80000000);   // index loads run ahead
  return true;
}

// (first batch, its group) of `nsub` pieces of the batch list: equal batch counts; a cut may fall
// inside a class-group (the sweeps are linear in the member rows, the kernels project partial sums)
#ifndef TEMX_GROUP_COST
#define TEMX_GROUP_COST 2
#endif
static int class_cuts(temx_plan* pl, int nsub, const int2** out, bool group_aligned = false) {
  const int key = group_aligned ? -nsub : nsub;
  auto it = pl->csplits.find(key);
  if (it == pl->csplits.end()) {
    std::vector<int> cut((size_t)2 * (nsub + 1));
    int g = 0;
    // group-aligned cuts balance batches + TEMX_GROUP_COST per class-group: the end of a group (exchange, reference,
    // 100-160 MFMAs) costs about two batches, and the table ends with the small classes (ne240: 361 groups of 2
    // batches), so that cuts by batch count alone gave the last workgroup twice the groups -- 1.9 instead of 1.5 ms
    // for ne240 x 128 x 1 (profiles/r03_lab20_class_order_d128_f32.log)
    const int64_t total = pl->cbatches + (int64_t)TEMX_GROUP_COST * pl->cgroups;
    for (int k = 0; k <= nsub; ++k) {
      const int64_t b = pl->cbatches * k / nsub;
      if (group_aligned) {     // the one-pass sweep stores whole-class sums: cut at the next group boundary
        const int64_t want = total * k / nsub;
        while (g < pl->cgroups && pl->gbatch0[(size_t)g] + (int64_t)TEMX_GROUP_COST * g < want) ++g;
        if (k == nsub) g = (int)pl->cgroups;
        cut[(size_t)2 * k] = pl->gbatch0[(size_t)g];
        cut[(size_t)2 * k + 1] = g;
        continue;
      }
      while (g + 1 < pl->cgroups && pl->gbatch0[(size_t)g + 1] <= b) ++g;
      cut[(size_t)2 * k] = (int)b;
      cut[(size_t)2 * k + 1] = g;
    }
    DevBuf b;
    int rc = upload(b, cut.data(), cut.size() * sizeof(int));
    if (rc) return rc;
    it = pl->csplits.emplace(key, b).first;
  }
  *out = static_cast<const int2*>(it->second.p);
  return TEMX_OK;
}

#ifndef TEMX_CLS_E_WPS
#define TEMX_CLS_E_WPS 3
#endif
#ifndef TEMX_CLS_E_PD
#define TEMX_CLS_E_PD 2
#endif
#ifndef TEMX_CLS_OP_WPS
#define TEMX_CLS_OP_WPS 1
#endif
#ifndef TEMX_CLS_OP_PD
#define TEMX_CLS_OP_PD 3
#endif
#ifndef TEMX_CLS_MINCHUNK
#define TEMX_CLS_MINCHUNK 1
#endif
#ifndef TEMX_CLS_OP_PD_F32
#define TEMX_CLS_OP_PD_F32 6
#endif
#ifndef TEMX_CLS_E_PD_F32
#define TEMX_CLS_E_PD_F32 4
#endif
#ifndef TEMX_CLS_Q_PD_F32
#define TEMX_CLS_Q_PD_F32 2
#endif
constexpr int CLS_PROJ_E_WPS = TEMX_CLS_E_WPS, CLS_PROJ_E_PD = TEMX_CLS_E_PD;   // one field per wave
// X batches a wave of the class project sweep holds (PD - 1 in flight).  A batch of fp32 carries half
// the bytes of an fp64 one in half the registers, so fp32 inputs get rings twice as deep: the same
// bytes in flight per wave.
template <typename T>
constexpr int cls_proj_pd(bool op, int nfw) {
  constexpr bool f32 = sizeof(T) == 4;
  return op ? (f32 ? TEMX_CLS_OP_PD_F32 : TEMX_CLS_OP_PD)
            : (nfw == 1 ? (f32 ? TEMX_CLS_E_PD_F32 : CLS_PROJ_E_PD) : (f32 ? TEMX_CLS_Q_PD_F32 : 2));
}

template <typename T, int NF>
static int launch_project_cls_t(temx_plan* pl, const FieldPtrs<NF>& fp, int64_t D, const double* colscale,
                                int sfield, double* partial, const Split& sp, hipStream_t st) {
  const int2* cuts = nullptr;
  if (int rc = class_cuts(pl, sp.nsplit, &cuts)) return rc;
  dim3 grid(sp.grid), block(256);
#define TEMX_LPC(TBSv, NFWv, WPSv, OPv)                                                             \
  hipLaunchKernelGGL((project_cls_kernel<T, NF, NFWv, TBSv, WPSv, cls_proj_pd<T>(OPv, NFWv), OPv>), grid, block, 0, st, fp, D, pl->K, \
                     pl->ycls.d(), static_cast<const int4*>(pl->crow.p), cuts, colscale, sfield,    \
                     partial, sp.nsplit, sp.ndt, (double*)nullptr)
  if (NF == 1 || sp.dpw == 1) {     // one field per wave (NF = 4: small ragged D, one d-tile per workgroup)
    switch (pl->TBS) {
      case 2: TEMX_LPC(2, 1, CLS_PROJ_E_WPS, false); break;
      case 4: TEMX_LPC(4, 1, CLS_PROJ_E_WPS, false); break;
      case 7: TEMX_LPC(7, 1, CLS_PROJ_E_WPS, false); break;
      default: TEMX_LPC(8, 1, CLS_PROJ_E_WPS, false); break;
    }
  } else {
    switch (pl->TBS) {
      case 2: TEMX_LPC(2, NF, 2, false); break;
      case 4: TEMX_LPC(4, NF, 2, false); break;
      case 7: TEMX_LPC(7, NF, 2, false); break;
      default: TEMX_LPC(8, NF, 2, false); break;
    }
  }
#undef TEMX_LPC
  HIPCHK(hipGetLastError());
  return TEMX_OK;
}

template <int NF>
static int launch_project_cls(temx_plan* pl, const FieldPtrs<NF>& fp, int dtype, int64_t D,
                              const double* colscale, int sfield, double* partial, const Split& sp,
                              hipStream_t st) {
  if (dtype == TEMX_F64) return launch_project_cls_t<double, NF>(pl, fp, D, colscale, sfield, partial, sp, st);
  if (dtype == TEMX_F32) return launch_project_cls_t<float, NF>(pl, fp, D, colscale, sfield, partial, sp, st);
  return fail(TEMX_EINVAL, "dtype must be TEMX_F64 or TEMX_F32");
}

template <typename T, int KIND>
static int launch_sweep_op_t(temx_plan* pl, const FieldPtrs<4>& fp, double* partial, const Split& sp, double* sums,
                             hipStream_t st) {
  const int2* cuts = nullptr;
  if (int rc = class_cuts(pl, sp.nsplit, &cuts, true)) return rc;
  dim3 grid(sp.grid), block(256);
  // the tracer sweep (42 accumulators) runs two waves per SIMD: a ring of 2 batches keeps it inside 256 registers
  constexpr int PDv = KIND == 1 ? (sizeof(T) == 4 ? 4 : 2) : (sizeof(T) == 4 ? TEMX_CLS_OP_PD_F32 : TEMX_CLS_OP_PD);
  // loads of 1 row x 64 columns (sweep_opr_kernel, kernels_op2.hpp) unless the last workgroup column would be
  // mostly padding (D = 72: 64 + 8) or TEMX_OP_MAP=tile asks for the tile form (A/B)
  const int64_t wcols = (pl->D + 63) / 64 * 64;
  // fp64 only: with fp32 inputs the tile form measured faster (ne240 x 128 x 1: 1.77 vs 2.08 ms, ne120 x 72 x 30: 7.2 vs 7.5)
  const bool row_map = sizeof(T) == 8 && !pl->op_tile && wcols * 100 <= pl->D * 115;
  constexpr int PDr = 2;
#define TEMX_LSO(TBSv)                                                                                \
  do {                                                                                                \
    if (row_map)                                                                                      \
      hipLaunchKernelGGL((sweep_opr_kernel<double, TBSv, PDr, KIND>), grid, block, 0, st, fp, pl->D, pl->K, pl->ycls.d(), \
                         static_cast<const int4*>(pl->crow.p), cuts, pl->colscale.d(), partial, sp.nsplit,  \
                         sp.ndt, sums);                                                               \
    else                                                                                              \
      hipLaunchKernelGGL((sweep_op_kernel<T, TBSv, PDv, KIND>), grid, block, 0, st, fp, pl->D, pl->K, pl->ycls.d(), \
                         static_cast<const int4*>(pl->crow.p), cuts, pl->colscale.d(), partial, sp.nsplit,   \
                         sp.ndt, sums);                                                               \
  } while (0)
  switch (pl->TBS) {
    case 2: TEMX_LSO(2); break;
    case 4: TEMX_LSO(4); break;
    case 7: TEMX_LSO(7); break;
    default: TEMX_LSO(8); break;
  }
#undef TEMX_LSO
  HIPCHK(hipGetLastError());
  return TEMX_OK;
}

// TEM + one tracer in one sweep (kernels_op2.hpp): (u, v, T, omega, q) -> csum, csq, 10 slabs per split
template <typename T>
static int launch_sweep_opw2_t(temx_plan* pl, const FieldPtrs<5>& fp, double* partial, const Split& sp, hipStream_t st) {
  const int2* cuts = nullptr;
  if (int rc = class_cuts(pl, sp.nsplit * 4, &cuts, true)) return rc;
  dim3 grid(sp.grid), block(256);
  constexpr int PDv = sizeof(T) == 4 ? TEMX_CLS_OP_PD_F32 : TEMX_CLS_OP_PD;
#define TEMX_LSW(TBSv)                                                                                       \
  hipLaunchKernelGGL((sweep_opw_kernel<T, TBSv, PDv, 2>), grid, block, 0, st, fp, pl->D, pl->K, pl->ycls.d(), \
                     static_cast<const int4*>(pl->crow.p), cuts, pl->colscale.d(), partial, sp.nsplit, sp.ndt, \
                     pl->csum.d(), pl->csq.d())
  switch (pl->TBS) {
    case 2: TEMX_LSW(2); break;
    case 4: TEMX_LSW(4); break;
    case 7: TEMX_LSW(7); break;
    default: TEMX_LSW(8); break;
  }
#undef TEMX_LSW
  HIPCHK(hipGetLastError());
  return TEMX_OK;
}

// KIND 0: (u, v, T, omega) -> csum; KIND 1: (q, v, omega) -> csq
template <int KIND>
static int launch_sweep_op(temx_plan* pl, const FieldPtrs<4>& fp, int dtype, double* partial, const Split& sp,
                           hipStream_t st) {
  double* sums = KIND == 0 ? pl->csum.d() : pl->csq.d();
  if (dtype == TEMX_F64) return launch_sweep_op_t<double, KIND>(pl, fp, partial, sp, sums, st);
  if (dtype == TEMX_F32) return launch_sweep_op_t<float, KIND>(pl, fp, partial, sp, sums, st);
  return fail(TEMX_EINVAL, "dtype must be TEMX_F64 or TEMX_F32");
}

template <typename T, int MODE, int DPW, int KIND>
static int launch_eddy_cls_d(temx_plan* pl, const FieldPtrs<4>& fp, const double* C, double* partial,
                             const Split& sp, const EddyOut& eo, hipStream_t st) {
  const int2* cuts = nullptr;
  if (int rc = class_cuts(pl, sp.nsplit * (8 / DPW), &cuts)) return rc;
  dim3 grid(sp.grid), block(512);
  constexpr int NFR = KIND == 0 ? 4 : 3;
#define TEMX_LEC(TBSv)                                                                                \
  do {                                                                                                \
    auto kern = eddy_cls_kernel<T, TBSv, MODE, DPW, KIND>;                                            \
    const size_t lds = ((size_t)DPW * NFR * 2 * TBSv * 64 + 8 * 2 * TBSv * 16) * sizeof(double);      \
    static std::atomic<uint64_t> attr_set{0};                                                         \
    if (int rc_ = lds_attr_once(attr_set, pl->device, reinterpret_cast<const void*>(kern), (int)lds)) \
      return rc_;                                                                                     \
    hipLaunchKernelGGL(kern, grid, block, lds, st, fp, pl->D, pl->K, pl->K4, pl->ycls.d(),            \
                       static_cast<const int4*>(pl->crow.p), cuts, pl->colscale.d(), C, partial,      \
                       sp.nsplit, sp.ndt, eo);                                                        \
  } while (0)
  switch (pl->TBS) {
    case 2: TEMX_LEC(2); break;
    case 4: TEMX_LEC(4); break;
    case 7: TEMX_LEC(7); break;
    default: TEMX_LEC(8); break;
  }
#undef TEMX_LEC
  HIPCHK(hipGetLastError());
  return TEMX_OK;
}

template <typename T, int MODE, int KIND>
static int launch_eddy_cls_t(temx_plan* pl, const FieldPtrs<4>& fp, const double* C, double* partial,
                             const Split& sp, const EddyOut& eo, hipStream_t st) {
  switch (sp.dpw) {
    case 1: return launch_eddy_cls_d<T, MODE, 1, KIND>(pl, fp, C, partial, sp, eo, st);
    case 2: return launch_eddy_cls_d<T, MODE, 2, KIND>(pl, fp, C, partial, sp, eo, st);
    default: return launch_eddy_cls_d<T, MODE, 4, KIND>(pl, fp, C, partial, sp, eo, st);
  }
}

template <int DPW, int KIND>
static int launch_flux_cls_d(temx_plan* pl, const double* C, double* partial, const Split& sp, hipStream_t st) {
  dim3 grid(sp.grid), block(512);
#define TEMX_LFC(TBSv)                                                                                \
  do {                                                                                                \
    auto kern = flux_cls_kernel<TBSv, DPW, KIND>;                                                     \
    const size_t lds = ((size_t)DPW * 4 * 2 * TBSv * 64 + 8 * 2 * TBSv * 16) * sizeof(double);        \
    static std::atomic<uint64_t> attr_set{0};                                                         \
    if (int rc_ = lds_attr_once(attr_set, pl->device, reinterpret_cast<const void*>(kern), (int)lds)) \
      return rc_;                                                                                     \
    hipLaunchKernelGGL(kern, grid, block, lds, st, pl->D, pl->K, pl->K4, pl->ycls.d(), pl->csum.d(),  \
                       pl->csq.d(), pl->ccnt.d(), pl->cgroups, C, partial, sp.nsplit, sp.ndt);        \
  } while (0)
  switch (pl->TBS) {
    case 2: TEMX_LFC(2); break;
    case 4: TEMX_LFC(4); break;
    case 7: TEMX_LFC(7); break;
    default: TEMX_LFC(8); break;
  }
#undef TEMX_LFC
  HIPCHK(hipGetLastError());
  return TEMX_OK;
}

template <int KIND>
static int launch_flux_cls(temx_plan* pl, const double* C, double* partial, const Split& sp, hipStream_t st) {
  switch (sp.dpw) {
    case 1: return launch_flux_cls_d<1, KIND>(pl, C, partial, sp, st);
    case 2: return launch_flux_cls_d<2, KIND>(pl, C, partial, sp, st);
    default: return launch_flux_cls_d<4, KIND>(pl, C, partial, sp, st);
  }
}

// ---- large-L class path ----------------------------------------------------------------------------
// sweep 1: the class sums of the four fields and of u v, u omega, v T (no projection)
static int launch_class_sums(temx_plan* pl, const FieldPtrs<4>& fp, int dtype, hipStream_t st) {
  const Split& sp = pl->sp_cproj4;
  const int2* cuts = nullptr;
  if (int rc = class_cuts(pl, sp.nsplit, &cuts, true)) return rc;
  dim3 grid(sp.grid), block(256);
#define TEMX_LCS(Tv)                                                                                  \
  hipLaunchKernelGGL((project_cls_kernel<Tv, 4, 4, 2, TEMX_CLS_OP_WPS, cls_proj_pd<Tv>(true, 4), true, false>), grid, block, 0, st, \
                     fp, pl->D, pl->K, (const double*)nullptr, static_cast<const int4*>(pl->crow.p), cuts,          \
                     pl->colscale.d(), 2, (double*)nullptr, sp.nsplit, sp.ndt, pl->csum.d())
  if (dtype == TEMX_F64) TEMX_LCS(double);
  else if (dtype == TEMX_F32) TEMX_LCS(float);
  else return fail(TEMX_EINVAL, "dtype must be TEMX_F64 or TEMX_F32");
#undef TEMX_LCS
  HIPCHK(hipGetLastError());
  return TEMX_OK;
}

// B[NQ][K][D] = sum over classes of Y_l (S_N +- S_S) for NQ sums of the records in rec, slice by slice
template <int NQ>
static int project_sums(temx_plan* pl, const double* rec, int RS, int row0, double* B, hipStream_t st) {
  const Split& sp = pl->sp_cflux;
  const int64_t D = pl->D;
  dim3 grid(sp.grid), block(512);
  int rc;
  for (int sl = 0; sl < pl->nslice; ++sl) {
    const int Ks = std::min(64, pl->K - 64 * sl);
#define TEMX_LSP(DPWv)                                                                                \
  do {                                                                                                \
    auto kern = sums_project_kernel<NQ, DPWv>;                                                        \
    const size_t lds = ((size_t)DPWv * NQ * 16 * 64 + 8 * 256) * sizeof(double);                      \
    static std::atomic<uint64_t> attr_set{0};                                                         \
    if (int rc_ = lds_attr_once(attr_set, pl->device, reinterpret_cast<const void*>(kern), (int)lds)) \
      return rc_;                                                                                     \
    hipLaunchKernelGGL(kern, grid, block, lds, st, D, pl->K, 64 * sl, pl->ycls_l.d() + sl * pl->ycls_lstride, \
                       rec, RS, row0, pl->cgroups, pl->partial.d(), sp.nsplit, sp.ndt);               \
  } while (0)
    switch (sp.dpw) {
      case 1: TEMX_LSP(1); break;
      case 2: TEMX_LSP(2); break;
      default: TEMX_LSP(4); break;
    }
#undef TEMX_LSP
    HIPCHK(hipGetLastError());
    if ((rc = launch_reduce(pl, pl->partial.d(), sp.nsplit, (int64_t)NQ * Ks * D, pl->Bs.d(), st))) return rc;
    for (int q = 0; q < NQ; ++q)
      HIPCHK(hipMemcpyAsync(B + ((int64_t)q * pl->K + 64 * sl) * D, pl->Bs.d() + (int64_t)q * Ks * D,
                            (size_t)Ks * D * 8, hipMemcpyDeviceToDevice, st));
  }
  return TEMX_OK;
}

static int launch_flux_large(temx_plan* pl, const double* C, hipStream_t st) {
  const Split& sp = pl->sp_lflux;
  dim3 grid(sp.grid), block(512);
#define TEMX_LFL(NSv)                                                                                 \
  do {                                                                                                \
    auto kern = flux_large_kernel<NSv>;                                                               \
    const size_t lds = ((size_t)NSv * 4 * 16 * 64 + 8 * 256) * sizeof(double);                        \
    static std::atomic<uint64_t> attr_set{0};                                                         \
    if (int rc_ = lds_attr_once(attr_set, pl->device, reinterpret_cast<const void*>(kern), (int)lds)) \
      return rc_;                                                                                     \
    hipLaunchKernelGGL(kern, grid, block, lds, st, pl->D, pl->K, pl->K4, pl->ycls_l.d(), pl->ycls_lstride, \
                       pl->csum.d(), pl->ccnt.d(), pl->cgroups, C, pl->pbuf.d(), sp.nsplit, sp.ndt);  \
  } while (0)
  switch (pl->nslice) {
    case 2: TEMX_LFL(2); break;
    case 3: TEMX_LFL(3); break;
    default: TEMX_LFL(4); break;
  }
#undef TEMX_LFL
  HIPCHK(hipGetLastError());
  return TEMX_OK;
}

template <typename T, int NF>
static int launch_project_sym_t(temx_plan* pl, const FieldPtrs<NF>& fp, int64_t D, const double* colscale,
                                int sfield, double* partial, const Split& sp, hipStream_t st) {
  dim3 grid(sp.grid), block(256);
#define TEMX_LPS(TBSv, NFWv, WPSv)                                                                  \
  hipLaunchKernelGGL((project_sym_kernel<T, NF, NFWv, TBSv, WPSv>), grid, block, 0, st, fp, D, pl->K, \
                     pl->ysym.d(), static_cast<const int*>(pl->rows.p), pl->npg, pl->npg_alloc * 4, \
                     colscale, sfield, partial, sp.nsplit, sp.ndt)
  if (NF == 4 && sp.dpw == 1) {     // small ragged D: one d-tile per workgroup, one field per wave
    switch (pl->TBS) {
      case 2: TEMX_LPS(2, 1, SYM_PROJ_E_WPS); break;
      case 4: TEMX_LPS(4, 1, SYM_PROJ_E_WPS); break;
      case 7: TEMX_LPS(7, 1, SYM_PROJ_E_WPS); break;
      default: TEMX_LPS(8, 1, SYM_PROJ_E_WPS); break;
    }
  } else {
    switch (pl->TBS) {
      case 2: TEMX_LPS(2, NF, 2); break;
      case 4: TEMX_LPS(4, NF, 2); break;
      case 7: TEMX_LPS(7, NF, 2); break;
      default: TEMX_LPS(8, NF, 2); break;
    }
  }
#undef TEMX_LPS
  HIPCHK(hipGetLastError());
  return TEMX_OK;
}

template <int NF>
static int launch_project_sym(temx_plan* pl, const FieldPtrs<NF>& fp, int dtype, int64_t D,
                              const double* colscale, int sfield, double* partial, const Split& sp,
                              hipStream_t st) {
  if (dtype == TEMX_F64) return launch_project_sym_t<double, NF>(pl, fp, D, colscale, sfield, partial, sp, st);
  if (dtype == TEMX_F32) return launch_project_sym_t<float, NF>(pl, fp, D, colscale, sfield, partial, sp, st);
  return fail(TEMX_EINVAL, "dtype must be TEMX_F64 or TEMX_F32");
}

template <typename T, int MODE, int DPW, int KIND>
static int launch_eddy_sym_d(temx_plan* pl, const FieldPtrs<4>& fp, const double* C, double* partial,
                             const Split& sp, const EddyOut& eo, hipStream_t st) {
  dim3 grid(sp.grid), block(512);
  constexpr int NFR = KIND == 0 ? 4 : 3;
#define TEMX_LES(TBSv)                                                                                \
  do {                                                                                                \
    auto kern = eddy_sym_kernel<T, TBSv, MODE, DPW, KIND>;                                            \
    const size_t lds = ((size_t)DPW * NFR * 2 * TBSv * 64 + 8 * 2 * TBSv * 16) * sizeof(double);      \
    static std::atomic<uint64_t> attr_set{0};   /* per instantiation, one bit per device */           \
    if (int rc_ = lds_attr_once(attr_set, pl->device, reinterpret_cast<const void*>(kern), (int)lds)) \
      return rc_;                                                                                     \
    hipLaunchKernelGGL(kern, grid, block, lds, st, fp, pl->D, pl->K, pl->K4, pl->ysym.d(),            \
                       static_cast<const int*>(pl->rows.p), pl->npg, pl->npg_alloc * 4, pl->npair,    \
                       pl->colscale.d(), C, partial, sp.nsplit, sp.ndt, eo);                          \
  } while (0)
  switch (pl->TBS) {
    case 2: TEMX_LES(2); break;
    case 4: TEMX_LES(4); break;
    case 7: TEMX_LES(7); break;
    default: TEMX_LES(8); break;
  }
#undef TEMX_LES
  HIPCHK(hipGetLastError());
  return TEMX_OK;
}

template <typename T, int MODE, int KIND>
static int launch_eddy_sym_t(temx_plan* pl, const FieldPtrs<4>& fp, const double* C, double* partial,
                             const Split& sp, const EddyOut& eo, hipStream_t st) {
  switch (sp.dpw) {
    case 1: return launch_eddy_sym_d<T, MODE, 1, KIND>(pl, fp, C, partial, sp, eo, st);
    case 2: return launch_eddy_sym_d<T, MODE, 2, KIND>(pl, fp, C, partial, sp, eo, st);
    default: return launch_eddy_sym_d<T, MODE, 4, KIND>(pl, fp, C, partial, sp, eo, st);
  }
}

// ------------------------------------------------------------------------------------------------
// host linear algebra: Cholesky inverse of the K x K Gram matrix (K <= 64)
// ------------------------------------------------------------------------------------------------
// Li = L^-1 for G = L L^T (long double); -1 when G is not numerically positive definite
static int spd_factor(const double* G, int K, std::vector<long double>& Li) {
  std::vector<long double> Lm((size_t)K * K, 0.0L);
  Li.assign((size_t)K * K, 0.0L);
  for (int i = 0; i < K; ++i) {
    for (int j = 0; j <= i; ++j) {
      long double s = G[i * K + j];
      for (int k = 0; k < j; ++k) s -= Lm[i * K + k] * Lm[j * K + k];
      if (i == j) {
        if (!(s > 0.0L) || !(s <= 1e300L)) return -1;
        Lm[i * K + i] = sqrtl(s);
      } else {
        Lm[i * K + j] = s / Lm[j * K + j];
      }
    }
  }
  // a rank-deficient Gram shows up as a tiny pivot relative to the diagonal
  for (int i = 0; i < K; ++i)
    if (Lm[i * K + i] * Lm[i * K + i] < 1e-13L * (long double)G[i * K + i]) return -1;
  for (int c = 0; c < K; ++c) {  // Li = L^-1 by forward substitution
    for (int i = c; i < K; ++i) {
      long double s = (i == c) ? 1.0L : 0.0L;
      for (int k = c; k < i; ++k) s -= Lm[i * K + k] * Li[k * K + c];
      Li[i * K + c] = s / Lm[i * K + i];
    }
  }
  return 0;
}

// G^-1 = L^-T L^-1
static void inverse_from_factor(const std::vector<long double>& Li, int K, double* Ginv) {
  for (int i = 0; i < K; ++i)
    for (int j = 0; j < K; ++j) {
      long double s = 0.0L;
      for (int k = std::max(i, j); k < K; ++k) s += Li[k * K + i] * Li[k * K + j];
      Ginv[i * K + j] = (double)s;
    }
}

// Pseudo-inverse of the symmetric positive semi-definite Gram matrix by cyclic Jacobi rotations
// (K <= 64).  Used when Cholesky fails: pinv(Y0) = pinv(G) Y0^T holds for any rank, which is the
// minimum-norm semantics of the reference's lstsq (gelsd) for a rank-deficient Y0 -- fewer distinct
// latitudes than harmonics.  Eigenvalues below 1e-12 * lambda_max are treated as zero.
static int sym_pinv(const double* G, int K, double* Ginv, int* rank_out) {
  std::vector<long double> A((size_t)K * K), V((size_t)K * K, 0.0L);
  for (int i = 0; i < K * K; ++i) A[i] = G[i];
  for (int i = 0; i < K; ++i) V[(size_t)i * K + i] = 1.0L;
  for (int sweep = 0; sweep < 100; ++sweep) {
    long double off = 0.0L, diag = 0.0L;
    for (int i = 0; i < K; ++i)
      for (int j = 0; j < K; ++j) (i == j ? diag : off) += A[(size_t)i * K + j] * A[(size_t)i * K + j];
    if (off <= 1e-60L * diag) break;
    for (int p = 0; p < K - 1; ++p)
      for (int q = p + 1; q < K; ++q) {
        const long double apq = A[(size_t)p * K + q];
        if (apq == 0.0L) continue;
        const long double theta = (A[(size_t)q * K + q] - A[(size_t)p * K + p]) / (2.0L * apq);
        const long double t = (theta >= 0 ? 1.0L : -1.0L) / (fabsl(theta) + sqrtl(theta * theta + 1.0L));
        const long double c = 1.0L / sqrtl(t * t + 1.0L), sn = t * c;
        for (int k = 0; k < K; ++k) {   // A <- A J
          const long double akp = A[(size_t)k * K + p], akq = A[(size_t)k * K + q];
          A[(size_t)k * K + p] = c * akp - sn * akq;
          A[(size_t)k * K + q] = sn * akp + c * akq;
        }
        for (int k = 0; k < K; ++k) {   // A <- J^T A
          const long double apk = A[(size_t)p * K + k], aqk = A[(size_t)q * K + k];
          A[(size_t)p * K + k] = c * apk - sn * aqk;
          A[(size_t)q * K + k] = sn * apk + c * aqk;
        }
        for (int k = 0; k < K; ++k) {   // V <- V J
          const long double vkp = V[(size_t)k * K + p], vkq = V[(size_t)k * K + q];
          V[(size_t)k * K + p] = c * vkp - sn * vkq;
          V[(size_t)k * K + q] = sn * vkp + c * vkq;
        }
      }
  }
  long double lmax = 0.0L;
  for (int i = 0; i < K; ++i) lmax = std::max(lmax, A[(size_t)i * K + i]);
  if (!(lmax > 0.0L)) return -1;
  int rank = 0;
  std::vector<long double> inv(K, 0.0L);
  for (int i = 0; i < K; ++i)
    if (A[(size_t)i * K + i] > 1e-12L * lmax) {
      inv[i] = 1.0L / A[(size_t)i * K + i];
      ++rank;
    }
  for (int i = 0; i < K; ++i)
    for (int j = 0; j < K; ++j) {
      long double s2 = 0.0L;
      for (int k = 0; k < K; ++k) s2 += V[(size_t)i * K + k] * inv[k] * V[(size_t)j * K + k];
      Ginv[(size_t)i * K + j] = (double)s2;
    }
  *rank_out = rank;
  return 0;
}

// np.gradient(f, x) coefficient table out[i] = a f[i-1] + b f[i] + c f[i+1], edge_order = 1
// (tem_util.py:154, 192).  numpy switches to the uniform formula only when diff(x) is bit-uniform.
static void gradient_table(const std::vector<double>& x, std::vector<double>& tab) {
  const int n = (int)x.size();
  tab.assign((size_t)n * 3, 0.0);
  std::vector<double> dx(n - 1);
  for (int i = 0; i + 1 < n; ++i) dx[i] = x[i + 1] - x[i];
  bool uniform = true;
  for (int i = 1; i + 1 < n; ++i) uniform = uniform && (dx[i] == dx[0]);
  for (int i = 1; i + 1 < n; ++i) {
    if (uniform) {
      tab[i * 3 + 0] = -1.0 / (2.0 * dx[0]);
      tab[i * 3 + 2] = 1.0 / (2.0 * dx[0]);
    } else {
      const double d1 = dx[i - 1], d2 = dx[i];
      tab[i * 3 + 0] = -(d2) / (d1 * (d1 + d2));
      tab[i * 3 + 1] = (d2 - d1) / (d1 * d2);
      tab[i * 3 + 2] = d1 / (d2 * (d1 + d2));
    }
  }
  tab[0 * 3 + 1] = -1.0 / dx[0];
  tab[0 * 3 + 2] = 1.0 / dx[0];
  tab[(n - 1) * 3 + 0] = -1.0 / dx[n - 2];
  tab[(n - 1) * 3 + 1] = 1.0 / dx[n - 2];
}

// The fused second sweep reads ONE set of Y0 blocks for the reconstruction and for the projection, so it
// serves neither K > 64 nor weights mode (projection rows scaled by 4 pi w, reconstruction rows not).
static inline bool unfused_stage2(const temx_plan* pl) { return pl->large || pl->weighted; }

static inline hipStream_t S_(void* s) { return static_cast<hipStream_t>(s); }
static inline const Split& eddy_split(const temx_plan* pl) {
  return pl->cls ? pl->sp_ceddy : (pl->sym ? pl->sp_seddy : pl->sp_eddy);
}
static inline int eddy_slabs(const temx_plan* pl) {   // partial slabs the eddy sweep writes per product
  if (pl->cls) return pl->sp_ceddy.nsplit;           // the class sweep adds its waves up in LDS first
  return eddy_split(pl).nsplit * (8 / eddy_split(pl).dpw);
}
static inline bool sym_project(const temx_plan* pl, int nf) {   // paired project sweep needs d-quads
  return pl->sym && (nf == 4 ? pl->sp_sproj4.nsplit : pl->sp_sproj1.nsplit) > 0;
}

template <int KIND>
static int run_eddy(temx_plan* pl, const FieldPtrs<4>& fp, int dtype, const double* C, double* partial,
                    const EddyOut* eo, hipStream_t st) {
  EddyOut none{};
  if (pl->cls) {
    if (dtype == TEMX_F64)
      return eo ? launch_eddy_cls_t<double, 1, KIND>(pl, fp, C, partial, pl->sp_ceddy, *eo, st)
                : launch_eddy_cls_t<double, 0, KIND>(pl, fp, C, partial, pl->sp_ceddy, none, st);
    if (dtype == TEMX_F32)
      return eo ? launch_eddy_cls_t<float, 1, KIND>(pl, fp, C, partial, pl->sp_ceddy, *eo, st)
                : launch_eddy_cls_t<float, 0, KIND>(pl, fp, C, partial, pl->sp_ceddy, none, st);
    return fail(TEMX_EINVAL, "dtype must be TEMX_F64 or TEMX_F32");
  }
  if (pl->sym) {
    if (dtype == TEMX_F64)
      return eo ? launch_eddy_sym_t<double, 1, KIND>(pl, fp, C, partial, pl->sp_seddy, *eo, st)
                : launch_eddy_sym_t<double, 0, KIND>(pl, fp, C, partial, pl->sp_seddy, none, st);
    if (dtype == TEMX_F32)
      return eo ? launch_eddy_sym_t<float, 1, KIND>(pl, fp, C, partial, pl->sp_seddy, *eo, st)
                : launch_eddy_sym_t<float, 0, KIND>(pl, fp, C, partial, pl->sp_seddy, none, st);
    return fail(TEMX_EINVAL, "dtype must be TEMX_F64 or TEMX_F32");
  }
  if (dtype == TEMX_F64) {
    return eo ? launch_eddy_t<double, 1, KIND>(pl, fp, C, partial, pl->sp_eddy, *eo, st)
              : launch_eddy_t<double, 0, KIND>(pl, fp, C, partial, pl->sp_eddy, none, st);
  }
  if (dtype == TEMX_F32) {
    return eo ? launch_eddy_t<float, 1, KIND>(pl, fp, C, partial, pl->sp_eddy, *eo, st)
              : launch_eddy_t<float, 0, KIND>(pl, fp, C, partial, pl->sp_eddy, none, st);
  }
  return fail(TEMX_EINVAL, "dtype must be TEMX_F64 or TEMX_F32");
}

static FieldPtrs<4> four(const void* a, const void* b, const void* c, const void* d) {
  FieldPtrs<4> fp;
  fp.p[0] = a; fp.p[1] = b; fp.p[2] = c; fp.p[3] = d;
  return fp;
}

// ---- single-sweep form ---------------------------------------------------------------------------------
// The class-sum stream costs sweep 1 2.2 of its 11.1 ms and feeds a 1.8 ms flux kernel.  This
// form needs neither: ONE sweep projects the four fields up to degree 2L and the three products up to degree
// L (sweep_os_kernel), and the eddy-product sums follow from the Legendre product linearisation
// (os_contract_kernel).  A band-limited reference of low degree, fitted to a subsample of class-groups in a
// short pre-pass with the same kernel, is subtracted first so that no term is a difference of large numbers
// (tools/proto/single_sweep_numerics.py).  Used by temx_tem_run only (the staged, all-reducible stages keep the
// class-sum path).

// Gauss-Legendre nodes and weights on [-1, 1] (long double, Newton on P_n)
static void gauss_legendre(int n, std::vector<long double>& x, std::vector<long double>& w) {
  x.assign(n, 0.0L);
  w.assign(n, 0.0L);
  const long double pi = 3.141592653589793238462643383279502884L;
  for (int i = 0; i < (n + 1) / 2; ++i) {
    long double z = cosl(pi * (i + 0.75L) / (n + 0.5L)), pp = 0.0L;
    for (int it = 0; it < 100; ++it) {
      long double p1 = 1.0L, p2 = 0.0L;
      for (int j = 1; j <= n; ++j) {
        const long double p3 = p2;
        p2 = p1;
        p1 = ((2.0L * j - 1.0L) * z * p2 - (j - 1.0L) * p3) / j;
      }
      pp = n * (z * p1 - p2) / (z * z - 1.0L);
      const long double dz = p1 / pp;
      z -= dz;
      if (fabsl(dz) < 1e-19L) break;
    }
    x[i] = -z;
    x[n - 1 - i] = z;
    w[i] = w[n - 1 - i] = 2.0L / ((1.0L - z * z) * pp * pp);
  }
}

// normalised Y_l^0 at x = cos(colat), l < n (long double)
static void ylm0_row(long double xv, int n, long double* y) {
  const long double pi = 3.141592653589793238462643383279502884L;
  long double pm1 = 1.0L, pc = xv;
  for (int l = 0; l < n; ++l) {
    long double P;
    if (l == 0) {
      P = 1.0L;
    } else if (l == 1) {
      P = xv;
    } else {
      const long double pn = ((2 * l - 1) * xv * pc - (l - 1) * pm1) / l;
      pm1 = pc;
      pc = pn;
      P = pn;
    }
    y[l] = sqrtl((2.0L * l + 1.0L) / (4.0L * pi)) * P;
  }
}

static int os_cuts(temx_plan* pl, bool sub, int nsub, const int2** out) {
  if (!sub) return class_cuts(pl, nsub, out, true);
  auto it = pl->csplits_s.find(nsub);
  if (it == pl->csplits_s.end()) {
    std::vector<int> cut((size_t)2 * (nsub + 1));
    int g = 0;
    const int64_t total = pl->sbatches + (int64_t)TEMX_GROUP_COST * pl->sgroups;     // as class_cuts
    for (int k = 0; k <= nsub; ++k) {
      const int64_t want = total * k / nsub;
      while (g < pl->sgroups && pl->sgbatch0[(size_t)g] + (int64_t)TEMX_GROUP_COST * g < want) ++g;
      if (k == nsub) g = (int)pl->sgroups;
      cut[(size_t)2 * k] = pl->sgbatch0[(size_t)g];
      cut[(size_t)2 * k + 1] = g;
    }
    DevBuf b;
    if (int rc = upload(b, cut.data(), cut.size() * sizeof(int))) return rc;
    it = pl->csplits_s.emplace(nsub, b).first;
  }
  *out = static_cast<const int2*>(it->second.p);
  return TEMX_OK;
}

// tables that depend on the grid and L only (built once per plan, at the first eligible temx_plan_set_tem)
static int build_os_tables(temx_plan* pl) {
  if (pl->os_built) return TEMX_OK;
  const int K = pl->K, L = pl->L;
  const int KX = 2 * L + 1;
  const int TBX = pl->TBS == 7 ? 13 : (pl->TBS == 4 ? 8 : 4);      // blocks per parity of the degree-2L basis (zero padded)
  const int KR = std::min(16, K);
  pl->KX = KX;
  pl->TBX = TBX;
  pl->KR = KR;
  int rc;
  // basis up to degree 2L at the class latitudes, Y basis (no re-orthogonalisation: these are raw projections)
  {
    std::vector<double> nx((size_t)8 * TBX + 8, 0.0);
    for (int l = 0; l < KX; ++l) nx[(size_t)l] = std::sqrt((2.0 * l + 1.0) / (4.0 * M_PI));
    DevBuf nd;
    if ((rc = upload(nd, nx.data(), nx.size() * 8))) return rc;
    rc = pl->ycx.ensure((size_t)(pl->cgroups + 1) * 2 * TBX * 16 * 8);
    if (!rc) {
      hipLaunchKernelGGL(cls_basis_kernel<512>, dim3((unsigned)((pl->cls_npad + 255) / 256)), dim3(256), 0, 0, pl->xc.d(),
                         pl->ncls, pl->cls_npad, KX, TBX, nd.d(), (const double*)nullptr, pl->ycx.d());
      // subsample of class-groups for the reference fit: every S-th group, its batches copied
      // (32 class-groups = 128 latitudes for the 16 coefficients of a column -- the fit only has to be decent, it is
      // removed again exactly; 96 groups cost 0.15 instead of 0.06 ms at ne120 x 72 x 30 and 0.5 instead of 0.2 ms at
      // ne30 x 72 x 91.  TEMX_OPT_OS_SUBSAMPLE / env TEMX_OS_SUBSAMPLE: groups kept)
      const char* ess = getenv("TEMX_OS_SUBSAMPLE");
      const int64_t keep = ess ? std::max(4, atoi(ess)) : pl->os_keep;
      const int64_t S = std::max<int64_t>(1, std::min<int64_t>(256, pl->cgroups / keep));
      std::vector<int> crow_s;
      std::vector<double> xc_s;
      pl->sgbatch0.clear();
      for (int64_t gi = 0; gi < pl->cgroups; gi += S) {
        pl->sgbatch0.push_back((int)(crow_s.size() / (4 * CLS_MB)));
        const size_t e0 = (size_t)pl->gbatch0[(size_t)gi] * 4 * CLS_MB, e1 = (size_t)pl->gbatch0[(size_t)gi + 1] * 4 * CLS_MB;
        crow_s.insert(crow_s.end(), pl->h_crow.begin() + e0, pl->h_crow.begin() + e1);
        for (int k = 0; k < 4; ++k) xc_s.push_back(pl->h_xc[(size_t)gi * 4 + k]);
      }
      pl->sgroups = (int64_t)pl->sgbatch0.size();
      pl->sbatches = (int64_t)(crow_s.size() / (4 * CLS_MB));
      pl->sgbatch0.push_back((int)pl->sbatches);
      {   // one row table per side, for the full table and for the subsample (sweep_os2_kernel: a wave per class side)
        SideTables stb;
        build_side_tables(pl->h_crow, pl->gbatch0, pl->cgroups, CLS_MB, CLS_PADB, CLS_SOUTH, CLS_FIRST, CLS_LAST, CLS_HASPAD_BIT, stb);
        for (int sd = 0; sd < 2 && !rc; ++sd)
          if (!(rc = upload(pl->side_crow[0][sd], stb.crow[sd].data(), stb.crow[sd].size() * sizeof(int))))
            rc = upload(pl->side_gfirst[0][sd], stb.gfirst[sd].data(), stb.gfirst[sd].size() * sizeof(int));
        std::vector<int> crow_full = crow_s;      // (not yet padded) + the terminating entry of sgbatch0
        build_side_tables(crow_full, pl->sgbatch0, pl->sgroups, CLS_MB, CLS_PADB, CLS_SOUTH, CLS_FIRST, CLS_LAST, CLS_HASPAD_BIT, stb);
        for (int sd = 0; sd < 2 && !rc; ++sd)
          if (!(rc = upload(pl->side_crow[1][sd], stb.crow[sd].data(), stb.crow[sd].size() * sizeof(int))))
            rc = upload(pl->side_gfirst[1][sd], stb.gfirst[sd].data(), stb.gfirst[sd].size() * sizeof(int));
      }
      crow_s.resize(crow_s.size() + (size_t)CLS_PADB * 4 * CLS_MB, (int)0x80000000);
      xc_s.resize(xc_s.size() + 4, 0.0);
      DevBuf xs;
      if (!rc && !(rc = upload(pl->crow_s, crow_s.data(), crow_s.size() * sizeof(int))) && !(rc = upload(xs, xc_s.data(), xc_s.size() * 8)) &&
          !(rc = pl->ycx_s.ensure((size_t)(pl->sgroups + 1) * 2 * TBX * 16 * 8))) {
        const int64_t np = (pl->sgroups + 1) * 4;
        // classes beyond the real ones in the last group of the full table have count 0 and x = 0: harmless rows
        hipLaunchKernelGGL(cls_basis_kernel<512>, dim3((unsigned)((np + 255) / 256)), dim3(256), 0, 0, xs.d(), pl->sgroups * 4,
                           np, KX, TBX, nd.d(), (const double*)nullptr, pl->ycx_s.d());
      }
      hipError_t e = hipDeviceSynchronize();
      xs.release();
      if (!rc && e != hipSuccess) rc = fail(TEMX_EHIP, "extended class basis failed: %s", hipGetErrorString(e));
      // Gram matrix of the subsample, degree < KR, and its inverse
      if (!rc) {
        std::vector<long double> y((size_t)KR);
        std::vector<double> Gs((size_t)KR * KR, 0.0);
        std::vector<long double> Gl((size_t)KR * KR, 0.0L);
        for (int64_t gi = 0; gi < pl->cgroups; gi += S)   // (the same groups as above)
          for (int k = 0; k < 4; ++k) {
            const long double nN = pl->h_cnt[(size_t)gi * 8 + k], nS = pl->h_cnt[(size_t)gi * 8 + 4 + k];
            if (nN + nS == 0.0L) continue;
            ylm0_row((long double)pl->h_xc[(size_t)gi * 4 + k], KR, y.data());
            for (int l = 0; l < KR; ++l)
              for (int m = 0; m < KR; ++m) Gl[(size_t)l * KR + m] += (nN + (((l + m) & 1) ? -nS : nS)) * y[l] * y[m];
          }
        for (size_t i = 0; i < Gs.size(); ++i) Gs[i] = (double)Gl[i];
        pl->h_Gs = Gs;
        std::vector<long double> Li;
        std::vector<double> Gi((size_t)KR * KR);
        if (spd_factor(Gs.data(), KR, Li) != 0) {
          // A rank of an ncol-sharded job owns a band of latitudes, which need not determine the fit on its own:
          // the job's subsample is the union over the ranks, its Gram matrix comes through
          // temx_plan_set_os_matrices (the sweeps refuse to run before that).
          if (pl->ext_G) {
            pl->os_need_global = true;
            std::fill(Gi.begin(), Gi.end(), 0.0);
            rc = upload(pl->Gsinv, Gi.data(), Gi.size() * 8);
          } else {
            rc = fail(TEMX_ERANK, "the subsample of latitude classes does not determine a degree-%d reference", KR - 1);
          }
        } else {
          inverse_from_factor(Li, KR, Gi.data());
          rc = upload(pl->Gsinv, Gi.data(), Gi.size() * 8);
        }
      }
    }
    nd.release();
    if (rc) return rc;
  }
  // Gauss-Legendre nodes for the transform form of the product linearisation (exact for degree 4L)
  {
    const int nq = 2 * L + 2;
    pl->NQ = nq;
    std::vector<long double> xq, wq;
    gauss_legendre(nq, xq, wq);
    std::vector<long double> row((size_t)KX);
    std::vector<double> Yq((size_t)nq * KX), w2((size_t)nq);
    const long double twopi = 2.0L * 3.141592653589793238462643383279502884L;
    for (int q = 0; q < nq; ++q) {
      ylm0_row(xq[(size_t)q], KX, row.data());
      for (int k = 0; k < KX; ++k) Yq[(size_t)q * KX + k] = (double)row[(size_t)k];
      w2[(size_t)q] = (double)(twopi * wq[(size_t)q]);
    }
    if ((rc = upload(pl->gaunt, Yq.data(), Yq.size() * 8))) return rc;       // (Yq[q][k])
    pl->h_Yq = Yq;
    if ((rc = upload(pl->wq2, w2.data(), w2.size() * 8))) return rc;
  }
  // Gx[l][k] = sum over the native columns of Y_l Y_k, l < K, k < KX (per latitude class; host threads)
  {
    const int nth = (int)std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
    std::vector<std::vector<double>> part((size_t)nth, std::vector<double>((size_t)K * KX, 0.0));
    std::vector<std::thread> th;
    auto stripe = [&](int t) {
        std::vector<double> yy((size_t)KX);
        std::vector<long double> y((size_t)KX);
        std::vector<double>& Gp = part[(size_t)t];
        for (int64_t ci = t; ci < pl->ncls; ci += nth) {
          const int64_t gi = ci >> 2;
          const int k4 = (int)(ci & 3);
          const double nN = pl->h_cnt[(size_t)gi * 8 + k4], nS = pl->h_cnt[(size_t)gi * 8 + 4 + k4];
          ylm0_row((long double)pl->h_xc[(size_t)ci], KX, y.data());
          for (int k = 0; k < KX; ++k) yy[(size_t)k] = (double)y[(size_t)k];
          const double se = nN + nS, so = nN - nS;
          for (int l = 0; l < K; ++l) {
            const double yl = yy[(size_t)l];
            double* row = &Gp[(size_t)l * KX];
            for (int k = (l & 1); k < KX; k += 2) row[k] += se * yl * yy[(size_t)k];        // l + k even
            for (int k = 1 - (l & 1); k < KX; k += 2) row[k] += so * yl * yy[(size_t)k];    // l + k odd
          }
        }
    };
    // stripe t of the classes per thread; a thread that cannot be started (std::system_error under a process /
    // thread limit -- this is an extern "C" call chain, nothing may propagate) leaves its stripe to the caller.
    // The partial sums stay per stripe, so the result has the same bits however the stripes were run.
    int started = 0;
    try {
      th.reserve((size_t)nth);
      for (; started < nth - 1; ++started) th.emplace_back(stripe, started);
    } catch (...) {
    }
    for (int t = started; t < nth; ++t) stripe(t);
    for (auto& x : th) x.join();
    std::vector<double> Gx((size_t)K * KX, 0.0);
    for (int t = 0; t < nth; ++t)          // fixed order: the same bits whatever the scheduling
      for (size_t i = 0; i < Gx.size(); ++i) Gx[i] += part[(size_t)t][i];
    if ((rc = upload(pl->Gx, Gx.data(), Gx.size() * 8))) return rc;
    pl->h_Gx = std::move(Gx);
  }
  pl->os_built = true;
  return os_upload_blocks(pl);
}

// 16 x 4 A-operand blocks (kernels_osc.hpp) of the R x C matrix A (row-major, leading dimension ld): nrb4 blocks of 4
// rows are padded to whole 16-row blocks, the columns to nkb blocks of 4; zero filled; appended to `out`
static void append_blocks16(std::vector<double>& out, const double* A, int R, int C, int ld, bool transpose, int nrb4, int nkb) {
  const size_t o = out.size();
  const int nrb = (nrb4 + 3) / 4;
  out.resize(o + (size_t)nrb * nkb * 64, 0.0);
  for (int rb = 0; rb < nrb; ++rb)
    for (int t = 0; t < nkb; ++t)
      for (int k = 0; k < 4; ++k)
        for (int m = 0; m < 16; ++m) {
          const int r = 16 * rb + m, c = 4 * t + k;         // element [r][c] of the (transposed) matrix
          if (r < R && c < C) out[o + ((size_t)rb * nkb + t) * 64 + k * 16 + m] = transpose ? A[(size_t)c * ld + r] : A[(size_t)r * ld + c];
        }
}

// the matrices of the single sweep's contraction as MFMA operand blocks: after build_os_tables, and again whenever
// one of them changes (temx_plan_finalize / refine: T, G2inv, G; temx_plan_set_os_matrices: Gx)
static int os_upload_blocks(temx_plan* pl) {
  if (pl->h_T.empty() || pl->h_Ginv.empty() || pl->h_G.empty() || pl->h_Yq.empty() || pl->h_Gx.empty()) return TEMX_OK;
  const int K = pl->K, KX = pl->KX, KR = pl->KR, NQ = pl->NQ;
  const int NBK = pl->TB, NBX = 2 * pl->TBX;
  if (4 * NBK < K || 4 * NBX < KX || 4 * NBX < NQ) return fail(TEMX_EUNSUPPORTED, "single-sweep contraction: no instantiation for L = %d", pl->L);
  std::vector<double> blk;
  size_t off[9];
  const double* T = pl->h_T.data();
  const double* Yq = pl->h_Yq.data();
  off[0] = blk.size(); append_blocks16(blk, T, K, K, K, true, NBK, NBK);                    // T^T
  off[1] = blk.size(); append_blocks16(blk, pl->h_Ginv.data(), K, K, K, false, NBK, NBK);   // G2inv
  off[2] = blk.size(); append_blocks16(blk, T, K, K, K, false, NBK, NBK);                   // T
  off[3] = blk.size(); append_blocks16(blk, pl->h_G.data(), K, KR, K, false, NBK, 4);       // G[:, :KR]
  off[4] = blk.size(); append_blocks16(blk, Yq, NQ, K, KX, false, NBX, NBK);                // Yq[:, :K]
  off[5] = blk.size(); append_blocks16(blk, Yq, NQ, KX, KX, false, NBX, NBX);               // Yq
  off[6] = blk.size(); append_blocks16(blk, Yq, K, NQ, KX, true, NBK, NBX);                 // Yq[:, :K]^T
  off[7] = blk.size(); append_blocks16(blk, Yq, KX, NQ, KX, true, NBX, NBX);                // Yq^T
  off[8] = blk.size(); append_blocks16(blk, pl->h_Gx.data(), K, KX, KX, false, NBK, NBX);   // Gx
  HIPCHK(hipDeviceSynchronize());              // (a launch in flight may still read the old blocks)
  if (int rc = upload(pl->oscblk, blk.data(), blk.size() * 8)) return rc;
  const double* b = pl->oscblk.d();
  pl->osc = OscMats{b + off[0], b + off[1], b + off[2], b + off[3], b + off[4], b + off[5], b + off[6], b + off[7], b + off[8]};
  return TEMX_OK;
}

#ifndef TEMX_OS_DEFER
#define TEMX_OS_DEFER 1
#endif
template <typename T, int KIND>
static int launch_sweep_os_t(temx_plan* pl, const FieldPtrs<4>& fp, bool sub, const double* rho, double* partial,
                             const Split& sp, hipStream_t st) {
  using KD = OsKind<KIND>;
  const int2* cuts = nullptr;
  if (int rc = os_cuts(pl, sub, sp.nsplit, &cuts)) return rc;
  dim3 grid(sp.grid), block(256);
  constexpr int NBR = 2;
  constexpr int PDv = sizeof(T) == 4 ? 4 : 2;       // (a 3-deep ring measured slower for the tracer kind: 8.7 vs 8.4 ms)
  constexpr int DF = TEMX_OS_DEFER;                 // projection of a finished class-group spread over the next 4 batches
  // loads of 1 row x 64 columns (sweep_osr_kernel, the default) or of 4 rows x 16 columns (TEMX_OS_MAP=tile, A/B)
  const bool tile_map = pl->os_tile;
  double* px = partial;
  // (the reference pre-pass needs the first KR rows of px only: pp == NULL tells the kernels to store nothing else --
  //  146 x 8 B per lane and workgroup otherwise, 150 MB at ne120 x 72 x 30 for a sweep that reads 240 MB)
  double* pp = sub ? nullptr : partial + (int64_t)sp.nsplit * KD::NFX * pl->KX * pl->D;
#define TEMX_LOS(TBSv, TBXv)                                                                                        \
  do {                                                                                                              \
    if constexpr (KIND == 3) {                                                                                      \
      if (tile_map) return fail(TEMX_EUNSUPPORTED, "two tracers per sweep: row-map sweeps only");                   \
    }                                                                                                               \
    if constexpr (KIND != 3) if (tile_map) {                                                                        \
      auto kern = sweep_os_kernel<T, TBSv, TBXv, NBR, PDv, KIND, DF>;                                               \
      const size_t lds = ((size_t)4 * (DF ? 2 : 1) * 2 * TBXv * 16 + (size_t)4 * KD::NF * 2 * NBR * 64 +            \
                          (size_t)4 * KD::NP * 2 * TBSv * 64) * 8;                                                  \
      static std::atomic<uint64_t> attr_set{0};                                                                     \
      if (int rc_ = lds_attr_once(attr_set, pl->device, reinterpret_cast<const void*>(kern), (int)lds)) return rc_; \
      hipLaunchKernelGGL(kern, grid, block, lds, st, fp, pl->D, pl->K, pl->KX, sub ? pl->ycx_s.d() : pl->ycx.d(),   \
                         static_cast<const int4*>(sub ? pl->crow_s.p : pl->crow.p), cuts, pl->colscale.d(), rho,    \
                         pl->KR, px, pp, sp.nsplit, sp.ndt);                                                        \
      break;                                                                                                        \
    }                                                                                                               \
    if (sizeof(T) == 4) {   /* fp32 inputs: two waves per SIMD (kernels_op2.hpp, sweep_os2_kernel) */              \
      auto kern = sweep_os2_kernel<float, TBSv, TBXv, NBR, 2, KIND>;                                                \
      const size_t lds = ((size_t)2 * 2 * TBXv * 16 + 16 + (size_t)4 * KD::NF * 2 * NBR * 64 +                      \
                          (size_t)8 * (KD::NP - KD::NPR) * TBSv * 64 + (size_t)(KD::NF + KD::NP) * 512) * 8;        \
      static std::atomic<uint64_t> attr_set{0};                                                                     \
      if (int rc_ = lds_attr_once(attr_set, pl->device, reinterpret_cast<const void*>(kern), (int)lds)) return rc_; \
      const int si = sub ? 1 : 0;                                                                                   \
      hipLaunchKernelGGL(kern, grid, dim3(512), lds, st, fp, pl->D, pl->K, pl->KX, sub ? pl->ycx_s.d() : pl->ycx.d(), \
                         static_cast<const int4*>(pl->side_crow[si][0].p), static_cast<const int4*>(pl->side_crow[si][1].p), \
                         static_cast<const int*>(pl->side_gfirst[si][0].p), static_cast<const int*>(pl->side_gfirst[si][1].p), \
                         cuts, pl->colscale.d(), rho, pl->KR, px, pp, sp.nsplit, sp.ndt);                           \
    } else {                                                                                                        \
      auto kern = sweep_osr_kernel<double, TBSv, TBXv, NBR, 2, KIND>;                                               \
      const size_t lds = ((size_t)2 * 2 * TBXv * 16 + 16 + (size_t)4 * KD::NF * 2 * NBR * 64 +                      \
                          (size_t)4 * (KD::NP - KD::NPR) * 2 * TBSv * 64 + (size_t)2 * (KD::NF + KD::NP) * 256) * 8; \
      static std::atomic<uint64_t> attr_set{0};                                                                     \
      if (int rc_ = lds_attr_once(attr_set, pl->device, reinterpret_cast<const void*>(kern), (int)lds)) return rc_; \
      hipLaunchKernelGGL(kern, grid, block, lds, st, fp, pl->D, pl->K, pl->KX, sub ? pl->ycx_s.d() : pl->ycx.d(),   \
                         static_cast<const int4*>(sub ? pl->crow_s.p : pl->crow.p), cuts, pl->colscale.d(), rho,    \
                         pl->KR, px, pp, sp.nsplit, sp.ndt);                                                        \
    }                                                                                                               \
  } while (0)
  if (pl->TBS == 7 && pl->TBX == 13) TEMX_LOS(7, 13);
  else if (pl->TBS == 4 && pl->TBX == 8) TEMX_LOS(4, 8);
  else if (pl->TBS == 2 && pl->TBX == 4) TEMX_LOS(2, 4);
  else return fail(TEMX_EUNSUPPORTED, "single-sweep form: no instantiation for L = %d", pl->L);
#undef TEMX_LOS
  HIPCHK(hipGetLastError());
  return TEMX_OK;
}

template <int KIND>
static int launch_sweep_os(temx_plan* pl, const FieldPtrs<4>& fp, int dtype, bool sub, const double* rho, double* partial,
                           const Split& sp, hipStream_t st) {
  return dtype == TEMX_F64 ? launch_sweep_os_t<double, KIND>(pl, fp, sub, rho, partial, sp, st)
                           : launch_sweep_os_t<float, KIND>(pl, fp, sub, rho, partial, sp, st);
}

static bool os_supported(const temx_plan* pl) {
  if (!pl->cls || pl->large || pl->weighted || !pl->qbasis || pl->h_crow.empty()) return false;
  const int tbx = (pl->L + 1 + 3) / 4;          // blocks of 4 even harmonics up to degree 2L
  return (pl->TBS == 7 && tbx <= 13) || (pl->TBS == 4 && tbx <= 8) || (pl->TBS == 2 && tbx <= 4);
}

// the single-sweep forms run for both input types (fp32 inputs take the two-waves-per-SIMD sweep);
// TEMX_SINGLE_SWEEP=0 at plan build selects the class-sum forms
static bool os_active(const temx_plan* pl, int dtype) {
  (void)dtype;
  return pl->os_on;
}

// The snapshots [t0, t0 + nts) the tail (solve, contraction, scan, epilogue) is about to describe.  Everything the
// tail leaves in the plan (C4, zb, Ax, tz ...) then has nlev * nts columns.
static void set_tail(temx_plan* pl, int64_t t0, int64_t nts) {
  if (pl->tt0 != t0 || pl->tnt != nts) pl->c4_valid = pl->os_valid = pl->op_valid = pl->tq_valid = pl->xb_valid = false;
  pl->tt0 = t0;
  pl->tnt = nts;
  pl->tD = (int64_t)pl->nlev * nts;
}
static inline bool tail_is_whole(const temx_plan* pl) { return pl->tt0 == 0 && pl->tnt == pl->nt; }

static int tem_stage3_impl(temx_plan* pl, const double* B3, double* results, double* zonal, hipStream_t st);
static int tracer_stage3_impl(temx_plan* pl, const double* Bq2, double* tres, double* tzon, hipStream_t st);

// Time slices of a reduction (kernels.hpp, SliceMap): nsl slices, rows_total rows per slice
static SliceMap slice_map(const temx_plan* pl, int nsl, int64_t rows_total, int64_t row0) {
  SliceMap m;
  if (nsl <= 1) return m;
  m.D = pl->D;
  m.nt = (int)pl->nt;
  m.W = nsl;
  m.chunk = rows_total * pl->nlev * ((pl->nt + nsl - 1) / nsl);
  m.row0 = row0;
  return m;
}

// the contraction on the matrix cores (kernels_osc.hpp): fields, then pairs
template <int NBK>
static int launch_osc_t(temx_plan* pl, const OscFieldsIn& fin, int nf, int nout, double* Bf, double* At, double* ab,
                        const OscPairsIn& pin, int np, double* Bp, int64_t Dt, hipStream_t st) {
  const unsigned gx = (unsigned)((Dt + 15) / 16);       // one workgroup (OSC_W waves) = one d-tile of one field / pair
  hipLaunchKernelGGL((osc_fields_kernel<NBK>), dim3(gx, nf, 2), dim3(OSC_W * 64), (size_t)osc_fields_lds(NBK) * 8, st, fin, pl->osc, pl->K,
                     pl->KX, pl->KR, pl->NQ, Dt, nout, Bf, At, ab, pl->D, (int)pl->tnt, (int)pl->nt, (int)pl->tt0);
  HIPCHK(hipGetLastError());
  hipLaunchKernelGGL((osc_pairs_kernel<NBK>), dim3(gx, np), dim3(OSC_W * 64), (size_t)osc_pairs_lds(NBK) * 8, st, pin, pl->osc, pl->wq2.d(),
                     pl->K, pl->KX, pl->NQ, Dt, Bp);
  HIPCHK(hipGetLastError());
  return TEMX_OK;
}
static int launch_osc(temx_plan* pl, const OscFieldsIn& fin, int nf, int nout, double* Bf, double* At, double* ab,
                      const OscPairsIn& pin, int np, double* Bp, int64_t Dt, hipStream_t st) {
  if (pl->TB == 13 && pl->TBX == 13) return launch_osc_t<13>(pl, fin, nf, nout, Bf, At, ab, pin, np, Bp, Dt, st);
  if (pl->TB == 8 && pl->TBX == 8) return launch_osc_t<8>(pl, fin, nf, nout, Bf, At, ab, pin, np, Bp, Dt, st);
  if (pl->TB == 4 && pl->TBX == 4) return launch_osc_t<4>(pl, fin, nf, nout, Bf, At, ab, pin, np, Bp, Dt, st);
  return fail(TEMX_EUNSUPPORTED, "single-sweep contraction: no instantiation for L = %d", pl->L);
}

// ---- the single sweep in three steps.  A single process runs them back to back (tem_run_os); an ncol-sharded job
// exchanges between them: (1) -> all-reduce of As (4 x KR x D doubles) -> (2) -> reduce-scatter of proj over time
// -> (3) on the snapshots the rank received.
// 1. reference pre-pass: the same sweep over the subsample of class-groups with a zero reference; As[4][KR][D] =
//    raw sums of the four fields on the first KR harmonics (this rank's rows)
static int os_prepass(temx_plan* pl, const FieldPtrs<4>& fp, int dtype, double* As, hipStream_t st) {
  int rc;
  const int64_t D = pl->D, KD4 = (int64_t)4 * pl->KX * D;
  pl->op_valid = pl->c4_valid = pl->tq_valid = pl->os_valid = false;       // no class sums on this path
  if ((rc = launch_sweep_os<0>(pl, fp, dtype, true, pl->rho0.d(), pl->partial.d(), pl->sp_os_s, st))) return rc;
  return launch_reduce_rows(pl, pl->partial.d(), pl->sp_os_s.nsplit, KD4, 4, pl->KX, pl->KR, D, As, st);
}

// 2. reference coefficients from the (global) subsample sums, the sweep, and its reduction: proj = [4 KX + 3 K] rows
//    (projections of the shifted fields to degree 2L, of their products to degree L), whole ([rows][D]) or cut into
//    nsl time slices ([nsl][rows][nlev][ceil(nt / nsl)], the input of a reduce-scatter)
static int os_sweep(temx_plan* pl, const FieldPtrs<4>& fp, int dtype, const double* As, int nsl, double* proj, hipStream_t st) {
  int rc;
  const int64_t D = pl->D;
  const int64_t KD4 = (int64_t)4 * pl->KX * D, KD3 = (int64_t)3 * pl->K * D;
  pl->op_valid = pl->c4_valid = pl->tq_valid = pl->os_valid = false;
  hipLaunchKernelGGL(os_ref_solve_kernel, dim3((unsigned)((D + 15) / 16), 4), dim3(256), 0, st, As, pl->KR, pl->KR, D,
                     pl->Gsinv.d(), pl->rho.d());
  HIPCHK(hipGetLastError());
  TimedLaunch tl{};
  time_begin(pl, 0, st, tl);
  rc = launch_sweep_os<0>(pl, fp, dtype, false, pl->rho.d(), pl->partial.d(), pl->sp_os, st);
  time_end(pl, 0, st, tl);
  if (rc) return rc;
  const int64_t rows = (int64_t)4 * pl->KX + 3 * pl->K;
  if ((rc = launch_reduce(pl, pl->partial.d(), pl->sp_os.nsplit, KD4, proj, st, -1, nullptr, slice_map(pl, nsl, rows, 0)))) return rc;
  return launch_reduce(pl, pl->partial.d() + (int64_t)pl->sp_os.nsplit * KD4, pl->sp_os.nsplit, KD3, nsl > 1 ? proj : proj + KD4, st, -1,
                       nullptr, slice_map(pl, nsl, rows, (int64_t)4 * pl->KX));
}

// 3. the tail for the snapshots [t0, t0 + nts): linearisation (raw sums of the fields and of the eddy products in the
//    plan's basis), coefficients and zonal means, epilogue.  proj_s: [4 KX + 3 K][nlev][nts], summed over the ranks.
static int os_tail(temx_plan* pl, const double* proj_s, int64_t t0, int64_t nts, double* results, double* zonal, hipStream_t st) {
  int rc;
  set_tail(pl, t0, nts);
  const int64_t Dt = pl->tD;
  const int64_t rows = (int64_t)4 * pl->KX + 3 * pl->K;
  if (proj_s != pl->Ax.d())       // the plan keeps the slice: the tracer's single sweep reuses the projections of v and omega
    HIPCHK(hipMemcpyAsync(pl->Ax.p, proj_s, (size_t)rows * Dt * 8, hipMemcpyDeviceToDevice, st));
  {
    TimedLaunch t2{};
    time_begin(pl, 1, st, t2);
    const size_t lds = os_contract_lds(pl->K, pl->KX, pl->NQ) * 8;
    static std::atomic<uint64_t> attr_set{0};
    if ((rc = lds_attr_once(attr_set, pl->device, reinterpret_cast<const void*>(os_contract_kernel<0>), 160 * 1024))) return rc;
    const double* Pp = pl->Ax.d() + (int64_t)4 * pl->KX * Dt;
    if (pl->osc_lds) {
      OsFields in;
      for (int f = 0; f < 4; ++f) {
        in.A[f] = pl->Ax.d() + (int64_t)f * pl->KX * Dt;
        in.rho[f] = pl->rho.d() + (int64_t)f * pl->KR * pl->D;
      }
      hipLaunchKernelGGL(os_contract_kernel<0>, dim3((unsigned)((Dt + OSC - 1) / OSC)), dim3(256), lds, st, in, Pp, pl->K, pl->KX,
                         pl->KR, pl->NQ, Dt, pl->T.d(), pl->Ginv.d(), pl->G.d(), pl->Gx.d(), pl->gaunt.d(), pl->wq2.d(),
                         pl->B4.d(), pl->B3.d(), pl->D, (int)nts, (int)pl->nt, (int)t0);
      HIPCHK(hipGetLastError());
    } else {
      using KD = OsKind<0>;
      const int64_t QD = (int64_t)pl->NQ * Dt;
      OscFieldsIn fin;
      OscPairsIn pin{};
      for (int f = 0; f < 4; ++f) {
        fin.A[f] = pl->Ax.d() + (int64_t)f * pl->KX * Dt;
        fin.rho[f] = pl->rho.d() + (int64_t)f * pl->KR * pl->D;
      }
      for (int k = 0; k < 3; ++k) {
        pin.At_a[k] = pl->osAt.d() + KD::pa(k) * QD; pin.At_b[k] = pl->osAt.d() + KD::pb(k) * QD;
        pin.ab_a[k] = pl->osAb.d() + KD::pa(k) * QD; pin.ab_b[k] = pl->osAb.d() + KD::pb(k) * QD;
        pin.P[k] = Pp + (int64_t)k * pl->K * Dt;
      }
      if ((rc = launch_osc(pl, fin, 4, 4, pl->B4.d(), pl->osAt.d(), pl->osAb.d(), pin, 3, pl->B3.d(), Dt, st))) return rc;
    }
    time_end(pl, 1, st, t2);
  }
  // as after stage 2: coefficients and zonal means of the four fields, then the epilogue
  if ((rc = launch_solve(pl, pl->B4.d(), 4, Dt, pl->C4.d(), pl->zb.d(), st))) return rc;
  if ((rc = tem_stage3_impl(pl, pl->B3.d(), results, zonal, st))) return rc;
  pl->c4_valid = true;
  pl->os_valid = true;            // Ax, rho describe these fields (and these snapshots): the tracer's single sweep may follow
  return TEMX_OK;
}

static int tem_run_os(temx_plan* pl, const FieldPtrs<4>& fp, int dtype, double* results, double* zonal, void* stream) {
  hipStream_t st = S_(stream);
  int rc;
  if ((rc = os_prepass(pl, fp, dtype, pl->Axs.d(), st))) return rc;
  if ((rc = os_sweep(pl, fp, dtype, pl->Axs.d(), 1, pl->Ax.d(), st))) return rc;
  return os_tail(pl, pl->Ax.d(), 0, pl->nt, results, zonal, st);
}

static int tracer_ws(temx_plan* pl);

// Tracers in the single-sweep form: (q, v, omega) -- or (q1, q2, v, omega): one read of v and omega for two tracers --
// read once, no class sums.  The degree-2L projections and the references of v and omega are those the TEM run left in
// the plan (os_valid): the same v and omega must be handed over.  Each q gets its own reference from a pre-pass, is
// projected to degree 2L, q v and q omega to degree L.  The same three steps as the TEM run; for nq tracers:
// Asq[nq][KR][D], projq = nq KX + 2 nq K rows (the q's, then q1 v, q1 omega, q2 v, q2 omega).
static int tracer_os_ws(temx_plan* pl, int nq) {
  const int64_t D = pl->D;
  int rc;
  if ((rc = tracer_ws(pl))) return rc;
  if ((rc = pl->Axq.ensure((size_t)nq * (pl->KX + 2 * pl->K + pl->KR) * D * 8))) return rc;   // projections, then the pre-pass sums
  if ((rc = pl->osAtq.ensure((size_t)nq * pl->NQ * D * 8))) return rc;
  if ((rc = pl->osAbq.ensure((size_t)nq * pl->NQ * D * 8))) return rc;
  if ((rc = pl->Bqp.ensure((size_t)nq * 3 * pl->K * D * 8))) return rc;                       // raw sums of the q's, then of the products
  return pl->rho_t.ensure((size_t)4 * pl->KR * D * 8);
}

// fp: (q, v, omega, -) for one tracer, (q1, q2, v, omega) for two
static int tracer_os_prepass(temx_plan* pl, int nq, const FieldPtrs<4>& fp, int dtype, double* Asq, hipStream_t st) {
  int rc;
  const int64_t D = pl->D;
  if ((rc = tracer_os_ws(pl, nq))) return rc;
  pl->tq_valid = false;
  // (the references of v, omega do not matter here: only the projections of the q's are used)
  rc = nq == 2 ? launch_sweep_os<3>(pl, fp, dtype, true, pl->rho0.d(), pl->partial.d(), pl->sp_os_s, st)
               : launch_sweep_os<1>(pl, fp, dtype, true, pl->rho0.d(), pl->partial.d(), pl->sp_os_s, st);
  if (rc) return rc;
  return launch_reduce_rows(pl, pl->partial.d(), pl->sp_os_s.nsplit, (int64_t)nq * pl->KX * D, nq, pl->KX, pl->KR, D, Asq, st);
}

static int tracer_os_sweep(temx_plan* pl, int nq, const FieldPtrs<4>& fp, int dtype, const double* Asq, int nsl, double* projq,
                           hipStream_t st) {
  int rc;
  const int64_t D = pl->D, KXD = (int64_t)pl->KX * D, KD = (int64_t)pl->K * D, KRD = (int64_t)pl->KR * D;
  if ((rc = tracer_os_ws(pl, nq))) return rc;
  pl->tq_valid = false;
  // references: (rho_q, rho_v, rho_omega) / (rho_q1, rho_q2, rho_v, rho_omega), v and omega as the TEM run fitted them
  hipLaunchKernelGGL(os_ref_solve_kernel, dim3((unsigned)((D + 15) / 16), nq), dim3(256), 0, st, Asq, pl->KR, pl->KR, D,
                     pl->Gsinv.d(), pl->rho_t.d());
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpyAsync(pl->rho_t.d() + nq * KRD, pl->rho.d() + 1 * KRD, (size_t)KRD * 8, hipMemcpyDeviceToDevice, st));         // v
  HIPCHK(hipMemcpyAsync(pl->rho_t.d() + (nq + 1) * KRD, pl->rho.d() + 3 * KRD, (size_t)KRD * 8, hipMemcpyDeviceToDevice, st));   // omega
  rc = nq == 2 ? launch_sweep_os<3>(pl, fp, dtype, false, pl->rho_t.d(), pl->partial.d(), pl->sp_os, st)
               : launch_sweep_os<1>(pl, fp, dtype, false, pl->rho_t.d(), pl->partial.d(), pl->sp_os, st);
  if (rc) return rc;
  const int64_t rows = (int64_t)nq * (pl->KX + 2 * pl->K);
  if ((rc = launch_reduce(pl, pl->partial.d(), pl->sp_os.nsplit, nq * KXD, projq, st, -1, nullptr, slice_map(pl, nsl, rows, 0)))) return rc;
  return launch_reduce(pl, pl->partial.d() + (int64_t)pl->sp_os.nsplit * nq * KXD, pl->sp_os.nsplit, 2 * nq * KD,
                       nsl > 1 ? projq : projq + nq * KXD, st, -1, nullptr, slice_map(pl, nsl, rows, (int64_t)nq * pl->KX));
}

// the tracers' tail for the snapshots the TEM tail worked on (set_tail): projq_s [nq KX + 2 nq K][nlev][tnt]
static int tracer_os_tail(temx_plan* pl, int nq, const double* projq_s, double* const* tres, double* const* tzon, hipStream_t st) {
  int rc;
  const int64_t Dt = pl->tD, KXD = (int64_t)pl->KX * Dt, KDt = (int64_t)pl->K * Dt, KRD = (int64_t)pl->KR * pl->D;
  double* Bq = pl->Bqp.d();                    // [nq][K][Dt]
  double* Bq2 = pl->Bqp.d() + nq * KDt;        // [2 nq][K][Dt]
  if (pl->osc_lds) {
    if (nq != 1) return fail(TEMX_EUNSUPPORTED, "two tracers per sweep need the contraction on the matrix cores (TEMX_OPT_OS_CONTRACT = 0)");
    const size_t lds = os_contract_lds(pl->K, pl->KX, pl->NQ) * 8;
    static std::atomic<uint64_t> attr_set{0};
    if ((rc = lds_attr_once(attr_set, pl->device, reinterpret_cast<const void*>(os_contract_kernel<1>), 160 * 1024))) return rc;
    OsFields in{};
    in.A[0] = projq_s;
    in.A[1] = pl->Ax.d() + 1 * KXD;
    in.A[2] = pl->Ax.d() + 3 * KXD;
    in.rho[0] = pl->rho_t.d();
    in.rho[1] = pl->rho.d() + 1 * KRD;
    in.rho[2] = pl->rho.d() + 3 * KRD;
    hipLaunchKernelGGL(os_contract_kernel<1>, dim3((unsigned)((Dt + OSC - 1) / OSC)), dim3(256), lds, st, in, projq_s + KXD,
                       pl->K, pl->KX, pl->KR, pl->NQ, Dt, pl->T.d(), pl->Ginv.d(), pl->G.d(), pl->Gx.d(), pl->gaunt.d(),
                       pl->wq2.d(), Bq, Bq2, pl->D, (int)pl->tnt, (int)pl->nt, (int)pl->tt0);
    HIPCHK(hipGetLastError());
  } else {
    // the q's are synthesised here; v and omega at the nodes are those the TEM tail left in the plan (osAt / osAb)
    const int64_t QD = (int64_t)pl->NQ * Dt;
    OscFieldsIn fin{};
    OscPairsIn pin{};
    const int other[2] = {1, 3};                   // v, omega among the TEM run's four fields
    for (int i = 0; i < nq; ++i) {
      fin.A[i] = projq_s + i * KXD;
      fin.rho[i] = pl->rho_t.d() + i * KRD;
      for (int k = 0; k < 2; ++k) {
        const int p = 2 * i + k;
        pin.At_a[p] = pl->osAtq.d() + i * QD; pin.ab_a[p] = pl->osAbq.d() + i * QD;
        pin.At_b[p] = pl->osAt.d() + other[k] * QD; pin.ab_b[p] = pl->osAb.d() + other[k] * QD;
        pin.P[p] = projq_s + nq * KXD + (int64_t)p * KDt;
      }
    }
    if ((rc = launch_osc(pl, fin, nq, nq, Bq, pl->osAtq.d(), pl->osAbq.d(), pin, 2 * nq, Bq2, Dt, st))) return rc;
  }
  // per tracer: coefficients Ct = (C_q, C_v, C_w) and qb -> tz[0], as the other tracer stage 2 forms leave them
  // (Ct / tz describe the LAST tracer afterwards: temx_tracer_eddy materialises that one)
  const size_t slab = (size_t)pl->K4 * Dt * 8;
  for (int i = 0; i < nq; ++i) {
    if ((rc = launch_solve(pl, Bq + i * KDt, 1, Dt, pl->Ct.d(), pl->tz.d(), st))) return rc;
    HIPCHK(hipMemcpyAsync((char*)pl->Ct.p + slab, (char*)pl->C4.p + slab, slab, hipMemcpyDeviceToDevice, st));
    HIPCHK(hipMemcpyAsync((char*)pl->Ct.p + 2 * slab, (char*)pl->C4.p + 3 * slab, slab, hipMemcpyDeviceToDevice, st));
    if ((rc = tracer_stage3_impl(pl, Bq2 + 2 * i * KDt, tres[i], tzon ? tzon[i] : nullptr, st))) return rc;
  }
  return TEMX_OK;
}

static FieldPtrs<4> tracer_fields(int nq, const void* const* q, const void* va, const void* wap) {
  return nq == 2 ? four(q[0], q[1], va, wap) : four(q[0], va, wap, nullptr);
}

static int tracer_run_os(temx_plan* pl, int nq, const void* const* q, const void* va, const void* wap, int dtype,
                         double* const* tres, double* const* tzon, void* stream) {
  hipStream_t st = S_(stream);
  int rc;
  if ((rc = tracer_os_ws(pl, nq))) return rc;
  const FieldPtrs<4> fp = tracer_fields(nq, q, va, wap);
  double* Asq = pl->Axq.d() + (int64_t)nq * (pl->KX + 2 * pl->K) * pl->D;
  if ((rc = tracer_os_prepass(pl, nq, fp, dtype, Asq, st))) return rc;
  if ((rc = tracer_os_sweep(pl, nq, fp, dtype, Asq, 1, pl->Axq.d(), st))) return rc;
  return tracer_os_tail(pl, nq, pl->Axq.d(), tres, tzon, st);
}

// ------------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------------
// No exception may cross the C ABI (the callers are ctypes and C): host allocations (std::bad_alloc) and thread
// starts (std::system_error) inside an entry point are the sources.  Every multi-statement entry point is a
// function-try-block ending in TEMX_CATCH.
#define TEMX_CATCH                                                                                     \
  catch (const std::bad_alloc&) { return fail(TEMX_ENOMEM, "out of host memory"); }                   \
  catch (const std::exception& e_) { return fail(TEMX_EINTERNAL, "unexpected C++ exception: %s", e_.what()); } \
  catch (...) { return fail(TEMX_EINTERNAL, "unexpected C++ exception"); }

extern "C" {

int temx_version(void) { return 401; }

const char* temx_last_error(void) { return g_err.c_str(); }

int temx_device_count(void) try {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
} TEMX_CATCH

void temx_plan_destroy(temx_plan* pl) {
  if (!pl) return;
  (void)hipSetDevice(pl->device);
  DevBuf* bufs[] = {&pl->x, &pl->Y0, &pl->yblk, &pl->yblk_w, &pl->Y0p, &pl->G, &pl->Ginv, &pl->norm,
                    &pl->flag, &pl->p, &pl->pg, &pl->lg, &pl->coslat, &pl->fcor, &pl->colscale,
                    &pl->B4, &pl->B3, &pl->C4, &pl->zb, &pl->partial, &pl->opB, &pl->opC,
                    &pl->Bq, &pl->Bq2, &pl->Ct, &pl->tz, &pl->rows, &pl->ysym, &pl->Bs, &pl->XB, &pl->P3};
  for (DevBuf* b : bufs) b->release();
  pl->crow.release();
  pl->ycls.release();
  pl->csum.release();
  pl->ccnt.release();
  pl->Pq.release();
  pl->csq.release();
  pl->Pq2.release();
  pl->ycls_l.release();
  pl->pbuf.release();
  pl->gblk.release();
  pl->ypblk.release();
  for (DevBuf* b : {&pl->T, &pl->Qp, &pl->GinvA, &pl->G2, &pl->xo, &pl->xc, &pl->ycx, &pl->ycx_s, &pl->crow_s, &pl->side_crow[0][0], &pl->side_crow[0][1], &pl->side_crow[1][0], &pl->side_crow[1][1],
                    &pl->side_gfirst[0][0], &pl->side_gfirst[0][1], &pl->side_gfirst[1][0], &pl->side_gfirst[1][1], &pl->rho,
                    &pl->rho0, &pl->gaunt, &pl->wq2, &pl->Axq, &pl->rho_t, &pl->Gx, &pl->Gsinv, &pl->Ax, &pl->Axs, &pl->oscblk, &pl->osAt, &pl->osAb, &pl->osAtq, &pl->osAbq, &pl->Bqp})
    b->release();
  for (auto& kv : pl->csplits_s) kv.second.release();
  for (auto& kv : pl->csplits) kv.second.release();
  for (int w = 0; w < 2; ++w)
    for (auto& tl : pl->timed[w]) {
      (void)hipEventDestroy(tl.a);
      (void)hipEventDestroy(tl.b);
    }
  delete pl;
}

// the basis kernels keep one row of K values per thread: 64 in registers, 512 (L <= 511) in scratch
#define TEMX_BASIS(kern, K, ...)                          \
  do {                                                    \
    if ((K) <= 64)                                        \
      hipLaunchKernelGGL(kern<64>, __VA_ARGS__);          \
    else                                                  \
      hipLaunchKernelGGL(kern<512>, __VA_ARGS__);         \
  } while (0)

// native rows: canonical [N][K] copy (Y0c, may be null) and the 4x4 blocks of the sweeps (yblk, may be null);
// T: null for Y0 itself, or the device copy of R^-1 for the rows of Q = Y0 R^-1
static int build_basis(temx_plan* pl, const double* rowscale_dev, const double* T, double* Y0c, double* yblk) {
  const int64_t npad = pl->nchunk * 16;
  TEMX_BASIS(basis_kernel, pl->K, dim3((unsigned)((npad + 255) / 256)), dim3(256), 0, 0, pl->x.d(), pl->N, npad, pl->K,
             pl->stride, pl->norm.d(), rowscale_dev, T, Y0c, yblk);
  HIPCHK(hipGetLastError());
  return TEMX_OK;
}

// rows at the output latitudes: canonical [M][K] into `dst`; for K <= 64 also the blocked copy of Qp
static int build_out_basis(temx_plan* pl, const double* T, double* dst, bool blocks) {
  const int64_t mch = (pl->M + 15) / 16;
  TEMX_BASIS(basis_kernel, pl->K, dim3((unsigned)((mch * 16 + 255) / 256)), dim3(256), 0, 0, pl->xo.d(), (int64_t)pl->M,
             mch * 16, pl->K, pl->stride, pl->norm.d(), (const double*)nullptr, T, dst, (double*)nullptr);
  HIPCHK(hipGetLastError());
  HIPCHK(hipDeviceSynchronize());
  if (blocks && pl->K <= 64) {   // blocked copy for solve_mfma_kernel
    std::vector<double> yp((size_t)pl->M * pl->K);
    HIPCHK(hipMemcpy(yp.data(), dst, yp.size() * 8, hipMemcpyDeviceToHost));
    return upload_blocks(pl->ypblk, yp.data(), pl->M, pl->K, pl->TB);
  }
  return TEMX_OK;
}

// basis rows at the class latitudes (kernels_cls.hpp): ycls, or the 64-harmonic slices ycls_l
static int build_cls_basis(temx_plan* pl, const double* T) {
  const unsigned nb = (unsigned)((pl->cls_npad + 255) / 256);
  if (pl->lcls) {
    for (int sl = 0; sl < pl->nslice; ++sl)
      TEMX_BASIS(cls_basis_slice_kernel, pl->K, dim3(nb), dim3(256), 0, 0, pl->xc.d(), pl->ncls, pl->cls_npad, pl->K,
                 64 * sl, pl->norm.d(), T, pl->ycls_l.d() + sl * pl->ycls_lstride);
  } else if (pl->cls) {
    TEMX_BASIS(cls_basis_kernel, pl->K, dim3(nb), dim3(256), 0, 0, pl->xc.d(), pl->ncls, pl->cls_npad, pl->K, pl->TBS,
               pl->norm.d(), T, pl->ycls.d());
  }
  HIPCHK(hipGetLastError());
  return TEMX_OK;
}

static int build_sym_basis(temx_plan* pl, const double* T) {
  if (!pl->sym) return TEMX_OK;
  const int64_t n4 = pl->npg_alloc * 4;
  TEMX_BASIS(sym_basis_kernel, pl->K, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, 0, pl->x.d(),
             static_cast<const int*>(pl->rows.p), pl->npair, n4, pl->K, pl->TBS, pl->norm.d(), T, pl->ysym.d());
  HIPCHK(hipGetLastError());
  return TEMX_OK;
}

// every projection / reconstruction operand of the plan in the basis T (null: Y0 itself)
static int build_all_bases(temx_plan* pl, const double* T) {
  int rc;
  if ((rc = build_basis(pl, nullptr, T, nullptr, pl->yblk.d()))) return rc;
  if ((rc = build_cls_basis(pl, T))) return rc;
  if ((rc = build_sym_basis(pl, T))) return rc;
  if ((rc = build_out_basis(pl, T, pl->Qp.d(), true))) return rc;
  HIPCHK(hipDeviceSynchronize());
  return TEMX_OK;
}

int temx_plan_create(temx_plan** out, int device, int64_t ncol, int L, int M,
                     const double* lat_deg_host, const double* lat_out_deg_host, int flags) try {
  if (!out || !lat_deg_host || !lat_out_deg_host) return fail(TEMX_EINVAL, "null argument");
  *out = nullptr;
  if (ncol < 1 || M < 1 || L < 0) return fail(TEMX_EINVAL, "ncol, M must be >= 1 and L >= 0");
  if (L > 511) return fail(TEMX_EUNSUPPORTED, "L = %d: this version supports L <= 511", L);
  int ndev = 0;
  HIPCHK(hipGetDeviceCount(&ndev));
  if (device < 0 || device >= ndev) return fail(TEMX_EHIP, "device %d not available (%d visible)", device, ndev);
  HIPCHK(hipSetDevice(device));
  hipDeviceProp_t prop;
  HIPCHK(hipGetDeviceProperties(&prop, device));
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
    return fail(TEMX_EUNSUPPORTED, "libtemx is built for gfx950 (MI355X); device is %s", prop.gcnArchName);

  temx_plan* pl = new temx_plan();
  pl->device = device;
  pl->no_qr = (flags & TEMX_NO_QR) != 0;
  const double tol_dflt = (flags & TEMX_LAT_TOL_F32) ? 1e-8 : 1e-11;   // degrees; see sym_tol_deg
  pl->num_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  pl->N = ncol;
  pl->nchunk = (ncol + 15) / 16;
  pl->L = L;
  pl->K = L + 1;
  // l-blocks of 4 harmonics per row; the sweeps are instantiated for TB in {4, 8, 13, 16}
  pl->TB = pl->K <= 16 ? 4 : (pl->K <= 32 ? 8 : (pl->K <= 52 ? 13 : 16));
  pl->stride = pl->TB;
  if (pl->K > 64) {   // several slices of 64 harmonics: sliced sweeps instead of the fused ones
    pl->large = true;
    pl->nslice = (pl->K + 63) / 64;
    pl->stride = 16 * pl->nslice;
  }
  pl->K4 = 4 * pl->stride;
  pl->M = M;
  pl->lat_out_deg.assign(lat_out_deg_host, lat_out_deg_host + M);
  int rc = TEMX_OK;
  auto bail = [&](int code) {
    temx_plan_destroy(pl);
    return code;
  };

  // x = cos(colat), colat = deg2rad(90 - lat)   (sph_zonal_mean.py:361)
  const double d2r = M_PI / 180.0;
  std::vector<double> xs((size_t)std::max<int64_t>(ncol, M));
  for (int64_t i = 0; i < ncol; ++i) xs[i] = std::cos((90.0 - lat_deg_host[i]) * d2r);
  if ((rc = upload(pl->x, xs.data(), (size_t)ncol * 8))) return bail(rc);
  std::vector<double> norm((size_t)std::max(64, pl->K4), 0.0);
  for (int l = 0; l < pl->K; ++l) norm[l] = std::sqrt((2.0 * l + 1.0) / (4.0 * M_PI));
  if ((rc = upload(pl->norm, norm.data(), norm.size() * 8))) return bail(rc);
  int zero = 0;
  if ((rc = upload(pl->flag, &zero, sizeof(int)))) return bail(rc);

  if ((rc = pl->Y0.ensure((size_t)ncol * pl->K * 8))) return bail(rc);
  // one extra chunk of blocks: the sweeps prefetch A operands one group / step ahead
  if ((rc = pl->yblk.ensure((size_t)(pl->nchunk + 1) * 4 * pl->stride * 16 * 8))) return bail(rc);
  if (hipMemset(pl->yblk.p, 0, pl->yblk.bytes) != hipSuccess) return bail(fail(TEMX_EHIP, "hipMemset of the Y0 blocks failed"));
  if ((rc = build_basis(pl, nullptr, nullptr, pl->Y0.d(), pl->yblk.d()))) return bail(rc);

  // Y0p on the output latitudes (sph_zonal_mean.py:367-370): same kernel, canonical copies only (Y0p the
  // attribute, Qp the device operand -- equal until temx_plan_finalize changes the basis)
  for (int m = 0; m < M; ++m) xs[m] = std::cos((90.0 - lat_out_deg_host[m]) * d2r);
  if ((rc = upload(pl->xo, xs.data(), (size_t)M * 8))) return bail(rc);
  if ((rc = pl->Y0p.ensure((size_t)M * pl->K * 8))) return bail(rc);
  if ((rc = pl->Qp.ensure((size_t)M * pl->K * 8))) return bail(rc);
  if ((rc = build_out_basis(pl, nullptr, pl->Y0p.d(), false))) return bail(rc);
  if ((rc = build_out_basis(pl, nullptr, pl->Qp.d(), true))) return bail(rc);

  // local Gram G = Y0^T Y0 through the projection sweep itself (A = Y0, D = K)
  {
    if ((rc = pl->G.ensure((size_t)pl->K * pl->K * 8))) return bail(rc);
    if ((rc = pl->Ginv.ensure((size_t)pl->K * pl->K * 8))) return bail(rc);
    Split sp = choose_split(pl->K, pl->nchunk, 2 * pl->num_cu);
    if ((rc = pl->partial.ensure((size_t)sp.nsplit * std::min(pl->K, 64) * pl->K * 8))) return bail(rc);
    FieldPtrs<1> fp;
    fp.p[0] = pl->Y0.p;
    if ((rc = project_all<1>(pl, fp, TEMX_F64, pl->K, nullptr, -1, sp, pl->G.d(), 0))) return bail(rc);
    hipError_t e = hipDeviceSynchronize();
    if (e != hipSuccess) return bail(fail(TEMX_EHIP, "gram kernel failed: %s", hipGetErrorString(e)));
  }
  {   // blocks of 4 even (or odd) harmonics in the paired / class sweeps
    const int nhalf = (pl->K + 1) / 2;                         // even harmonics (>= odd ones)
    const int tbs = (nhalf + 3) / 4;
    pl->TBS = tbs <= 2 ? 2 : (tbs <= 4 ? 4 : (tbs <= 7 ? 7 : 8));
  }
  // latitude classes (kernels_cls.hpp): columns that share |lat| share a basis row -- the MFMA work
  // is per class and the sweeps become HBM streams (cubed-sphere: 16 columns per class)
  {
    const char* e0 = getenv("TEMX_NO_SYM");
    const char* e1 = getenv("TEMX_NO_CLS");
    ClassTables ct;
    if ((!pl->large || pl->K <= 256) && !(flags & (TEMX_NO_SYMMETRY | TEMX_NO_CLASSES)) && !(e0 && e0[0] == '1') &&
        !(e1 && e1[0] == '1') && build_classes(lat_deg_host, ncol, ct, tol_dflt)) {
      if ((rc = upload(pl->crow, ct.crow.data(), ct.crow.size() * sizeof(int)))) return bail(rc);

      if ((rc = upload(pl->ccnt, ct.cnt.data(), ct.cnt.size() * 8))) return bail(rc);
      if ((rc = upload(pl->xc, ct.xc.data(), ct.xc.size() * 8))) return bail(rc);
      pl->h_xc = ct.xc;
      pl->h_cnt = ct.cnt;
      pl->h_crow = ct.crow;
      if (const char* dp = getenv("TEMX_DUMP_CROW")) {     // development: the row table as tools/sweep_lab.hip reads it (LAB_CROW)
        if (FILE* fh = fopen(dp, "wb")) {
          const int32_t hdr[2] = {(int32_t)ct.ngroups, (int32_t)ct.crow.size()};
          fwrite(hdr, 4, 2, fh);
          fwrite(ct.gbatch0.data(), 4, (size_t)ct.ngroups + 1, fh);
          fwrite(ct.crow.data(), 4, ct.crow.size(), fh);
          fclose(fh);
        }
      }
      pl->cls_npad = (ct.ngroups + 1) * 4;
      pl->gbatch0 = std::move(ct.gbatch0);
      pl->cgroups = ct.ngroups;
      pl->cbatches = ct.nbatch;
      pl->ncls = ct.ncls;
      if (pl->large) {   // class sums first, sliced basis at the class latitudes (kernels_cls.hpp)
        pl->ycls_lstride = (ct.ngroups + 1) * 256;
        if ((rc = pl->ycls_l.ensure((size_t)pl->nslice * pl->ycls_lstride * 8))) return bail(rc);
        pl->lcls = true;
      } else {
        if ((rc = pl->ycls.ensure((size_t)(ct.ngroups + 1) * 2 * pl->TBS * 16 * 8))) return bail(rc);
        pl->cls = true;
      }
      if ((rc = build_cls_basis(pl, nullptr))) return bail(rc);
      {
        hipError_t e2 = hipDeviceSynchronize();
        if (e2 != hipSuccess) return bail(fail(TEMX_EHIP, "class basis kernel failed: %s", hipGetErrorString(e2)));
      }
    }
  }
  // mirror pairing (kernels_sym.hpp): ~46 % fewer MFMAs on equatorially symmetric grids
  if (!pl->cls) {
    const char* e = getenv("TEMX_NO_SYM");
    std::vector<int> rN, rS;
    if (!pl->large && !(flags & TEMX_NO_SYMMETRY) && !(e && e[0] == '1') && find_mirror_pairs(lat_deg_host, ncol, rN, rS, tol_dflt)) {
      pl->npair = (int64_t)rN.size();
      pl->npg = (pl->npair + 3) / 4;
      pl->npg_alloc = ((pl->npg + SYM_PROJ_CH - 1) / SYM_PROJ_CH + 1) * SYM_PROJ_CH;   // whole chunks + 1 chunk
      const int64_t n4 = pl->npg_alloc * 4;
      std::vector<int> rows((size_t)2 * n4, 0);
      for (int64_t k = 0; k < n4; ++k) rows[(size_t)n4 + k] = -1;
      for (int64_t k = 0; k < pl->npair; ++k) {
        rows[(size_t)k] = rN[(size_t)k];
        rows[(size_t)n4 + k] = rS[(size_t)k];
      }
      if ((rc = upload(pl->rows, rows.data(), rows.size() * sizeof(int)))) return bail(rc);
      if ((rc = pl->ysym.ensure((size_t)pl->npg_alloc * 2 * pl->TBS * 16 * 8))) return bail(rc);
      pl->sym = true;
      if ((rc = build_sym_basis(pl, nullptr))) return bail(rc);
      hipError_t e2 = hipDeviceSynchronize();
      if (e2 != hipSuccess) return bail(fail(TEMX_EHIP, "sym basis kernel failed: %s", hipGetErrorString(e2)));
    }
  }
  if (!(flags & TEMX_DEFER_FINALIZE)) {
    if ((rc = temx_plan_finalize(pl, nullptr))) return bail(rc);
  }
  *out = pl;
  return TEMX_OK;
} TEMX_CATCH

int temx_plan_is_paired(const temx_plan* pl) { return pl && (pl->sym || pl->cls) ? 1 : 0; }

int temx_plan_sweep_mode(const temx_plan* pl) try {
  return !pl ? -1 : ((pl->cls || (pl->lcls && pl->lone)) ? 2 : (pl->sym ? 1 : 0));
} TEMX_CATCH

int temx_plan_one_pass(const temx_plan* pl) try {
  return pl && ((pl->cls && pl->onepass) || (pl->lcls && pl->lone)) ? 1 : 0;
} TEMX_CATCH

int temx_plan_single_sweep(const temx_plan* pl) { return pl && pl->os_on ? 1 : 0; }

// G2 (device) = Q^T Q over this plan's rows, through the projection sweep (A = Q, D = K)
static int gram_of_q(temx_plan* pl) {
  DevBuf Qc;
  int rc = Qc.ensure((size_t)pl->N * pl->K * 8);
  if (rc) return rc;
  if ((rc = pl->G2.ensure((size_t)pl->K * pl->K * 8)) == TEMX_OK &&
      (rc = build_basis(pl, nullptr, pl->T.d(), Qc.d(), nullptr)) == TEMX_OK) {
    Split sp = choose_split(pl->K, pl->nchunk, 2 * pl->num_cu);
    FieldPtrs<1> fp;
    fp.p[0] = Qc.p;
    rc = project_all<1>(pl, fp, TEMX_F64, pl->K, nullptr, -1, sp, pl->G2.d(), 0);
  }
  hipError_t e = hipDeviceSynchronize();
  Qc.release();
  if (rc) return rc;
  if (e != hipSuccess) return fail(TEMX_EHIP, "gram kernel failed: %s", hipGetErrorString(e));
  return TEMX_OK;
}

// Replaces lstsq(Y0, I_N) (sph_zonal_mean.py:389).  The Gram matrix G = Y0^T Y0 squares the condition
// number of Y0, and multiplying by an explicit G^-1 squares it once more (measured on a random grid with
// cond(G) = 5.9e3: 1.4e-9 from the reference's SVD solution, profiles/r02_fuzz_seed_887.log).  So the plan
// re-orthogonalises (Cholesky-QR2): R from the Cholesky factorisation of G (long double), every basis block
// of the sweeps rebuilt for Q = Y0 R^-1 -- a row of Q still depends on latitude only, and with an
// equatorially symmetric grid R does not mix even and odd harmonics, so the class / paired sweeps keep
// working -- then the Gram matrix of Q (the identity up to cond(G) eps) factorised once more
// (temx_plan_refine).  The sweeps project on Q, the K x K "solve" multiplies by (Q^T Q)^-1 ~ I, and Qp = Y0p
// R^-1 takes the coefficients to the output latitudes: errors of order cond(Y0) eps.  Attributes (Y0, Y0p,
// Y0inv = G^-1 Y0^T) are unchanged.  TEMX_NO_QR=1 keeps the plain normal equations (A/B runs).
int temx_plan_finalize(temx_plan* pl, const double* G_host) try {
  if (!pl) return fail(TEMX_EINVAL, "null plan");
  HIPCHK(hipSetDevice(pl->device));
  const int K = pl->K;
  std::vector<double> G((size_t)K * K), Gi((size_t)K * K);
  if (G_host) {
    std::copy(G_host, G_host + (size_t)K * K, G.begin());
    HIPCHK(hipMemcpy(pl->G.p, G.data(), G.size() * 8, hipMemcpyHostToDevice));
  } else {
    HIPCHK(hipMemcpy(G.data(), pl->G.p, G.size() * 8, hipMemcpyDeviceToHost));
  }
  for (double v : G)
    if (!std::isfinite(v)) return fail(TEMX_EINVAL, "Gram matrix is not finite (NaN latitudes?)");
  pl->h_G = G;                    // (as on the device: before the odd entries are cleared)
  if (pl->qbasis) {               // finalised before: back to the Y0 basis the Gram matrix refers to
    if (int rcb = build_all_bases(pl, nullptr)) return rcb;
    pl->qbasis = false;
  }
  // On an equatorially symmetric grid G[l][m] vanishes for l + m odd; the entries that rounding left
  // there are cleared, so that R (and Q) keep the parity the class / paired sweeps rely on.
  const bool parity_paths = pl->sym || pl->cls || pl->lcls;
  double odd_max = 0.0, diag_max = 0.0;
  for (int i = 0; i < K; ++i) {
    diag_max = std::max(diag_max, std::fabs(G[(size_t)i * K + i]));
    for (int j = 0; j < K; ++j)
      if ((i + j) & 1) odd_max = std::max(odd_max, std::fabs(G[(size_t)i * K + j]));
  }
  const bool checker = odd_max <= 1e-10 * diag_max;
  pl->ext_G = G_host != nullptr;
  pl->g_checker = checker && (G_host != nullptr || parity_paths);
  if (pl->g_checker)
    for (int i = 0; i < K; ++i)
      for (int j = 0; j < K; ++j)
        if ((i + j) & 1) G[(size_t)i * K + j] = 0.0;
  std::vector<long double> Li;
  const bool spd = spd_factor(G.data(), K, Li) == 0;
  if (!spd) {
    // rank-deficient Y0: pseudo-inverse, like the reference's lstsq (sph_zonal_mean.py:389)
    int rank = 0;
    if (sym_pinv(G.data(), K, Gi.data(), &rank) != 0 || rank == 0)
      return fail(TEMX_ERANK, "Y0^T Y0 has no positive eigenvalue (N=%lld, K=%d)", (long long)pl->N, K);
    pl->rank = rank;
  } else {
    inverse_from_factor(Li, K, Gi.data());
    pl->rank = K;
  }
  if (int rca = upload(pl->GinvA, Gi.data(), Gi.size() * 8)) return rca;
  const char* eq = getenv("TEMX_NO_QR");
  const bool no_qr = eq ? eq[0] == '1' : pl->no_qr;
  // Q = Y0 R^-1 keeps the parity of the harmonics only when G is a checkerboard (an equatorially symmetric grid);
  // the class / paired sweeps need that.  With an external G (ncol-sharded: the all-reduced one) every rank must
  // make the SAME choice whatever sweeps its own block of columns runs, so the choice depends on G alone there.
  const bool want_q = spd && !no_qr && (G_host ? checker : (!parity_paths || checker));
  if (want_q) {
    std::vector<double> T((size_t)K * K, 0.0), I((size_t)K * K, 0.0);
    for (int l = 0; l < K; ++l) {
      I[(size_t)l * K + l] = 1.0;
      for (int j = l; j < K; ++j) T[(size_t)l * K + j] = (double)Li[(size_t)j * K + l];   // R^-1 = (L^-1)^T
    }
    if (int rct = upload(pl->T, T.data(), T.size() * 8)) return rct;
    pl->h_T = T;
    if (int rcb = build_all_bases(pl, pl->T.d())) return rcb;
    pl->qbasis = true;
    if (int rcg = set_ginv(pl, I.data())) return rcg;      // until temx_plan_refine
  } else {
    if (int rcg = set_ginv(pl, Gi.data())) return rcg;
  }
  int zero = 0;
  HIPCHK(hipMemcpy(pl->flag.p, &zero, sizeof(int), hipMemcpyHostToDevice));
  pl->finalized = true;
  pl->op_valid = pl->c4_valid = pl->xb_valid = pl->tq_valid = false;   // sums / coefficients of another basis
  // one process owns all the rows: second pass of the re-orthogonalisation with its own Gram matrix of Q
  if (want_q && !G_host) return temx_plan_refine(pl, nullptr);
  return TEMX_OK;
} TEMX_CATCH

int temx_plan_refine(temx_plan* pl, const double* G2_host) try {
  if (!pl) return fail(TEMX_EINVAL, "null plan");
  if (!pl->finalized) return fail(TEMX_ESTATE, "plan not finalised");
  if (!pl->qbasis) return TEMX_OK;                 // normal equations / pseudo-inverse / weights: nothing to refine
  HIPCHK(hipSetDevice(pl->device));
  const int K = pl->K;
  std::vector<double> G2((size_t)K * K), Gi((size_t)K * K);
  if (G2_host) {
    std::copy(G2_host, G2_host + (size_t)K * K, G2.begin());
  } else {
    if (int rc = gram_of_q(pl)) return rc;
    HIPCHK(hipMemcpy(G2.data(), pl->G2.p, G2.size() * 8, hipMemcpyDeviceToHost));
  }
  for (double v : G2)
    if (!std::isfinite(v)) return fail(TEMX_EINVAL, "second Gram matrix is not finite");
  if (pl->g_checker)                               // same parity argument as in temx_plan_finalize
    for (int i = 0; i < K; ++i)
      for (int j = 0; j < K; ++j)
        if ((i + j) & 1) G2[(size_t)i * K + j] = 0.0;
  std::vector<long double> Li;
  if (spd_factor(G2.data(), K, Li) != 0) return TEMX_OK;   // (cannot happen for Q^T Q ~ I; keep the identity)
  inverse_from_factor(Li, K, Gi.data());
  return set_ginv(pl, Gi.data());
} TEMX_CATCH

int temx_plan_set_weights(temx_plan* pl, const double* w_host) try {
  if (!pl || !w_host) return fail(TEMX_EINVAL, "null argument");
  HIPCHK(hipSetDevice(pl->device));
  // Y0inv = Y0^T diag(4 pi w)  (sph_zonal_mean.py:181, 385): scale the projection operand rows,
  // and make the "Gram inverse" the identity.
  std::vector<double> w((size_t)pl->N);
  for (int64_t i = 0; i < pl->N; ++i) w[i] = w_host[i] * 4.0 * M_PI;
  if (pl->qbasis) {               // finalised before: the weighted operator works in the Y0 basis
    if (int rcb = build_all_bases(pl, nullptr)) return rcb;
    pl->qbasis = false;
  }
  DevBuf wd;
  int rc = upload(wd, w.data(), w.size() * 8);
  if (rc) return rc;
  rc = pl->yblk_w.ensure(pl->yblk.bytes);
  if (rc) {
    wd.release();
    return rc;
  }
  (void)hipMemset(pl->yblk_w.p, 0, pl->yblk_w.bytes);
  rc = build_basis(pl, wd.d(), nullptr, nullptr, pl->yblk_w.d());
  hipError_t e = hipDeviceSynchronize();
  wd.release();
  if (rc) return rc;
  if (e != hipSuccess) return fail(TEMX_EHIP, "basis kernel failed: %s", hipGetErrorString(e));
  std::vector<double> I((size_t)pl->K * pl->K, 0.0);
  for (int k = 0; k < pl->K; ++k) I[(size_t)k * pl->K + k] = 1.0;
  if (int rcg = set_ginv(pl, I.data())) return rcg;
  if (int rca = upload(pl->GinvA, I.data(), I.size() * 8)) return rca;
  // weighted rows of one latitude no longer share a basis row: every latitude-structured path is off
  // (the large-L class path too: its class basis ycls_l is unweighted)
  pl->sym = pl->cls = pl->lcls = false;
  pl->lone = pl->onepass = pl->op_valid = pl->xb_valid = false;
  pl->tem = false;                // splits / workspaces belong to the path: set_tem again
  pl->weighted = true;
  pl->finalized = true;
  return TEMX_OK;
} TEMX_CATCH

int temx_get_matrix(temx_plan* pl, int which, double* dst, void* stream) try {
  if (!pl || !dst) return fail(TEMX_EINVAL, "null argument");
  HIPCHK(hipSetDevice(pl->device));
  hipStream_t st = S_(stream);
  const size_t KK = (size_t)pl->K * pl->K * 8;
  switch (which) {
    case TEMX_MAT_Y0:
      HIPCHK(hipMemcpyAsync(dst, pl->Y0.p, (size_t)pl->N * pl->K * 8, hipMemcpyDeviceToDevice, st));
      return TEMX_OK;
    case TEMX_MAT_Y0P:
      HIPCHK(hipMemcpyAsync(dst, pl->Y0p.p, (size_t)pl->M * pl->K * 8, hipMemcpyDeviceToDevice, st));
      return TEMX_OK;
    case TEMX_MAT_GRAM:
      HIPCHK(hipMemcpyAsync(dst, pl->G.p, KK, hipMemcpyDeviceToDevice, st));
      return TEMX_OK;
    case TEMX_MAT_GINV:
      if (!pl->finalized) return fail(TEMX_ESTATE, "plan not finalised");
      HIPCHK(hipMemcpyAsync(dst, pl->GinvA.p, KK, hipMemcpyDeviceToDevice, st));
      return TEMX_OK;
    case TEMX_MAT_Y0INV:
      if (!pl->finalized) return fail(TEMX_ESTATE, "plan not finalised");
      hipLaunchKernelGGL(y0inv_kernel, dim3((unsigned)((pl->N + 255) / 256)), dim3(256), 0, st, pl->Y0.d(),
                         pl->GinvA.d(), pl->N, pl->K, dst);
      HIPCHK(hipGetLastError());
      return TEMX_OK;
    case TEMX_MAT_GRAM2: {
      // this rank's share of Q^T Q (ncol-sharded callers all-reduce it and hand it to temx_plan_refine);
      // the identity when the plan does not run on a re-orthogonalised basis
      if (!pl->finalized) return fail(TEMX_ESTATE, "plan not finalised");
      if (!pl->qbasis) {
        std::vector<double> I((size_t)pl->K * pl->K, 0.0);
        for (int k = 0; k < pl->K; ++k) I[(size_t)k * pl->K + k] = 1.0;
        HIPCHK(hipStreamSynchronize(st));
        HIPCHK(hipMemcpy(dst, I.data(), KK, hipMemcpyHostToDevice));
        return TEMX_OK;
      }
      if (int rc2 = gram_of_q(pl)) return rc2;
      HIPCHK(hipMemcpyAsync(dst, pl->G2.p, KK, hipMemcpyDeviceToDevice, st));
      return TEMX_OK;
    }
    case TEMX_MAT_GX:
    case TEMX_MAT_GSUB: {
      if (!pl->os_built) return fail(TEMX_ESTATE, "the single-sweep tables are not built (temx_plan_set_tem on a plan that takes that form)");
      const std::vector<double>& h = which == TEMX_MAT_GX ? pl->h_Gx : pl->h_Gs;
      HIPCHK(hipStreamSynchronize(st));
      HIPCHK(hipMemcpy(dst, h.data(), h.size() * 8, hipMemcpyHostToDevice));
      return TEMX_OK;
    }
    default:
      return fail(TEMX_EINVAL, "unknown matrix id %d", which);
  }
} TEMX_CATCH

// ncol-sharded single sweep: the two matrices of build_os_tables that sum over the rows -- Gx = Y0^T Y0ext [K][2L+1]
// and the Gram matrix of the reference subsample [KR][KR] -- summed over the ranks (all-reduce of
// temx_get_matrix(TEMX_MAT_GX / TEMX_MAT_GSUB)), handed back.
int temx_plan_set_os_matrices(temx_plan* pl, const double* Gx_host, const double* Gs_host) try {
  if (!pl || !Gx_host || !Gs_host) return fail(TEMX_EINVAL, "null argument");
  if (!pl->os_built) return fail(TEMX_ESTATE, "the single-sweep tables are not built");
  HIPCHK(hipSetDevice(pl->device));
  const int KR = pl->KR;
  std::vector<long double> Li;
  std::vector<double> Gi((size_t)KR * KR);
  if (spd_factor(Gs_host, KR, Li) != 0)
    return fail(TEMX_ERANK, "the subsample of latitude classes does not determine a degree-%d reference", KR - 1);
  inverse_from_factor(Li, KR, Gi.data());
  HIPCHK(hipDeviceSynchronize());
  if (int rc = upload(pl->Gsinv, Gi.data(), Gi.size() * 8)) return rc;
  if (int rc = upload(pl->Gx, Gx_host, (size_t)pl->K * pl->KX * 8)) return rc;
  pl->h_Gx.assign(Gx_host, Gx_host + (size_t)pl->K * pl->KX);
  if (int rc = os_upload_blocks(pl)) return rc;
  pl->os_need_global = false;
  pl->os_valid = false;
  return TEMX_OK;
} TEMX_CATCH

// ---- operator API --------------------------------------------------------------------------------
int temx_project(temx_plan* pl, const void* A, int dtype, int64_t D, double* B, void* stream) try {
  if (!pl || !A || !B) return fail(TEMX_EINVAL, "null argument");
  if (D < 1 || D >= ((int64_t)1 << 28)) return fail(TEMX_EINVAL, "D must be in [1, 2^28)");
  HIPCHK(hipSetDevice(pl->device));
  FieldPtrs<1> fp;
  fp.p[0] = A;
  if (pl->cls) {      // class sweep: one basis row per latitude class
    Split spc = choose_split(D, std::max<int64_t>(1, pl->cbatches / 4), CLS_PROJ_E_WPS * pl->num_cu, 4);
    int rcc = pl->partial.ensure((size_t)spc.nsplit * pl->K * D * 8);
    if (rcc) return rcc;
    if ((rcc = launch_project_cls<1>(pl, fp, dtype, D, nullptr, -1, pl->partial.d(), spc, S_(stream)))) return rcc;
    return launch_reduce(pl, pl->partial.d(), spc.nsplit, (int64_t)pl->K * D, B, S_(stream));
  }
  Split sp = choose_split(D, pl->nchunk, 2 * pl->num_cu);
  int rc = pl->partial.ensure((size_t)sp.nsplit * std::min(pl->K, 64) * D * 8);
  if (rc) return rc;
  return project_all<1>(pl, fp, dtype, D, nullptr, -1, sp, B, S_(stream));
} TEMX_CATCH

int temx_zonal_mean_from_sums(temx_plan* pl, const double* B, int64_t D, double* out, int native,
                              void* stream) try {
  if (!pl || !B || !out) return fail(TEMX_EINVAL, "null argument");
  if (!pl->finalized) return fail(TEMX_ESTATE, "plan not finalised");
  HIPCHK(hipSetDevice(pl->device));
  int rc;
  if (!native) return launch_solve(pl, B, 1, D, nullptr, out, S_(stream));
  if ((rc = pl->opC.ensure((size_t)pl->K4 * D * 8))) return rc;
  if ((rc = launch_solve(pl, B, 1, D, pl->opC.d(), nullptr, S_(stream)))) return rc;
  return launch_recon(pl, D, pl->opC.d(), out, S_(stream));
} TEMX_CATCH

int temx_zonal_mean(temx_plan* pl, const void* A, int dtype, int64_t D, double* out, int native,
                    void* stream) try {
  if (!pl || !A || !out) return fail(TEMX_EINVAL, "null argument");
  if (!pl->finalized) return fail(TEMX_ESTATE, "plan not finalised");
  int rc = pl->opB.ensure((size_t)pl->K * std::max<int64_t>(D, 1) * 8);
  if (rc) return rc;
  if ((rc = temx_project(pl, A, dtype, D, pl->opB.d(), stream))) return rc;
  return temx_zonal_mean_from_sums(pl, pl->opB.d(), D, out, native, stream);
} TEMX_CATCH

// ---- path selection: temx_plan_configure sets the options, the TEMX_* environment variables override them ------
struct FormChoice {
  bool two_pass;      // the class path in its two-pass form
  bool force_op;      // the one-pass form wherever it is possible (lifts the size threshold)
  int os;             // single sweep: 0 never, 1 wherever the instantiations exist, -1 automatic
};
static FormChoice form_choice(const temx_plan* pl) {
  FormChoice c{false, false, -1};
  switch (pl->opt_form) {
    case TEMX_FORM_TWO_PASS: c.two_pass = true; c.os = 0; break;
    case TEMX_FORM_CLASS_SUMS: c.force_op = true; c.os = 0; break;
    case TEMX_FORM_SINGLE_SWEEP: c.force_op = true; c.os = 1; break;
    case TEMX_FORM_NO_SINGLE_SWEEP: c.os = 0; break;
    default: break;
  }
  if (const char* e = getenv("TEMX_TWO_PASS")) c.two_pass = e[0] == '1';
  if (const char* e = getenv("TEMX_ONE_PASS")) c.force_op = c.force_op || e[0] == '1';
  if (const char* e = getenv("TEMX_SINGLE_SWEEP")) c.os = e[0] == '0' ? 0 : (e[0] == '1' ? 1 : c.os);
  return c;
}
// 1: loads of 4 rows x 16 columns (the MFMA tile), 0: loads of 1 row x 64 columns
static bool tile_map(int opt, const char* env) {
  if (const char* e = getenv(env)) return !strcmp(e, "tile");
  return opt == 1;
}
static bool tracer_one_pass_wanted(const temx_plan* pl) {
  if (const char* e = getenv("TEMX_TRACER_ONE_PASS")) return e[0] == '1';
  return pl->opt_tracer_one_pass == 1;
}

int temx_plan_configure(temx_plan* pl, int option, int value) try {
  if (!pl) return fail(TEMX_EINVAL, "null plan");
  switch (option) {
    case TEMX_OPT_FORM:
      if (value < -1 || value > TEMX_FORM_NO_SINGLE_SWEEP) return fail(TEMX_EINVAL, "TEMX_OPT_FORM: unknown form %d", value);
      pl->opt_form = value;
      break;
    case TEMX_OPT_OS_MAP: pl->opt_os_map = value; break;
    case TEMX_OPT_OP_MAP: pl->opt_op_map = value; break;
    case TEMX_OPT_TRACER_ONE_PASS: pl->opt_tracer_one_pass = value; break;
    case TEMX_OPT_OS_SUBSAMPLE:
      if (value < 4) return fail(TEMX_EINVAL, "TEMX_OPT_OS_SUBSAMPLE: at least 4 class-groups");
      if (pl->os_built && value != pl->os_keep) return fail(TEMX_ESTATE, "the single-sweep tables of this plan are built: set the subsample before temx_plan_set_tem");
      pl->os_keep = value;
      break;
    case TEMX_OPT_SINGLE_SWEEP_MIN_GROUPS: pl->opt_single_sweep_min_groups = value; break;
    case TEMX_OPT_OS_CONTRACT: pl->opt_os_contract = value; break;
    default: return fail(TEMX_EINVAL, "unknown option %d", option);
  }
  pl->tem = false;                // the choice is made in temx_plan_set_tem: call it (again)
  return TEMX_OK;
} TEMX_CATCH

int temx_plan_option(const temx_plan* pl, int option) try {
  if (!pl) return -1;
  switch (option) {
    case TEMX_OPT_FORM: return pl->os_on ? TEMX_FORM_SINGLE_SWEEP : ((pl->cls && pl->onepass) || (pl->lcls && pl->lone) ? TEMX_FORM_CLASS_SUMS : TEMX_FORM_TWO_PASS);
    case TEMX_OPT_OS_MAP: return tile_map(pl->opt_os_map, "TEMX_OS_MAP") ? 1 : 0;
    case TEMX_OPT_OP_MAP: return tile_map(pl->opt_op_map, "TEMX_OP_MAP") ? 1 : 0;
    case TEMX_OPT_TRACER_ONE_PASS: return tracer_one_pass_wanted(pl) ? 1 : 0;
    case TEMX_OPT_OS_SUBSAMPLE: return pl->os_keep;
    case TEMX_OPT_SINGLE_SWEEP_MIN_GROUPS: return pl->opt_single_sweep_min_groups;
    case TEMX_OPT_OS_CONTRACT: return pl->osc_lds ? 1 : 0;
    default: return -1;
  }
} TEMX_CATCH

// ---- TEM pipeline --------------------------------------------------------------------------------
int temx_plan_set_tem(temx_plan* pl, int nlev, int64_t nt, const double* p_pa_host, double p0) try {
  if (!pl || !p_pa_host) return fail(TEMX_EINVAL, "null argument");
  if (nlev < 2 || nt < 1) return fail(TEMX_EINVAL, "need nlev >= 2 and nt >= 1");
  if ((int64_t)nlev * nt >= ((int64_t)1 << 28)) return fail(TEMX_EINVAL, "nlev*nt must be < 2^28");
  if (pl->M < 2) return fail(TEMX_EINVAL, "need at least 2 zonal-mean latitudes");
  HIPCHK(hipSetDevice(pl->device));
  std::vector<double> p(p_pa_host, p_pa_host + nlev), tab;
  for (int j = 1; j < nlev; ++j)
    if (!(p[j] > p[j - 1])) return fail(TEMX_EINVAL, "pressure must be strictly ascending (front end flips)");
  // a failure below must not leave an earlier configuration half replaced: the plan is unconfigured
  // (tem_ready fails) until the last allocation has succeeded
  pl->tem = false;
  pl->onepass = pl->lone = pl->op_valid = pl->xb_valid = pl->tq_valid = false;
  pl->os_on = pl->os_valid = false;
  pl->os_tile = tile_map(pl->opt_os_map, "TEMX_OS_MAP");
  {
    const char* e = getenv("TEMX_OS_CONTRACT");
    pl->osc_lds = e ? !strcmp(e, "lds") : pl->opt_os_contract == 1;
  }
  pl->op_tile = tile_map(pl->opt_op_map, "TEMX_OP_MAP");
  pl->nlev = nlev;
  pl->nt = nt;
  pl->D = (int64_t)nlev * nt;
  pl->tD = pl->D;
  pl->tnt = nt;
  pl->tt0 = 0;
  pl->p0 = p0;
  const int M = pl->M;
  const int64_t D = pl->D;
  int rc;
  if ((rc = upload(pl->p, p.data(), (size_t)nlev * 8))) return rc;
  gradient_table(p, tab);
  if ((rc = upload(pl->pg, tab.data(), tab.size() * 8))) return rc;
  // latitude tables: f and cos(lat) use lat*pi/180 (tem_diagnostics.py:401-402), the gradient
  // uses np.deg2rad(lat) = lat*(pi/180) (:586)
  std::vector<double> latr(M), cosl(M), fc(M);
  for (int m = 0; m < M; ++m) {
    const double lat = pl->lat_out_deg[m];
    latr[m] = lat * (M_PI / 180.0);
    cosl[m] = std::cos(lat * M_PI / 180.0);
    fc[m] = 2 * kOm * std::sin(lat * M_PI / 180.0);
  }
  gradient_table(latr, tab);
  if ((rc = upload(pl->lg, tab.data(), tab.size() * 8))) return rc;
  if ((rc = upload(pl->coslat, cosl.data(), (size_t)M * 8))) return rc;
  if ((rc = upload(pl->fcor, fc.data(), (size_t)M * 8))) return rc;
  // theta = T (p0/p)^k, k = R/Cp  (tem_diagnostics.py:498, constants.py:12): per-column scale
  std::vector<double> cs((size_t)D);
  const double kap = kR / kCp;
  for (int j = 0; j < nlev; ++j) {
    const double s = std::pow(p0 / p[j], kap);
    for (int64_t t = 0; t < nt; ++t) cs[(size_t)j * nt + t] = s;
  }
  if ((rc = upload(pl->colscale, cs.data(), cs.size() * 8))) return rc;

  const int ndt_ = (int)((D + 15) / 16);
  const int pdpw = proj_dpw(4, ndt_);
  pl->sp_proj4 = choose_split(D, pl->nchunk, proj_wps(4, pdpw) * pl->num_cu, pdpw);
  // eddy sweep: one 8-wave workgroup per CU (LDS slabs) owning dpw d-tiles; the 8/dpw waves on a
  // d-tile split its chunk range -> 8/dpw partial slabs per split
  const int edpw = pick_dpw(ndt_, 4);
  pl->sp_eddy = choose_split(D, pl->nchunk / (8 / edpw), pl->num_cu, edpw);
  const size_t need =
      (size_t)std::max(pl->sp_proj4.nsplit * 4, pl->sp_eddy.nsplit * (8 / edpw) * 3) * pl->K * D * 8;
  if ((rc = pl->partial.ensure(need))) return rc;
  if ((rc = pl->B4.ensure((size_t)4 * pl->K * D * 8))) return rc;
  if ((rc = pl->B3.ensure((size_t)3 * pl->K * D * 8))) return rc;
  if ((rc = pl->C4.ensure((size_t)4 * pl->K4 * D * 8))) return rc;
  if ((rc = pl->zb.ensure((size_t)8 * M * D * 8))) return rc;
  pl->sp_proj1 = choose_split(D, pl->nchunk, 2 * pl->num_cu);
  if (unfused_stage2(pl)) {   // products are projected three at a time, 64 harmonics per pass
    const size_t need3 = (size_t)pl->sp_proj1.nsplit * 3 * 64 * D * 8;
    if ((rc = pl->partial.ensure(std::max(need3, pl->partial.bytes)))) return rc;
  }
  pl->lone = false;
  pl->xb_valid = false;
  pl->op_valid = false;      // class sums of an earlier configuration are void
  if (pl->lcls) {    // large-L class path: class sums first (kernels_cls.hpp), if they fit
    const size_t need_cs = (size_t)pl->cgroups * ndt_ * 14 * 64 * 8, need_pb = (size_t)pl->cgroups * ndt_ * 3 * 128 * 8;
    size_t fr = 0, tot = 0;
    if (ndt_ >= 4 && !form_choice(pl).two_pass && hipMemGetInfo(&fr, &tot) == hipSuccess &&
        (pl->csum.bytes >= need_cs || need_cs + need_pb < fr / 2) && alloc_write_stream(pl->csum, need_cs) == TEMX_OK &&
        pl->pbuf.ensure(need_pb) == TEMX_OK) {
      HIPCHK(hipMemset(pl->csum.p, 0, pl->csum.bytes));
      HIPCHK(hipMemset(pl->pbuf.p, 0, pl->pbuf.bytes));
      const int64_t cunits = std::max<int64_t>(1, pl->cbatches / 4);
      pl->sp_cproj4 = choose_split(D, cunits, TEMX_CLS_OP_WPS * pl->num_cu, 4, 8);
      pl->sp_cflux = choose_split(D, std::max<int64_t>(1, pl->cgroups / (8 / edpw)), pl->num_cu, edpw, TEMX_CLS_MINCHUNK);
      pl->sp_lflux = choose_split(D, std::max<int64_t>(1, pl->cgroups / 8), pl->num_cu, 1, TEMX_CLS_MINCHUNK);
      const size_t need4 = (size_t)pl->sp_cflux.nsplit * 4 * 64 * D * 8;
      if ((rc = pl->partial.ensure(std::max(need4, pl->partial.bytes)))) return rc;
      if ((rc = pl->Bs.ensure((size_t)4 * 64 * D * 8))) return rc;
      const int2* cuts_unused = nullptr;
      if ((rc = class_cuts(pl, pl->sp_cproj4.nsplit, &cuts_unused, true))) return rc;
      pl->lone = true;
    }
  }
  if (pl->cls) {
    const bool quad = pick_dpw(ndt_, 4) == 4;
    const int64_t cunits = std::max<int64_t>(1, pl->cbatches / 4);   // work units of ~4 batches (one cubed-sphere class-group)
    pl->sp_cproj4 = choose_split(D, cunits, (quad ? 2 : CLS_PROJ_E_WPS) * pl->num_cu, quad ? 4 : 1, TEMX_CLS_MINCHUNK);
    // one-pass form of sweep 1: quads of d-tiles, pieces of >= 8 class-groups (its cuts are group aligned)
    Split sp_op = choose_split(D, cunits, TEMX_CLS_OP_WPS * pl->num_cu, 4, 8);
    pl->sp_cproj1 = choose_split(D, cunits, CLS_PROJ_E_WPS * pl->num_cu, 4, TEMX_CLS_MINCHUNK);
    pl->sp_ceddy = choose_split(D, cunits / (8 / edpw), pl->num_cu, edpw, TEMX_CLS_MINCHUNK);
    const size_t need3 = (size_t)std::max({pl->sp_cproj4.nsplit * 4, pl->sp_ceddy.nsplit * (8 / edpw) * 3,
                                           pl->sp_cproj1.nsplit}) * pl->K * D * 8;
    if ((rc = pl->partial.ensure(std::max(need3, pl->partial.bytes)))) return rc;
    // one-pass form: quads of d-tiles, enough class-groups per piece for group-aligned cuts to balance,
    // and room for the class sums (8 x 512 B per class-group and d-tile)
    pl->onepass = false;
    pl->op_valid = false;
    {
      const FormChoice fc = form_choice(pl);
      const bool force = fc.force_op;             // whenever possible (tests)
      // a ragged last quad only idles a few waves.  The one-pass sweep runs one wave per SIMD, which
      // pays once there is enough work: measured break-even near 1.2e7 elements per field
      // (ne30x72x2, 7e6: 0.137 ms two-pass vs 0.156; ne30x72x4, 1.4e7: 0.223 vs 0.203;
      // ne120x72x2, 1.1e8: 1.44 vs 1.18)
      const bool quad_op = ndt_ >= 4 && (force || (double)pl->N * (double)D >= 1.2e7);
      if (quad_op && !fc.two_pass) {
        const size_t need_cs = (size_t)pl->cgroups * ndt_ * 8 * 64 * 8;   // 4 {north, south} pairs per lane
        size_t fr = 0, tot = 0;
        bool have = pl->csum.bytes >= need_cs;
        if (!have && hipMemGetInfo(&fr, &tot) == hipSuccess && need_cs < fr / 2 && alloc_write_stream(pl->csum, need_cs) == TEMX_OK) {
          // lanes beyond a ragged last d-tile are never written but are read (and ignored) by the flux kernel
          HIPCHK(hipMemset(pl->csum.p, 0, pl->csum.bytes));
          have = true;
        }
        if (have) {
          pl->onepass = true;
          pl->sp_cproj4 = sp_op;
          pl->sp_cflux = choose_split(D, std::max<int64_t>(1, pl->cgroups / (8 / edpw)), pl->num_cu, edpw, TEMX_CLS_MINCHUNK);
          const size_t need4 = (size_t)std::max(pl->sp_cflux.nsplit * 3, sp_op.nsplit * 7) * pl->K * D * 8;
          if ((rc = pl->partial.ensure(std::max(need4, pl->partial.bytes)))) return rc;
          if ((rc = pl->Pq.ensure((size_t)3 * pl->K * D * 8))) return rc;
          // TEM + one tracer in one sweep: the four waves of a workgroup share a d-tile (one workgroup per CU)
          pl->sp_copw = choose_split(D, std::max<int64_t>(1, cunits / 4), pl->num_cu, 1, 8);
          // single-sweep form of temx_tem_run (no class-sum stream), the default where the one-pass path runs
          // and the grid has enough latitude classes for the reference fit; TEMX_SINGLE_SWEEP=0 keeps the
          // class-sum form, =1 also takes it on small grids
          {
            const bool off = fc.os == 0, forced = fc.os == 1;
            // Both forms cost in proportion to D; the class-sum form also in proportion to the rows (its store stream and
            // the flux kernel), the single sweep pays a pre-pass and a contraction that do not depend on them.  Measured
            // break-even near 600 class-groups (ne30, 760 groups: 2.49 against 2.80 ms at 72 x 91, 0.86 against 0.96 at
            // 72 x 30; round 3, with the contraction in LDS and 96 groups in the pre-pass: 2048).
            const int64_t min_groups = pl->opt_single_sweep_min_groups >= 0 ? pl->opt_single_sweep_min_groups : 640;
            bool want = !off && os_supported(pl) && (forced || pl->cgroups >= min_groups);
            if (want) {
              rc = build_os_tables(pl);
              if (rc == TEMX_ERANK) want = false;          // too few distinct latitudes in the subsample: class-sum form
              else if (rc) return rc;
            }
            if (want) {
              pl->sp_os = choose_split(D, cunits, pl->num_cu, 4, 8);
              // (one round of workgroups: the pre-pass is all prologue and epilogue)
              // (the pre-pass stores next to nothing since round 4, so it may be cut as finely as the chip has CUs: at
              //  D = 6552 one split per column quad left it on 103 workgroups, 0.18 ms for 430 MB)
              pl->sp_os_s = choose_split(D, std::max<int64_t>(1, pl->sbatches / 4), pl->num_cu, 4, 2);
              const size_t per = ((size_t)4 * pl->KX + 3 * pl->K) * D * 8;
              if ((rc = pl->partial.ensure(std::max((size_t)std::max(pl->sp_os.nsplit, pl->sp_os_s.nsplit) * per, pl->partial.bytes)))) return rc;
              if ((rc = pl->Ax.ensure(((size_t)4 * pl->KX + 3 * pl->K) * D * 8))) return rc;
              if ((rc = pl->Axs.ensure((size_t)4 * pl->KR * D * 8))) return rc;
              if ((rc = pl->osAt.ensure((size_t)4 * pl->NQ * D * 8))) return rc;
              if ((rc = pl->osAb.ensure((size_t)4 * pl->NQ * D * 8))) return rc;
              if ((rc = pl->rho.ensure((size_t)4 * pl->KR * D * 8))) return rc;
              if ((rc = pl->rho0.ensure((size_t)4 * pl->KR * D * 8))) return rc;
              HIPCHK(hipMemset(pl->rho0.p, 0, pl->rho0.bytes));
              const int2* cu = nullptr;
              if ((rc = os_cuts(pl, false, pl->sp_os.nsplit, &cu))) return rc;
              if ((rc = os_cuts(pl, true, pl->sp_os_s.nsplit, &cu))) return rc;
              pl->os_on = true;
            }
          }
        }
      }
    }
    // work cuts now, not at the first launch: launches must stay legal inside a stream capture
    const int2* cuts_unused = nullptr;
    if (pl->onepass && (rc = class_cuts(pl, pl->sp_cproj4.nsplit, &cuts_unused, true))) return rc;
    if (pl->onepass && (rc = class_cuts(pl, pl->sp_copw.nsplit * 4, &cuts_unused, true))) return rc;
    for (int nsub : {pl->sp_cproj4.nsplit, pl->sp_cproj1.nsplit, pl->sp_ceddy.nsplit * (8 / edpw)})
      if ((rc = class_cuts(pl, nsub, &cuts_unused))) return rc;
  }
  if (pl->sym) {
    const int64_t nch = (pl->npg + SYM_PROJ_CH - 1) / SYM_PROJ_CH;
    pl->sp_sproj4 = Split();
    pl->sp_sproj1 = Split();
    if (pick_dpw(ndt_, 4) == 4) {     // quads of d-tiles, all four fields per wave
      pl->sp_sproj4 = choose_split(D, nch, 2 * pl->num_cu, 4);
      pl->sp_sproj1 = choose_split(D, nch, 2 * pl->num_cu, 4);
    } else {                          // small ragged D: one d-tile per workgroup, one field per wave
      pl->sp_sproj4 = choose_split(D, nch, SYM_PROJ_E_WPS * pl->num_cu, 1);
    }
    pl->sp_seddy = choose_split(D, pl->npg / (8 / edpw), pl->num_cu, edpw);
    const size_t need2 = (size_t)std::max({pl->sp_sproj4.nsplit * 4, pl->sp_seddy.nsplit * (8 / edpw) * 3,
                                           pl->sp_sproj1.nsplit}) * pl->K * D * 8;
    if ((rc = pl->partial.ensure(std::max(need2, pl->partial.bytes)))) return rc;
  }
  pl->tem = true;
  return TEMX_OK;
} TEMX_CATCH

static int tem_ready(temx_plan* pl) {
  if (!pl) return fail(TEMX_EINVAL, "null plan");
  if (!pl->finalized) return fail(TEMX_ESTATE, "plan not finalised");
  if (!pl->tem) return fail(TEMX_ESTATE, "temx_plan_set_tem has not been called");
  return TEMX_OK;
}

int temx_tem_stage1(temx_plan* pl, const void* ua, const void* va, const void* ta, const void* wap,
                    int dtype, double* B4, void* stream) try {
  int rc = tem_ready(pl);
  if (rc) return rc;
  if (!ua || !va || !ta || !wap || !B4) return fail(TEMX_EINVAL, "null argument");
  HIPCHK(hipSetDevice(pl->device));
  hipStream_t st = S_(stream);
  FieldPtrs<4> fp;
  fp.p[0] = ua; fp.p[1] = va; fp.p[2] = ta; fp.p[3] = wap;
  if (pl->large) set_tail(pl, 0, pl->nt);
  if (pl->large && pl->lone) {   // class sums first, then their projection slice by slice
    pl->op_valid = false;
    if ((rc = launch_class_sums(pl, fp, dtype, st))) return rc;
    if ((rc = project_sums<4>(pl, pl->csum.d(), 7, 0, B4, st))) return rc;
    pl->op_valid = true;
    return TEMX_OK;
  }
  if (pl->large) return project_all<4>(pl, fp, dtype, pl->D, pl->colscale.d(), 2, pl->sp_proj4, B4, st);
  TimedLaunch tl{};
  time_begin(pl, 0, st, tl);
  const bool sp4 = sym_project(pl, 4);
  const Split& sp = pl->cls ? pl->sp_cproj4 : (sp4 ? pl->sp_sproj4 : pl->sp_proj4);
  const bool op = pl->cls && pl->onepass;
  set_tail(pl, 0, pl->nt);
  pl->op_valid = false;
  pl->os_valid = false;
  pl->c4_valid = false;          // C4 still describes the previous fields until a stage-2 solve has run
  pl->tq_valid = false;          // tracer class sums pair with the v, omega sums of one TEM run
  rc = op      ? launch_sweep_op<0>(pl, fp, dtype, pl->partial.d(), sp, st)
       : pl->cls ? launch_project_cls<4>(pl, fp, dtype, pl->D, pl->colscale.d(), 2, pl->partial.d(), sp, st)
       : sp4   ? launch_project_sym<4>(pl, fp, dtype, pl->D, pl->colscale.d(), 2, pl->partial.d(), sp, st)
               : launch_project<4>(pl, fp, dtype, pl->D, pl->colscale.d(), 2, pl->partial.d(), sp, st);
  time_end(pl, 0, st, tl);
  if (rc) return rc;
  const int64_t KD = (int64_t)pl->K * pl->D;
  if (!op) return launch_reduce(pl, pl->partial.d(), sp.nsplit, 4 * KD, B4, st);
  // one-pass form: 7 slabs per split -- the four fields, then the products u v, u omega, v theta
  if ((rc = launch_reduce(pl, pl->partial.d(), sp.nsplit, 4 * KD, B4, st, 7 * KD))) return rc;
  if ((rc = launch_reduce(pl, pl->partial.d() + 4 * KD, sp.nsplit, 3 * KD, pl->Pq.d(), st, 7 * KD))) return rc;
  pl->op_valid = true;           // csum / Pq now describe these fields (temx_tem_stage2_from_sums)
  return TEMX_OK;
} TEMX_CATCH

// ---- large-L (K > 64) second sweep: the fused eddy kernel keeps all coefficients of a d-tile in LDS,
// which stops at 64 harmonics.  Here the native zonal means are materialised by accumulating
// reconstruction passes (64 harmonics each), the eddies/products by one elementwise kernel, and
// the products are projected slice by slice.  ~8x the HBM traffic of the fused path; correct for
// any L <= 511.
static int large_ws(temx_plan* pl) {
  const size_t nd = (size_t)pl->N * pl->D * 8;
  int rc;
  // 8 arrays of ncol x nlev x nt doubles (107 GB at ne120 x 72 x 30): say so before the allocator does
  size_t fr = 0, tot = 0;
  const size_t have = pl->XB.bytes + pl->P3.bytes;
  if (have < 8 * nd && hipMemGetInfo(&fr, &tot) == hipSuccess && 8 * nd - have > fr)
    return fail(TEMX_ENOMEM, "the unfused second sweep (L > 63 without latitude classes, or a weighted plan) needs "
                             "%zu bytes of workspace (8 x ncol x nlev x nt doubles), %zu are free; process the "
                             "snapshots in smaller blocks (temx_plan_set_tem with a smaller nt)", 8 * nd - have, fr);
  if ((rc = pl->XB.ensure(5 * nd))) return rc;     // ub vb thetab wapb (+ qb for tracers), native
  return pl->P3.ensure(3 * nd);
}

static int launch_eddy_from_xbar(temx_plan* pl, const FieldPtrs<4>& fp, int dtype, const FieldPtrs<4>& xb,
                                 const double* colscale, const EddyOut& eo, hipStream_t st, int64_t nrows = -1) {
  if (nrows < 0) nrows = pl->N;
  dim3 grid((unsigned)(pl->num_cu * 8)), block(256);
  if (dtype == TEMX_F64)
    hipLaunchKernelGGL(eddy_from_xbar_kernel<double>, grid, block, 0, st, fp, xb, nrows, pl->D, colscale, eo);
  else if (dtype == TEMX_F32)
    hipLaunchKernelGGL(eddy_from_xbar_kernel<float>, grid, block, 0, st, fp, xb, nrows, pl->D, colscale, eo);
  else
    return fail(TEMX_EINVAL, "dtype must be TEMX_F64 or TEMX_F32");
  HIPCHK(hipGetLastError());
  return TEMX_OK;
}

static FieldPtrs<4> native_means(temx_plan* pl, int a, int b, int c, int d) {
  const int64_t nd = pl->N * pl->D;
  FieldPtrs<4> xb;
  xb.p[0] = pl->XB.d() + a * nd; xb.p[1] = pl->XB.d() + b * nd;
  xb.p[2] = pl->XB.d() + c * nd; xb.p[3] = pl->XB.d() + d * nd;
  return xb;
}

// native zonal means of the four fields from the coefficients of the last solve (large-L paths)
static int ensure_xb(temx_plan* pl, hipStream_t st) {
  if (pl->xb_valid) return TEMX_OK;
  int rc;
  if ((rc = large_ws(pl))) return rc;
  const int64_t nd = pl->N * pl->D;
  for (int f = 0; f < 4; ++f)
    if ((rc = launch_recon(pl, pl->D, pl->C4.d() + (int64_t)f * pl->K4 * pl->D, pl->XB.d() + f * nd, st))) return rc;
  pl->xb_valid = true;
  return TEMX_OK;
}

static int tem_stage2_large(temx_plan* pl, const FieldPtrs<4>& fp, int dtype, const double* B4, double* B3,
                            hipStream_t st) {
  int rc;
  if ((rc = large_ws(pl))) return rc;
  const int64_t nd = pl->N * pl->D;
  if ((rc = launch_solve(pl, B4, 4, pl->D, pl->C4.d(), pl->zb.d(), st))) return rc;
  pl->c4_valid = true;
  for (int f = 0; f < 4; ++f)
    if ((rc = launch_recon(pl, pl->D, pl->C4.d() + (int64_t)f * pl->K4 * pl->D, pl->XB.d() + f * nd, st))) return rc;
  EddyOut eo{};
  FieldPtrs<3> f3;
  for (int i = 0; i < 3; ++i) {
    eo.p[4 + i] = pl->P3.d() + i * nd;
    f3.p[i] = eo.p[4 + i];
  }
  pl->xb_valid = true;
  if ((rc = launch_eddy_from_xbar(pl, fp, dtype, native_means(pl, 0, 1, 2, 3), pl->colscale.d(), eo, st))) return rc;
  return project_all<3>(pl, f3, TEMX_F64, pl->D, nullptr, -1, pl->sp_proj1, B3, st);
}

int temx_tem_stage2(temx_plan* pl, const void* ua, const void* va, const void* ta, const void* wap,
                    int dtype, const double* B4, double* B3, void* stream) try {
  int rc = tem_ready(pl);
  if (rc) return rc;
  if (!ua || !va || !ta || !wap || !B4 || !B3) return fail(TEMX_EINVAL, "null argument");
  HIPCHK(hipSetDevice(pl->device));
  hipStream_t st = S_(stream);
  set_tail(pl, 0, pl->nt);
  if (unfused_stage2(pl)) return tem_stage2_large(pl, four(ua, va, ta, wap), dtype, B4, B3, st);
  // C = G^-1 B4 and the four zonal means ub vb thetab wapb -> zb[0..3]
  if ((rc = launch_solve(pl, B4, 4, pl->D, pl->C4.d(), pl->zb.d(), st))) return rc;
  pl->c4_valid = true;
  TimedLaunch tl{};
  time_begin(pl, 1, st, tl);
  rc = run_eddy<0>(pl, four(ua, va, ta, wap), dtype, pl->C4.d(), pl->partial.d(), nullptr, st);
  time_end(pl, 1, st, tl);
  if (rc) return rc;
  return launch_reduce(pl, pl->partial.d(), eddy_slabs(pl), (int64_t)3 * pl->K * pl->D, B3, st);
} TEMX_CATCH

int temx_tem_stage2_from_sums(temx_plan* pl, const double* B4, double* B3, void* stream) try {
  int rc = tem_ready(pl);
  if (rc) return rc;
  if (!B4 || !B3) return fail(TEMX_EINVAL, "null argument");
  if (!((pl->cls && pl->onepass) || (pl->large && pl->lone)))
    return fail(TEMX_ESTATE, "the plan does not run the one-pass class path (temx_plan_one_pass)");
  if (!pl->op_valid)
    return fail(TEMX_ESTATE, "no class sums: temx_tem_stage1 must precede temx_tem_stage2_from_sums");
  HIPCHK(hipSetDevice(pl->device));
  hipStream_t st = S_(stream);
  if (!tail_is_whole(pl)) return fail(TEMX_ESTATE, "a time-sliced tail ran since the last temx_tem_stage1");
  if ((rc = launch_solve(pl, B4, 4, pl->D, pl->C4.d(), pl->zb.d(), st))) return rc;
  pl->c4_valid = true;
  if (pl->large) {
    if ((rc = launch_flux_large(pl, pl->C4.d(), st))) return rc;
    pl->xb_valid = false;          // the native means are not materialised on this path
    return project_sums<3>(pl, pl->pbuf.d(), 3, 0, B3, st);
  }
  TimedLaunch tl{};
  time_begin(pl, 1, st, tl);
  rc = launch_flux_cls<0>(pl, pl->C4.d(), pl->partial.d(), pl->sp_cflux, st);
  time_end(pl, 1, st, tl);
  if (rc) return rc;
  // B3 = (projections of u v, u omega, v theta from sweep 1) + (projected corrections)
  return launch_reduce(pl, pl->partial.d(), pl->sp_cflux.nsplit, (int64_t)3 * pl->K * pl->D, B3, st, -1,
                       pl->Pq.d());
} TEMX_CATCH

int temx_tem_stage3(temx_plan* pl, const double* B3, double* results, double* zonal, void* stream) try {
  int rc = tem_ready(pl);
  if (rc) return rc;
  if (!B3 || !results) return fail(TEMX_EINVAL, "null argument");
  HIPCHK(hipSetDevice(pl->device));
  if (!tail_is_whole(pl))
    return fail(TEMX_ESTATE, "the zonal means in the plan are those of a time slice (temx_tem_os_tail); stage 3 needs a "
                             "temx_tem_stage2 / stage2_from_sums on the whole run first");
  return tem_stage3_impl(pl, B3, results, zonal, S_(stream));
} TEMX_CATCH

// flux zonal means, derivatives, psi, integral and the ten diagnostics for the snapshots of the tail (pl->tnt of them;
// zb[0..3] hold the zonal means of the four fields for the same snapshots)
static int tem_stage3_impl(temx_plan* pl, const double* B3, double* results, double* zonal, hipStream_t st) {
  int rc;
  const int64_t Dt = pl->tD, nts = pl->tnt;
  const int64_t MD = (int64_t)pl->M * Dt;
  // flux zonal means upvpb upwappb vptpb -> zb[4..6]
  if ((rc = launch_solve(pl, B3, 3, Dt, nullptr, pl->zb.d() + 4 * MD, st))) return rc;
  // int_vbdp -> zb[7]: by a wavefront scan; inside the epilogue only for short columns on small zonal grids
  // (measured: at nlev = 72 the O(nlev) loop per point costs what the extra launch saves, at 128 more)
  EpiTables tb{pl->p.d(), pl->pg.d(), pl->lg.d(), pl->coslat.d(), pl->fcor.d()};
  static const bool epi_wg = !(getenv("TEMX_EPI_WG") && atoi(getenv("TEMX_EPI_WG")) == 0);   // A/B only
  if (epi_wg && Dt <= EPI_WG_MAXD) {          // small zonal grid: the scan inside the epilogue, one workgroup per latitude
    hipLaunchKernelGGL(tem_epilogue_kernel<2>, dim3((unsigned)pl->M), dim3(256), 0, st, pl->zb.d(), pl->M, pl->nlev, nts, tb,
                       pl->p0, results, zonal);
  } else if (pl->nlev > 40 || MD > ((int64_t)1 << 17)) {
    const int64_t ncols = (int64_t)pl->M * nts;
    hipLaunchKernelGGL(pint_scan_kernel, dim3((unsigned)((ncols + 3) / 4)), dim3(256), 0, st, pl->zb.d() + 1 * MD,
                       pl->p.d(), pl->M, pl->nlev, nts, pl->zb.d() + 7 * MD);
    HIPCHK(hipGetLastError());
    hipLaunchKernelGGL(tem_epilogue_kernel<0>, dim3((unsigned)((MD + 255) / 256)), dim3(256), 0, st, pl->zb.d(),
                       pl->M, pl->nlev, nts, tb, pl->p0, results, zonal);
  } else {
    hipLaunchKernelGGL(tem_epilogue_kernel<1>, dim3((unsigned)((MD + 255) / 256)), dim3(256), 0, st, pl->zb.d(),
                       pl->M, pl->nlev, nts, tb, pl->p0, results, zonal);
  }
  HIPCHK(hipGetLastError());
  return TEMX_OK;
}

// ---- the single sweep in three steps (ncol-sharded jobs exchange between them; see include/temx.h) -----------
static int os_ready(temx_plan* pl) {
  int rc = tem_ready(pl);
  if (rc) return rc;
  if (!pl->os_on) return fail(TEMX_ESTATE, "the plan does not run the single-sweep form (temx_plan_single_sweep)");
  if (pl->os_need_global)
    return fail(TEMX_ESTATE, "ncol-sharded single sweep: temx_plan_set_os_matrices with the all-reduced TEMX_MAT_GX / TEMX_MAT_GSUB first");
  return TEMX_OK;
}
static int slices_ok(const temx_plan* pl, int nslices) {
  if (nslices < 1 || nslices > pl->nt)
    return fail(TEMX_EINVAL, "nslices = %d: a time-sliced tail needs 1 <= nslices <= nt = %lld", nslices, (long long)pl->nt);
  return TEMX_OK;
}

int temx_tem_os_prepass(temx_plan* pl, const void* ua, const void* va, const void* ta, const void* wap, int dtype,
                        double* As, void* stream) try {
  int rc = os_ready(pl);
  if (rc) return rc;
  if (!ua || !va || !ta || !wap || !As) return fail(TEMX_EINVAL, "null argument");
  if (dtype != TEMX_F64 && dtype != TEMX_F32) return fail(TEMX_EINVAL, "dtype must be TEMX_F64 or TEMX_F32");
  HIPCHK(hipSetDevice(pl->device));
  return os_prepass(pl, four(ua, va, ta, wap), dtype, As, S_(stream));
} TEMX_CATCH

int temx_tem_os_sweep(temx_plan* pl, const void* ua, const void* va, const void* ta, const void* wap, int dtype,
                      const double* As, int nslices, double* proj, void* stream) try {
  int rc = os_ready(pl);
  if (rc) return rc;
  if (!ua || !va || !ta || !wap || !As || !proj) return fail(TEMX_EINVAL, "null argument");
  if (dtype != TEMX_F64 && dtype != TEMX_F32) return fail(TEMX_EINVAL, "dtype must be TEMX_F64 or TEMX_F32");
  if ((rc = slices_ok(pl, nslices))) return rc;
  HIPCHK(hipSetDevice(pl->device));
  return os_sweep(pl, four(ua, va, ta, wap), dtype, As, nslices, proj, S_(stream));
} TEMX_CATCH

int temx_tem_os_tail(temx_plan* pl, const double* proj_slice, int64_t t0, int64_t nts, double* results, double* zonal,
                     void* stream) try {
  int rc = os_ready(pl);
  if (rc) return rc;
  if (!proj_slice || !results) return fail(TEMX_EINVAL, "null argument");
  if (t0 < 0 || nts < 1 || t0 + nts > pl->nt)
    return fail(TEMX_EINVAL, "snapshots [%lld, %lld) are not inside the run (nt = %lld)", (long long)t0, (long long)(t0 + nts), (long long)pl->nt);
  HIPCHK(hipSetDevice(pl->device));
  return os_tail(pl, proj_slice, t0, nts, results, zonal, S_(stream));
} TEMX_CATCH

static int tracers_args(const temx_plan* pl, int nq, const void* const* q_host, const void* va, const void* wap, int dtype) {
  if (nq != 1 && nq != 2) return fail(TEMX_EINVAL, "nq = %d: one or two tracers per sweep", nq);
  if (!q_host || !q_host[0] || (nq == 2 && !q_host[1]) || !va || !wap) return fail(TEMX_EINVAL, "null argument");
  if (dtype != TEMX_F64 && dtype != TEMX_F32) return fail(TEMX_EINVAL, "dtype must be TEMX_F64 or TEMX_F32");
  if (nq == 2 && (pl->os_tile || pl->osc_lds))
    return fail(TEMX_EUNSUPPORTED, "two tracers per sweep: not with TEMX_OPT_OS_MAP = tile / TEMX_OPT_OS_CONTRACT = lds");
  return TEMX_OK;
}

int temx_tracers_os_prepass(temx_plan* pl, int nq, const void* const* q_host, const void* va, const void* wap, int dtype,
                            double* Asq, void* stream) try {
  int rc = os_ready(pl);
  if (rc) return rc;
  if ((rc = tracers_args(pl, nq, q_host, va, wap, dtype))) return rc;
  if (!Asq) return fail(TEMX_EINVAL, "null argument");
  HIPCHK(hipSetDevice(pl->device));
  return tracer_os_prepass(pl, nq, tracer_fields(nq, q_host, va, wap), dtype, Asq, S_(stream));
} TEMX_CATCH

int temx_tracers_os_sweep(temx_plan* pl, int nq, const void* const* q_host, const void* va, const void* wap, int dtype,
                          const double* Asq, int nslices, double* projq, void* stream) try {
  int rc = os_ready(pl);
  if (rc) return rc;
  if ((rc = tracers_args(pl, nq, q_host, va, wap, dtype))) return rc;
  if (!Asq || !projq) return fail(TEMX_EINVAL, "null argument");
  if ((rc = slices_ok(pl, nslices))) return rc;
  HIPCHK(hipSetDevice(pl->device));
  return tracer_os_sweep(pl, nq, tracer_fields(nq, q_host, va, wap), dtype, Asq, nslices, projq, S_(stream));
} TEMX_CATCH

int temx_tracers_os_tail(temx_plan* pl, int nq, const double* projq_slice, double* const* tres_host, double* const* tzon_host,
                         void* stream) try {
  int rc = os_ready(pl);
  if (rc) return rc;
  if (nq != 1 && nq != 2) return fail(TEMX_EINVAL, "nq = %d: one or two tracers per sweep", nq);
  if (!projq_slice || !tres_host || !tres_host[0] || (nq == 2 && !tres_host[1])) return fail(TEMX_EINVAL, "null argument");
  if (!pl->os_valid || !pl->c4_valid)
    return fail(TEMX_ESTATE, "the tracers' tail needs the state of a temx_tem_os_tail on the same snapshots");
  HIPCHK(hipSetDevice(pl->device));
  if ((rc = tracer_os_ws(pl, nq))) return rc;
  return tracer_os_tail(pl, nq, projq_slice, tres_host, tzon_host, S_(stream));
} TEMX_CATCH

// stages 2b + 3 on a time slice, from raw sums of any form of the sweeps: B4s [4][K][nlev][nts], B3s [3][K][nlev][nts]
int temx_tem_tail_from_sums(temx_plan* pl, const double* B4s, const double* B3s, int64_t t0, int64_t nts, double* results,
                            double* zonal, void* stream) try {
  int rc = tem_ready(pl);
  if (rc) return rc;
  if (!B4s || !B3s || !results) return fail(TEMX_EINVAL, "null argument");
  if (t0 < 0 || nts < 1 || t0 + nts > pl->nt)
    return fail(TEMX_EINVAL, "snapshots [%lld, %lld) are not inside the run (nt = %lld)", (long long)t0, (long long)(t0 + nts), (long long)pl->nt);
  HIPCHK(hipSetDevice(pl->device));
  hipStream_t st = S_(stream);
  set_tail(pl, t0, nts);
  pl->c4_valid = pl->os_valid = false;                        // zb describes the slice from here on
  if ((rc = launch_solve(pl, B4s, 4, pl->tD, nullptr, pl->zb.d(), st))) return rc;
  return tem_stage3_impl(pl, B3s, results, zonal, st);
} TEMX_CATCH

// Cut rows of [nlev][nt] columns into the time slices a reduce-scatter wants: out[w][row][lev][t - t0(w)], slice w
// padded to rows * nlev * ceil(nt / nslices) doubles (kernels.hpp, SliceMap).
int temx_time_slices(temx_plan* pl, const double* B, int64_t rows, int nslices, double* out, void* stream) try {
  int rc = tem_ready(pl);
  if (rc) return rc;
  if (!B || !out || rows < 1) return fail(TEMX_EINVAL, "bad argument");
  if ((rc = slices_ok(pl, nslices))) return rc;
  HIPCHK(hipSetDevice(pl->device));
  return launch_reduce(pl, B, 1, rows * pl->D, out, S_(stream), -1, nullptr, slice_map(pl, nslices, rows, 0));
} TEMX_CATCH

int temx_tem_run(temx_plan* pl, const void* ua, const void* va, const void* ta, const void* wap,
                 int dtype, double* results, double* zonal, void* stream) try {
  int rc = tem_ready(pl);
  if (rc) return rc;
  if (os_active(pl, dtype)) {
    if ((rc = os_ready(pl))) return rc;
    if (!ua || !va || !ta || !wap || !results) return fail(TEMX_EINVAL, "null argument");
    if (dtype != TEMX_F64 && dtype != TEMX_F32) return fail(TEMX_EINVAL, "dtype must be TEMX_F64 or TEMX_F32");
    HIPCHK(hipSetDevice(pl->device));
    return tem_run_os(pl, four(ua, va, ta, wap), dtype, results, zonal, stream);
  }
  if ((rc = temx_tem_stage1(pl, ua, va, ta, wap, dtype, pl->B4.d(), stream))) return rc;
  if (temx_plan_one_pass(pl))
    rc = temx_tem_stage2_from_sums(pl, pl->B4.d(), pl->B3.d(), stream);
  else
    rc = temx_tem_stage2(pl, ua, va, ta, wap, dtype, pl->B4.d(), pl->B3.d(), stream);
  if (rc) return rc;
  return temx_tem_stage3(pl, pl->B3.d(), results, zonal, stream);
} TEMX_CATCH

int temx_tem_eddy(temx_plan* pl, const void* ua, const void* va, const void* ta, const void* wap,
                  int dtype, double* const* eddy_ptrs_host, void* stream) try {
  int rc = tem_ready(pl);
  if (rc) return rc;
  if (!ua || !va || !ta || !wap || !eddy_ptrs_host) return fail(TEMX_EINVAL, "null argument");
  HIPCHK(hipSetDevice(pl->device));
  if (!tail_is_whole(pl)) return fail(TEMX_ESTATE, "the plan holds the coefficients of a time slice (temx_tem_os_tail), not of the whole run");
  EddyOut eo;
  for (int i = 0; i < TEMX_NEDDY; ++i) eo.p[i] = eddy_ptrs_host[i];
  if (unfused_stage2(pl)) {
    if ((rc = ensure_xb(pl, S_(stream)))) return rc;
    return launch_eddy_from_xbar(pl, four(ua, va, ta, wap), dtype, native_means(pl, 0, 1, 2, 3),
                                 pl->colscale.d(), eo, S_(stream));
  }
  return run_eddy<0>(pl, four(ua, va, ta, wap), dtype, pl->C4.d(), nullptr, &eo, S_(stream));
} TEMX_CATCH

int temx_tem_eddy_rows(temx_plan* pl, const void* ua, const void* va, const void* ta, const void* wap,
                       int dtype, int64_t row0, int64_t nrows, double* const* eddy_ptrs_host, void* stream) try {
  int rc = tem_ready(pl);
  if (rc) return rc;
  if (!ua || !va || !ta || !wap || !eddy_ptrs_host) return fail(TEMX_EINVAL, "null argument");
  if (row0 < 0 || nrows < 1 || row0 + nrows > pl->N || (row0 & 15))
    return fail(TEMX_EINVAL, "row range [%lld, %lld) must lie inside the grid and start at a multiple of 16",
                (long long)row0, (long long)(row0 + nrows));
  if (dtype != TEMX_F64 && dtype != TEMX_F32) return fail(TEMX_EINVAL, "dtype must be TEMX_F64 or TEMX_F32");
  HIPCHK(hipSetDevice(pl->device));
  if (!tail_is_whole(pl)) return fail(TEMX_ESTATE, "the plan holds the coefficients of a time slice (temx_tem_os_tail), not of the whole run");
  hipStream_t st = S_(stream);
  const int64_t D = pl->D, nd = nrows * D;
  // native zonal means of the rows (coefficients of the last temx_tem_stage2), then elementwise eddies
  DevBuf& ws = pl->opC;                       // operator-API workspace doubles as the row-chunk buffer
  if ((rc = ws.ensure((size_t)4 * nd * 8))) return rc;
  FieldPtrs<4> xb, fp;
  const void* src[4] = {ua, va, ta, wap};
  const size_t es = dtype == TEMX_F64 ? 8 : 4;
  for (int f = 0; f < 4; ++f) {
    double* o = ws.d() + (int64_t)f * nd;
    if ((rc = launch_recon(pl, D, pl->C4.d() + (int64_t)f * pl->K4 * D, o, st, row0, nrows))) return rc;
    xb.p[f] = o;
    fp.p[f] = static_cast<const char*>(src[f]) + (size_t)row0 * D * es;
  }
  EddyOut eo;
  for (int i = 0; i < TEMX_NEDDY; ++i) eo.p[i] = eddy_ptrs_host[i];
  return launch_eddy_from_xbar(pl, fp, dtype, xb, pl->colscale.d(), eo, st, nrows);
} TEMX_CATCH

// ---- tracer TEM -----------------------------------------------------------------------------------
static int tracer_ws(temx_plan* pl) {
  const int64_t D = pl->D;
  int rc;
  if ((rc = pl->Bq.ensure((size_t)pl->K * D * 8))) return rc;
  if ((rc = pl->Bq2.ensure((size_t)2 * pl->K * D * 8))) return rc;
  if ((rc = pl->Ct.ensure((size_t)3 * pl->K4 * D * 8))) return rc;
  if ((rc = pl->tz.ensure((size_t)3 * pl->M * D * 8))) return rc;
  const size_t need = (size_t)std::max(pl->sp_proj1.nsplit, pl->sp_cproj1.nsplit) * pl->K * D * 8;
  return pl->partial.ensure(std::max(need, pl->partial.bytes));
}

int temx_tracer_stage1(temx_plan* pl, const void* q, int dtype, double* Bq, void* stream) try {
  int rc = tem_ready(pl);
  if (rc) return rc;
  if (!q || !Bq) return fail(TEMX_EINVAL, "null argument");
  HIPCHK(hipSetDevice(pl->device));
  if ((rc = tracer_ws(pl))) return rc;
  FieldPtrs<1> fp;
  fp.p[0] = q;
  if (pl->large) return project_all<1>(pl, fp, dtype, pl->D, nullptr, -1, pl->sp_proj1, Bq, S_(stream));
  const bool sp1 = sym_project(pl, 1);
  const Split& sp = pl->cls ? pl->sp_cproj1 : (sp1 ? pl->sp_sproj1 : pl->sp_proj1);
  rc = pl->cls ? launch_project_cls<1>(pl, fp, dtype, pl->D, nullptr, -1, pl->partial.d(), sp, S_(stream))
       : sp1   ? launch_project_sym<1>(pl, fp, dtype, pl->D, nullptr, -1, pl->partial.d(), sp, S_(stream))
               : launch_project<1>(pl, fp, dtype, pl->D, nullptr, -1, pl->partial.d(), sp, S_(stream));
  if (rc) return rc;
  return launch_reduce(pl, pl->partial.d(), sp.nsplit, (int64_t)pl->K * pl->D, Bq, S_(stream));
} TEMX_CATCH

int temx_tracer_stage2(temx_plan* pl, const void* q, const void* va, const void* wap, int dtype,
                       const double* Bq, double* Bq2, void* stream) try {
  int rc = tem_ready(pl);
  if (rc) return rc;
  if (!q || !va || !wap || !Bq || !Bq2) return fail(TEMX_EINVAL, "null argument");
  HIPCHK(hipSetDevice(pl->device));
  if (!tail_is_whole(pl)) return fail(TEMX_ESTATE, "the plan holds the coefficients of a time slice (temx_tem_os_tail), not of the whole run");
  if ((rc = tracer_ws(pl))) return rc;
  hipStream_t st = S_(stream);
  const size_t slab = (size_t)pl->K4 * pl->D * 8;
  // coefficients: Ct = (C_q, C_v, C_w); qb -> tz[0]
  if ((rc = launch_solve(pl, Bq, 1, pl->D, pl->Ct.d(), pl->tz.d(), st))) return rc;
  if (unfused_stage2(pl)) {   // q' v' and q' omega' from the native means of the last temx_tem_stage2
    if ((rc = ensure_xb(pl, st))) return rc;
    const int64_t nd = pl->N * pl->D;
    if ((rc = launch_recon(pl, pl->D, pl->Ct.d(), pl->XB.d() + 4 * nd, st))) return rc;
    EddyOut eo{};
    eo.p[4] = pl->P3.d();
    eo.p[5] = pl->P3.d() + nd;
    if ((rc = launch_eddy_from_xbar(pl, four(q, va, va, wap), dtype, native_means(pl, 4, 1, 1, 3), nullptr, eo, st)))
      return rc;
    for (int i = 0; i < 2; ++i) {
      FieldPtrs<1> f1;
      f1.p[0] = eo.p[4 + i];
      if ((rc = project_all<1>(pl, f1, TEMX_F64, pl->D, nullptr, -1, pl->sp_proj1, Bq2 + (int64_t)i * pl->K * pl->D, st)))
        return rc;
    }
    return TEMX_OK;
  }
  HIPCHK(hipMemcpyAsync((char*)pl->Ct.p + slab, (char*)pl->C4.p + slab, slab, hipMemcpyDeviceToDevice, st));
  HIPCHK(hipMemcpyAsync((char*)pl->Ct.p + 2 * slab, (char*)pl->C4.p + 3 * slab, slab, hipMemcpyDeviceToDevice, st));
  if ((rc = run_eddy<1>(pl, four(q, va, wap, nullptr), dtype, pl->Ct.d(), pl->partial.d(), nullptr, st))) return rc;
  return launch_reduce(pl, pl->partial.d(), eddy_slabs(pl), (int64_t)2 * pl->K * pl->D, Bq2, st);
} TEMX_CATCH

int temx_tracer_stage3(temx_plan* pl, const double* Bq2, double* tres, double* tzon, void* stream) try {
  int rc = tem_ready(pl);
  if (rc) return rc;
  if (!Bq2 || !tres) return fail(TEMX_EINVAL, "null argument");
  if (!pl->tz.p) return fail(TEMX_ESTATE, "temx_tracer_stage2 has not been called");
  if (!tail_is_whole(pl)) return fail(TEMX_ESTATE, "the plan holds the zonal means of a time slice (temx_tem_os_tail)");
  HIPCHK(hipSetDevice(pl->device));
  return tracer_stage3_impl(pl, Bq2, tres, tzon, S_(stream));
} TEMX_CATCH

static int tracer_stage3_impl(temx_plan* pl, const double* Bq2, double* tres, double* tzon, hipStream_t st) {
  int rc;
  const int64_t Dt = pl->tD;
  const int64_t MD = (int64_t)pl->M * Dt;
  if ((rc = launch_solve(pl, Bq2, 2, Dt, nullptr, pl->tz.d() + MD, st))) return rc;   // qpvpb, qpwappb
  EpiTables tb{pl->p.d(), pl->pg.d(), pl->lg.d(), pl->coslat.d(), pl->fcor.d()};
  hipLaunchKernelGGL(tracer_epilogue_kernel, dim3((unsigned)((MD + 255) / 256)), dim3(256), 0, st, pl->zb.d(),
                     pl->tz.d(), pl->M, pl->nlev, pl->tnt, tb, pl->p0, tres, tzon);
  HIPCHK(hipGetLastError());
  return TEMX_OK;
}

// the one-pass tracer stages pair q's class sums with the v, omega class sums (csum: stage 1) and with the v,
// omega coefficients (C4: stage 2) of the latest TEM run; both must describe the same fields
static inline bool tracer_one_pass(const temx_plan* pl) { return pl->cls && pl->onepass && pl->op_valid; }

int temx_tracer_stage1_sums(temx_plan* pl, const void* q, const void* va, const void* wap, int dtype,
                            double* Bq, void* stream) try {
  int rc = tem_ready(pl);
  if (rc) return rc;
  if (!q || !va || !wap || !Bq) return fail(TEMX_EINVAL, "null argument");
  if (!tracer_one_pass(pl))
    return fail(TEMX_ESTATE, "needs the one-pass class path and the class sums of a TEM run on this plan "
                             "(temx_plan_one_pass, temx_tem_stage1)");
  HIPCHK(hipSetDevice(pl->device));
  if ((rc = tracer_ws(pl))) return rc;
  hipStream_t st = S_(stream);
  const Split& sp = pl->sp_cproj4;
  const int64_t KD = (int64_t)pl->K * pl->D;
  if ((rc = pl->csq.ensure((size_t)pl->cgroups * sp.ndt * 2 * 64 * 8))) return rc;
  if ((rc = pl->Pq2.ensure((size_t)2 * KD * 8))) return rc;
  pl->tq_valid = false;
  // one read of (q, v, omega): 3 slabs per split -- q, then the co-moments of q v and q omega
  if ((rc = launch_sweep_op<1>(pl, four(q, va, wap, nullptr), dtype, pl->partial.d(), sp, st))) return rc;
  if ((rc = launch_reduce(pl, pl->partial.d(), sp.nsplit, KD, Bq, st, 3 * KD))) return rc;
  if ((rc = launch_reduce(pl, pl->partial.d() + KD, sp.nsplit, 2 * KD, pl->Pq2.d(), st, 3 * KD))) return rc;
  pl->tq_valid = true;
  return TEMX_OK;
} TEMX_CATCH

int temx_tracer_stage2_from_sums(temx_plan* pl, const double* Bq, double* Bq2, void* stream) try {
  int rc = tem_ready(pl);
  if (rc) return rc;
  if (!Bq || !Bq2) return fail(TEMX_EINVAL, "null argument");
  if (tracer_one_pass(pl) && !pl->c4_valid)
    return fail(TEMX_ESTATE, "temx_tem_stage1 ran on new fields but no temx_tem_stage2 since: the v, omega "
                             "coefficients belong to the previous fields");
  if (!tracer_one_pass(pl) || !pl->tq_valid)
    return fail(TEMX_ESTATE, "no tracer class sums: temx_tracer_stage1_sums must precede temx_tracer_stage2_from_sums");
  HIPCHK(hipSetDevice(pl->device));
  if (!tail_is_whole(pl)) return fail(TEMX_ESTATE, "the plan holds the coefficients of a time slice (temx_tem_os_tail), not of the whole run");
  hipStream_t st = S_(stream);
  const size_t slab = (size_t)pl->K4 * pl->D * 8;
  // coefficients: Ct = (C_q, C_v, C_w); qb -> tz[0]
  if ((rc = launch_solve(pl, Bq, 1, pl->D, pl->Ct.d(), pl->tz.d(), st))) return rc;
  HIPCHK(hipMemcpyAsync((char*)pl->Ct.p + slab, (char*)pl->C4.p + slab, slab, hipMemcpyDeviceToDevice, st));
  HIPCHK(hipMemcpyAsync((char*)pl->Ct.p + 2 * slab, (char*)pl->C4.p + 3 * slab, slab, hipMemcpyDeviceToDevice, st));
  if ((rc = launch_flux_cls<1>(pl, pl->Ct.d(), pl->partial.d(), pl->sp_cflux, st))) return rc;
  // Bq2 = (projected co-moments of q v, q omega from the sweep) + (projected n (m_q - qb)(m_v - vb) terms)
  return launch_reduce(pl, pl->partial.d(), pl->sp_cflux.nsplit, (int64_t)2 * pl->K * pl->D, Bq2, st, -1,
                       pl->Pq2.d());
} TEMX_CATCH

// TEM stage 1 and the tracer's one-pass stage 1 in ONE sweep over (u, v, T, omega, q): the fields are read once
// for both (40 B per grid point instead of 32 + 24).  State afterwards: as after temx_tem_stage1 followed by
// temx_tracer_stage1_sums.
int temx_tem_tracer_stage1(temx_plan* pl, const void* ua, const void* va, const void* ta, const void* wap,
                           const void* q, int dtype, double* B4, double* Bq, void* stream) try {
  int rc = tem_ready(pl);
  if (rc) return rc;
  if (!ua || !va || !ta || !wap || !q || !B4 || !Bq) return fail(TEMX_EINVAL, "null argument");
  if (dtype != TEMX_F64 && dtype != TEMX_F32) return fail(TEMX_EINVAL, "dtype must be TEMX_F64 or TEMX_F32");
  if (!(pl->cls && pl->onepass) || pl->large)
    return fail(TEMX_ESTATE, "the fused TEM + tracer sweep needs the one-pass class path (temx_plan_one_pass)");
  HIPCHK(hipSetDevice(pl->device));
  if ((rc = tracer_ws(pl))) return rc;
  hipStream_t st = S_(stream);
  const Split& sp = pl->sp_copw;
  const int64_t KD = (int64_t)pl->K * pl->D;
  if ((rc = pl->csq.ensure((size_t)pl->cgroups * sp.ndt * 2 * 64 * 8))) return rc;
  if ((rc = pl->Pq2.ensure((size_t)2 * KD * 8))) return rc;
  if ((rc = pl->partial.ensure(std::max((size_t)sp.nsplit * 10 * KD * 8, pl->partial.bytes)))) return rc;
  set_tail(pl, 0, pl->nt);
  pl->op_valid = pl->c4_valid = pl->tq_valid = pl->os_valid = false;
  FieldPtrs<5> fp;
  fp.p[0] = ua; fp.p[1] = va; fp.p[2] = ta; fp.p[3] = wap; fp.p[4] = q;
  TimedLaunch tl{};
  time_begin(pl, 0, st, tl);
  rc = dtype == TEMX_F64 ? launch_sweep_opw2_t<double>(pl, fp, pl->partial.d(), sp, st)
                         : launch_sweep_opw2_t<float>(pl, fp, pl->partial.d(), sp, st);
  time_end(pl, 0, st, tl);
  if (rc) return rc;
  // 10 slabs per split: u v theta omega | q | u v, u omega, v theta | q v, q omega
  const double* P = pl->partial.d();
  if ((rc = launch_reduce(pl, P, sp.nsplit, 4 * KD, B4, st, 10 * KD))) return rc;
  if ((rc = launch_reduce(pl, P + 4 * KD, sp.nsplit, KD, Bq, st, 10 * KD))) return rc;
  if ((rc = launch_reduce(pl, P + 5 * KD, sp.nsplit, 3 * KD, pl->Pq.d(), st, 10 * KD))) return rc;
  if ((rc = launch_reduce(pl, P + 8 * KD, sp.nsplit, 2 * KD, pl->Pq2.d(), st, 10 * KD))) return rc;
  pl->op_valid = true;
  pl->tq_valid = true;
  return TEMX_OK;
} TEMX_CATCH

int temx_tem_tracer_run(temx_plan* pl, const void* ua, const void* va, const void* ta, const void* wap,
                        const void* q, int dtype, double* results, double* zonal, double* tres, double* tzon,
                        void* stream) try {
  int rc = tem_ready(pl);
  if (rc) return rc;
  if (!q || !tres) return fail(TEMX_EINVAL, "null argument");
  // any other path -- or the single-sweep forms, which beat the fused class-sum sweep -- : the two runs one after the other
  if (!(pl->cls && pl->onepass) || pl->large || os_active(pl, dtype)) {
    if ((rc = temx_tem_run(pl, ua, va, ta, wap, dtype, results, zonal, stream))) return rc;
    return temx_tracer_run(pl, q, va, wap, dtype, tres, tzon, stream);
  }
  if ((rc = tracer_ws(pl))) return rc;
  if ((rc = temx_tem_tracer_stage1(pl, ua, va, ta, wap, q, dtype, pl->B4.d(), pl->Bq.d(), stream))) return rc;
  if ((rc = temx_tem_stage2_from_sums(pl, pl->B4.d(), pl->B3.d(), stream))) return rc;
  if ((rc = temx_tem_stage3(pl, pl->B3.d(), results, zonal, stream))) return rc;
  if ((rc = temx_tracer_stage2_from_sums(pl, pl->Bq.d(), pl->Bq2.d(), stream))) return rc;
  return temx_tracer_stage3(pl, pl->Bq2.d(), tres, tzon, stream);
} TEMX_CATCH

int temx_tracer_run(temx_plan* pl, const void* q, const void* va, const void* wap, int dtype,
                    double* tres, double* tzon, void* stream) try {
  int rc = tem_ready(pl);
  if (rc) return rc;
  if (pl->os_valid && pl->c4_valid && tail_is_whole(pl) && os_active(pl, dtype)) {   // after a single-sweep TEM run: the tracer's single sweep
    if (!q || !va || !wap || !tres) return fail(TEMX_EINVAL, "null argument");
    if (dtype != TEMX_F64 && dtype != TEMX_F32) return fail(TEMX_EINVAL, "dtype must be TEMX_F64 or TEMX_F32");
    HIPCHK(hipSetDevice(pl->device));
    const void* qs[1] = {q};
    double* tr[1] = {tres};
    double* tz[1] = {tzon};
    return tracer_run_os(pl, 1, qs, va, wap, dtype, tr, tz, stream);
  }
  if ((rc = tracer_ws(pl))) return rc;
  // The one-pass form reads (q, v, omega) once instead of q + (q, v, omega), but its sweep shares a SIMD
  // with fewer waves than the two-pass kernels: measured on ne120 x 72 x 30 it is no faster (10.2 ms
  // against 9.6 ms), so the two-pass stages stay the default and TEMX_TRACER_ONE_PASS=1 selects it.
  if (tracer_one_pass_wanted(pl) && tracer_one_pass(pl)) {
    if ((rc = temx_tracer_stage1_sums(pl, q, va, wap, dtype, pl->Bq.d(), stream))) return rc;
    if ((rc = temx_tracer_stage2_from_sums(pl, pl->Bq.d(), pl->Bq2.d(), stream))) return rc;
    return temx_tracer_stage3(pl, pl->Bq2.d(), tres, tzon, stream);
  }
  if ((rc = temx_tracer_stage1(pl, q, dtype, pl->Bq.d(), stream))) return rc;
  if ((rc = temx_tracer_stage2(pl, q, va, wap, dtype, pl->Bq.d(), pl->Bq2.d(), stream))) return rc;
  return temx_tracer_stage3(pl, pl->Bq2.d(), tres, tzon, stream);
} TEMX_CATCH

// nq tracers of one TEM run (tem_diagnostics.py:281-301 takes a list): after a single-sweep TEM run they are swept
// in PAIRS -- (q1, q2, v, omega) read once: the traffic of the TEM sweep for two tracers instead of 3/4 of it for each
// -- and a last odd one alone; on any other path one temx_tracer_run after the other.
int temx_tracers_run(temx_plan* pl, int nq, const void* const* q_host, const void* va, const void* wap, int dtype,
                     double* const* tres_host, double* const* tzon_host, void* stream) try {
  int rc = tem_ready(pl);
  if (rc) return rc;
  if (nq < 1 || !q_host || !tres_host) return fail(TEMX_EINVAL, "bad argument");
  for (int i = 0; i < nq; ++i)
    if (!q_host[i] || !tres_host[i]) return fail(TEMX_EINVAL, "null argument");
  const bool os = pl->os_valid && pl->c4_valid && tail_is_whole(pl) && os_active(pl, dtype) && !pl->os_tile && !pl->osc_lds &&
                  (dtype == TEMX_F64 || dtype == TEMX_F32);
  int i = 0;
  if (os) {
    if (!va || !wap) return fail(TEMX_EINVAL, "null argument");
    if ((rc = os_ready(pl))) return rc;
    HIPCHK(hipSetDevice(pl->device));
    for (; i + 1 < nq; i += 2)
      if ((rc = tracer_run_os(pl, 2, q_host + i, va, wap, dtype, tres_host + i, tzon_host ? tzon_host + i : nullptr, stream))) return rc;
  }
  for (; i < nq; ++i)
    if ((rc = temx_tracer_run(pl, q_host[i], va, wap, dtype, tres_host[i], tzon_host ? tzon_host[i] : nullptr, stream))) return rc;
  return TEMX_OK;
} TEMX_CATCH

int temx_tracer_eddy(temx_plan* pl, const void* q, const void* va, const void* wap, int dtype,
                     double* const* ptrs3_host, void* stream) try {
  int rc = tem_ready(pl);
  if (rc) return rc;
  if (!q || !va || !wap || !ptrs3_host) return fail(TEMX_EINVAL, "null argument");
  if (!pl->Ct.p) return fail(TEMX_ESTATE, "temx_tracer_stage2 has not been called");
  HIPCHK(hipSetDevice(pl->device));
  if (!tail_is_whole(pl)) return fail(TEMX_ESTATE, "the plan holds the coefficients of a time slice (temx_tem_os_tail), not of the whole run");
  EddyOut eo{};
  eo.p[0] = ptrs3_host[0];
  eo.p[4] = ptrs3_host[1];
  eo.p[5] = ptrs3_host[2];
  if (unfused_stage2(pl)) {
    if (!pl->xb_valid) return fail(TEMX_ESTATE, "temx_tracer_stage2 has not been called since the last TEM run");
    return launch_eddy_from_xbar(pl, four(q, va, va, wap), dtype, native_means(pl, 4, 1, 1, 3), nullptr, eo,
                                 S_(stream));
  }
  return run_eddy<1>(pl, four(q, va, wap, nullptr), dtype, pl->Ct.d(), nullptr, &eo, S_(stream));
} TEMX_CATCH

int temx_status(temx_plan* pl, int* nonfinite, void* stream) try {
  if (!pl || !nonfinite) return fail(TEMX_EINVAL, "null argument");
  HIPCHK(hipSetDevice(pl->device));
  HIPCHK(hipStreamSynchronize(S_(stream)));
  int f = 0;
  HIPCHK(hipMemcpy(&f, pl->flag.p, sizeof(int), hipMemcpyDeviceToHost));
  *nonfinite = f;
  if (f) {
    int zero = 0;
    HIPCHK(hipMemcpy(pl->flag.p, &zero, sizeof(int), hipMemcpyHostToDevice));
  }
  return TEMX_OK;
} TEMX_CATCH

// ---- measurement helpers -------------------------------------------------------------------------
int temx_synth_fields(int device, int64_t ncol, int nlev, int64_t nt, int64_t t0, const double* lat_deg,
                      const double* lon_deg, const double* plev_hpa, int dtype, uint64_t seed, void* ua,
                      void* va, void* ta, void* wap, void* stream) try {
  if (!lat_deg || !lon_deg || !plev_hpa || !ua || !va || !ta || !wap) return fail(TEMX_EINVAL, "null argument");
  HIPCHK(hipSetDevice(device));
  dim3 grid(256 * 16), block(256);
  if (dtype == TEMX_F64) {
    hipLaunchKernelGGL(synth_kernel<double>, grid, block, 0, S_(stream), ncol, nlev, nt, t0, lat_deg, lon_deg,
                       plev_hpa, seed, (double*)ua, (double*)va, (double*)ta, (double*)wap);
  } else if (dtype == TEMX_F32) {
    hipLaunchKernelGGL(synth_kernel<float>, grid, block, 0, S_(stream), ncol, nlev, nt, t0, lat_deg, lon_deg,
                       plev_hpa, seed, (float*)ua, (float*)va, (float*)ta, (float*)wap);
  } else {
    return fail(TEMX_EINVAL, "dtype must be TEMX_F64 or TEMX_F32");
  }
  HIPCHK(hipGetLastError());
  return TEMX_OK;
} TEMX_CATCH

int temx_mfma_f64_peak(int device, int iters, double* tflops_out) try {
  if (!tflops_out || iters < 1) return fail(TEMX_EINVAL, "bad argument");
  HIPCHK(hipSetDevice(device));
  hipDeviceProp_t prop;
  HIPCHK(hipGetDeviceProperties(&prop, device));
  double* sink = nullptr;
  HIPCHK(hipMalloc(&sink, 8));
  const int blocks = prop.multiProcessorCount * 2;  // 8 waves per CU, 2 per SIMD
  hipEvent_t a = nullptr, b = nullptr;
  float ms = 0.f;
  hipError_t e = hipEventCreate(&a);
  if (e == hipSuccess) e = hipEventCreate(&b);
  if (e == hipSuccess) {
    hipLaunchKernelGGL(mfma_f64_peak_kernel, dim3(blocks), dim3(256), 0, 0, iters, sink);  // warm-up
    e = hipEventRecord(a, 0);
  }
  if (e == hipSuccess) {
    hipLaunchKernelGGL(mfma_f64_peak_kernel, dim3(blocks), dim3(256), 0, 0, iters, sink);
    e = hipEventRecord(b, 0);
  }
  if (e == hipSuccess) e = hipEventSynchronize(b);
  if (e == hipSuccess) e = hipEventElapsedTime(&ms, a, b);
  if (a) (void)hipEventDestroy(a);
  if (b) (void)hipEventDestroy(b);
  (void)hipFree(sink);
  if (e != hipSuccess) return fail(TEMX_EHIP, "mfma peak benchmark failed: %s", hipGetErrorString(e));
  const double flops = (double)blocks * 4.0 * iters * 8.0 * 512.0;
  *tflops_out = flops / (ms * 1e-3) / 1e12;
  return TEMX_OK;
} TEMX_CATCH

int temx_kernel_timing(temx_plan* pl, int enable) try {
  if (!pl) return fail(TEMX_EINVAL, "null plan");
  HIPCHK(hipSetDevice(pl->device));
  pl->timing = enable != 0;
  for (int w = 0; w < 2; ++w) {
    for (auto& tl : pl->timed[w]) {
      (void)hipEventDestroy(tl.a);
      (void)hipEventDestroy(tl.b);
    }
    pl->timed[w].clear();
  }
  return TEMX_OK;
} TEMX_CATCH

int temx_kernel_timing_read(temx_plan* pl, int which, double* avg_ms, int* launches) try {
  if (!pl || !avg_ms || !launches || which < 0 || which > 1) return fail(TEMX_EINVAL, "bad argument");
  HIPCHK(hipSetDevice(pl->device));
  double tot = 0.0;
  int n = 0;
  for (auto& tl : pl->timed[which]) {
    HIPCHK(hipEventSynchronize(tl.b));
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, tl.a, tl.b));
    tot += ms;
    ++n;
  }
  *avg_ms = n ? tot / n : 0.0;
  *launches = n;
  return TEMX_OK;
} TEMX_CATCH

int temx_selftest_exception(int kind) try {
  if (kind == 0) throw std::bad_alloc();
  if (kind == 1) throw std::runtime_error("self-test");
  if (kind == 2) throw 42;
  return fail(TEMX_EINVAL, "kind must be 0, 1 or 2");
} TEMX_CATCH

}  // extern "C"
