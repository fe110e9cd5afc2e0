// kernels.hpp -- gfx950 (CDNA4) device code of libtemx.  Written for MI355X only.
//
// Vocabulary (reference: PyTEMDiags/sph_zonal_mean.py, tem_diagnostics.py):
//   N  = ncol native columns, K = L+1 harmonics, M zonal-grid latitudes, D = nlev*nt,
//   field  = [N][D] row-major,  group = 4 consecutive native columns, chunk = 4 groups,
//   d-tile = 16 consecutive (lev,time) columns, l-block = 4 consecutive harmonics (TB per row).
//
// Matrix instruction: v_mfma_f64_4x4x4_4b_f64 (4 independent 4x4x4 blocks per wave).  Measured on
// MI355X (tools/ubench_f64.hip): 18 cycles/instruction/SIMD = 67-68 TFLOP/s already at one wave
// per SIMD, against 105 cycles (48 TFLOP/s) for v_mfma_f64_16x16x4_f64 and 57 TFLOP/s for
// v_fma_f64 -- so this is the fp64 primitive the sweeps are tiled for.  Lane maps, probed with
// one-hot operands (tools/probe_mfma4.hip):
//   A: lane = 16 k + 4 b + i      B: lane = 16 k + 4 b + j      D: lane = 16 i + 4 b + j
// Mapping block b to column group 4b..4b+3 makes one instruction a [4 x 4] . [4 x 16] product:
//   B / D operands: lane holds element [row = lane>>4][col = lane&15] of a 4 x 16 tile -- the
//     natural coalesced layout of 4 rows x 128 B of a field, and a D tile is directly the B
//     operand of a following product that contracts over its rows;
//   A operand: a 4 x 4 block, element [i = lane&3][k = lane>>4], replicated over the 4 blocks:
//     16 doubles = one 128-byte line, read by every lane as a broadcast.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#define TEMX_MFMA4(a, b, c) __builtin_amdgcn_mfma_f64_4x4x4f64((a), (b), (c), 0, 0, 0)

namespace temx {

template <int NF>
struct FieldPtrs {
  const void* p[NF];
};

struct EddyOut {
  double* p[7];
};

// ------------------------------------------------------------------------------------------------
// ylm0_basis: replaces the scipy.special.sph_harm loops (sph_zonal_mean.py:360-363, 367-370).
// Y_l^0 = sqrt((2l+1)/4pi) P_l(x), x = cos(colat); l P_l = (2l-1) x P_{l-1} - (l-1) P_{l-2}.
// One thread per native column; writes the canonical row-major matrix and the 4x4-blocked copy
// the sweeps stream as MFMA A operands:
//   yblk[group][t][k*4 + i] = Y0[4 group + k][4 t + i]        (t < TB, 128 B per block)
// Rows >= N and harmonics >= K are zero, so tails need no masking in the sweeps.
// rowscale (weights mode, sph_zonal_mean.py:385) scales the blocked copy only.
// ------------------------------------------------------------------------------------------------
// One row of the projection basis at x = cos(colat):  q[l] = Y_l^0(x) for l < K, or -- with T, the
// inverse of the Cholesky factor of the Gram matrix (upper triangular, row-major K x K) --
//     q[j] = sum_{l <= j} Y_l^0(x) T[l][j],
// the row of Q = Y0 R^-1 (temx_plan_finalize: the sweeps then project on an orthonormal basis and the
// K x K solve is the identity up to rounding).  In place from the highest column down: q[j] needs only l <= j.
template <int KMAX>
__device__ __forceinline__ void basis_row(double xv, int K, const double* __restrict__ norm,
                                          const double* __restrict__ T, double* q) {
  double pm1 = 1.0, pc = xv;
  for (int l = 0; l < K; ++l) {
    double P;
    if (l == 0) {
      P = 1.0;
    } else if (l == 1) {
      P = xv;
    } else {
      double pn = ((2 * l - 1) * xv * pc - (l - 1) * pm1) / l;
      pm1 = pc;
      pc = pn;
      P = pn;
    }
    q[l] = norm[l] * P;
  }
  if (T != nullptr) {
    for (int j = K - 1; j >= 0; --j) {
      double a = 0.0;
      for (int l = 0; l <= j; ++l) a += q[l] * T[(int64_t)l * K + j];
      q[j] = a;
    }
  }
}

template <int KMAX>
__global__ void basis_kernel(const double* __restrict__ x, int64_t N, int64_t nrow_pad, int K, int TB /* blocks stored per group */,
                             const double* __restrict__ norm, const double* __restrict__ rowscale,
                             const double* __restrict__ T, double* __restrict__ Y0, double* __restrict__ yblk) {
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= nrow_pad) return;
  const bool valid = i < N;
  const double xv = valid ? x[i] : 0.0;
  const double rs = (valid && rowscale) ? rowscale[i] : 1.0;
  const int64_t group = i >> 2;
  const int k = (int)(i & 3);
  double q[KMAX];
  basis_row<KMAX>(xv, K, norm, T, q);
  for (int l = 0; l < 4 * TB; ++l) {
    const double val = (valid && l < K) ? q[l] : 0.0;
    if (Y0 && valid && l < K) Y0[i * K + l] = val;
    if (yblk) yblk[((group * TB + (l >> 2)) * 16) + k * 4 + (l & 3)] = val * rs;
  }
}

// wave-work decomposition shared by the sweeps: work id -> (split over chunks, d-tile).
// Workgroups are dealt to XCDs round-robin (blockIdx % 8); remapping so that each XCD owns a
// contiguous run of work ids keeps the d-tiles of one chunk range (which stream the same
// Y0 blocks) behind one L2.
__device__ __forceinline__ int uniform_wave() {
  return __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));   // SGPR: addresses stay scalar
}

// workgroup-work decomposition shared by the sweeps: work id -> (split over chunks, group of
// d-tiles).  The waves of a workgroup walk the same chunk range of the same group of d-tiles.
// Workgroups are dealt to XCDs round-robin
// (blockIdx % 8); remapping so that each XCD owns a contiguous run of work ids keeps the groups of
// one chunk range (which stream the same Y0 blocks) behind one L2.
__device__ __forceinline__ bool wg_work(int ndq, int nsplit, int& split, int& dq) {
  const int cpx = gridDim.x >> 3;
  const int w = (blockIdx.x & 7) * cpx + (blockIdx.x >> 3);
  if (w >= ndq * nsplit) return false;
  split = w / ndq;
  dq = w % ndq;
  return true;
}

// ------------------------------------------------------------------------------------------------
// project sweep: partial[split][f][l][d] = sum_{i in split} Y0[i][l] * X_f[i][d]
// Replaces the inner np.matmul(Y0inv, AA) of sph_zonal_mean.py:251 (reduction over ncol); the
// G^-1 factor is applied afterwards on the K x D sums (solve_kernel).  For the TEM pipeline NF=4
// with theta = T (p0/p)^kappa fused into the load of field `sfield` (tem_diagnostics.py:498).
// One wave owns one d-tile, all TB l-blocks and NFW of the NF fields (NFW*TB accumulator
// registers).  X is read exactly once from HBM in the MFMA B layout, straight into registers, PD
// chunks ahead (re-loaded into the registers just consumed).  The Y0 blocks of a chunk are staged
// through LDS by the whole workgroup (double buffered, one barrier per chunk): loaded per operand
// they would make every wait also wait for the youngest HBM loads (vector loads retire in order).
// ------------------------------------------------------------------------------------------------
template <typename T, int NF, int NFW, int TB, int PD, int WPS>
__global__ void __launch_bounds__(256, WPS)
project_kernel(FieldPtrs<NF> fp, int64_t N, int64_t D, int K, const double* __restrict__ yblk,
               int gstride /* blocks stored per group */, int tb_off /* first block of this harmonic slice */,
               int64_t nchunk, const double* __restrict__ colscale, int sfield,
               double* __restrict__ partial, int nsplit, int ndt) {
  // NFW fields per wave: a workgroup's 4 waves cover DPW = 4*NFW/NF d-tiles x NF/NFW field groups.
  // Fewer fields per wave = fewer accumulator registers = room for a deeper X ring and a third
  // wave per SIMD (more independent instruction streams to overlap HBM latency with MFMA issue).
  constexpr int DPW = 4 * NFW / NF;
  // Y0 blocks of one chunk (4 groups x TB blocks x 16), double buffered.  They come through LDS
  // rather than straight into registers because vector loads retire in order: an operand load
  // issued every few MFMAs would make every wait also wait for the youngest HBM loads of X.
  __shared__ double ystage[2][4 * TB * 16];
  int split, dq;
  if (!wg_work((ndt + DPW - 1) / DPW, nsplit, split, dq)) return;
  const int wave = uniform_wave();
  const int tid = threadIdx.x, lane = tid & 63;
  const int c = lane & 15, g = lane >> 4;
  const int dt = dq * DPW + wave % DPW;
  const int f0 = (wave / DPW) * NFW;                   // first field of this wave
  const bool active = dt < ndt;                        // ragged last workgroup: helper waves only stage
  const int64_t d = (int64_t)dt * 16 + c;
  const bool dvalid = active && d < D;
  const int64_t dcl = d < D ? d : D - 1;
  const int c0 = (int)(nchunk * split / nsplit), c1 = (int)(nchunk * (split + 1) / nsplit);   // uniform

  // addressing: wave-uniform base (SGPR) + one 32-bit lane offset
  const uint32_t loff = (uint32_t)(g * D + dcl);          // host guarantees 4*D < 2^31
  const uint32_t yoff = (uint32_t)(g * 4 + (lane & 3));   // A[i = lane&3 -> l][k = lane>>4 -> row]
  double sc[NFW];
  const T* fb[NFW];
#pragma unroll
  for (int f = 0; f < NFW; ++f) {
    sc[f] = (colscale != nullptr && f0 + f == sfield) ? colscale[dcl] : 1.0;
    fb[f] = reinterpret_cast<const T*>(fp.p[f0 + f]);
  }

  double acc[NFW][TB];
#pragma unroll
  for (int f = 0; f < NFW; ++f)
#pragma unroll
    for (int t = 0; t < TB; ++t) acc[f][t] = 0.0;

  constexpr int YE = 4 * TB * 16;            // doubles of Y0 blocks per chunk
  constexpr int YJ = (YE + 255) / 256;       // staging loads per thread
  // X runs PD chunks ahead in a register ring (HBM latency under load is longer than the MFMA
  // time of one chunk); the chunk loop is unrolled by PD so ring slots are compile-time.
  T xn[PD][NFW][4];
  double ys[YJ];
  const int nfull = (int)(N >> 4);           // chunks whose 16 rows all exist
  auto load_x = [&](int chunk, auto slotc, int ti, auto fastc) __attribute__((always_inline)) {
    constexpr int slot = decltype(slotc)::value;
    const int64_t gb = (int64_t)chunk * 16 + ti * 4;   // first row of the group (uniform)
    if (decltype(fastc)::value) {
#pragma unroll
      for (int f = 0; f < NFW; ++f) xn[slot][f][ti] = (fb[f] + gb * D)[loff];
    } else {                                  // ragged tail of the grid: clamp per lane
      int64_t row = gb + g;
      row = row < N ? row : N - 1;
#pragma unroll
      for (int f = 0; f < NFW; ++f) xn[slot][f][ti] = fb[f][row * D + dcl];
    }
  };
  // element tid + 256 j of a chunk's staging image = (group gi, block t, element e); in yblk the
  // groups of a chunk are gstride blocks apart and this slice starts at block tb_off (gstride == TB,
  // tb_off == 0 when all harmonics fit one slice).  The per-lane offsets are loop invariant.
  uint32_t yso[YJ];
#pragma unroll
  for (int j = 0; j < YJ; ++j) {
    const int li = tid + 256 * j;
    const int gi = li / (TB * 16), rem = li % (TB * 16);
    yso[j] = (uint32_t)((gi < 4 ? gi : 3) * gstride * 16 + tb_off * 16 + rem);
  }
  const int64_t ychunk = (int64_t)4 * gstride * 16;
  auto load_ys = [&](int chunk) __attribute__((always_inline)) {   // yblk is padded: never leaves it
#pragma unroll
    for (int j = 0; j < YJ; ++j) ys[j] = (yblk + (int64_t)chunk * ychunk)[yso[j]];
  };
  // one chunk: stage Y0 blocks, barrier, prefetch Y of chunk+1 and X of chunk+PD,
  // 4 groups x TB x NFW MFMAs
  auto do_chunk = [&](int chunk, auto slotc, auto fastc) __attribute__((always_inline)) {
    constexpr bool FAST = decltype(fastc)::value;     // FAST: chunk+PD < c1 and all its rows exist
    constexpr int slot = decltype(slotc)::value;
    double* yst = ystage[(chunk - c0) & 1];
#pragma unroll
    for (int j = 0; j < YJ; ++j)
      if (tid + 256 * j < YE) yst[tid + 256 * j] = ys[j];
    __syncthreads();
    if (FAST || chunk + 1 < c1) load_ys(chunk + 1);
    const bool more = FAST || chunk + PD < c1;
    if (active) {
#pragma unroll
      for (int ti = 0; ti < 4; ++ti) {
        double xs[NFW];
#pragma unroll
        for (int f = 0; f < NFW; ++f) xs[f] = (double)xn[slot][f][ti] * sc[f];
        if (more) load_x(chunk + PD, slotc, ti, fastc);
#pragma unroll
        for (int t = 0; t < TB; ++t) {
          const double ya = yst[(ti * TB + t) * 16 + yoff];
#pragma unroll
          for (int f = 0; f < NFW; ++f) acc[f][t] = TEMX_MFMA4(ya, xs[f], acc[f][t]);
        }
      }
    }
  };
  // PD consecutive chunks, ring slot = position in the unrolled group
  auto do_group = [&](int chunk, auto fastc) __attribute__((always_inline)) {
    do_chunk(chunk, std::integral_constant<int, 0>{}, fastc);
    if (PD > 1 && (decltype(fastc)::value || chunk + 1 < c1))
      do_chunk(chunk + 1, std::integral_constant<int, (PD > 1 ? 1 : 0)>{}, fastc);
    if (PD > 2 && (decltype(fastc)::value || chunk + 2 < c1))
      do_chunk(chunk + 2, std::integral_constant<int, (PD > 2 ? 2 : 0)>{}, fastc);
  };

  if (c0 < c1) {
    load_ys(c0);
    if (active) {
#pragma unroll
      for (int ti = 0; ti < 4; ++ti) load_x(c0, std::integral_constant<int, 0>{}, ti, std::false_type{});
      if (PD > 1 && c0 + 1 < c1) {
#pragma unroll
        for (int ti = 0; ti < 4; ++ti)
          load_x(c0 + 1, std::integral_constant<int, (PD > 1 ? 1 : 0)>{}, ti, std::false_type{});
      }
      if (PD > 2 && c0 + 2 < c1) {
#pragma unroll
        for (int ti = 0; ti < 4; ++ti)
          load_x(c0 + 2, std::integral_constant<int, (PD > 2 ? 2 : 0)>{}, ti, std::false_type{});
      }
    }
  }
  // chunks c with c + 2*PD - 1 < min(c1, nfull) run whole groups on the clamp-free path
  const int cfast = (c1 < nfull ? c1 : nfull) - (2 * PD - 1);
  int chunk = c0;
  for (; chunk < cfast; chunk += PD) do_group(chunk, std::true_type{});
  for (; chunk < c1; chunk += PD) do_group(chunk, std::false_type{});

  if (dvalid) {
#pragma unroll
    for (int f = 0; f < NFW; ++f)
#pragma unroll
      for (int t = 0; t < TB; ++t) {
        const int l = t * 4 + g;
        if (l < K) partial[(((int64_t)split * NF + f0 + f) * K + l) * D + d] = acc[f][t];
      }
  }
}

// Deterministic sum of the per-split slabs; flags non-finite sums (NaN inputs).
// A block owns 16 consecutive entries; its 256 threads are 16 entries x 16 split lanes, so small
// problems (few entries, hundreds of splits) still spread over hundreds of blocks.  Each split lane
// sums its splits in ascending order, the 16 lane sums are combined in a fixed order.
// Slab sp starts at partial + sp * stride (stride >= n: a sweep may interleave other slabs);
// `addend` (NULL or [n]) is added last.
//
// Where the sum of entry idx goes.  The entries are rows of [nlev][nt] columns (time fastest).  An ncol-sharded job
// exchanges them by a reduce-scatter over TIME -- everything after the zonal sums (solve, contraction, vertical
// stencils, epilogue; tem_diagnostics.py:574-797) acts along latitude and pressure only, so a rank can finish the
// snapshots it receives on its own.  The reduction therefore writes the layout the collective wants,
//   out[w][row0 + row][lev][t - t0(w)],   chunk w = `chunk` doubles, rows of nlev x ntw(w) columns inside it,
// the snapshots cut like sharding.shard_bounds (the first nt % W ranks hold one more).  D == 0: out[idx] (no map).
struct SliceMap {
  int64_t D = 0;       // columns per row (nlev * nt); 0 = identity
  int nt = 1, W = 1;   // snapshots, slices
  int64_t chunk = 0;   // doubles per slice of the output
  int64_t row0 = 0;    // first row of this reduction inside a slice
};
__device__ __forceinline__ int64_t slice_index(const SliceMap& m, int64_t idx) {
  if (m.D == 0) return idx;
  const int64_t row = idx / m.D, d = idx - row * m.D;
  const int lev = (int)(d / m.nt), t = (int)(d - (int64_t)lev * m.nt);
  const int base = m.nt / m.W, extra = m.nt - base * m.W, tb = extra * (base + 1);
  int w, tl, ntw;
  if (t < tb) { w = t / (base + 1); tl = t - w * (base + 1); ntw = base + 1; }
  else { const int u = t - tb; w = extra + u / base; tl = u - (w - extra) * base; ntw = base; }
  const int nlev = (int)(m.D / m.nt);
  return (int64_t)w * m.chunk + ((m.row0 + row) * nlev + lev) * ntw + tl;
}

__global__ void __launch_bounds__(256)
reduce_partials_kernel(const double* __restrict__ partial, int nsplit, int64_t stride, int64_t n,
                       const double* __restrict__ addend, double* __restrict__ B, int* __restrict__ flag,
                       SliceMap map) {
  __shared__ double sh[16][17];
  const int e = threadIdx.x & 15, sl = threadIdx.x >> 4;
  const int64_t idx = (int64_t)blockIdx.x * 16 + e;
  double s = 0.0;
  if (idx < n)
    for (int sp = sl; sp < nsplit; sp += 16) s += partial[(int64_t)sp * stride + idx];
  sh[sl][e] = s;
  __syncthreads();
  if (sl == 0 && idx < n) {
    double t = 0.0;
#pragma unroll
    for (int j = 0; j < 16; ++j) t += sh[j][e];
    if (addend != nullptr) t += addend[idx];
    B[slice_index(map, idx)] = t;
    if (!(fabs(t) <= 1.79769313486231570815e308)) atomicOr(flag, 1);
  }
}

// Many entries, few slabs (nsplit <= NS <= 16): one thread per entry, slabs summed in ascending order --
// the same association as the kernel above (each of its 16 split lanes then holds at most one slab),
// so both give identical bits.  NS = 2, 4, 8 or 16 slots are loaded (a slot past nsplit re-reads slab 0 and adds
// zero): with 16 slots whatever nsplit, a two-slab sum at D = 6552 spent 32 us on fourteen redundant loads per entry.
template <int NS>
__global__ void __launch_bounds__(256)
reduce_partials_flat_kernel(const double* __restrict__ partial, int nsplit, int64_t stride, int64_t n,
                            const double* __restrict__ addend, double* __restrict__ B, int* __restrict__ flag,
                            SliceMap map) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= n) return;
  double v[NS];
#pragma unroll
  for (int sp = 0; sp < NS; ++sp) v[sp] = partial[(int64_t)(sp < nsplit ? sp : 0) * stride + idx];
  double t = 0.0;
#pragma unroll
  for (int sp = 0; sp < NS; ++sp) t += sp < nsplit ? v[sp] : 0.0;
  if (addend != nullptr) t += addend[idx];
  B[slice_index(map, idx)] = t;
  if (!(fabs(t) <= 1.79769313486231570815e308)) atomicOr(flag, 1);
}

// The first `rows` rows of each of nf blocks of `rows_in` rows: out[f][r][d] = sum over the slabs of
// partial[sp * stride + (f * rows_in + r) * D + d].  (The reference pre-pass of the single sweep needs the
// projections on the first KR harmonics only; ascending slab order, as above.)
__global__ void __launch_bounds__(256)
reduce_rows_kernel(const double* __restrict__ partial, int nsplit, int64_t stride, int nf, int rows_in, int rows,
                   int64_t D, double* __restrict__ out, int* __restrict__ flag) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t per = (int64_t)rows * D;
  if (idx >= per * nf) return;
  const int64_t f = idx / per, rem = idx - f * per;
  const double* src = partial + f * rows_in * D + rem;
  double t = 0.0;
  for (int sp = 0; sp < nsplit; ++sp) t += src[(int64_t)sp * stride];
  out[idx] = t;
  if (!(fabs(t) <= 1.79769313486231570815e308)) atomicOr(flag, 1);
}
// The same sums for many slabs (nsplit > 16: small D, where the pre-pass is cut into many row ranges): sixteen
// threads per entry, each adding every 16th slab, then the sixteen partial sums in ascending order -- as
// reduce_partials_kernel does.  (One thread per entry is a chain of nsplit dependent adds: 26 us at nsplit = 128.)
__global__ void __launch_bounds__(256)
reduce_rows_wide_kernel(const double* __restrict__ partial, int nsplit, int64_t stride, int nf, int rows_in, int rows,
                        int64_t D, double* __restrict__ out, int* __restrict__ flag) {
  __shared__ double sh[16][17];
  const int e = threadIdx.x & 15, sl = threadIdx.x >> 4;
  const int64_t idx = (int64_t)blockIdx.x * 16 + e;
  const int64_t per = (int64_t)rows * D;
  const bool ok = idx < per * nf;
  double s = 0.0;
  if (ok) {
    const int64_t f = idx / per, rem = idx - f * per;
    const double* src = partial + f * rows_in * D + rem;
    for (int sp = sl; sp < nsplit; sp += 16) s += src[(int64_t)sp * stride];
  }
  sh[sl][e] = s;
  __syncthreads();
  if (sl == 0 && ok) {
    double t = 0.0;
#pragma unroll
    for (int j = 0; j < 16; ++j) t += sh[j][e];
    out[idx] = t;
    if (!(fabs(t) <= 1.79769313486231570815e308)) atomicOr(flag, 1);
  }
}

// ------------------------------------------------------------------------------------------------
// solve: C = Ginv . B (harmonic coefficients, pinv(Y0) A = G^-1 Y0^T A, replaces the lstsq of
// sph_zonal_mean.py:389) and Xb = Y0p . C (the outer matmul with Y = Y0p of :251).
// One block per (16 columns, field, slice of the M output latitudes); every slice recomputes the
// K x 16 coefficients (cheap) so that small problems still fill the chip.  C is stored with
// K4 = 4*TB rows (rows >= K zero) by slice 0.
// ------------------------------------------------------------------------------------------------
// This scalar version serves K > 64 (64 < K <= 512); K <= 64 uses solve_mfma_kernel below.
__global__ void __launch_bounds__(1024)
solve_kernel(const double* __restrict__ B, int K, int K4, int M, int64_t D,
             const double* __restrict__ Ginv, const double* __restrict__ Y0p,
             double* __restrict__ C, double* __restrict__ Xb) {
  extern __shared__ double slds[];          // sb[K4][17], scf[K4][17]
  double* sb = slds;
  double* scf = slds + (size_t)K4 * 17;
  const int f = blockIdx.y;
  const int64_t d0 = (int64_t)blockIdx.x * 16;
  const int tid = threadIdx.x;
  const int mper = (M + gridDim.z - 1) / gridDim.z;
  const int m0 = blockIdx.z * mper;
  const int m1 = m0 + mper < M ? m0 + mper : M;
  for (int idx = tid; idx < K * 16; idx += 1024) {
    const int k = idx >> 4, dd = idx & 15;
    const int64_t d = d0 + dd;
    sb[k * 17 + dd] = d < D ? B[((int64_t)f * K + k) * D + d] : 0.0;
  }
  __syncthreads();
  for (int idx = tid; idx < K4 * 16; idx += 1024) {
    const int k = idx >> 4, dd = idx & 15;
    const int64_t d = d0 + dd;
    double v = 0.0;
    if (k < K) {
      const double* gr = Ginv + (int64_t)k * K;
      double v0 = 0.0, v1 = 0.0, v2 = 0.0, v3 = 0.0;      // four chains: the dot is latency bound
      int kk = 0;
      for (; kk + 4 <= K; kk += 4) {
        v0 += gr[kk] * sb[kk * 17 + dd];
        v1 += gr[kk + 1] * sb[(kk + 1) * 17 + dd];
        v2 += gr[kk + 2] * sb[(kk + 2) * 17 + dd];
        v3 += gr[kk + 3] * sb[(kk + 3) * 17 + dd];
      }
      for (; kk < K; ++kk) v0 += gr[kk] * sb[kk * 17 + dd];
      v = (v0 + v1) + (v2 + v3);
    }
    scf[k * 17 + dd] = v;
    if (C != nullptr && blockIdx.z == 0 && d < D) C[((int64_t)f * K4 + k) * D + d] = v;
  }
  __syncthreads();
  if (Xb != nullptr) {
    for (int idx = tid; idx < (m1 - m0) * 16; idx += 1024) {
      const int m = m0 + (idx >> 4), dd = idx & 15;
      const int64_t d = d0 + dd;
      if (d >= D) continue;
      const double* yr = Y0p + (int64_t)m * K;
      double v0 = 0.0, v1 = 0.0, v2 = 0.0, v3 = 0.0;
      int kk = 0;
      for (; kk + 4 <= K; kk += 4) {
        v0 += yr[kk] * scf[kk * 17 + dd];
        v1 += yr[kk + 1] * scf[(kk + 1) * 17 + dd];
        v2 += yr[kk + 2] * scf[(kk + 2) * 17 + dd];
        v3 += yr[kk + 3] * scf[(kk + 3) * 17 + dd];
      }
      for (; kk < K; ++kk) v0 += yr[kk] * scf[kk * 17 + dd];
      Xb[((int64_t)f * M + m) * D + d] = (v0 + v1) + (v2 + v3);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// solve, K <= 64: C = Ginv . B and Xb = Y0p . C as two small MFMA GEMMs per d-tile.  One wave = one
// d-tile (16 columns) of one field: the B tile is loaded straight into the MFMA B layout, Ginv and
// this block's slice of Y0p are staged in LDS as 4x4 blocks (block[k*4+i] = A[4r+i][4t+k], built
// on the host), the C tile comes out in the B layout and feeds the second product with no lane
// movement.  gridDim.z slices the output latitudes (16 blocks of 4 per slice); every slice
// recomputes C (TB*TB MFMAs).
// ------------------------------------------------------------------------------------------------
constexpr int SOLVE_MB = 16;          // 4-row blocks of output latitudes per z-slice

template <int TB>
__global__ void __launch_bounds__(256)
solve_mfma_kernel(const double* __restrict__ B, int K, int M, int64_t D, const double* __restrict__ gblk,
                  const double* __restrict__ ypblk, double* __restrict__ C, double* __restrict__ Xb,
                  int mbs /* 4-row blocks of output latitudes per z-slice, <= SOLVE_MB */) {
  extern __shared__ double slds[];          // sg[TB*TB*16], sy[SOLVE_MB*TB*16]
  double* sg = slds;
  double* sy = slds + TB * TB * 16;
  const int f = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 4, c = lane & 15;
  const int64_t d = ((int64_t)blockIdx.x * 4 + wave) * 16 + c;
  const bool dvalid = d < D;
  const int64_t dcl = dvalid ? d : D - 1;
  const int mb0 = blockIdx.z * mbs;
  const int nmb_all = (M + 3) >> 2;
  const int nmb = Xb != nullptr ? (nmb_all - mb0 < mbs ? nmb_all - mb0 : mbs) : 0;

  double breg[TB];
#pragma unroll
  for (int t = 0; t < TB; ++t) {            // branch free: clamped row + select
    const int row = 4 * t + g;
    const int rc = row < K ? row : K - 1;
    const double v = B[((int64_t)f * K + rc) * D + dcl];
    breg[t] = row < K ? v : 0.0;
  }
  {   // staging: every global load is issued before the first LDS store (one round trip, not one per element)
    constexpr int JG = (TB * TB * 16 + 255) / 256, JY = (SOLVE_MB * TB * 16 + 255) / 256;
    double tg[JG], ty[JY];
    const double* yp = ypblk + (int64_t)mb0 * TB * 16;
    const int ny = nmb * TB * 16;
#pragma unroll
    for (int j = 0; j < JG; ++j) tg[j] = gblk[tid + 256 * j < TB * TB * 16 ? tid + 256 * j : 0];
#pragma unroll
    for (int j = 0; j < JY; ++j) ty[j] = yp[tid + 256 * j < ny ? tid + 256 * j : 0];
#pragma unroll
    for (int j = 0; j < JG; ++j)
      if (tid + 256 * j < TB * TB * 16) sg[tid + 256 * j] = tg[j];
#pragma unroll
    for (int j = 0; j < JY; ++j)
      if (tid + 256 * j < ny) sy[tid + 256 * j] = ty[j];
  }
  __syncthreads();

  const uint32_t yoff = (uint32_t)(g * 4 + (lane & 3));
  double creg[TB];
#pragma unroll
  for (int r = 0; r < TB; ++r) creg[r] = 0.0;
#pragma unroll
  for (int t = 0; t < TB; ++t)
#pragma unroll
    for (int r = 0; r < TB; ++r) creg[r] = TEMX_MFMA4(sg[(r * TB + t) * 16 + yoff], breg[t], creg[r]);
  if (C != nullptr && blockIdx.z == 0 && dvalid) {
#pragma unroll
    for (int r = 0; r < TB; ++r) C[((int64_t)f * (4 * TB) + 4 * r + g) * D + d] = creg[r];
  }
  for (int mb = 0; mb < nmb; mb += 4) {     // 4 independent accumulators per pass
    double acc[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int t = 0; t < TB; ++t)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int mbj = mb + j < nmb ? mb + j : nmb - 1;
        acc[j] = TEMX_MFMA4(sy[(mbj * TB + t) * 16 + yoff], creg[t], acc[j]);
      }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int m = 4 * (mb0 + mb + j) + g;
      if (mb + j < nmb && m < M && dvalid) Xb[((int64_t)f * M + m) * D + d] = acc[j];
    }
  }
}

// ------------------------------------------------------------------------------------------------
// eddy / flux sweep (the dominant kernel).  Per step (8 columns) x d-tile (16), one wave:
//   xbar_f = Y0[chunk] . C_f                (4 reconstructions; contraction over harmonics)
//            = sph_zonal_mean_native of tem_diagnostics.py:517-529, never stored
//   x'_f   = x_f - xbar_f                    (eddies; theta = T (p0/p)^kappa fused in the load)
//   u'v', u'w', v'theta'                     (tem_diagnostics.py:547-555)
//   partial[q][l][d] += Y0[chunk]^T . (product q)    (3 projections; contraction over columns)
// = 14*TB MFMAs per step.  The reconstruction result tile has the B-operand layout, so the products
// feed the projection with no lane movement.
// The coefficient B operands (4 fields x TB k-steps x 16 columns) are loop invariant per d-tile;
// they live in an LDS slab shared by the 8/DPW waves that split the step range of that d-tile.
// Each wave writes the whole slab itself before reading it (identical values from every writer),
// so no barrier is needed anywhere.
// MODE 1 additionally stores the eddies / products (lazy properties up, vp, ... of :420-433).
// ------------------------------------------------------------------------------------------------
constexpr int EDDY_GR = 2;   // groups (of 4 columns) per eddy step: 8 columns x 16 (lev,time)

// KIND 0: TEM      fields (u, v, T->theta, omega); products u'v', u'w', v'theta'
// KIND 1: tracer   fields (q, v, omega);           products q'v', q'w'   (tem_diagnostics.py:532-538, 560-570)
template <typename T, int TB, int MODE, int DPW, int KIND>
__global__ void __launch_bounds__(512, 2)
eddy_kernel(FieldPtrs<4> fp, int64_t N, int64_t D, int K, const double* __restrict__ yblk,
            int64_t nchunk, const double* __restrict__ colscale,
            const double* __restrict__ C, double* __restrict__ partial, int nsplit, int ndt,
            EddyOut eo) {
  // A workgroup (8 waves) owns DPW d-tiles (4, 2 or 1); the NP = 8/DPW waves on one d-tile share
  // its coefficient slab and split the chunk range NP ways (small D: fewer, evenly loaded tiles).
  // LDS: [DPW d-tiles][4 fields][TB][64] coefficient slabs, then one private copy per wave of the
  // Y0 blocks of the current step (GR x TB x 16), read by the reconstruction as
  // A[column][harmonic] and by the projection as A[harmonic][column].  Wave-private staging
  // (each wave re-reads the blocks from L2) keeps the kernel free of barriers: with one
  // 8-wave workgroup per CU a per-step barrier stalled both waves of every SIMD at once
  // (measured 22.3 -> 19.8 ms on ne120x72x30).  LDS operations of one wave execute in order, so
  // a single buffer is enough: the next step's blocks are written after this step's reads.
  extern __shared__ double lds[];
  constexpr int GR = EDDY_GR;
  constexpr int YE = GR * TB * 16;           // doubles of Y0 blocks per step
  constexpr int YJ = (YE + 63) / 64;         // staging loads per lane
  constexpr int NP = 8 / DPW;
  constexpr int NFR = KIND == 0 ? 4 : 3;     // fields reconstructed
  constexpr int NPR = KIND == 0 ? 3 : 2;     // products projected
  int split, dq;
  if (!wg_work((ndt + DPW - 1) / DPW, nsplit, split, dq)) return;
  const int wave = uniform_wave(), lane = threadIdx.x & 63;
  const int w4 = wave % DPW, part = wave / DPW;
  const int c = lane & 15, g = lane >> 4;
  const int dt = dq * DPW + w4;
  const bool active = dt < ndt;
  const int64_t d = (int64_t)dt * 16 + c;
  const bool dvalid = active && d < D;
  const int64_t dcl = d < D ? d : D - 1;
  const int64_t nstep = nchunk * (4 / GR);
  const int64_t sub = (int64_t)split * NP + part, nsub = (int64_t)nsplit * NP;
  const int c0 = (int)(nstep * sub / nsub), c1 = (int)(nstep * (sub + 1) / nsub);   // uniform (SGPR)

  // coefficient B operands: cb[f][s][lane] = C_f[4 s + g][d]; all NP waves on the d-tile write
  // identical values (no barrier needed)
  {
    double* cb = lds + (size_t)w4 * (NFR * TB * 64) + lane;
#pragma unroll
    for (int f = 0; f < NFR; ++f)
#pragma unroll
      for (int s = 0; s < TB; ++s) cb[(f * TB + s) * 64] = C[((int64_t)f * 4 * TB + 4 * s + g) * D + dcl];
  }
  int cbi = w4 * (NFR * TB * 64) + lane;   // index of this lane's first slab element in lds[]
  double* yst = lds + DPW * NFR * TB * 64 + wave * YE;   // this wave's copy of the step's Y0 blocks

  const double sth = (KIND == 0 && colscale != nullptr) ? colscale[dcl] : 1.0;
  const uint32_t loff = (uint32_t)(g * D + dcl);
  const uint32_t aoff_r = (uint32_t)((lane & 3) * 4 + g);   // reconstruction A[i -> column][k -> harmonic]
  const uint32_t aoff_p = (uint32_t)(g * 4 + (lane & 3));   // projection     A[i -> harmonic][k -> column]

  double acc[NPR][TB];
#pragma unroll
  for (int q = 0; q < NPR; ++q)
#pragma unroll
    for (int t = 0; t < TB; ++t) acc[q][t] = 0.0;

  T xn[NFR][GR];
  double ys[YJ];
  const int nfull = (int)(N / (4 * GR));     // steps whose rows all exist
  auto load_x = [&](int step, auto fastc) __attribute__((always_inline)) {
#pragma unroll
    for (int ti = 0; ti < GR; ++ti) {
      const int64_t gb = ((int64_t)step * GR + ti) * 4;
      if (decltype(fastc)::value) {
#pragma unroll
        for (int f = 0; f < NFR; ++f) xn[f][ti] = (reinterpret_cast<const T*>(fp.p[f]) + gb * D)[loff];
      } else {
        int64_t row = gb + g;
        row = row < N ? row : N - 1;
#pragma unroll
        for (int f = 0; f < NFR; ++f) xn[f][ti] = reinterpret_cast<const T*>(fp.p[f])[row * D + dcl];
      }
    }
  };
  auto load_ys = [&](int step) __attribute__((always_inline)) {   // the blocked array is padded by one chunk
#pragma unroll
    for (int j = 0; j < YJ; ++j) ys[j] = (yblk + (int64_t)step * YE)[lane + 64 * j];
  };

  // one step.  FAST: the next step exists for this wave and its rows are all inside the grid
  // (no per-lane clamping, no liveness tests).
  auto do_step = [&](int step, auto fastc) __attribute__((always_inline)) {
    constexpr bool FAST = decltype(fastc)::value;
#pragma unroll
    for (int j = 0; j < YJ; ++j)
      if (lane + 64 * j < YE) yst[lane + 64 * j] = ys[j];
    const bool more = FAST || step + 1 < c1;
    if (more) load_ys(step + 1);

    double xs[NFR][GR];
#pragma unroll
    for (int f = 0; f < NFR; ++f)
#pragma unroll
      for (int ti = 0; ti < GR; ++ti) xs[f][ti] = (double)xn[f][ti];
    if (KIND == 0) {
#pragma unroll
      for (int ti = 0; ti < GR; ++ti) xs[2][ti] *= sth;
    }
    if (more) load_x(step + 1, fastc);

    // The slab is loop invariant: without this, hipcc hoists all 4*TB LDS reads out of the loop
    // into 8*TB registers and spills.  Laundering the index keeps them as in-loop ds_reads.
    asm volatile("" : "+v"(cbi));
    const double* cbr = lds + cbi;

    // ---- reconstruction: rec[f][ti] = sum_s Y0blk[ti][s] . C_f[s] ----
    double rec[NFR][GR];
#pragma unroll
    for (int f = 0; f < NFR; ++f)
#pragma unroll
      for (int ti = 0; ti < GR; ++ti) rec[f][ti] = 0.0;
#pragma unroll
    for (int s = 0; s < TB; ++s) {
      double cbc[NFR];
#pragma unroll
      for (int f = 0; f < NFR; ++f) cbc[f] = cbr[(f * TB + s) * 64];
#pragma unroll
      for (int ti = 0; ti < GR; ++ti) {
        const double ya = yst[(ti * TB + s) * 16 + aoff_r];
#pragma unroll
        for (int f = 0; f < NFR; ++f) rec[f][ti] = TEMX_MFMA4(ya, cbc[f], rec[f][ti]);
      }
    }

    // ---- eddies and products (tem_diagnostics.py:517-529, 547-555; tracers :537, :563-567) ----
    double p[NPR][GR];
#pragma unroll
    for (int ti = 0; ti < GR; ++ti) {
      double e[NFR];
#pragma unroll
      for (int f = 0; f < NFR; ++f) e[f] = xs[f][ti] - rec[f][ti];
      if (KIND == 0) {
        p[0][ti] = e[0] * e[1];                    // u'v'
        p[1][ti] = e[0] * e[NFR - 1];              // u'w'
        p[NPR - 1][ti] = e[1] * e[2];              // v'theta'
      } else {
        p[0][ti] = e[0] * e[1];                    // q'v'
        p[1][ti] = e[0] * e[2];                    // q'w'
      }
      if (MODE == 1) {
        const int64_t row = ((int64_t)step * GR + ti) * 4 + g;
        if (dvalid && row < N) {
          const int64_t o = row * D + d;
          if (KIND == 0) {
#pragma unroll
            for (int f = 0; f < NFR; ++f)
              if (eo.p[f]) eo.p[f][o] = e[f];
          } else if (eo.p[0]) {
            eo.p[0][o] = e[0];
          }
#pragma unroll
          for (int q = 0; q < NPR; ++q)
            if (eo.p[4 + q]) eo.p[4 + q][o] = p[q][ti];
        }
      }
    }

    // ---- projection of the three products ----
#pragma unroll
    for (int ti = 0; ti < GR; ++ti)
#pragma unroll
      for (int t = 0; t < TB; ++t) {
        const double ya = yst[(ti * TB + t) * 16 + aoff_p];
#pragma unroll
        for (int q = 0; q < NPR; ++q) acc[q][t] = TEMX_MFMA4(ya, p[q][ti], acc[q][t]);
      }
  };

  if (!active) return;                      // ragged last quad: nothing to do, nobody waits
  if (c0 < c1) {
    load_ys(c0);
    load_x(c0, std::false_type{});
  }
  const int cfast = (c1 < nfull ? c1 : nfull) - 1;
  int step = c0;
  for (; step < cfast; ++step) do_step(step, std::true_type{});
  for (; step < c1; ++step) do_step(step, std::false_type{});

  if (dvalid && partial != nullptr) {
    const int64_t slab = sub;
#pragma unroll
    for (int q = 0; q < NPR; ++q)
#pragma unroll
      for (int t = 0; t < TB; ++t) {
        const int l = t * 4 + g;
        if (l < K) partial[((slab * NPR + q) * K + l) * D + d] = acc[q][t];
      }
  }
}

// native-grid zonal mean out[i][d] = sum_l Y0[i][l] C[l][d]  (sph_zonal_mean_native,
// sph_zonal_mean.py:285-290, outer matmul with Y = Y0).  Same tile scheme as the eddy sweep.
template <int TB>
__global__ void __launch_bounds__(256, 2)
recon_kernel(int64_t N, int64_t D, const double* __restrict__ yblk, int gstride, int tb_off, int64_t nchunk,
             const double* __restrict__ C /* rows of this slice */, double* __restrict__ out, int accumulate,
             int nsplit, int ndt) {
  extern __shared__ double lds[];
  int split, dq;
  if (!wg_work((ndt + 3) >> 2, nsplit, split, dq)) return;
  const int wave = uniform_wave();
  const int dt = dq * 4 + wave;
  if (dt >= ndt) return;
  const int lane = threadIdx.x & 63;
  const int c = lane & 15, g = lane >> 4;
  const int64_t d = (int64_t)dt * 16 + c;
  const bool dvalid = d < D;
  const int64_t dcl = dvalid ? d : D - 1;
  const int64_t c0 = nchunk * 4 * split / nsplit, c1 = nchunk * 4 * (split + 1) / nsplit;  // groups
  double* cb = lds + (size_t)wave * (TB * 64) + lane;
#pragma unroll
  for (int s = 0; s < TB; ++s) cb[s * 64] = C[((int64_t)4 * s + g) * D + dcl];
  const double* yb = yblk + (int64_t)tb_off * 16 + (lane & 3) * 4 + g;
  for (int64_t group = c0; group < c1; ++group) {
    double rec = 0.0;
#pragma unroll
    for (int s = 0; s < TB; ++s) rec = TEMX_MFMA4(yb[(group * gstride + s) * 16], cb[s * 64], rec);
    const int64_t row = group * 4 + g;
    if (dvalid && row < N) {
      if (accumulate) rec += out[row * D + d];
      out[row * D + d] = rec;
    }
  }
}

// eddies and eddy products from stored native-grid means (large-L path, where the harmonics do not
// fit one fused sweep):  e_f = x_f - xbar_f (theta scaled), p = u'v', u'w', v'theta'.
// eo.p[0..3] eddies, eo.p[4..6] products; NULL entries are skipped.
template <typename T>
__global__ void __launch_bounds__(256)
eddy_from_xbar_kernel(FieldPtrs<4> fp, FieldPtrs<4> xbar /* native zonal means, [N][D] doubles each */,
                      int64_t N, int64_t D, const double* __restrict__ colscale, EddyOut eo) {
  const int64_t total = N * D;
  for (int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    double e[4];
#pragma unroll
    for (int f = 0; f < 4; ++f) {
      double x = (double)reinterpret_cast<const T*>(fp.p[f])[idx];
      if (f == 2 && colscale != nullptr) x *= colscale[idx % D];
      e[f] = x - reinterpret_cast<const double*>(xbar.p[f])[idx];
    }
#pragma unroll
    for (int f = 0; f < 4; ++f)
      if (eo.p[f]) eo.p[f][idx] = e[f];
    if (eo.p[4]) eo.p[4][idx] = e[0] * e[1];
    if (eo.p[5]) eo.p[5][idx] = e[0] * e[3];
    if (eo.p[6]) eo.p[6][idx] = e[1] * e[2];
  }
}

// ------------------------------------------------------------------------------------------------
// zonal-grid epilogue (everything below works on [M][nlev][nt], < 0.5 % of the native bytes)
// ------------------------------------------------------------------------------------------------
struct EpiTables {
  const double* p;       // [nlev]   pressure, Pa            (tem_diagnostics.py:385)
  const double* pg;      // [nlev][3] np.gradient coefficients along p   (tem_util.py:192)
  const double* lg;      // [M][3]    np.gradient coefficients along lat [rad] (tem_util.py:154)
  const double* coslat;  // [M]      (tem_diagnostics.py:402)
  const double* fcor;    // [M]      (tem_diagnostics.py:401)
};

// int_vbdp: cumulative trapezoid from the model top (tem_util.py:230-232, np.trapz:
// sum(diff(p) * (y[1:] + y[:-1]) / 2)).  One wave per (lat, time) column, lanes along lev,
// inclusive scan by wavefront shuffles, carry across 64-level chunks.
__global__ void __launch_bounds__(256)
pint_scan_kernel(const double* __restrict__ vb, const double* __restrict__ p, int M, int nlev,
                 int64_t nt, double* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int64_t col = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (col >= (int64_t)M * nt) return;
  const int64_t m = col / nt, t = col % nt;
  const double* v = vb + m * nlev * nt + t;
  double* o = out + m * nlev * nt + t;
  double carry = 0.0;
  for (int j0 = 0; j0 < nlev; j0 += 64) {
    const int j = j0 + lane;
    double term = 0.0;
    if (j < nlev && j > 0) term = (p[j] - p[j - 1]) * (v[(int64_t)j * nt] + v[(int64_t)(j - 1) * nt]) / 2.0;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const double up = __shfl_up(term, off, 64);
      if (lane >= off) term += up;
    }
    if (j < nlev) o[(int64_t)j * nt] = carry + term;
    carry += __shfl(term, 63, 64);
  }
}

// One thread per zonal grid point; every stencil is recomputed from the seven zonal means
// (all L1/L2 resident), so the ten GM16 Table-A1 outputs come out of a single launch.
// zb: [8][M][D] = ub vb thetab wapb upvpb upwappb vptpb int_vbdp.
// INT_MODE 1: int_vbdp (the cumulative trapezoid of vb from the model top, tem_util.py:230-232) is summed
// by the thread itself -- O(nlev) L1/L2 reads per point instead of a launch of pint_scan_kernel -- and
// stored to zb[7] for later readers.  Pays for short columns only (the host decides): measured on
// ne30 x 72 x 1 the loop costs the 5 us the launch saves, on nlev = 128 more.
// INT_MODE 2 (small zonal grids, nlev * nt <= EPI_WG_MAXD): one workgroup per latitude; its waves first run the scan
// of pint_scan_kernel (same association, same bits) for the latitude's columns into LDS, then the threads walk the
// latitude's points -- the scan without a launch of its own and without the O(nlev) loop per point of mode 1.
constexpr int EPI_WG_MAXD = 1024;

template <int INT_MODE>
__device__ __forceinline__ void tem_epilogue_point(double* __restrict__ zb, int M, int nlev, int64_t nt, const EpiTables& tb,
                                                   double p0, double* __restrict__ res, double* __restrict__ zon,
                                                   int64_t idx, double intv_in) {
  // constants.py:6-14 (NB: pi is the reference's truncated value, used by psitem only)
  constexpr double a_e = 6.37123e6, g0 = 9.80665, Hs = 7000.0, pi_ref = 3.14159;
  const int64_t D = (int64_t)nlev * nt;
  const int64_t MD = (int64_t)M * D;
  const int m = (int)(idx / D);
  const int64_t dd = idx % D;
  const int j = (int)(dd / nt);
  const int64_t t = dd % nt;
  const double* ub = zb;
  const double* vb = zb + MD;
  const double* thb = zb + 2 * MD;
  const double* wb = zb + 3 * MD;
  const double* upvpb = zb + 4 * MD;
  const double* upwb = zb + 5 * MD;
  const double* vptpb = zb + 6 * MD;

  auto at = [&](const double* A, int mm, int jj) { return A[((int64_t)mm * nlev + jj) * nt + t]; };
  auto clj = [&](int jj) { return jj < 0 ? 0 : (jj >= nlev ? nlev - 1 : jj); };
  auto clm = [&](int mm) { return mm < 0 ? 0 : (mm >= M ? M - 1 : mm); };
  double intv0;                              // int_vbdp at (m, j)
  if constexpr (INT_MODE == 1) {
    double a0 = 0.0, a1 = 0.0;               // two chains: the adds are the critical path
    double vprev = at(vb, m, 0);
    int jj = 1;
    for (; jj + 1 <= j; jj += 2) {
      const double v1 = at(vb, m, jj), v2 = at(vb, m, jj + 1);
      a0 += (tb.p[jj] - tb.p[jj - 1]) * (v1 + vprev) / 2.0;
      a1 += (tb.p[jj + 1] - tb.p[jj]) * (v2 + v1) / 2.0;
      vprev = v2;
    }
    if (jj <= j) a0 += (tb.p[jj] - tb.p[jj - 1]) * (at(vb, m, jj) + vprev) / 2.0;
    intv0 = a0 + a1;
    zb[7 * MD + idx] = intv0;
  } else if constexpr (INT_MODE == 2) {
    intv0 = intv_in;
    zb[7 * MD + idx] = intv0;
  } else {
    intv0 = zb[7 * MD + idx];
  }
  // d/dp with numpy's second-order non-uniform interior, first-order edges (tem_util.py:192)
  auto ddp = [&](const double* A, int mm, int jj) {
    return tb.pg[jj * 3 + 0] * at(A, mm, clj(jj - 1)) + tb.pg[jj * 3 + 1] * at(A, mm, jj) +
           tb.pg[jj * 3 + 2] * at(A, mm, clj(jj + 1));
  };
  auto psi_at = [&](int mm, int jj) { return at(vptpb, mm, jj) / ddp(thb, mm, jj); };  // :590
  // d(ub cos)/dlat (tem_diagnostics.py:584-586)
  auto dubcos_dlat = [&](int mm, int jj) {
    const int ma = clm(mm - 1), mb = clm(mm + 1);
    return tb.lg[mm * 3 + 0] * (at(ub, ma, jj) * tb.coslat[ma]) +
           tb.lg[mm * 3 + 1] * (at(ub, mm, jj) * tb.coslat[mm]) +
           tb.lg[mm * 3 + 2] * (at(ub, mb, jj) * tb.coslat[mb]);
  };
  // EP flux components in log-pressure form (tem_diagnostics.py:691-692, 709-710)
  auto epfy_at = [&](int mm, int jj, double ps) {
    const double x = (ddp(ub, mm, jj) * ps - at(upvpb, mm, jj)) * (a_e * tb.coslat[mm]);
    return x * (tb.p[jj] / p0);
  };
  auto epfz_at = [&](int mm, int jj, double ps) {
    const double x = tb.fcor[mm] - dubcos_dlat(mm, jj) * (1.0 / (a_e * tb.coslat[mm]));
    return -Hs / p0 * ((x * ps - at(upwb, mm, jj)) * (a_e * tb.coslat[mm]));
  };

  const int jm = clj(j - 1), jp = clj(j + 1), mm1 = clm(m - 1), mp1 = clm(m + 1);
  const double cosm = tb.coslat[m];
  const double inv_acos = 1.0 / (a_e * cosm);
  const double psi0 = psi_at(m, j);
  const double psi_jm = psi_at(m, jm), psi_jp = psi_at(m, jp);
  const double psi_mm = psi_at(mm1, j), psi_mp = psi_at(mp1, j);
  const double dub_dp = ddp(ub, m, j);
  const double dth_dp = ddp(thb, m, j);
  const double dpsi_dp = tb.pg[j * 3 + 0] * psi_jm + tb.pg[j * 3 + 1] * psi0 + tb.pg[j * 3 + 2] * psi_jp;
  const double psicos = psi0 * cosm;
  const double dpsicos_dlat = tb.lg[m * 3 + 0] * (psi_mm * tb.coslat[mm1]) + tb.lg[m * 3 + 1] * psicos +
                              tb.lg[m * 3 + 2] * (psi_mp * tb.coslat[mp1]);
  const double dubcos = dubcos_dlat(m, j);

  const double vtem = at(vb, m, j) - dpsi_dp;                                    // :622
  const double omegatem = at(wb, m, j) + dpsicos_dlat * inv_acos;                // :639
  const double wtem = omegatem * (-Hs / tb.p[j]);                                // :657
  const double psitem = 2 * pi_ref * a_e / g0 * ((intv0 - psi0) * cosm);           // :674
  const double epfy = epfy_at(m, j, psi0);
  const double epfz = epfz_at(m, j, psi0);
  // EP flux divergence (tem_diagnostics.py:730-736)
  const double p0_p = p0 / tb.p[j];
  const double Fphi_cos_m = epfy_at(mm1, j, psi_mm) * p0_p * tb.coslat[mm1];
  const double Fphi_cos_0 = epfy * p0_p * cosm;
  const double Fphi_cos_p = epfy_at(mp1, j, psi_mp) * p0_p * tb.coslat[mp1];
  const double dFphi = tb.lg[m * 3 + 0] * Fphi_cos_m + tb.lg[m * 3 + 1] * Fphi_cos_0 + tb.lg[m * 3 + 2] * Fphi_cos_p;
  const double Fp_m = epfz_at(m, jm, psi_jm) * -p0 / Hs;
  const double Fp_0 = epfz * -p0 / Hs;
  const double Fp_p = epfz_at(m, jp, psi_jp) * -p0 / Hs;
  const double dFp = tb.pg[j * 3 + 0] * Fp_m + tb.pg[j * 3 + 1] * Fp_0 + tb.pg[j * 3 + 2] * Fp_p;
  const double epdiv = dFphi * inv_acos + dFp;
  const double utendepfd = epdiv * inv_acos;                                     // :753
  const double utendvtem = vtem * (tb.fcor[m] - dubcos * inv_acos);              // :772-773
  const double utendwtem = -omegatem * dub_dp;                                   // :791

  res[0 * MD + idx] = vtem;
  res[1 * MD + idx] = omegatem;
  res[2 * MD + idx] = wtem;
  res[3 * MD + idx] = psitem;
  res[4 * MD + idx] = epfy;
  res[5 * MD + idx] = epfz;
  res[6 * MD + idx] = epdiv;
  res[7 * MD + idx] = utendepfd;
  res[8 * MD + idx] = utendvtem;
  res[9 * MD + idx] = utendwtem;
  if (zon != nullptr) {
#pragma unroll
    for (int q = 0; q < 7; ++q) zon[q * MD + idx] = zb[q * MD + idx];
    zon[7 * MD + idx] = dub_dp;
    zon[8 * MD + idx] = dth_dp;
    zon[9 * MD + idx] = at(ub, m, j) * cosm;
    zon[10 * MD + idx] = dubcos;
    zon[11 * MD + idx] = psi0;
    zon[12 * MD + idx] = psicos;
    zon[13 * MD + idx] = dpsicos_dlat;
    zon[14 * MD + idx] = dpsi_dp;
    zon[15 * MD + idx] = intv0;
  }
}

template <int INT_MODE>
__global__ void __launch_bounds__(256)
tem_epilogue_kernel(double* __restrict__ zb, int M, int nlev, int64_t nt, EpiTables tb,
                    double p0, double* __restrict__ res, double* __restrict__ zon) {
  const int64_t D = (int64_t)nlev * nt;
  const int64_t MD = (int64_t)M * D;
  if constexpr (INT_MODE == 2) {
    __shared__ double s_int[EPI_WG_MAXD];
    const int m = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const double* v = zb + MD + (int64_t)m * D;       // vb of this latitude, [nlev][nt]
    for (int64_t t = wave; t < nt; t += 4) {
      double carry = 0.0;
      for (int j0 = 0; j0 < nlev; j0 += 64) {
        const int j = j0 + lane;
        double term = 0.0;
        if (j < nlev && j > 0) term = (tb.p[j] - tb.p[j - 1]) * (v[(int64_t)j * nt + t] + v[(int64_t)(j - 1) * nt + t]) / 2.0;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
          const double up = __shfl_up(term, off, 64);
          if (lane >= off) term += up;
        }
        if (j < nlev) s_int[(int64_t)j * nt + t] = carry + term;
        carry += __shfl(term, 63, 64);
      }
    }
    __syncthreads();
    for (int64_t dd = threadIdx.x; dd < D; dd += 256)
      tem_epilogue_point<2>(zb, M, nlev, nt, tb, p0, res, zon, (int64_t)m * D + dd, s_int[dd]);
  } else {
    const int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (idx >= MD) return;
    tem_epilogue_point<INT_MODE>(zb, M, nlev, nt, tb, p0, res, zon, idx, 0.0);
  }
}

// Tracer TEM epilogue (Abalos+ 2017; tem_diagnostics.py:602-611, 801-991).  One thread per zonal
// grid point, stencils recomputed like tem_epilogue_kernel.
// zb: the TEM zonal means of the plan ([8][M][D]); tz: [3][M][D] = qb qpvpb qpwappb.
// tres: [6][M][D] = etfy etfz etdiv qtendetfd qtendvtem qtendwtem;
// tzon: NULL or [6][M][D] = qb qpvpb qpwappb dqb_dp qbcoslat dqbcoslat_dlat.
__global__ void __launch_bounds__(256)
tracer_epilogue_kernel(const double* __restrict__ zb, const double* __restrict__ tz, int M, int nlev,
                       int64_t nt, EpiTables tb, double p0, double* __restrict__ tres,
                       double* __restrict__ tzon) {
  constexpr double a_e = 6.37123e6, Hs = 7000.0;
  const int64_t D = (int64_t)nlev * nt;
  const int64_t MD = (int64_t)M * D;
  const int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (idx >= MD) return;
  const int m = (int)(idx / D);
  const int64_t dd = idx % D;
  const int j = (int)(dd / nt);
  const int64_t t = dd % nt;
  const double* vb = zb + MD;
  const double* thb = zb + 2 * MD;
  const double* wb = zb + 3 * MD;
  const double* vptpb = zb + 6 * MD;
  const double* qb = tz;
  const double* qpvpb = tz + MD;
  const double* qpwb = tz + 2 * MD;

  auto at = [&](const double* A, int mm, int jj) { return A[((int64_t)mm * nlev + jj) * nt + t]; };
  auto clj = [&](int jj) { return jj < 0 ? 0 : (jj >= nlev ? nlev - 1 : jj); };
  auto clm = [&](int mm) { return mm < 0 ? 0 : (mm >= M ? M - 1 : mm); };
  auto ddp = [&](const double* A, int mm, int jj) {
    return tb.pg[jj * 3 + 0] * at(A, mm, clj(jj - 1)) + tb.pg[jj * 3 + 1] * at(A, mm, jj) +
           tb.pg[jj * 3 + 2] * at(A, mm, clj(jj + 1));
  };
  auto psi_at = [&](int mm, int jj) { return at(vptpb, mm, jj) / ddp(thb, mm, jj); };
  auto dqbcos_dlat = [&](int mm, int jj) {                              // :608-610
    const int ma = clm(mm - 1), mb = clm(mm + 1);
    return tb.lg[mm * 3 + 0] * (at(qb, ma, jj) * tb.coslat[ma]) +
           tb.lg[mm * 3 + 1] * (at(qb, mm, jj) * tb.coslat[mm]) +
           tb.lg[mm * 3 + 2] * (at(qb, mb, jj) * tb.coslat[mb]);
  };
  auto etfy_at = [&](int mm, int jj, double ps) {                       // :823-824
    const double x = (ddp(qb, mm, jj) * ps - at(qpvpb, mm, jj)) * (a_e * tb.coslat[mm]);
    return x * (tb.p[jj] / p0);
  };
  auto etfz_at = [&](int mm, int jj, double ps) {                       // :856-857
    const double x = -(dqbcos_dlat(mm, jj) * (1.0 / (a_e * tb.coslat[mm])));
    return -Hs / p0 * ((x * ps - at(qpwb, mm, jj)) * (a_e * tb.coslat[mm]));
  };

  const int jm = clj(j - 1), jp = clj(j + 1), mm1 = clm(m - 1), mp1 = clm(m + 1);
  const double cosm = tb.coslat[m];
  const double inv_acos = 1.0 / (a_e * cosm);
  const double psi0 = psi_at(m, j);
  const double psi_jm = psi_at(m, jm), psi_jp = psi_at(m, jp);
  const double psi_mm = psi_at(mm1, j), psi_mp = psi_at(mp1, j);
  const double dpsi_dp = tb.pg[j * 3 + 0] * psi_jm + tb.pg[j * 3 + 1] * psi0 + tb.pg[j * 3 + 2] * psi_jp;
  const double dpsicos_dlat = tb.lg[m * 3 + 0] * (psi_mm * tb.coslat[mm1]) + tb.lg[m * 3 + 1] * (psi0 * cosm) +
                              tb.lg[m * 3 + 2] * (psi_mp * tb.coslat[mp1]);
  const double vtem = at(vb, m, j) - dpsi_dp;                            // :622
  const double omegatem = at(wb, m, j) + dpsicos_dlat * inv_acos;        // :639
  const double dqb_dp = ddp(qb, m, j);
  const double dqbcos = dqbcos_dlat(m, j);

  const double etfy = etfy_at(m, j, psi0);
  const double etfz = etfz_at(m, j, psi0);
  const double p0_p = p0 / tb.p[j];                                      // :887-893
  const double dM = tb.lg[m * 3 + 0] * (etfy_at(mm1, j, psi_mm) * p0_p * tb.coslat[mm1]) +
                    tb.lg[m * 3 + 1] * (etfy * p0_p * cosm) +
                    tb.lg[m * 3 + 2] * (etfy_at(mp1, j, psi_mp) * p0_p * tb.coslat[mp1]);
  const double dMp = tb.pg[j * 3 + 0] * (etfz_at(m, jm, psi_jm) * -p0 / Hs) + tb.pg[j * 3 + 1] * (etfz * -p0 / Hs) +
                     tb.pg[j * 3 + 2] * (etfz_at(m, jp, psi_jp) * -p0 / Hs);
  const double etdiv = dM * inv_acos + dMp;
  tres[0 * MD + idx] = etfy;
  tres[1 * MD + idx] = etfz;
  tres[2 * MD + idx] = etdiv;
  tres[3 * MD + idx] = etdiv * inv_acos;                                 // qtendetfd :921
  tres[4 * MD + idx] = -vtem * (dqbcos * inv_acos);                      // qtendvtem :952-953
  tres[5 * MD + idx] = -omegatem * dqb_dp;                               // qtendwtem :984-985
  if (tzon != nullptr) {
    tzon[0 * MD + idx] = at(qb, m, j);
    tzon[1 * MD + idx] = at(qpvpb, m, j);
    tzon[2 * MD + idx] = at(qpwb, m, j);
    tzon[3 * MD + idx] = dqb_dp;
    tzon[4 * MD + idx] = at(qb, m, j) * cosm;
    tzon[5 * MD + idx] = dqbcos;
  }
}

// Y0inv[k][i] = sum_k' Ginv[k][k'] Y0[i][k']  (attribute .Y0inv only; never on the hot path)
__global__ void y0inv_kernel(const double* __restrict__ Y0, const double* __restrict__ Ginv,
                             int64_t N, int K, double* __restrict__ out) {
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= N) return;
  for (int k = 0; k < K; ++k) {
    double v = 0.0;
    for (int kk = 0; kk < K; ++kk) v += Ginv[k * K + kk] * Y0[i * K + kk];
    out[(int64_t)k * N + i] = v;
  }
}

// ------------------------------------------------------------------------------------------------
// measurement helpers
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint64_t splitmix64(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}

__device__ __forceinline__ double hash_normal(uint64_t seed, uint64_t field, uint64_t idx) {
  const uint64_t h1 = splitmix64(seed * 0x100000001B3ull + field * 0x9E3779B97F4A7C15ull + idx * 2);
  const uint64_t h2 = splitmix64(h1 ^ 0xD1B54A32D192ED03ull);
  const double u1 = ((double)(h1 >> 11) + 0.5) * (1.0 / 9007199254740992.0);
  const double u2 = ((double)(h2 >> 11) + 0.5) * (1.0 / 9007199254740992.0);
  return sqrt(-2.0 * log(u1)) * cos(6.283185307179586 * u2);
}

// SURVEY section 8(d) synthetic fields (same analytic part as pytemdiags_amd/synth.py).
template <typename T>
__global__ void synth_kernel(int64_t N, int nlev, int64_t nt, int64_t t0, const double* __restrict__ lat,
                             const double* __restrict__ lon, const double* __restrict__ plev,
                             uint64_t seed, T* __restrict__ ua, T* __restrict__ va,
                             T* __restrict__ ta, T* __restrict__ wap) {
  const int64_t D = (int64_t)nlev * nt;
  const int64_t total = N * D;
  for (int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    const int64_t i = idx / D, dd = idx % D;
    const int j = (int)(dd / nt);
    const double t = (double)(t0 + dd % nt);
    const double dr = 0.017453292519943295;
    const double phi = lat[i] * dr, lam = lon[i] * dr;
    const double z = -7.0 * log(plev[j] / 1000.0);
    const double s = sin(phi), c = cos(phi);
    const double s2p = sin(2 * phi);
    double T_ = 300.0 - 60.0 * s * s - 6.5 * fmin(z, 12.0) + 2.0 * fmax(z - 20.0, 0.0) +
                3.0 * cos(3 * lam + 0.3 * t) * c;
    double u = 30.0 * s2p * s2p * exp(-((z - 12.0) / 8.0) * ((z - 12.0) / 8.0)) +
               8.0 * sin(4 * lam + 0.2 * t) * c * c;
    double v = 6.0 * cos(4 * lam + 0.2 * t) * c * c * exp(-((z - 10.0) / 10.0) * ((z - 10.0) / 10.0)) +
               0.5 * s2p;
    double w = 0.05 * sin(4 * lam + 0.2 * t + 0.7) * c * c + 0.01 * cos(3 * phi);
    // the noise index uses the absolute time so time shards of one job are consistent
    const uint64_t nidx = (uint64_t)((i * nlev + j) * (int64_t)1000003 + (t0 + dd % nt));
    u += 0.1 * hash_normal(seed, 0, nidx);
    v += 0.1 * hash_normal(seed, 1, nidx);
    T_ += 0.1 * hash_normal(seed, 2, nidx);
    w += 0.1 * hash_normal(seed, 3, nidx);
    ua[idx] = (T)u;
    va[idx] = (T)v;
    ta[idx] = (T)T_;
    wap[idx] = (T)w;
  }
}

// bare fp64 MFMA issue loop (v_mfma_f64_4x4x4_4b_f64): 8 independent accumulators per wave.
__global__ void __launch_bounds__(256) mfma_f64_peak_kernel(int iters, double* sink) {
  double a[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) a[i] = 0.0;
  double x = 1.0 + threadIdx.x * 1e-3, y = 0.5 - threadIdx.x * 1e-4;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int j = 0; j < 8; ++j) a[j] = TEMX_MFMA4(x, y, a[j]);
  }
  double r = 0.0;
#pragma unroll
  for (int i = 0; i < 8; ++i) r += a[i];
  if (r == 12345.678) sink[0] = r;
}

}  // namespace temx
