// kernels_cls.hpp -- latitude-class sweeps: native columns that share a latitude share a basis row.
//
// Y_l^0 depends on latitude only.  On the grids this code meets (cubed-sphere: 8 columns per
// latitude and hemisphere by the symmetry of the cube; lat-lon / Gaussian: NLON columns per
// latitude) many native columns have the same |lat|.  For a class c of such columns, with northern
// members N(c) and southern members S(c) (Y_l^0(-lat) = (-1)^l Y_l^0(lat)):
//   projection      sum_i Y0[i][l] x_i  =  sum_c Y_l(lat_c) ( sum_{N(c)} x  +-  sum_{S(c)} x )
//   reconstruction  xbar is the same for every member of a side:  xbar_N = E + O,  xbar_S = E - O
// so the MFMA work is per CLASS, not per column: 1/8 of the generic sweeps on a cubed sphere, and
// the sweeps turn from MFMA bound into pure HBM streams (one v_add / a few VALU ops per element).
// The mirror-paired sweeps of kernels_sym.hpp are the special case of one member per side.
//
// Tables (built on the host, temx.hip build_classes):
//   classes are sorted by (members N, members S) so that the 4 classes of a class-group (the MFMA
//   k dimension) have equal counts; a group is walked in batches of CLS_MB member rows per class,
//   first its northern batches, then its southern ones.
//   crow[batch][g][j]  int32: member row j of class g of the batch's group (one int4 per 16-lane
//                      group), | batch-has-padding << 27 | side << 28 | first-batch-of-group << 29 |
//                      last << 30; bit 31 = padding entry (no member: row 0 is read and weighted 0).
//                      Every entry of a batch carries the batch flags.  Padded by CLS_PADB batches.
//   ycls[group][2*TBS][16]  4x4 blocks at the class latitudes, layout of kernels_sym.hpp's ysym.
//   csplit[nsub+1]     (first batch, its group) of every piece of work; equal batch counts, cuts may
//                      fall inside a group (both sweeps are linear in the member rows).
// No workgroup barriers in the sweeps' main loops: each wave stages its own Y blocks (wave-private
// LDS) and walks its own flat batch list with the X loads PD - 1 batches ahead and the row indices one
// batch further.
//
// One-pass form (project_cls_kernel<.., OP = true> + flux_cls_kernel): inside a class side xbar is a
// constant, so   sum (u - ub)(v - vb) = C_uv + n (m_u - ub)(m_v - vb),   m = S / n the side's mean and
// C_uv = sum (u - m_u)(v - m_v) its centred co-moment.  Sweep 1 also accumulates C_uv, C_uw, C_vT per
// class and side (about the side's first member, so nothing large cancels).  They enter the projected
// eddy-product sums only linearly, so sweep 1 projects them itself (seven projections per wave) and
// stores, per (class-group, d-tile), only the four field sums
//   csum[group][dt][4][64][2] sums S_u S_v S_theta S_w, per lane (= 16 class + column) the
//                             {northern, southern} pair: 4 x 16-byte stores / loads per lane
// from which flux_cls_kernel forms and projects  n (m_u - ub)(m_v - vb)  after the solve: the fields
// are read once.  Its work cuts are aligned to class-groups (a stored sum must be
// complete).  The large-L class path (PROJ = false) stores all seven sums (records of 7 pairs).
#pragma once
#include "kernels_sym.hpp"

namespace temx {

// X is read once per sweep.  Non-temporal loads (-DTEMX_NT_LOADS) were measured: round 1 +2 % on
// ne120x72x30 for the two-pass sweeps but -20 % on the ne30 shapes; round 2, one-pass sweep: +-0 on
// ne120x72x30, -26 % on ne30x72x91 and on ne240x128x1 fp32.  Plain loads stay the default.  (A 4-deep X
// ring in the one-pass sweep, -DTEMX_CLS_OP_PD=4, measured +-0 as well.)
#ifdef TEMX_NT_LOADS
#define TEMX_XLOAD(p) __builtin_nontemporal_load(p)
#else
#define TEMX_XLOAD(p) (*(p))
#endif

// class-sum record stores.  The records are written once and read by a later kernel; measured on ne120 x
// 72 x 30 (tools/sweep_lab.hip, profiles/r03_lab_*.log): non-temporal stores 10.9-11.1 ms whatever the
// placement of the buffer, plain stores 10.9-11.1 / 11.5-11.9 ms (fast / slow placement, see
// alloc_write_stream in temx.hip), sc0 sc1 stores 10.9-11.0 / 11.2-11.6 ms.  The lab builds swap flavours.
typedef double temx_d2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void temx_nt_store2(double2* p, double2 v) {
  temx_d2v t = {v.x, v.y};
  __builtin_nontemporal_store(t, reinterpret_cast<temx_d2v*>(p));
}
__device__ __forceinline__ void temx_sc1_store2(double2* p, double2 v) {
  temx_d2v t = {v.x, v.y};
  asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(t) : "memory");
}
#if defined(TEMX_CSTORE_NONE)
#define TEMX_CSTORE(p, v) ((void)(p), (void)(v))
#elif defined(TEMX_CSTORE_SC1)       // write-through, the line is dropped from L2
#define TEMX_CSTORE(p, v) temx_sc1_store2((p), (v))
#elif defined(TEMX_CSTORE_PLAIN)
#define TEMX_CSTORE(p, v) (*(p) = (v))
#else
#define TEMX_CSTORE(p, v) temx_nt_store2((p), (v))
#endif
// index of the class-sum record of (class-group, d-tile); lab builds try other layouts
#ifdef TEMX_LAB
__device__ int64_t temx_lab_ngr;
#endif
#ifndef TEMX_CSUM_REC
#define TEMX_CSUM_REC(grp, dt, ndt) ((int64_t)(grp) * (ndt) + (dt))
#endif

constexpr int CLS_MB = 4;                     // member rows per class and batch
constexpr int CLS_PADB = 10;                  // batches of padding behind crow (index loads run up to PD + 1 ahead)

// compile-time unrolled loop: f(integral_constant<int, 0>) ... f(integral_constant<int, N - 1>)
template <int N, int I = 0, typename F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<N, I + 1>(f);
  }
}
constexpr int CLS_ROWMASK = 0x07FFFFFF;
constexpr int CLS_HASPAD_BIT = 1 << 27;       // the batch has at least one padding entry (set in all its entries)
constexpr int CLS_SOUTH = 1, CLS_FIRST = 2, CLS_LAST = 4;   // flags, stored at bit 28

template <int KMAX>
__global__ void cls_basis_kernel(const double* __restrict__ xc, int64_t ncls, int64_t ncls_pad, int K, int TBS,
                                 const double* __restrict__ norm, const double* __restrict__ T,
                                 double* __restrict__ ycls) {
  int64_t ci = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (ci >= ncls_pad) return;
  const bool valid = ci < ncls;
  const double xv = valid ? xc[ci] : 0.0;
  const int64_t grp = ci >> 2;
  const int k = (int)(ci & 3);
  double* blk = ycls + grp * (2 * TBS * 16);
  double q[KMAX];
  basis_row<KMAX>(xv, K, norm, T, q);          // (T keeps the parity of a column: temx_plan_finalize)
  for (int l = 0; l < 8 * TBS; ++l) {
    const double val = (valid && l < K) ? q[l] : 0.0;
    const int h = l >> 1;
    const int t = (l & 1) * TBS + (h >> 2);
    blk[t * 16 + k * 4 + (h & 3)] = val;
  }
}

// ------------------------------------------------------------------------------------------------
// class project sweep (sweep 1)
// ------------------------------------------------------------------------------------------------
// PD = X batches held in registers (PD - 1 in flight while one is consumed); the small one-field-
// per-wave configuration has the registers for a deeper ring, which is what short batch lists
// (small D) need: they are latency, not bandwidth, bound.
// OP with PROJ = false (NF = NFW = 4 only; the large-L class path, 64 < K <= 256): no projection, the
// sweep only accumulates, per class and side, the four field sums and the sums of u v, u omega, v T and
// stores the 7 x 2 sums of every (class-group, d-tile) in csum; sums_project_kernel and
// flux_large_kernel work on them 64 harmonics at a time.  Needs work cuts at group boundaries.
// (The one-pass form for K <= 64, which projects while it sums, is sweep_op_kernel in kernels_op.hpp.)
template <typename T, int NF, int NFW, int TBS, int WPS, int PD, bool OP, bool PROJ = true>
__global__ void __launch_bounds__(256, WPS)
project_cls_kernel(FieldPtrs<NF> fp, int64_t D, int K, const double* __restrict__ ycls,
                   const int4* __restrict__ crow, const int2* __restrict__ csplit,
                   const double* __restrict__ colscale, int sfield, double* __restrict__ partial, int nsplit,
                   int ndt, double* __restrict__ csum) {
  static_assert(!OP || (NF == 4 && NFW == 4), "one-pass sums need all four fields in one wave");
  static_assert(PROJ != OP, "either the projections or the class sums (sweep_op_kernel produces both)");
  constexpr int DPW = 4 * NFW / NF;
  constexpr int NB = 2 * TBS;
  constexpr int YE = NB * 16;
  constexpr int YJ = (YE + 63) / 64;
  constexpr int MB = CLS_MB;
  __shared__ double ystage[4][YE];            // wave private
  int split, dq;
  if (!wg_work((ndt + DPW - 1) / DPW, nsplit, split, dq)) return;
  const int wave = uniform_wave(), lane = threadIdx.x & 63;
  const int c = lane & 15, g = lane >> 4;
  const int dt = dq * DPW + wave % DPW;
  const int f0 = (wave / DPW) * NFW;
  if (dt >= ndt) return;                      // (no barriers below)
  const int64_t d = (int64_t)dt * 16 + c;
  const bool dvalid = d < D;
  const int64_t dcl = dvalid ? d : D - 1;
  const int b0 = __builtin_amdgcn_readfirstlane(csplit[split].x);
  const int b1 = __builtin_amdgcn_readfirstlane(csplit[split + 1].x);
  int grp = __builtin_amdgcn_readfirstlane(csplit[split].y);
  const uint32_t yoff = (uint32_t)(g * 4 + (lane & 3));
  double* yst = ystage[wave];

  double sc[NFW];
  const T* fb[NFW];
#pragma unroll
  for (int f = 0; f < NFW; ++f) {
    sc[f] = (colscale != nullptr && f0 + f == sfield) ? colscale[dcl] : 1.0;
    fb[f] = reinterpret_cast<const T*>(fp.p[f0 + f]) + dcl;
  }
  constexpr int NA = (OP && PROJ) ? NFW + 3 : NFW;   // one-pass form: + projections of u v, u omega, v theta
  constexpr int NFP = (OP && PROJ) ? NF + 3 : NF;    // slabs per split in `partial`
  constexpr int RS = PROJ ? 4 : 7;                   // {north, south} pairs per csum record
  double acc[NA][NB];
#pragma unroll
  for (int f = 0; f < NA; ++f)
#pragma unroll
    for (int t = 0; t < NB; ++t) acc[f][t] = 0.0;
  double sN[NFW], sS[NFW];
#pragma unroll
  for (int f = 0; f < NFW; ++f) sN[f] = sS[f] = 0.0;
  double qN[3] = {0.0, 0.0, 0.0}, qS[3] = {0.0, 0.0, 0.0};   // OP: sums of u v, u omega, v T

  T xb[PD][MB][NFW];
  int er[PD][MB];
  int fl[PD];
  double ys[YJ];
  auto load_ys = [&](int gi) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < YJ; ++j) ys[j] = (ycls + (int64_t)gi * YE)[(lane + 64 * j) < YE ? (lane + 64 * j) : 0];
  };
  auto issue = [&](auto pc, const int4 rv) __attribute__((always_inline)) {
    constexpr int P = decltype(pc)::value;
    er[P][0] = rv.x; er[P][1] = rv.y; er[P][2] = rv.z; er[P][3] = rv.w;
    fl[P] = __builtin_amdgcn_readfirstlane(rv.x) >> 28;
#pragma unroll
    for (int j = 0; j < MB; ++j) {
      const int64_t off = (int64_t)(er[P][j] & CLS_ROWMASK) * D;
#pragma unroll
      for (int f = 0; f < NFW; ++f) xb[P][j][f] = TEMX_XLOAD(fb[f] + off);
    }
  };
  int4 rn;
  auto step = [&](auto pc, int b) __attribute__((always_inline)) {
    constexpr int P = decltype(pc)::value;
    if (b + (PD - 1) < b1) {                  // index load first: it must not queue behind the X loads
      const int4 r1 = rn;
      rn = crow[(int64_t)(b + PD) * 4 + g];
      issue(std::integral_constant<int, (P + PD - 1) % PD>{}, r1);
    }
    const int flags = fl[P] | (b + 1 == b1 ? CLS_LAST : 0);   // a piece may end inside a group: project its partial sums
    double wt[MB];                            // padding entries read row 0 and weigh nothing
#pragma unroll
    for (int j = 0; j < MB; ++j) wt[j] = er[P][j] < 0 ? 0.0 : 1.0;
    if constexpr (OP) {
      double* sx = (flags & CLS_SOUTH) ? sS : sN;
      double* sq = (flags & CLS_SOUTH) ? qS : qN;
#pragma unroll
      for (int j = 0; j < MB; ++j) {
        const double u = wt[j] * (double)xb[P][j][0], v = wt[j] * (double)xb[P][j][1];
        const double tt = (double)xb[P][j][2], om = (double)xb[P][j][3];
        sx[0] += u;
        sx[1] += v;
        sx[2] += wt[j] * tt;
        sx[3] += wt[j] * om;
        sq[0] += u * (double)xb[P][j][1];
        sq[1] += u * om;
        sq[2] += v * tt;
      }
    } else if (flags & CLS_SOUTH) {
#pragma unroll
      for (int j = 0; j < MB; ++j)
#pragma unroll
        for (int f = 0; f < NFW; ++f) sS[f] += wt[j] * (double)xb[P][j][f];
    } else {
#pragma unroll
      for (int j = 0; j < MB; ++j)
#pragma unroll
        for (int f = 0; f < NFW; ++f) sN[f] += wt[j] * (double)xb[P][j][f];
    }
    if (flags & CLS_LAST) {
      if constexpr (PROJ) {
#pragma unroll
        for (int j = 0; j < YJ; ++j)
          if (lane + 64 * j < YE) yst[lane + 64 * j] = ys[j];
      }
      if constexpr (OP) {                     // RS sums per side of this (group, d-tile), theta-scaled
        if (dvalid) {
          // row s of the record = {northern, southern} value of sum s per lane: RS stores of 16 B
          double2* o = reinterpret_cast<double2*>(csum + (((int64_t)grp * ndt + dt) * (2 * RS)) * 64) + lane;
#pragma unroll
          for (int f = 0; f < 4; ++f) o[f * 64] = make_double2(sN[f] * sc[f], sS[f] * sc[f]);
          if constexpr (!PROJ) {              // large-L class path: the product sums are projected later
            o[4 * 64] = make_double2(qN[0], qS[0]);
            o[5 * 64] = make_double2(qN[1], qS[1]);
            o[6 * 64] = make_double2(qN[2] * sc[2], qS[2] * sc[2]);
          }
        }
      }
      ++grp;
      if constexpr (PROJ) {
        load_ys(grp);                         // ycls is padded by one group
        double ss[NA], dd[NA];
#pragma unroll
        for (int f = 0; f < NFW; ++f) {
          ss[f] = (sN[f] + sS[f]) * sc[f];
          dd[f] = (sN[f] - sS[f]) * sc[f];
        }
        if constexpr (OP) {                   // the class co-moments are projected like field sums
#pragma unroll
          for (int q = 0; q < 3; ++q) {
            const double qs = q == 2 ? sc[2] : 1.0;
            ss[NFW + q] = (qN[q] + qS[q]) * qs;
            dd[NFW + q] = (qN[q] - qS[q]) * qs;
          }
        }
#pragma unroll
        for (int t = 0; t < NB; ++t) {
          const double ya = yst[t * 16 + yoff];
#pragma unroll
          for (int f = 0; f < NA; ++f) acc[f][t] = TEMX_MFMA4(ya, t < TBS ? ss[f] : dd[f], acc[f][t]);
        }
      }
      if constexpr (OP) {
#pragma unroll
        for (int q = 0; q < 3; ++q) qN[q] = qS[q] = 0.0;
      }
#pragma unroll
      for (int f = 0; f < NFW; ++f) sN[f] = sS[f] = 0.0;
    }
  };

  if (b0 < b1) {
    static_assert(PD + 1 <= CLS_PADB, "crow padding must cover the index prefetch");
    if constexpr (PROJ) load_ys(grp);
    // prologue: X of the first PD - 1 batches (the table is padded, a short list just loads padding)
    rn = crow[(int64_t)b0 * 4 + g];
    static_for<PD - 1>([&](auto kc) __attribute__((always_inline)) {
      constexpr int k = decltype(kc)::value;
      const int4 r0 = rn;
      rn = crow[(int64_t)(b0 + k + 1) * 4 + g];
      if (k == 0 || b0 + k < b1) issue(kc, r0);
    });
    for (int b = b0; b < b1; b += PD)
      static_for<PD>([&](auto kc) __attribute__((always_inline)) {
        constexpr int k = decltype(kc)::value;
        if (k == 0 || b + k < b1) step(kc, b + k);
      });
  }

  // (an empty range still stores its zero slab: the reduction sums every slab)
  if (PROJ && dvalid) {
#pragma unroll
    for (int f = 0; f < NA; ++f)
#pragma unroll
      for (int t = 0; t < NB; ++t) {
        const int l = sym_harm<TBS>(t, g);
        if (l < K) partial[(((int64_t)split * NFP + f0 + f) * K + l) * D + d] = acc[f][t];
      }
  }
}

// ------------------------------------------------------------------------------------------------
// class eddy / flux sweep (sweep 2).  Per class-group: 4 reconstructions (once), the eddies and
// products of every member row against its side's zonal mean (VALU), 3 projections of the
// per-class product sums (once).  Template parameters as eddy_sym_kernel.
// ------------------------------------------------------------------------------------------------
template <typename T, int TBS, int MODE, int DPW, int KIND>
__global__ void __launch_bounds__(512, 2)
eddy_cls_kernel(FieldPtrs<4> fp, int64_t D, int K, int K4, const double* __restrict__ ycls,
                const int4* __restrict__ crow, const int2* __restrict__ csplit,
                const double* __restrict__ colscale, const double* __restrict__ C,
                double* __restrict__ partial, int nsplit, int ndt, EddyOut eo) {
  extern __shared__ double lds[];
  constexpr int NB = 2 * TBS;
  constexpr int YE = NB * 16;
  constexpr int YJ = (YE + 63) / 64;
  constexpr int NP = 8 / DPW;
  constexpr int NFR = KIND == 0 ? 4 : 3;
  constexpr int NPR = KIND == 0 ? 3 : 2;
  constexpr int MB = CLS_MB;
  int split, dq;
  if (!wg_work((ndt + DPW - 1) / DPW, nsplit, split, dq)) return;
  const int wave = uniform_wave(), lane = threadIdx.x & 63;
  const int w4 = wave % DPW, part = wave / DPW;
  const int c = lane & 15, g = lane >> 4;
  const int dt = dq * DPW + w4;
  const bool active = dt < ndt;               // waves beyond the ragged end idle up to the final barriers
  const int64_t d = (int64_t)dt * 16 + c;
  const bool dvalid = active && d < D;
  const int64_t dcl = dvalid ? d : D - 1;
  const int64_t sub = (int64_t)split * NP + part;
  const int b0 = __builtin_amdgcn_readfirstlane(csplit[sub].x);
  const int b1 = __builtin_amdgcn_readfirstlane(csplit[sub + 1].x);
  int grp = __builtin_amdgcn_readfirstlane(csplit[sub].y);

  // coefficient B operands, even blocks then odd blocks: cb[f][tb][lane] = C_f[harm(tb, g)][d]
  if (active) {
    double* cb = lds + (size_t)w4 * (NFR * NB * 64) + lane;
#pragma unroll
    for (int f = 0; f < NFR; ++f) {
      double v[NB];
#pragma unroll
      for (int tb = 0; tb < NB; ++tb) {
        const int l = sym_harm<TBS>(tb, g);
        const int lc = l < K ? l : K - 1;
        v[tb] = C[((int64_t)f * K4 + lc) * D + dcl];
      }
#pragma unroll
      for (int tb = 0; tb < NB; ++tb) cb[(f * NB + tb) * 64] = sym_harm<TBS>(tb, g) < K ? v[tb] : 0.0;
    }
  }
  int cbi = w4 * (NFR * NB * 64) + lane;
  double* yst = lds + DPW * NFR * NB * 64 + wave * YE;
  // the slab is shared by the NP waves of a d-tile; each wave wrote all of it (same values), so a
  // wave only depends on its own stores -- no barrier

  const double sth = (KIND == 0 && colscale != nullptr) ? colscale[dcl] : 1.0;
  const uint32_t aoff_r = (uint32_t)((lane & 3) * 4 + g);
  const uint32_t aoff_p = (uint32_t)(g * 4 + (lane & 3));
  const T* fb[NFR];
#pragma unroll
  for (int f = 0; f < NFR; ++f) fb[f] = reinterpret_cast<const T*>(fp.p[f]) + dcl;

  double acc[NPR][NB];
#pragma unroll
  for (int q = 0; q < NPR; ++q)
#pragma unroll
    for (int t = 0; t < NB; ++t) acc[q][t] = 0.0;
  double xbN[NFR], xbS[NFR], pN[NPR], pS[NPR];
#pragma unroll
  for (int f = 0; f < NFR; ++f) xbN[f] = xbS[f] = 0.0;
#pragma unroll
  for (int q = 0; q < NPR; ++q) pN[q] = pS[q] = 0.0;

  T xb[2][MB][NFR];
  int er[2][MB];
  int fl[2] = {0, 0};
  double ys[YJ];
  auto load_ys = [&](int gi) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < YJ; ++j) ys[j] = (ycls + (int64_t)gi * YE)[(lane + 64 * j) < YE ? (lane + 64 * j) : 0];
  };
  auto issue = [&](auto pc, const int4 rv) __attribute__((always_inline)) {
    constexpr int P = decltype(pc)::value;
    er[P][0] = rv.x; er[P][1] = rv.y; er[P][2] = rv.z; er[P][3] = rv.w;
    fl[P] = __builtin_amdgcn_readfirstlane(rv.x) >> 28;
#pragma unroll
    for (int j = 0; j < MB; ++j) {
      const int64_t off = (int64_t)(er[P][j] & CLS_ROWMASK) * D;
#pragma unroll
      for (int f = 0; f < NFR; ++f) xb[P][j][f] = TEMX_XLOAD(fb[f] + off);
    }
  };
  int4 rn;
  auto step = [&](auto pc, int b) __attribute__((always_inline)) {
    constexpr int P = decltype(pc)::value;
    if (b + 1 < b1) {
      const int4 r1 = rn;
      rn = crow[(int64_t)(b + 2) * 4 + g];
      issue(std::integral_constant<int, P ^ 1>{}, r1);
    }
    // a piece may begin / end inside a group: reconstruct there, project the partial product sums
    const int flags = fl[P] | (b == b0 ? CLS_FIRST : 0) | (b + 1 == b1 ? CLS_LAST : 0);
    if (flags & CLS_FIRST) {
      // ---- reconstruction at the class latitudes: E = even-harmonic part, O = odd-harmonic part ----
#pragma unroll
      for (int j = 0; j < YJ; ++j)
        if (lane + 64 * j < YE) yst[lane + 64 * j] = ys[j];
      load_ys(grp + 1);                        // ycls is padded by one group
      asm volatile("" : "+v"(cbi));            // keep the loop-invariant slab reads inside the loop
      const double* cbr = lds + cbi;
      double E[NFR], O[NFR];
#pragma unroll
      for (int f = 0; f < NFR; ++f) E[f] = O[f] = 0.0;
#pragma unroll
      for (int tb = 0; tb < NB; ++tb) {
        const double ya = yst[tb * 16 + aoff_r];
#pragma unroll
        for (int f = 0; f < NFR; ++f) {
          if (tb < TBS)
            E[f] = TEMX_MFMA4(ya, cbr[(f * NB + tb) * 64], E[f]);
          else
            O[f] = TEMX_MFMA4(ya, cbr[(f * NB + tb) * 64], O[f]);
        }
      }
#pragma unroll
      for (int f = 0; f < NFR; ++f) {
        xbN[f] = E[f] + O[f];
        xbS[f] = E[f] - O[f];
      }
    }
    // ---- eddies and products of this batch's member rows ----
    const bool south = (flags & CLS_SOUTH) != 0;
#pragma unroll
    for (int j = 0; j < MB; ++j) {
      const int ent = er[P][j];
      const double w = ent < 0 ? 0.0 : 1.0;
      double e[NFR], p[NPR];
#pragma unroll
      for (int f = 0; f < NFR; ++f) {
        double x = (double)xb[P][j][f];
        if (KIND == 0 && f == 2) x *= sth;
        e[f] = x - (south ? xbS[f] : xbN[f]);
      }
      if (KIND == 0) {
        p[0] = e[0] * e[1];
        p[1] = e[0] * e[NFR - 1];
        p[NPR - 1] = e[1] * e[2];
      } else {
        p[0] = e[0] * e[1];
        p[1] = e[0] * e[2];
      }
      if (south) {
#pragma unroll
        for (int q = 0; q < NPR; ++q) pS[q] += w * p[q];
      } else {
#pragma unroll
        for (int q = 0; q < NPR; ++q) pN[q] += w * p[q];
      }
      if (MODE == 1) {
        if (dvalid && ent >= 0) {
          const int64_t o = (int64_t)(ent & CLS_ROWMASK) * D + d;
          if (KIND == 0) {
#pragma unroll
            for (int f = 0; f < NFR; ++f)
              if (eo.p[f]) eo.p[f][o] = e[f];
          } else if (eo.p[0]) {
            eo.p[0][o] = e[0];
          }
#pragma unroll
          for (int q = 0; q < NPR; ++q)
            if (eo.p[4 + q]) eo.p[4 + q][o] = p[q];
        }
      }
    }
    if (flags & CLS_LAST) {
      // ---- projection: even harmonics see the class sums, odd harmonics the N - S differences ----
      double sp[NPR], dp[NPR];
#pragma unroll
      for (int q = 0; q < NPR; ++q) {
        sp[q] = pN[q] + pS[q];
        dp[q] = pN[q] - pS[q];
        pN[q] = pS[q] = 0.0;
      }
#pragma unroll
      for (int tb = 0; tb < NB; ++tb) {
        const double ya = yst[tb * 16 + aoff_p];
#pragma unroll
        for (int q = 0; q < NPR; ++q) acc[q][tb] = TEMX_MFMA4(ya, tb < TBS ? sp[q] : dp[q], acc[q][tb]);
      }
      ++grp;
    }
  };

  if (active && b0 < b1) {
    load_ys(grp);
    rn = crow[(int64_t)b0 * 4 + g];
    const int4 r0 = rn;
    rn = crow[(int64_t)(b0 + 1) * 4 + g];
    issue(std::integral_constant<int, 0>{}, r0);
    for (int b = b0; b < b1; b += 2) {
      step(std::integral_constant<int, 0>{}, b);
      if (b + 1 < b1) step(std::integral_constant<int, 1>{}, b + 1);
    }
  }

  if (partial == nullptr) return;             // materialising launch (uniform: same for the whole grid)
  // ---- the NP waves that split this d-tile's batch range add up in LDS, in a fixed order, so the
  //      workgroup stores one slab per (split, d-tile) instead of NP
  __syncthreads();                             // every wave is done with the coefficient slab
  {
    double* red = lds + (size_t)w4 * (NFR * NB * 64) + lane;
    for (int pw = 0; pw < NP; ++pw) {
      if (part == pw) {
#pragma unroll
        for (int q = 0; q < NPR; ++q)
#pragma unroll
          for (int tb = 0; tb < NB; ++tb) {
            double* r = red + (q * NB + tb) * 64;
            *r = pw == 0 ? acc[q][tb] : *r + acc[q][tb];
          }
      }
      __syncthreads();
    }
    if (dvalid) {
#pragma unroll
      for (int q = 0; q < NPR; ++q)
#pragma unroll
        for (int tb = 0; tb < NB; ++tb) {
          const int l = sym_harm<TBS>(tb, g);
          if ((q * NB + tb) % NP == part && l < K)
            partial[(((int64_t)split * NPR + q) * K + l) * D + d] = red[(q * NB + tb) * 64];
        }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// flux_cls_kernel: second phase of the one-pass class path.  Inside a class side the zonal mean xb is a
// constant, so with the side's mean m = S / n and its centred co-moment C_uv = sum (u - m_u)(v - m_v)
//     sum (u - ub)(v - vb) = C_uv + n (m_u - ub)(m_v - vb)          (tem_diagnostics.py:547-557)
// Sweep 1 projected the co-moments next to the four fields (they enter linearly); this kernel adds the
// second term, which needs only the four field sums of every class side.  Per (class-group, d-tile):
// reconstruct the zonal means at the class latitudes (4 x 14 MFMAs), form n (m_u - ub)(m_v - vb)
// (and the same for u omega, v theta) and project it (3 x 14 MFMAs).  Reads 8 x 512 B per (group,
// d-tile) instead of the fields; the caller adds the co-moment projections of sweep 1 (launch_reduce
// with an addend).  No term is a difference of large numbers.
// ------------------------------------------------------------------------------------------------
// KIND 1 (tracer): fields (q, v, omega) -- the sums of q come from csq (records of one pair), those of v and
// omega from rows 1 and 3 of the TEM run's csum; products q'v', q'omega'.
template <int TBS, int DPW, int KIND = 0>
__global__ void __launch_bounds__(512, 2)
flux_cls_kernel(int64_t D, int K, int K4, const double* __restrict__ ycls, const double* __restrict__ csum,
                const double* __restrict__ csq, const double* __restrict__ ccnt, int64_t ngroups,
                const double* __restrict__ C, double* __restrict__ partial, int nsplit, int ndt) {
  extern __shared__ double lds[];
  constexpr int NB = 2 * TBS;
  constexpr int YE = NB * 16;
  constexpr int YJ = (YE + 63) / 64;
  constexpr int NP = 8 / DPW;
  constexpr int NFR = KIND == 0 ? 4 : 3, NPR = KIND == 0 ? 3 : 2;
  int split, dq;
  if (!wg_work((ndt + DPW - 1) / DPW, nsplit, split, dq)) return;
  const int wave = uniform_wave(), lane = threadIdx.x & 63;
  const int w4 = wave % DPW, part = wave / DPW;
  const int c = lane & 15, g = lane >> 4;
  const int dt = dq * DPW + w4;
  const bool active = dt < ndt;
  const int64_t d = (int64_t)dt * 16 + c;
  const bool dvalid = active && d < D;
  const int64_t dcl = dvalid ? d : D - 1;
  const int64_t sub = (int64_t)split * NP + part, nsub = (int64_t)nsplit * NP;
  const int g0 = (int)(ngroups * sub / nsub), g1 = (int)(ngroups * (sub + 1) / nsub);

  if (active) {
    double* cb = lds + (size_t)w4 * (NFR * NB * 64) + lane;
#pragma unroll
    for (int f = 0; f < NFR; ++f) {
      double v[NB];
#pragma unroll
      for (int tb = 0; tb < NB; ++tb) {
        const int l = sym_harm<TBS>(tb, g);
        const int lc = l < K ? l : K - 1;
        v[tb] = C[((int64_t)f * K4 + lc) * D + dcl];
      }
#pragma unroll
      for (int tb = 0; tb < NB; ++tb) cb[(f * NB + tb) * 64] = sym_harm<TBS>(tb, g) < K ? v[tb] : 0.0;
    }
  }
  int cbi = w4 * (NFR * NB * 64) + lane;
  double* yst = lds + DPW * NFR * NB * 64 + wave * YE;
  const uint32_t aoff_r = (uint32_t)((lane & 3) * 4 + g);
  const uint32_t aoff_p = (uint32_t)(g * 4 + (lane & 3));

  double acc[NPR][NB];
#pragma unroll
  for (int q = 0; q < NPR; ++q)
#pragma unroll
    for (int t = 0; t < NB; ++t) acc[q][t] = 0.0;

  double sv[2][2 * NFR], cn[2][2], ys[2][YJ];
  auto load = [&](auto pc, int gi) __attribute__((always_inline)) {
    constexpr int P = decltype(pc)::value;
    const double2* base = reinterpret_cast<const double2*>(csum + (((int64_t)gi * ndt + (active ? dt : 0)) * 8) * 64) + lane;
#pragma unroll
    for (int j = 0; j < YJ; ++j) ys[P][j] = (ycls + (int64_t)gi * YE)[(lane + 64 * j) < YE ? (lane + 64 * j) : 0];
    cn[P][0] = ccnt[(int64_t)gi * 8 + g];
    cn[P][1] = ccnt[(int64_t)gi * 8 + 4 + g];
#pragma unroll
    for (int s_ = 0; s_ < NFR; ++s_) {
      double2 v2;
      if (KIND == 0)
        v2 = base[s_ * 64];
      else if (s_ == 0)           // q: its own records of one {north, south} pair
        v2 = (reinterpret_cast<const double2*>(csq + (((int64_t)gi * ndt + (active ? dt : 0)) * 2) * 64) + lane)[0];
      else                        // v, omega: rows 1 and 3 of the TEM record
        v2 = base[(s_ == 1 ? 1 : 3) * 64];
      sv[P][s_] = v2.x;           // northern members
      sv[P][NFR + s_] = v2.y;     // southern members
    }
  };
  auto step = [&](auto pc, int gi) __attribute__((always_inline)) {
    constexpr int P = decltype(pc)::value;
    if (gi + 1 < g1) load(std::integral_constant<int, P ^ 1>{}, gi + 1);
#pragma unroll
    for (int j = 0; j < YJ; ++j)
      if (lane + 64 * j < YE) yst[lane + 64 * j] = ys[P][j];
    asm volatile("" : "+v"(cbi));
    const double* cbr = lds + cbi;
    double E[NFR], O[NFR];
#pragma unroll
    for (int f = 0; f < NFR; ++f) E[f] = O[f] = 0.0;
#pragma unroll
    for (int tb = 0; tb < NB; ++tb) {
      const double ya = yst[tb * 16 + aoff_r];
#pragma unroll
      for (int f = 0; f < NFR; ++f) {
        if (tb < TBS)
          E[f] = TEMX_MFMA4(ya, cbr[(f * NB + tb) * 64], E[f]);
        else
          O[f] = TEMX_MFMA4(ya, cbr[(f * NB + tb) * 64], O[f]);
      }
    }
    double pr[2][NPR];
#pragma unroll
    for (int sd = 0; sd < 2; ++sd) {
      const double* S = sv[P] + NFR * sd;
      const double n = cn[P][sd], rn = n > 0.0 ? 1.0 / n : 0.0;
      // class-side mean minus the zonal mean at the class latitude (E + O north, E - O south)
      double dl[NFR];
#pragma unroll
      for (int f = 0; f < NFR; ++f) dl[f] = S[f] * rn - (sd ? E[f] - O[f] : E[f] + O[f]);
      if (KIND == 0) {
        pr[sd][0] = n * dl[0] * dl[1];             // sum u'v'     - C_uv
        pr[sd][1] = n * dl[0] * dl[NFR - 1];       // sum u'omega' - C_uw
        pr[sd][NPR - 1] = n * dl[1] * dl[2];       // sum v'theta' - C_vtheta
      } else {
        pr[sd][0] = n * dl[0] * dl[1];             // sum q'v'     - C_qv
        pr[sd][1] = n * dl[0] * dl[2];             // sum q'omega' - C_qw
      }
    }
#pragma unroll
    for (int tb = 0; tb < NB; ++tb) {
      const double ya = yst[tb * 16 + aoff_p];
#pragma unroll
      for (int q = 0; q < NPR; ++q)
        acc[q][tb] = TEMX_MFMA4(ya, tb < TBS ? pr[0][q] + pr[1][q] : pr[0][q] - pr[1][q], acc[q][tb]);
    }
  };
  if (active && g0 < g1) {
    load(std::integral_constant<int, 0>{}, g0);
    for (int gi = g0; gi < g1; gi += 2) {
      step(std::integral_constant<int, 0>{}, gi);
      if (gi + 1 < g1) step(std::integral_constant<int, 1>{}, gi + 1);
    }
  }

  __syncthreads();
  {
    double* red = lds + (size_t)w4 * (NFR * NB * 64) + lane;
    for (int pw = 0; pw < NP; ++pw) {
      if (part == pw) {
#pragma unroll
        for (int q = 0; q < NPR; ++q)
#pragma unroll
          for (int tb = 0; tb < NB; ++tb) {
            double* r = red + (q * NB + tb) * 64;
            *r = pw == 0 ? acc[q][tb] : *r + acc[q][tb];
          }
      }
      __syncthreads();
    }
    if (dvalid) {
#pragma unroll
      for (int q = 0; q < NPR; ++q)
#pragma unroll
        for (int tb = 0; tb < NB; ++tb) {
          const int l = sym_harm<TBS>(tb, g);
          if ((q * NB + tb) % NP == part && l < K)
            partial[(((int64_t)split * NPR + q) * K + l) * D + d] = red[(q * NB + tb) * 64];
        }
    }
  }
}

// Tried and measured, not kept (round 3): the same phase with the 4 x 14 coefficient operands of a d-tile in
// REGISTERS (one wave per SIMD, 244 VGPR + 100 AGPR, LDS serving only the 2 x 14 Y operands of a group).
// flux_cls_kernel reads one LDS operand per reconstruction MFMA and its LDS traffic (43 KB per class-group
// and d-tile) takes about as long as its MFMAs at two waves per SIMD -- 1.79 ms on ne120 x 72 x 30 against
// an MFMA bound of 1.05 ms -- but the register form ran at 2.02 ms: a single in-order wave does not overlap
// its loads, LDS staging and product arithmetic with the matrix pipe (profiles/r03_ab_flux_registers.log).
// ------------------------------------------------------------------------------------------------
// Large-L class path (64 < K <= 256 on a grid with latitude classes): the class sums do not depend on
// L, so the fields are swept once (project_cls_kernel<.., OP, PROJ = false>) and everything else
// works on the sums, 64 harmonics (8 even + 8 odd blocks) per slice:
//   cls_basis_slice_kernel   ycls_l[slice][group][16][16], harmonics 64 s ... 64 s + 63
//   sums_project_kernel<NQ>  partial[split][q][l - 64 s][d] = sum_c Y_l(c) (S_q,N +- S_q,S) for NQ sums of
//                            a record (the 4 field sums of csum, or the 3 product sums of pbuf)
//   flux_large_kernel        xbar at the class latitudes from all slices, algebraic eddy-product sums
//                            -> pbuf[group][dt][3][64][2]
// ------------------------------------------------------------------------------------------------
template <int KMAX>
__global__ void cls_basis_slice_kernel(const double* __restrict__ xc, int64_t ncls, int64_t ncls_pad, int K,
                                       int l0, const double* __restrict__ norm, const double* __restrict__ T,
                                       double* __restrict__ ycls) {
  int64_t ci = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (ci >= ncls_pad) return;
  const bool valid = ci < ncls;
  const double xv = valid ? xc[ci] : 0.0;
  double* blk = ycls + (ci >> 2) * (16 * 16);
  const int k = (int)(ci & 3);
  double q[KMAX];
  basis_row<KMAX>(xv, K, norm, T, q);
  for (int l = l0; l < l0 + 64; ++l) {
    const double val = (valid && l < K) ? q[l] : 0.0;
    const int h = (l - l0) >> 1;
    const int t = ((l - l0) & 1) * 8 + (h >> 2);
    blk[t * 16 + k * 4 + (h & 3)] = val;
  }
}

template <int NQ, int DPW>
__global__ void __launch_bounds__(512, 2)
sums_project_kernel(int64_t D, int K, int l0, const double* __restrict__ ycls /* this slice */,
                    const double* __restrict__ rec /* [group][dt][RS][64][2] */, int RS, int row0,
                    int64_t ngroups, double* __restrict__ partial, int nsplit, int ndt) {
  extern __shared__ double lds[];
  constexpr int TBS = 8, NB = 16, YE = NB * 16, YJ = YE / 64, NP = 8 / DPW;
  int split, dq;
  if (!wg_work((ndt + DPW - 1) / DPW, nsplit, split, dq)) return;
  const int wave = uniform_wave(), lane = threadIdx.x & 63;
  const int w4 = wave % DPW, part = wave / DPW;
  const int c = lane & 15, g = lane >> 4;
  const int dt = dq * DPW + w4;
  const bool active = dt < ndt;
  const int64_t d = (int64_t)dt * 16 + c;
  const bool dvalid = active && d < D;
  const int64_t sub = (int64_t)split * NP + part, nsub = (int64_t)nsplit * NP;
  const int g0 = (int)(ngroups * sub / nsub), g1 = (int)(ngroups * (sub + 1) / nsub);
  double* yst = lds + DPW * NQ * NB * 64 + wave * YE;
  const uint32_t aoff_p = (uint32_t)(g * 4 + (lane & 3));
  double acc[NQ][NB];
#pragma unroll
  for (int q = 0; q < NQ; ++q)
#pragma unroll
    for (int t = 0; t < NB; ++t) acc[q][t] = 0.0;
  double2 sv[2][NQ];
  double ys[2][YJ];
  auto load = [&](auto pc, int gi) __attribute__((always_inline)) {
    constexpr int P = decltype(pc)::value;
    const double2* base = reinterpret_cast<const double2*>(rec + (((int64_t)gi * ndt + dt) * RS + row0) * 128) + lane;
#pragma unroll
    for (int j = 0; j < YJ; ++j) ys[P][j] = (ycls + (int64_t)gi * YE)[lane + 64 * j];
#pragma unroll
    for (int q = 0; q < NQ; ++q) sv[P][q] = base[q * 64];
  };
  auto step = [&](auto pc, int gi) __attribute__((always_inline)) {
    constexpr int P = decltype(pc)::value;
    if (gi + 1 < g1) load(std::integral_constant<int, P ^ 1>{}, gi + 1);
#pragma unroll
    for (int j = 0; j < YJ; ++j) yst[lane + 64 * j] = ys[P][j];
#pragma unroll
    for (int tb = 0; tb < NB; ++tb) {
      const double ya = yst[tb * 16 + aoff_p];
#pragma unroll
      for (int q = 0; q < NQ; ++q)
        acc[q][tb] = TEMX_MFMA4(ya, tb < TBS ? sv[P][q].x + sv[P][q].y : sv[P][q].x - sv[P][q].y, acc[q][tb]);
    }
  };
  if (active && g0 < g1) {
    load(std::integral_constant<int, 0>{}, g0);
    for (int gi = g0; gi < g1; gi += 2) {
      step(std::integral_constant<int, 0>{}, gi);
      if (gi + 1 < g1) step(std::integral_constant<int, 1>{}, gi + 1);
    }
  }
  __syncthreads();
  {
    double* red = lds + (size_t)w4 * (NQ * NB * 64) + lane;
    for (int pw = 0; pw < NP; ++pw) {
      if (part == pw) {
#pragma unroll
        for (int q = 0; q < NQ; ++q)
#pragma unroll
          for (int tb = 0; tb < NB; ++tb) {
            double* r = red + (q * NB + tb) * 64;
            *r = pw == 0 ? acc[q][tb] : *r + acc[q][tb];
          }
      }
      __syncthreads();
    }
    const int Ks = K - l0 < 64 ? K - l0 : 64;
    if (dvalid) {
#pragma unroll
      for (int q = 0; q < NQ; ++q)
#pragma unroll
        for (int tb = 0; tb < NB; ++tb) {
          const int ll = sym_harm<TBS>(tb, g);          // harmonic inside the slice
          if ((q * NB + tb) % NP == part && ll < Ks)
            partial[(((int64_t)split * NQ + q) * Ks + ll) * D + d] = red[(q * NB + tb) * 64];
        }
    }
  }
}

// one workgroup = one d-tile (8 waves split the class-groups); the coefficients of all NS slices of the
// four fields stay in LDS (NS * 32 KB)
template <int NS>
__global__ void __launch_bounds__(512, 2)
flux_large_kernel(int64_t D, int K, int K4, const double* __restrict__ ycls_l /* [NS][groups+1][16][16] */,
                  int64_t gstride /* doubles per slice of ycls_l */, const double* __restrict__ csum,
                  const double* __restrict__ ccnt, int64_t ngroups, const double* __restrict__ C,
                  double* __restrict__ pbuf, int nsplit, int ndt) {
  extern __shared__ double lds[];
  constexpr int TBS = 8, NB = 16, YE = NB * 16, YJ = YE / 64, NFR = 4;
  int split, dt;
  if (!wg_work(ndt, nsplit, split, dt)) return;
  const int wave = uniform_wave(), lane = threadIdx.x & 63;
  const int c = lane & 15, g = lane >> 4;
  const int64_t d = (int64_t)dt * 16 + c;
  const bool dvalid = d < D;
  const int64_t dcl = dvalid ? d : D - 1;
  const int64_t sub = (int64_t)split * 8 + wave, nsub = (int64_t)nsplit * 8;
  const int g0 = (int)(ngroups * sub / nsub), g1 = (int)(ngroups * (sub + 1) / nsub);
  // coefficient B operands of every slice: cb[s][f][tb][lane] = C_f[64 s + harm(tb, g)][d]
  {
    double* cb = lds + lane;
    for (int s = 0; s < NS; ++s)
#pragma unroll
      for (int f = 0; f < NFR; ++f) {
        double v[NB];
#pragma unroll
        for (int tb = 0; tb < NB; ++tb) {
          const int l = 64 * s + sym_harm<TBS>(tb, g);
          const int lc = l < K ? l : K - 1;
          v[tb] = C[((int64_t)f * K4 + lc) * D + dcl];
        }
#pragma unroll
        for (int tb = 0; tb < NB; ++tb)
          cb[((s * NFR + f) * NB + tb) * 64] = 64 * s + sym_harm<TBS>(tb, g) < K ? v[tb] : 0.0;
      }
  }
  int cbi = lane;
  double* yst = lds + NS * NFR * NB * 64 + wave * YE;
  const uint32_t aoff_r = (uint32_t)((lane & 3) * 4 + g);
  for (int gi = g0; gi < g1; ++gi) {
    const double2* base = reinterpret_cast<const double2*>(csum + (((int64_t)gi * ndt + dt) * 14) * 64) + lane;
    double2 S[7];
#pragma unroll
    for (int s_ = 0; s_ < 7; ++s_) S[s_] = base[s_ * 64];
    const double nN = ccnt[(int64_t)gi * 8 + g], nS = ccnt[(int64_t)gi * 8 + 4 + g];
    double E[NFR], O[NFR];
#pragma unroll
    for (int f = 0; f < NFR; ++f) E[f] = O[f] = 0.0;
    for (int s = 0; s < NS; ++s) {
      double ys[YJ];
#pragma unroll
      for (int j = 0; j < YJ; ++j) ys[j] = (ycls_l + s * gstride + (int64_t)gi * YE)[lane + 64 * j];
#pragma unroll
      for (int j = 0; j < YJ; ++j) yst[lane + 64 * j] = ys[j];
      asm volatile("" : "+v"(cbi));
      const double* cbr = lds + cbi + (size_t)s * (NFR * NB * 64);
#pragma unroll
      for (int tb = 0; tb < NB; ++tb) {
        const double ya = yst[tb * 16 + aoff_r];
#pragma unroll
        for (int f = 0; f < NFR; ++f) {
          if (tb < TBS)
            E[f] = TEMX_MFMA4(ya, cbr[(f * NB + tb) * 64], E[f]);
          else
            O[f] = TEMX_MFMA4(ya, cbr[(f * NB + tb) * 64], O[f]);
        }
      }
    }
    double2 pr[3];
    {
      const double ub = E[0] + O[0], vb = E[1] + O[1], tb_ = E[2] + O[2], wb = E[3] + O[3];
      pr[0].x = S[4].x - vb * S[0].x - ub * S[1].x + nN * ub * vb;
      pr[1].x = S[5].x - wb * S[0].x - ub * S[3].x + nN * ub * wb;
      pr[2].x = S[6].x - tb_ * S[1].x - vb * S[2].x + nN * vb * tb_;
    }
    {
      const double ub = E[0] - O[0], vb = E[1] - O[1], tb_ = E[2] - O[2], wb = E[3] - O[3];
      pr[0].y = S[4].y - vb * S[0].y - ub * S[1].y + nS * ub * vb;
      pr[1].y = S[5].y - wb * S[0].y - ub * S[3].y + nS * ub * wb;
      pr[2].y = S[6].y - tb_ * S[1].y - vb * S[2].y + nS * vb * tb_;
    }
    if (dvalid) {
      double2* o = reinterpret_cast<double2*>(pbuf + (((int64_t)gi * ndt + dt) * 3) * 128) + lane;
#pragma unroll
      for (int q = 0; q < 3; ++q) o[q * 64] = pr[q];
    }
  }
}

}  // namespace temx
