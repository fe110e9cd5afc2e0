// lab_kernels.hpp -- forms of sweep 1 of the one-pass latitude-class path that were built and measured in round 3
//  and are NOT launched by libtemx.so; kept for tools/sweep_lab.hip only.
//
//  * row pairs (sweep_op16_kernel): a lane owns two adjacent columns of a member row and the two halves
//    of a 16-lane group own the northern and the southern side of a class: lane = 16 k + 8 h + cp (class
//    k of the group, side h, column pair cp).  One load instruction is 8 rows x 128 B (fp64; 16 bytes per
//    lane) instead of 4 rows x 128 B, both sides of a class-group are walked at once, and the only
//    cross-lane traffic is one swap with lane ^ 8 per finished sum (DPP), after which lane (k, h, cp)
//    holds the {north, south} pair of column 2 cp + h -- the MFMA operand layout with the columns of the
//    d-tile permuted (a permutation of the 16 independent columns of the product).
//    Row table: crow16[batch][8 lane groups][MBV], same entry format as crow (built by the lab only).
//
//  * parity pair with redundant loads (sweep_opp_kernel, VERDICT r02's option (a)): 48.8 ms.  Loads served
//    by L1 / L2 instead of HBM are far from free at this rate.
#pragma once
#include "../pytemdiags_amd/csrc/kernels_op2.hpp"

namespace temx {

// ------------------------------------------------------------------------------------------------
// row pairs: lane = 16 k + 8 h + cp reads columns 2 cp, 2 cp + 1 of a member row of side h of class k
// ------------------------------------------------------------------------------------------------
template <typename T> struct Pair2;
template <> struct Pair2<double> { using type = double2; };
template <> struct Pair2<float> { using type = float2; };

// value of lane ^ 8 (the other side of the class): row_ror:8 inside the row of 16 lanes
__device__ __forceinline__ double swap8(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(0, lo, 0x128, 0xF, 0xF, false);
  hi = __builtin_amdgcn_update_dpp(0, hi, 0x128, 0xF, 0xF, false);
  return __hiloint2double(hi, lo);
}

template <int MBV> struct RowVec;
template <> struct RowVec<2> { using type = int2; };
template <> struct RowVec<4> { using type = int4; };

// WS = 1: a workgroup covers four d-tiles, every wave holds all NA x 2 TBS accumulators (one wave per SIMD)
// WS = 4: the four waves share one d-tile as in sweep_opw_kernel (two waves per SIMD)
// needs D even and 2 * sizeof(T)-aligned field pointers (the launcher checks)
// RB > 0 (WS = 1): the records of RB consecutive class-groups are collected in LDS and stored together
template <typename T, int TBS, int MBV, int PD, int KIND, int WS, int RB = 0>
__global__ void __launch_bounds__(256, WS == 1 ? 1 : 2)
sweep_op16_kernel(FieldPtrs<4> fp, int64_t D, int K, const double* __restrict__ ycls,
                  const int* __restrict__ crow16, const int2* __restrict__ csplit,
                  const double* __restrict__ colscale, double* __restrict__ partial, int nsplit, int ndt,
                  double* __restrict__ csum) {
  using KD = OpKind<KIND>;
  using V2 = typename Pair2<T>::type;
  using RV = typename RowVec<MBV>::type;
  constexpr int NB = 2 * TBS;
  constexpr int YE = NB * 16;
  constexpr int YJ = (YE + 63) / 64;
  constexpr int NFLD = KD::NFLD, NST = KD::NST, NQ = KD::NQ;
  constexpr int NA = NST + NQ;
  constexpr int NBW = (TBS + 1) / 2;
  constexpr int NBA = WS == 1 ? NB : NBW;     // accumulator blocks per wave
  static_assert(WS == 1 || WS == 4, "one d-tile per wave or one per workgroup");
  static_assert(PD + 1 <= CLS_PADB, "table padding must cover the index prefetch");
  using Shared = typename std::conditional<WS == 1, double[4][YE], OpwSlot<NA, NB>[2][4]>::type;
  __shared__ Shared sh;
  static_assert(RB == 0 || WS == 1, "record buffering is for the one-wave-per-d-tile form");
  __shared__ double2 recbuf[RB > 0 ? 4 * RB * NST * 64 : 1];
  int nbuf = 0, grp_buf0 = 0;
  int split, dq;
  if (!wg_work(WS == 1 ? (ndt + 3) / 4 : ndt, nsplit, split, dq)) return;
  const int wave = uniform_wave(), lane = threadIdx.x & 63;
  const int g = lane >> 4, h = (lane >> 3) & 1, cp = lane & 7, lg = lane >> 3;
  const int dt = WS == 1 ? dq * 4 + wave : dq;
  if (dt >= ndt) return;                      // (WS = 1 only; no barriers there)
  const int64_t d0 = (int64_t)dt * 16 + 2 * cp;           // the lane's column pair
  const bool pvalid = d0 < D;                              // D is even: both columns or neither
  const int64_t dcl = pvalid ? d0 : D - 2;
  const int64_t dmine = d0 + h;                            // the column this lane holds after the swap
  const int piece = WS == 1 ? split : split * 4 + wave;
  const int b0 = __builtin_amdgcn_readfirstlane(csplit[piece].x);
  const int b1 = __builtin_amdgcn_readfirstlane(csplit[piece + 1].x);
  int grp = __builtin_amdgcn_readfirstlane(csplit[piece].y);
  int rounds = 0;
  if constexpr (WS == 4) {
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const int n = __builtin_amdgcn_readfirstlane(csplit[split * 4 + w + 1].y - csplit[split * 4 + w].y);
      rounds = n > rounds ? n : rounds;
    }
  }
  const int par = wave >> 1;
  const int t0 = WS == 1 ? 0 : par * TBS + (wave & 1) * NBW;
  const int nbw = WS == 1 ? NB : ((wave & 1) ? TBS - NBW : NBW);
  const uint32_t yoff = (uint32_t)(g * 4 + (lane & 3));
  double sth[2] = {1.0, 1.0};
  if (KIND == 0 && colscale != nullptr) {
    sth[0] = colscale[dcl];
    sth[1] = colscale[dcl + 1];
  }
  const T* fb[NFLD];
#pragma unroll
  for (int f = 0; f < NFLD; ++f) fb[f] = reinterpret_cast<const T*>(fp.p[f]) + dcl;

  double acc[NA][NBA];
#pragma unroll
  for (int f = 0; f < NA; ++f)
#pragma unroll
    for (int t = 0; t < NBA; ++t) acc[f][t] = 0.0;
  double s[NFLD][2], q[NQ][2], x0[NFLD][2], cnt = 0.0;
#pragma unroll
  for (int f = 0; f < NFLD; ++f) s[f][0] = s[f][1] = x0[f][0] = x0[f][1] = 0.0;
#pragma unroll
  for (int k = 0; k < NQ; ++k) q[k][0] = q[k][1] = 0.0;
  const uint32_t D32 = (uint32_t)D;
  int buf = 0, done = 0;

  V2 xb[PD][MBV][NFLD];
  int er[PD][MBV];
  double ys[YJ];
  auto load_ys = [&](int gi) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < YJ; ++j) ys[j] = (ycls + (int64_t)gi * YE)[(lane + 64 * j) < YE ? (lane + 64 * j) : 0];
  };
  auto rows_of = [&](int b) __attribute__((always_inline)) {
    return reinterpret_cast<const RV*>(crow16)[(int64_t)b * 8 + lg];
  };
  auto issue = [&](auto pc, const RV rv) __attribute__((always_inline)) {
    constexpr int P = decltype(pc)::value;
    er[P][0] = rv.x; er[P][1] = rv.y;
    if constexpr (MBV == 4) { er[P][2] = rv.z; er[P][3] = rv.w; }
#pragma unroll
    for (int j = 0; j < MBV; ++j) {
      const uint64_t off = (uint64_t)(uint32_t)(er[P][j] & CLS_ROWMASK) * D32;
#pragma unroll
      for (int f = 0; f < NFLD; ++f) xb[P][j][f] = TEMX_XLOAD(reinterpret_cast<const V2*>(fb[f] + off));
    }
  };
  auto flush_records = [&]() __attribute__((always_inline)) {
    if constexpr (RB > 0) {
      const bool lvalid = (int64_t)dt * 16 + (lane & 15) < D;
      for (int r = 0; r < nbuf; ++r) {
        const double2* src = recbuf + (wave * RB + r) * (NST * 64) + lane;
        double2* o = reinterpret_cast<double2*>(csum + TEMX_CSUM_REC(grp_buf0 + r, dt, ndt) * (2 * NST) * 64) + lane;
        double2 v[NST];
#pragma unroll
        for (int f = 0; f < NST; ++f) v[f] = src[f * 64];
        if (lvalid) {
#pragma unroll
          for (int f = 0; f < NST; ++f) TEMX_CSTORE(o + f * 64, v[f]);
        }
      }
      nbuf = 0;
    }
  };
  RV rn;
  auto step = [&](auto pc, int b) __attribute__((always_inline)) {
    constexpr int P = decltype(pc)::value;
    if (b + (PD - 1) < b1) {                  // index load first: it must not queue behind the X loads
      const RV r1 = rn;
      rn = rows_of(b + PD);
      issue(std::integral_constant<int, (P + PD - 1) % PD>{}, r1);
    }
    const int fl = __builtin_amdgcn_readfirstlane(er[P][0]) >> 27;    // haspad, -, first, last
    if (fl & (CLS_FIRST << 1)) {              // first batch of the group: each side's first member is its origin
#pragma unroll
      for (int f = 0; f < NFLD; ++f) {
        x0[f][0] = (double)xb[P][0][f].x;
        x0[f][1] = (double)xb[P][0][f].y;
      }
    }
    if (fl & 1) {                             // a padding entry reads row 0 and weighs nothing
#pragma unroll
      for (int j = 0; j < MBV; ++j) {
        const double w = er[P][j] < 0 ? 0.0 : 1.0;
        double dx[NFLD][2];
#pragma unroll
        for (int f = 0; f < NFLD; ++f) {
          dx[f][0] = (double)xb[P][j][f].x - x0[f][0];
          dx[f][1] = (double)xb[P][j][f].y - x0[f][1];
        }
#pragma unroll
        for (int cc = 0; cc < 2; ++cc) {
#pragma unroll
          for (int f = 0; f < NFLD; ++f) s[f][cc] += w * dx[f][cc];
#pragma unroll
          for (int k = 0; k < NQ; ++k) q[k][cc] += (w * dx[KD::pa(k)][cc]) * dx[KD::pb(k)][cc];
        }
        cnt += w;
      }
    } else {
#pragma unroll
      for (int j = 0; j < MBV; ++j) {
        double dx[NFLD][2];
#pragma unroll
        for (int f = 0; f < NFLD; ++f) {
          dx[f][0] = (double)xb[P][j][f].x - x0[f][0];
          dx[f][1] = (double)xb[P][j][f].y - x0[f][1];
        }
#pragma unroll
        for (int cc = 0; cc < 2; ++cc) {
#pragma unroll
          for (int f = 0; f < NFLD; ++f) s[f][cc] += dx[f][cc];
#pragma unroll
          for (int k = 0; k < NQ; ++k) q[k][cc] += dx[KD::pa(k)][cc] * dx[KD::pb(k)][cc];
        }
      }
      cnt += (double)MBV;
    }
    if (fl & (CLS_LAST << 1)) {
      // ---- both sides of the group are complete: true sums and centred co-moments of this lane's side
      const double rcn = cnt > 0.0 ? 1.0 / cnt : 0.0;
      double keep[NA], recv[NA];
#pragma unroll
      for (int i = 0; i < NA; ++i) {
        double v[2];
#pragma unroll
        for (int cc = 0; cc < 2; ++cc) {
          if (i < NST) {
            v[cc] = s[i][cc] + cnt * x0[i][cc];
            if (KIND == 0 && i == 2) v[cc] *= sth[cc];
          } else {
            const int k = i - NST;
            v[cc] = q[k][cc] - s[KD::pa(k)][cc] * s[KD::pb(k)][cc] * rcn;
            if (KIND == 0 && k == NQ - 1) v[cc] *= sth[cc];
          }
        }
        // the lane keeps column 2 cp + h and sends the other one to lane ^ 8
        keep[i] = h ? v[1] : v[0];
        recv[i] = swap8(h ? v[0] : v[1]);
      }
#pragma unroll
      for (int f = 0; f < NFLD; ++f) s[f][0] = s[f][1] = 0.0;
#pragma unroll
      for (int k = 0; k < NQ; ++k) q[k][0] = q[k][1] = 0.0;
      cnt = 0.0;
      if constexpr (RB > 0) {
        if (nbuf == 0) grp_buf0 = grp;
        double2* o = recbuf + (wave * RB + nbuf) * (NST * 64) + (16 * g + 2 * cp + h);
#pragma unroll
        for (int f = 0; f < NST; ++f) o[f * 64] = h ? make_double2(recv[f], keep[f]) : make_double2(keep[f], recv[f]);
        if (++nbuf == RB) flush_records();
      } else if (dmine < D) {                 // record row f = {northern, southern} sum of field f, lane slot 16 k + column
        double2* o = reinterpret_cast<double2*>(csum + TEMX_CSUM_REC(grp, dt, ndt) * (2 * NST) * 64) + (16 * g + 2 * cp + h);
#pragma unroll
        for (int f = 0; f < NST; ++f)
          TEMX_CSTORE(o + f * 64, h ? make_double2(recv[f], keep[f]) : make_double2(keep[f], recv[f]));
      }
      double ss[NA], dd[NA];
#pragma unroll
      for (int i = 0; i < NA; ++i) {
        ss[i] = keep[i] + recv[i];
        dd[i] = h ? recv[i] - keep[i] : keep[i] - recv[i];
      }
      if constexpr (WS == 1) {
        double* yst = sh[wave];
#pragma unroll
        for (int j = 0; j < YJ; ++j)
          if (lane + 64 * j < YE) yst[lane + 64 * j] = ys[j];
        ++grp;
        load_ys(grp);                         // ycls is padded by one group
#pragma unroll
        for (int t = 0; t < NB; ++t) {
          const double ya = yst[t * 16 + yoff];
#pragma unroll
          for (int f = 0; f < NA; ++f) acc[f][t] = TEMX_MFMA4(ya, t < TBS ? ss[f] : dd[f], acc[f][t]);
        }
      } else {
        OpwSlot<NA, NB>& me = sh[buf][wave];
#pragma unroll
        for (int j = 0; j < YJ; ++j)
          if (lane + 64 * j < YE) me.y[lane + 64 * j] = ys[j];
        ++grp;
        load_ys(grp);
#pragma unroll
        for (int i = 0; i < NA; ++i) {
          me.sd[0][i][lane] = ss[i];
          me.sd[1][i][lane] = dd[i];
        }
        opw_round<TBS, NA>(sh[buf], par, t0, nbw, yoff, lane, acc);
        buf ^= 1;
        ++done;
      }
    }
  };

  if (b0 < b1) {
    load_ys(grp);
    rn = rows_of(b0);
    static_for<PD - 1>([&](auto kc) __attribute__((always_inline)) {
      constexpr int k = decltype(kc)::value;
      const RV r0 = rn;
      rn = rows_of(b0 + k + 1);
      if (k == 0 || b0 + k < b1) issue(kc, r0);
    });
    for (int b = b0; b < b1; b += PD)
      static_for<PD>([&](auto kc) __attribute__((always_inline)) {
        constexpr int k = decltype(kc)::value;
        if (k == 0 || b + k < b1) step(kc, b + k);
      });
  }
  flush_records();
  if constexpr (WS == 4) {
    for (; done < rounds; ++done) {
      OpwSlot<NA, NB>& me = sh[buf][wave];
#pragma unroll
      for (int f = 0; f < NA; ++f) me.sd[0][f][lane] = me.sd[1][f][lane] = 0.0;
#pragma unroll
      for (int j = 0; j < YJ; ++j)
        if (lane + 64 * j < YE) me.y[lane + 64 * j] = 0.0;
      opw_round<TBS, NA>(sh[buf], par, t0, nbw, yoff, lane, acc);
      buf ^= 1;
    }
  }

  // (an empty range still stores its zero slab: the reduction sums every slab)
  if (dmine < D) {
#pragma unroll
    for (int f = 0; f < NA; ++f)
#pragma unroll
      for (int tl = 0; tl < NBA; ++tl)
        if (tl < nbw) {
          const int l = sym_harm<TBS>(t0 + tl, g);
          if (l < K) partial[(((int64_t)split * NA + f) * K + l) * D + dmine] = acc[f][tl];
        }
  }
}

// ------------------------------------------------------------------------------------------------
// parity pair with redundant loads (measured alternative): two waves of a workgroup walk the SAME d-tile
// and the SAME batches -- both issue every load (the second one is served by L1 / L2, HBM traffic is
// unchanged), both form the class sums; wave 0 of the pair accumulates the even harmonics and stores
// the class-sum records, wave 1 the odd harmonics: NA x TBS accumulators each, no LDS hand-over, no barrier.
// Workgroup = 2 d-tiles x 2 parities; cuts as sweep_op_kernel.
// ------------------------------------------------------------------------------------------------
template <typename T, int TBS, int PD, int KIND>
__global__ void __launch_bounds__(256, 2)
sweep_opp_kernel(FieldPtrs<4> fp, int64_t D, int K, const double* __restrict__ ycls,
                 const int4* __restrict__ crow, const int2* __restrict__ csplit,
                 const double* __restrict__ colscale, double* __restrict__ partial, int nsplit, int ndt,
                 double* __restrict__ csum) {
  using KD = OpKind<KIND>;
  constexpr int YE = TBS * 16;                // this parity's blocks only
  constexpr int YJ = (YE + 63) / 64;
  constexpr int MB = CLS_MB;
  constexpr int NFLD = KD::NFLD, NST = KD::NST, NQ = KD::NQ;
  constexpr int NA = NST + NQ;
  __shared__ double ystage[4][YE];
  int split, dq;
  if (!wg_work((ndt + 1) / 2, nsplit, split, dq)) return;
  const int wave = uniform_wave(), lane = threadIdx.x & 63;
  const int c = lane & 15, g = lane >> 4;
  const int par = wave & 1;
  const int dt = dq * 2 + (wave >> 1);
  if (dt >= ndt) return;
  const int64_t d = (int64_t)dt * 16 + c;
  const bool dvalid = d < D;
  const int64_t dcl = dvalid ? d : D - 1;
  const int b0 = __builtin_amdgcn_readfirstlane(csplit[split].x);
  const int b1 = __builtin_amdgcn_readfirstlane(csplit[split + 1].x);
  int grp = __builtin_amdgcn_readfirstlane(csplit[split].y);
  const uint32_t yoff = (uint32_t)(g * 4 + (lane & 3));
  double* yst = ystage[wave];
  const double sth = (KIND == 0 && colscale != nullptr) ? colscale[dcl] : 1.0;
  const T* fb[NFLD];
#pragma unroll
  for (int f = 0; f < NFLD; ++f) fb[f] = reinterpret_cast<const T*>(fp.p[f]) + dcl;
  double acc[NA][TBS];
#pragma unroll
  for (int f = 0; f < NA; ++f)
#pragma unroll
    for (int t = 0; t < TBS; ++t) acc[f][t] = 0.0;
  double s[NFLD], q[NQ], x0[NFLD], cnt = 0.0;
  double sN[NST], qN[NQ];
#pragma unroll
  for (int f = 0; f < NFLD; ++f) s[f] = x0[f] = 0.0;
#pragma unroll
  for (int k = 0; k < NQ; ++k) q[k] = qN[k] = 0.0;
#pragma unroll
  for (int f = 0; f < NST; ++f) sN[f] = 0.0;
  bool north_open = false, prev_south = false;
  const uint32_t D32 = (uint32_t)D;
  T xb[PD][MB][NFLD];
  int er[PD][MB];
  double ys[YJ];
  auto load_ys = [&](int gi) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < YJ; ++j)
      ys[j] = (ycls + ((int64_t)gi * 2 + par) * YE)[(lane + 64 * j) < YE ? (lane + 64 * j) : 0];
  };
  auto issue = [&](auto pc, const int4 rv) __attribute__((always_inline)) {
    constexpr int P = decltype(pc)::value;
    er[P][0] = rv.x; er[P][1] = rv.y; er[P][2] = rv.z; er[P][3] = rv.w;
#pragma unroll
    for (int j = 0; j < MB; ++j) {
      const uint64_t off = (uint64_t)(uint32_t)(er[P][j] & CLS_ROWMASK) * D32;
#pragma unroll
      for (int f = 0; f < NFLD; ++f) xb[P][j][f] = TEMX_XLOAD(fb[f] + off);
    }
  };
  auto finish_side = [&](double* so, double* qo) __attribute__((always_inline)) {
    const double rn = cnt > 0.0 ? 1.0 / cnt : 0.0;
#pragma unroll
    for (int k = 0; k < NQ; ++k) qo[k] = q[k] - s[KD::pa(k)] * s[KD::pb(k)] * rn;
#pragma unroll
    for (int f = 0; f < NST; ++f) so[f] = s[f] + cnt * x0[f];
#pragma unroll
    for (int f = 0; f < NFLD; ++f) s[f] = 0.0;
#pragma unroll
    for (int k = 0; k < NQ; ++k) q[k] = 0.0;
    cnt = 0.0;
  };
  int4 rn;
  auto step = [&](auto pc, int b) __attribute__((always_inline)) {
    constexpr int P = decltype(pc)::value;
    if (b + (PD - 1) < b1) {
      const int4 r1 = rn;
      rn = crow[(int64_t)(b + PD) * 4 + g];
      issue(std::integral_constant<int, (P + PD - 1) % PD>{}, r1);
    }
    const int fl = __builtin_amdgcn_readfirstlane(er[P][0]) >> 27;
    const bool south = (fl & (CLS_SOUTH << 1)) != 0;
    if ((fl & (CLS_FIRST << 1)) || (south && !prev_south)) {
      if (south && north_open) finish_side(sN, qN);
      north_open = !south;
#pragma unroll
      for (int f = 0; f < NFLD; ++f) x0[f] = (double)xb[P][0][f];
    }
    prev_south = south;
    if (fl & 1) {
#pragma unroll
      for (int j = 0; j < MB; ++j) {
        const double w = er[P][j] < 0 ? 0.0 : 1.0;
        double dx[NFLD];
#pragma unroll
        for (int f = 0; f < NFLD; ++f) dx[f] = (double)xb[P][j][f] - x0[f];
#pragma unroll
        for (int f = 0; f < NFLD; ++f) s[f] += w * dx[f];
#pragma unroll
        for (int k = 0; k < NQ; ++k) q[k] += (w * dx[KD::pa(k)]) * dx[KD::pb(k)];
        cnt += w;
      }
    } else {
#pragma unroll
      for (int j = 0; j < MB; ++j) {
        double dx[NFLD];
#pragma unroll
        for (int f = 0; f < NFLD; ++f) dx[f] = (double)xb[P][j][f] - x0[f];
#pragma unroll
        for (int f = 0; f < NFLD; ++f) s[f] += dx[f];
#pragma unroll
        for (int k = 0; k < NQ; ++k) q[k] += dx[KD::pa(k)] * dx[KD::pb(k)];
      }
      cnt += (double)MB;
    }
    if (fl & (CLS_LAST << 1)) {
      prev_south = false;
#pragma unroll
      for (int j = 0; j < YJ; ++j)
        if (lane + 64 * j < YE) yst[lane + 64 * j] = ys[j];
      double sS[NST], qS[NQ];
#pragma unroll
      for (int f = 0; f < NST; ++f) sS[f] = 0.0;
#pragma unroll
      for (int k = 0; k < NQ; ++k) qS[k] = 0.0;
      if (north_open)
        finish_side(sN, qN);
      else
        finish_side(sS, qS);
      north_open = false;
      if (KIND == 0) {
        sN[NST > 2 ? 2 : 0] *= sth; sS[NST > 2 ? 2 : 0] *= sth;
        qN[NQ - 1] *= sth; qS[NQ - 1] *= sth;
      }
      if (dvalid && par == 0) {
        double2* o = reinterpret_cast<double2*>(csum + TEMX_CSUM_REC(grp, dt, ndt) * (2 * NST) * 64) + lane;
#pragma unroll
        for (int f = 0; f < NST; ++f) TEMX_CSTORE(o + f * 64, make_double2(sN[f], sS[f]));
      }
      ++grp;
      load_ys(grp);
      double op[NA];                          // sums for the even harmonics, differences for the odd ones
#pragma unroll
      for (int f = 0; f < NST; ++f) op[f] = par ? sN[f] - sS[f] : sN[f] + sS[f];
#pragma unroll
      for (int k = 0; k < NQ; ++k) op[NST + k] = par ? qN[k] - qS[k] : qN[k] + qS[k];
#pragma unroll
      for (int t = 0; t < TBS; ++t) {
        const double ya = yst[t * 16 + yoff];
#pragma unroll
        for (int f = 0; f < NA; ++f) acc[f][t] = TEMX_MFMA4(ya, op[f], acc[f][t]);
      }
#pragma unroll
      for (int f = 0; f < NST; ++f) sN[f] = 0.0;
#pragma unroll
      for (int k = 0; k < NQ; ++k) qN[k] = 0.0;
    }
  };
  if (b0 < b1) {
    load_ys(grp);
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
  if (dvalid) {
#pragma unroll
    for (int f = 0; f < NA; ++f)
#pragma unroll
      for (int t = 0; t < TBS; ++t) {
        const int l = sym_harm<TBS>(par * TBS + t, g);
        if (l < K) partial[(((int64_t)split * NA + f) * K + l) * D + d] = acc[f][t];
      }
  }
}


}  // namespace temx
