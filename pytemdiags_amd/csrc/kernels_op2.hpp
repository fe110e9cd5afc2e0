// kernels_op2.hpp -- the sweeps of the latitude-class path that the library launches by default, and the kernels
// around the single sweep.  (The algebra and the outputs of the class-sum forms are those of sweep_op_kernel in
// kernels_op.hpp: class-sum records + partial[split][NA][K][D].)
//
//  * sweep_opw_kernel (shared d-tile): the four waves of a workgroup work on ONE d-tile.  Each wave walks its own
//    run of class-groups (its own member rows: nothing is read twice) and forms that group's sums; at the end of
//    every class-group the four waves exchange the NA sums + NA differences per lane and the group's Y blocks
//    through LDS (double buffered, ONE workgroup barrier per round; the barrier does not wait for the global loads
//    in flight), and every wave projects all four groups of the round onto ITS quarter of the harmonic blocks.
//    The library runs its five-field instantiation (KIND = 2: TEM + one tracer in one sweep, ten projections --
//    more accumulators than one wave can hold).
//  * the single sweep: sweep_os_kernel (tile-shaped loads; A/B: TEMX_OS_MAP=tile),
//    os_ref_solve_kernel, os_contract_kernel;
//  * the sweeps that load 1 row x 64 columns per instruction: RowLoad / row_wait (hand-issued
//    loads), sweep_osr_kernel (single sweep, fp64 inputs), sweep_os2_kernel (single sweep with two waves per SIMD,
//    fp32 inputs; row tables per side from side_tables.hpp), sweep_opr_kernel (sweep 1 of the class-sum form, fp64).
//
// Round 3 measured two more forms of sweep 1 that the library never launches (row pairs with 16-byte loads,
// a parity pair with redundant loads): they live in tools/lab_kernels.hpp, for tools/sweep_lab.hip.
#pragma once
#include "kernels_op.hpp"

namespace temx {

// hand-over slot of one wave for one round: the NA sums and NA differences per lane, the group's Y blocks
template <int NA, int NB>
struct OpwSlot {
  double sd[2][NA][64];
  double y[NB * 16];
};

// one round of the shared-d-tile form: all four waves have written their slot; project the four groups
// onto this wave's blocks t0 .. t0 + nbw - 1 (all of one parity: par = 0 sums / even harmonics, 1 differences)
template <int TBS, int NA>
__device__ __forceinline__ void opw_round(const OpwSlot<NA, 2 * TBS>* slots, int par, int t0, int nbw,
                                          uint32_t yoff, int lane, double (&acc)[NA][(TBS + 1) / 2]) {
  // LDS writes of this wave done, then the workgroup barrier; global loads stay in flight across it
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    double b[NA];
#pragma unroll
    for (int f = 0; f < NA; ++f) b[f] = slots[j].sd[par][f][lane];
#pragma unroll
    for (int tl = 0; tl < (TBS + 1) / 2; ++tl)
      if (tl < nbw) {
        const double ya = slots[j].y[(t0 + tl) * 16 + yoff];
#pragma unroll
        for (int f = 0; f < NA; ++f) acc[f][tl] = TEMX_MFMA4(ya, b[f], acc[f][tl]);
      }
  }
}

// ------------------------------------------------------------------------------------------------
// shared d-tile, 8-byte loads (row table crow of kernels_cls.hpp)
// csplit holds 4 * nsplit + 1 group-aligned cuts: workgroup `split` owns pieces 4 split .. 4 split + 3
// ------------------------------------------------------------------------------------------------
// KIND 2 (TEM + one tracer, 10 projections, 5 fields): one wave per SIMD; the tracer's class sums go to csq.
#ifndef TEMX_OPW_WPS
#define TEMX_OPW_WPS 2
#endif
template <typename T, int TBS, int PD, int KIND>
__global__ void __launch_bounds__(256, KIND == 2 ? 1 : TEMX_OPW_WPS)
sweep_opw_kernel(FieldPtrs<OpKind<KIND>::NPTR> fp, int64_t D, int K, const double* __restrict__ ycls,
                 const int4* __restrict__ crow, const int2* __restrict__ csplit,
                 const double* __restrict__ colscale, double* __restrict__ partial, int nsplit, int ndt,
                 double* __restrict__ csum, double* __restrict__ csq) {
  using KD = OpKind<KIND>;
  constexpr int NB = 2 * TBS;
  constexpr int YE = NB * 16;
  constexpr int YJ = (YE + 63) / 64;
  constexpr int MB = CLS_MB;
  constexpr int NFLD = KD::NFLD, NST = KD::NST, NQ = KD::NQ;
  constexpr int NA = NST + NQ;
  constexpr int NBW = (TBS + 1) / 2;          // blocks of the first wave of a parity
  static_assert(MB == 4, "a lane group reads its 4 member rows as one int4");
  static_assert(PD + 1 <= CLS_PADB, "table padding must cover the index prefetch");
  __shared__ OpwSlot<NA, NB> slot[2][4];
  int split, dt;
  if (!wg_work(ndt, nsplit, split, dt)) return;
  const int wave = uniform_wave(), lane = threadIdx.x & 63;
  const int c = lane & 15, g = lane >> 4;
  const int64_t d = (int64_t)dt * 16 + c;
  const bool dvalid = d < D;
  const int64_t dcl = dvalid ? d : D - 1;
  const int piece = split * 4 + wave;
  const int b0 = __builtin_amdgcn_readfirstlane(csplit[piece].x);
  const int b1 = __builtin_amdgcn_readfirstlane(csplit[piece + 1].x);
  int grp = __builtin_amdgcn_readfirstlane(csplit[piece].y);
  int rounds = 0;                             // the most class-groups any of the four waves walks
#pragma unroll
  for (int w = 0; w < 4; ++w) {
    const int n = __builtin_amdgcn_readfirstlane(csplit[split * 4 + w + 1].y - csplit[split * 4 + w].y);
    rounds = n > rounds ? n : rounds;
  }
  const int par = wave >> 1;
  const int t0 = par * TBS + (wave & 1) * NBW, nbw = (wave & 1) ? TBS - NBW : NBW;
  const uint32_t yoff = (uint32_t)(g * 4 + (lane & 3));
  const double sth = (KD::TF >= 0 && colscale != nullptr) ? colscale[dcl] : 1.0;
  const T* fb[NFLD];
#pragma unroll
  for (int f = 0; f < NFLD; ++f) fb[f] = reinterpret_cast<const T*>(fp.p[f]) + dcl;

  double acc[NA][NBW];
#pragma unroll
  for (int f = 0; f < NA; ++f)
#pragma unroll
    for (int t = 0; t < NBW; ++t) acc[f][t] = 0.0;
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
  int buf = 0, done = 0;

  T xb[PD][MB][NFLD];
  int er[PD][MB];
  double ys[YJ];
  auto load_ys = [&](int gi) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < YJ; ++j) ys[j] = (ycls + (int64_t)gi * YE)[(lane + 64 * j) < YE ? (lane + 64 * j) : 0];
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
      OpwSlot<NA, NB>& me = slot[buf][wave];
#pragma unroll
      for (int j = 0; j < YJ; ++j)
        if (lane + 64 * j < YE) me.y[lane + 64 * j] = ys[j];
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
      if constexpr (KD::TF >= 0) {            // T -> theta: its sum and the v theta co-moment
        sN[KD::TF] *= sth; sS[KD::TF] *= sth;
        qN[KD::TQ] *= sth; qS[KD::TQ] *= sth;
      }
      if (dvalid) {
        double2* o = reinterpret_cast<double2*>(csum + TEMX_CSUM_REC(grp, dt, ndt) * (2 * KD::NSTA) * 64) + lane;
#pragma unroll
        for (int f = 0; f < KD::NSTA; ++f) TEMX_CSTORE(o + f * 64, make_double2(sN[f], sS[f]));
        if constexpr (NST > KD::NSTA) {       // the tracer's own records
          double2* oq = reinterpret_cast<double2*>(csq + TEMX_CSUM_REC(grp, dt, ndt) * (2 * (NST - KD::NSTA)) * 64) + lane;
#pragma unroll
          for (int f = KD::NSTA; f < NST; ++f) TEMX_CSTORE(oq + (f - KD::NSTA) * 64, make_double2(sN[f], sS[f]));
        }
      }
      ++grp;
      load_ys(grp);
#pragma unroll
      for (int f = 0; f < NST; ++f) {
        me.sd[0][f][lane] = sN[f] + sS[f];
        me.sd[1][f][lane] = sN[f] - sS[f];
        sN[f] = 0.0;
      }
#pragma unroll
      for (int k = 0; k < NQ; ++k) {
        me.sd[0][NST + k][lane] = qN[k] + qS[k];
        me.sd[1][NST + k][lane] = qN[k] - qS[k];
        qN[k] = 0.0;
      }
      opw_round<TBS, NA>(slot[buf], par, t0, nbw, yoff, lane, acc);
      buf ^= 1;
      ++done;
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
  // a wave with fewer class-groups than its neighbours hands over zeros for the remaining rounds
  for (; done < rounds; ++done) {
    OpwSlot<NA, NB>& me = slot[buf][wave];
#pragma unroll
    for (int f = 0; f < NA; ++f) me.sd[0][f][lane] = me.sd[1][f][lane] = 0.0;
#pragma unroll
    for (int j = 0; j < YJ; ++j)
      if (lane + 64 * j < YE) me.y[lane + 64 * j] = 0.0;
    opw_round<TBS, NA>(slot[buf], par, t0, nbw, yoff, lane, acc);
    buf ^= 1;
  }

  if (dvalid) {
#pragma unroll
    for (int f = 0; f < NA; ++f)
#pragma unroll
      for (int tl = 0; tl < NBW; ++tl)
        if (tl < nbw) {
          const int l = sym_harm<TBS>(t0 + tl, g);
          if (l < K) partial[(((int64_t)split * NA + f) * K + l) * D + d] = acc[f][tl];
        }
  }
}

// ------------------------------------------------------------------------------------------------
// sweep_os_kernel -- ONE sweep, no class-sum stream: the eddy-product sums follow afterwards from
//   F_l = P_l - sum_m beta_m M^a_lm - sum_m alpha_m M^b_lm + alpha^T T_l beta,   M^a_lm = sum_k g(l,m,k) A_k
// (Legendre product linearisation), which needs the projections A_k of the four fields up to degree 2L and
// the projections P_l of the three products up to degree L -- all of fields from which a band-limited
// REFERENCE r = sum_m rho_m Y_m (coefficients rho from a subsample pre-pass) has been subtracted, so that
// no term is a difference of large numbers.  Per class side (sums about the first member x0 as in
// sweep_op_kernel, n members, S~ = sum (x - x0), q~ = sum (a - a0)(b - b0)):
//     S'  = S~ + n (x0 - r)                      sum of the shifted field
//     P'  = q~ - S~_a S~_b / n + n (m_a - r_a)(m_b - r_b),   m - r = S~ / n + x0 - r
// with r reconstructed at the class latitudes per class-group (4 x 2 NBR MFMAs, operands rho in LDS).
// Same row table and work cuts as sweep_op_kernel; one wave = one d-tile, one wave per SIMD.
//   ycx[group][2 TBX][16]   basis blocks up to degree 2L (TBX even blocks, then TBX odd ones)
//   rho[4][4 * 2 NBR ...]   reference coefficients, rows harm(tb, g) as C in flux_cls_kernel: rho[f][K4][D]
//   px[split][4][KX][D]     field projections (KX = 2L + 1),   pp[split][3][K][D]  product projections
// ------------------------------------------------------------------------------------------------
template <int TBX>
__device__ __forceinline__ constexpr int symx_harm(int tb, int i) {
  return tb < TBX ? 2 * (4 * tb + i) : 2 * (4 * (tb - TBX) + i) + 1;
}

// KIND 0: TEM     fields (u, v, T -> theta, omega), all four projected to degree 2L; products u v, u omega, v theta
// KIND 1: tracer  fields (q, v, omega): only q is projected to degree 2L (the projections of v and omega are
//                 those of the TEM run, whose references for v and omega must be handed over again); products q v, q omega
template <int KIND> struct OsKind;
//                 those of the TEM run, whose references for v and omega must be handed over again); products q v, q omega
// KIND 3: two tracers  fields (q1, q2, v, omega): q1 and q2 projected to degree 2L; products q1 v, q1 omega, q2 v, q2 omega.
//                 One read of v and omega serves two tracers (tem_diagnostics.py:281-301 takes a LIST of tracers):
//                 2 x 26 + 4 x 14 = 108 accumulators; NPR of the products keep theirs in registers (the fields need
//                 half the registers of KIND 0), the others in LDS as before.
template <> struct OsKind<0> {
  static constexpr int NF = 4, NFX = 4, NP = 3, TF = 2, TP = 2, NPR = 0;
  __host__ __device__ static constexpr int pa(int k) { return k == 2 ? 1 : 0; }
  __host__ __device__ static constexpr int pb(int k) { return k == 0 ? 1 : (k == 1 ? 3 : 2); }
};
template <> struct OsKind<1> {
  static constexpr int NF = 3, NFX = 1, NP = 2, TF = -1, TP = -1, NPR = 0;
  __host__ __device__ static constexpr int pa(int) { return 0; }
  __host__ __device__ static constexpr int pb(int k) { return k == 0 ? 1 : 2; }
};
template <> struct OsKind<3> {
  static constexpr int NF = 4, NFX = 2, NP = 4, TF = -1, TP = -1, NPR = 2;
  __host__ __device__ static constexpr int pa(int k) { return k >> 1; }
  __host__ __device__ static constexpr int pb(int k) { return 2 + (k & 1); }
};

#ifndef TEMX_OS_SKIP
#define TEMX_OS_SKIP 0      // lab builds leave parts of the work out (tools/sweep_lab.hip)
#endif
// DEFER: the projection MFMAs of a finished class-group (4 x 26 + 3 x 14 of them, with the read-modify-write of
// the LDS accumulators) are not issued in one block at the group's end -- 2600 cycles in which the wave,
// alone on its SIMD, requests nothing from memory -- but in NCH chunks, one per following batch, each right
// after that batch's loads have been issued.  The finished group's operands wait in registers, its Y blocks
// in a second LDS buffer.
template <typename T, int TBS, int TBX, int NBR, int PD, int KIND = 0, int DEFER = 0>
__global__ void __launch_bounds__(256, 1)
sweep_os_kernel(FieldPtrs<4> fp, int64_t D, int K, int KX, const double* __restrict__ ycx,
                const int4* __restrict__ crow, const int2* __restrict__ csplit,
                const double* __restrict__ colscale, const double* __restrict__ rho, int K4,
                double* __restrict__ px, double* __restrict__ pp, int nsplit, int ndt) {
  using KD = OsKind<KIND>;
  constexpr int NF = KD::NF, NFX = KD::NFX, NP = KD::NP;
  constexpr int NBX = 2 * TBX;                // blocks of the extended basis
  constexpr int YE = NBX * 16;
  constexpr int YJ = (YE + 63) / 64;
  constexpr int MB = CLS_MB;
  static_assert(NBR <= TBS && TBS <= TBX, "reference degree <= L <= 2L");
  // the 3 x 2 TBS product accumulators live in wave-private LDS (146 accumulators do not fit the register file
  // next to the load ring): read-modify-write once per class-group
  extern __shared__ double lds[];             // [4 waves][YE] Y blocks, [4 waves][4][2 NBR][64] reference operands, [4 waves][3][2 TBS][64] product accumulators
  int split, dq;
  if (!wg_work((ndt + 3) / 4, nsplit, split, dq)) return;
  const int wave = uniform_wave(), lane = threadIdx.x & 63;
  const int c = lane & 15, g = lane >> 4;
  const int dt = dq * 4 + wave;
  if (dt >= ndt) return;
  const int64_t d = (int64_t)dt * 16 + c;
  const bool dvalid = d < D;
  const int64_t dcl = dvalid ? d : D - 1;
  const int b0 = __builtin_amdgcn_readfirstlane(csplit[split].x);
  const int b1 = __builtin_amdgcn_readfirstlane(csplit[split + 1].x);
  int grp = __builtin_amdgcn_readfirstlane(csplit[split].y);
  const uint32_t aoff_p = (uint32_t)(g * 4 + (lane & 3));
  const uint32_t aoff_r = (uint32_t)((lane & 3) * 4 + g);
  constexpr int NYB = DEFER != 0 ? 2 : 1;          // Y buffers per wave
  constexpr int NCH = 4;                      // chunks of a deferred projection
  double* ybase = lds + wave * (NYB * YE);
  double* yst = ybase;
  double* cb = lds + 4 * NYB * YE + wave * (NF * 2 * NBR * 64) + lane;
  double* apl = lds + 4 * NYB * YE + 4 * (NF * 2 * NBR * 64) + wave * (NP * 2 * TBS * 64) + lane;
  const double sth = (KD::TF >= 0 && colscale != nullptr) ? colscale[dcl] : 1.0;
#pragma unroll
  for (int i = 0; i < NP * 2 * TBS; ++i) apl[i * 64] = 0.0;
#pragma unroll
  for (int f = 0; f < NF; ++f)
#pragma unroll
    for (int tb = 0; tb < 2 * NBR; ++tb) {   // (reference blocks: the first NBR even and the first NBR odd blocks)
      const int l = tb < NBR ? 2 * (4 * tb + g) : 2 * (4 * (tb - NBR) + g) + 1;
      const double v = rho[((int64_t)f * K4 + (l < K ? l : K - 1)) * D + dcl];
      cb[(f * 2 * NBR + tb) * 64] = l < K ? v : 0.0;
    }
  const T* fb[NF];
#pragma unroll
  for (int f = 0; f < NF; ++f) fb[f] = reinterpret_cast<const T*>(fp.p[f]) + dcl;

  double ax[NFX][NBX];
#pragma unroll
  for (int f = 0; f < NFX; ++f)
#pragma unroll
    for (int t = 0; t < NBX; ++t) ax[f][t] = 0.0;
  double s[NF], q[NP], x0[NF], cnt = 0.0;
  double sN[NF], qN[NP], x0N[NF], cntN = 0.0;   // the finished northern side, still about its own origin
#pragma unroll
  for (int f = 0; f < NF; ++f) s[f] = x0[f] = sN[f] = x0N[f] = 0.0;
#pragma unroll
  for (int k = 0; k < NP; ++k) q[k] = qN[k] = 0.0;
  bool north_open = false, prev_south = false;
  const uint32_t D32 = (uint32_t)D;

  T xb[PD][MB][NF];
  int er[PD][MB];
  double ys[YJ];
  auto load_ys = [&](int gi) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < YJ; ++j) ys[j] = (ycx + (int64_t)gi * YE)[(lane + 64 * j) < YE ? (lane + 64 * j) : 0];
  };
  auto issue = [&](auto pc, const int4 rv) __attribute__((always_inline)) {
    constexpr int P = decltype(pc)::value;
    er[P][0] = rv.x; er[P][1] = rv.y; er[P][2] = rv.z; er[P][3] = rv.w;
#pragma unroll
    for (int j = 0; j < MB; ++j) {
      const uint64_t off = (uint64_t)(uint32_t)(er[P][j] & CLS_ROWMASK) * D32;
#pragma unroll
      for (int f = 0; f < NF; ++f) xb[P][j][f] = TEMX_XLOAD(fb[f] + off);
    }
  };
  // operands of the group whose projection is (partly) pending, the Y buffer they go with, chunks done so far
  double dS[NFX][2], dP[NP][2];
  const double* yprev = ybase;
  int left = 0;                               // chunks of it still to run
#pragma unroll
  for (int f = 0; f < NFX; ++f) dS[f][0] = dS[f][1] = 0.0;
#pragma unroll
  for (int k = 0; k < NP; ++k) dP[k][0] = dP[k][1] = 0.0;
  // DEFER 1: fields and products deferred; 2: the fields only (8 operands fewer to hold)
  auto project_blocks = [&](auto t0c, auto t1c, auto whatc, const double* yb, const double (&S)[NFX][2], const double (&P)[NP][2])
      __attribute__((always_inline)) {
    constexpr int T0 = decltype(t0c)::value, T1 = decltype(t1c)::value, WHAT = decltype(whatc)::value;
#pragma unroll
    for (int t = T0; t < T1; ++t) {
      const double ya = yb[t * 16 + aoff_p];
      if constexpr (WHAT & 1) {
#pragma unroll
        for (int f = 0; f < NFX; ++f) ax[f][t] = TEMX_MFMA4(ya, S[f][t < TBX ? 0 : 1], ax[f][t]);
      }
      const int tp = t < TBX ? t : t - TBX;                     // the product blocks are the first TBS of each parity
      if ((WHAT & 2) && tp < TBS) {
        const int ta = t < TBX ? tp : TBS + tp;
        double v[NP];
#pragma unroll
        for (int k = 0; k < NP; ++k) v[k] = apl[(k * 2 * TBS + ta) * 64];
#pragma unroll
        for (int k = 0; k < NP; ++k) v[k] = TEMX_MFMA4(ya, P[k][t < TBX ? 0 : 1], v[k]);
#pragma unroll
        for (int k = 0; k < NP; ++k) apl[(k * 2 * TBS + ta) * 64] = v[k];
      }
    }
  };
  // chunk c of the pending projection: blocks [c NBX/NCH, (c+1) NBX/NCH).  The chunk a step runs is its position
  // in the 4-step unrolled loop (static, so ax[][] stays statically indexed); any NCH consecutive steps cover all.
  auto pending_chunk = [&](auto cc) __attribute__((always_inline)) {
    constexpr int C = decltype(cc)::value;
    project_blocks(std::integral_constant<int, C * NBX / NCH>{}, std::integral_constant<int, (C + 1) * NBX / NCH>{},
                   std::integral_constant<int, DEFER == 2 ? 1 : 3>{}, yprev, dS, dP);
  };
  auto park_north = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int f = 0; f < NF; ++f) { sN[f] = s[f]; x0N[f] = x0[f]; s[f] = 0.0; }
#pragma unroll
    for (int k = 0; k < NP; ++k) { qN[k] = q[k]; q[k] = 0.0; }
    cntN = cnt;
    cnt = 0.0;
  };
  int4 rn;
  auto step = [&](auto posc, int b) __attribute__((always_inline)) {
    constexpr int POS = decltype(posc)::value % NCH;   // position in the unrolled loop
    constexpr int P = decltype(posc)::value % PD;      // slot of the load ring
    if (b + (PD - 1) < b1) {
      const int4 r1 = rn;
      rn = crow[(int64_t)(b + PD) * 4 + g];
      issue(std::integral_constant<int, (P + PD - 1) % PD>{}, r1);
    }
    if constexpr (DEFER != 0) {
      if (left > 0) {                         // the loads of the next batch are in flight meanwhile
        if (!(TEMX_OS_SKIP & 1)) pending_chunk(std::integral_constant<int, POS>{});
        --left;
      }
    }
    const int fl = __builtin_amdgcn_readfirstlane(er[P][0]) >> 27;
    const bool south = (fl & (CLS_SOUTH << 1)) != 0;
    if ((fl & (CLS_FIRST << 1)) || (south && !prev_south)) {
      if (south && north_open) park_north();
      north_open = !south;
#pragma unroll
      for (int f = 0; f < NF; ++f) x0[f] = (double)xb[P][0][f];
    }
    prev_south = south;
    if (TEMX_OS_SKIP & 4) {                   // (lab: no accumulation; the loads stay live)
#pragma unroll
      for (int j = 0; j < MB; ++j)
#pragma unroll
        for (int f = 0; f < NF; ++f) asm volatile("" ::"v"(xb[P][j][f]));
    } else if (fl & 1) {
#pragma unroll
      for (int j = 0; j < MB; ++j) {
        const double w = er[P][j] < 0 ? 0.0 : 1.0;
        double dx[NF];
#pragma unroll
        for (int f = 0; f < NF; ++f) dx[f] = (double)xb[P][j][f] - x0[f];
#pragma unroll
        for (int f = 0; f < NF; ++f) s[f] += w * dx[f];
#pragma unroll
        for (int k = 0; k < NP; ++k) q[k] += (w * dx[KD::pa(k)]) * dx[KD::pb(k)];
        cnt += w;
      }
    } else {
#pragma unroll
      for (int j = 0; j < MB; ++j) {
        double dx[NF];
#pragma unroll
        for (int f = 0; f < NF; ++f) dx[f] = (double)xb[P][j][f] - x0[f];
#pragma unroll
        for (int f = 0; f < NF; ++f) s[f] += dx[f];
#pragma unroll
        for (int k = 0; k < NP; ++k) q[k] += dx[KD::pa(k)] * dx[KD::pb(k)];
      }
      cnt += (double)MB;
    }
    if (fl & (CLS_LAST << 1)) {
      prev_south = false;
      if constexpr (DEFER != 0) {
        if (left > 0)                         // (a group shorter than NCH steps: what is left of the previous projection)
          static_for<NCH>([&](auto cc) __attribute__((always_inline)) {
            if (((decltype(cc)::value - POS - 1) & (NCH - 1)) < left) pending_chunk(cc);
          });
        yst = yst == ybase ? ybase + YE : ybase;
      }
#pragma unroll
      for (int j = 0; j < YJ; ++j)
        if (lane + 64 * j < YE) yst[lane + 64 * j] = ys[j];
      if (north_open) park_north();             // the group has no southern batch: the open side is the northern one
      north_open = false;
      ++grp;
      load_ys(grp);
      // ---- reference at the class latitudes: E = even part, O = odd part; r_N = E + O, r_S = E - O
      double E[NF], O[NF];
#pragma unroll
      for (int f = 0; f < NF; ++f) E[f] = O[f] = 0.0;
#pragma unroll
      for (int tb = 0; tb < ((TEMX_OS_SKIP & 2) ? 0 : 2 * NBR); ++tb) {
        const int blk = tb < NBR ? tb : TBX + (tb - NBR);
        const double ya = yst[blk * 16 + aoff_r];
#pragma unroll
        for (int f = 0; f < NF; ++f) {
          if (tb < NBR)
            E[f] = TEMX_MFMA4(ya, cb[(f * 2 * NBR + tb) * 64], E[f]);
          else
            O[f] = TEMX_MFMA4(ya, cb[(f * 2 * NBR + tb) * 64], O[f]);
        }
      }
      // ---- sums of the shifted fields and of their products, per side (theta = T x the column scale)
      double SN[NF], SS[NF], PN[NP], PS[NP], mN[NF], mS[NF];
      const double rnN = cntN > 0.0 ? 1.0 / cntN : 0.0, rnS = cnt > 0.0 ? 1.0 / cnt : 0.0;
#pragma unroll
      for (int f = 0; f < NF; ++f) {
        const double sc = f == KD::TF ? sth : 1.0;
        mN[f] = (sN[f] * rnN + x0N[f]) * sc - (E[f] + O[f]);     // side mean minus the reference
        mS[f] = (s[f] * rnS + x0[f]) * sc - (E[f] - O[f]);
        SN[f] = cntN * mN[f];
        SS[f] = cnt * mS[f];
      }
#pragma unroll
      for (int k = 0; k < NP; ++k) {
        const double sc = k == KD::TP ? sth : 1.0;
        PN[k] = (qN[k] - sN[KD::pa(k)] * sN[KD::pb(k)] * rnN) * sc + cntN * mN[KD::pa(k)] * mN[KD::pb(k)];
        PS[k] = (q[k] - s[KD::pa(k)] * s[KD::pb(k)] * rnS) * sc + cnt * mS[KD::pa(k)] * mS[KD::pb(k)];
      }
#pragma unroll
      for (int f = 0; f < NFX; ++f) {
        dS[f][0] = SN[f] + SS[f];
        dS[f][1] = SN[f] - SS[f];
      }
#pragma unroll
      for (int k = 0; k < NP; ++k) {
        dP[k][0] = PN[k] + PS[k];
        dP[k][1] = PN[k] - PS[k];
      }
      if constexpr (DEFER != 0) {
        yprev = yst;
        left = NCH;
        if constexpr (DEFER == 2)
          project_blocks(std::integral_constant<int, 0>{}, std::integral_constant<int, NBX>{}, std::integral_constant<int, 2>{}, yst, dS, dP);
      } else {
        project_blocks(std::integral_constant<int, 0>{}, std::integral_constant<int, NBX>{}, std::integral_constant<int, 3>{}, yst, dS, dP);
      }
#pragma unroll
      for (int f = 0; f < NF; ++f) s[f] = sN[f] = 0.0;
#pragma unroll
      for (int k = 0; k < NP; ++k) q[k] = qN[k] = 0.0;
      cnt = cntN = 0.0;
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
    constexpr int UNR = DEFER != 0 ? (PD % 4 == 0 ? PD : PD % 2 == 0 ? 2 * PD : 4 * PD) : PD;   // lcm(PD, NCH)
    static_assert(UNR % PD == 0 && (DEFER == 0 || UNR % NCH == 0), "unrolled loop covers whole rings and whole chunk rounds");
    for (int b = b0; b < b1; b += UNR)
      static_for<UNR>([&](auto kc) __attribute__((always_inline)) {
        constexpr int k = decltype(kc)::value;
        if (k == 0 || b + k < b1) step(kc, b + k);
      });
  }
  if constexpr (DEFER != 0) {                 // the last group's projection (its chunks in any order)
    if (left > 0) {
      const int first = (b1 - b0) & (NCH - 1);          // position the next step would have had
      static_for<NCH>([&](auto cc) __attribute__((always_inline)) {
        if (((decltype(cc)::value - first) & (NCH - 1)) < left) pending_chunk(cc);
      });
    }
  }
  if (dvalid) {
#pragma unroll
    for (int f = 0; f < NFX; ++f)
#pragma unroll
      for (int t = 0; t < NBX; ++t) {
        const int l = symx_harm<TBX>(t, g);
        if (l < (pp != nullptr ? KX : K4)) px[(((int64_t)split * NFX + f) * KX + l) * D + d] = ax[f][t];   // (pre-pass, pp == NULL: only the K4 rows of the reference fit)
      }
#pragma unroll
    for (int k = 0; k < NP; ++k)
#pragma unroll
      for (int t = 0; t < 2 * TBS; ++t) {
        const int l = sym_harm<TBS>(t, g);
        if (pp != nullptr && l < K) pp[(((int64_t)split * NP + k) * K + l) * D + d] = apl[(k * 2 * TBS + t) * 64];
      }
  }
}

// Loads of the row-map sweeps: wave-uniform row address (SGPR pair) + one 32-bit lane offset, issued by hand.
// Written as ordinary loads the compiler (ROCm 7.2) either folds the lane offset into per-lane pointers -- one
// VALU add per load whose result registers it takes from the batch still in flight, so the issue waits for that
// batch -- or mixes both forms under SGPR pressure.  Issued by hand the loads are invisible to its wait-count
// pass: row_wait<N>() is the s_waitcnt that makes the named values usable (N = loads issued after them).
#ifndef TEMX_ROWLOAD_MOD
#define TEMX_ROWLOAD_MOD "nt"     // cache policy of the row loads (lab A/B: "", "sc1", "sc0 sc1" -- all 8.85-9.02 ms, no difference;
                                  // round 4, rows that are not whole 128-byte lines: "" 2.384 against 2.391 ms on ne30 x 72 x 91 -- the
                                  // lines two neighbouring windows share are fetched twice under either policy)
#endif
template <typename T> struct RowLoad;
template <> struct RowLoad<double> {
  static __device__ __forceinline__ void ld(double& dst, uint32_t voff, uint64_t sbase) {
    asm volatile("global_load_dwordx2 %0, %1, %2 " TEMX_ROWLOAD_MOD : "=v"(dst) : "v"(voff), "s"(sbase) : "memory");
  }
};
template <> struct RowLoad<float> {
  static __device__ __forceinline__ void ld(float& dst, uint32_t voff, uint64_t sbase) {
    asm volatile("global_load_dword %0, %1, %2 " TEMX_ROWLOAD_MOD : "=v"(dst) : "v"(voff), "s"(sbase) : "memory");
  }
};
// 1 / n for a member count n (a small positive integer): v_rcp_f64 and one Newton step instead of the division's
// twenty instructions, twice per class-group and wave
__device__ __forceinline__ double temx_rcp_count(double n) {
  double r = __builtin_amdgcn_rcp(n);
  return __builtin_fma(__builtin_fma(-n, r, 1.0), r, r);
}
template <int N, typename T>
__device__ __forceinline__ void row_wait(T& a, T& b, T& c, T& d) {
  asm volatile("s_waitcnt vmcnt(%4)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "n"(N) : "memory");
}
template <int N, typename T>
__device__ __forceinline__ void row_wait(T& a, T& b, T& c) {
  asm volatile("s_waitcnt vmcnt(%3)" : "+v"(a), "+v"(b), "+v"(c) : "n"(N) : "memory");
}
template <typename T>
__device__ __forceinline__ void row_touch(T& a) {      // orders the uses of a behind the waits executed so far
  asm volatile("" : "+v"(a)::"memory");
}

// ------------------------------------------------------------------------------------------------
// The single sweep with row-contiguous loads.  tools/ubench_lanemap.hip: the same bytes, rows and ring depth read
// at 5.5 TB/s when one load instruction takes 4 rows x 16 columns (the MFMA B tile, as every other sweep here
// loads) and at 6.3 TB/s when it takes 1 row x 64 columns (fp32: 3.1 vs 5.1 TB/s; 512-byte rows: 3.2 vs 6.1).
// The class sums are lane-wise sums over the rows of a class, so the lane map of the loads is free:
//   * while it reads, wave w of the workgroup owns class slot w of every class-group and all 64 columns of the
//     workgroup (lane = column): every load is one row x 64 columns with a wave-uniform row address;
//   * at the end of a class-group the side means and central co-moments of (class, column) cross through LDS
//     (two workgroup barriers that leave the global loads in flight), and wave w becomes the owner of d-tile w
//     (lane = class slot x column, the MFMA B layout) for the reference, the shifted sums and the projection,
//     which is deferred over the next four batches as in sweep_os_kernel<DEFER = 1>.
// The Y blocks of a group are the same for the four waves: one shared copy, double buffered.
// ------------------------------------------------------------------------------------------------
template <typename T, int TBS, int TBX, int NBR, int PD, int KIND = 0>
__global__ void __launch_bounds__(256, 1)
sweep_osr_kernel(FieldPtrs<4> fp, int64_t D, int K, int KX, const double* __restrict__ ycx,
                 const int4* __restrict__ crow, const int2* __restrict__ csplit,
                 const double* __restrict__ colscale, const double* __restrict__ rho, int K4,
                 double* __restrict__ px, double* __restrict__ pp, int nsplit, int ndt) {
  using KD = OsKind<KIND>;
  constexpr int NF = KD::NF, NFX = KD::NFX, NP = KD::NP;
  constexpr int NBX = 2 * TBX;
  constexpr int YE = NBX * 16;
  constexpr int YJ = (YE + 255) / 256;        // Y elements per thread of the workgroup
  constexpr int MB = CLS_MB;
  constexpr int NCH = 4;
  constexpr int NV = 2 * (NF + NP);           // exchanged per (class, column): side means, central co-moments
  constexpr int NPR = KD::NPR, NPL = NP - NPR; // product accumulators in registers / in LDS
  static_assert(NBR <= TBS && TBS <= TBX, "reference degree <= L <= 2L");
  static_assert(YJ <= 2, "Y quarter per thread");
  static_assert((PD - 1) * CLS_MB * KD::NF + (CLS_MB - 1) * KD::NF + 2 <= 63, "the ring is counted in vmcnt (6 bits)");
  // [2][YE] Y blocks | [16] member counts | [4 waves][NF][2 NBR][64] reference operands |
  // [4 waves][NP - NPR][2 TBS][64] product accumulators | [NV][4 classes][64 columns] exchange
  extern __shared__ double lds[];
  int split, dq;
  if (!wg_work((ndt + 3) / 4, nsplit, split, dq)) return;
  const int tid = threadIdx.x;
  const int wave = uniform_wave(), lane = tid & 63;
  const int c = lane & 15, g = lane >> 4;
  // reading role: class slot `wave`, column dq * 64 + lane
  const int64_t colr = (int64_t)dq * 64 + lane < D ? (int64_t)dq * 64 + lane : D - 1;
  const uint32_t colb32 = (uint32_t)colr * (uint32_t)sizeof(T);   // byte offset of my column in a row
  // tile role: d-tile dq * 4 + wave, lane = (class slot g, column c)
  const int dt = dq * 4 + wave;
  const int64_t d = (int64_t)dt * 16 + c;
  const bool dvalid = dt < ndt && d < D;
  const int64_t dcl = dvalid ? d : D - 1;
  const int b0 = __builtin_amdgcn_readfirstlane(csplit[split].x);
  const int b1 = __builtin_amdgcn_readfirstlane(csplit[split + 1].x);
  int grp = __builtin_amdgcn_readfirstlane(csplit[split].y);
  const uint32_t aoff_p = (uint32_t)(g * 4 + (lane & 3));
  const uint32_t aoff_r = (uint32_t)((lane & 3) * 4 + g);
  double* ybase = lds;
  double* cn = lds + 2 * YE;
  double* cb = lds + 2 * YE + 16 + wave * (NF * 2 * NBR * 64) + lane;
  double* apl = lds + 2 * YE + 16 + 4 * (NF * 2 * NBR * 64) + wave * (NPL * 2 * TBS * 64) + lane;
  double* ex = lds + 2 * YE + 16 + 4 * (NF * 2 * NBR * 64) + 4 * (NPL * 2 * TBS * 64);
  double* exw = ex + wave * 64 + lane;                   // [v][my class][my column]
  const double* exr = ex + g * 64 + wave * 16 + c;       // [v][class g][column of my d-tile]
  const double sth = (KD::TF >= 0 && colscale != nullptr) ? colscale[colr] : 1.0;
#pragma unroll
  for (int i = 0; i < NPL * 2 * TBS; ++i) apl[i * 64] = 0.0;
  double apr[NPR > 0 ? NPR : 1][2 * TBS];
#pragma unroll
  for (int k = 0; k < (NPR > 0 ? NPR : 1); ++k)
#pragma unroll
    for (int t = 0; t < 2 * TBS; ++t) apr[k][t] = 0.0;
#pragma unroll
  for (int f = 0; f < NF; ++f)
#pragma unroll
    for (int tb = 0; tb < 2 * NBR; ++tb) {
      const int l = tb < NBR ? 2 * (4 * tb + g) : 2 * (4 * (tb - NBR) + g) + 1;
      const double v = rho[((int64_t)f * K4 + (l < K ? l : K - 1)) * D + dcl];
      cb[(f * 2 * NBR + tb) * 64] = l < K ? v : 0.0;
    }
  uint64_t fbase[NF];
#pragma unroll
  for (int f = 0; f < NF; ++f) fbase[f] = reinterpret_cast<uint64_t>(fp.p[f]);
  { double sth_ready = sth; asm volatile("" : "+v"(sth_ready)); }     // (see sweep_os2_kernel: no tracked load may stay pending into the loop)

  double ax[NFX][NBX];
#pragma unroll
  for (int f = 0; f < NFX; ++f)
#pragma unroll
    for (int t = 0; t < NBX; ++t) ax[f][t] = 0.0;
  double s[NF], q[NP], x0[NF], cnt = 0.0;
  double sN[NF], qN[NP], x0N[NF], cntN = 0.0;
#pragma unroll
  for (int f = 0; f < NF; ++f) s[f] = x0[f] = sN[f] = x0N[f] = 0.0;
#pragma unroll
  for (int k = 0; k < NP; ++k) q[k] = qN[k] = 0.0;
  bool north_open = false, prev_south = false;
  const uint32_t rowbytes = (uint32_t)D * (uint32_t)sizeof(T);   // host guarantees D < 2^28

  T xb[PD][MB][NF];
  int er[PD][MB];                             // wave-uniform: the rows of this wave's class slot
  double ys[YJ];
  const uint32_t yoff32[2] = {(uint32_t)(tid < YE ? tid : 0) * 8u, (uint32_t)(tid + 256 < YE ? tid + 256 : 0) * 8u};
  auto load_ys = [&](int gi) __attribute__((always_inline)) {   // (by hand as well: a compiler-tracked load would be
#pragma unroll                                                  // waited for with vmcnt(0), draining the ring once per group)
    for (int j = 0; j < YJ; ++j) RowLoad<double>::ld(ys[j], yoff32[j], reinterpret_cast<uint64_t>(ycx) + (uint64_t)gi * (YE * 8));
  };
  auto issue = [&](auto pc, const int4 rv) __attribute__((always_inline)) {
    constexpr int P = decltype(pc)::value;
    er[P][0] = rv.x; er[P][1] = rv.y; er[P][2] = rv.z; er[P][3] = rv.w;
#pragma unroll
    for (int j = 0; j < MB; ++j) {
      const uint64_t off = (uint64_t)(uint32_t)(er[P][j] & CLS_ROWMASK) * rowbytes;   // wave-uniform, one 32 x 32 -> 64 multiply
#pragma unroll
      for (int f = 0; f < NF; ++f) RowLoad<T>::ld(xb[P][j][f], colb32, fbase[f] + off);
    }
  };
  double dS[NFX][2], dP[NP][2];
  const double* yprev = ybase;
  int ycur = 0;
  int left = 0;
#pragma unroll
  for (int f = 0; f < NFX; ++f) dS[f][0] = dS[f][1] = 0.0;
#pragma unroll
  for (int k = 0; k < NP; ++k) dP[k][0] = dP[k][1] = 0.0;
  auto pending_chunk = [&](auto cc) __attribute__((always_inline)) {
    constexpr int C = decltype(cc)::value;
#pragma unroll
    for (int t = C * NBX / NCH; t < (C + 1) * NBX / NCH; ++t) {
      const double ya = yprev[t * 16 + aoff_p];
#pragma unroll
      for (int f = 0; f < NFX; ++f) ax[f][t] = TEMX_MFMA4(ya, dS[f][t < TBX ? 0 : 1], ax[f][t]);
      const int tp = t < TBX ? t : t - TBX;
      if (tp < TBS) {
        const int ta = t < TBX ? tp : TBS + tp;
#pragma unroll
        for (int k = 0; k < NPR; ++k) apr[k][ta] = TEMX_MFMA4(ya, dP[k][t < TBX ? 0 : 1], apr[k][ta]);
        if constexpr (NPL > 0) {
          double v[NPL > 0 ? NPL : 1];
#pragma unroll
          for (int k = 0; k < NPL; ++k) v[k] = apl[(k * 2 * TBS + ta) * 64];
#pragma unroll
          for (int k = 0; k < NPL; ++k) v[k] = TEMX_MFMA4(ya, dP[NPR + k][t < TBX ? 0 : 1], v[k]);
#pragma unroll
          for (int k = 0; k < NPL; ++k) apl[(k * 2 * TBS + ta) * 64] = v[k];
        }
      }
    }
  };
  auto park_north = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int f = 0; f < NF; ++f) { sN[f] = s[f]; x0N[f] = x0[f]; s[f] = 0.0; }
#pragma unroll
    for (int k = 0; k < NP; ++k) { qN[k] = q[k]; q[k] = 0.0; }
    cntN = cnt;
    cnt = 0.0;
  };
  int4 rn;
  auto step = [&](auto posc, int b) __attribute__((always_inline)) {
    constexpr int POS = decltype(posc)::value % NCH;
    constexpr int P = decltype(posc)::value % PD;
    {                                         // (past b1: the next cut's rows or the table's padding, never used)
      const int4 r1 = rn;
      rn = crow[(int64_t)(b + PD) * 4 + wave];
      issue(std::integral_constant<int, (P + PD - 1) % PD>{}, r1);
    }
    if (left > 0) {                           // the loads of the next batch are in flight meanwhile
      if (!(TEMX_OS_SKIP & 1)) pending_chunk(std::integral_constant<int, POS>{});
      --left;
    }
    // PD - 1 batches were issued after this one; its rows become usable one by one
    static_assert(NF == 4 || NF == 3, "row_wait has forms for 3 and 4 fields");
    static_for<MB>([&](auto jc) __attribute__((always_inline)) {
      constexpr int j = decltype(jc)::value;
      constexpr int NW = (PD - 1) * MB * NF + (MB - 1 - j) * NF;
      if constexpr (NF == 4) row_wait<NW>(xb[P][j][0], xb[P][j][1], xb[P][j][2], xb[P][j][3]);
      else row_wait<NW>(xb[P][j][0], xb[P][j][1], xb[P][j][2]);
    });
#pragma unroll
    for (int j = 0; j < YJ; ++j) row_touch(ys[j]);   // (issued at least one step ago, in order before this batch)
    const int fl = er[P][0] >> 27;            // (flags are those of the batch: the same in its four class slots)
    const bool south = (fl & (CLS_SOUTH << 1)) != 0;
    if ((fl & (CLS_FIRST << 1)) || (south && !prev_south)) {
      if (south && north_open) park_north();
      north_open = !south;
#pragma unroll
      for (int f = 0; f < NF; ++f) x0[f] = (double)xb[P][0][f];
    }
    prev_south = south;
#pragma unroll
    for (int j = 0; j < ((TEMX_OS_SKIP & 4) ? 0 : MB); ++j) {
      const double w = er[P][j] < 0 ? 0.0 : 1.0;   // (a padding entry: the whole row of this wave)
      double dx[NF];
#pragma unroll
      for (int f = 0; f < NF; ++f) dx[f] = (double)xb[P][j][f] - x0[f];
#pragma unroll
      for (int f = 0; f < NF; ++f) s[f] += w * dx[f];
#pragma unroll
      for (int k = 0; k < NP; ++k) q[k] += (w * dx[KD::pa(k)]) * dx[KD::pb(k)];
      cnt += w;
    }
    if (fl & (CLS_LAST << 1)) {
      prev_south = false;
      if (north_open) park_north();           // the group has no southern batch: the open side is the northern one
      north_open = false;
      if (left > 0)                           // (a group shorter than NCH steps: what is left of the previous projection)
        static_for<NCH>([&](auto cc) __attribute__((always_inline)) {
          if (((decltype(cc)::value - POS - 1) & (NCH - 1)) < left) pending_chunk(cc);
        });
      // ---- reading role: side means (theta = T x the column scale) and central co-moments of my class
      const double rnN = cntN > 0.0 ? temx_rcp_count(cntN) : 0.0, rnS = cnt > 0.0 ? temx_rcp_count(cnt) : 0.0;
      double val[NV];
#pragma unroll
      for (int f = 0; f < NF; ++f) {
        const double sc = f == KD::TF ? sth : 1.0;
        val[f] = (sN[f] * rnN + x0N[f]) * sc;
        val[NF + NP + f] = (s[f] * rnS + x0[f]) * sc;
      }
#pragma unroll
      for (int k = 0; k < NP; ++k) {
        const double sc = k == KD::TP ? sth : 1.0;
        val[NF + k] = (qN[k] - sN[KD::pa(k)] * sN[KD::pb(k)] * rnN) * sc;
        val[2 * NF + NP + k] = (q[k] - s[KD::pa(k)] * s[KD::pb(k)] * rnS) * sc;
      }
      // every wave is done with the exchange area of the previous group (LDS reads retired, loads stay in flight)
      if (!(TEMX_OS_SKIP & 8)) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      ycur ^= 1;
      double* yw = ybase + ycur * YE;
#pragma unroll
      for (int v = 0; v < NV; ++v) exw[v * 256] = val[v];
      if (lane == 0) {
        cn[wave] = cntN;
        cn[4 + wave] = cnt;
      }
#pragma unroll
      for (int j = 0; j < YJ; ++j)
        if (tid + 256 * j < YE) yw[tid + 256 * j] = ys[j];
      if (!(TEMX_OS_SKIP & 8)) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      ++grp;
      load_ys(grp);
      // ---- tile role: class slot g, column c of d-tile `wave`
      double m2[2][NF], c2[2][NP];
#pragma unroll
      for (int f = 0; f < NF; ++f) {
        m2[0][f] = exr[f * 256];
        m2[1][f] = exr[(NF + NP + f) * 256];
      }
#pragma unroll
      for (int k = 0; k < NP; ++k) {
        c2[0][k] = exr[(NF + k) * 256];
        c2[1][k] = exr[(2 * NF + NP + k) * 256];
      }
      const double nN = cn[g], nS = cn[4 + g];
      // reference at the class latitudes: E = even part, O = odd part; r_N = E + O, r_S = E - O
      double E[NF], O[NF];
#pragma unroll
      for (int f = 0; f < NF; ++f) E[f] = O[f] = 0.0;
#pragma unroll
      for (int tb = 0; tb < ((TEMX_OS_SKIP & 2) ? 0 : 2 * NBR); ++tb) {
        const int blk = tb < NBR ? tb : TBX + (tb - NBR);
        const double ya = yw[blk * 16 + aoff_r];
#pragma unroll
        for (int f = 0; f < NF; ++f) {
          if (tb < NBR)
            E[f] = TEMX_MFMA4(ya, cb[(f * 2 * NBR + tb) * 64], E[f]);
          else
            O[f] = TEMX_MFMA4(ya, cb[(f * 2 * NBR + tb) * 64], O[f]);
        }
      }
#pragma unroll
      for (int f = 0; f < NF; ++f) {
        m2[0][f] -= E[f] + O[f];              // side mean minus the reference
        m2[1][f] -= E[f] - O[f];
      }
#pragma unroll
      for (int f = 0; f < NFX; ++f) {
        const double SNf = nN * m2[0][f], SSf = nS * m2[1][f];
        dS[f][0] = SNf + SSf;
        dS[f][1] = SNf - SSf;
      }
#pragma unroll
      for (int k = 0; k < NP; ++k) {
        const double PNk = c2[0][k] + nN * m2[0][KD::pa(k)] * m2[0][KD::pb(k)];
        const double PSk = c2[1][k] + nS * m2[1][KD::pa(k)] * m2[1][KD::pb(k)];
        dP[k][0] = PNk + PSk;
        dP[k][1] = PNk - PSk;
      }
      yprev = yw;
      left = NCH;
#pragma unroll
      for (int f = 0; f < NF; ++f) s[f] = sN[f] = 0.0;
#pragma unroll
      for (int k = 0; k < NP; ++k) q[k] = qN[k] = 0.0;
      cnt = cntN = 0.0;
    }
  };

  if (b0 < b1) {
    load_ys(grp);
    rn = crow[(int64_t)b0 * 4 + wave];
    static_for<PD - 1>([&](auto kc) __attribute__((always_inline)) {
      constexpr int k = decltype(kc)::value;
      const int4 r0 = rn;
      rn = crow[(int64_t)(b0 + k + 1) * 4 + wave];
      issue(kc, r0);
    });
    constexpr int UNR = PD % 4 == 0 ? PD : PD % 2 == 0 ? 2 * PD : 4 * PD;   // lcm(PD, NCH)
    for (int b = b0; b < b1; b += UNR)
      static_for<UNR>([&](auto kc) __attribute__((always_inline)) {
        constexpr int k = decltype(kc)::value;
        if (k == 0 || b + k < b1) step(kc, b + k);
      });
    // the batches issued past b1 and the last Y prefetch are still landing in registers the compiler believes
    // free from here on: drain them before anything else is written there
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  if (left > 0) {
    const int first = (b1 - b0) & (NCH - 1);
    static_for<NCH>([&](auto cc) __attribute__((always_inline)) {
      if (((decltype(cc)::value - first) & (NCH - 1)) < left) pending_chunk(cc);
    });
  }
  if (dvalid) {
#pragma unroll
    for (int f = 0; f < NFX; ++f)
#pragma unroll
      for (int t = 0; t < NBX; ++t) {
        const int l = symx_harm<TBX>(t, g);
        if (l < (pp != nullptr ? KX : K4)) px[(((int64_t)split * NFX + f) * KX + l) * D + d] = ax[f][t];   // (pre-pass, pp == NULL: only the K4 rows of the reference fit)
      }
#pragma unroll
    for (int k = 0; k < NP; ++k)
#pragma unroll
      for (int t = 0; t < 2 * TBS; ++t) {
        const int l = sym_harm<TBS>(t, g);
        if (pp != nullptr && l < K) pp[(((int64_t)split * NP + k) * K + l) * D + d] = k < NPR ? apr[k < NPR ? k : 0][t] : apl[((k < NPR ? 0 : k - NPR) * 2 * TBS + t) * 64];
      }
  }
}

// ------------------------------------------------------------------------------------------------
// The single sweep with TWO waves per SIMD.  sweep_osr_kernel is one in-order wave per SIMD: with fp32 inputs its
// instruction stream (7.5 ms at ne120 x 72 x 30) no longer fits under its loads (5.2 ms), and nothing runs while it
// waits.  146 accumulators per d-tile leave no room for a second wave -- unless a wave holds half of them.  Here a
// workgroup is 8 waves:
//   * reading role: wave w owns class slot w & 3 on side w >> 2 (north / south) and the workgroup's 64 columns; it
//     walks the row table of ITS side (side_tables.hpp), so it carries one side's sums only;
//   * tile role: wave w owns d-tile w & 3 and PARITY w >> 2 of the harmonics: the even blocks take the sum of the
//     two sides' operands, the odd blocks their difference -- 4 x 13 field accumulators in registers and 3 x 7
//     product accumulators in LDS per wave.  Both parity waves reconstruct the whole reference (8 MFMAs twice).
// The exchange, the barriers, the deferred projection (two chunks here: a side of a cubed-sphere class is two
// batches) and the hand-issued loads are those of sweep_osr_kernel.
// ------------------------------------------------------------------------------------------------
#ifndef TEMX_OS2_STAGGER
#define TEMX_OS2_STAGGER 0
#endif
#ifndef TEMX_OS2_ACC32
#define TEMX_OS2_ACC32 1
#endif
template <typename T, int TBS, int TBX, int NBR, int PD, int KIND = 0>
__global__ void __launch_bounds__(512, 1)
sweep_os2_kernel(FieldPtrs<4> fp, int64_t D, int K, int KX, const double* __restrict__ ycx,
                 const int4* __restrict__ crowN, const int4* __restrict__ crowS,
                 const int* __restrict__ gfirstN, const int* __restrict__ gfirstS, const int2* __restrict__ csplit,
                 const double* __restrict__ colscale, const double* __restrict__ rho, int K4,
                 double* __restrict__ px, double* __restrict__ pp, int nsplit, int ndt) {
  using KD = OsKind<KIND>;
  constexpr int NF = KD::NF, NFX = KD::NFX, NP = KD::NP;
  constexpr int NBX = 2 * TBX;
  constexpr int YE = NBX * 16;
  constexpr int MB = CLS_MB;
  constexpr int NCH = 2;
  constexpr int NV = NF + NP;                 // exchanged per (class side, column): mean of each field, central co-moments
  constexpr int NPR = KD::NPR, NPL = NP - NPR; // product accumulators in registers / in LDS
  static_assert(NBR <= TBS && TBS <= TBX, "reference degree <= L <= 2L");
  static_assert(YE <= 512, "one Y element per thread");
  static_assert((PD - 1) * MB * NF + (MB - 1) * NF + 1 <= 63, "the ring is counted in vmcnt (6 bits)");
  // [2][YE] Y blocks | [16] member counts | [4 d-tiles][NF][2 NBR][64] reference operands |
  // [8 waves][NP - NPR][TBS][64] product accumulators | [NV][8 class sides][64 columns] exchange
  extern __shared__ double lds[];
  int split, dq;
  if (!wg_work((ndt + 3) / 4, nsplit, split, dq)) return;
  const int tid = threadIdx.x;
  const int wave = uniform_wave(), lane = tid & 63;
  const int c = lane & 15, g = lane >> 4;
  // reading role: class slot wave & 3 of side wave >> 2, column dq * 64 + lane
  const int side = wave >> 2;
  const int64_t colr = (int64_t)dq * 64 + lane < D ? (int64_t)dq * 64 + lane : D - 1;
  const uint32_t colb32 = (uint32_t)colr * (uint32_t)sizeof(T);
  // tile role: d-tile dq * 4 + (wave & 3), harmonics of parity wave >> 2, lane = (class slot g, column c)
  const int tl = wave & 3, par = wave >> 2;
  const int dt = dq * 4 + tl;
  const int64_t d = (int64_t)dt * 16 + c;
  const bool dvalid = dt < ndt && d < D;
  const int64_t dcl = dvalid ? d : D - 1;
  const int4* __restrict__ crow = side ? crowS : crowN;
  const int* __restrict__ gfirst = side ? gfirstS : gfirstN;
  int grp = __builtin_amdgcn_readfirstlane(csplit[split].y);
  const int grp1 = __builtin_amdgcn_readfirstlane(csplit[split + 1].y);
  const int b0 = __builtin_amdgcn_readfirstlane(gfirst[grp]);
  const int b1 = __builtin_amdgcn_readfirstlane(gfirst[grp1]);
  const uint32_t aoff_p = (uint32_t)(g * 4 + (lane & 3));
  const uint32_t aoff_r = (uint32_t)((lane & 3) * 4 + g);
  double* ybase = lds;
  double* cn = lds + 2 * YE;
  double* cb = lds + 2 * YE + 16 + tl * (NF * 2 * NBR * 64) + lane;
  double* apl = lds + 2 * YE + 16 + 4 * (NF * 2 * NBR * 64) + wave * (NPL * TBS * 64) + lane;
  double* ex = lds + 2 * YE + 16 + 4 * (NF * 2 * NBR * 64) + 8 * (NPL * TBS * 64);
  double* exw = ex + wave * 64 + lane;                   // [v][my class side][my column]
  const double* exr = ex + g * 64 + tl * 16 + c;         // [v][north side of class g][column of my d-tile]; south: + 256
  const double sth = (KD::TF >= 0 && colscale != nullptr) ? colscale[colr] : 1.0;
#pragma unroll
  for (int i = 0; i < NPL * TBS; ++i) apl[i * 64] = 0.0;
  double apr[NPR > 0 ? NPR : 1][TBS];
#pragma unroll
  for (int k = 0; k < (NPR > 0 ? NPR : 1); ++k)
#pragma unroll
    for (int t = 0; t < TBS; ++t) apr[k][t] = 0.0;
  if (par == 0) {
#pragma unroll
    for (int f = 0; f < NF; ++f)
#pragma unroll
      for (int tb = 0; tb < 2 * NBR; ++tb) {
        const int l = tb < NBR ? 2 * (4 * tb + g) : 2 * (4 * (tb - NBR) + g) + 1;
        const double v = rho[((int64_t)f * K4 + (l < K ? l : K - 1)) * D + dcl];
        cb[(f * 2 * NBR + tb) * 64] = l < K ? v : 0.0;
      }
  }
  uint64_t fbase[NF];
#pragma unroll
  for (int f = 0; f < NF; ++f) fbase[f] = reinterpret_cast<uint64_t>(fp.p[f]);
  // The column scale is the one load of this kernel the compiler tracks; used first inside the loop, its wait would
  // be placed THERE -- as `s_waitcnt vmcnt(0)`, since the pass cannot see the hand-issued loads behind it: a drain of
  // the ring once per class-group (found by tools/isa_check.py in round 4; ne120 x 72 x 30 fp32: 6.6 -> see DESIGN 5d).
  // A use here settles it before the first row load is issued.
  { double sth_ready = sth; asm volatile("" : "+v"(sth_ready)); }

  double ax[NFX][TBX];
#pragma unroll
  for (int f = 0; f < NFX; ++f)
#pragma unroll
    for (int t = 0; t < TBX; ++t) ax[f][t] = 0.0;
  // fp32 inputs: the sums of a class side about its first member, S~ = sum (x - x0) and q~ = sum (a - a0)(b - b0), are
  // accumulated in fp32 (TEMX_OS2_ACC32).  The differences are eddy-sized and a side has 8 members on a cubed sphere:
  // the rounding of these sums is ~1e-7 of the EDDY amplitude, four orders below the 2e-5 the fp32 path is held to
  // (the reference's own fp32-input run differs from its fp64 run by 5e-6), and everything from the side
  // means on is fp64 as before.  It takes the conversions and the quarter-rate fp64 VALU work out of the inner loop.
  using AT = typename std::conditional<(sizeof(T) == 4 && TEMX_OS2_ACC32), float, double>::type;
  AT s[NF], q[NP], x0[NF], cnt = 0;
#pragma unroll
  for (int f = 0; f < NF; ++f) s[f] = x0[f] = 0;
#pragma unroll
  for (int k = 0; k < NP; ++k) q[k] = 0;
  const uint32_t rowbytes = (uint32_t)D * (uint32_t)sizeof(T);   // host guarantees D < 2^28

  T xb[PD][MB][NF];
  int er[PD][MB];                             // wave-uniform: the rows of this wave's class side
  double ys;
  const uint32_t yoff32 = (uint32_t)(tid < YE ? tid : 0) * 8u;
  auto load_ys = [&](int gi) __attribute__((always_inline)) {
    RowLoad<double>::ld(ys, yoff32, reinterpret_cast<uint64_t>(ycx) + (uint64_t)gi * (YE * 8));
  };
  auto issue = [&](auto pc, const int4 rv) __attribute__((always_inline)) {
    constexpr int P = decltype(pc)::value;
    er[P][0] = rv.x; er[P][1] = rv.y; er[P][2] = rv.z; er[P][3] = rv.w;
#pragma unroll
    for (int j = 0; j < MB; ++j) {
      const uint64_t off = (uint64_t)(uint32_t)(er[P][j] & CLS_ROWMASK) * rowbytes;   // wave-uniform
#pragma unroll
      for (int f = 0; f < NF; ++f) RowLoad<T>::ld(xb[P][j][f], colb32, fbase[f] + off);
    }
  };
  double dS[NFX], dP[NP];                     // operands of the pending projection: my parity's combination
#pragma unroll
  for (int f = 0; f < NFX; ++f) dS[f] = 0.0;
#pragma unroll
  for (int k = 0; k < NP; ++k) dP[k] = 0.0;
  const double* yprev = ybase + par * (TBX * 16);
  int ycur = 0, left = 0;
  auto pending_chunk = [&](auto cc) __attribute__((always_inline)) {
    constexpr int C = decltype(cc)::value;
#pragma unroll
    for (int t = C * TBX / NCH; t < (C + 1) * TBX / NCH; ++t) {
      const double ya = yprev[t * 16 + aoff_p];
#pragma unroll
      for (int f = 0; f < NFX; ++f) ax[f][t] = TEMX_MFMA4(ya, dS[f], ax[f][t]);
      if (t < TBS) {                          // the product blocks are the first TBS of the parity
#pragma unroll
        for (int k = 0; k < NPR; ++k) apr[k][t] = TEMX_MFMA4(ya, dP[k], apr[k][t]);
        if constexpr (NPL > 0) {
          double v[NPL > 0 ? NPL : 1];
#pragma unroll
          for (int k = 0; k < NPL; ++k) v[k] = apl[(k * TBS + t) * 64];
#pragma unroll
          for (int k = 0; k < NPL; ++k) v[k] = TEMX_MFMA4(ya, dP[NPR + k], v[k]);
#pragma unroll
          for (int k = 0; k < NPL; ++k) apl[(k * TBS + t) * 64] = v[k];
        }
      }
    }
  };
  int4 rn;
  auto step = [&](auto posc, int b) __attribute__((always_inline)) {
    constexpr int POS = decltype(posc)::value % NCH;
    constexpr int P = decltype(posc)::value % PD;
    {                                         // (past b1: the next cut's rows or the table's padding, never used)
      const int4 r1 = rn;
      rn = crow[(int64_t)(b + PD) * 4 + (wave & 3)];
      issue(std::integral_constant<int, (P + PD - 1) % PD>{}, r1);
    }
    // The two waves of a SIMD (one per side of the same classes) reach every barrier together, so left alone they
    // also run the same phase at the same time: both on the matrix pipe, then both on the VALU, then both waiting.
    // TEMX_OS2_STAGGER: the southern wave runs its projection chunk AFTER the accumulation instead of before it, so
    // that one wave's MFMAs meet the other's VALU work (the pipes are separate).
    const bool early = !(TEMX_OS2_STAGGER) || side == 0;
    if (early && left > 0) {                  // the loads of the next batch are in flight meanwhile
      pending_chunk(std::integral_constant<int, POS>{});
      --left;
    }
    static_for<MB>([&](auto jc) __attribute__((always_inline)) {
      constexpr int j = decltype(jc)::value;
      constexpr int NW = (PD - 1) * MB * NF + (MB - 1 - j) * NF;
      if constexpr (NF == 4) row_wait<NW>(xb[P][j][0], xb[P][j][1], xb[P][j][2], xb[P][j][3]);
      else row_wait<NW>(xb[P][j][0], xb[P][j][1], xb[P][j][2]);
    });
    row_touch(ys);
    const int fl = er[P][0] >> 27;            // has-padding, (south), first, last: of the batch on this side
    if (fl & (CLS_FIRST << 1)) {
#pragma unroll
      for (int f = 0; f < NF; ++f) x0[f] = (AT)xb[P][0][f];
    }
#pragma unroll
    for (int j = 0; j < MB; ++j) {
      const AT w = er[P][j] < 0 ? (AT)0 : (AT)1;   // (a padding entry: the whole row of this wave)
      AT dx[NF];
#pragma unroll
      for (int f = 0; f < NF; ++f) dx[f] = (AT)xb[P][j][f] - x0[f];
#pragma unroll
      for (int f = 0; f < NF; ++f) s[f] += w * dx[f];
#pragma unroll
      for (int k = 0; k < NP; ++k) q[k] += (w * dx[KD::pa(k)]) * dx[KD::pb(k)];
      cnt += w;
    }
    if (!early && left > 0) {
      pending_chunk(std::integral_constant<int, POS>{});
      --left;
    }
    if (fl & (CLS_LAST << 1)) {
      if (left > 0)                           // (a side shorter than NCH steps: what is left of the previous projection)
        static_for<NCH>([&](auto cc) __attribute__((always_inline)) {
          if (((decltype(cc)::value - POS - 1) & (NCH - 1)) < left) pending_chunk(cc);
        });
      // ---- reading role: mean (theta = T x the column scale) and central co-moments of my class side
      const double cntd = (double)cnt;
      const double rcn = cntd > 0.0 ? temx_rcp_count(cntd) : 0.0;
      double val[NV];
#pragma unroll
      for (int f = 0; f < NF; ++f) val[f] = ((double)s[f] * rcn + (double)x0[f]) * (f == KD::TF ? sth : 1.0);
#pragma unroll
      for (int k = 0; k < NP; ++k)
        val[NF + k] = ((double)q[k] - (double)s[KD::pa(k)] * (double)s[KD::pb(k)] * rcn) * (k == KD::TP ? sth : 1.0);
      // every wave is done with the exchange area of the previous group (LDS reads retired, loads stay in flight)
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      ycur ^= 1;
      double* yw = ybase + ycur * YE;
#pragma unroll
      for (int v = 0; v < NV; ++v) exw[v * 512] = val[v];
      if (lane == 0) cn[wave] = cntd;
      if (tid < YE) yw[tid] = ys;
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      ++grp;
      load_ys(grp);
      // ---- tile role: class slot g, column c of d-tile tl; both sides of the class
      double m2[2][NF], c2[2][NP];
#pragma unroll
      for (int f = 0; f < NF; ++f) {
        m2[0][f] = exr[f * 512];
        m2[1][f] = exr[f * 512 + 256];
      }
#pragma unroll
      for (int k = 0; k < NP; ++k) {
        c2[0][k] = exr[(NF + k) * 512];
        c2[1][k] = exr[(NF + k) * 512 + 256];
      }
      const double nN = cn[g], nS = cn[4 + g];
      // reference at the class latitudes: E = even part, O = odd part; r_N = E + O, r_S = E - O
      double E[NF], O[NF];
#pragma unroll
      for (int f = 0; f < NF; ++f) E[f] = O[f] = 0.0;
#pragma unroll
      for (int tb = 0; tb < 2 * NBR; ++tb) {
        const int blk = tb < NBR ? tb : TBX + (tb - NBR);
        const double ya = yw[blk * 16 + aoff_r];
#pragma unroll
        for (int f = 0; f < NF; ++f) {
          if (tb < NBR)
            E[f] = TEMX_MFMA4(ya, cb[(f * 2 * NBR + tb) * 64], E[f]);
          else
            O[f] = TEMX_MFMA4(ya, cb[(f * 2 * NBR + tb) * 64], O[f]);
        }
      }
#pragma unroll
      for (int f = 0; f < NF; ++f) {
        m2[0][f] -= E[f] + O[f];              // side mean minus the reference
        m2[1][f] -= E[f] - O[f];
      }
      const double sg = par ? -1.0 : 1.0;     // even harmonics: north + south; odd: north - south
#pragma unroll
      for (int f = 0; f < NFX; ++f) dS[f] = nN * m2[0][f] + sg * (nS * m2[1][f]);
#pragma unroll
      for (int k = 0; k < NP; ++k) {
        const double PNk = c2[0][k] + nN * m2[0][KD::pa(k)] * m2[0][KD::pb(k)];
        const double PSk = c2[1][k] + nS * m2[1][KD::pa(k)] * m2[1][KD::pb(k)];
        dP[k] = PNk + sg * PSk;
      }
      yprev = yw + par * (TBX * 16);
      left = NCH;
#pragma unroll
      for (int f = 0; f < NF; ++f) s[f] = 0;
#pragma unroll
      for (int k = 0; k < NP; ++k) q[k] = 0;
      cnt = 0;
    }
  };

  if (b0 < b1) {
    load_ys(grp);
    rn = crow[(int64_t)b0 * 4 + (wave & 3)];
    static_for<PD - 1>([&](auto kc) __attribute__((always_inline)) {
      constexpr int k = decltype(kc)::value;
      const int4 r0 = rn;
      rn = crow[(int64_t)(b0 + k + 1) * 4 + (wave & 3)];
      issue(kc, r0);
    });
    constexpr int UNR = PD % 2 == 0 ? PD : 2 * PD;   // lcm(PD, NCH)
    for (int b = b0; b < b1; b += UNR)
      static_for<UNR>([&](auto kc) __attribute__((always_inline)) {
        constexpr int k = decltype(kc)::value;
        if (k == 0 || b + k < b1) step(kc, b + k);
      });
    // loads issued past b1 and the last Y prefetch are still landing in registers the compiler believes free
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  if (left > 0) {
    const int first = (b1 - b0) & (NCH - 1);
    static_for<NCH>([&](auto cc) __attribute__((always_inline)) {
      if (((decltype(cc)::value - first) & (NCH - 1)) < left) pending_chunk(cc);
    });
  }
  if (dvalid) {
#pragma unroll
    for (int f = 0; f < NFX; ++f)
#pragma unroll
      for (int t = 0; t < TBX; ++t) {
        const int l = 2 * (4 * t + g) + par;  // block t of parity par, row g (symx_harm<TBX>(par * TBX + t, g))
        if (l < (pp != nullptr ? KX : K4)) px[(((int64_t)split * NFX + f) * KX + l) * D + d] = ax[f][t];   // (pre-pass, pp == NULL: only the K4 rows of the reference fit)
      }
#pragma unroll
    for (int k = 0; k < NP; ++k)
#pragma unroll
      for (int t = 0; t < TBS; ++t) {
        const int l = 2 * (4 * t + g) + par;
        if (pp != nullptr && l < K) pp[(((int64_t)split * NP + k) * K + l) * D + d] = k < NPR ? apr[k < NPR ? k : 0][t] : apl[((k < NPR ? 0 : k - NPR) * TBS + t) * 64];
      }
  }
}

// ------------------------------------------------------------------------------------------------
// Sweep 1 of the class-sum form with row-contiguous loads: sweep_op_kernel (kernels_op.hpp: same inputs, class-sum
// records, partial slabs and arithmetic) read the way sweep_osr_kernel reads.  While it reads, wave w owns class
// slot w of every class-group and the workgroup's 64 columns; at the end of a group it stores its class's sums
// into the four records of the group (16 columns = 256 B of each of the four d-tiles per instruction), the
// {sum, difference} operands of (class, column) cross through LDS, and wave w projects d-tile w, deferred over
// the next four batches.
// ------------------------------------------------------------------------------------------------
template <typename T, int TBS, int PD, int KIND>
__global__ void __launch_bounds__(256, OpKind<KIND>::WPS)
sweep_opr_kernel(FieldPtrs<4> fp, int64_t D, int K, const double* __restrict__ ycls,
                 const int4* __restrict__ crow,
                 const int2* __restrict__ csplit, const double* __restrict__ colscale,
                 double* __restrict__ partial, int nsplit, int ndt, double* __restrict__ csum) {
  using KD = OpKind<KIND>;
  constexpr int NB = 2 * TBS;
  constexpr int YE = NB * 16;
  constexpr int MB = CLS_MB;
  constexpr int NFLD = KD::NFLD, NST = KD::NST, NQ = KD::NQ;
  constexpr int NA = NST + NQ;                // projections: the stored sums, then the co-moments
  constexpr int NCH = 4;
  static_assert(KIND == 0 || KIND == 1, "TEM or tracer");
  static_assert(YE <= 256, "one Y element per thread");
  static_assert(PD + 1 <= CLS_PADB, "table padding must cover the index prefetch");
  static_assert((PD - 1) * MB * NFLD + (MB - 1) * NFLD + NST + 1 <= 63, "the ring is counted in vmcnt (6 bits)");
  __shared__ double ybuf[2][YE];              // the group's Y blocks, shared by the four waves
  __shared__ double ex[2 * NA][4][64];        // [sums then differences][class slot][column]
  int split, dq;
  if (!wg_work((ndt + 3) / 4, nsplit, split, dq)) return;
  const int tid = threadIdx.x;
  const int wave = uniform_wave(), lane = tid & 63;
  const int c = lane & 15, g = lane >> 4;
  // reading role: class slot `wave`, column dq * 64 + lane
  const int64_t colr_ = (int64_t)dq * 64 + lane;
  const bool rvalid = colr_ < D;
  const int64_t colr = rvalid ? colr_ : D - 1;
  const uint32_t colb32 = (uint32_t)colr * (uint32_t)sizeof(T);
  // tile role: d-tile dq * 4 + wave, lane = (class slot g, column c)
  const int dt = dq * 4 + wave;
  const int64_t d = (int64_t)dt * 16 + c;
  const bool dvalid = dt < ndt && d < D;
  const int b0 = __builtin_amdgcn_readfirstlane(csplit[split].x);
  const int b1 = __builtin_amdgcn_readfirstlane(csplit[split + 1].x);
  int grp = __builtin_amdgcn_readfirstlane(csplit[split].y);
  const uint32_t yoff = (uint32_t)(g * 4 + (lane & 3));
  const double sth = (KIND == 0 && colscale != nullptr) ? colscale[colr] : 1.0;
  uint64_t fbase[NFLD];
#pragma unroll
  for (int f = 0; f < NFLD; ++f) fbase[f] = reinterpret_cast<uint64_t>(fp.p[f]);

  double acc[NA][NB];
#pragma unroll
  for (int f = 0; f < NA; ++f)
#pragma unroll
    for (int t = 0; t < NB; ++t) acc[f][t] = 0.0;
  double s[NFLD], q[NQ], x0[NFLD], cnt = 0.0;
  double sN[NST], qN[NQ];
#pragma unroll
  for (int f = 0; f < NFLD; ++f) s[f] = x0[f] = 0.0;
#pragma unroll
  for (int k = 0; k < NQ; ++k) q[k] = qN[k] = 0.0;
#pragma unroll
  for (int f = 0; f < NST; ++f) sN[f] = 0.0;
  bool north_open = false, prev_south = false;
  const uint32_t rowbytes = (uint32_t)D * (uint32_t)sizeof(T);   // host guarantees D < 2^28

  T xb[PD][MB][NFLD];
  int er[PD][MB];                             // wave-uniform: the rows of this wave's class slot
  double ys;
  const uint32_t yoff32 = (uint32_t)(tid < YE ? tid : 0) * 8u;
  auto load_ys = [&](int gi) __attribute__((always_inline)) {
    RowLoad<double>::ld(ys, yoff32, reinterpret_cast<uint64_t>(ycls) + (uint64_t)gi * (YE * 8));
  };
  auto issue = [&](auto pc, const int4 rv) __attribute__((always_inline)) {
    constexpr int P = decltype(pc)::value;
    er[P][0] = rv.x; er[P][1] = rv.y; er[P][2] = rv.z; er[P][3] = rv.w;
#pragma unroll
    for (int j = 0; j < MB; ++j) {
      const uint64_t off = (uint64_t)(uint32_t)(er[P][j] & CLS_ROWMASK) * rowbytes;   // wave-uniform, one 32 x 32 -> 64 multiply
#pragma unroll
      for (int f = 0; f < NFLD; ++f) RowLoad<T>::ld(xb[P][j][f], colb32, fbase[f] + off);
    }
  };
  auto finish_side = [&](double* so, double* qo) __attribute__((always_inline)) {
    const double rn = cnt > 0.0 ? temx_rcp_count(cnt) : 0.0;
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
  // operands of the group whose projection is pending (tile role), its Y buffer, chunks still to run
  double pS[NA], pD[NA];
#pragma unroll
  for (int f = 0; f < NA; ++f) pS[f] = pD[f] = 0.0;
  const double* yprev = ybuf[0];
  int ycur = 0, left = 0;
  auto pending_chunk = [&](auto cc) __attribute__((always_inline)) {
    constexpr int C = decltype(cc)::value;
#pragma unroll
    for (int t = C * NB / NCH; t < (C + 1) * NB / NCH; ++t) {
      const double ya = yprev[t * 16 + yoff];
#pragma unroll
      for (int f = 0; f < NA; ++f) acc[f][t] = TEMX_MFMA4(ya, t < TBS ? pS[f] : pD[f], acc[f][t]);
    }
  };
  int4 rn;
  auto step = [&](auto posc, int b) __attribute__((always_inline)) {
    constexpr int POS = decltype(posc)::value % NCH;
    constexpr int P = decltype(posc)::value % PD;
    {                                         // (past b1: the next cut's rows or the table's padding, never used)
      const int4 r1 = rn;
      rn = crow[(int64_t)(b + PD) * 4 + wave];
      issue(std::integral_constant<int, (P + PD - 1) % PD>{}, r1);
    }
    if (left > 0) {
      pending_chunk(std::integral_constant<int, POS>{});
      --left;
    }
    static_for<MB>([&](auto jc) __attribute__((always_inline)) {
      constexpr int j = decltype(jc)::value;
      constexpr int NW = (PD - 1) * MB * NFLD + (MB - 1 - j) * NFLD;
      if constexpr (NFLD == 4) row_wait<NW>(xb[P][j][0], xb[P][j][1], xb[P][j][2], xb[P][j][3]);
      else row_wait<NW>(xb[P][j][0], xb[P][j][1], xb[P][j][2]);
    });
    row_touch(ys);
    const int fl = er[P][0] >> 27;            // haspad, south, first, last: those of the batch
    const bool south = (fl & (CLS_SOUTH << 1)) != 0;
    if ((fl & (CLS_FIRST << 1)) || (south && !prev_south)) {
      if (south && north_open) finish_side(sN, qN);
      north_open = !south;
#pragma unroll
      for (int f = 0; f < NFLD; ++f) x0[f] = (double)xb[P][0][f];
    }
    prev_south = south;
#pragma unroll
    for (int j = 0; j < MB; ++j) {
      const double w = er[P][j] < 0 ? 0.0 : 1.0;   // (a padding entry: the whole row of this wave)
      double dx[NFLD];
#pragma unroll
      for (int f = 0; f < NFLD; ++f) dx[f] = (double)xb[P][j][f] - x0[f];
#pragma unroll
      for (int f = 0; f < NFLD; ++f) s[f] += w * dx[f];
#pragma unroll
      for (int k = 0; k < NQ; ++k) q[k] += (w * dx[KD::pa(k)]) * dx[KD::pb(k)];
      cnt += w;
    }
    if (fl & (CLS_LAST << 1)) {
      prev_south = false;
      double sS[NST], qS[NQ];
#pragma unroll
      for (int f = 0; f < NST; ++f) sS[f] = 0.0;
#pragma unroll
      for (int k = 0; k < NQ; ++k) qS[k] = 0.0;
      if (north_open)
        finish_side(sN, qN);                  // the group has no southern batch
      else
        finish_side(sS, qS);
      north_open = false;
      if (KIND == 0) {                        // T -> theta: field 2 and the v theta co-moment
        sN[NST > 2 ? 2 : 0] *= sth; sS[NST > 2 ? 2 : 0] *= sth;
        qN[NQ - 1] *= sth; qS[NQ - 1] *= sth;
      }
      if (rvalid) {                           // record (grp, d-tile lane >> 4), row f, element [class slot][column]
        double2* o = reinterpret_cast<double2*>(csum + TEMX_CSUM_REC(grp, dq * 4 + g, ndt) * (2 * NST) * 64) + wave * 16 + c;
#pragma unroll
        for (int f = 0; f < NST; ++f) TEMX_CSTORE(o + f * 64, make_double2(sN[f], sS[f]));
      }
      if (left > 0)                           // (a group shorter than NCH steps: what is left of the previous projection)
        static_for<NCH>([&](auto cc) __attribute__((always_inline)) {
          if (((decltype(cc)::value - POS - 1) & (NCH - 1)) < left) pending_chunk(cc);
        });
      // every wave is done with the exchange area of the previous group (LDS reads retired, loads stay in flight)
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      ycur ^= 1;
#pragma unroll
      for (int f = 0; f < NST; ++f) {
        ex[f][wave][lane] = sN[f] + sS[f];
        ex[NA + f][wave][lane] = sN[f] - sS[f];
      }
#pragma unroll
      for (int k = 0; k < NQ; ++k) {          // the class co-moments are projected like field sums
        ex[NST + k][wave][lane] = qN[k] + qS[k];
        ex[NA + NST + k][wave][lane] = qN[k] - qS[k];
      }
      if (tid < YE) ybuf[ycur][tid] = ys;
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      ++grp;
      load_ys(grp);                           // ycls is padded by one group
#pragma unroll
      for (int f = 0; f < NA; ++f) {
        pS[f] = ex[f][g][wave * 16 + c];
        pD[f] = ex[NA + f][g][wave * 16 + c];
      }
      yprev = ybuf[ycur];
      left = NCH;
#pragma unroll
      for (int f = 0; f < NST; ++f) sN[f] = 0.0;
#pragma unroll
      for (int k = 0; k < NQ; ++k) qN[k] = 0.0;
    }
  };

  if (b0 < b1) {
    load_ys(grp);
    rn = crow[(int64_t)b0 * 4 + wave];
    static_for<PD - 1>([&](auto kc) __attribute__((always_inline)) {
      constexpr int k = decltype(kc)::value;
      const int4 r0 = rn;
      rn = crow[(int64_t)(b0 + k + 1) * 4 + wave];
      issue(kc, r0);
    });
    constexpr int UNR = PD % 4 == 0 ? PD : PD % 2 == 0 ? 2 * PD : 4 * PD;   // lcm(PD, NCH)
    for (int b = b0; b < b1; b += UNR)
      static_for<UNR>([&](auto kc) __attribute__((always_inline)) {
        constexpr int k = decltype(kc)::value;
        if (k == 0 || b + k < b1) step(kc, b + k);
      });
    // loads issued past b1 and the last Y prefetch are still landing in registers the compiler believes free
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  if (left > 0) {
    const int first = (b1 - b0) & (NCH - 1);
    static_for<NCH>([&](auto cc) __attribute__((always_inline)) {
      if (((decltype(cc)::value - first) & (NCH - 1)) < left) pending_chunk(cc);
    });
  }
  // (an empty range still stores its zero slab: the reduction sums every slab)
  if (dvalid) {
#pragma unroll
    for (int f = 0; f < NA; ++f)
#pragma unroll
      for (int t = 0; t < NB; ++t) {
        const int l = sym_harm<TBS>(t, g);
        if (l < K) partial[(((int64_t)split * NA + f) * K + l) * D + d] = acc[f][t];
      }
  }
}

// reference coefficients of the single-sweep form: rho[f][l][d] = sum_m Gsinv[l][m] As[f][m][d], l, m < KR
// (As: projections of a subsample of class-groups onto the first KR harmonics, the first rows of each field of a
// [nf][KX][D] array; summed over the ranks when the job is ncol-sharded)
__global__ void __launch_bounds__(256)
os_ref_solve_kernel(const double* __restrict__ As, int KX /* rows per field in As */, int KR, int64_t D,
                    const double* __restrict__ Gsinv, double* __restrict__ rho) {
  // one thread per (l, d): 16 x the threads of a thread per column, which was latency bound at 22 us (36 workgroups)
  __shared__ double sg[16 * 16];
  const int f = blockIdx.y;
  for (int i = threadIdx.x; i < KR * KR; i += blockDim.x) sg[i] = Gsinv[i];
  __syncthreads();
  const int64_t d = blockIdx.x * (int64_t)(blockDim.x / 16) + (threadIdx.x & 15);
  const int l = threadIdx.x >> 4;
  if (d >= D || l >= KR) return;
  const double* a = As + (int64_t)f * KX * D + d;
  double v0 = 0.0, v1 = 0.0;
  int m = 0;
  for (; m + 2 <= KR; m += 2) {
    v0 += sg[l * KR + m] * a[(int64_t)m * D];
    v1 += sg[l * KR + m + 1] * a[(int64_t)(m + 1) * D];
  }
  if (m < KR) v0 += sg[l * KR + m] * a[(int64_t)m * D];
  rho[((int64_t)f * KR + l) * D + d] = v0 + v1;
}

// ------------------------------------------------------------------------------------------------
// os_contract_kernel: from the projections of the single sweep to the raw sums the rest of the pipeline
// takes.  Per column d, in the Y basis (A^f_k, k < KX: shifted fields; P^p_l, l < K: shifted products; rho: the
// reference that was subtracted), with x_q, w_q the Gauss-Legendre nodes and weights (NQ = 2L + 2: exact
// for degree 4L) and Yq[q][k] = Y_k(x_q):
//   alpha^f = T (G2inv (T^T A^f[:K]))                    coefficients of the zonal mean of the shifted field
//   B4^f    = T^T (A^f[:K] + G[:, :KR] rho^f)             raw sums of the ORIGINAL field, plan basis
//   At^f_q  = sum_k Yq[q][k] A^f_k,   ab^f_q = sum_l Yq[q][l] alpha^f_l      (synthesis at the nodes)
//   sum_i Y_l abar b   = sum_q 2 pi w_q Y_l(x_q) ab^a_q At^b_q               (Y_l abar is a polynomial of degree <= 2L,
//                                                                            so only the degree-2L projection of b matters)
//   c_k  = sum_q 2 pi w_q Yq[q][k] ab^a_q ab^b_q          the product of the two zonal means, degree <= 2L
//   F_l  = P_l - sum_q 2 pi w_q Yq[q][l] (ab^b_q At^a_q + ab^a_q At^b_q) + sum_k Gx_lk c_k
//   B3^p = T^T F                                          raw sums of the eddy products, plan basis
// (the transform form of the Legendre product linearisation: 1.3e5 multiply-adds per column instead of the
// 1.6e6 of the explicit g(l,m,k) sums).  One workgroup = OSC columns; thread = (row slot, column); every
// matrix operand is staged in LDS before it is used.
// ------------------------------------------------------------------------------------------------
constexpr int OSC = 4;                          // columns per workgroup
constexpr int OSR = 256 / OSC;                  // row slots

__device__ __forceinline__ void os_stage(double* dst, const double* __restrict__ src, int n, int tid) {
  for (int i = tid; i < n; i += 256) dst[i] = src[i];
}

// LDS doubles of os_contract_kernel
__host__ __device__ inline size_t os_contract_lds(int K, int KX, int NQ) {
  const size_t big = (size_t)NQ * KX > (size_t)2 * K * K ? (size_t)NQ * KX : (size_t)2 * K * K;
  return ((size_t)4 * KX + 8 * K + 8 * NQ + 3 * KX + 3 * K) * OSC + big;
}

// per field: its degree-2L projections [KX][D] and its reference coefficients [KR][D]
struct OsFields {
  const double* A[4];
  const double* rho[4];
};

// KIND 0: TEM (4 fields, raw sums of all four out, 3 products); KIND 1: tracer (q, v, omega with the projections
// and references of v and omega taken from the TEM run; raw sums of q out, 2 products)
template <int KIND>
__global__ void __launch_bounds__(256)
os_contract_kernel(OsFields in, const double* __restrict__ Pp,
                   int K, int KX, int KR, int NQ, int64_t D, const double* __restrict__ Tm,
                   const double* __restrict__ G2inv, const double* __restrict__ G, const double* __restrict__ Gx,
                   const double* __restrict__ Yq, const double* __restrict__ wq2, double* __restrict__ B4,
                   double* __restrict__ B3, int64_t Drho, int nts, int nt, int t0) {
  using KD = OsKind<KIND>;
  constexpr int NF = KD::NF, NP = KD::NP, NOUT = KIND == 0 ? 4 : 1;
  static_assert(OSC == 4, "a thread owns one output row for the four columns of the workgroup");
  extern __shared__ double sm[];
  const int KK = K * K;
  double* sA = sm;                              // [4][KX][OSC]   projections of the shifted fields
  double* sAl = sA + 4 * KX * OSC;              // [4][K][OSC]    scratch / Y-basis sums
  double* sW = sAl + 4 * K * OSC;               // [4][K][OSC]    alpha (from phase 3 on)
  double* sAt = sW + 4 * K * OSC;               // [4][NQ][OSC]   At
  double* sAb = sAt + 4 * NQ * OSC;             // [4][NQ][OSC]   ab
  double* sC = sAb + 4 * NQ * OSC;              // [3][KX][OSC]   c_k per pair
  double* sF = sC + 3 * KX * OSC;               // [3][K][OSC]    cross terms, then F
  double* big = sF + 3 * K * OSC;               // T | G2inv, then Yq, then Gx, then T
  const int tid = threadIdx.x;
  const int64_t d0 = (int64_t)blockIdx.x * OSC;
  // a matrix row times the OSC columns of a vector block: one scattered LDS read of the matrix element and one
  // broadcast read of the four column values per term
  auto dot4 = [&](const double* mrow, int mstride, const double* v, int n, double (&acc)[OSC]) __attribute__((always_inline)) {
    // (unrolled: the loop is LDS-latency bound with one workgroup per CU; eight terms in flight)
#pragma unroll 8
    for (int k = 0; k < n; ++k) {
      const double m = mrow[(size_t)k * mstride];
      const double4 x = *reinterpret_cast<const double4*>(v + k * OSC);
      acc[0] += m * x.x; acc[1] += m * x.y; acc[2] += m * x.z; acc[3] += m * x.w;
    }
  };
  auto put4 = [&](double* dst, const double (&a)[OSC]) __attribute__((always_inline)) {
    *reinterpret_cast<double4*>(dst) = make_double4(a[0], a[1], a[2], a[3]);
  };
  for (int i = tid; i < NF * KX * OSC; i += 256) {
    const int row = i / OSC, c = i % OSC;
    const int64_t d = d0 + c < D ? d0 + c : D - 1;
    sA[i] = in.A[row / KX][(int64_t)(row % KX) * D + d];
  }
  os_stage(big, Tm, KK, tid);
  os_stage(big + KK, G2inv, KK, tid);
  __syncthreads();
  const double* sT = big;
  const double* sG2 = big + KK;
  for (int i = tid; i < NF * K; i += 256) {     // y = T^T A[:K]   (row j of T^T = column j of T, rows l <= j)
    const int f = i / K, j = i % K;
    double a[OSC] = {0.0, 0.0, 0.0, 0.0};
    dot4(sT + j, K, sA + (size_t)f * KX * OSC, j + 1, a);
    put4(sW + (size_t)i * OSC, a);
  }
  __syncthreads();
  for (int i = tid; i < NF * K; i += 256) {     // C' = G2inv y
    const int f = i / K, j = i % K;
    double a[OSC] = {0.0, 0.0, 0.0, 0.0};
    dot4(sG2 + j * K, 1, sW + (size_t)f * K * OSC, K, a);
    put4(sAl + (size_t)i * OSC, a);
  }
  __syncthreads();
  for (int i = tid; i < NF * K; i += 256) {     // alpha = T C'   (row l of T: columns j >= l)
    const int f = i / K, l = i % K;
    double a[OSC] = {0.0, 0.0, 0.0, 0.0};
    dot4(sT + l * K + l, 1, sAl + ((size_t)f * K + l) * OSC, K - l, a);
    put4(sW + (size_t)i * OSC, a);
  }
  __syncthreads();
  for (int i = tid; i < K * KR; i += 256) big[KK + i] = G[(i / KR) * K + (i % KR)];   // G[:, :KR] over G2inv
  for (int i = tid; i < NOUT * KR * OSC; i += 256) {                                  // rho block -> sAt (free until the synthesis)
    const int row = i / OSC, c = i % OSC;
    const int64_t d = d0 + c < D ? d0 + c : D - 1;
    // the references are those of the whole run ([KR][Drho], Drho = nlev * nt); this call may work on the
    // snapshots [t0, t0 + nts) only (a time-sliced tail): column (lev, t) of the slice is column (lev, t0 + t) there
    const int64_t lev = d / nts, dg = lev * nt + t0 + (d - lev * nts);
    sAt[i] = in.rho[row / KR][(int64_t)(row % KR) * Drho + dg];
  }
  __syncthreads();
  for (int i = tid; i < NOUT * K; i += 256) {   // Y-basis sums of the original fields: A[:K] + G[:, :KR] rho
    const int f = i / K, l = i % K;
    double a[OSC];
#pragma unroll
    for (int c = 0; c < OSC; ++c) a[c] = sA[((size_t)f * KX + l) * OSC + c];
    dot4(big + KK + l * KR, 1, sAt + (size_t)f * KR * OSC, KR, a);
    put4(sAl + (size_t)i * OSC, a);
  }
  __syncthreads();
  for (int i = tid; i < NOUT * K; i += 256) {   // B4 = T^T (.)
    const int f = i / K, j = i % K;
    double a[OSC] = {0.0, 0.0, 0.0, 0.0};
    dot4(sT + j, K, sAl + (size_t)f * K * OSC, j + 1, a);
#pragma unroll
    for (int c = 0; c < OSC; ++c)
      if (d0 + c < D) B4[((int64_t)f * K + j) * D + d0 + c] = a[c];
  }
  __syncthreads();
  os_stage(big, Yq, NQ * KX, tid);              // Yq[q][k]
  __syncthreads();
  for (int i = tid; i < NF * NQ; i += 256) {    // synthesis at the nodes
    const int f = i / NQ, q = i % NQ;
    double a[OSC] = {0.0, 0.0, 0.0, 0.0}, b[OSC] = {0.0, 0.0, 0.0, 0.0};
    dot4(big + q * KX, 1, sA + (size_t)f * KX * OSC, KX, a);
    dot4(big + q * KX, 1, sW + (size_t)f * K * OSC, K, b);
    put4(sAt + (size_t)i * OSC, a);
    put4(sAb + (size_t)i * OSC, b);
  }
  __syncthreads();
  for (int i = tid; i < NP * KX; i += 256) {    // cross terms (rows l < K) and c_k, per pair
    const int p = i / KX, k = i % KX;
    const int fa = KD::pa(p), fb = KD::pb(p);
    double x[OSC] = {0.0, 0.0, 0.0, 0.0}, cc[OSC] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll 4
    for (int q = 0; q < NQ; ++q) {
      const double yw = big[q * KX + k] * wq2[q];
      const double4 aa = *reinterpret_cast<const double4*>(sAb + ((size_t)fa * NQ + q) * OSC);
      const double4 bb = *reinterpret_cast<const double4*>(sAb + ((size_t)fb * NQ + q) * OSC);
      cc[0] += yw * aa.x * bb.x; cc[1] += yw * aa.y * bb.y; cc[2] += yw * aa.z * bb.z; cc[3] += yw * aa.w * bb.w;
      if (k < K) {
        const double4 ta = *reinterpret_cast<const double4*>(sAt + ((size_t)fa * NQ + q) * OSC);
        const double4 tb = *reinterpret_cast<const double4*>(sAt + ((size_t)fb * NQ + q) * OSC);
        x[0] += yw * (bb.x * ta.x + aa.x * tb.x); x[1] += yw * (bb.y * ta.y + aa.y * tb.y);
        x[2] += yw * (bb.z * ta.z + aa.z * tb.z); x[3] += yw * (bb.w * ta.w + aa.w * tb.w);
      }
    }
    put4(sC + (size_t)i * OSC, cc);
    if (k < K) put4(sF + ((size_t)p * K + k) * OSC, x);
  }
  __syncthreads();
  os_stage(big, Gx, K * KX, tid);
  __syncthreads();
  for (int i = tid; i < NP * K; i += 256) {     // F = P - cross + Gx c
    const int p = i / K, l = i % K;
    double t3[OSC] = {0.0, 0.0, 0.0, 0.0};
    dot4(big + l * KX, 1, sC + (size_t)p * KX * OSC, KX, t3);
#pragma unroll
    for (int c = 0; c < OSC; ++c) {
      const int64_t d = d0 + c < D ? d0 + c : D - 1;
      t3[c] = Pp[((int64_t)p * K + l) * D + d] - sF[(size_t)i * OSC + c] + t3[c];
    }
    put4(sAl + (size_t)i * OSC, t3);            // (F in sAl: sF is still being read by other threads)
  }
  __syncthreads();
  os_stage(big, Tm, KK, tid);
  __syncthreads();
  for (int i = tid; i < NP * K; i += 256) {     // B3 = T^T F
    const int p = i / K, j = i % K;
    double a[OSC] = {0.0, 0.0, 0.0, 0.0};
    dot4(big + j, K, sAl + (size_t)p * K * OSC, j + 1, a);
#pragma unroll
    for (int c = 0; c < OSC; ++c)
      if (d0 + c < D) B3[((int64_t)p * K + j) * D + d0 + c] = a[c];
  }
}

}  // namespace temx
