// kernels_op.hpp -- sweep 1 of the one-pass latitude-class path (see kernels_cls.hpp for the class
// tables and the algebra).  One wave = one d-tile (16 (lev, time) columns) x a list of member-row
// batches; it reads u, v, T, omega ONCE and produces, per class-group:
//   * the four field sums of every class side                    -> csum (8 x 512 B per group, d-tile)
//   * seven projections: the four fields and the centred class co-moments of u v, u omega, v theta
//                                                                 -> partial[split][7][K][D]
// The class sums are accumulated about the side's first member (S~ = sum (x - x0), q~ = sum
// (u - u0)(v - v0)) and converted when the side is complete:  S = S~ + n x0,  C_uv = q~ - S~_u S~_v / n.
//
// This is the dominant kernel of the pipeline and an HBM stream.  Round 3 (tools/sweep_lab.hip):
// built without its class-sum stores it reads at 5.98 TB/s -- the chip's streaming ceiling -- and forms with
// two waves per SIMD or 16-byte loads (kernels_op2.hpp) are no faster with the stores in place: the 12.5 % of
// store traffic costs 2.2 of the 11.1 ms, and that is a property of the memory system (an independent writer
// kernel costs the same).  One wave per SIMD (98 fp64 accumulators); the loop body is kept lean:
//   * D is passed as 32 bits, so the element offset of a member row is one v_mad_u64_u32;
//   * the sums of the side being walked live in ONE set of registers (no per-batch north / south
//     selects); they move to the "north" set when the southern batches of the group begin;
//   * batches without padding entries (the rule: cubed-sphere classes have 8 members a side) skip the
//     member weights.
// Row table: crow of kernels_cls.hpp (one int4 of member rows per class and batch, batch flags in the
// top bits of every entry).
#pragma once
#include "kernels_cls.hpp"

namespace temx {


// KIND 0: TEM     fields (u, v, T -> theta, omega); stored sums of all four; co-moments u v, u omega, v theta
// KIND 1: tracer  fields (q, v, omega); stored sum of q only (those of v and omega are in the TEM run's
//                 csum); co-moments q v, q omega   (tem_diagnostics.py:532-538, 560-570)
// KIND 2: TEM + one tracer in one sweep (kernels_op2.hpp, sweep_opw_kernel only): fields (u, v, T -> theta, omega, q);
//                 stored sums: the four TEM ones (csum) and q's (csq); co-moments u v, u omega, v theta, q v, q omega
// NPTR: field pointers handed over; TF / TQ: index of theta among the stored sums / of the v theta co-moment (-1: none);
// NSTA: pairs of a class-sum record in `csum` (the remaining NST - NSTA go to records of `csq`)
template <int KIND> struct OpKind;
template <> struct OpKind<0> {
  static constexpr int NFLD = 4, NST = 4, NQ = 3, WPS = 1, NPTR = 4, TF = 2, TQ = 2, NSTA = 4;
  __host__ __device__ static constexpr int pa(int k) { return k == 2 ? 1 : 0; }              // u u v
  __host__ __device__ static constexpr int pb(int k) { return k == 0 ? 1 : (k == 1 ? 3 : 2); }   // v w theta
};
template <> struct OpKind<1> {
  static constexpr int NFLD = 3, NST = 1, NQ = 2, WPS = 2, NPTR = 4, TF = -1, TQ = -1, NSTA = 1;
  __host__ __device__ static constexpr int pa(int) { return 0; }                              // q q
  __host__ __device__ static constexpr int pb(int k) { return k == 0 ? 1 : 2; }               // v w
};
template <> struct OpKind<2> {
  static constexpr int NFLD = 5, NST = 5, NQ = 5, WPS = 1, NPTR = 5, TF = 2, TQ = 2, NSTA = 4;
  __host__ __device__ static constexpr int pa(int k) { return k == 2 ? 1 : (k >= 3 ? 4 : 0); }                // u u v q q
  __host__ __device__ static constexpr int pb(int k) { return k == 0 ? 1 : (k == 1 ? 3 : (k == 2 ? 2 : (k == 3 ? 1 : 3))); }   // v w theta v w
};

template <typename T, int TBS, int PD, int KIND>
__global__ void __launch_bounds__(256, OpKind<KIND>::WPS)
sweep_op_kernel(FieldPtrs<4> fp, int64_t D, int K, const double* __restrict__ ycls,
                const int4* __restrict__ crow,
                const int2* __restrict__ csplit, const double* __restrict__ colscale,
                double* __restrict__ partial, int nsplit, int ndt, double* __restrict__ csum) {
  using KD = OpKind<KIND>;
  constexpr int NB = 2 * TBS;
  constexpr int YE = NB * 16;
  constexpr int YJ = (YE + 63) / 64;
  constexpr int MB = CLS_MB;
  constexpr int NFLD = KD::NFLD, NST = KD::NST, NQ = KD::NQ;
  constexpr int NA = NST + NQ;                // projections: the stored sums, then the co-moments
  static_assert(MB == 4, "a lane group reads its 4 member rows as one int4");
  static_assert(PD + 1 <= CLS_PADB, "table padding must cover the index prefetch");
  __shared__ double ystage[4][YE];            // wave private
  int split, dq;
  if (!wg_work((ndt + 3) / 4, nsplit, split, dq)) return;
  const int wave = uniform_wave(), lane = threadIdx.x & 63;
  const int c = lane & 15, g = lane >> 4;
  const int dt = dq * 4 + wave;
  if (dt >= ndt) return;                      // (no barriers below)
  const int64_t d = (int64_t)dt * 16 + c;
  const bool dvalid = d < D;
  const int64_t dcl = dvalid ? d : D - 1;
  const int b0 = __builtin_amdgcn_readfirstlane(csplit[split].x);
  const int b1 = __builtin_amdgcn_readfirstlane(csplit[split + 1].x);
  int grp = __builtin_amdgcn_readfirstlane(csplit[split].y);
  const uint32_t yoff = (uint32_t)(g * 4 + (lane & 3));
  double* yst = ystage[wave];
  // theta = T (p0/p)^kappa: a per-column scale, applied to the finished sums (TEM only)
  const double sth = (KIND == 0 && colscale != nullptr) ? colscale[dcl] : 1.0;
  const T* fb[NFLD];
#pragma unroll
  for (int f = 0; f < NFLD; ++f) fb[f] = reinterpret_cast<const T*>(fp.p[f]) + dcl;

  double acc[NA][NB];
#pragma unroll
  for (int f = 0; f < NA; ++f)
#pragma unroll
    for (int t = 0; t < NB; ++t) acc[f][t] = 0.0;
  // side being walked (shifted sums), and the finished northern side of the group (true sums, co-moments)
  double s[NFLD], q[NQ], x0[NFLD], cnt = 0.0;
  double sN[NST], qN[NQ];
#pragma unroll
  for (int f = 0; f < NFLD; ++f) s[f] = x0[f] = 0.0;
#pragma unroll
  for (int k = 0; k < NQ; ++k) q[k] = qN[k] = 0.0;
#pragma unroll
  for (int f = 0; f < NST; ++f) sN[f] = 0.0;
  bool north_open = false;                    // uniform: the side being walked is a northern one
  bool prev_south = false;                    // uniform: the previous batch of this group was a southern one
  const uint32_t D32 = (uint32_t)D;           // host guarantees D < 2^28

  T xb[PD][MB][NFLD];
  int er[PD][MB];                             // table entries: row (27 bits), batch flags, padding (sign)
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
  // shifted sums of the side just walked -> true sums and centred co-moments; the side's registers restart
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
    if (b + (PD - 1) < b1) {                  // index load first: it must not queue behind the X loads
      const int4 r1 = rn;
      rn = crow[(int64_t)(b + PD) * 4 + g];
      issue(std::integral_constant<int, (P + PD - 1) % PD>{}, r1);
    }
    const int fl = __builtin_amdgcn_readfirstlane(er[P][0]) >> 27;    // haspad, south, first, last
    const bool south = (fl & (CLS_SOUTH << 1)) != 0;
    if ((fl & (CLS_FIRST << 1)) || (south && !prev_south)) {          // first batch of a side: its first member is the origin
      if (south && north_open) finish_side(sN, qN);
      north_open = !south;
#pragma unroll
      for (int f = 0; f < NFLD; ++f) x0[f] = (double)xb[P][0][f];
    }
    prev_south = south;
    if (fl & 1) {                             // a padding entry reads row 0 and weighs nothing
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
        finish_side(sN, qN);                  // the group has no southern batch
      else
        finish_side(sS, qS);
      north_open = false;
      if (KIND == 0) {                        // T -> theta: field 2 and the v theta co-moment
        sN[NST > 2 ? 2 : 0] *= sth; sS[NST > 2 ? 2 : 0] *= sth;
        qN[NQ - 1] *= sth; qS[NQ - 1] *= sth;
      }
      if (dvalid) {                           // record row f = {northern, southern} sum of field f per lane
        double2* o = reinterpret_cast<double2*>(csum + TEMX_CSUM_REC(grp, dt, ndt) * (2 * NST) * 64) + lane;
#pragma unroll
        for (int f = 0; f < NST; ++f) TEMX_CSTORE(o + f * 64, make_double2(sN[f], sS[f]));
      }
      ++grp;
      load_ys(grp);                           // ycls is padded by one group
      double ss[NA], dd[NA];
#pragma unroll
      for (int f = 0; f < NST; ++f) {
        ss[f] = sN[f] + sS[f];
        dd[f] = sN[f] - sS[f];
      }
#pragma unroll
      for (int k = 0; k < NQ; ++k) {          // the class co-moments are projected like field sums
        ss[NST + k] = qN[k] + qS[k];
        dd[NST + k] = qN[k] - qS[k];
      }
#pragma unroll
      for (int t = 0; t < NB; ++t) {
        const double ya = yst[t * 16 + yoff];
#pragma unroll
        for (int f = 0; f < NA; ++f) acc[f][t] = TEMX_MFMA4(ya, t < TBS ? ss[f] : dd[f], acc[f][t]);
      }
#pragma unroll
      for (int f = 0; f < NST; ++f) sN[f] = 0.0;
#pragma unroll
      for (int k = 0; k < NQ; ++k) qN[k] = 0.0;
    }
  };

  if (b0 < b1) {
    load_ys(grp);
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

// Tried and measured, not kept (round 2): the same sweep with 16-byte loads -- a lane loading 2 (fp64) or
// 4 (fp32) adjacent columns of 8 or 16 member rows per instruction, the partial sums of the lanes that
// share a class added and transposed into the MFMA operand layout through wave-private LDS.  A bare
// gather at one wave per SIMD gains 50 % from the wider loads (tools/ubench_gather32.hip: 4.0 -> 6.1
// TB/s), this kernel lost 4 % on fp64 (ne120 x 72 x 30: 12.3 vs 11.85 ms) and 29 % on fp32 (ne240 x 128
// x 1: 2.77 vs 2.15 ms): with 98 accumulators and one wave per SIMD the extra LDS hand-overs and VALU
// sit on the same in-order instruction stream as the loads.  (Lesson kept: lanes that hand data to each
// other through LDS without a workgroup barrier need a compiler barrier -- asm volatile("" ::: "memory")
// -- between the writes and the reads, or the reads of other lanes' values are hoisted above the writes;
// a C++ fence at wavefront scope is also correct but drains the global loads in flight.)
//
// Also tried and measured, not kept: two waves per SIMD by pairing waves on a d-tile -- one wave walks the
// northern member rows of every class-group, the other the southern ones (per-hemisphere row tables),
// they exchange their 4 sums + 3 co-moments through LDS behind one workgroup barrier per class-group,
// and the northern wave projects N + S onto the even harmonics, the southern N - S onto the odd ones
// (49 accumulators each).  Correct (parity 1.2e-13), but 18.4 ms against 11.3 ms on ne120 x 72 x 30 and
// 3.5 ms against 1.8 ms on ne240 x 128 x 1 fp32: a barrier every two batches phase-locks all eight waves
// of the CU, and 256 registers per wave still spill the loop invariants.
//
// And: two INDEPENDENT waves per SIMD by letting a wave cover 8 of a d-tile's 16 columns, the four blocks
// of the 4x4x4 MFMA used as 2 column-blocks x 2 harmonic-blocks (49 accumulators, 248 registers, no spill,
// no exchange between waves beyond one lane-swap per class side).  Correct (1.3e-13), but the member rows
// are then read as 64-byte pieces: 17.6 ms against 10.8 ms on ne120 x 72 x 30 (3.0 TB/s).

}  // namespace temx
