// kernels_osc.hpp -- the contraction of the single sweep on the matrix cores (round 4; replaces os_contract_kernel
// of kernels_op2.hpp, which stays for A/B: TEMX_OPT_OS_CONTRACT / TEMX_OS_CONTRACT=lds).
//
// From the projections of the single sweep to the raw sums the rest of the pipeline takes (the algebra is stated at
// os_contract_kernel; tem_diagnostics.py:547-557 are the products it replaces).  Every step is a small fixed matrix
// applied to columns, so it runs as 16x16x4 fp64 MFMAs on [4 x 16] tiles of 16 columns: one workgroup = one d-tile of
// one field (or of one pair of fields), a tile of a product comes out in the B-operand layout of the next product.  Two
// kernels instead of one, because the pair step needs the synthesis of two fields:
//
//   osc_fields_kernel  per (field, d-tile):  alpha = T (G2inv (T^T A[:K])),  B4 = T^T (A[:K] + G[:, :KR] rho),
//                      At = Yq A,  ab = Yq[:, :K] alpha            -> At, ab [field][NQ rows][Dt]  (and B4)
//   osc_pairs_kernel   per (pair, d-tile):   U = w (ab_b At_a + ab_a At_b),  V = w ab_a ab_b  (per node, VALU),
//                      cross = Yq[:, :K]^T U,  c = Yq^T V,  F = P - cross + Gx c,  B3 = T^T F
//
// The first version of this contraction (os_contract_kernel) stages every matrix in LDS and walks them with scalar
// FMAs, 4 columns per workgroup: 0.23 ms at D = 2160 and growing with D beyond one round of workgroups -- the reason
// the single sweep stopped at grids with >= 2048 class-groups.  A one-kernel MFMA form was measured in round 3 at
// 0.20 ms (59 dependent phases of one L2 latency each on 135 workgroups).  Here the vectors of a d-tile live in LDS in
// the operand layout, the matrix blocks stream from L2 through a register ring PF steps ahead of the MFMAs that use
// them (the loops are rolled: written unrolled, the compiler hoisted every block load of a product and spilled), and
// the four waves of a workgroup share one d-tile, each computing every fourth output block of a product.
//
// Matrices are 16 x 4 A-operand blocks built on the host (append_blocks16): blk[(MB * NKB + kb) * 64 + k * 16 + m] =
// M[16 MB + m][4 kb + k], zero padded to whole blocks; lane (k = lane >> 4, m = lane & 15) loads element `lane` of a block.
#pragma once
#include "kernels_op2.hpp"

namespace temx {

struct OscMats {
  const double* Tt;      // [NBK][NBK]  T^T
  const double* G2inv;   // [NBK][NBK]
  const double* T;       // [NBK][NBK]
  const double* Gk;      // [NBK][4]    G[:, :KR]  (Gram matrix of Y0, Y basis)
  const double* YqK;     // [NBX][NBK]  Yq[:, :K]
  const double* Yq;      // [NBX][NBX]  Yq[q][k], k < KX
  const double* YqKt;    // [NBK][NBX]  Yq[:, :K]^T
  const double* Yqt;     // [NBX][NBX]  Yq^T
  const double* Gx;      // [NBK][NBX]
};

// (Round 4 also built the variant with the matrix of each product staged in LDS by a two-wave workgroup -- an LDS read
// broadcasts, while streamed from L2 every lane of the four 16-lane groups fetches the same 16 doubles of a block:
// 512 bytes of L1 bandwidth per MFMA.  It measured 0.21 ms against 0.088: 144-153 KB of LDS leave one workgroup of
// two waves per CU, and its seven stage-and-barrier phases wait for L2 with nothing else to run.  Not kept.)
#ifndef TEMX_OSC_PF
#define TEMX_OSC_PF 4
#endif
constexpr int OSC_PF = TEMX_OSC_PF;       // matrix blocks are loaded this many steps ahead of their MFMAs

// A workgroup is OSC_W = 4 waves on ONE d-tile: the vectors of the tile sit in LDS once, each wave computes every
// fourth output block of a product (wave, wave + 4, ...), a workgroup barrier separates the products.  A d-tile's
// ~1700 MFMAs then run on four SIMDs instead of one: the contraction of a time slice of an ncol-sharded job (17
// d-tiles) or of ne240 x 128 x 1 (8 d-tiles) was one long wave per d-tile on an otherwise idle chip.
// (Build constants measured on the 16x16x4 form, both kernels, D = 2160 / 6552 / 128: W4 PF4 41.8 / 103.7 / 29.1 us,
// W8 PF4 47.1 / 113.0 / 26.1, W4 PF2 51.8 / 123.1 / 40.5, W8 PF2 47.7 / 122.5 / 34.6, W4 PF6 42.1 / 100.9 / 27.3.)
#ifndef TEMX_OSC_W
#define TEMX_OSC_W 4
#endif
constexpr int OSC_W = TEMX_OSC_W;

// M . x on 16x16x4 fp64 MFMAs:  out rows [4 mb, 4 mb + 4), mb < MO, = init(mb) + sum_kb M[.., 4 kb .. 4 kb + 3] . x[kb],
// kb < KI.  One instruction takes a 16 x 4 block of the matrix as its A operand -- 64 distinct doubles, one per
// lane, a fully coalesced 512-byte load -- against the 4 x 16 operand tile x[kb] (xl[kb * 64] is this lane's element)
// and accumulates 16 rows x 16 columns: result register i of a lane is row 4 i + g of the block, i.e. the lane's own
// slot of the 4-row tile 4 MB + i, so a product's output goes to LDS in the operand layout of the next product with no
// lane movement.  (The first form of this loop used the 4x4x4 instruction with the matrix block shared by its four
// sub-blocks: every lane group fetched the same 16 doubles, 512 bytes of L1 bandwidth per 512 flop, and the kernels ran
// at 30 % of the MFMA rate waiting for the texture addresser -- 120 + 83 us at D = 6552.  This form moves a quarter of
// the bytes per flop.)  Wave w computes the 16-row blocks w, w + OSC_W, ...; the loop over kb is rolled (PF steps per
// trip, the remainder peeled); every step issues its block loads unconditionally (past the end: block KI - 1 again,
// never used; a slot past the last 16-row block: the last block again, discarded) -- a conditional issue makes the
// compiler's wait counts conservative.
typedef double osc_d4 __attribute__((ext_vector_type(4)));

template <int MO, int KI, typename Init, typename Out>
__device__ __forceinline__ void osc_mm(const double* __restrict__ blk, int wave, const double* xl, int lane,
                                       Init init, Out out) {
  constexpr int MO16 = (MO + 3) / 4;
  constexpr int MC = (MO16 + OSC_W - 1) / OSC_W;
  constexpr int PF = OSC_PF < KI ? OSC_PF : KI;
  osc_d4 acc[MC];
  double ring[PF][MC];
  const double* __restrict__ b[MC];
#pragma unroll
  for (int j = 0; j < MC; ++j) {
    const int MB = wave + j * OSC_W < MO16 ? wave + j * OSC_W : MO16 - 1;
    b[j] = blk + (size_t)MB * KI * 64 + lane;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int mb = 4 * MB + i;
      acc[j][i] = mb < MO ? init(mb) : 0.0;
    }
  }
  static_for<PF - 1>([&](auto pc) __attribute__((always_inline)) {
    constexpr int p = decltype(pc)::value;
#pragma unroll
    for (int j = 0; j < MC; ++j) ring[p][j] = b[j][p * 64];
  });
  auto step = [&](auto jc, int k) __attribute__((always_inline)) {
    constexpr int r = decltype(jc)::value;
    const int kn = k + PF - 1 < KI ? k + PF - 1 : KI - 1;
#pragma unroll
    for (int j = 0; j < MC; ++j) ring[(r + PF - 1) % PF][j] = b[j][kn * 64];
    const double x = xl[k * 64];
#pragma unroll
    for (int j = 0; j < MC; ++j) acc[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(ring[r][j], x, acc[j], 0, 0, 0);
  };
  constexpr int TRIPS = KI / PF, REM = KI % PF;
#pragma unroll 1
  for (int t = 0; t < TRIPS; ++t)
    static_for<PF>([&](auto jc) __attribute__((always_inline)) { step(jc, t * PF + decltype(jc)::value); });
  static_for<REM>([&](auto jc) __attribute__((always_inline)) { step(jc, TRIPS * PF + decltype(jc)::value); });
#pragma unroll
  for (int j = 0; j < MC; ++j)
    if (wave + j * OSC_W < MO16) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int mb = 4 * (wave + j * OSC_W) + i;
        if (mb < MO) out(mb, acc[j][i]);
      }
    }
}

// nf fields: A[f] [KX][Dt] (projections of the shifted field), rho[f] [KR][Drho] (only for f < nout)
struct OscFieldsIn {
  const double* A[4];
  const double* rho[4];
};

// LDS doubles per workgroup
__host__ __device__ constexpr int osc_fields_lds(int NBK) { return (2 * NBK + 2 * NBK + 4) * 64; }
__host__ __device__ constexpr int osc_pairs_lds(int NBK) { return (3 * 2 * NBK + 2 * NBK) * 64; }

template <int NBK>
__global__ void __launch_bounds__(OSC_W * 64)
osc_fields_kernel(OscFieldsIn in, OscMats m, int K, int KX, int KR, int NQ, int64_t Dt, int nout,
                  double* __restrict__ B4 /* [nout][K][Dt] */, double* __restrict__ At /* [nf][NQ][Dt] */,
                  double* __restrict__ ab /* [nf][NQ][Dt] */, int64_t Drho, int nts, int nt, int t0) {
  constexpr int NBX = 2 * NBK;
  extern __shared__ double sm[];             // [NBX] a | [NBK] t0 | [NBK] t1 | [4] rho   tiles of 64 lanes
  const int f = blockIdx.y;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int g = lane >> 4, c = lane & 15;
  const int64_t d = (int64_t)blockIdx.x * 16 + c;
  const bool dvalid = d < Dt;
  const int64_t dcl = dvalid ? d : Dt - 1;
  double* va = sm + lane;
  double* vt0 = va + NBX * 64;
  double* vt1 = vt0 + NBK * 64;
  double* vr = vt1 + NBK * 64;
  const double* __restrict__ Af = in.A[f];
  for (int kb = wave; kb < NBX; kb += OSC_W) {
    const int row = 4 * kb + g;
    const double v = Af[(int64_t)(row < KX ? row : KX - 1) * Dt + dcl];
    va[kb * 64] = row < KX ? v : 0.0;
  }
  if (wave == 0 && f < nout) {
    // the references are those of the whole run ([KR][Drho]); this call may work on the snapshots [t0, t0 + nts)
    const int64_t lev = dcl / nts, dg = lev * nt + t0 + (dcl - lev * nts);
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
      const int row = 4 * kb + g;
      const double v = in.rho[f][(int64_t)(row < KR ? row : KR - 1) * Drho + dg];
      vr[kb * 64] = row < KR ? v : 0.0;
    }
  }
  __syncthreads();
  auto zero = [](int) { return 0.0; };
  // gridDim.z == 2: the synthesis of the field itself, At = Yq A (676 of a d-tile's 1742 MFMAs at L = 50), does not
  // depend on alpha and runs in a workgroup of its own (blockIdx.z == 1)
  if (gridDim.z == 2 && blockIdx.z == 1) {
    osc_mm<NBX, NBX>(m.Yq, wave, va, lane, zero, [&](int mb, double v) {
      if (dvalid && 4 * mb + g < NQ) At[((int64_t)f * NQ + 4 * mb + g) * Dt + d] = v;
    });
    return;
  }
  // (rows >= K of A ride along in the last block of A[:K]: the matrices are zero there)
  // ---- coefficients of the zonal mean of the shifted field: alpha = T (G2inv (T^T A[:K]))  -> vt0
  osc_mm<NBK, NBK>(m.Tt, wave, va, lane, zero, [&](int mb, double v) { vt0[mb * 64] = v; });
  __syncthreads();
  osc_mm<NBK, NBK>(m.G2inv, wave, vt0, lane, zero, [&](int mb, double v) { vt1[mb * 64] = v; });
  __syncthreads();
  osc_mm<NBK, NBK>(m.T, wave, vt1, lane, zero, [&](int mb, double v) { vt0[mb * 64] = v; });
  __syncthreads();                           // alpha in vt0; vt1 is free again
  // ---- raw sums of the ORIGINAL field in the plan's basis: B4 = T^T (A[:K] + G[:, :KR] rho)   (f: block uniform)
  if (f < nout) {
    osc_mm<NBK, 4>(m.Gk, wave, vr, lane, [&](int mb) { return va[mb * 64]; }, [&](int mb, double v) { vt1[mb * 64] = v; });
    __syncthreads();
    osc_mm<NBK, NBK>(m.Tt, wave, vt1, lane, zero, [&](int mb, double v) {
      if (dvalid && 4 * mb + g < K) B4[((int64_t)f * K + 4 * mb + g) * Dt + d] = v;
    });
  }
  // ---- synthesis at the Gauss-Legendre nodes: ab = Yq[:, :K] alpha, At = Yq A
  osc_mm<NBX, NBK>(m.YqK, wave, vt0, lane, zero, [&](int mb, double v) {
    if (dvalid && 4 * mb + g < NQ) ab[((int64_t)f * NQ + 4 * mb + g) * Dt + d] = v;
  });
  if (gridDim.z == 1)
    osc_mm<NBX, NBX>(m.Yq, wave, va, lane, zero, [&](int mb, double v) {
      if (dvalid && 4 * mb + g < NQ) At[((int64_t)f * NQ + 4 * mb + g) * Dt + d] = v;
    });
}

// per pair p: the synthesised fields a = pa(p), b = pb(p) ([NQ][Dt] each) and the projection of their product [K][Dt]
struct OscPairsIn {
  const double* At_a[4];
  const double* At_b[4];
  const double* ab_a[4];
  const double* ab_b[4];
  const double* P[4];
};

template <int NBK>
__global__ void __launch_bounds__(OSC_W * 64)
osc_pairs_kernel(OscPairsIn in, OscMats m, const double* __restrict__ wq2, int K, int KX, int NQ, int64_t Dt,
                 double* __restrict__ B3 /* [np][K][Dt] */) {
  constexpr int NBX = 2 * NBK;
  extern __shared__ double sm[];             // [NBX] U | [NBX] V | [NBX] c | [NBK] cross | [NBK] F
  const int p = blockIdx.y;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int g = lane >> 4, c = lane & 15;
  const int64_t d = (int64_t)blockIdx.x * 16 + c;
  const bool dvalid = d < Dt;
  const int64_t dcl = dvalid ? d : Dt - 1;
  double* vU = sm + lane;
  double* vV = vU + NBX * 64;
  double* vc = vV + NBX * 64;
  double* vx = vc + NBX * 64;
  double* vF = vx + NBK * 64;
  const double* __restrict__ Ata = in.At_a[p];
  const double* __restrict__ Atb = in.At_b[p];
  const double* __restrict__ aba = in.ab_a[p];
  const double* __restrict__ abb = in.ab_b[p];
  // ---- products at the nodes:  U_q = w_q (bb At_a + aa At_b),  V_q = w_q aa bb
  for (int kb = wave; kb < NBX; kb += OSC_W) {
    const int q = 4 * kb + g;
    const int qc = q < NQ ? q : NQ - 1;
    const double w = q < NQ ? wq2[qc] : 0.0;
    const int64_t o = (int64_t)qc * Dt + dcl;
    const double ta = Ata[o], tb = Atb[o], aa = aba[o], bb = abb[o];
    vU[kb * 64] = w * (bb * ta + aa * tb);
    vV[kb * 64] = w * (aa * bb);
  }
  __syncthreads();
  auto zero = [](int) { return 0.0; };
  // ---- projected back:  cross_l = sum_q Y_l(x_q) U_q,   c_k = sum_q Y_k(x_q) V_q
  osc_mm<NBK, NBX>(m.YqKt, wave, vU, lane, zero, [&](int mb, double v) { vx[mb * 64] = v; });
  osc_mm<NBX, NBX>(m.Yqt, wave, vV, lane, zero, [&](int mb, double v) { vc[mb * 64] = v; });
  __syncthreads();
  // ---- F = P - cross + Gx c,  B3 = T^T F
  const double* __restrict__ Pp = in.P[p];
  osc_mm<NBK, NBX>(m.Gx, wave, vc, lane,
                   [&](int mb) {
                     const int row = 4 * mb + g;
                     const double v = Pp[(int64_t)(row < K ? row : K - 1) * Dt + dcl];
                     return (row < K ? v : 0.0) - vx[mb * 64];
                   },
                   [&](int mb, double v) { vF[mb * 64] = v; });
  __syncthreads();
  osc_mm<NBK, NBK>(m.Tt, wave, vF, lane, zero, [&](int mb, double v) {
    if (dvalid && 4 * mb + g < K) B3[((int64_t)p * K + 4 * mb + g) * Dt + d] = v;
  });
}

}  // namespace temx
