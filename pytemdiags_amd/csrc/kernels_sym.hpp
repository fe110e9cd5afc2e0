// kernels_sym.hpp -- mirror-paired sweeps for equatorially symmetric native grids (gfx950).
//
// Y_l^0(-lat) = (-1)^l Y_l^0(lat).  When every native column has a mirror column at the opposite
// latitude (any longitude: the basis does not depend on it) -- true for cubed-sphere, lat-lon and
// Gaussian grids -- a pair (n, s) of rows contributes to the projection
//     even l:  Y0[n][l] (x_n + x_s)          odd l:  Y0[n][l] (x_n - x_s)
// and the reconstruction at the pair is   xbar_n = E + O,  xbar_s = E - O   with
//     E = sum_{even l} Y0[n][l] C[l],   O = sum_{odd l} Y0[n][l] C[l].
// So one pair of rows needs TBS even blocks + TBS odd blocks of 4 harmonics (2*7 = 14 for L = 50)
// instead of 2 * 13: 54 % of the MFMAs of the generic sweeps, same bytes, same operator
// (results differ from the generic path by rounding only).  Equator columns (and padding) are
// pairs without a southern partner: x_s := 0 and their odd harmonics vanish by themselves.
//
// Layout (built by sym_basis_kernel):
//   rows[2][npg4]        northern / southern row index per pair (-1: no partner), pair-group major
//   ysym[pg][2*TBS][16]  4x4 blocks for pair-group pg (4 pairs): first the TBS even-harmonic
//                        blocks, then the TBS odd ones; block[k*4+i] = Y0[north row of pair 4pg+k][l(i)]
//                        with l = 2(4t+i) (even) or 2(4t+i)+1 (odd); zero for l >= K or padding pairs.
// Tile scheme, operand layouts and staging are those of kernels.hpp.
#pragma once
#include "kernels.hpp"

namespace temx {

constexpr int SYM_PROJ_CH = 3;   // pair-groups per chunk of the paired project sweep

template <int KMAX>
__global__ void sym_basis_kernel(const double* __restrict__ x, const int* __restrict__ rowN, int64_t npair,
                                 int64_t npair_pad, int K, int TBS, const double* __restrict__ norm,
                                 const double* __restrict__ T, double* __restrict__ ysym) {
  int64_t pi = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (pi >= npair_pad) return;
  const bool valid = pi < npair;
  const double xv = valid ? x[rowN[pi]] : 0.0;
  const int64_t pg = pi >> 2;
  const int k = (int)(pi & 3);
  double* blk = ysym + pg * (2 * TBS * 16);
  double q[KMAX];
  basis_row<KMAX>(xv, K, norm, T, q);          // (T keeps the parity of a column: temx_plan_finalize)
  for (int l = 0; l < 8 * TBS; ++l) {
    const double val = (valid && l < K) ? q[l] : 0.0;
    const int h = l >> 1;                       // index among the even (or odd) harmonics
    const int t = (l & 1) * TBS + (h >> 2);     // block: even blocks first, then odd
    blk[t * 16 + k * 4 + (h & 3)] = val;
  }
}

// harmonic of accumulator / coefficient row (block tb in [0, 2*TBS), element i)
template <int TBS>
__device__ __forceinline__ constexpr int sym_harm(int tb, int i) {
  return tb < TBS ? 2 * (4 * tb + i) : 2 * (4 * (tb - TBS) + i) + 1;
}

// ------------------------------------------------------------------------------------------------
// paired project sweep (sweep 1): partial[split][f][l][d] over this split's pairs
// ------------------------------------------------------------------------------------------------
// NFW fields per wave: the workgroup's 4 waves cover DPW = 4*NFW/NF d-tiles x NF/NFW field groups
// (NFW = 1 for small ragged D, like the generic sweep's config E).
template <typename T, int NF, int NFW, int TBS, int WPS>
__global__ void __launch_bounds__(256, WPS)
project_sym_kernel(FieldPtrs<NF> fp, int64_t D, int K, const double* __restrict__ ysym,
                   const int* __restrict__ rows, int64_t npg, int64_t npg4, const double* __restrict__ colscale,
                   int sfield, double* __restrict__ partial, int nsplit, int ndt) {
  constexpr int DPW = 4 * NFW / NF;
  constexpr int NB = 2 * TBS;                 // blocks per pair-group
  constexpr int CH = SYM_PROJ_CH;             // pair-groups per chunk (CH*8 physical rows)
  constexpr int YE = CH * NB * 16;
  constexpr int YJ = (YE + 255) / 256;
  __shared__ double ystage[2][YE];
  int split, dq;
  if (!wg_work((ndt + DPW - 1) / DPW, nsplit, split, dq)) return;
  const int wave = uniform_wave();
  const int tid = threadIdx.x, lane = tid & 63;
  const int c = lane & 15, g = lane >> 4;
  const int dt = dq * DPW + wave % DPW;
  const int f0 = (wave / DPW) * NFW;          // first field of this wave
  const bool active = dt < ndt;
  const int64_t d = (int64_t)dt * 16 + c;
  const bool dvalid = active && d < D;
  const int64_t dcl = d < D ? d : D - 1;
  const int64_t nchunk = (npg + CH - 1) / CH;             // rows[] and ysym are padded to whole chunks (+1); npg4 = entries per half of rows[]
  const int c0 = (int)(nchunk * split / nsplit), c1 = (int)(nchunk * (split + 1) / nsplit);
  const uint32_t yoff = (uint32_t)(g * 4 + (lane & 3));

  double sc[NFW];
  const T* fb[NFW];
#pragma unroll
  for (int f = 0; f < NFW; ++f) {
    sc[f] = (colscale != nullptr && f0 + f == sfield) ? colscale[dcl] : 1.0;
    fb[f] = reinterpret_cast<const T*>(fp.p[f0 + f]) + dcl;
  }
  double acc[NFW][NB];
#pragma unroll
  for (int f = 0; f < NFW; ++f)
#pragma unroll
    for (int t = 0; t < NB; ++t) acc[f][t] = 0.0;

  T xn[NFW][CH], xs[NFW][CH];
  int rn[CH], rs[CH];       // row indices of the chunk after next (index loads run two chunks ahead)
  double ys[YJ];
  auto load_rows = [&](int chunk) __attribute__((always_inline)) {
#pragma unroll
    for (int ti = 0; ti < CH; ++ti) {
      rn[ti] = rows[((int64_t)chunk * CH + ti) * 4 + g];
      rs[ti] = rows[npg4 + ((int64_t)chunk * CH + ti) * 4 + g];
    }
  };
  auto load_x = [&](int ti) __attribute__((always_inline)) {   // uses rn/rs currently held
    const int64_t on = (int64_t)rn[ti] * D;
    const int64_t os = (int64_t)(rs[ti] < 0 ? rn[ti] : rs[ti]) * D;
#pragma unroll
    for (int f = 0; f < NFW; ++f) {
      xn[f][ti] = fb[f][on];
      xs[f][ti] = fb[f][os];
    }
  };
  auto load_ys = [&](int chunk) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < YJ; ++j) ys[j] = (ysym + (int64_t)chunk * YE)[(tid + 256 * j) < YE ? (tid + 256 * j) : 0];
  };

  // (an empty range still stores its zero slab below: the reduction sums every slab)
  int has_s[CH];
  if (c0 < c1) {
    load_ys(c0);
    load_rows(c0);
    if (active) {
#pragma unroll
      for (int ti = 0; ti < CH; ++ti) {
        load_x(ti);
        has_s[ti] = rs[ti] >= 0;
      }
    }
    load_rows(c0 + 1);        // padded: always in bounds
  }
  for (int chunk = c0; chunk < c1; ++chunk) {
    double* yst = ystage[(chunk - c0) & 1];
#pragma unroll
    for (int j = 0; j < YJ; ++j)
      if (tid + 256 * j < YE) yst[tid + 256 * j] = ys[j];
    __syncthreads();
    const bool more = chunk + 1 < c1;
    if (more) load_ys(chunk + 1);
    if (active) {
#pragma unroll
      for (int ti = 0; ti < CH; ++ti) {
        double ss[NFW], dd[NFW];
#pragma unroll
        for (int f = 0; f < NFW; ++f) {
          const double a = (double)xn[f][ti] * sc[f];
          const double b = has_s[ti] ? (double)xs[f][ti] * sc[f] : 0.0;
          ss[f] = a + b;
          dd[f] = a - b;
        }
        if (more) {           // rn/rs hold chunk+1's rows: issue its X loads into the freed registers
          load_x(ti);
          has_s[ti] = rs[ti] >= 0;
        }
#pragma unroll
        for (int t = 0; t < NB; ++t) {
          const double ya = yst[(ti * NB + t) * 16 + yoff];
#pragma unroll
          for (int f = 0; f < NFW; ++f) acc[f][t] = TEMX_MFMA4(ya, t < TBS ? ss[f] : dd[f], acc[f][t]);
        }
      }
    }
    if (more) load_rows(chunk + 2);   // padded by one chunk; clamped on the host side by allocation
  }

  if (dvalid) {
#pragma unroll
    for (int f = 0; f < NFW; ++f)
#pragma unroll
      for (int t = 0; t < NB; ++t) {
        const int l = sym_harm<TBS>(t, g);
        if (l < K) partial[(((int64_t)split * NF + f0 + f) * K + l) * D + d] = acc[f][t];
      }
  }
}

// ------------------------------------------------------------------------------------------------
// paired eddy / flux sweep (sweep 2).  One step = one pair-group (4 pairs = 8 physical rows).
// ------------------------------------------------------------------------------------------------
template <typename T, int TBS, int MODE, int DPW, int KIND>
__global__ void __launch_bounds__(512, 2)
eddy_sym_kernel(FieldPtrs<4> fp, int64_t D, int K, int K4, const double* __restrict__ ysym,
                const int* __restrict__ rows, int64_t npg, int64_t npg4, int64_t npair,
                const double* __restrict__ colscale,
                const double* __restrict__ C, double* __restrict__ partial, int nsplit, int ndt,
                EddyOut eo) {
  extern __shared__ double lds[];
  constexpr int NB = 2 * TBS;
  constexpr int YE = NB * 16;
  constexpr int YJ = (YE + 63) / 64;
  constexpr int NP = 8 / DPW;
  constexpr int NFR = KIND == 0 ? 4 : 3;
  constexpr int NPR = KIND == 0 ? 3 : 2;
  int split, dq;
  if (!wg_work((ndt + DPW - 1) / DPW, nsplit, split, dq)) return;
  const int wave = uniform_wave(), lane = threadIdx.x & 63;
  const int w4 = wave % DPW, part = wave / DPW;
  const int c = lane & 15, g = lane >> 4;
  const int dt = dq * DPW + w4;
  if (dt >= ndt) return;
  const int64_t d = (int64_t)dt * 16 + c;
  const bool dvalid = d < D;
  const int64_t dcl = dvalid ? d : D - 1;
  const int64_t sub = (int64_t)split * NP + part, nsub = (int64_t)nsplit * NP;
  const int c0 = (int)(npg * sub / nsub), c1 = (int)(npg * (sub + 1) / nsub);

  // coefficient B operands, even blocks then odd blocks: cb[f][tb][lane] = C_f[harm(tb, g)][d]
  {
    // branch free (clamped row + select) so the NB loads of a field are issued back to back; a
    // guarded load per element serialised NFR*NB L2 round trips in every wave's prologue
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

  // (an empty range still stores its zero slab below: the reduction sums every slab)
  T xn[NFR], xs[NFR];
  double ys[YJ];
  const int64_t cs = c0 < c1 ? c0 : 0;
  int rn = rows[cs * 4 + g], rs = rows[npg4 + cs * 4 + g];
  auto load_x = [&]() __attribute__((always_inline)) {
    const int64_t on = (int64_t)rn * D;
    const int64_t os = (int64_t)(rs < 0 ? rn : rs) * D;
#pragma unroll
    for (int f = 0; f < NFR; ++f) {
      xn[f] = fb[f][on];
      xs[f] = fb[f][os];
    }
  };
  auto load_ys = [&](int step) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < YJ; ++j) ys[j] = (ysym + (int64_t)step * YE)[(lane + 64 * j) < YE ? (lane + 64 * j) : 0];
  };
  load_ys((int)cs);
  load_x();
  int crn = rn, crs = rs;                              // rows of the step whose X is in xn/xs
  rn = rows[(cs + 1) * 4 + g];                         // padded: in bounds
  rs = rows[npg4 + (cs + 1) * 4 + g];

  for (int step = c0; step < c1; ++step) {
#pragma unroll
    for (int j = 0; j < YJ; ++j)
      if (lane + 64 * j < YE) yst[lane + 64 * j] = ys[j];
    const bool more = step + 1 < c1;
    if (more) load_ys(step + 1);

    const bool has_s = crs >= 0;
    const int my_rn = crn, my_rs = crs;
    double xN[NFR], xS[NFR];
#pragma unroll
    for (int f = 0; f < NFR; ++f) {
      xN[f] = (double)xn[f];
      xS[f] = (double)xs[f];
    }
    if (KIND == 0) {
      xN[2] *= sth;
      xS[2] *= sth;
    }
    if (more) {                      // rn/rs hold step+1's rows
      load_x();
      crn = rn;
      crs = rs;
      rn = rows[(int64_t)(step + 2) * 4 + g];
      rs = rows[npg4 + (int64_t)(step + 2) * 4 + g];
    }

    asm volatile("" : "+v"(cbi));    // keep the loop-invariant slab reads inside the loop
    const double* cbr = lds + cbi;

    // ---- reconstruction: E = even-harmonic part, O = odd-harmonic part at the northern rows ----
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

    // ---- eddies at both rows of the pair, products, even / odd combinations ----
    double eN[NFR], eS[NFR], sp[NPR], dp[NPR], pN[NPR], pS[NPR];
#pragma unroll
    for (int f = 0; f < NFR; ++f) {
      eN[f] = xN[f] - (E[f] + O[f]);
      eS[f] = has_s ? xS[f] - (E[f] - O[f]) : 0.0;
    }
    if (KIND == 0) {
      pN[0] = eN[0] * eN[1];
      pS[0] = eS[0] * eS[1];
      pN[1] = eN[0] * eN[NFR - 1];
      pS[1] = eS[0] * eS[NFR - 1];
      pN[NPR - 1] = eN[1] * eN[2];
      pS[NPR - 1] = eS[1] * eS[2];
    } else {
      pN[0] = eN[0] * eN[1];
      pS[0] = eS[0] * eS[1];
      pN[1] = eN[0] * eN[2];
      pS[1] = eS[0] * eS[2];
    }
#pragma unroll
    for (int q = 0; q < NPR; ++q) {
      sp[q] = pN[q] + pS[q];
      dp[q] = pN[q] - pS[q];
    }
    if (MODE == 1) {
      if (dvalid && (int64_t)step * 4 + g < npair) {
        const int64_t on = (int64_t)my_rn * D + d, os = (int64_t)my_rs * D + d;
        if (KIND == 0) {
#pragma unroll
          for (int f = 0; f < NFR; ++f)
            if (eo.p[f]) {
              eo.p[f][on] = eN[f];
              if (has_s) eo.p[f][os] = eS[f];
            }
        } else if (eo.p[0]) {
          eo.p[0][on] = eN[0];
          if (has_s) eo.p[0][os] = eS[0];
        }
#pragma unroll
        for (int q = 0; q < NPR; ++q)
          if (eo.p[4 + q]) {
            eo.p[4 + q][on] = pN[q];
            if (has_s) eo.p[4 + q][os] = pS[q];
          }
      }
    }

    // ---- projection: even harmonics see the sums, odd harmonics the differences ----
#pragma unroll
    for (int tb = 0; tb < NB; ++tb) {
      const double ya = yst[tb * 16 + aoff_p];
#pragma unroll
      for (int q = 0; q < NPR; ++q) acc[q][tb] = TEMX_MFMA4(ya, tb < TBS ? sp[q] : dp[q], acc[q][tb]);
    }
  }

  if (dvalid && partial != nullptr) {
#pragma unroll
    for (int q = 0; q < NPR; ++q)
#pragma unroll
      for (int tb = 0; tb < NB; ++tb) {
        const int l = sym_harm<TBS>(tb, g);
        if (l < K) partial[((sub * NPR + q) * K + l) * D + d] = acc[q][tb];
      }
  }
}

}  // namespace temx
