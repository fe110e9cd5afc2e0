#!/usr/bin/env python3
"""Numerics prototype (CPU, numpy): eddy-product sums WITHOUT per-class storage.
   F_l = sum_i Y_l (a_i - abar_i)(b_i - bbar_i)   with abar = Y alpha (band-limited, degree <= L)
       = P_l - sum_m beta_m M^a_lm - sum_m alpha_m M^b_lm + alpha^T T_l beta
   P_l    = sum_i Y_l a_i b_i                      (projection of the product, degree <= L)
   M^a_lm = sum_i Y_l Y_m a_i = sum_k g(l,m,k) A_k (A_k = sum_i Y_k a_i, k <= 2L: Legendre product linearisation)
   T_l,mn = sum_i Y_l Y_m Y_n = sum_k g(m,n,k) Gx_lk, Gx = Y^T Y_ext
 with an optional reference subtracted from every field first (a constant per column, or a band-limited fit
 on a subsample): the eddies do not change, the cancellation does."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from oracle import tem_oracle as orc
from pytemdiags_amd import synth

ne = int(sys.argv[1]) if len(sys.argv) > 1 else 16
L = int(sys.argv[2]) if len(sys.argv) > 2 else 50
K, K2 = L + 1, 2 * L + 1
lat, lon = synth.cubed_sphere_gll(ne)
nlev, nt = 6, 2
plev = synth.pressure_levels(nlev)
ua, va, ta, wap = synth.analytic_fields(lat, lon, plev, nt, seed=3)
theta = ta * ((orc.P0 / (plev * 100.0)) ** orc.k)[None, :, None]
N = lat.size
F4 = [x.reshape(N, -1) for x in (ua, va, theta, wap)]
D = F4[0].shape[1]
Yx = orc.ylm0_matrix_recurrence(lat, 2 * L)          # [N][2L+1]
Y = Yx[:, :K]
Q, R = np.linalg.qr(Y)
coef = lambda A: np.linalg.solve(R, Q.T @ A)         # least-squares coefficients (degree <= L)

# Gaunt-type coefficients of the m = 0 harmonics by Gauss-Legendre quadrature (exact: degree <= 4L)
xg, wg = np.polynomial.legendre.leggauss(2 * L + 2)
Yg = orc.ylm0_matrix_recurrence(np.rad2deg(np.arcsin(xg)), 2 * L)      # [nodes][2L+1]
# Y_l Y_m = sum_k g[l,m,k] Y_k ; orthonormality on the sphere: int Y_k Y_k' dOmega = delta -> weight 2 pi w
g = np.einsum("q,ql,qm,qk->lmk", 2 * np.pi * wg, Yg[:, :K], Yg[:, :K], Yg)
chk = np.max(np.abs(np.einsum("lmk,ik->ilm", g, Yx[:200]) - Yx[:200, :K, None] * Yx[:200, None, :K]))
print("N=%d L=%d D=%d | linearisation identity on the grid: max err %.2e" % (N, L, D, chk))
Gx = Y.T @ Yx                                       # [K][2L+1]
T = np.einsum("mnk,lk->lmn", g, Gx)                  # [K][K][K]

def direct(a, b):
    ap, bp = a - Y @ coef(a), b - Y @ coef(b)
    return Y.T @ (ap * bp)

def linearised(a, b, ra, rb):
    a, b = a - ra, b - rb                            # reference subtraction (band-limited: eddies unchanged)
    A, B = Yx.T @ a, Yx.T @ b                        # [2L+1][D]
    P = Y.T @ (a * b)
    al, be = coef(a), coef(b)                        # (in the engine: from A[:K], B[:K])
    Ma = np.einsum("lmk,kd->lmd", g, A)
    Mb = np.einsum("lmk,kd->lmd", g, B)
    t1 = np.einsum("lmd,md->ld", Ma, be)
    t2 = np.einsum("lmd,md->ld", Mb, al)
    t3 = np.einsum("lmn,md,nd->ld", T, al, be)
    return P - t1 - t2 + t3, (np.max(np.abs(P)), np.max(np.abs(t1)), np.max(np.abs(t2)), np.max(np.abs(t3)))

def refs(kind):
    out = []
    for f in F4:
        if kind == "none":
            out.append(np.zeros((1, D)))
        elif kind == "const":                        # one row's values: a constant per column
            out.append(f[:1, :].copy())
        else:                                        # band-limited least-squares fit on a 1/16 subsample, degree <= LREF
            LREF = int(os.environ.get("PROTO_LREF", str(L)))
            idx = np.sort(np.random.default_rng(5).choice(N, N // 16, replace=False))
            q_, r_ = np.linalg.qr(Y[idx][:, :LREF + 1])
            out.append(Y[:, :LREF + 1] @ np.linalg.solve(r_, q_.T @ f[idx]))
    return out

pairs = [(0, 1, "u v"), (0, 3, "u w"), (1, 2, "v theta")]
zm = lambda B3: np.linalg.solve(R, np.linalg.solve(R.T, B3))          # G^-1 B3 -> coefficients of the flux zonal mean
for kind in ("none", "const", "fit16"):
    r = refs(kind)
    for ia, ib, name in pairs:
        Fd = direct(F4[ia], F4[ib])
        Fl, mags = linearised(F4[ia], F4[ib], r[ia], r[ib])
        e_raw = np.max(np.abs(Fl - Fd)) / np.max(np.abs(Fd))
        Yp = orc.ylm0_matrix_recurrence(np.arange(-89.5, 90, 1.0), L)
        zd, zl = Yp @ zm(Fd), Yp @ zm(Fl)
        e_zm = np.max(np.abs(zl - zd)) / np.max(np.abs(zd))
        print("reference %-6s %-8s raw sums err %.2e | zonal mean of the flux err %.2e | |P| %.1e |t1| %.1e |t2| %.1e |t3| %.1e |F| %.1e"
              % (kind, name, e_raw, e_zm, *mags, np.max(np.abs(Fd))))

# ---- the ten results through the oracle's epilogue: exact zonal means of the four fields, flux means from either form
if os.environ.get("PROTO_EPILOGUE", "1") == "1":
    lat_zm = orc.zm_latitudes(1, False)
    Yp = orc.ylm0_matrix_recurrence(lat_zm, L)
    shape = (lat_zm.size, nlev, nt)
    zmean = lambda f: (Yp @ coef(f)).reshape(shape)
    base = {"ub": zmean(F4[0]), "vb": zmean(F4[1]), "thetab": zmean(F4[2]), "wapb": zmean(F4[3])}
    r = refs("fit16")
    out = {}
    for form in ("direct", "lin"):
        z = dict(base)
        for (ia, ib, name), key in zip(pairs, ("upvpb", "upwappb", "vptpb")):
            Fm = direct(F4[ia], F4[ib]) if form == "direct" else linearised(F4[ia], F4[ib], r[ia], r[ib])[0]
            z[key] = (Yp @ zm(Fm)).reshape(shape)
        o = orc.TEMOracle.from_zonal_means(z, plev)
        out[form] = {n: np.asarray(getattr(o, n)(), float) for n in ("vtem", "omegatem", "wtem", "psitem", "epfy", "epfz", "epdiv", "utendepfd", "utendvtem", "utendwtem")}
    worst = 0.0
    for n in out["direct"]:
        e = np.max(np.abs(out["lin"][n] - out["direct"][n])) / np.max(np.abs(out["direct"][n]))
        worst = max(worst, e)
        print("  result %-10s linearised vs direct %.2e" % (n, e))
    print("worst over the ten results: %.2e" % worst)
