"""world_size-2 gloo tests (CPU) of the multi-GPU driver in pytemdiags_amd/sharding.py.

The driver only sequences stages and collectives; here its backend is a CPU stand-in with the
engine.Plan stage interface built on the oracle, so the test checks that ncol-sharded partial
sums + all-reduce + redundant solve reproduce the single-process result, and that time sharding
+ ragged gather reassemble the time axis."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from oracle import tem_oracle as orc          # noqa: E402
from pytemdiags_amd import sharding, synth    # noqa: E402

NE, NLEV, NT, L = 4, 6, 5, 12


class OracleBackend:
    """engine.Plan's stage interface on CPU tensors (test stand-in only)."""

    def __init__(self, lat_local, lat_out, L, plev):
        self.Y0 = orc.ylm0_matrix(lat_local, L)
        self.Y0p = orc.ylm0_matrix(lat_out, L)
        self.plev = np.asarray(plev)
        self.scale = (orc.P0 / (self.plev * 100)) ** orc.k
        self.Ginv = None

    def matrix(self, which):
        assert which == sharding.MAT_GRAM
        return torch.from_numpy(self.Y0.T @ self.Y0)

    def finalize(self, G):
        self.Ginv = np.linalg.inv(G)

    def _f(self, ua, va, ta, wap):
        th = orc.multiply_p(ta, self.scale)
        return [np.asarray(x).reshape(x.shape[0], -1) for x in (ua, va, th, wap)], ua.shape[1:]

    def tem_stage1(self, ua, va, ta, wap):
        X, _ = self._f(ua, va, ta, wap)
        return torch.from_numpy(np.stack([self.Y0.T @ x for x in X]))

    def tem_stage2(self, ua, va, ta, wap, B4):
        X, tr = self._f(ua, va, ta, wap)
        C = [self.Ginv @ b for b in B4.numpy()]
        self.zon = {n: (self.Y0p @ c).reshape((-1,) + tr) for n, c in zip(("ub", "vb", "thetab", "wapb"), C)}
        e = [x - self.Y0 @ c for x, c in zip(X, C)]
        self.e = e
        prods = [e[0] * e[1], e[0] * e[3], e[1] * e[2]]
        self.tr = tr
        return torch.from_numpy(np.stack([self.Y0.T @ p for p in prods]))

    def tem_stage3(self, B3, want_zonal=False):
        for n, b in zip(("upvpb", "upwappb", "vptpb"), B3.numpy()):
            self.zon[n] = (self.Y0p @ (self.Ginv @ b)).reshape((-1,) + self.tr)
        o = orc.TEMOracle.from_zonal_means(self.zon, self.plev)
        res = torch.from_numpy(np.stack([getattr(o, n)() for n in orc.RESULTS]))
        return res, None

    def tracer_stage1(self, q):
        self.qf = np.asarray(q).reshape(q.shape[0], -1)
        return torch.from_numpy(self.Y0.T @ self.qf)

    def tracer_stage2(self, q, va, wap, Bq):
        cq = self.Ginv @ Bq.numpy()
        self.zon_q = {"qb": (self.Y0p @ cq).reshape((-1,) + self.tr)}
        qp = self.qf - self.Y0 @ cq
        return torch.from_numpy(np.stack([self.Y0.T @ (qp * self.e[1]), self.Y0.T @ (qp * self.e[3])]))

    def tracer_stage3(self, Bq2, want_zonal=False):
        for n, b in zip(("qpvpb", "qpwappb"), Bq2.numpy()):
            self.zon_q[n] = (self.Y0p @ (self.Ginv @ b)).reshape((-1,) + self.tr)
        o = orc.TEMOracle.from_zonal_means(self.zon, self.plev)
        o.q = [np.empty(0)]
        o.qb, o.qpvpb, o.qpwappb = [self.zon_q["qb"]], [self.zon_q["qpvpb"]], [self.zon_q["qpwappb"]]
        o._derivatives()
        return torch.from_numpy(np.stack([getattr(o, n)(0) for n in orc.TRACER_RESULTS])), None

    def tem_run(self, ua, va, ta, wap, want_zonal=False):
        B4 = self.tem_stage1(ua, va, ta, wap)
        return self.tem_stage3(self.tem_stage2(ua, va, ta, wap, B4), want_zonal)


class OnePassOracleBackend(OracleBackend):
    """The same stand-in with the one-pass stage interface of engine.Plan: stage 1 leaves state (the
    engine: per-class sums) that the from-sums stages consume without being handed the fields again."""
    one_pass = True
    tracer_one_pass = True

    def tem_stage1(self, ua, va, ta, wap):
        self._kept = (ua, va, ta, wap)
        return super().tem_stage1(ua, va, ta, wap)

    def tem_stage2(self, *a):
        raise AssertionError("the driver must use tem_stage2_from_sums on a one-pass backend")

    def tem_stage2_from_sums(self, B4):
        return OracleBackend.tem_stage2(self, *self._kept, B4)

    def tracer_stage1(self, q):
        raise AssertionError("the driver must use tracer_stage1_sums on a one-pass backend")

    def tracer_stage1_sums(self, q, va, wap):
        self._kept_q = (q, va, wap)
        return OracleBackend.tracer_stage1(self, q)

    def tracer_stage2_from_sums(self, Bq):
        return OracleBackend.tracer_stage2(self, *self._kept_q, Bq)


class SingleSweepOracleBackend(OracleBackend):
    """Stand-in with the three-step interface of the single sweep (engine.Plan.tem_os_prepass / _sweep / _tail and
    the tracer's three; include/temx.h): the sweep hands over projections of the fields (minus a low-degree
    reference fitted to a subsample) up to degree 2L and of their products up to degree L, written as time slices;
    the tail gets the eddy-product sums of ITS snapshots from them by the Legendre product linearisation on
    Gauss-Legendre nodes -- the algebra of os_contract_kernel in numpy, which is what makes a time-sliced tail
    possible: nothing in it needs a second look at the rows."""
    single_sweep = True
    device = torch.device("cpu")

    def __init__(self, lat_local, lat_out, L, plev, nt):
        super().__init__(lat_local, lat_out, L, plev)
        self.L, self.K, self.KX, self.KR = L, L + 1, 2 * L + 1, min(16, L + 1)
        self.nlev, self.nt, self.D = len(plev), nt, len(plev) * nt
        self.os_rows = 4 * self.KX + 3 * self.K
        self.Yx = orc.ylm0_matrix_recurrence(lat_local, 2 * L)
        self.sub = np.arange(0, lat_local.size, 3)                    # this rank's share of the reference subsample
        xg, wg = np.polynomial.legendre.leggauss(2 * L + 2)
        self.Yg = orc.ylm0_matrix_recurrence(np.rad2deg(np.arcsin(xg)), 2 * L)
        self.wg = 2 * np.pi * wg
        self.Gx = self.Gs = None

    def matrix(self, which):
        if which == sharding.MAT_GX:
            return torch.from_numpy(self.Y0.T @ self.Yx)
        if which == sharding.MAT_GSUB:
            Ys = self.Y0[self.sub][:, :self.KR]
            return torch.from_numpy(Ys.T @ Ys)
        return super().matrix(which)

    def finalize(self, G):
        self.G = np.array(G)
        super().finalize(G)

    def configure(self, **kw):
        pass

    def set_tem(self, nlev, nt, p_pa, p0=None):
        assert (nlev, nt) == (self.nlev, self.nt)

    def set_os_matrices(self, Gx, Gs):
        self.Gx, self.Gs = np.array(Gx), np.array(Gs)

    def _slices(self, rows2d, W):      # [rows][D] -> [W][rows * nlev * ntmax], slice w packed as [rows][nlev][ntw]
        a = rows2d.reshape(rows2d.shape[0], self.nlev, self.nt)
        ntmax = -(-self.nt // W)
        out = np.zeros((W, rows2d.shape[0] * self.nlev * ntmax))
        for w in range(W):
            t0, t1 = sharding.shard_bounds(self.nt, W, w)
            out[w, : a.shape[0] * self.nlev * (t1 - t0)] = a[:, :, t0:t1].reshape(-1)
        return torch.from_numpy(out if W > 1 else a.copy())

    def _prepass(self, X):
        Ys = self.Y0[self.sub][:, :self.KR]
        return np.stack([Ys.T @ x[self.sub] for x in X])

    def tem_os_prepass(self, ua, va, ta, wap, out=None):
        return torch.from_numpy(self._prepass(self._f(ua, va, ta, wap)[0]))

    def _sweep(self, X, As, pairs):
        assert self.Gs is not None, "set_os_matrices (the all-reduced matrices) must precede the sweep"
        rho = [np.linalg.solve(self.Gs, a) for a in np.asarray(As)]            # [KR][D] per field
        Xs = [x - self.Y0[:, :self.KR] @ r for x, r in zip(X, rho)]            # shifted fields: same eddies
        return rho, Xs, [self.Y0.T @ (Xs[a] * Xs[b]) for a, b in pairs]

    def tem_os_sweep(self, ua, va, ta, wap, As, nslices=1, out=None):
        X, _ = self._f(ua, va, ta, wap)
        self.rho, Xs, P = self._sweep(X, As.numpy().reshape(4, self.KR, self.D), [(0, 1), (0, 3), (1, 2)])
        self._kept_vw = (Xs[1], Xs[3])
        return self._slices(np.concatenate([self.Yx.T @ x for x in Xs] + P), nslices)

    def _lin(self, A, al, B, be, P):
        """sum_i Y_l (a - abar)(b - bbar) from A, B (degree-2L projections), the coefficients of the zonal means and P"""
        At, Bt = self.Yg @ A, self.Yg @ B                                      # synthesis at the nodes
        ab, bb = self.Yg[:, :self.K] @ al, self.Yg[:, :self.K] @ be
        cross = self.Yg[:, :self.K].T @ (self.wg[:, None] * (bb * At + ab * Bt))
        c = self.Yg.T @ (self.wg[:, None] * (ab * bb))
        return P - cross + self.Gx @ c

    def _cols(self, t0, nts):          # columns (lev, t0 + t) of the whole run
        return (np.arange(self.nlev)[:, None] * self.nt + t0 + np.arange(nts)[None, :]).reshape(-1)

    def tem_os_tail(self, mine, t0, nts, want_zonal=False):
        m = mine.numpy().reshape(-1)[: self.os_rows * self.nlev * nts].reshape(self.os_rows, self.nlev * nts)
        A = [m[f * self.KX:(f + 1) * self.KX] for f in range(4)]
        P = [m[4 * self.KX + k * self.K: 4 * self.KX + (k + 1) * self.K] for k in range(3)]
        al = [self.Ginv @ a[:self.K] for a in A]
        cols = self._cols(t0, nts)
        tr = (self.nlev, nts)
        # zonal means of the ORIGINAL fields: the reference comes back exactly
        C = [a + np.pad(r[:, cols], ((0, self.K - self.KR), (0, 0))) for a, r in zip(al, self.rho)]
        self.zon = {n: (self.Y0p @ c).reshape((-1,) + tr) for n, c in zip(("ub", "vb", "thetab", "wapb"), C)}
        for n, (a, b), p in zip(("upvpb", "upwappb", "vptpb"), [(0, 1), (0, 3), (1, 2)], P):
            F = self._lin(A[a], al[a], A[b], al[b], p)
            self.zon[n] = (self.Y0p @ (self.Ginv @ F)).reshape((-1,) + tr)
        self._tail = (t0, nts, A, al)
        o = orc.TEMOracle.from_zonal_means(self.zon, self.plev)
        return torch.from_numpy(np.stack([getattr(o, n)() for n in orc.RESULTS])), None

    def tracers_os_prepass(self, qs, va, wap, out=None):
        return torch.from_numpy(np.stack(self._prepass([np.asarray(q).reshape(q.shape[0], -1) for q in qs])))

    def tracers_os_sweep(self, qs, va, wap, Asq, nslices=1, out=None):
        qf = [np.asarray(q).reshape(q.shape[0], -1) for q in qs]
        self.rho_q = [np.linalg.solve(self.Gs, a) for a in Asq.numpy()]
        qsft = [x - self.Y0[:, :self.KR] @ r for x, r in zip(qf, self.rho_q)]
        vs, ws = self._kept_vw                                                # (the same v, omega as the TEM run)
        prods = [self.Y0.T @ (x * y) for x in qsft for y in (vs, ws)]          # q1 v, q1 omega, q2 v, q2 omega
        return self._slices(np.concatenate([self.Yx.T @ x for x in qsft] + prods), nslices)

    def tracers_os_tail(self, nq, mine, nts, want_zonal=False):
        t0, nts0, A, al = self._tail
        assert nts == nts0
        rows = nq * (self.KX + 2 * self.K)
        m = mine.numpy().reshape(-1)[: rows * self.nlev * nts].reshape(rows, self.nlev * nts)
        tr = (self.nlev, nts)
        outs = []
        for i in range(nq):
            Aq = m[i * self.KX:(i + 1) * self.KX]
            P = [m[nq * self.KX + (2 * i + k) * self.K: nq * self.KX + (2 * i + k + 1) * self.K] for k in range(2)]
            alq = self.Ginv @ Aq[:self.K]
            cq = alq + np.pad(self.rho_q[i][:, self._cols(t0, nts)], ((0, self.K - self.KR), (0, 0)))
            zq = {"qb": (self.Y0p @ cq).reshape((-1,) + tr)}
            for n, fld, p in zip(("qpvpb", "qpwappb"), (1, 3), P):
                zq[n] = (self.Y0p @ (self.Ginv @ self._lin(Aq, alq, A[fld], al[fld], p))).reshape((-1,) + tr)
            o = orc.TEMOracle.from_zonal_means(self.zon, self.plev)
            o.q = [np.empty(0)]
            o.qb, o.qpvpb, o.qpwappb = [zq["qb"]], [zq["qpvpb"]], [zq["qpwappb"]]
            o._derivatives()
            outs.append((torch.from_numpy(np.stack([getattr(o, n)(0) for n in orc.TRACER_RESULTS])), None))
        return outs


_LON = synth.cubed_sphere_gll(NE)[1]


def _data():
    lat, lon = synth.cubed_sphere_gll(NE)
    plev = synth.pressure_levels(NLEV)
    f = synth.analytic_fields(lat, lon, plev, NT, seed=3)
    return lat, plev, f


def _worker(rank, world, port, mode, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        lat, plev, f = _data()
        lat_zm = orc.zm_latitudes(1)
        if mode in ("ncol", "ncol-one-pass", "ncol-sliced"):
            i0, i1 = sharding.shard_bounds(lat.size, world, rank)
            if mode == "ncol-sliced":
                be = SingleSweepOracleBackend(lat[i0:i1], lat_zm, L, plev, NT)
                runner = sharding.NcolShardedTEM(be, tail="sliced")
                runner.set_tem(NLEV, NT, plev * 100)
                assert runner.sliced
            else:
                be = (OracleBackend if mode == "ncol" else OnePassOracleBackend)(lat[i0:i1], lat_zm, L, plev)
                runner = sharding.NcolShardedTEM(be)
            res, _ = runner.run(*[x[i0:i1] for x in f])
            q = synth.analytic_tracer(lat, _LON, plev, NT)
            tres, _ = runner.run_tracer(q[i0:i1], f[1][i0:i1], f[3][i0:i1])
            if mode == "ncol-sliced":      # the list form: a pair in one sweep must repeat the single run
                q2 = synth.analytic_tracer(lat, _LON, plev, NT, which=1)
                both = runner.run_tracers([q[i0:i1], q2[i0:i1]], f[1][i0:i1], f[3][i0:i1])
                assert len(both) == 2 and float((both[0][0] - tres).abs().max()) <= 1e-12 * float(tres.abs().max())
            if mode == "ncol-sliced":      # every rank holds its own snapshots: ragged 3 + 2 of NT = 5
                t0, t1 = runner.my_snapshots()
                assert (t0, t1) == sharding.shard_bounds(NT, world, rank) and res.shape[-1] == t1 - t0
                res, tres = sharding.gather_time(res), sharding.gather_time(tres)
            res = torch.cat([res, tres])
        else:
            be = OracleBackend(lat, lat_zm, L, plev)
            be.finalize((be.Y0.T @ be.Y0))
            runner = sharding.TimeShardedTEM(be, NT)
            t0, t1 = runner.t0, runner.t1
            res, _ = runner.run(*[np.ascontiguousarray(x[:, :, t0:t1]) for x in f])
            res = sharding.gather_time(res)
        if rank == 0:
            ret.put(res.numpy())
    finally:
        dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("mode", ["ncol", "ncol-one-pass", "ncol-sliced", "time"])
def test_sharded_pipeline_world2_gloo(mode):
    lat, plev, f = _data()
    q = synth.analytic_tracer(lat, _LON, plev, NT)
    ref = orc.TEMOracle(*f, lat, plev, L=L, mode="factorised", q=q)
    ctx = mp.get_context("spawn")
    ret = ctx.Queue()
    port = _free_port()
    world = int(os.environ.get("TEMX_GLOO_WORLD", "2"))      # 2 in the suite; more ranks (<= NT) to rehearse by hand
    procs = [ctx.Process(target=_worker, args=(r, world, port, mode, ret)) for r in range(world)]
    for p in procs:
        p.start()
    got = ret.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert got.shape == ((16 if mode.startswith("ncol") else 10), 180, NLEV, NT)
    for i, n in enumerate(orc.RESULTS):
        r = getattr(ref, n)()
        err = np.max(np.abs(got[i] - r)) / np.max(np.abs(r))
        assert err <= 1e-10, (mode, n, err)
    if mode.startswith("ncol"):      # tracer TEM through the sharded driver (two more all-reduces)
        for i, n in enumerate(orc.TRACER_RESULTS):
            r = getattr(ref, n)(0)
            err = np.max(np.abs(got[10 + i] - r)) / np.max(np.abs(r))
            assert err <= 1e-10, (mode, n, err)


def test_world1_is_a_no_op():
    t = torch.ones(3)
    assert sharding.allreduce_sum_(t) is t and float(t.sum()) == 3.0
    assert sharding.gather_time(t) is t


def test_shard_bounds_in_multiples():
    """Time blocks of whole cache lines per row (sharding.aligned_snapshots): every rank but possibly the last gets a
    multiple, the blocks tile [0, n), sizes differ by at most one multiple."""
    assert sharding.aligned_snapshots(72, 8) == 2 and sharding.aligned_snapshots(72, 4) == 4
    assert sharding.aligned_snapshots(128, 4) == 1 and sharding.aligned_snapshots(30, 8) == 8
    b = [sharding.shard_bounds(730, 8, r, 2) for r in range(8)]
    assert [hi - lo for lo, hi in b] == [92, 92, 92, 92, 92, 90, 90, 90]
    for n, w, m in ((730, 8, 2), (731, 8, 2), (30, 8, 4), (7, 8, 2), (100, 3, 8), (64, 4, 16)):
        b = [sharding.shard_bounds(n, w, r, m) for r in range(w)]
        assert b[0][0] == 0 and b[-1][1] == n and all(b[r][1] == b[r + 1][0] for r in range(w - 1)), (n, w, m, b)
        if n // m >= w:
            assert all((hi - lo) % m == 0 for lo, hi in b[:-1]), (n, w, m, b)
            assert max(hi - lo for lo, hi in b) - min(hi - lo for lo, hi in b[:-1]) <= m, (n, w, m, b)
    assert [sharding.shard_bounds(5, 2, r) for r in range(2)] == [(0, 3), (3, 5)]      # the default is unchanged
