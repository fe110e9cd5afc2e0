"""The single sweep in three steps and its time-sliced tail (include/temx.h: temx_tem_os_prepass / _sweep / _tail,
temx_tracers_os_*, temx_tem_tail_from_sums, temx_time_slices) -- the form an ncol-sharded job runs: the zonal sums are
linear in the rows (sph_zonal_mean.py:251) and everything after them acts along latitude and pressure only
(tem_diagnostics.py:574-797), so the ranks exchange the sums by a reduce-scatter over time and each finishes its own
snapshots.  Here the collectives are written out as explicit sums on one GPU; tests/test_gpu_multiproc.py runs the
same flow through torch.distributed with two processes."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _skip_if_forced_elsewhere():
    if any(os.environ.get(k) == "1" for k in ("TEMX_NO_SYM", "TEMX_NO_CLS", "TEMX_TWO_PASS", "TEMX_NO_QR")) or \
            os.environ.get("TEMX_SINGLE_SWEEP") == "0":
        pytest.skip("the environment forces another form of the sweeps")


def _case(ne, nlev, nt, dtype=np.float64, seed=17):
    from pytemdiags_amd import synth
    lat, lon = synth.cubed_sphere_gll(ne)
    plev = synth.pressure_levels(nlev)
    f = synth.analytic_fields(lat, lon, plev, nt, seed=seed, dtype=dtype)
    q = synth.analytic_tracer(lat, lon, plev, nt).astype(dtype)
    lat_zm = (np.arange(-90, 91, 1.0)[1:] + np.arange(-90, 91, 1.0)[:-1]) / 2
    return lat, lon, plev, f, q, lat_zm


def _dev(x):
    return torch.as_tensor(np.ascontiguousarray(x), device="cuda:0")


def _relerr(a, b):
    return float((a - b).abs().max()) / float(b.abs().max())


@pytest.mark.parametrize("ne,nlev,nt,dtype,L,W", [
    (12, 10, 7, np.float64, 50, 3),     # ragged slices 3 + 2 + 2
    (10, 9, 8, np.float64, 28, 8),      # one snapshot per slice, TBS = 4
    (12, 16, 5, np.float32, 50, 2),     # fp32 inputs (two waves per SIMD)
])
def test_three_steps_and_time_slices_equal_the_whole_run(ne, nlev, nt, dtype, L, W):
    from pytemdiags_amd import _lib, engine, sharding
    _skip_if_forced_elsewhere()
    lat, lon, plev, f, q, lat_zm = _case(ne, nlev, nt, dtype)
    d = [_dev(x) for x in f]
    dq = _dev(q)
    plan = engine.Plan(lat, lat_zm, L, form="single-sweep")          # path selection through the ABI, not the environment
    plan.set_tem(nlev, nt, plev * 100)
    assert plan.single_sweep and plan.option(_lib.OPT_FORM) == _lib.FORM_SINGLE_SWEEP
    res, zon = plan.tem_run(*d, want_zonal=True)
    tres, tzon = plan.tracer_run(dq, d[1], d[3], want_zonal=True)
    assert not plan.status()
    # the three steps with one slice are temx_tem_run
    As = plan.tem_os_prepass(*d)
    assert As.shape == (4, plan.KR, plan.D)
    proj = plan.tem_os_sweep(*d, As)
    assert proj.shape == (plan.os_rows, nlev, nt)
    r1, z1 = plan.tem_os_tail(proj, 0, nt, want_zonal=True)
    assert torch.equal(r1, res) and torch.equal(z1, zon)
    # W time slices: slice w holds the snapshots shard_bounds(nt, W, w) of every row, packed
    projW = plan.tem_os_sweep(*d, As, nslices=W)
    ntmax = -(-nt // W)
    assert projW.shape == (W, plan.os_rows * nlev * ntmax)
    Asq = plan.tracers_os_prepass([dq], d[1], d[3])
    projq = plan.tracers_os_sweep([dq], d[1], d[3], Asq, nslices=W)
    for w in range(W):
        t0, t1 = sharding.shard_bounds(nt, W, w)
        mine = projW[w][: plan.os_rows * nlev * (t1 - t0)].reshape(plan.os_rows, nlev, t1 - t0)
        assert torch.equal(mine, proj[:, :, t0:t1])                  # the reduction wrote the slices itself
        rw, zw = plan.tem_os_tail(projW[w], t0, t1 - t0, want_zonal=True)
        # a column of the tail does not see its neighbours: the slice reproduces the whole run's bits
        assert torch.equal(rw, res[..., t0:t1]) and torch.equal(zw, zon[..., t0:t1])
        with pytest.raises(_lib.TemxError):                          # the plan now describes a slice: whole-run stage 3 refuses
            plan.tem_stage3(torch.zeros((3, plan.K, plan.D), dtype=torch.float64, device="cuda:0"))
        (tw, tzw), = plan.tracers_os_tail(1, projq[w], t1 - t0, want_zonal=True)
        assert torch.equal(tw, tres[..., t0:t1]) and torch.equal(tzw, tzon[..., t0:t1])
    # back to the whole run: the staged class-sum entry points work again and agree
    B4 = plan.tem_stage1(*d)
    B3 = plan.tem_stage2_from_sums(B4)
    r3, _ = plan.tem_stage3(B3)
    tol = 1e-11 if dtype == np.float64 else 1e-5
    for i, n in enumerate(_lib.RESULT_NAMES):
        assert _relerr(r3[i], res[i]) <= tol, n
    # the time-sliced tail for raw sums of any form: temx_time_slices + temx_tem_tail_from_sums
    B4W, B3W = plan.time_slices(B4, W), plan.time_slices(B3, W)
    for w in range(W):
        t0, t1 = sharding.shard_bounds(nt, W, w)
        rw, _ = plan.tem_tail_from_sums(B4W[w], B3W[w], t0, t1 - t0)
        assert torch.equal(rw, r3[..., t0:t1])
    assert not plan.status()
    plan.close()


@pytest.mark.parametrize("ne,nlev,nt,W", [(16, 12, 5, 2), (12, 8, 9, 4)])
def test_ncol_shards_with_time_sliced_tail_emulated_on_one_gpu(ne, nlev, nt, W):
    """W ranks' plans on one GPU, the collectives of sharding.NcolShardedTEM written out as sums: Gram matrices, the
    two matrices of the single sweep that sum over the rows (TEMX_MAT_GX, TEMX_MAT_GSUB), the pre-pass sums
    (all-reduce) and the projections (reduce-scatter over time).  Against the unsharded run, TEM and tracer."""
    from pytemdiags_amd import _lib, engine, sharding
    _skip_if_forced_elsewhere()
    L = 50
    lat, lon, plev, f, q, lat_zm = _case(ne, nlev, nt)
    whole = engine.Plan(lat, lat_zm, L, form="single-sweep")
    whole.set_tem(nlev, nt, plev * 100)
    d = [_dev(x) for x in f]
    ref, _ = whole.tem_run(*d)
    tref, _ = whole.tracer_run(_dev(q), d[1], d[3])
    whole.close()
    shards = sharding.symmetric_ncol_shards(lat, W)
    plans = [engine.Plan(lat[m], lat_zm, L, defer_finalize=True, form="single-sweep") for m in shards]
    G = sum(p.matrix(_lib.MAT_GRAM) for p in plans).cpu().numpy()
    for p in plans:
        p.finalize(G)
    G2 = sum(p.matrix(_lib.MAT_GRAM2) for p in plans).cpu().numpy()
    for p in plans:
        p.refine(G2)
        p.configure(os_subsample=12)
        p.set_tem(nlev, nt, plev * 100)
        assert p.single_sweep
    Gx = sum(p.matrix(_lib.MAT_GX) for p in plans).cpu().numpy()
    Gs = sum(p.matrix(_lib.MAT_GSUB) for p in plans).cpu().numpy()
    for p in plans:
        p.set_os_matrices(Gx, Gs)
    loc = [[_dev(x[m]) for x in f] for m in shards]
    locq = [_dev(q[m]) for m in shards]
    As = sum(p.tem_os_prepass(*x) for p, x in zip(plans, loc))
    proj = sum(p.tem_os_sweep(*x, As, nslices=W) for p, x in zip(plans, loc))
    Asq = sum(p.tracers_os_prepass([xq], x[1], x[3]) for p, x, xq in zip(plans, loc, locq))
    projq = None
    outs, touts = [], []
    for w, p in enumerate(plans):
        t0, t1 = sharding.shard_bounds(nt, W, w)
        r, _ = p.tem_os_tail(proj[w], t0, t1 - t0)
        outs.append(r)
    # (the tracer's sweep needs its plan's references of v and omega: those of the latest tem_os_sweep, still in place)
    projq = sum(p.tracers_os_sweep([xq], x[1], x[3], Asq, nslices=W) for p, x, xq in zip(plans, loc, locq))
    for w, p in enumerate(plans):
        t0, t1 = sharding.shard_bounds(nt, W, w)
        (t, _), = p.tracers_os_tail(1, projq[w], t1 - t0)
        touts.append(t)
    got, tgot = torch.cat(outs, dim=-1), torch.cat(touts, dim=-1)
    for p in plans:
        assert not p.status()
        p.close()
    for i, n in enumerate(_lib.RESULT_NAMES):
        assert _relerr(got[i], ref[i]) <= 1e-11, n
    for i, n in enumerate(_lib.TRACER_RESULT_NAMES):
        assert _relerr(tgot[i], tref[i]) <= 1e-11, n


def test_nan_input_on_the_single_sweep_path():
    """SURVEY Q14 (sph_zonal_mean.py:219-221: NaN anywhere in an input raises) on the path the headline runs."""
    from pytemdiags_amd import engine
    _skip_if_forced_elsewhere()
    lat, lon, plev, f, q, lat_zm = _case(12, 16, 4)
    plan = engine.Plan(lat, lat_zm, 50, form="single-sweep")
    plan.set_tem(16, 4, plev * 100)
    assert plan.single_sweep
    d = [_dev(x) for x in f]
    plan.tem_run(*d)
    assert not plan.status()
    for fld, col in ((0, 5), (2, lat.size - 1), (3, lat.size // 2)):
        bad = [x.clone() for x in d]
        bad[fld][col, 3, 1] = float("nan")
        plan.tem_run(*bad)
        assert plan.status(), (fld, col)
        plan.tem_run(*d)
        assert not plan.status()
    plan.close()


def test_configure_selects_the_forms():
    """the path is a property of the plan (temx_plan_configure), two plans in one process differ."""
    from pytemdiags_amd import _lib, engine
    _skip_if_forced_elsewhere()
    if os.environ.get("TEMX_ONE_PASS") == "1":
        pytest.skip("TEMX_ONE_PASS=1 overrides the two-pass option")
    lat, lon, plev, f, q, lat_zm = _case(12, 16, 4)
    d = [_dev(x) for x in f]
    forms = {"two-pass": _lib.FORM_TWO_PASS, "class-sums": _lib.FORM_CLASS_SUMS, "single-sweep": _lib.FORM_SINGLE_SWEEP}
    plans = {k: engine.Plan(lat, lat_zm, 50, form=k) for k in forms}
    res = {}
    for k, p in plans.items():
        p.set_tem(16, 4, plev * 100)
        assert p.option(_lib.OPT_FORM) == forms[k], k
        assert p.single_sweep == (k == "single-sweep") and p.one_pass == (k != "two-pass")
        res[k] = p.tem_run(*d)[0]
    for k in ("class-sums", "single-sweep"):
        for i, n in enumerate(_lib.RESULT_NAMES):
            assert _relerr(res[k][i], res["two-pass"][i]) <= 1e-11, (k, n)
    p = plans["single-sweep"]
    p.configure(os_map="tile")
    with pytest.raises(_lib.TemxError):        # configure asks for temx_plan_set_tem again
        p.tem_run(*d)
    p.set_tem(16, 4, plev * 100)
    assert p.option(_lib.OPT_OS_MAP) == 1
    rt = p.tem_run(*d)[0]
    for i, n in enumerate(_lib.RESULT_NAMES):
        assert _relerr(rt[i], res["single-sweep"][i]) <= 1e-11, n
    for p in plans.values():
        p.close()


@pytest.mark.parametrize("ne,nlev,nt,dtype,L", [(12, 10, 7, np.float64, 50), (10, 9, 8, np.float64, 28), (10, 16, 5, np.float64, 12),
                                                (12, 16, 5, np.float32, 40)])
def test_contraction_on_the_matrix_cores_equals_the_lds_form(ne, nlev, nt, dtype, L):
    """The contraction after the single sweep (kernels_osc.hpp: two MFMA kernels) against its round-3 form
    (os_contract_kernel, matrices staged in LDS, TEMX_OPT_OS_CONTRACT = 1): the same algebra, TEM and tracer."""
    from pytemdiags_amd import _lib, engine
    _skip_if_forced_elsewhere()
    if os.environ.get("TEMX_OS_CONTRACT"):
        pytest.skip("TEMX_OS_CONTRACT overrides the option")
    lat, lon, plev, f, q, lat_zm = _case(ne, nlev, nt, dtype)
    d = [_dev(x) for x in f]
    dq = _dev(q)
    out = {}
    for form in ("mfma", "lds"):
        plan = engine.Plan(lat, lat_zm, L, form="single-sweep")
        plan.configure(os_contract=form)
        plan.set_tem(nlev, nt, plev * 100)
        assert plan.single_sweep and plan.option(_lib.OPT_OS_CONTRACT) == (1 if form == "lds" else 0)
        r, z = plan.tem_run(*d, want_zonal=True)
        t, _ = plan.tracer_run(dq, d[1], d[3])
        assert not plan.status()
        out[form] = (r, z, t)
        plan.close()
    tol = 1e-12 if dtype == np.float64 else 1e-6
    for a, b, names in zip(out["mfma"], out["lds"], (_lib.RESULT_NAMES, _lib.ZONAL_NAMES, _lib.TRACER_RESULT_NAMES)):
        for i, n in enumerate(names):
            assert _relerr(a[i], b[i]) <= tol, n


@pytest.mark.parametrize("ne,nlev,nt,dtype,L,ntr", [
    (12, 10, 7, np.float64, 50, 2),
    (12, 16, 5, np.float32, 50, 3),     # fp32 inputs (two waves per SIMD); an odd tracer left over
    (10, 9, 8, np.float64, 28, 5),      # TBS = 4
])
def test_two_tracers_per_sweep(ne, nlev, nt, dtype, L, ntr):
    """The reference takes a LIST of tracers (tem_diagnostics.py:281-301, 532-538, 560-570).  After a single-sweep TEM
    run they are swept in pairs -- (q1, q2, v, omega) read once -- and must give what one tracer at a time gives
    (temx_tracer_run, itself held to the oracle and the reference's goldens elsewhere); also sliced in time."""
    from pytemdiags_amd import _lib, engine, sharding, synth
    _skip_if_forced_elsewhere()
    if os.environ.get("TEMX_OS_MAP") == "tile" or os.environ.get("TEMX_OS_CONTRACT") == "lds":
        pytest.skip("two tracers per sweep need the row-map sweep and the contraction on the matrix cores")
    lat, lon, plev, f, q, lat_zm = _case(ne, nlev, nt, dtype)
    qs = [_dev(synth.analytic_tracer(lat, lon, plev, nt, which=i).astype(dtype)) for i in range(ntr)]
    d = [_dev(x) for x in f]
    plan = engine.Plan(lat, lat_zm, L, form="single-sweep")
    plan.set_tem(nlev, nt, plev * 100)
    res, _ = plan.tem_run(*d)
    single = [plan.tracer_run(x, d[1], d[3], want_zonal=True) for x in qs]
    paired = plan.tracers_run(qs, d[1], d[3], want_zonal=True)
    assert len(paired) == ntr and not plan.status()
    tol = 1e-12 if dtype == np.float64 else 1e-6
    for i in range(ntr):
        for a, b, names in zip(paired[i], single[i], (_lib.TRACER_RESULT_NAMES, _lib.TRACER_ZONAL_NAMES)):
            for k, n in enumerate(names):
                assert _relerr(a[k], b[k]) <= tol, (i, n)
    again = plan.tracers_run(qs, d[1], d[3], want_zonal=True)
    assert all(torch.equal(x[0], y[0]) for x, y in zip(paired, again))      # fixed-order reductions
    # the pair in three steps, its tail on time slices
    W = 2
    Asq = plan.tracers_os_prepass(qs[:2], d[1], d[3])
    assert Asq.shape == (2, plan.KR, plan.D)
    projq = plan.tracers_os_sweep(qs[:2], d[1], d[3], Asq, nslices=W)
    As = plan.tem_os_prepass(*d)
    proj = plan.tem_os_sweep(*d, As, nslices=W)
    for w in range(W):
        t0, t1 = sharding.shard_bounds(nt, W, w)
        plan.tem_os_tail(proj[w], t0, t1 - t0)
        outs = plan.tracers_os_tail(2, projq[w], t1 - t0, want_zonal=True)
        for i in range(2):
            assert torch.equal(outs[i][0], paired[i][0][..., t0:t1]) and torch.equal(outs[i][1], paired[i][1][..., t0:t1])
    assert not plan.status()
    plan.close()


def test_reference_two_tracer_golden_through_the_pair_sweep():
    """The reference's own run with a LIST of two tracers (tests/golden/tracer_ne4_10x2_f64.npz, made by
    tools/make_goldens.py from the unmodified reference) through the pair sweep.  The golden has D = 10 x 2 columns,
    fewer than the one-pass forms take; every step of the pipeline is independent per snapshot
    (tem_diagnostics.py:510-797 never mix times), so the two snapshots are repeated four times and every repetition
    must reproduce the reference's numbers."""
    from conftest import GOLDEN, fieldnorm_err
    from pytemdiags_amd import _lib, engine
    _skip_if_forced_elsewhere()
    if os.environ.get("TEMX_OS_MAP") == "tile" or os.environ.get("TEMX_OS_CONTRACT") == "lds":
        pytest.skip("two tracers per sweep need the row-map sweep and the contraction on the matrix cores")
    g = np.load(os.path.join(GOLDEN, "tracer_ne4_10x2_f64.npz"), allow_pickle=True)
    assert int(g["ntrac"]) == 2
    rep = 4
    tile = lambda x: _dev(np.tile(np.asarray(x), (1, 1, rep)))                # noqa: E731  [ncol][plev][time x rep]
    d = [tile(g[k]) for k in ("ua", "va", "ta", "wap")]
    qs = [tile(g["q0"]), tile(g["q1"])]
    nlev, nt = g["ua"].shape[1], g["ua"].shape[2] * rep
    lat_zm = (np.arange(-90, 91, 1.0)[1:] + np.arange(-90, 91, 1.0)[:-1]) / 2
    plan = engine.Plan(g["lat"], lat_zm, 50, form="single-sweep")
    plan.set_tem(nlev, nt, np.asarray(g["plev"]) * 100)
    assert plan.single_sweep
    res, _ = plan.tem_run(*d)
    out = plan.tracers_run(qs, d[1], d[3], want_zonal=True)
    assert not plan.status()
    res = res.cpu().numpy()
    for r in range(rep):
        sl = slice(2 * r, 2 * r + 2)
        for i, n in enumerate(_lib.RESULT_NAMES):
            assert fieldnorm_err(res[i][..., sl], g["res_" + n]) <= 1e-10, (r, n)
        for qi in range(2):
            tres, tzon = out[qi][0].cpu().numpy(), out[qi][1].cpu().numpy()
            for k, n in enumerate(_lib.TRACER_RESULT_NAMES):
                assert fieldnorm_err(tres[k][..., sl], g["q%d_res_%s" % (qi, n)]) <= 1e-10, (r, qi, n)
            for k, n in enumerate(_lib.TRACER_ZONAL_NAMES):
                assert fieldnorm_err(tzon[k][..., sl], np.asarray(g["q%d_%s" % (qi, n)], np.float64)) <= 1e-10, (r, qi, n)
    plan.close()
