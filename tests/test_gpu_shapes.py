"""GPU parity at shapes that stress the decomposition (ragged ncol / D, several d-tiles per
workgroup configurations, fp32 inputs) against the CPU oracle on the same seeded inputs, plus
size-independent properties at a BASELINE-sized grid."""
import numpy as np
import pytest

from conftest import fieldnorm_err

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def run_case(ne, nlev, nt, L=50, dtype=np.float64, seed=0):
    from oracle import tem_oracle as orc
    from pytemdiags_amd import _lib, engine, synth
    lat, lon = synth.cubed_sphere_gll(ne)
    plev = synth.pressure_levels(nlev)
    f = synth.analytic_fields(lat, lon, plev, nt, seed=seed, dtype=dtype)
    ref = orc.TEMOracle(*f, lat, plev, L=L, mode="factorised")
    plan = engine.Plan(lat, ref.lat, L)
    plan.set_tem(nlev, nt, plev * 100)
    dev = [torch.as_tensor(x, device="cuda:0") for x in f]
    res, zon = plan.tem_run(*dev, want_zonal=True)
    assert not plan.status()
    res, zon = res.cpu().numpy(), zon.cpu().numpy()
    tol = 1e-10 if dtype == np.float64 else 2e-5
    for i, n in enumerate(_lib.RESULT_NAMES):
        e = fieldnorm_err(res[i], getattr(ref, n)())
        assert e <= tol, (ne, nlev, nt, n, e)
    for i, n in enumerate(_lib.ZONAL_NAMES):
        e = fieldnorm_err(zon[i], getattr(ref, n))
        assert e <= tol, (ne, nlev, nt, n, e)
    plan.close()


@pytest.mark.parametrize("ne,nlev,nt,L", [
    (4, 7, 3, 50),      # D = 21: two d-tiles, ragged
    (4, 2, 1, 50),      # D = 2: a single partial d-tile (nlev = 2 is the minimum np.gradient takes)
    (8, 72, 1, 50),     # the ne30x72x1 shape in small: 5 d-tiles -> 1 d-tile per workgroup
    (8, 16, 8, 50),     # D = 128: 8 d-tiles -> 4 per workgroup
    (8, 9, 5, 15),      # TB = 4
    (8, 12, 2, 31),     # TB = 8
    (8, 12, 2, 60),     # TB = 16
    (16, 24, 3, 50),    # D = 72 again with more columns per split
    (16, 10, 3, 80),    # K = 81 > 64: sliced large-L path, 2 slices
    (16, 6, 2, 130),    # K = 131: 3 slices, ragged last slice
    (12, 16, 4, 80),    # K = 81, D = 64: large-L class path (class sums first, 2 slices)
    (16, 20, 5, 130),   # K = 131, D = 100: 3 slices, ragged d-tiles
    (30, 16, 4, 200),   # K = 201: 4 slices
    (8, 128, 8, 50),    # D = 1024, the largest zonal grid of the workgroup-per-latitude epilogue: two 64-level chunks
                        # of its scan with a carry, two snapshots per wave
    (8, 130, 7, 20),    # D = 910: ragged second chunk of the scan (130 levels), seven snapshots on four waves
    (8, 129, 8, 20),    # D = 1032: just past it -- the scan kernel and the one-thread-per-point epilogue
])
def test_pipeline_shapes_fp64(ne, nlev, nt, L):
    run_case(ne, nlev, nt, L)


def test_pipeline_fp32_inputs():
    run_case(8, 30, 4, dtype=np.float32)


def test_properties_at_baseline_grid_size():
    """ne30 (48 602 columns) x 72 x 2: properties that need no CPU reference."""
    from pytemdiags_amd import _lib, engine, synth
    lat, lon = synth.cubed_sphere_gll(30)
    plev = synth.pressure_levels(72)
    lat_zm = (np.arange(-90, 91, 1.0)[1:] + np.arange(-90, 91, 1.0)[:-1]) / 2
    plan = engine.Plan(lat, lat_zm, 50)
    plan.set_tem(72, 2, plev * 100)
    # (1) G^-1 G = I  (the sanity numbers the reference prints, sph_zonal_mean.py:393-398)
    P = (plan.matrix(_lib.MAT_GINV) @ plan.matrix(_lib.MAT_GRAM)).cpu().numpy()
    assert np.max(np.abs(P - np.eye(51))) < 1e-12
    # (2) a zonally symmetric field (a function of latitude only, inside the span of the basis)
    #     has zero eddies and is reproduced by both zonal means
    Y0 = plan.matrix(_lib.MAT_Y0)
    coef = torch.linspace(1.0, 0.1, 51, dtype=torch.float64, device="cuda:0")
    sym = (Y0 @ coef)[:, None, None].expand(-1, 72, 2).contiguous()
    zm_n = plan.zonal_mean(sym, native=True)
    assert float((zm_n - sym).abs().max()) < 1e-11 * float(sym.abs().max())
    Y0p = plan.matrix(_lib.MAT_Y0P)
    zm = plan.zonal_mean(sym)
    assert float((zm - (Y0p @ coef)[:, None, None]).abs().max()) < 1e-11
    # (3) linearity of the operator:  zm(a x + b y) = a zm(x) + b zm(y)
    f = engine.synth_fields(0, lat, lon, plev, 2)
    lhs = plan.zonal_mean(2.5 * f[0] - 0.5 * f[1])
    rhs = 2.5 * plan.zonal_mean(f[0]) - 0.5 * plan.zonal_mean(f[1])
    assert float((lhs - rhs).abs().max()) < 1e-11 * float(rhs.abs().max())
    # (4) idempotence on the native grid: zmn(zmn(x)) = zmn(x)
    z1 = plan.zonal_mean(f[0], native=True)
    z2 = plan.zonal_mean(z1, native=True)
    assert float((z2 - z1).abs().max()) < 1e-11 * float(z1.abs().max())
    # (5) symmetric inputs -> zero eddy fluxes; the pipeline still runs (psi = 0/.. guarded by T)
    res, zon = plan.tem_run(sym, sym, f[2], sym, want_zonal=True)
    assert not plan.status()
    scale = float(sym.abs().max()) ** 2
    # (the one-pass class path gets the product sums by differences of O(|u||v|) sums: rounding
    # noise relative to the field scale, not exact zeros)
    assert float(zon[_lib.ZONAL_NAMES.index("upvpb")].abs().max()) < 1e-15 * scale + 1e-20
    # (6) run-to-run determinism (fixed-order reductions)
    r1, _ = plan.tem_run(*f)
    r2, _ = plan.tem_run(*f)
    assert torch.equal(r1, r2)
    plan.close()


def test_pole_points_and_coarse_zonal_grid_vs_oracle():
    """zm_pole_points=True (cos(lat) ~ 6e-17 at the poles: finite but huge, exactly as numpy gives)
    and zm_dlat=2 through the front end; the oracle implements tem_diagnostics.py:388-396."""
    from oracle import tem_oracle as orc
    from pytemdiags_amd import TEMDiagnostics, synth
    lat, lon = synth.cubed_sphere_gll(8)
    plev = synth.pressure_levels(12)
    f = synth.analytic_fields(lat, lon, plev, 2, seed=4)
    for pole in (False, True):
        ref = orc.TEMOracle(*f, lat, plev, L=30, zm_dlat=2, zm_pole_points=pole, mode="factorised")
        tem = TEMDiagnostics(*f, lat, plev=plev, L=30, zm_dlat=2, zm_pole_points=pole, debug_level=0)
        assert tem.ZM_N == (91 if pole else 90)
        np.testing.assert_array_equal(tem.lat, ref.lat)
        inner = slice(1, -1) if pole else slice(None)     # the two pole rows are 1/cos blow-ups
        for n in orc.RESULTS:
            r, g = getattr(tem, n)(), getattr(ref, n)()
            assert fieldnorm_err(r[inner], g[inner]) <= 1e-10, (pole, n)
            if pole:      # same blow-up as numpy at the poles, compared relatively
                assert np.all(np.isfinite(r) == np.isfinite(g))


def test_pipeline_is_hip_graph_capturable():
    """temx_tem_run is stream ordered and allocation free after set_tem: it captures into a HIP
    graph and the replay is bit-identical to the eager run."""
    from pytemdiags_amd import engine, synth
    lat, lon = synth.cubed_sphere_gll(8)
    plev = synth.pressure_levels(20)
    lat_zm = (np.arange(-90, 91, 1.0)[1:] + np.arange(-90, 91, 1.0)[:-1]) / 2
    plan = engine.Plan(lat, lat_zm, 50)
    plan.set_tem(20, 3, plev * 100)
    f = engine.synth_fields(0, lat, lon, plev, 3)
    out = plan._alloc_results(False)
    plan.tem_run(*f, out=out)
    torch.cuda.synchronize()
    ref = out[0].clone()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        plan.tem_run(*f, out=out)          # warm-up on the capture stream
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            plan.tem_run(*f, out=out)
    out[0].zero_()
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(out[0], ref) and not plan.status()
    plan.close()


@pytest.mark.parametrize("lat", [
    np.array([-61.0, -20.5, 3.0, 17.25, 44.0, 71.5, 88.0]),            # 7 columns, not symmetric
    np.array([-70.0, -35.5, -10.0, 0.0, 10.0, 35.5, 70.0]),            # symmetric with an equator column
    np.array([-80.0, 80.0, -45.0, 45.0, -5.0, 5.0]),                   # symmetric, unordered, no equator
    np.linspace(-85, 85, 19),                                          # 19 columns (one ragged group)
])
def test_tiny_grids(lat):
    """Fewer columns than one chunk, counts that are not multiples of 4, both sweep flavours."""
    from oracle import tem_oracle as orc
    from pytemdiags_amd import _lib, engine
    rng = np.random.default_rng(1)
    N, nlev, nt, L = lat.size, 5, 3, 3
    plev = np.array([50.0, 150.0, 400.0, 700.0, 950.0])
    f = [rng.standard_normal((N, nlev, nt)) + off for off in (0, 0, 280, 0)]
    ref = orc.TEMOracle(*f, lat, plev, L=L, zm_dlat=10, mode="factorised")
    for symmetry in (True, False):
        plan = engine.Plan(lat, ref.lat, L, symmetry=symmetry)
        plan.set_tem(nlev, nt, plev * 100)
        res, _ = plan.tem_run(*[torch.as_tensor(x, device="cuda:0") for x in f])
        assert not plan.status()
        res = res.cpu().numpy()
        for i, n in enumerate(_lib.RESULT_NAMES):
            assert fieldnorm_err(res[i], getattr(ref, n)()) <= 1e-10, (symmetry, plan.paired, n)
        plan.close()


@pytest.mark.parametrize("symmetric_shards", [True, False])
def test_ncol_sharded_flow_emulated_on_one_gpu(symmetric_shards):
    """The staged C ABI as NcolShardedTEM drives it, with the all-reduces replaced by explicit sums
    over two 'ranks' that live on the same GPU: Gram, [4][K][D] sums, [3][K][D] flux sums, tracer
    sums.  Must reproduce the unsharded run."""
    from pytemdiags_amd import _lib, engine, sharding, synth
    lat, lon = synth.cubed_sphere_gll(8)
    plev = synth.pressure_levels(10)
    nt = 3
    lat_zm = (np.arange(-90, 91, 1.0)[1:] + np.arange(-90, 91, 1.0)[:-1]) / 2
    f = [torch.as_tensor(x, device="cuda:0") for x in synth.analytic_fields(lat, lon, plev, nt, seed=9)]
    q = torch.as_tensor(synth.analytic_tracer(lat, lon, plev, nt), device="cuda:0")
    full = engine.Plan(lat, lat_zm, 50)
    full.set_tem(10, nt, plev * 100)
    ref, _ = full.tem_run(*f)
    tref, _ = full.tracer_run(q, f[1], f[3])

    if symmetric_shards:
        parts = sharding.symmetric_ncol_shards(lat, 2)
    else:
        parts = [np.arange(*sharding.shard_bounds(lat.size, 2, r)) for r in range(2)]
    plans = [engine.Plan(lat[p], lat_zm, 50, defer_finalize=True) for p in parts]
    import os
    nosym = os.environ.get("TEMX_NO_SYM") == "1"
    if symmetric_shards:
        assert [pl.paired for pl in plans] == [not nosym] * 2
    G = sum(pl.matrix(_lib.MAT_GRAM) for pl in plans)                   # all-reduce (i)
    loc = []
    for pl, p in zip(plans, parts):
        pl.finalize(G.cpu().numpy())
        pl.set_tem(10, nt, plev * 100)
        idx = torch.as_tensor(p, device="cuda:0")
        loc.append([x[idx].contiguous() for x in f] + [q[idx].contiguous()])
    B4 = sum(pl.tem_stage1(*l[:4]) for pl, l in zip(plans, loc))        # all-reduce (ii)
    B3 = sum(pl.tem_stage2_from_sums(B4) if pl.one_pass else pl.tem_stage2(*l[:4], B4)
             for pl, l in zip(plans, loc))                              # all-reduce (iii)
    for pl in plans:                                                    # every rank: same epilogue
        res, _ = pl.tem_stage3(B3)
        for i, n in enumerate(_lib.RESULT_NAMES):
            assert fieldnorm_err(res[i].cpu().numpy(), ref[i].cpu().numpy()) <= 1e-11, n
    Bq = sum(pl.tracer_stage1(l[4]) for pl, l in zip(plans, loc))
    Bq2 = sum(pl.tracer_stage2(l[4], l[1], l[3], Bq) for pl, l in zip(plans, loc))
    tres, _ = plans[0].tracer_stage3(Bq2)
    for i, n in enumerate(_lib.TRACER_RESULT_NAMES):
        assert fieldnorm_err(tres[i].cpu().numpy(), tref[i].cpu().numpy()) <= 1e-11, n
    assert not any(pl.status() for pl in plans)
    for pl in plans + [full]:
        pl.close()


@pytest.mark.parametrize("nlev,nt", [(9, 2), (16, 4)])      # D = 18: generic sliced path; D = 64: class sums first
def test_large_L_eddies_and_tracer_vs_oracle(nlev, nt):
    """K > 64 (sliced sweeps): native eddies/products and the tracer TEM against the oracle."""
    from oracle import tem_oracle as orc
    from pytemdiags_amd import _lib, engine, synth
    ne, L = 12, 70
    lat, lon = synth.cubed_sphere_gll(ne)
    plev = synth.pressure_levels(nlev)
    f = synth.analytic_fields(lat, lon, plev, nt, seed=3)
    q = synth.analytic_tracer(lat, lon, plev, nt, which=1)
    ref = orc.TEMOracle(*f, lat, plev, L=L, mode="factorised", q=[q])
    plan = engine.Plan(lat, ref.lat, L)
    assert not plan.paired                                  # the paired sweeps stop at 64 harmonics
    plan.set_tem(nlev, nt, plev * 100)
    d = [torch.as_tensor(x, device="cuda:0") for x in f]
    qd = torch.as_tensor(q, device="cuda:0")
    res, _ = plan.tem_run(*d)
    ed = plan.tem_eddy(*d)
    for n in _lib.EDDY_NAMES:
        assert fieldnorm_err(ed[n].cpu().numpy(), getattr(ref, n)) <= 1e-10, n
    tres, tzon = plan.tracer_run(qd, d[1], d[3], want_zonal=True)
    assert not plan.status()
    for i, n in enumerate(_lib.TRACER_RESULT_NAMES):
        e = fieldnorm_err(tres[i].cpu().numpy(), getattr(ref, n)(0))
        assert e <= 1e-10, (n, e)
    for i, n in enumerate(_lib.TRACER_ZONAL_NAMES):
        e = fieldnorm_err(tzon[i].cpu().numpy(), getattr(ref, n)[0])
        assert e <= 1e-10, (n, e)
    te = plan.tracer_eddy(qd, d[1], d[3])
    for n in _lib.TRACER_EDDY_NAMES:
        assert fieldnorm_err(te[n].cpu().numpy(), getattr(ref, n)[0]) <= 1e-10, n
    plan.close()


@pytest.mark.parametrize("seed,dtype", [(0, np.float64), (1, np.float64), (2, np.float32)])
def test_latitude_classes_irregular_grid(seed, dtype):
    """Latitude-class sweeps (kernels_cls.hpp) on a grid with uneven classes: 1..11 columns per
    latitude on one or both hemispheres, equator and pole columns, rows in random order --
    the whole pipeline, the native eddies, the tracer TEM and the operator API vs the oracle."""
    import os
    from oracle import tem_oracle as orc
    from pytemdiags_amd import _lib, engine, synth
    rng = np.random.default_rng(seed)
    alat = np.concatenate([[0.0, 90.0], rng.uniform(0.5, 89.5, 260)])
    lats = []
    for a in alat:
        nn, ns = rng.integers(0, 12, 2)
        if nn + ns == 0:
            nn = 1
        if a == 0.0:
            nn, ns = nn + ns, 0
        lats += [a] * nn + [-a] * ns
    lat = np.array(lats)
    rng.shuffle(lat)
    lon = rng.uniform(0, 360, lat.size)
    nlev, nt, L = 7, 3, 40
    plev = synth.pressure_levels(nlev)
    f = synth.analytic_fields(lat, lon, plev, nt, seed=seed, dtype=dtype)
    q = synth.analytic_tracer(lat, lon, plev, nt, which=0, dtype=dtype)
    ref = orc.TEMOracle(*f, lat, plev, L=L, mode="factorised", q=[q])
    plan = engine.Plan(lat, ref.lat, L)
    nosym = os.environ.get("TEMX_NO_SYM") == "1" or os.environ.get("TEMX_NO_CLS") == "1"
    assert plan.sweep_mode == (0 if nosym else 2)
    plan.set_tem(nlev, nt, plev * 100)
    d = [torch.as_tensor(x, device="cuda:0") for x in f]
    qd = torch.as_tensor(q, device="cuda:0")
    tol = 1e-10 if dtype == np.float64 else 2e-5
    res, zon = plan.tem_run(*d, want_zonal=True)
    for i, n in enumerate(_lib.RESULT_NAMES):
        e = fieldnorm_err(res[i].cpu().numpy(), getattr(ref, n)())
        assert e <= tol, (n, e)
    for i, n in enumerate(_lib.ZONAL_NAMES):
        e = fieldnorm_err(zon[i].cpu().numpy(), getattr(ref, n))
        assert e <= tol, (n, e)
    ed = plan.tem_eddy(*d)
    for n in _lib.EDDY_NAMES:
        assert fieldnorm_err(ed[n].cpu().numpy(), getattr(ref, n)) <= tol, n
    tres, _ = plan.tracer_run(qd, d[1], d[3])
    for i, n in enumerate(_lib.TRACER_RESULT_NAMES):
        e = fieldnorm_err(tres[i].cpu().numpy(), getattr(ref, n)(0))
        assert e <= tol, (n, e)
    te = plan.tracer_eddy(qd, d[1], d[3])
    for n in _lib.TRACER_EDDY_NAMES:
        assert fieldnorm_err(te[n].cpu().numpy(), getattr(ref, n)[0]) <= tol, n
    A = rng.standard_normal((lat.size, 37))
    zm = plan.zonal_mean(torch.as_tensor(A, device="cuda:0")).cpu().numpy()
    assert fieldnorm_err(zm, ref.ZM.zonal_mean(A)) <= 1e-10
    assert not plan.status()
    plan.close()


def test_sweep_modes_agree_on_cubed_sphere():
    """generic, mirror-paired and latitude-class sweeps are the same operator (rounding apart)."""
    from pytemdiags_amd import engine, synth
    lat, lon = synth.cubed_sphere_gll(12)
    plev = synth.pressure_levels(20)
    f = [torch.as_tensor(x, device="cuda:0") for x in synth.analytic_fields(lat, lon, plev, 3, seed=5)]
    lat_zm = (np.arange(-90, 91, 1.0)[1:] + np.arange(-90, 91, 1.0)[:-1]) / 2
    outs, modes = [], []
    for kw in ({}, {"classes": False}, {"symmetry": False}):
        plan = engine.Plan(lat, lat_zm, 50, **kw)
        plan.set_tem(20, 3, plev * 100)
        modes.append(plan.sweep_mode)
        outs.append(plan.tem_run(*f)[0].cpu().numpy())
        assert not plan.status()
        plan.close()
    import os
    if os.environ.get("TEMX_NO_SYM") != "1" and os.environ.get("TEMX_NO_CLS") != "1":
        assert modes == [2, 1, 0]
    for o in outs[1:]:
        for i in range(o.shape[0]):
            assert fieldnorm_err(o[i], outs[0][i]) <= 1e-11


def test_latlon_grid_tem_vs_oracle():
    """A structured lat-lon grid (NLON columns per latitude: few, very long latitude classes, so the
    work cuts fall inside class-groups) through the whole pipeline, eddies included."""
    from oracle import tem_oracle as orc
    from pytemdiags_amd import _lib, engine, synth
    nlat, nlon, nlev, nt, L = 91, 180, 5, 2, 30
    lat = np.repeat(np.linspace(-90, 90, nlat), nlon)
    lon = np.tile(np.arange(nlon) * (360.0 / nlon), nlat)
    plev = synth.pressure_levels(nlev)
    f = synth.analytic_fields(lat, lon, plev, nt, seed=11)
    ref = orc.TEMOracle(*f, lat, plev, L=L, mode="factorised")
    plan = engine.Plan(lat, ref.lat, L)
    plan.set_tem(nlev, nt, plev * 100)
    d = [torch.as_tensor(x, device="cuda:0") for x in f]
    res, zon = plan.tem_run(*d, want_zonal=True)
    for i, n in enumerate(_lib.RESULT_NAMES):
        e = fieldnorm_err(res[i].cpu().numpy(), getattr(ref, n)())
        assert e <= 1e-10, (n, e)
    for i, n in enumerate(_lib.ZONAL_NAMES):
        e = fieldnorm_err(zon[i].cpu().numpy(), getattr(ref, n))
        assert e <= 1e-10, (n, e)
    ed = plan.tem_eddy(*d)
    for n in _lib.EDDY_NAMES:
        assert fieldnorm_err(ed[n].cpu().numpy(), getattr(ref, n)) <= 1e-10, n
    assert not plan.status()
    plan.close()


def _one_pass_expected():
    import os
    return not any(os.environ.get(k) == "1" for k in ("TEMX_NO_SYM", "TEMX_NO_CLS", "TEMX_TWO_PASS"))


@pytest.fixture
def force_one_pass(monkeypatch):
    """the one-pass form is normally reserved for >= 1.2e7 elements per field; the parity tests use it on
    small shapes"""
    import os
    if os.environ.get("TEMX_TWO_PASS") != "1":
        monkeypatch.setenv("TEMX_ONE_PASS", "1")


@pytest.mark.parametrize("ne,nlev,nt,dtype,L", [
    (16, 16, 8, np.float64, 50),     # D = 128: two exact quads of d-tiles
    (8, 40, 5, np.float64, 50),      # D = 200: 13 d-tiles, ragged last quad and ragged last tile
    (12, 30, 6, np.float32, 50),     # fp32 inputs
    (8, 16, 5, np.float64, 12),      # 2 blocks of even / odd harmonics (TBS = 2)
    (8, 16, 5, np.float64, 30),      # TBS = 4
    (8, 16, 5, np.float64, 61),      # TBS = 8
])
def test_one_pass_class_path_vs_oracle(ne, nlev, nt, dtype, L, force_one_pass):
    """One-pass form of the class path (kernels_cls.hpp): sweep 1 stores per-class sums of u v,
    u omega, v theta; the eddy-product sums follow algebraically, the fields are read once."""
    from oracle import tem_oracle as orc
    from pytemdiags_amd import _lib, engine, synth
    lat, lon = synth.cubed_sphere_gll(ne)
    plev = synth.pressure_levels(nlev)
    f = synth.analytic_fields(lat, lon, plev, nt, seed=ne, dtype=dtype)
    ref = orc.TEMOracle(*f, lat, plev, L=L, mode="factorised")
    plan = engine.Plan(lat, ref.lat, L)
    plan.set_tem(nlev, nt, plev * 100)
    assert plan.one_pass == _one_pass_expected()
    d = [torch.as_tensor(x, device="cuda:0") for x in f]
    res, zon = plan.tem_run(*d, want_zonal=True)
    assert not plan.status()
    tol = 1e-10 if dtype == np.float64 else 2e-5
    for i, n in enumerate(_lib.RESULT_NAMES):
        e = fieldnorm_err(res[i].cpu().numpy(), getattr(ref, n)())
        assert e <= tol, (n, e)
    for i, n in enumerate(_lib.ZONAL_NAMES):
        e = fieldnorm_err(zon[i].cpu().numpy(), getattr(ref, n))
        assert e <= tol, (n, e)
    # staged == fused; stage 2 given fields always re-reads them (two-pass sweep): same numbers up
    # to rounding
    B4 = plan.tem_stage1(*d)
    B3 = plan.tem_stage2_from_sums(B4) if plan.one_pass else plan.tem_stage2(*d, B4)
    res2, _ = plan.tem_stage3(B3)
    assert torch.equal(res, res2)
    d2 = [x.clone() for x in d]
    B3b = plan.tem_stage2(*d2, B4)
    den = float(B3.abs().max())
    assert float((B3b - B3).abs().max()) <= 1e-11 * den
    # eddies (always the two-pass kernel) and the tracer TEM still work next to it
    ed = plan.tem_eddy(*d)
    for n in ("up", "vptp"):
        assert fieldnorm_err(ed[n].cpu().numpy(), getattr(ref, n)) <= tol, n
    if L == 50 and dtype == np.float64:
        q = synth.analytic_tracer(lat, lon, plev, nt, which=1)
        refq = orc.TEMOracle(*f, lat, plev, L=L, mode="factorised", q=[q])
        tres, _ = plan.tracer_run(torch.as_tensor(q, device="cuda:0"), d[1], d[3])
        for i, n in enumerate(_lib.TRACER_RESULT_NAMES):
            e = fieldnorm_err(tres[i].cpu().numpy(), getattr(refq, n)(0))
            assert e <= 1e-10, (n, e)
    assert not plan.status()
    plan.close()


def test_one_pass_on_uneven_classes(force_one_pass):
    """one-pass sums with uneven class sizes, classes on one hemisphere only, padding rows."""
    from oracle import tem_oracle as orc
    from pytemdiags_amd import _lib, engine, synth
    rng = np.random.default_rng(21)
    lats = []
    for a in np.concatenate([[0.0, 90.0], rng.uniform(0.5, 89.5, 700)]):
        nn, ns = rng.integers(0, 10, 2)
        if nn + ns == 0:
            ns = 2
        if a == 0.0:
            nn, ns = nn + ns, 0
        lats += [a] * nn + [-a] * ns
    lat = np.array(lats)
    rng.shuffle(lat)
    lon = rng.uniform(0, 360, lat.size)
    nlev, nt, L = 19, 7, 30                                   # D = 133: 9 d-tiles
    plev = synth.pressure_levels(nlev)
    f = synth.analytic_fields(lat, lon, plev, nt, seed=4)
    ref = orc.TEMOracle(*f, lat, plev, L=L, mode="factorised")
    plan = engine.Plan(lat, ref.lat, L)
    plan.set_tem(nlev, nt, plev * 100)
    assert plan.one_pass == _one_pass_expected()
    res, _ = plan.tem_run(*[torch.as_tensor(x, device="cuda:0") for x in f])
    assert not plan.status()
    for i, n in enumerate(_lib.RESULT_NAMES):
        e = fieldnorm_err(res[i].cpu().numpy(), getattr(ref, n)())
        assert e <= 1e-10, (n, e)
    plan.close()


def test_one_pass_equals_two_pass_at_baseline_grid_size(monkeypatch):
    """ne30 (48 602 columns) x 72 x 4 (18 d-tiles: one-pass by default): the one-pass and the two-pass
    form of the class path are the same computation up to rounding -- no CPU reference needed."""
    from pytemdiags_amd import _lib, engine, synth
    lat, lon = synth.cubed_sphere_gll(30)
    plev = synth.pressure_levels(72)
    lat_zm = (np.arange(-90, 91, 1.0)[1:] + np.arange(-90, 91, 1.0)[:-1]) / 2
    f = engine.synth_fields(0, lat, lon, plev, 4)
    outs, forms = [], []
    expected = _one_pass_expected()
    for two in (False, True):
        if two:
            monkeypatch.setenv("TEMX_TWO_PASS", "1")
        plan = engine.Plan(lat, lat_zm, 50)
        plan.set_tem(72, 4, plev * 100)
        forms.append(plan.one_pass)
        res, zon = plan.tem_run(*f, want_zonal=True)
        assert not plan.status()
        outs.append((res.cpu().numpy(), zon.cpu().numpy()))
        plan.close()
    assert forms[1] is False
    assert forms[0] == expected
    for a, b, names in ((outs[0][0], outs[1][0], _lib.RESULT_NAMES), (outs[0][1], outs[1][1], _lib.ZONAL_NAMES)):
        for i, n in enumerate(names):
            assert fieldnorm_err(a[i], b[i]) <= 1e-11, (n, fieldnorm_err(a[i], b[i]))


@pytest.mark.parametrize("seed", range(int(__import__("os").environ.get("TEMX_FUZZ_N", "24"))))
def test_fuzz_random_grids_and_shapes(seed, monkeypatch):
    """Randomised grids (class sizes 1..20, some classes on one hemisphere, optional structure-free
    columns), random ncol / nlev / nt / L / dtype and sweep form, whole pipeline vs the oracle."""
    from oracle import tem_oracle as orc
    from pytemdiags_amd import _lib, engine, synth
    rng = np.random.default_rng(1000 + seed)
    nuniq = int(rng.integers(40, 400))
    maxm = int(rng.choice([1, 2, 5, 20]))
    lats = []
    for a in rng.uniform(0.2, 89.8, nuniq):
        nn, ns = rng.integers(0, maxm + 1, 2)
        if nn + ns == 0:
            nn = 1
        lats += [a] * int(nn) + [-a] * int(ns)
    if rng.random() < 0.5:
        lats += [0.0] * int(rng.integers(1, 6))
    if rng.random() < 0.5:
        lats += [90.0, -90.0]
    if rng.random() < 0.3:
        lats += list(rng.uniform(-89, 89, int(rng.integers(1, 50))))     # columns without partners
    lat = np.array(lats)
    rng.shuffle(lat)
    lon = rng.uniform(0, 360, lat.size)
    nlev = int(rng.integers(2, 24))
    nt = int(rng.integers(1, 9))
    L = int(rng.integers(3, min(63, nuniq // 2)))
    if nuniq >= 300 and rng.random() < 0.5:                  # the sliced large-L paths (64 < K)
        L = int(rng.integers(64, min(160, nuniq // 2)))
    dtype = np.float32 if rng.random() < 0.25 else np.float64
    if rng.random() < 0.5:
        monkeypatch.setenv("TEMX_ONE_PASS", "1")
    plev = synth.pressure_levels(nlev)
    f = synth.analytic_fields(lat, lon, plev, nt, seed=seed, dtype=dtype)
    ref = orc.TEMOracle(*f, lat, plev, L=L, mode="factorised")
    plan = engine.Plan(lat, ref.lat, L)
    plan.set_tem(nlev, nt, plev * 100)
    res, zon = plan.tem_run(*[torch.as_tensor(x, device="cuda:0") for x in f], want_zonal=True)
    assert not plan.status()
    # Both sides work on an orthonormalised basis now (engine: Cholesky-QR2 at plan build, temx_plan_finalize;
    # oracle: Householder QR), errors of order cond(Y0) eps: the tolerance is the plain one up to
    # cond(G) = cond(Y0)^2 = 1e8 (round 2 scaled it from cond(G) = 2e3 up).  The BASELINE grids: cond(G) <= 20.
    cond = np.linalg.cond(ref.ZM.Y0.T @ ref.ZM.Y0)
    tol = (1e-10 if dtype == np.float64 else 2e-5) * max(1.0, cond / 1e8)
    if __import__("os").environ.get("TEMX_NO_QR") == "1":    # A/B runs on the plain normal equations: round 2's scaling
        tol = (1e-10 if dtype == np.float64 else 2e-5) * max(1.0, cond / 2e3)
    info = (seed, lat.size, nlev, nt, L, dtype.__name__, plan.sweep_mode, plan.one_pass, "cond %.1e" % cond)
    for i, n in enumerate(_lib.RESULT_NAMES):
        e = fieldnorm_err(res[i].cpu().numpy(), getattr(ref, n)())
        assert e <= tol, (n, e) + info
    for i, n in enumerate(_lib.ZONAL_NAMES):
        e = fieldnorm_err(zon[i].cpu().numpy(), getattr(ref, n))
        assert e <= tol, (n, e) + info
    plan.close()


def test_one_pass_tracer_reads_the_fields_once(force_one_pass, monkeypatch):
    """Tracer TEM on the one-pass class path (tem_diagnostics.py:532-538, 560-570): temx_tracer_stage1_sums
    reads (q, v, omega) once and stores the class sums of q; temx_tracer_stage2_from_sums forms the
    q'v', q'omega' sums from them and the TEM run's class sums of v and omega.  Against the oracle, against
    the two-pass stages, and the state checks of the explicit contract."""
    from oracle import tem_oracle as orc
    from pytemdiags_amd import _lib, engine, synth
    monkeypatch.setenv("TEMX_TRACER_ONE_PASS", "1")     # tracer_run takes the one-pass stages (read at first use)
    lat, lon = synth.cubed_sphere_gll(8)
    plev = synth.pressure_levels(16)
    nt = 4
    f = synth.analytic_fields(lat, lon, plev, nt, seed=21)
    qs = [synth.analytic_tracer(lat, lon, plev, nt, which=i) for i in range(2)]
    ref = orc.TEMOracle(*f, lat, plev, mode="factorised", q=qs)
    plan = engine.Plan(lat, ref.lat, 50)
    plan.set_tem(16, nt, plev * 100)
    d = [torch.as_tensor(x, device="cuda:0") for x in f]
    dq = [torch.as_tensor(x, device="cuda:0") for x in qs]
    if not plan.one_pass:                       # TEMX_NO_CLS / TEMX_TWO_PASS runs of the suite
        with pytest.raises(_lib.TemxError):
            plan.tracer_stage1_sums(dq[0], d[1], d[3])
        plan.close()
        return
    with pytest.raises(_lib.TemxError):         # no TEM run yet: no class sums of v and omega
        plan.tracer_stage1_sums(dq[0], d[1], d[3])
    plan.tem_run(*d)
    for i in range(2):
        tres, tzon = plan.tracer_run(dq[i], d[1], d[3], want_zonal=True)
        for k, n in enumerate(_lib.TRACER_RESULT_NAMES):
            e = fieldnorm_err(tres[k].cpu().numpy(), getattr(ref, n)(i))
            assert e <= 1e-10, (i, n, e)
        for k, n in enumerate(_lib.TRACER_ZONAL_NAMES):
            e = fieldnorm_err(tzon[k].cpu().numpy(), getattr(ref, n)[i])
            assert e <= 1e-10, (i, n, e)
        # staged form == fused form, bit for bit; two-pass stages agree to rounding
        Bq = plan.tracer_stage1_sums(dq[i], d[1], d[3])
        Bq2 = plan.tracer_stage2_from_sums(Bq)
        t2, _ = plan.tracer_stage3(Bq2)     # (tracer_run: these stages, or the two-pass ones -- a process-wide switch)
        assert float((t2 - tres).abs().max()) <= 1e-11 * float(tres.abs().max())
        Bq_b = plan.tracer_stage1(dq[i])
        Bq2_b = plan.tracer_stage2(dq[i], d[1], d[3], Bq_b)
        assert float((Bq_b - Bq).abs().max()) <= 1e-12 * float(Bq.abs().max())
        assert float((Bq2_b - Bq2).abs().max()) <= 1e-11 * float(Bq2.abs().max())
        # the native-grid tracer eddies still come from the two-pass kernel
        ed = plan.tracer_eddy(dq[i], d[1], d[3])
        assert fieldnorm_err(ed["qpvp"].cpu().numpy(), ref.qpvp[i]) <= 1e-10
    # a new TEM stage 1 voids the tracer sums (they pair with one TEM run's sums of v and omega)
    Bq = plan.tracer_stage1_sums(dq[0], d[1], d[3])
    plan.tem_stage1(*d)
    with pytest.raises(_lib.TemxError):
        plan.tracer_stage2_from_sums(Bq)
    # TEM stage 1 on NEW fields, tracer stage 1, and no TEM stage 2 in between -- the v, omega
    # coefficients (C4) still belong to the previous fields, so the tracer's stage 2 must refuse ...
    f2 = synth.analytic_fields(lat, lon, plev, nt, seed=22)
    d2 = [torch.as_tensor(x, device="cuda:0") for x in f2]
    B4 = plan.tem_stage1(*d2)
    Bq = plan.tracer_stage1_sums(dq[0], d2[1], d2[3])
    with pytest.raises(_lib.TemxError) as ei:
        plan.tracer_stage2_from_sums(Bq)
    assert ei.value.code == -5
    # ... and accept once the stage-2 solve of these fields has run; the answer is the one for (q, f2)
    plan.tem_stage3(plan.tem_stage2_from_sums(B4))      # (stage 3 leaves psi, vtem, omegatem for the tracer epilogue)
    t3, _ = plan.tracer_stage3(plan.tracer_stage2_from_sums(Bq))
    ref2 = orc.TEMOracle(*f2, lat, plev, mode="factorised", q=[qs[0]])
    for k, n in enumerate(_lib.TRACER_RESULT_NAMES):
        e = fieldnorm_err(t3[k].cpu().numpy(), getattr(ref2, n)(0))
        assert e <= 1e-10, (n, e)
    assert not plan.status()
    plan.close()


@pytest.mark.parametrize("ne,nlev,nt,dtype,L", [
    (8, 16, 4, np.float64, 50),      # D = 64: exact d-tiles
    (8, 13, 3, np.float64, 50),      # D = 39: ragged last d-tile, work cuts with unequal group counts
    (12, 20, 3, np.float32, 50),     # fp32 inputs
    (8, 16, 5, np.float64, 12),      # TBS = 2
])
def test_tem_and_tracer_in_one_sweep(force_one_pass, ne, nlev, nt, dtype, L):
    """TEMDiagnostics(q=...) knows its tracer up front (tem_diagnostics.py:241-259, 532-538,
    560-570), so temx_tem_tracer_run reads (u, v, T, omega, q) ONCE (four waves share a d-tile, ten
    projections).  Against the oracle, against the separate TEM and tracer runs, in staged form, and the
    state contract."""
    from oracle import tem_oracle as orc
    from pytemdiags_amd import _lib, engine, synth
    lat, lon = synth.cubed_sphere_gll(ne)
    plev = synth.pressure_levels(nlev)
    f = synth.analytic_fields(lat, lon, plev, nt, seed=31, dtype=dtype)
    q = synth.analytic_tracer(lat, lon, plev, nt).astype(dtype)
    ref = orc.TEMOracle(*f, lat, plev, L=L, mode="factorised", q=[q])
    plan = engine.Plan(lat, ref.lat, L)
    plan.set_tem(nlev, nt, plev * 100)
    d = [torch.as_tensor(x, device="cuda:0") for x in f]
    dq = torch.as_tensor(q, device="cuda:0")
    tol = 1e-10 if dtype == np.float64 else 2e-5
    res, zon, tres, tzon = plan.tem_tracer_run(*d, dq, want_zonal=True)
    assert not plan.status()
    for i, n in enumerate(_lib.RESULT_NAMES):
        assert fieldnorm_err(res[i].cpu().numpy(), getattr(ref, n)()) <= tol, n
    for i, n in enumerate(_lib.ZONAL_NAMES):
        assert fieldnorm_err(zon[i].cpu().numpy(), getattr(ref, n)) <= tol, n
    for k, n in enumerate(_lib.TRACER_RESULT_NAMES):
        assert fieldnorm_err(tres[k].cpu().numpy(), getattr(ref, n)(0)) <= tol, n
    for k, n in enumerate(_lib.TRACER_ZONAL_NAMES):
        assert fieldnorm_err(tzon[k].cpu().numpy(), getattr(ref, n)[0]) <= tol, n
    # the separate runs (TEM sweep, then the tracer's own stages) agree to rounding
    r2, _ = plan.tem_run(*d)
    t2, _ = plan.tracer_run(dq, d[1], d[3])
    for a, b in ((res, r2), (tres, t2)):
        for i in range(a.shape[0]):
            assert float((a[i] - b[i]).abs().max()) <= (1e-11 if dtype == np.float64 else 1e-5) * float(b[i].abs().max()), i
    if plan.one_pass:
        # staged form == fused form bit for bit; B4 / Bq are what an ncol-sharded caller all-reduces
        B4, Bq = plan.tem_tracer_stage1(*d, dq)
        B4b = plan.tem_stage1(*d)
        assert float((B4 - B4b).abs().max()) <= 1e-12 * float(B4b.abs().max())
        with pytest.raises(_lib.TemxError):             # stage 1 on its own voids the tracer's class sums
            plan.tracer_stage2_from_sums(Bq)
        B4, Bq = plan.tem_tracer_stage1(*d, dq)
        with pytest.raises(_lib.TemxError):             # no TEM stage 2 yet: v, omega coefficients of older fields
            plan.tracer_stage2_from_sums(Bq)
        r3, _ = plan.tem_stage3(plan.tem_stage2_from_sums(B4))
        t3, _ = plan.tracer_stage3(plan.tracer_stage2_from_sums(Bq))
        assert torch.equal(r3, res) and torch.equal(t3, tres)
    assert not plan.status()
    plan.close()


@pytest.mark.parametrize("ne,nlev,nt,dtype,L", [
    (16, 16, 8, np.float64, 50),     # D = 128
    (12, 13, 5, np.float64, 50),     # D = 65: ragged last d-tile
    (12, 30, 6, np.float32, 50),     # fp32 inputs
    (12, 16, 5, np.float64, 28),     # TBS = 4: degree-2L basis of 8 blocks per parity
    (10, 16, 5, np.float64, 12),     # TBS = 2
    (10, 16, 5, np.float64, 40),     # TBS = 7 with a zero-padded degree-2L basis
])
def test_single_sweep_form(monkeypatch, ne, nlev, nt, dtype, L):
    """The single-sweep form of temx_tem_run (include/temx.h, temx_plan_single_sweep): no class-sum stream, the
    eddy-product sums from the Legendre product linearisation.  Against the oracle, against the class-sum
    form on the same plan inputs, run to run bit for bit, and the tracer / eddy entry points that follow a
    TEM run (tem_diagnostics.py:515-570)."""
    from oracle import tem_oracle as orc
    from pytemdiags_amd import _lib, engine, synth
    import os
    if any(os.environ.get(k) == "1" for k in ("TEMX_NO_SYM", "TEMX_NO_CLS", "TEMX_TWO_PASS", "TEMX_NO_QR")):
        pytest.skip("the single-sweep form needs the one-pass class path on the re-orthogonalised basis")
    lat, lon = synth.cubed_sphere_gll(ne)
    plev = synth.pressure_levels(nlev)
    f = synth.analytic_fields(lat, lon, plev, nt, seed=41, dtype=dtype)
    q = synth.analytic_tracer(lat, lon, plev, nt).astype(dtype)
    ref = orc.TEMOracle(*f, lat, plev, L=L, mode="factorised", q=[q])
    d = [torch.as_tensor(x, device="cuda:0") for x in f]
    dq = torch.as_tensor(q, device="cuda:0")
    tol = 1e-10 if dtype == np.float64 else 2e-5
    monkeypatch.setenv("TEMX_ONE_PASS", "1")
    monkeypatch.setenv("TEMX_SINGLE_SWEEP", "1")
    plan = engine.Plan(lat, ref.lat, L)
    plan.set_tem(nlev, nt, plev * 100)
    assert plan.one_pass and plan.single_sweep
    res, zon = plan.tem_run(*d, want_zonal=True)
    assert not plan.status()
    for i, n in enumerate(_lib.RESULT_NAMES):
        assert fieldnorm_err(res[i].cpu().numpy(), getattr(ref, n)()) <= tol, n
    for i, n in enumerate(_lib.ZONAL_NAMES):
        assert fieldnorm_err(zon[i].cpu().numpy(), getattr(ref, n)) <= tol, n
    res2, _ = plan.tem_run(*d)
    assert torch.equal(res, res2)                                   # fixed-order reductions
    # what follows a TEM run: native eddies from its coefficients; the tracer in its own single sweep (it reuses
    # the degree-2L projections and references of v and omega the TEM run left in the plan)
    ed = plan.tem_eddy(*d)
    assert fieldnorm_err(ed["upvp"].cpu().numpy(), ref.upvp) <= tol
    tres, tzon = plan.tracer_run(dq, d[1], d[3], want_zonal=True)
    for k, n in enumerate(_lib.TRACER_RESULT_NAMES):
        assert fieldnorm_err(tres[k].cpu().numpy(), getattr(ref, n)(0)) <= tol, n
    for k, n in enumerate(_lib.TRACER_ZONAL_NAMES):
        assert fieldnorm_err(tzon[k].cpu().numpy(), getattr(ref, n)[0]) <= tol, n
    ted = plan.tracer_eddy(dq, d[1], d[3])
    assert fieldnorm_err(ted["qpvp"].cpu().numpy(), ref.qpvp[0]) <= tol
    r4, _, t4, _ = plan.tem_tracer_run(*d, dq)                     # TEM + tracer in one call: the two single sweeps
    assert torch.equal(r4, res) and torch.equal(t4, tres)
    # the staged entry points keep the class-sum form and agree
    B4 = plan.tem_stage1(*d)
    r3, _ = plan.tem_stage3(plan.tem_stage2_from_sums(B4))
    for i, n in enumerate(_lib.RESULT_NAMES):
        assert float((r3[i] - res[i]).abs().max()) <= (1e-11 if dtype == np.float64 else 1e-5) * float(res[i].abs().max()), n
    plan.close()
    monkeypatch.setenv("TEMX_SINGLE_SWEEP", "0")
    plan = engine.Plan(lat, ref.lat, L)
    plan.set_tem(nlev, nt, plev * 100)
    assert plan.one_pass and not plan.single_sweep
    plan.close()


@pytest.mark.parametrize("ne,nlev,nt,dtype", [
    (16, 16, 8, np.float64),     # D = 128: two full workgroup columns
    (12, 24, 15, np.float64),    # D = 360: ragged last workgroup column (384)
    (12, 30, 6, np.float32),     # fp32 inputs (the class-sum sweep keeps the tile form for them)
])
def test_row_map_sweeps_equal_tile_map_sweeps(monkeypatch, ne, nlev, nt, dtype):
    """The sweeps that load 1 row x 64 columns per instruction (sweep_osr_kernel, sweep_opr_kernel)
    against the forms that load the MFMA B tile (TEMX_OS_MAP / TEMX_OP_MAP = tile): same sums in another order,
    for the single-sweep run, its tracer, and the staged class-sum entry points."""
    from pytemdiags_amd import _lib, engine, synth
    import os
    if any(os.environ.get(k) == "1" for k in ("TEMX_NO_SYM", "TEMX_NO_CLS", "TEMX_TWO_PASS", "TEMX_NO_QR")):
        pytest.skip("needs the one-pass class path")
    lat, lon = synth.cubed_sphere_gll(ne)
    plev = synth.pressure_levels(nlev)
    f = synth.analytic_fields(lat, lon, plev, nt, seed=43, dtype=dtype)
    q = synth.analytic_tracer(lat, lon, plev, nt).astype(dtype)
    d = [torch.as_tensor(x, device="cuda:0") for x in f]
    dq = torch.as_tensor(q, device="cuda:0")
    lat_zm = np.arange(-88.0, 89.0, 4.0)
    monkeypatch.setenv("TEMX_ONE_PASS", "1")
    monkeypatch.setenv("TEMX_SINGLE_SWEEP", "1")
    out = {}
    for form in ("row", "tile"):
        if form == "tile":
            monkeypatch.setenv("TEMX_OS_MAP", "tile")
            monkeypatch.setenv("TEMX_OP_MAP", "tile")
        plan = engine.Plan(lat, lat_zm, 50)
        plan.set_tem(nlev, nt, plev * 100)
        assert plan.single_sweep
        res, zon = plan.tem_run(*d, want_zonal=True)
        tres, _ = plan.tracer_run(dq, d[1], d[3])
        B4 = plan.tem_stage1(*d)                               # class-sum form: sweep_opr_kernel / sweep_op_kernel
        B3 = plan.tem_stage2_from_sums(B4)
        sres, _ = plan.tem_stage3(B3)
        assert not plan.status()
        out[form] = [x.clone() for x in (res, zon, tres, B4, B3, sres)]
        plan.close()
    # fp32 inputs: the row-map single sweep (sweep_os2_kernel) accumulates the sums of a class side about its first
    # member in fp32, the tile-map kernels in fp64: they agree to ~1e-7 of the eddy amplitude, not to
    # rounding of fp64 -- both far inside the 2e-5 the fp32 path is held to against the oracle.
    tol = 1e-11 if dtype == np.float64 else 5e-6
    for a, b, what in zip(out["row"], out["tile"], ("results", "zonal", "tracer results", "B4", "B3", "staged results")):
        for i in range(a.shape[0]):
            scale = float(b[i].abs().max())
            assert float((a[i] - b[i]).abs().max()) <= tol * scale, (what, i)


@pytest.mark.parametrize("dtype,single", [(np.float64, True), (np.float32, True), (np.float64, False)])
def test_row_map_kernels_on_uneven_classes(monkeypatch, dtype, single):
    """The sweeps of DESIGN.md 5d (sweep_osr_kernel for fp64, sweep_os2_kernel for fp32 inputs, sweep_opr_kernel for
    the class-sum form) on a grid with uneven class sizes, classes on one hemisphere only, padding rows, groups
    without a southern (or northern) batch -- against the oracle, TEM and tracer."""
    from oracle import tem_oracle as orc
    from pytemdiags_amd import _lib, engine, synth
    import os
    if any(os.environ.get(k) == "1" for k in ("TEMX_NO_SYM", "TEMX_NO_CLS", "TEMX_TWO_PASS", "TEMX_NO_QR")):
        pytest.skip("needs the one-pass class path on the re-orthogonalised basis")
    rng = np.random.default_rng(23)
    lats = []
    for a in np.concatenate([[0.0, 90.0], rng.uniform(0.5, 89.5, 900)]):
        nn, ns = rng.integers(0, 12, 2)
        if single:                 # the single sweep needs the re-orthogonalised basis, which keeps the parity of the
            ns = nn = max(nn, 1)   # harmonics only on a mirror-symmetric grid: uneven sizes, but the same on both sides
        if nn + ns == 0:
            nn = 3
        if a == 0.0:
            nn, ns = nn + ns, 0    # (the equator class: northern batches only)
        lats += [a] * nn + [-a] * ns
    lat = np.array(lats)
    rng.shuffle(lat)
    lon = rng.uniform(0, 360, lat.size)
    nlev, nt, L = 16, 12, 30                                  # D = 192: three full workgroup columns
    plev = synth.pressure_levels(nlev)
    f = synth.analytic_fields(lat, lon, plev, nt, seed=6, dtype=dtype)
    q = synth.analytic_tracer(lat, lon, plev, nt).astype(dtype)
    ref = orc.TEMOracle(*f, lat, plev, L=L, mode="factorised", q=[q])
    monkeypatch.setenv("TEMX_ONE_PASS", "1")
    monkeypatch.setenv("TEMX_SINGLE_SWEEP", "1" if single else "0")
    plan = engine.Plan(lat, ref.lat, L)
    plan.set_tem(nlev, nt, plev * 100)
    assert plan.one_pass and plan.single_sweep == single
    d = [torch.as_tensor(x, device="cuda:0") for x in f]
    dq = torch.as_tensor(q, device="cuda:0")
    tol = 1e-10 if dtype == np.float64 else 2e-5
    res, zon = plan.tem_run(*d, want_zonal=True)
    tres, _ = plan.tracer_run(dq, d[1], d[3])
    assert not plan.status()
    for i, n in enumerate(_lib.RESULT_NAMES):
        e = fieldnorm_err(res[i].cpu().numpy(), getattr(ref, n)())
        assert e <= tol, (n, e)
    for i, n in enumerate(_lib.ZONAL_NAMES):
        e = fieldnorm_err(zon[i].cpu().numpy(), getattr(ref, n))
        assert e <= tol, (n, e)
    for k, n in enumerate(_lib.TRACER_RESULT_NAMES):
        e = fieldnorm_err(tres[k].cpu().numpy(), getattr(ref, n)(0))
        assert e <= tol, (n, e)
    plan.close()


def test_latitude_noise_of_real_grids_keeps_the_class_sweeps(monkeypatch):
    """VERDICT r03 #6.  (a) The natural construction of the cubed sphere (no bit-for-bit mirror rebuild: latitudes of
    a class agree to round-off only) stays on the latitude-class sweeps with the default tolerance and holds the
    fp64 parity tolerance.  (b) fp32 fields: a latitude coordinate with 3e-9 degrees of noise -- far outside the fp64
    tolerance -- keeps the class sweeps through TEMX_LAT_TOL_F32 (which the front end sets for fp32 inputs) and holds
    2e-5; without the flag the same grid falls back to another form.  (sph_zonal_mean.py:360-363: Y depends on
    latitude only.)"""
    import os
    from oracle import tem_oracle as orc
    from pytemdiags_amd import _lib, engine, synth
    if any(os.environ.get(k) for k in ("TEMX_NO_SYM", "TEMX_NO_CLS", "TEMX_SYM_TOL_DEG")):
        pytest.skip("the environment fixes the sweeps / the tolerance")
    ne, nlev, nt = 16, 8, 3
    lat, lon = synth.cubed_sphere_gll(ne, mirror=False)
    lat_m, _ = synth.cubed_sphere_gll(ne)
    assert lat.size == lat_m.size
    plev = synth.pressure_levels(nlev)
    f = synth.analytic_fields(lat, lon, plev, nt, seed=5)
    ref = orc.TEMOracle(*f, lat, plev, mode="factorised")
    plan = engine.Plan(lat, ref.lat, 50)
    assert plan.sweep_mode == 2
    plan.set_tem(nlev, nt, plev * 100)
    res, _ = plan.tem_run(*[torch.as_tensor(x, device="cuda:0") for x in f])
    for i, n in enumerate(_lib.RESULT_NAMES):
        assert fieldnorm_err(res[i].cpu().numpy(), getattr(ref, n)()) <= 1e-10, n
    plan.close()
    rng = np.random.default_rng(3)
    latn = lat_m + rng.uniform(-3e-9, 3e-9, lat_m.size)
    f32 = synth.analytic_fields(latn, lon, plev, nt, seed=5, dtype=np.float32)
    ref = orc.TEMOracle(*f32, latn, plev, mode="factorised")
    p0 = engine.Plan(latn, ref.lat, 50)
    assert p0.sweep_mode != 2                                    # fp64 tolerance: no classes on this coordinate
    p0.close()
    plan = engine.Plan(latn, ref.lat, 50, fp32_fields=True)
    assert plan.sweep_mode == 2
    plan.set_tem(nlev, nt, plev * 100)
    res, _ = plan.tem_run(*[torch.as_tensor(x, device="cuda:0") for x in f32])
    for i, n in enumerate(_lib.RESULT_NAMES):
        assert fieldnorm_err(res[i].cpu().numpy(), np.asarray(getattr(ref, n)(), np.float64)) <= 2e-5, n
    plan.close()
    # (c) a latitude coordinate that went through float32 (half an ulp at 90 degrees: 3.8e-6 degrees) needs no
    # tolerance at all: rounding is deterministic, columns whose latitudes agreed before agree after, north and south
    # round alike.  The class sweeps stay, at the fp64 tolerance, on that coordinate.
    lat32 = lat_m.astype(np.float32).astype(np.float64)
    f = synth.analytic_fields(lat32, lon, plev, nt, seed=5)
    ref = orc.TEMOracle(*f, lat32, plev, mode="factorised")
    plan = engine.Plan(lat32, ref.lat, 50)
    assert plan.sweep_mode == 2
    plan.set_tem(nlev, nt, plev * 100)
    res, _ = plan.tem_run(*[torch.as_tensor(x, device="cuda:0") for x in f])
    for i, n in enumerate(_lib.RESULT_NAMES):
        assert fieldnorm_err(res[i].cpu().numpy(), getattr(ref, n)()) <= 1e-10, n
    plan.close()
    # (d) noise that is independent per column and beyond the automatic tolerances (5e-6 degrees) with fp32 fields:
    # TEMX_SYM_TOL_DEG=1e-5 is the knob -- a basis row then moves by L x 1.7e-7 rad = 9e-6 of its size at worst,
    # which the 2e-5 of fp32 fields still covers
    latn = lat_m + rng.uniform(-2.5e-6, 2.5e-6, lat_m.size)
    f32 = synth.analytic_fields(latn, lon, plev, nt, seed=5, dtype=np.float32)
    ref = orc.TEMOracle(*f32, latn, plev, mode="factorised")
    p0 = engine.Plan(latn, ref.lat, 50, fp32_fields=True)
    assert p0.sweep_mode != 2
    p0.close()
    monkeypatch.setenv("TEMX_SYM_TOL_DEG", "1e-5")
    plan = engine.Plan(latn, ref.lat, 50, fp32_fields=True)
    assert plan.sweep_mode == 2
    plan.set_tem(nlev, nt, plev * 100)
    res, _ = plan.tem_run(*[torch.as_tensor(x, device="cuda:0") for x in f32])
    for i, n in enumerate(_lib.RESULT_NAMES):
        assert fieldnorm_err(res[i].cpu().numpy(), np.asarray(getattr(ref, n)(), np.float64)) <= 2e-5, n
    plan.close()


def test_sharded_job_on_a_grid_that_is_not_symmetric_mixed_sweep_forms():
    """temx_plan_finalize used to decide Q = Y0 R^-1 against Y0 from which sweeps the RANK's own block
    takes; on a job whose grid is not equatorially symmetric, a rank with latitude classes kept Y0 and a rank without
    them took Q, and the all-reduced sums mixed two bases.  With a Gram matrix from outside the choice now depends on
    that (global) matrix alone.  Here: rank A owns a symmetric band of a cubed sphere (class sweeps), rank B the columns
    poleward of it with their latitudes jittered by up to 1e-3 degrees (no classes, no mirror pairs: generic sweeps);
    the job's grid, their union, covers the sphere (cond(G) = 2.3) but is not symmetric: the odd entries of its Gram
    matrix are 2e-5 of the diagonal, not rounding noise.  The replicated flow with explicit sums against the oracle
    and against the unsharded run on the same grid."""
    from pytemdiags_amd import _lib, engine, synth
    import os
    if any(os.environ.get(k) == "1" for k in ("TEMX_NO_SYM", "TEMX_NO_CLS")):
        pytest.skip("needs ranks that take different sweeps")
    lat0, lon0 = synth.cubed_sphere_gll(10)
    rng = np.random.default_rng(12)
    a = np.flatnonzero(np.abs(lat0) <= 35.0)
    b = np.flatnonzero(np.abs(lat0) > 35.0)
    lat = lat0.copy()
    lat[b] += rng.uniform(-1e-3, 1e-3, b.size)
    keep = np.concatenate([a, b])
    lat, lon = lat[keep], lon0[keep]
    parts = [np.arange(a.size), a.size + np.arange(b.size)]
    nlev, nt = 8, 3
    plev = synth.pressure_levels(nlev)
    lat_zm = (np.arange(-90, 91, 2.0)[1:] + np.arange(-90, 91, 2.0)[:-1]) / 2
    L = 50
    fh = synth.analytic_fields(lat, lon, plev, nt, seed=4)
    f = [torch.as_tensor(x, device="cuda:0") for x in fh]
    from oracle import tem_oracle as orc
    o = orc.TEMOracle(*fh, lat, plev, zm_dlat=2, L=L, mode="factorised")
    full = engine.Plan(lat, lat_zm, L)
    full.set_tem(nlev, nt, plev * 100)
    ref, _ = full.tem_run(*f)
    assert not full.status()
    for i, n in enumerate(_lib.RESULT_NAMES):
        assert fieldnorm_err(ref[i].cpu().numpy(), getattr(o, n)()) <= 1e-10, n
    plans = [engine.Plan(lat[p], lat_zm, L, defer_finalize=True) for p in parts]
    assert plans[0].sweep_mode == 2 and plans[1].sweep_mode == 0         # class sweeps / generic sweeps
    G = sum(pl.matrix(_lib.MAT_GRAM) for pl in plans).cpu().numpy()
    assert np.max(np.abs(G[0::2, 1::2])) > 1e-6 * np.max(np.abs(np.diag(G)))     # not a checkerboard: the job is not symmetric
    for pl in plans:
        pl.finalize(G)
    G2 = sum(pl.matrix(_lib.MAT_GRAM2) for pl in plans).cpu().numpy()
    loc = []
    for pl, p in zip(plans, parts):
        pl.refine(G2)
        pl.set_tem(nlev, nt, plev * 100)
        idx = torch.as_tensor(p, device="cuda:0")
        loc.append([x[idx].contiguous() for x in f])
    B4 = sum(pl.tem_stage1(*l) for pl, l in zip(plans, loc))
    B3 = sum(pl.tem_stage2_from_sums(B4) if pl.one_pass else pl.tem_stage2(*l, B4) for pl, l in zip(plans, loc))
    for pl in plans:
        res, _ = pl.tem_stage3(B3)
        for i, n in enumerate(_lib.RESULT_NAMES):
            assert fieldnorm_err(res[i].cpu().numpy(), ref[i].cpu().numpy()) <= 1e-10, n
    assert not any(pl.status() for pl in plans)
    for pl in plans + [full]:
        pl.close()
