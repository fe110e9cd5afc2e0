"""GPU tests of the drop-in front end (TEMDiagnostics / sph_zonal_averager) against the goldens
the reference produced: values, dtypes, dims, input kinds and error behaviour."""
import os

import numpy as np
import pytest

from conftest import GOLDEN, fieldnorm_err

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

RESULTS = ("vtem", "omegatem", "wtem", "psitem", "epfy", "epfz", "epdiv", "utendepfd", "utendvtem", "utendwtem")
ZONAL = ("ub", "vb", "thetab", "wapb", "upvpb", "upwappb", "vptpb", "dub_dp", "dthetab_dp", "ubcoslat",
         "dubcoslat_dlat", "psi", "psicoslat", "dpsicoslat_dlat", "dpsi_dp", "int_vbdp")
NATIVE = ("up", "vp", "thetap", "wapp", "upvp", "upwapp", "vptp")


def load(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


def labeled(g, k):
    from pytemdiags_amd import LabeledArray
    return LabeledArray(g[k], ("ncol", "plev", "time"), {"plev": g["plev"], "time": g["time"]}, name=k)


def vals(x):
    v = x.values if hasattr(x, "dims") else x
    return v.cpu().numpy() if hasattr(v, "cpu") else np.asarray(v)


@pytest.mark.parametrize("case", ["tem_ne4_30x1_f64", "tem_ne4_30x1_f32", "tem_ne4_30x1_desc"])
def test_temdiagnostics_labeled_matches_reference(case):
    from pytemdiags_amd import TEMDiagnostics, LabeledArray
    g = load(case)
    tol = 2e-5 if g["ua"].dtype == np.float32 else 1e-10
    lat = LabeledArray(g["lat"], ("ncol",))
    tem = TEMDiagnostics(labeled(g, "ua"), labeled(g, "va"), labeled(g, "ta"), labeled(g, "wap"), lat,
                         debug_level=0)
    assert (tem.NCOL, tem.NLEV, tem.NT, tem.ZM_N) == (866, 30, 1, 180)
    np.testing.assert_array_equal(tem.lat, g["lat_zm"])
    assert tem.plev[0] < tem.plev[-1]                     # always ascending (tem_diagnostics.py:372-382)
    for n in RESULTS:
        r = getattr(tem, n)()
        assert isinstance(r, LabeledArray) and r.dims == ("lat", "plev", "time") and r.name == n
        assert r.dtype == g["res_" + n].dtype, n
        assert fieldnorm_err(r.values, g["res_" + n]) <= tol, n
        np.testing.assert_array_equal(r.coords["lat"], g["lat_zm"])
    for n in ZONAL:
        z = getattr(tem, n)
        assert z.dtype == g["zm_" + n].dtype, n
        assert fieldnorm_err(z.values, g["zm_" + n]) <= tol, n
    if "nat_up" in g.files:
        for n in NATIVE:
            e = getattr(tem, n)
            assert e.dims == ("ncol", "plev", "time") and e.dtype == g["nat_" + n].dtype, n
            assert fieldnorm_err(e.values, g["nat_" + n]) <= tol, n
        th = tem.theta
        assert th.dtype == np.float64 and fieldnorm_err(th.values, g["theta"]) <= 1e-12


def test_raw_numpy_torch_and_transposed_inputs():
    from pytemdiags_amd import TEMDiagnostics
    g = load("tem_ne4_12x3_L20_dlat3")
    kw = dict(L=int(g["L"]), zm_dlat=float(g["zm_dlat"]), debug_level=0)
    # raw numpy
    t1 = TEMDiagnostics(g["ua"], g["va"], g["ta"], g["wap"], g["lat"], plev=g["plev"], time=g["time"], **kw)
    r1 = t1.epdiv()
    assert isinstance(r1, np.ndarray) and fieldnorm_err(r1, g["res_epdiv"]) <= 1e-10
    # raw torch, (time, plev, ncol) order
    tt = [torch.as_tensor(np.ascontiguousarray(np.transpose(g[k], (2, 1, 0))), device="cuda:0")
          for k in ("ua", "va", "ta", "wap")]
    t2 = TEMDiagnostics(*tt, g["lat"], plev=g["plev"], dims=("time", "plev", "ncol"), **kw)
    r2 = t2.psitem()
    assert isinstance(r2, torch.Tensor) and r2.is_cuda
    assert fieldnorm_err(r2.cpu().numpy(), g["res_psitem"]) <= 1e-10
    assert fieldnorm_err(t2.ub.cpu().numpy(), g["zm_ub"]) <= 1e-10


def test_two_dimensional_input_is_accepted():
    """The reference's 2-D path is broken; here (ncol, plev) works as one snapshot."""
    from pytemdiags_amd import TEMDiagnostics
    g = load("tem_ne4_30x1_f64")
    f = [g[k][:, :, 0] for k in ("ua", "va", "ta", "wap")]
    tem = TEMDiagnostics(*f, g["lat"], plev=g["plev"], debug_level=0)
    assert fieldnorm_err(tem.vtem(), g["res_vtem"]) <= 1e-10


def test_operator_frontend_and_errors():
    from pytemdiags_amd import sph_zonal_averager, LabeledArray
    g = load("op_ne4_L30")
    ZM = sph_zonal_averager(g["lat"], g["lat_out"], int(g["L"]))
    assert ZM.Y0 is None and ZM.Y0p is None and ZM.N == 866 and ZM.M == 90
    assert ZM.grid_name == "ncol866" and ZM.grid_out_name == "2.0deg"
    A = LabeledArray(g["in_rand3d"], ("ncol", "a", "b"), name="x")
    with pytest.raises(RuntimeError):
        ZM.sph_zonal_mean(A)                               # matrices undefined (sph_zonal_mean.py:213)
    ZM.sph_compute_matrices()
    assert np.max(np.abs(ZM.Y0 - g["Y0"])) < 2e-12 and np.max(np.abs(ZM.Y0p - g["Y0p"])) < 2e-12
    assert np.max(np.abs(ZM.Y0inv @ ZM.Y0 - np.eye(31))) < 1e-12
    d, o = ZM.sanity_check()
    assert abs(d - 31) < 1e-10 and abs(o) < 1e-10
    zm = ZM.sph_zonal_mean(A)
    assert zm.dims == ("lat", "a", "b") and zm.attrs["long_name"] == "zonal mean of x"
    assert np.max(np.abs(zm.values - g["zm_rand3d"])) < 1e-10
    zn = ZM.sph_zonal_mean_native(A)
    assert zn.dims == ("ncol", "a", "b") and np.max(np.abs(zn.values - g["zmn_rand3d"])) < 1e-10
    a32 = ZM.sph_zonal_mean(g["in_rand3d_f32"])            # raw ndarray in -> ndarray out, input dtype
    assert a32.dtype == np.float32 and np.max(np.abs(a32 - g["zm_rand3d_f32"])) < 1e-5
    bad = g["in_y20"].copy(); bad[5] = np.nan
    with pytest.raises(RuntimeError, match="nans"):
        ZM.sph_zonal_mean(bad)
    with pytest.raises(RuntimeError, match="leftmost"):
        ZM.sph_zonal_mean(LabeledArray(g["in_rand3d"].transpose(1, 0, 2).copy(), ("a", "ncol", "b")))
    with pytest.raises(RuntimeError, match="leftmost"):
        ZM.sph_zonal_mean(np.zeros(17))


def test_tem_nan_input_raises():
    from pytemdiags_amd import TEMDiagnostics
    g = load("tem_ne4_12x3_L20_dlat3")
    ta = g["ta"].copy(); ta[100, 2, 1] = np.nan
    with pytest.raises(RuntimeError, match="nans"):
        TEMDiagnostics(g["ua"], g["va"], ta, g["wap"], g["lat"], plev=g["plev"], L=20, zm_dlat=3, debug_level=0)


TRES = ("etfy", "etfz", "etdiv", "qtendetfd", "qtendvtem", "qtendwtem")
TZON = ("qb", "qpvpb", "qpwappb", "dqb_dp", "qbcoslat", "dqbcoslat_dlat")
TNAT = ("qp", "qpvp", "qpwapp")


@pytest.mark.parametrize("case", ["tracer_ne4_10x2_f64", "tracer_ne4_10x2_qf32"])
def test_tracer_tem_matches_reference(case):
    """Abalos+ 2017 tracer TEM (tem_diagnostics.py:532-538, 560-570, 602-611, 801-991)."""
    from pytemdiags_amd import TEMDiagnostics, LabeledArray
    g = load(case)
    nq = int(g["ntrac"])
    f32 = g["q0"].dtype == np.float32
    tol = 2e-5 if f32 else 1e-10
    q = [LabeledArray(g["q%d" % i], ("ncol", "plev", "time"), {"plev": g["plev"], "time": g["time"]},
                      name="Q%d" % i) for i in range(nq)]
    tem = TEMDiagnostics(labeled(g, "ua"), labeled(g, "va"), labeled(g, "ta"), labeled(g, "wap"),
                         LabeledArray(g["lat"], ("ncol",)), q=q if nq > 1 else q[0], debug_level=0)
    assert tem.ntrac == nq
    for n in RESULTS:
        assert fieldnorm_err(getattr(tem, n)().values, g["res_" + n]) <= (1e-10 if not f32 else 1e-10), n
    if nq > 1:
        with pytest.raises(RuntimeError, match="qi must be passed"):      # tem_diagnostics.py:815
            tem.etfy()
    else:
        assert fieldnorm_err(tem.etfy().values, g["q0_res_etfy"]) <= tol
    for i in range(nq):
        for n in TRES:
            r = getattr(tem, n)(i)
            ref = g["q%d_res_%s" % (i, n)]
            assert r.dims == ("lat", "plev", "time") and r.dtype == ref.dtype, n
            assert fieldnorm_err(r.values, ref) <= tol, (i, n, fieldnorm_err(r.values, ref))
        for n in TZON:
            z = getattr(tem, n)[i]
            ref = g["q%d_%s" % (i, n)]
            assert z.dtype == ref.dtype, (n, z.dtype, ref.dtype)
            assert fieldnorm_err(z.values, ref) <= tol, (i, n)
        for n in TNAT:
            e = getattr(tem, n)[i]
            ref = g["q%d_%s" % (i, n)]
            assert e.dims == ("ncol", "plev", "time") and e.dtype == ref.dtype, (n, e.dtype, ref.dtype)
            assert fieldnorm_err(e.values, ref) <= tol, (i, n)


def test_tracer_staged_equals_fused():
    from pytemdiags_amd import engine
    g = load("tracer_ne4_10x2_f64")
    lat_zm = (np.arange(-90, 91, 1.0)[1:] + np.arange(-90, 91, 1.0)[:-1]) / 2
    plan = engine.Plan(g["lat"], lat_zm, 50)
    plan.set_tem(10, 2, g["plev"] * 100)
    d = {k: torch.as_tensor(g[k], device="cuda:0") for k in ("ua", "va", "ta", "wap", "q0", "q1")}
    plan.tem_run(d["ua"], d["va"], d["ta"], d["wap"])
    t1, _ = plan.tracer_run(d["q1"], d["va"], d["wap"])
    Bq = plan.tracer_stage1(d["q1"])
    Bq2 = plan.tracer_stage2(d["q1"], d["va"], d["wap"], Bq)
    t2, _ = plan.tracer_stage3(Bq2)
    assert torch.equal(t1, t2) and not plan.status()
    plan.close()


def test_latlon_grid_zonal_mean_is_the_longitude_average():
    """Structured data through format_latlon_data (tem_util.py:247-342).  On a regular lat-lon grid
    Y0^T A only sees the longitude sums, so for a profile inside span{Y_l^0, l <= L} (a polynomial
    of degree <= L in sin(lat)) plus any zonally varying part that averages out, the
    spherical-harmonic zonal mean IS the profile -- no CPU reference needed, any size."""
    from pytemdiags_amd import sph_zonal_averager, LabeledArray
    from pytemdiags_amd.tem_util import format_latlon_data
    nlat, nlon, nlev = 181, 360, 7
    lat = np.linspace(-90, 90, nlat)
    lon = np.arange(nlon) * (360.0 / nlon)
    s = np.sin(np.deg2rad(lat))
    prof = lambda x, k: 1.0 + 0.5 * x - 2.0 * x ** 2 + 0.3 * x ** (5 + k)       # noqa: E731
    lam = np.deg2rad(lon)
    f = np.empty((nlat, nlon, nlev))
    for k in range(nlev):
        f[:, :, k] = prof(s, k)[:, None] + (3.0 + k) * np.cos((k + 1) * lam)[None, :] * np.cos(np.deg2rad(lat))[:, None] ** 2
    data = format_latlon_data({"lat": LabeledArray(lat, ("lat",)), "lon": LabeledArray(lon, ("lon",)),
                               "F": LabeledArray(f, ("lat", "lon", "lev"), name="F")})
    lat_out = np.linspace(-88, 88, 45)
    ZM = sph_zonal_averager(data["lat"].values, lat_out, 20)
    ZM.sph_compute_matrices()
    zm = ZM.sph_zonal_mean(data["F"])
    assert zm.dims == ("lat", "lev")
    so = np.sin(np.deg2rad(lat_out))
    for k in range(nlev):
        assert np.max(np.abs(zm.values[:, k] - prof(so, k))) < 1e-11, k
    zmn = ZM.sph_zonal_mean_native(data["F"]).values.reshape(nlat, nlon, nlev)
    for k in range(nlev):
        assert np.max(np.abs(zmn[:, :, k] - prof(s, k)[:, None])) < 1e-11, k


def test_to_netcdf_round_trip(tmp_path):
    """to_netcdf / q_to_netcdf (tem_diagnostics.py:995-1103): file naming, variable set (incl. the
    reference's 'wawpp' and 'dqp_dp' keys), dims and values; NetCDF-3 through scipy when xarray is absent."""
    from pytemdiags_amd import TEMDiagnostics, LabeledArray, ncio
    g = load("tracer_ne4_10x2_f64")
    q = [LabeledArray(g["q%d" % i], ("ncol", "plev", "time"), {"plev": g["plev"], "time": g["time"]},
                      name="Q%d" % i) for i in range(int(g["ntrac"]))]
    tem = TEMDiagnostics(labeled(g, "ua"), labeled(g, "va"), labeled(g, "ta"), labeled(g, "wap"),
                         LabeledArray(g["lat"], ("ncol",)), q=q, debug_level=0)
    path = tem.to_netcdf(loc=str(tmp_path), prefix="run1", include_attrs=True)
    assert path == "%s/run1_TEM_ncol866_1.0deg_L50.nc" % tmp_path and tem.out_file == path
    ds = ncio.read_dataset(path)
    want = set(RESULTS) | (set(ZONAL) | set(NATIVE) | {"wawpp"}) - {"wapp"}
    assert want <= set(ds) and {"lat", "plev", "time"} <= set(ds)
    for n in RESULTS:
        dims, v = ds[n]
        assert dims == ("lat", "plev", "time") and v.dtype == g["res_" + n].dtype
        assert fieldnorm_err(v, g["res_" + n]) <= 1e-10, n
    assert ds["wawpp"][0] == ("ncol", "plev", "time") and np.array_equal(ds["wawpp"][1], vals(tem.wapp))
    np.testing.assert_array_equal(ds["lat"][1], tem.lat)
    np.testing.assert_allclose(ds["plev"][1], np.sort(g["plev"]))
    files = tem.q_to_netcdf(loc=str(tmp_path), include_attrs=True)
    assert files[1].endswith("TEM_ncol866_1.0deg_L50_TRACER-Q1.nc")
    dq = ncio.read_dataset(files[1])
    assert {"etfy", "etfz", "etdiv", "qtendetfd", "qtendvtem", "qtendwtem", "dqp_dp", "qpvp"} <= set(dq)
    assert fieldnorm_err(dq["etdiv"][1], g["q1_res_etdiv"]) <= 1e-10
    assert dq["qpvp"][0] == ("ncol", "plev", "time")


def test_native_attributes_stream_in_column_blocks():
    """iter_native / temx_tem_eddy_rows: the native-grid attributes block by block equal the whole arrays
    (streaming them for large runs with bounded device memory)."""
    from pytemdiags_amd import TEMDiagnostics, LabeledArray
    g = load("tem_ne4_30x1_f32")
    tem = TEMDiagnostics(labeled(g, "ua"), labeled(g, "va"), labeled(g, "ta"), labeled(g, "wap"),
                         LabeledArray(g["lat"], ("ncol",)), debug_level=0)
    whole = {n: vals(getattr(tem, n)) for n in NATIVE}
    seen = 0
    for c0, c1, blk in tem.iter_native(chunk_cols=200):          # 192-column blocks, ragged last one
        assert c0 == seen and c0 % 16 == 0 and set(blk) == set(NATIVE)
        for n in NATIVE:
            assert blk[n].dtype == whole[n].dtype and blk[n].shape == (c1 - c0, 30, 1)
            np.testing.assert_allclose(blk[n], whole[n][c0:c1], rtol=0, atol=1e-6 * np.max(np.abs(whole[n])))
        seen = c1
    assert seen == 866
    for n in NATIVE:
        assert fieldnorm_err(whole[n], g["nat_" + n]) <= 2e-5


def test_map_cache_files_round_trip(tmp_path):
    """The reference's NetCDF map cache (sph_zonal_mean.py:329-345, 400-417): file names, variables
    ``Y0[ncol, l]``, ``Y0inv[l, ncol]``, ``Y0p[ncol, l]``; written after the build, read back by the next
    averager, deleted by ``overwrite``; a cache of another L is ignored."""
    import os
    from pytemdiags_amd import sph_zonal_averager, ncio
    g = load("op_ne4_L30")
    L = int(g["L"])
    dest = str(tmp_path / "maps")
    Z = sph_zonal_averager(g["lat"], g["lat_out"], L, save_dest=dest, grid_name="ne4np4", grid_out_name="2deg")
    assert Z.Y0 is None                                    # nothing cached yet (read_only probe, :177)
    Z.sph_compute_matrices()
    assert Z.Y0_file_out == "%s/Y0_ne4np4_L%d.nc" % (dest, L)
    assert Z.Y0p_file_out == "%s/Y0p_ne4np4_2deg_L%d.nc" % (dest, L)
    assert os.path.isfile(Z.Y0_file_out) and os.path.isfile(Z.Y0p_file_out) and not Z.map_cache_used
    a, b = ncio.read_dataset(Z.Y0_file_out), ncio.read_dataset(Z.Y0p_file_out)
    assert a["Y0"][0] == ("ncol", "l") and a["Y0inv"][0] == ("l", "ncol") and b["Y0p"][0] == ("ncol", "l")
    assert np.max(np.abs(a["Y0"][1] - g["Y0"])) < 2e-12 and np.max(np.abs(b["Y0p"][1] - g["Y0p"])) < 2e-12
    assert np.max(np.abs(a["Y0inv"][1] @ a["Y0"][1] - np.eye(L + 1))) < 1e-10
    # a second averager finds the cache in its constructor, like the reference
    Z2 = sph_zonal_averager(g["lat"], g["lat_out"], L, save_dest=dest, grid_name="ne4np4", grid_out_name="2deg")
    assert Z2.map_cache_used and np.array_equal(Z2.Y0, a["Y0"][1]) and np.array_equal(Z2.Y0inv, a["Y0inv"][1])
    zm = Z2.sph_zonal_mean(g["in_rand3d"])
    assert np.max(np.abs(zm - g["zm_rand3d"])) <= 1e-10 * max(1.0, float(np.max(np.abs(g["in_rand3d"]))))
    # overwrite deletes and rebuilds (:332-334); no_write leaves no file
    t0 = os.path.getmtime(Z.Y0_file_out)
    Z3 = sph_zonal_averager(g["lat"], g["lat_out"], L, save_dest=dest, grid_name="ne4np4", grid_out_name="2deg",
                            overwrite=True)
    assert Z3.Y0 is None and not os.path.isfile(Z.Y0_file_out)
    Z3.sph_compute_matrices(no_write=True)
    assert not os.path.isfile(Z.Y0_file_out) and Z3.Y0 is not None
    # a file of the right name but another grid is reported and ignored
    Z.sph_compute_matrices()
    assert os.path.getmtime(Z.Y0_file_out) >= t0
    lat_other = g["lat"][::-1].copy()
    with pytest.warns(UserWarning):
        Z4 = sph_zonal_averager(lat_other, g["lat_out"], L, save_dest=dest, grid_name="ne4np4", grid_out_name="2deg")
    assert not Z4.map_cache_used
    assert np.max(np.abs(Z4.sph_zonal_mean(g["in_rand3d"][::-1]) - g["zm_rand3d"])) <= 1e-10 * 5
    # without save_dest nothing is read or written (see the class docstring)
    Z5 = sph_zonal_averager(g["lat"], g["lat_out"], L)
    Z5.sph_compute_matrices()
    assert not os.path.exists(Z5.Y0_file_out)


def test_map_cache_with_weights(tmp_path):
    """a map cache on file together with ``weights=``.  The constructor probes the cache
    before the reference scales its ``weights`` attribute by 4 pi (sph_zonal_mean.py:177-181); the
    engine must get the caller's weights either way, and quadrature weights do not make ``Y0inv Y0`` the
    identity, so the cache is validated against ``Y0^T diag(4 pi w)``.  Checked against the reference's
    own weighted results (tests/golden/opw_*.npz)."""
    from pytemdiags_amd import sph_zonal_averager
    g = load("opw_gauss24x48_L10")
    L = int(g["L"])
    dest = str(tmp_path / "maps")
    kw = dict(save_dest=dest, grid_name="gauss", grid_out_name="out")
    Z = sph_zonal_averager(g["lat"], g["lat_out"], L, weights=list(g["weights"]), **kw)   # a list is fine
    assert Z.Y0 is None
    Z.sph_compute_matrices()
    assert np.allclose(Z.weights, g["weights"] * 4 * np.pi, rtol=0, atol=0)
    zm = Z.sph_zonal_mean(g["in_rand3d"])
    assert np.max(np.abs(zm - g["zm_rand3d"])) <= 1e-10 * float(np.max(np.abs(g["zm_rand3d"])))
    # second averager: cache hit in the constructor, weights not yet scaled there
    Z2 = sph_zonal_averager(g["lat"], g["lat_out"], L, weights=g["weights"].copy(), **kw)
    assert Z2.map_cache_used and Z2.Y0 is not None
    assert np.allclose(Z2.weights, g["weights"] * 4 * np.pi, rtol=0, atol=0)
    for k in ("y20", "rand3d"):
        zm2 = Z2.sph_zonal_mean(g["in_" + k])
        assert np.max(np.abs(zm2 - g["zm_" + k])) <= 1e-10 * max(1.0, float(np.max(np.abs(g["zm_" + k])))), k
        zn2 = Z2.sph_zonal_mean_native(g["in_" + k])
        assert np.max(np.abs(zn2 - g["zmn_" + k])) <= 1e-10 * max(1.0, float(np.max(np.abs(g["zmn_" + k])))), k
    assert np.max(np.abs(Z2.Y0inv - Z.Y0inv)) == 0.0
    # the same files, other weights: reported, and the operator follows the arguments
    w2 = g["weights"][::-1].copy() * 0.5 + 0.5 / g["weights"].size
    with pytest.warns(UserWarning):
        Z3 = sph_zonal_averager(g["lat"], g["lat_out"], L, weights=w2, **kw)
    assert not Z3.map_cache_used
    assert np.max(np.abs(Z3.Y0inv - Z3.Y0.T * (w2 * 4 * np.pi)[None, :])) == 0.0
