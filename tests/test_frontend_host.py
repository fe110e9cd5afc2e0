"""CPU-only tests: host-side validation of the front end (errors raised before any device work),
the C-ABI library's exported symbols, and the sharding helpers."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_symbol_of_the_header():
    from pytemdiags_amd import _lib
    lib = _lib.load()                                   # raises if libtemx.so is missing / stale
    header = open(os.path.join(ROOT, "include", "temx.h")).read()
    declared = set(re.findall(r"\b(temx_[a-z0-9_]+)\s*\(", header))
    declared.discard("temx_plan")
    bound = {n for n, _, _ in _lib.SIGNATURES}
    assert declared == bound, (declared ^ bound)
    for n in declared:
        assert hasattr(lib, n), n
    assert lib.temx_version() >= 100
    assert isinstance(lib.temx_last_error(), bytes)


def test_no_exception_crosses_the_c_abi():
    """Every entry point of libtemx.so is a function-try-block: a C++ exception inside comes back as an error code
    (host allocation failure -> TEMX_ENOMEM, anything else -> TEMX_EINTERNAL), never as std::terminate."""
    from pytemdiags_amd import _lib
    lib = _lib.load()
    assert lib.temx_selftest_exception(0) == -3 and b"memory" in lib.temx_last_error()
    assert lib.temx_selftest_exception(1) == -7 and b"self-test" in lib.temx_last_error()
    assert lib.temx_selftest_exception(2) == -7
    assert lib.temx_selftest_exception(9) == -1
    src = open(os.path.join(ROOT, "pytemdiags_amd", "csrc", "temx.hip")).read()
    body = src[src.index('extern "C" {'):src.index('}  // extern "C"')]
    multi = [m for m in re.finditer(r"^int temx_\w+\([^;{]*\)\s*(try )?\{\s*$", body, re.M)]
    assert len(multi) >= 40 and all(m.group(1) for m in multi), [m.group(0)[:60] for m in multi if not m.group(1)]


def test_engine_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from pytemdiags_amd import _lib
    lib = _lib.load()
    h = ctypes.c_void_p()
    lat = (ctypes.c_double * 4)(0, 10, 20, 30)
    rc = lib.temx_plan_create(ctypes.byref(h), 0, 4, 2, 4, lat, lat, 0)
    assert rc == -2 and not h.value                      # TEMX_EHIP: no device, no fallback
    assert b"" != lib.temx_last_error()
    from pytemdiags_amd import sph_zonal_averager
    ZM = sph_zonal_averager(np.linspace(-80, 80, 50), np.linspace(-60, 60, 7), 5)
    with pytest.raises(_lib.TemxError):
        ZM.sph_compute_matrices()


def test_frontend_validation_errors_match_reference():
    from pytemdiags_amd import TEMDiagnostics, LabeledArray
    N, nlev, nt = 40, 5, 2
    lat = np.linspace(-85, 85, N)
    plev = np.linspace(100, 900, nlev)

    def mk(dims=("ncol", "plev", "time"), shape=(N, nlev, nt)):
        return LabeledArray(np.ones(shape), dims, {"plev": plev, "time": np.arange(nt)})
    good = mk()
    with pytest.raises(RuntimeError, match="does not contain dim"):            # tem_diagnostics.py:316-318
        TEMDiagnostics(mk(dims=("cells", "plev", "time")), good, good, good, lat, debug_level=0)
    with pytest.raises(RuntimeError, match="these must match"):               # :320-323
        TEMDiagnostics(good, good, good, good, lat[:-1], debug_level=0)
    with pytest.raises(RuntimeError, match="dims"):                            # :326-329
        TEMDiagnostics(LabeledArray(np.ones((N, nlev, nt, 2)), ("ncol", "plev", "time", "x"), {"plev": plev}),
                       good, good, good, lat, debug_level=0)
    with pytest.raises(AssertionError):                                        # :389
        TEMDiagnostics(good, good, good, good, lat, zm_dlat=7, debug_level=0)
    with pytest.raises(RuntimeError, match="xarray DataArray"):                # :313
        TEMDiagnostics([1, 2], [1, 2], [1, 2], [1, 2], lat, plev=plev, debug_level=0)
    with pytest.raises(RuntimeError, match="tracers"):                          # :294
        TEMDiagnostics(good, good, good, good, lat, q=["not an array"], debug_level=0)
    with pytest.raises(RuntimeError, match="same kind"):
        TEMDiagnostics(good, good, good, good, lat, q=[np.ones((N, nlev, nt))], debug_level=0)


def test_averager_constructor_attributes_without_gpu():
    from pytemdiags_amd import sph_zonal_averager
    lat = np.linspace(-89, 89, 120)
    lat_out = np.arange(-89.5, 90, 1.0)
    Z = sph_zonal_averager(lat, lat_out, 50, save_dest="/tmp/maps")
    assert (Z.N, Z.M, Z.L) == (120, 180, 50) and list(Z.l[:3]) == [0, 1, 2]
    assert Z.Y0 is None and Z.Y0inv is None and Z.Y0p is None
    assert Z.Y0_file_out == "/tmp/maps/Y0_ncol120_L50.nc"                       # sph_zonal_mean.py:169
    assert Z.Y0p_file_out == "/tmp/maps/Y0p_ncol120_1.0deg_L50.nc"              # :173
    w = np.full(120, 1 / 120)
    Zw = sph_zonal_averager(lat, lat_out, 5, weights=w)
    assert np.allclose(Zw.weights, 4 * np.pi / 120) and np.allclose(w, 1 / 120)  # scaled copy (:181)


def test_shard_bounds():
    from pytemdiags_amd.sharding import shard_bounds
    b = [shard_bounds(730, 8, r) for r in range(8)]
    assert b[0] == (0, 92) and b[-1][1] == 730 and sorted(e - s for s, e in b) == [91] * 6 + [92] * 2
    assert all(b[i][1] == b[i + 1][0] for i in range(7))
    assert [shard_bounds(777602, 8, r)[1] - shard_bounds(777602, 8, r)[0] for r in range(8)].count(97200) == 6
    assert shard_bounds(5, 8, 7) == (5, 5)


def test_symmetric_ncol_shards():
    from pytemdiags_amd import sharding, synth
    lat, _ = synth.cubed_sphere_gll(4)
    parts = sharding.symmetric_ncol_shards(lat, 3)
    allidx = np.sort(np.concatenate(parts))
    assert np.array_equal(allidx, np.arange(lat.size))            # a partition
    for p in parts:
        l = np.sort(lat[p])
        assert np.allclose(l, -l[::-1], atol=1e-12)                # each block is mirror symmetric
        # whole latitude classes: no |lat| value is shared between two blocks
    al = [np.unique(np.round(np.abs(lat[p]), 9)) for p in parts]
    assert not (set(al[0]) & set(al[1])) and not (set(al[1]) & set(al[2])) and not (set(al[0]) & set(al[2]))
    sizes = [p.size for p in parts]
    assert max(sizes) - min(sizes) <= 32                            # cuts move to class boundaries (<= 16 columns)
    # a grid without repeated latitudes gets |lat| bands of near-equal size
    lat2 = np.linspace(-80, 85, 50)
    parts = sharding.symmetric_ncol_shards(lat2, 4)
    assert np.array_equal(np.sort(np.concatenate(parts)), np.arange(50))
    assert [p.size for p in parts] == [13, 13, 12, 12]
    assert max(np.abs(lat2[parts[0]])) <= min(np.abs(lat2[parts[1]]))


def test_format_latlon_data_stacks_lat_major():
    """tem_util.py:247-342: (lat, lon) -> ncol = lat*NLON + lon, per-column lat/lon, midpoint bounds."""
    from pytemdiags_amd import LabeledArray
    from pytemdiags_amd.tem_util import format_latlon_data
    lat = np.array([-60.0, 0.0, 45.0])
    lon = np.array([0.0, 90.0, 180.0, 270.0])
    plev = np.array([10.0, 100.0])
    u = np.arange(2 * 3 * 4, dtype=np.float32).reshape(2, 3, 4)            # (plev, lat, lon)
    data = {"lat": LabeledArray(lat, ("lat",)), "lon": LabeledArray(lon, ("lon",)),
            "U": LabeledArray(u, ("plev", "lat", "lon"), {"plev": plev}, name="U", attrs={"units": "m/s"}),
            "P0": LabeledArray(np.array(1e5), ())}
    out = format_latlon_data(data)
    assert out["U"].dims == ("ncol", "plev") and out["U"].shape == (12, 2) and out["U"].dtype == np.float32
    assert out["U"].attrs["units"] == "m/s" and np.array_equal(out["U"].coords["plev"], plev)
    for i in range(3):
        for j in range(4):
            assert np.array_equal(out["U"].values[i * 4 + j], u[:, i, j])
    assert np.array_equal(out["lat"].values, np.repeat(lat, 4)) and out["lat"].dims == ("ncol",)
    assert np.array_equal(out["lon"].values, np.tile(lon, 3))
    assert out["lat_bnds"].dims == ("ncol", "nbnd") and out["lon_bnds"].shape == (12, 2)
    assert np.allclose(out["lat_bnds"].values[0], [-90.0, -30.0])          # first cell: lat[0] -+ diff/2
    assert np.allclose(out["lat_bnds"].values[-1], [22.5, 67.5])           # last cell as wide as the one before
    assert np.allclose(out["lon_bnds"].values[3], [225.0, 315.0])
    assert out["P0"] is data["P0"]
    bad = dict(data, lat_bnds=LabeledArray(np.zeros((3, 2)), ("lat", "bnds")))
    with pytest.raises(RuntimeError, match="does not have dimension nbnd"):
        format_latlon_data(bad)


def test_c_abi_header_is_plain_c_and_links(tmp_path):
    """include/temx.h compiles as C and every entry point resolves against libtemx.so from a C program."""
    import shutil
    import subprocess
    from pytemdiags_amd import _lib
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "link_check")
    libdir = os.path.dirname(_lib.LIB_PATH)
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(root, "include"),
                    os.path.join(root, "tests", "c_abi", "link_check.c"), "-o", exe,
                    "-L", libdir, "-ltemx", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"], check=True)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "temx_version=%d" % _lib.ABI_VERSION in out.stdout and "symbols=%d/%d" % (len(_lib.SIGNATURES), len(_lib.SIGNATURES)) in out.stdout and "null_plan_rc=-1" in out.stdout
    # the header declares exactly what the ctypes table binds
    hdr = open(os.path.join(root, "include", "temx.h")).read()
    import re
    declared = set(re.findall(r"\b(temx_[a-z0-9_]+)\s*\(", hdr))
    assert declared == {n for n, _, _ in _lib.SIGNATURES}


def test_ncio_round_trip(tmp_path):
    """NetCDF-3 writer behind to_netcdf when xarray is absent: dims, dtypes, coordinates, attributes."""
    from pytemdiags_amd import ncio, LabeledArray
    x = np.arange(24.0).reshape(2, 3, 4)
    y = LabeledArray(np.float32([1.5, 2.5]), ("lat",), name="y", attrs={"units": "K"})
    path = ncio.write_dataset(str(tmp_path / "a.nc"), {"x": (("lat", "plev", "time"), x, {"long_name": "test"}),
                                                       "y": (("lat",), y, y.attrs)},
                              {"lat": [-1.0, 1.0], "plev": [1, 2, 3.0]}, attrs={"title": "t"})
    ds = ncio.read_dataset(path)
    assert ds["x"][0] == ("lat", "plev", "time") and ds["x"][1].dtype == np.float64 and np.array_equal(ds["x"][1], x)
    assert ds["y"][1].dtype == np.float32 and np.array_equal(ds["y"][1], [1.5, 2.5])
    assert np.array_equal(ds["lat"][1], [-1.0, 1.0]) and np.array_equal(ds["plev"][1], [1.0, 2.0, 3.0])
    assert "time" not in ds                                 # no coordinate given for it
    with pytest.raises(ValueError):
        ncio.write_dataset(str(tmp_path / "b.nc"), {"x": (("lat", "plev"), x)})


def test_map_cache_file_format_host(tmp_path):
    """ncio writes / reads the map-cache variables of sph_zonal_mean.py:400-417 (host only, NetCDF-3)."""
    from pytemdiags_amd import ncio
    rng = np.random.default_rng(0)
    Y0, Y0inv = rng.standard_normal((40, 7)), rng.standard_normal((7, 40))
    p = ncio.write_dataset(str(tmp_path / "Y0_g_L6.nc"), {
        "Y0": (("ncol", "l"), Y0, {"long_name": "Matrix Y0 for grid g"}),
        "Y0inv": (("l", "ncol"), Y0inv, {"long_name": "Matrix Y0inv for grid g"})})
    d = ncio.read_any(p)
    assert d["Y0"][0] == ("ncol", "l") and d["Y0inv"][0] == ("l", "ncol")
    assert np.array_equal(d["Y0"][1], Y0) and np.array_equal(d["Y0inv"][1], Y0inv)
    with open(p, "rb") as fh:
        assert fh.read(3) == b"CDF"


def test_bench_self_launches_one_rank_per_gpu():
    """`python bench.py --gpus 2` without a launcher starts torch.distributed.run itself (before anything
    touches a GPU) and returns the child's exit code.  Without a GPU every rank stops with the
    no-fallback message, so the exit code is non-zero -- which is what this checks, on CPU."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["HIP_VISIBLE_DEVICES"] = ""                 # also on a GPU box: the ranks must not find a device
    env["CUDA_VISIBLE_DEVICES"] = ""
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                        "--no-cpu-baseline", "--also", ""], capture_output=True, text=True, timeout=300, env=env)
    out = p.stdout + p.stderr
    assert p.returncode != 0, out[-2000:]
    assert out.count("bench.py needs a GPU") >= 2, out[-2000:]      # one message per rank: two ranks were started
