"""Pin the CPU oracle (oracle/tem_oracle.py) to golden vectors computed by the reference itself
(tools/make_goldens.py) and to the analytic known-answers of the reference's own test-suite
(PyTEMDiags/tests/tests_sph_zonal_mean.py:331-347, 465-475)."""
import os

import numpy as np
import pytest

from conftest import GOLDEN, fieldnorm_err
from oracle import tem_oracle as orc

TEM_CASES = ["tem_ne4_30x1_f64", "tem_ne4_30x1_f32", "tem_ne4_30x1_desc",
             "tem_ne4_12x3_L20_dlat3", "tem_ne8_20x2_f64", "tem_ne4_16x4_f64"]


def load(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


@pytest.mark.parametrize("case", TEM_CASES)
@pytest.mark.parametrize("mode", ["literal", "factorised"])
def test_tem_oracle_matches_reference_goldens(case, mode):
    g = load(case)
    if mode == "literal" and g["lat"].size > 2000:
        pytest.skip("literal mode is O(N^2); covered on ne4")
    o = orc.TEMOracle(g["ua"], g["va"], g["ta"], g["wap"], g["lat"], g["plev"],
                      L=int(g["L"]), zm_dlat=float(g["zm_dlat"]), mode=mode)
    f32 = g["ua"].dtype == np.float32
    # literal mode repeats the reference's operation order: agreement is at round-off.
    # factorised mode re-associates Y (G^-1 (Y0^T A)): <= 1e-10 field-normalised (fp64).
    tol = {("literal", False): 1e-12, ("factorised", False): 1e-10,
           ("literal", True): 2e-6, ("factorised", True): 2e-5}[(mode, f32)]
    np.testing.assert_allclose(o.lat, g["lat_zm"], rtol=0, atol=0)
    for n in orc.RESULTS:
        r = getattr(o, n)()
        assert r.dtype == g["res_" + n].dtype, n
        assert r.shape == g["res_" + n].shape
        assert fieldnorm_err(r, g["res_" + n]) <= tol, (n, fieldnorm_err(r, g["res_" + n]))
    for n in orc.ZONAL_ATTRS:
        assert getattr(o, n).dtype == g["zm_" + n].dtype, n
        assert fieldnorm_err(getattr(o, n), g["zm_" + n]) <= tol, n
    assert o.theta.dtype == np.float64          # Q5: theta promoted by the fp64 p einsum
    if "nat_up" in g.files:
        for n in orc.NATIVE_ATTRS:
            assert getattr(o, n).dtype == g["nat_" + n].dtype, n
            assert fieldnorm_err(getattr(o, n), g["nat_" + n]) <= tol, n


def test_descending_plev_equals_ascending():
    a, d = load("tem_ne4_30x1_f64"), load("tem_ne4_30x1_desc")
    for n in orc.RESULTS:       # reference flips to ascending p (tem_diagnostics.py:372-382)
        np.testing.assert_array_equal(a["res_" + n], d["res_" + n])


@pytest.mark.parametrize("mode", ["literal", "factorised"])
def test_operator_goldens(mode):
    g = load("op_ne4_L30")
    Z = orc.ZonalAverager(g["lat"], g["lat_out"], int(g["L"]), mode=mode)
    np.testing.assert_allclose(Z.Y0, g["Y0"], rtol=0, atol=1e-15)
    np.testing.assert_allclose(Z.Y0p, g["Y0p"], rtol=0, atol=1e-15)
    for k in ("y20", "y21", "sinlon", "lat2p1", "rand3d", "rand3d_f32"):
        A = g["in_" + k]
        tol = 1e-5 if A.dtype == np.float32 else 1e-11
        zm, zmn = Z.zonal_mean(A), Z.zonal_mean_native(A)
        assert zm.dtype == A.dtype and zmn.dtype == A.dtype
        assert zm.shape == (g["lat_out"].size,) + A.shape[1:]
        den = max(1.0, float(np.max(np.abs(A))))
        assert np.max(np.abs(zm.astype(float) - g["zm_" + k])) <= tol * den, k
        assert np.max(np.abs(zmn.astype(float) - g["zmn_" + k])) <= tol * den, k


@pytest.mark.parametrize("case", ["opw_gauss24x48_L10", "opw_gauss24x48_L70", "opw_ne4_L10"])
def test_weights_mode_goldens(case):
    """`weights` mode, Y0inv = Y0^T diag(4 pi w) (sph_zonal_mean.py:180-181, 383-386): the oracle's
    weights branch against the reference called with ``weights=``."""
    g = load(case)
    Z = orc.ZonalAverager(g["lat"], g["lat_out"], int(g["L"]), weights=g["weights"])
    for k in ("y20", "lat2p1", "rand3d", "rand3d_f32"):
        A = g["in_" + k]
        tol = 1e-5 if A.dtype == np.float32 else 1e-12
        zm, zmn = Z.zonal_mean(A), Z.zonal_mean_native(A)
        assert zm.dtype == A.dtype and zmn.dtype == A.dtype
        assert fieldnorm_err(zm, g["zm_" + k]) <= tol, k
        assert fieldnorm_err(zmn, g["zmn_" + k]) <= tol, k
    np.testing.assert_allclose(Z.Y0inv @ Z.Y0, g["Y0inv_Y0"], rtol=0, atol=1e-12)
    if case == "opw_gauss24x48_L10":      # exact quadrature weights: Y0^T diag(4 pi w) Y0 = I
        assert np.max(np.abs(g["Y0inv_Y0"] - np.eye(11))) < 1e-13


def test_reference_known_answers():
    """tests_sph_zonal_mean.py:465-475: zm(Y_2^0) = Y_2^0(lat_out); zm(lat^2+1) ~ f(lat_out)."""
    g = load("op_ne4_L30")
    Z = orc.ZonalAverager(g["lat"], g["lat_out"], int(g["L"]), mode="factorised")
    y20_out = orc.ylm0_matrix(g["lat_out"], 2)[:, 2]
    assert np.max(np.abs(Z.zonal_mean(g["in_y20"]) - y20_out)) < 1e-12
    f2 = np.deg2rad(g["lat_out"]) ** 2 + 1
    assert np.max(np.abs(Z.zonal_mean(g["in_lat2p1"]) - f2)) < 1e-1   # coarse ne4 grid, L=30: cusp of lat^2 at the poles (the reference itself gives 0.0565)
    # a zonally symmetric field is reproduced on the native grid (test_sph_decomp, :152-293)
    assert np.max(np.abs(Z.zonal_mean_native(g["in_y20"]) - g["in_y20"])) < 1e-12
    d, o = Z.sanity()
    assert abs(d - (int(g["L"]) + 1)) < 1e-9 and abs(o) < 1e-9


def test_recurrence_basis_matches_scipy():
    lat = np.linspace(-90, 90, 721)
    A, B = orc.ylm0_matrix(lat, 50), orc.ylm0_matrix_recurrence(lat, 50)
    assert np.max(np.abs(A - B)) < 1e-12


def test_nan_and_shape_errors():
    g = load("op_ne4_L30")
    Z = orc.ZonalAverager(g["lat"], g["lat_out"], 10, mode="factorised")
    A = g["in_y20"].copy()
    A[3] = np.nan
    with pytest.raises(RuntimeError):
        Z.zonal_mean(A)
    with pytest.raises(RuntimeError):
        Z.zonal_mean(np.zeros(7))


@pytest.mark.parametrize("case", ["tracer_ne4_10x2_f64", "tracer_ne4_10x2_qf32"])
@pytest.mark.parametrize("mode", ["literal", "factorised"])
def test_tracer_oracle_matches_reference_goldens(case, mode):
    """Tracer TEM (Abalos+ 2017): tem_diagnostics.py:532-538, 560-570, 602-611, 801-991."""
    g = load(case)
    nq = int(g["ntrac"])
    q = [g["q%d" % i] for i in range(nq)]
    o = orc.TEMOracle(g["ua"], g["va"], g["ta"], g["wap"], g["lat"], g["plev"], mode=mode, q=q)
    f32 = q[0].dtype == np.float32
    tol = {("literal", False): 1e-12, ("factorised", False): 1e-10,
           ("literal", True): 2e-6, ("factorised", True): 2e-5}[(mode, f32)]
    for n in orc.RESULTS:
        assert fieldnorm_err(getattr(o, n)(), g["res_" + n]) <= tol
    for i in range(nq):
        for n in orc.TRACER_RESULTS:
            r = getattr(o, n)(i)
            assert r.dtype == g["q%d_res_%s" % (i, n)].dtype, n
            assert fieldnorm_err(r, g["q%d_res_%s" % (i, n)]) <= tol, (i, n)
        for n in orc.TRACER_ZONAL + orc.TRACER_NATIVE:
            x = getattr(o, n)[i]
            assert x.dtype == g["q%d_%s" % (i, n)].dtype, n
            assert fieldnorm_err(x, g["q%d_%s" % (i, n)]) <= tol, (i, n)
