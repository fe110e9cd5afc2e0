"""Deterministic synthetic grids and fields for tests and benchmarks (host / numpy side).

Nothing here is on the product path: it only manufactures inputs of the shapes
named in BASELINE.json.

* ``cubed_sphere_gll(ne)``  -- unique GLL (np=4) nodes of an equiangular cubed sphere,
  6*ne^2*9+2 points (866 / 3458 / 48602 / 777602 / 3110402 for ne4/8/30/120/240).
* ``pressure_levels(nlev)`` -- log-spaced 1..1000 hPa, top -> bottom.
* ``analytic_fields(...)``  -- stably stratified T, jet-like u, wavy v / omega
  (+ optional Gaussian noise from ``np.random.default_rng``).

The device-side generator with the same analytic part (noise from a counter hash)
lives in csrc/synth.hip and is exposed as ``temx_synth_fields``.
"""
from __future__ import annotations

import numpy as np

__all__ = ["cubed_sphere_gll", "pressure_levels", "analytic_fields", "analytic_tracer", "ncol_of_ne"]


def ncol_of_ne(ne: int) -> int:
    return 6 * ne * ne * 9 + 2


def cubed_sphere_gll(ne: int, mirror: bool = True):
    """Return (lat_deg, lon_deg) of the unique np=4 GLL nodes of an ne x ne x 6 cubed sphere.

    Points are sorted lexicographically by rounded (x, y, z) so the order is
    deterministic but *unstructured* with respect to latitude.  ``mirror=False`` leaves the
    latitudes as the construction gives them: equal across a latitude class only up to
    round-off (~1e-12 degrees at ne240), the state real grid files are in.
    """
    gll = np.array([-1.0, -1.0 / np.sqrt(5.0), 1.0 / np.sqrt(5.0), 1.0])
    edges = np.linspace(-np.pi / 4, np.pi / 4, ne + 1)
    mid = 0.5 * (edges[1:] + edges[:-1])
    half = 0.5 * (edges[1:] - edges[:-1])
    ang = (mid[:, None] + half[:, None] * gll[None, :]).ravel()
    ang = np.unique(np.round(ang, 14))
    assert ang.size == 3 * ne + 1
    t = np.tan(ang)
    X, Y = np.meshgrid(t, t, indexing="ij")
    X = X.ravel()
    Y = Y.ravel()
    one = np.ones_like(X)
    faces = [
        (one, X, Y), (-one, -X, Y), (-X, one, Y),
        (X, -one, Y), (-Y, X, one), (Y, X, -one),
    ]
    P = np.concatenate([np.stack(f, axis=1) for f in faces], axis=0)
    P /= np.linalg.norm(P, axis=1, keepdims=True)
    P = np.unique(np.round(P, 12), axis=0)
    assert P.shape[0] == ncol_of_ne(ne), (P.shape, ncol_of_ne(ne))
    P /= np.linalg.norm(P, axis=1, keepdims=True)
    # The ideal grid is mirror symmetric about the equator, but the 12-digit rounding above resolves
    # the two face-copies of an edge node independently in each hemisphere (1e-12-level asymmetry at
    # ne240).  Rebuild the south as the exact mirror of the north so latitudes are bitwise +-.
    if mirror:
        north = P[P[:, 2] > 1e-9]
        eq = P[np.abs(P[:, 2]) <= 1e-9].copy()
        eq[:, 2] = 0.0
        eq /= np.linalg.norm(eq, axis=1, keepdims=True)
        assert 2 * north.shape[0] + eq.shape[0] == P.shape[0]
        P = np.concatenate([north, north * np.array([1.0, 1.0, -1.0]), eq], axis=0)
    P = P[np.lexsort((np.round(P[:, 2], 12), np.round(P[:, 1], 12), np.round(P[:, 0], 12)))]
    lat = np.rad2deg(np.arcsin(np.clip(P[:, 2], -1.0, 1.0)))
    lon = np.mod(np.rad2deg(np.arctan2(P[:, 1], P[:, 0])), 360.0)
    return lat, lon


def pressure_levels(nlev: int) -> np.ndarray:
    """plev in hPa, ascending pressure (model top first)."""
    return np.exp(np.linspace(np.log(1.0), np.log(1000.0), nlev))


def analytic_fields(lat_deg, lon_deg, plev_hpa, nt, noise=0.1, seed=0, dtype=np.float64):
    """Return (ua, va, ta, wap), each ``[ncol][nlev][nt]`` C-contiguous.

    Noise is drawn in the order u, v, T, omega from ``default_rng(seed)``.
    """
    phi = np.deg2rad(np.asarray(lat_deg, dtype=np.float64))[:, None, None]
    lam = np.deg2rad(np.asarray(lon_deg, dtype=np.float64))[:, None, None]
    p = np.asarray(plev_hpa, dtype=np.float64)[None, :, None]
    t = np.arange(nt, dtype=np.float64)[None, None, :]
    z = -7.0 * np.log(p / 1000.0)
    s, c = np.sin(phi), np.cos(phi)
    T = (300.0 - 60.0 * s**2 - 6.5 * np.minimum(z, 12.0) + 2.0 * np.maximum(z - 20.0, 0.0)
         + 3.0 * np.cos(3 * lam + 0.3 * t) * c)
    u = (30.0 * np.sin(2 * phi) ** 2 * np.exp(-((z - 12.0) / 8.0) ** 2)
         + 8.0 * np.sin(4 * lam + 0.2 * t) * c**2 + 0.0 * z)
    v = (6.0 * np.cos(4 * lam + 0.2 * t) * c**2 * np.exp(-((z - 10.0) / 10.0) ** 2)
         + 0.5 * np.sin(2 * phi) + 0.0 * t)
    w = (0.05 * np.sin(4 * lam + 0.2 * t + 0.7) * c**2 + 0.01 * np.cos(3 * phi) + 0.0 * z)
    shape = (phi.shape[0], p.shape[1], nt)
    u, v, T, w = (np.broadcast_to(a, shape).copy() for a in (u, v, T, w))
    if noise:
        rng = np.random.default_rng(seed)
        for a in (u, v, T, w):
            a += noise * rng.standard_normal(shape)
    return tuple(np.ascontiguousarray(a.astype(dtype)) for a in (u, v, T, w))


def analytic_tracer(lat_deg, lon_deg, plev_hpa, nt, which=0, noise=0.02, seed=100, dtype=np.float64):
    """A smooth positive mixing ratio ``[ncol][nlev][nt]`` (kg/kg-like magnitudes) for tracer TEM tests."""
    phi = np.deg2rad(np.asarray(lat_deg, dtype=np.float64))[:, None, None]
    lam = np.deg2rad(np.asarray(lon_deg, dtype=np.float64))[:, None, None]
    p = np.asarray(plev_hpa, dtype=np.float64)[None, :, None]
    t = np.arange(nt, dtype=np.float64)[None, None, :]
    z = -7.0 * np.log(p / 1000.0)
    q = (1e-3 * (1.0 + 0.5 * np.sin((which + 1) * phi)) * np.exp(-z / (7.0 + 3.0 * which))
         * (1.0 + 0.1 * np.cos((2 + which) * lam + 0.1 * t) * np.cos(phi)))
    if noise:
        rng = np.random.default_rng(seed + which)
        q = q * (1.0 + noise * rng.standard_normal(q.shape))
    return np.ascontiguousarray(q.astype(dtype))
