"""NetCDF output without xarray: the writers behind ``TEMDiagnostics.to_netcdf`` / ``q_to_netcdf``
(tem_diagnostics.py:995-1103) when xarray is not importable.

Files are NetCDF-3 (64-bit offset) written with ``scipy.io.netcdf_file`` -- the format xarray itself
falls back to when netCDF4 is missing -- with the reference's variable names, dimension order
``(lat | ncol, plev, time)`` and coordinate variables.  Host-side I/O only; no numerics.
"""
from __future__ import annotations

import numpy as np


def _np(x):
    v = x.values if hasattr(x, "values") and hasattr(x, "dims") else x
    if hasattr(v, "detach"):
        v = v.detach().cpu().numpy()
    return np.ascontiguousarray(v)


def write_dataset(path, variables, coords=None, attrs=None):
    """``variables``: name -> (dims, values[, attrs]); ``coords``: dim name -> 1-D values."""
    from scipy.io import netcdf_file
    coords = dict(coords or {})
    with netcdf_file(path, "w", version=2) as nc:
        for k, v in (attrs or {}).items():
            setattr(nc, k, v)
        sizes = {}
        for name, spec in variables.items():
            dims, vals = spec[0], _np(spec[1])
            if len(dims) != vals.ndim:
                raise ValueError("variable %r: dims %r do not match shape %r" % (name, dims, vals.shape))
            for dn, n in zip(dims, vals.shape):
                if sizes.setdefault(dn, n) != n:
                    raise ValueError("dimension %r has two lengths (%d, %d)" % (dn, sizes[dn], n))
        for dn, n in sizes.items():
            nc.createDimension(dn, n)
        for dn, cv in coords.items():
            if dn in sizes and dn not in variables:
                cv = np.asarray(_np(cv), dtype=np.float64)
                var = nc.createVariable(dn, "d", (dn,))
                var[:] = cv
        for name, spec in variables.items():
            dims, vals = spec[0], _np(spec[1])
            if vals.dtype not in (np.float32, np.float64):
                vals = vals.astype(np.float64)
            var = nc.createVariable(name, "f" if vals.dtype == np.float32 else "d", tuple(dims))
            var[:] = vals
            for k, v in (spec[2] if len(spec) > 2 and spec[2] else {}).items():
                if isinstance(v, (str, int, float)):
                    setattr(var, k, v)
    return path


def read_dataset(path):
    """name -> (dims, ndarray) of every variable in a NetCDF-3 file (tests, round trips)."""
    from scipy.io import netcdf_file
    out = {}
    with netcdf_file(path, "r", mmap=False) as nc:
        for name, var in nc.variables.items():
            a = np.array(var[:])
            out[name] = (tuple(var.dimensions), a.astype(a.dtype.newbyteorder("=")))
    return out


def read_any(path):
    """Like :func:`read_dataset`, for NetCDF-3 (scipy) and -- when xarray is importable -- NetCDF-4 files
    (the reference writes its map cache through xarray, sph_zonal_mean.py:400-417)."""
    with open(path, "rb") as fh:
        magic = fh.read(4)
    if magic[:3] == b"CDF":
        return read_dataset(path)
    import xarray as xr                      # NetCDF-4 / HDF5: needs xarray + netCDF4 or h5netcdf
    with xr.open_dataset(path) as ds:
        return {k: (tuple(v.dims), np.asarray(v.values)) for k, v in ds.variables.items()}
