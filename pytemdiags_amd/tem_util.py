"""Caller-side data formatting (host only; no numerics of the TEM path live here).

``format_latlon_data`` mirrors PyTEMDiags' ``tem_util.format_latlon_data`` (tem_util.py:247-342):
structured (lat, lon) data are stacked to the unstructured ``ncol`` layout the zonal averager
works on.  The reference takes an ``xarray.Dataset``; this front end takes a mapping
``name -> LabeledArray`` (or anything with ``.dims/.values``) and, when xarray is importable, a
real ``Dataset`` as well, returning the same kind.
"""
from __future__ import annotations

import numpy as np

from .containers import LabeledArray, _xr


def _values(x):
    v = x.values if hasattr(x, "values") else x
    if hasattr(v, "detach"):
        v = v.detach().cpu().numpy()
    return np.asarray(v)


def _cell_bounds(c):
    """Midpoint bounds of a 1-D coordinate, last cell as wide as the one before
    (tem_util.py:306-308, 319-321)."""
    c = np.asarray(c, dtype=np.float64)
    diff = np.diff(np.hstack([c, c[-1] + (c[-1] - c[-2])]))
    return np.vstack([c - diff / 2, c + diff / 2]).T


def format_latlon_data(data, lat_name="lat", lon_name="lon", latbnd_name="lat_bnds",
                       lonbnd_name="lon_bnds", bnddim_name="nbnd"):
    """Stack the (lat, lon) dimensions of every variable into a leading ``ncol`` dimension of
    length NLAT*NLON (lat-major, like ``Dataset.stack(ncol=(lat, lon))``), keep the per-column
    latitudes and longitudes as variables ``lat``/``lon`` on ``ncol``, and add cell-bound
    variables when the input has none.  Same arguments and errors as the reference."""
    if _xr is not None and isinstance(data, _xr.Dataset):  # pragma: no cover - xarray absent here
        as_map = {k: LabeledArray(v.values, v.dims, name=k, attrs=dict(v.attrs)) for k, v in data.variables.items()}
        out = format_latlon_data(as_map, lat_name, lon_name, latbnd_name, lonbnd_name, bnddim_name)
        return _xr.Dataset({k: (v.dims, v.values, v.attrs) for k, v in out.items()})

    data = dict(data)
    if lat_name not in data or lon_name not in data:
        raise KeyError("format_latlon_data needs the coordinate variables %r and %r" % (lat_name, lon_name))
    lat = _values(data[lat_name]).astype(np.float64).ravel()
    lon = _values(data[lon_name]).astype(np.float64).ravel()
    nlat, nlon = lat.size, lon.size

    for bname, coord, cname in ((latbnd_name, lat, lat_name), (lonbnd_name, lon, lon_name)):
        if bname not in data:
            data[bname] = LabeledArray(_cell_bounds(coord), (cname, bnddim_name), name=bname)
        elif bnddim_name not in tuple(data[bname].dims):
            raise RuntimeError(
                "Variable {} does not have dimension {}. Dimensions are: {}. Did you specify the "
                "latbnd_name, lonbnd_name, and bnddim_name arguments to format_latlon_data() "
                "correctly?".format(bname, bnddim_name, tuple(data[bname].dims)))

    out = {}
    for name, var in data.items():
        if name in (lat_name, lon_name):
            continue
        if not hasattr(var, "dims"):
            raise RuntimeError("variable %r has no dims; wrap it in a LabeledArray" % name)
        dims = tuple(var.dims)
        if lat_name not in dims and lon_name not in dims:
            out[name] = var
            continue
        v = _values(var)
        # broadcast a variable that has only one of the two dims (e.g. lat_bnds) over the other
        if lat_name not in dims:
            v, dims = np.broadcast_to(v[None], (nlat,) + v.shape), (lat_name,) + dims
        if lon_name not in dims:
            v, dims = np.broadcast_to(v[None], (nlon,) + v.shape), (lon_name,) + dims
        order = [dims.index(lat_name), dims.index(lon_name)] + [i for i, d in enumerate(dims)
                                                               if d not in (lat_name, lon_name)]
        v = np.transpose(v, order)
        if v.shape[0] != nlat or v.shape[1] != nlon:
            raise RuntimeError("variable %r does not match the lat/lon coordinates" % name)
        rest = tuple(dims[i] for i in order[2:])
        coords = {d: c for d, c in getattr(var, "coords", {}).items() if d in rest}
        out[name] = LabeledArray(np.ascontiguousarray(v.reshape((nlat * nlon,) + v.shape[2:])),
                                 ("ncol",) + rest, coords, name=getattr(var, "name", name),
                                 attrs=getattr(var, "attrs", None))
    out[lat_name] = LabeledArray(np.repeat(lat, nlon), ("ncol",), name=lat_name)
    out[lon_name] = LabeledArray(np.tile(lon, nlat), ("ncol",), name=lon_name)
    return out
