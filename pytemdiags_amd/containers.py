"""Duck-typed labelled arrays for the drop-in front end.

The reference requires ``xarray.DataArray`` inputs (tem_diagnostics.py:312, sph_zonal_mean.py:217).
This front end accepts, and returns the same kind it was given:

* real ``xarray.DataArray`` objects when xarray is importable (any object exposing ``.dims``,
  ``.values`` and ``.coords`` is treated the same way);
* ``LabeledArray`` -- a minimal stand-in defined here (xarray is not installed on the build or
  GPU images);
* raw ``numpy.ndarray`` / ``torch.Tensor`` together with explicit ``plev=`` / ``time=``
  coordinates and a ``dims=`` order.
"""
from __future__ import annotations

import numpy as np

try:  # pragma: no cover - not installed in this image
    import xarray as _xr
except Exception:  # noqa: BLE001
    _xr = None


class LabeledArray:
    """values + dims + coords + name + attrs; just enough of the DataArray surface."""

    def __init__(self, values, dims, coords=None, name=None, attrs=None):
        self.values = values
        self.dims = tuple(dims)
        self.coords = dict(coords or {})
        self.name = name
        self.attrs = dict(attrs or {})
        if len(self.dims) != self.values.ndim:
            raise ValueError("dims %r do not match a %d-d array" % (self.dims, self.values.ndim))

    @property
    def shape(self):
        return tuple(self.values.shape)

    @property
    def dtype(self):
        return self.values.dtype

    @property
    def ndim(self):
        return self.values.ndim

    def __len__(self):
        return self.values.shape[0]

    def __getitem__(self, key):
        if isinstance(key, str):
            c = self.coords.get(key)
            if c is None:
                c = np.arange(self.shape[self.dims.index(key)])
            return LabeledArray(np.asarray(c), (key,), {key: np.asarray(c)}, name=key)
        return self.values[key]

    def __array__(self, dtype=None, copy=None):
        v = self.values
        if hasattr(v, "detach"):
            v = v.detach().cpu().numpy()
        return np.asarray(v, dtype=dtype)

    def astype(self, dt):
        v = self.values.astype(dt) if isinstance(self.values, np.ndarray) else self.values.to(dt)
        return LabeledArray(v, self.dims, self.coords, self.name, self.attrs)

    def __repr__(self):
        return "LabeledArray(name=%r, dims=%r, shape=%r, dtype=%s)" % (self.name, self.dims, self.shape, self.dtype)


def is_labeled(x):
    return hasattr(x, "dims") and hasattr(x, "values")


def is_xarray(x):
    return _xr is not None and isinstance(x, _xr.DataArray)


def coord_of(x, name):
    """1-D coordinate values of a labelled array as float ndarray (index if absent)."""
    try:
        c = x[name]
        c = c.values if hasattr(c, "values") else c
    except Exception:  # noqa: BLE001
        c = np.arange(x.shape[list(x.dims).index(name)])
    if hasattr(c, "detach"):
        c = c.detach().cpu().numpy()
    return np.asarray(c)


def make_like(kind, values, dims, coords, name, attrs=None):
    """Build the output container: kind in {'xarray', 'labeled', 'raw'}."""
    if kind == "raw":
        return values
    if hasattr(values, "detach") and kind == "xarray":
        values = values.detach().cpu().numpy()
    if kind == "xarray" and _xr is not None:
        return _xr.DataArray(values, dims=dims, coords=coords, name=name, attrs=attrs or {})
    return LabeledArray(values, dims, coords, name, attrs)
