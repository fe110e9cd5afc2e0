"""CONTAINER-ONLY labelled-array stand-in used by tools/make_goldens.py.

xarray is not installed in the build image and there is no network.  The reference
(PyTEMDiags) uses xarray purely as a *labelled container*: every number it produces comes
from numpy / scipy calls on ``.values``.  This module provides just enough of that container
surface for the unmodified reference source to import and run, so golden vectors can be
generated from the reference's own arithmetic.  It performs no arithmetic of its own beyond
forwarding operators / ufuncs to numpy on the wrapped ndarray.

It is never imported by the package, the tests, bench.py or anything on the GPU box.
"""
import copy as _copy

import numpy as np


class _Coords(dict):
    pass


class DataArray(np.lib.mixins.NDArrayOperatorsMixin):
    __array_priority__ = 100

    def __init__(self, data, dims=None, coords=None, name=None, attrs=None):
        if isinstance(data, DataArray):
            dims = data.dims if dims is None else dims
            coords = dict(data.coords) if coords is None else coords
            name = data.name if name is None else name
            attrs = dict(data.attrs) if attrs is None else attrs
            data = data._v
        object.__setattr__(self, "_v", np.asarray(data))
        if dims is None:
            dims = tuple("dim_%d" % i for i in range(self._v.ndim))
        object.__setattr__(self, "dims", tuple(dims))
        object.__setattr__(self, "coords", _Coords(coords or {}))
        object.__setattr__(self, "name", name)
        object.__setattr__(self, "attrs", dict(attrs or {}))
        assert len(self.dims) == self._v.ndim, (self.dims, self._v.shape)

    # ---- container surface -------------------------------------------------------------------
    @property
    def values(self):
        return self._v

    @values.setter
    def values(self, v):
        v = np.asarray(v)
        assert v.shape == self._v.shape, (v.shape, self._v.shape)
        object.__setattr__(self, "_v", v)

    @property
    def shape(self):
        return self._v.shape

    @property
    def dtype(self):
        return self._v.dtype

    @property
    def ndim(self):
        return self._v.ndim

    def __len__(self):
        return self._v.shape[0]

    def __getattr__(self, key):
        if key.startswith("_"):
            raise AttributeError(key)
        attrs = object.__getattribute__(self, "attrs")
        if key in attrs:
            return attrs[key]
        raise AttributeError(key)

    def __setattr__(self, key, value):
        if key == "values":
            DataArray.values.fset(self, value)
        else:
            object.__setattr__(self, key, value)

    def _like(self, v, dims=None):
        return DataArray(v, dims=self.dims if dims is None else dims, coords=dict(self.coords),
                         name=self.name, attrs=dict(self.attrs))

    def copy(self, deep=True):
        out = self._like(self._v.copy() if deep else self._v)
        object.__setattr__(out, "coords", _Coords({k: np.array(v, copy=True) for k, v in self.coords.items()}))
        return out

    def __deepcopy__(self, memo):
        return self.copy(deep=True)

    def astype(self, dt):
        return self._like(self._v.astype(dt))

    def transpose(self, *dims):
        perm = [self.dims.index(d) for d in dims]
        return self._like(np.transpose(self._v, perm), dims=tuple(dims))

    def expand_dims(self, name, axis=None):
        axis = self._v.ndim if axis is None else axis
        dims = list(self.dims)
        dims.insert(axis, name)
        return self._like(np.expand_dims(self._v, axis), dims=tuple(dims))

    def reindex(self, indexers):
        out = self
        for dim, new in indexers.items():
            new = np.asarray(new.values if isinstance(new, DataArray) else new)
            old = np.asarray(out.coords[dim])
            idx = np.array([int(np.nonzero(old == v)[0][0]) for v in new])
            ax = out.dims.index(dim)
            res = out._like(np.take(out._v, idx, axis=ax))
            res.coords[dim] = new.copy()
            out = res
        return out

    def isel(self, **kw):
        sl = [slice(None)] * self._v.ndim
        for d, s in kw.items():
            sl[self.dims.index(d)] = s
        out = self._like(self._v[tuple(sl)])
        for d, s in kw.items():
            if d in out.coords:
                out.coords[d] = np.asarray(out.coords[d])[s]
        return out

    def rename(self, mapping):
        dims = tuple(mapping.get(d, d) for d in self.dims)
        out = self._like(self._v, dims=dims)
        object.__setattr__(out, "coords", _Coords({mapping.get(k, k): v for k, v in self.coords.items()}))
        return out

    # ---- indexing ----------------------------------------------------------------------------
    def __getitem__(self, key):
        if isinstance(key, str):
            c = self.coords.get(key)
            if c is None:
                c = np.arange(self._v.shape[self.dims.index(key)])
            return DataArray(np.asarray(c), dims=(key,), coords={key: np.asarray(c)}, name=key)
        v = self._v[key]
        if np.ndim(v) == self._v.ndim:
            return DataArray(v, dims=self.dims, name=self.name, attrs=dict(self.attrs))
        if np.ndim(v) == 0:
            return DataArray(v, dims=())
        # dropped integer-indexed dims
        if not isinstance(key, tuple):
            key = (key,)
        key = key + (slice(None),) * (self._v.ndim - len(key))
        dims = tuple(d for d, kx in zip(self.dims, key) if not isinstance(kx, (int, np.integer)))
        return DataArray(v, dims=dims)

    def __setitem__(self, key, value):
        self._v[key] = value._v if isinstance(value, DataArray) else value

    # ---- numpy protocol ----------------------------------------------------------------------
    def __array__(self, dtype=None, copy=None):
        v = self._v
        if dtype is not None:
            v = v.astype(dtype, copy=False)
        return v

    def __array_ufunc__(self, ufunc, method, *inputs, **kwargs):
        proto = next(x for x in inputs if isinstance(x, DataArray) and x._v.ndim == max(
            (y._v.ndim for y in inputs if isinstance(y, DataArray))))
        raw = tuple(x._v if isinstance(x, DataArray) else x for x in inputs)
        if "out" in kwargs:
            kwargs["out"] = tuple(o._v if isinstance(o, DataArray) else o for o in kwargs["out"])
        res = getattr(ufunc, method)(*raw, **kwargs)
        if isinstance(res, np.ndarray) and res.shape == proto._v.shape:
            return proto._like(res)
        return res

    def __bool__(self):
        return bool(self._v)

    def __repr__(self):
        return "shim.DataArray(name=%r, dims=%r, shape=%r, dtype=%s)" % (
            self.name, self.dims, self._v.shape, self._v.dtype)


class Dataset(dict):
    def to_netcdf(self, *a, **k):
        return None


def merge(objs):
    return Dataset({o.name: o for o in objs})


def open_dataset(path, *a, **k):
    raise FileNotFoundError(path)


DataArray.to_netcdf = lambda self, *a, **k: None

from . import core  # noqa: E402,F401
