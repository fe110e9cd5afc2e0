"""Drop-in front end for ``PyTEMDiags.sph_zonal_averager`` (reference: PyTEMDiags/sph_zonal_mean.py).

Same constructor signature, attributes, methods and error behaviour; the arithmetic runs in the
HIP engine (libtemx.so) through the factorised operator ``Y (G^-1 (Y0^T A))`` instead of the
reference's dense ``lstsq(Y0, I_N)`` and N x N products (sph_zonal_mean.py:389, :251).
There is no CPU fallback: without the built extension and a GPU the constructor of the plan raises.
"""
from __future__ import annotations

import pathlib

import numpy as np

from . import _lib, containers

SAVE_DEST = "{}/../maps".format(pathlib.Path(__file__).parent.resolve())     # sph_zonal_mean.py:26
DEFAULT_LAT_ATTRS = {"long_name": "Latitude of Grid Cell Centers", "standard_name": "latitude",
                     "units": "degrees_north", "axis": "Y"}                     # sph_zonal_mean.py:27-28


def _as_numpy(x):
    if hasattr(x, "values") and not isinstance(x, np.ndarray):
        x = x.values
    if hasattr(x, "detach"):
        x = x.detach().cpu().numpy()
    return np.asarray(x)


class sph_zonal_averager:
    """Zonal averaging on an unstructured grid by projection on the Y_l^0 (sph_zonal_mean.py:35).

    Extra keyword (not in the reference): ``device`` -- CUDA device index (default 0).

    Map cache (sph_zonal_mean.py:329-345, 400-417).  The cache file names are derived exactly like the
    reference (:165-174) and the files hold the same variables -- ``Y0[ncol, l]`` and ``Y0inv[l, ncol]``
    in ``Y0_{grid}_L{L}.nc``, ``Y0p[ncol, l]`` in ``Y0p_{grid}_{gridout}_L{L}.nc``.  Two deliberate
    differences: (1) the cache is used only when ``save_dest`` is given -- the reference defaults to a
    ``maps`` directory next to the package and always writes, but the dense ``Y0inv`` is K x N doubles
    (317 MB at ne120) that this engine never needs, and rebuilding the basis on the device takes
    < 1 ms; (2) without netCDF4 / xarray the files are NetCDF-3 (``scipy.io``), which xarray reads
    back, while a NetCDF-4 file written by the reference can only be read here when xarray is
    importable.  Matrices read from the cache are checked against the device-built basis; a file
    that belongs to another grid or L is reported and ignored (the reference would use it silently).
    """

    def __init__(self, lat, lat_out, L, weights=None, grid_name=None, grid_out_name=None,
                 ncoldim="ncol", overwrite=False, save_dest=None, debug=False, logfile=None,
                 device=None, fp32_fields=False):
        self.L = L
        # (not in the reference's signature) the arrays to be averaged are fp32: latitudes that agree to 1e-8 degrees
        # share a basis row (include/temx.h, TEMX_LAT_TOL_F32)
        self._fp32_fields = bool(fp32_fields)
        self.lat = _as_numpy(lat)                    # sph_zonal_mean.py:148-151
        self.lat_out = _as_numpy(lat_out)
        # the caller's weights as float64 (sum to 1); the reference scales its attribute by 4 pi only AFTER the
        # cache probe below (:177-181) -- the engine always takes the unscaled ones
        self._w_raw = None if weights is None else _as_numpy(weights).astype(np.float64)
        self.weights = self._w_raw
        self.grid_name = grid_name
        self.grid_out_name = grid_out_name
        self.save_dest = save_dest
        self.ncoldim = ncoldim
        self.debug = debug
        self.logfile = logfile
        self.device = 0 if device is None else device

        self._cache_on = save_dest is not None       # see the class docstring
        self.N = len(self.lat)                       # :154
        self.M = len(self.lat_out)                   # :155
        self.l = np.arange(L + 1)                    # :156
        self.diagw = None
        self._plan = None
        self._cache = {}
        if self.save_dest is None:
            self.save_dest = SAVE_DEST
        if self.grid_name is None:
            self.grid_name = "ncol{}".format(self.N)
        self.Y0_file_out = "{}/Y0_{}_L{}.nc".format(self.save_dest, self.grid_name, self.L)
        if self.grid_out_name is None:
            dlat_out = np.diff(self.lat_out)[0]
            self.grid_out_name = "{}deg".format(dlat_out)
        self.Y0p_file_out = "{}/Y0p_{}_{}_L{}.nc".format(self.save_dest, self.grid_name,
                                                       self.grid_out_name, self.L)
        # the reference probes the map cache here (read_only=True, :177); there is no cache
        self.sph_compute_matrices(read_only=True, overwrite=overwrite)
        # scale grid weights to unit sphere surface area (:180-181); not in place
        if self._w_raw is not None:
            self.weights = self._w_raw * (4 * np.pi)

    # ---- matrices (attributes Y0, Y0inv, Y0p of the reference, fetched lazily from the device) ----
    def _matrix(self, which):
        if self._plan is None:
            return None
        if which not in self._cache:
            self._cache[which] = self._plan.matrix(which).cpu().numpy()
        return self._cache[which]

    @property
    def Y0(self):
        return self._matrix(_lib.MAT_Y0)

    @property
    def Y0p(self):
        return self._matrix(_lib.MAT_Y0P)

    @property
    def Y0inv(self):
        if self._plan is None:
            return None
        if self._w_raw is not None:                  # Y0inv = Y0^T diag(4 pi w)  (:385, weights scaled at :181)
            return self.Y0.T * (self._w_raw * (4 * np.pi))[None, :]
        return self._matrix(_lib.MAT_Y0INV)

    # ---- map cache files (sph_zonal_mean.py:329-345, 400-417) ----
    def _read_map_cache(self):
        """(Y0, Y0inv, Y0p) from the cache files, or None (missing, unreadable or of another shape)."""
        import os
        from . import ncio
        if not (os.path.isfile(self.Y0_file_out) and os.path.isfile(self.Y0p_file_out)):
            return None
        try:
            a, b = ncio.read_any(self.Y0_file_out), ncio.read_any(self.Y0p_file_out)
            Y0, Y0inv, Y0p = a["Y0"][1], a["Y0inv"][1], b["Y0p"][1]
        except Exception:   # noqa: BLE001 - an unreadable cache is no cache (e.g. NetCDF-4 without netCDF4)
            return None
        K = self.L + 1
        if Y0.shape != (self.N, K) or Y0inv.shape != (K, self.N) or Y0p.shape != (self.M, K):
            return None
        return Y0, Y0inv, Y0p

    def _write_map_cache(self):
        import os
        from . import ncio
        os.makedirs(self.save_dest, exist_ok=True)
        ncio.write_dataset(self.Y0_file_out, {
            "Y0": (("ncol", "l"), self.Y0, {"long_name": "Matrix Y0 for grid {}".format(self.grid_name)}),
            "Y0inv": (("l", "ncol"), self.Y0inv, {"long_name": "Matrix Y0inv for grid {}".format(self.grid_name)})})
        ncio.write_dataset(self.Y0p_file_out, {
            "Y0p": (("ncol", "l"), self.Y0p, {"long_name": "Matrix Y0p for grid {}".format(self.grid_out_name)})})

    def sph_compute_matrices(self, overwrite=False, read_only=False, no_write=False):
        """Build Y0, Y0p and the (factorised) inverse on the device (sph_zonal_mean.py:302-422); with a
        ``save_dest``, read / write the reference's map cache files around it."""
        import os
        if self._cache_on and overwrite:             # :332-334
            for fn in (self.Y0_file_out, self.Y0p_file_out):
                if os.path.isfile(fn):
                    os.remove(fn)
        cached = self._read_map_cache() if self._cache_on else None
        if read_only and cached is None:
            return                                   # no cache on file (:343-345)
        from . import engine
        weighted = self._w_raw is not None
        if weighted and len(self._w_raw) != len(self.lat):
            raise RuntimeError("number of weights must equal number of native grid latitudes!")   # :353-354
        if self._plan is not None:
            self._plan.close()
        self._cache = {}
        self._plan = engine.Plan(self.lat, self.lat_out, self.L, device=self.device, defer_finalize=weighted,
                                 fp32_fields=self._fp32_fields)
        if weighted:
            self._plan.set_weights(self._w_raw)
        if cached is not None:
            # the engine applies its own factorisation of the same operator (built above for THIS grid, L and
            # weights); a cache is accepted only if it describes them too.  Quadrature weights do not make
            # Y0inv Y0 the identity, so in weights mode the cached Y0inv is compared with Y0^T diag(4 pi w).
            ok = np.max(np.abs(cached[0] - self.Y0)) <= 1e-9 and np.max(np.abs(cached[2] - self.Y0p)) <= 1e-9
            if ok and weighted:
                ref = self.Y0inv
                ok = np.max(np.abs(cached[1] - ref)) <= 1e-9 * max(1.0, float(np.max(np.abs(ref))))
            elif ok:
                ok = np.max(np.abs(cached[1] @ cached[0] - np.eye(self.L + 1))) <= 1e-6
            if ok:
                self._cache[_lib.MAT_Y0], self._cache[_lib.MAT_Y0P] = cached[0], cached[2]
                if not weighted:
                    self._cache[_lib.MAT_Y0INV] = cached[1]
                self.map_cache_used = True
                return
            import warnings
            warnings.warn("map cache {} does not match this grid / L / weights; ignored (the operator was "
                          "built from the arguments)".format(self.Y0_file_out))
        self.map_cache_used = False
        if self._cache_on and not no_write and not read_only:
            self._write_map_cache()                  # :400-417

    def sanity_check(self):
        """(sum(diag(Y0inv Y0)), sum(offdiag)) -- the numbers the reference prints (:393-398)."""
        P = self._matrix(_lib.MAT_GINV) @ self._matrix(_lib.MAT_GRAM)
        d = float(np.sum(np.diagonal(P)))
        return d, float(np.sum(P) - d)

    # ---- the operator ----
    def _sph_zonal_mean_generic(self, A, native):
        """sph_zonal_mean.py:187-283."""
        import torch
        if self._plan is None:
            raise RuntimeError("Matrices Y0, Y0inv, and/or Y0p are undefined; either verify grid_name,"
                               "grid_name_out, and save_dest, or call sph_compute_matrices()"
                               "before sph_zonal_mean() or sph_zonal_mean_native()!")        # :213-216
        labeled = containers.is_labeled(A)
        name = getattr(A, "name", None) if labeled else None
        if name is None:
            name = "{unnamed variable}"                                                      # :228-229
        vals = A.values if labeled else A
        if labeled:
            dims = tuple(A.dims)
            if dims[0] != self.ncoldim or vals.shape[0] != self.N:                            # :234-237
                raise RuntimeError("(sph_zonal_mean_generic() Expected the first (leftmost) "
                                   "dimension of variable {} to be {} of length {}".format(
                                       name, self.ncoldim, self.N))
        elif vals.shape[0] != self.N:
            raise RuntimeError("(sph_zonal_mean_generic() Expected the first (leftmost) "
                               "dimension of variable {} to be {} of length {}".format(
                                   name, self.ncoldim, self.N))
        is_torch = isinstance(vals, torch.Tensor)
        dev = self._plan.device
        t = vals if is_torch else torch.as_tensor(np.ascontiguousarray(vals))
        in_dtype = t.dtype
        if t.dtype not in (torch.float64, torch.float32):
            t = t.to(torch.float64)
        out = self._plan.zonal_mean(t.to(dev), native=native)
        if self._plan.status():                                                               # :219-221
            raise RuntimeError("Variable {} has nans! Spectral zonal averager cannot handle nans; "
                               "please replace or remove them".format(name))
        # cast the zonal mean back to the precision of the input data (:277-282)
        out = out.to(in_dtype) if in_dtype.is_floating_point else out
        if is_torch:
            out_vals = out.to(vals.device)
        else:
            out_vals = out.cpu().numpy()
        if not labeled:
            return out_vals
        kind = "xarray" if containers.is_xarray(A) else "labeled"
        attrs = dict(getattr(A, "attrs", {}) or {})
        coords = {k: v for k, v in dict(getattr(A, "coords", {}) or {}).items() if k != self.ncoldim}
        if not native:                                                                        # :267-273
            dims = ("lat",) + dims[1:]
            coords = dict(coords)
            coords["lat"] = self.lat_out
            attrs = dict(DEFAULT_LAT_ATTRS)
        attrs["long_name"] = "zonal mean of {}".format(name)                                 # :275
        return containers.make_like(kind, out_vals, dims, coords, name, attrs)

    def sph_zonal_mean_native(self, A):          # sph_zonal_mean.py:285-290
        return self._sph_zonal_mean_generic(A, True)

    def sph_zonal_mean(self, A):                 # sph_zonal_mean.py:291-296
        return self._sph_zonal_mean_generic(A, False)
