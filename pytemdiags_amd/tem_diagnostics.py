"""Drop-in front end for ``PyTEMDiags.TEMDiagnostics`` (reference: PyTEMDiags/tem_diagnostics.py).

Same constructor, methods, properties and error behaviour.  All numerics -- potential
temperature, the 7 + 4 zonal means, eddy products, p / lat derivatives, psi, the cumulative
pressure integral and the ten GM16 Table-A1 diagnostics -- run on the MI355X in the HIP engine
(libtemx.so); this module only validates, reshapes and labels.

Input kinds (the output kind follows the input kind):
  * labelled arrays -- ``xarray.DataArray`` when xarray is importable, or anything with
    ``.dims`` / ``.values`` and a ``plev`` coordinate (``containers.LabeledArray``);
  * raw ``numpy.ndarray`` / ``torch.Tensor`` laid out as ``dims`` (default
    ``(ncol, plev, time)``) with explicit ``plev=`` [hPa] and optional ``time=``.
"""
from __future__ import annotations

import os
import warnings

import numpy as np

from . import _lib, containers
from .constants import P0, Om
from .sph_zonal_mean import sph_zonal_averager

DEFAULT_DIMS = {"horz": "ncol", "vert": "plev", "time": "time"}        # tem_diagnostics.py:25

# dtype each stored quantity has in the reference when the inputs are not fp64:
# theta is promoted to fp64 by the fp64 pressure einsum (tem_diagnostics.py:498); everything
# downstream of theta, of cos(lat) products or of p_integral is fp64; the rest keeps the input dtype.
_F64_ALWAYS = {"thetab", "vptpb", "dthetab_dp", "ubcoslat", "dubcoslat_dlat", "psi", "psicoslat",
               "dpsicoslat_dlat", "dpsi_dp", "int_vbdp", "thetap", "vptp", "theta"}


class TEMDiagnostics:
    def __init__(self, ua, va, ta, wap, lat_native, q=None, p0=P0, zm_dlat=1, L=50,
                 dim_names=DEFAULT_DIMS, grid_name=None, zm_grid_name=None, map_save_dest=None,
                 overwrite_map=False, zm_pole_points=False, debug_level=1, logfile=None,
                 *, plev=None, time=None, dims=None, device=None):
        # ---- arguments (tem_diagnostics.py:217-236) ----
        self.p0 = p0
        self.q = q
        self.ntrac = None
        self.lat_native = lat_native
        self.L = L
        self.zm_dlat = zm_dlat
        self.dim_names = dim_names
        self.zm_pole_points = zm_pole_points
        self.grid_name = grid_name
        self.zm_grid_name = zm_grid_name
        self.map_save_dest = map_save_dest
        self.overwrite_map = overwrite_map
        self.debug_level = debug_level
        self.logfile = logfile
        self._device = 0 if device is None else device
        self._raw_dims = dims
        self._raw_plev, self._raw_time = plev, time
        self._in = {"ua": ua, "va": va, "ta": ta, "wap": wap}

        self._config_dims()

        # ---- zonal averaging object (tem_diagnostics.py:243-249) ----
        self.ZM = sph_zonal_averager(self._lat_native_np, self._lat_zm, self.L, grid_name=grid_name,
                                     grid_out_name=zm_grid_name, save_dest=map_save_dest,
                                     debug=debug_level > 1, overwrite=overwrite_map, device=self._device,
                                     fp32_fields=str(self._work_dtype) == "torch.float32")
        if self.ZM.Y0 is None or self.ZM.Y0p is None:
            self.ZM.sph_compute_matrices(overwrite=overwrite_map)
        self._zonal_mean = self.ZM.sph_zonal_mean

        # ---- the whole numeric pipeline: one engine call (tem_diagnostics.py:252-259) ----
        plan = self.ZM._plan
        plan.set_tem(self.NLEV, self.NT, self._p_np, float(self.p0))
        # (class-sum forms: the first tracer, when there is one, shares the sweep of the fields, temx_tem_tracer_run;
        #  single sweep: the tracers follow the TEM run in pairs, temx_tracers_run)
        fused = None
        if self.ntrac and not plan.single_sweep:
            self._res, self._zon, *fused = plan.tem_tracer_run(*self._dev_fields, self._dev_q[0], want_zonal=True)
        else:
            self._res, self._zon = plan.tem_run(*self._dev_fields, want_zonal=True)
        if plan.status():                                   # sph_zonal_mean.py:219-221
            raise RuntimeError("Variable has nans! Spectral zonal averager cannot handle nans; "
                               "please replace or remove them")
        self._eddy = None
        self._theta = None
        self._out_file = None
        # ---- tracers (tem_diagnostics.py:532-538, 560-570, 602-611): one engine call each ----
        self._tres, self._tzon, self._teddy = [], [], [None] * self.ntrac
        self._last_tracer = None
        if self.ntrac and plan.single_sweep:
            # the list of tracers in one engine call: two per sweep, (q1, q2, v, omega) read once
            for tres, tzon in plan.tracers_run(self._dev_q, self._dev_fields[1], self._dev_fields[3], want_zonal=True):
                self._tres.append(tres)
                self._tzon.append(tzon)
            self._last_tracer = self.ntrac - 1
        else:
            for i in range(self.ntrac):
                if i == 0 and fused is not None:
                    tres, tzon = fused
                else:
                    tres, tzon = plan.tracer_run(self._dev_q[i], self._dev_fields[1], self._dev_fields[3], want_zonal=True)
                self._tres.append(tres)
                self._tzon.append(tzon)
                self._last_tracer = i
        if self.ntrac and plan.status():
            raise RuntimeError("Variable has nans! Spectral zonal averager cannot handle nans; "
                               "please replace or remove them")

    # ------------------------------------------------------------------------------------------
    def _config_dims(self):
        """Validation and reshaping of tem_diagnostics.py:266-405 (host side only)."""
        import torch
        self.ncolname = self.dim_names["horz"]
        self.plevname = self.dim_names["vert"]
        try:
            self.timename = self.dim_names["time"]
        except KeyError:
            self.timename = DEFAULT_DIMS["time"]
        self.data_dims = (self.ncolname, self.plevname, self.timename)

        # tracers (tem_diagnostics.py:281-301): a labelled array / raw array, or a list of them
        if self.q is not None:
            if not isinstance(self.q, list):
                self.q = [self.q]
            ok = all(containers.is_labeled(x) or isinstance(x, (np.ndarray, torch.Tensor)) for x in self.q)
            if not ok or len(self.q) == 0:
                raise RuntimeError("tracers q must be passed as an xarray DataArray, or"
                                   "a list of xarray DataArrays")
            self.ntrac = len(self.q)
        else:
            self.q = []
            self.ntrac = 0
        self._q_out_file = [None] * self.ntrac
        for i, x in enumerate(self.q):
            self._in["q{}".format(i)] = x

        lat = self.lat_native
        self._lat_native_np = np.asarray(lat.values if containers.is_labeled(lat) else
                                         (lat.detach().cpu().numpy() if hasattr(lat, "detach") else lat),
                                         dtype=np.float64)
        nlat = self._lat_native_np.shape[0]

        labeled = [containers.is_labeled(v) for v in self._in.values()]
        if any(labeled) and not all(labeled):
            raise RuntimeError("Input data for args ua, va, ta, wap (and q) must all be of the same kind")
        self._kind = "raw"
        if all(labeled):
            self._kind = "xarray" if containers.is_xarray(self._in["ua"]) else "labeled"

        vals = {}
        for var, dat in self._in.items():
            if self._kind == "raw":
                if not (isinstance(dat, np.ndarray) or isinstance(dat, torch.Tensor)):
                    raise RuntimeError("Input data for arg '{}' must be an xarray DataArray".format(var))   # :313
                ddims = tuple(self._raw_dims) if self._raw_dims is not None else self.data_dims[:dat.ndim]
                v = dat
            else:
                ddims = tuple(dat.dims)
                v = dat.values
            if self.ncolname not in ddims:                                                    # :316-318
                raise RuntimeError("Input data {} does not contain dim {}".format(var, self.ncolname))
            if v.shape[ddims.index(self.ncolname)] != nlat:                                   # :320-323
                raise RuntimeError("Dimension {} in variable {} is length {}, but input parameter lat is "
                                   "length {}; these must match!".format(
                                       self.ncolname, var, v.shape[ddims.index(self.ncolname)], nlat))
            if len(ddims) < 2 or len(ddims) > 3:                                              # :326-329
                raise RuntimeError("Input data has {0} dims, expected either 2 ({1}, {2}) or 3 ({1}, {2}, {3})"
                                   .format(len(ddims), self.ncolname, self.plevname, self.timename))
            t = v if isinstance(v, torch.Tensor) else torch.as_tensor(np.asarray(v))
            if self.timename not in ddims:                  # 2-D input: add a length-1 time axis (:332-335)
                t = t.unsqueeze(-1)
                ddims = ddims + (self.timename,)
            if self.plevname not in ddims:
                raise RuntimeError("Input data {} does not contain dim {}".format(var, self.plevname))
            perm = [ddims.index(n) for n in self.data_dims]                                   # :343-353
            vals[var] = t.permute(*perm)

        ua = self._in["ua"]
        # ---- coordinates (tem_diagnostics.py:360-367) ----
        if self._kind == "raw":
            if self._raw_plev is None:
                raise RuntimeError("raw array inputs need plev= (pressure levels in hPa)")
            plev = np.asarray(self._raw_plev, dtype=np.float64)
            nt = vals["ua"].shape[2]
            time = np.asarray(self._raw_time) if self._raw_time is not None else np.arange(nt)
        else:
            plev = np.asarray(containers.coord_of(ua, self.plevname), dtype=np.float64)
            time = containers.coord_of(ua, self.timename) if self.timename in ua.dims else np.zeros(1)
        self.NCOL, self.NLEV, self.NT = (int(s) for s in vals["ua"].shape)
        if plev.shape[0] != self.NLEV:
            raise RuntimeError("plev has {} entries but the data have {} levels".format(plev.shape[0], self.NLEV))
        for var in [v for v in vals if v != "ua"]:
            if tuple(vals[var].shape) != tuple(vals["ua"].shape):
                raise RuntimeError("Input data {} has shape {}, expected {}".format(
                    var, tuple(vals[var].shape), tuple(vals["ua"].shape)))

        # ---- pressure direction: model top first (tem_diagnostics.py:369-382) ----
        if plev[0] > plev[-1]:
            vals = {k: torch.flip(v, dims=(1,)) for k, v in vals.items()}
            plev = plev[::-1].copy()
        self.plev = plev
        self.time = time
        self.p = self.plev * 100                                                              # :385
        self._p_np = np.asarray(self.p, dtype=np.float64)

        # ---- zonal-mean latitudes (tem_diagnostics.py:387-398) ----
        tol = 1e-6
        assert (180 / self.zm_dlat).is_integer(), "180 must be divisible by dlat_out"
        self._lat_zm = np.arange(-90, 90 + self.zm_dlat, self.zm_dlat)
        if self._lat_zm[-1] > 90 + tol:
            self._lat_zm = self._lat_zm[:-1]
        if not self.zm_pole_points:
            self._lat_zm = (self._lat_zm[1:] + self._lat_zm[:-1]) / 2
        self.ZM_N = len(self._lat_zm)
        self._f_zm = 2 * Om * np.sin(self._lat_zm * np.pi / 180)                              # :401
        self._coslat_zm = np.cos(self._lat_zm * np.pi / 180)                                  # :402
        self.lat, self.coslat = self._lat_zm, self._coslat_zm
        self.f = self._f_zm[:, np.newaxis, np.newaxis]

        # ---- device residency: contiguous [ncol][plev][time], one dtype ----
        dts = {v.dtype for v in vals.values()}
        self._in_dtype = {k: v.dtype for k, v in vals.items()}
        work = torch.float32 if dts == {torch.float32} else torch.float64
        dev = torch.device("cuda", int(self._device) if not isinstance(self._device, torch.device)
                           else (self._device.index or 0))
        self._torch_out = isinstance(self._in["ua"], torch.Tensor) or (
            self._kind != "raw" and isinstance(self._in["ua"].values, torch.Tensor))
        self._dev_fields = [vals[k].to(device=dev, dtype=work).contiguous() for k in ("ua", "va", "ta", "wap")]
        self.ua, self.va, self.ta, self.wap = self._dev_fields
        self._work_dtype = work
        self._dev_q = [vals["q{}".format(i)].to(device=dev, dtype=work).contiguous() for i in range(self.ntrac)]
        self._tracer_names = [getattr(x, "name", None) for x in self.q]

    # ------------------------------------------------------------------------------------------
    def _np_dtype(self, var):
        import torch
        if isinstance(var, tuple):           # numpy promotion of a product of two inputs
            return np.result_type(*[self._np_dtype(v) for v in var]).type
        return {torch.float32: np.float32, torch.float64: np.float64}.get(self._in_dtype[var], np.float64)

    def _wrap(self, t, name, src_var, native=False, force64=False):
        """Label a device result; cast like the reference's astype."""
        import torch
        dt = np.float64 if (force64 or name in _F64_ALWAYS) else self._np_dtype(src_var)
        dt = np.float64 if dt == np.float64 else np.float32
        tdt = torch.float64 if dt == np.float64 else torch.float32
        t = t.to(tdt)
        vals = t if self._torch_out else t.cpu().numpy()
        if self._kind == "raw":
            return vals
        first = self.ncolname if native else "lat"
        dims = (first, self.plevname, self.timename)
        coords = {self.plevname: self.plev, self.timename: self.time}
        if not native:
            coords["lat"] = self._lat_zm
        return containers.make_like(self._kind, vals, dims, coords, name)

    def _zonal(self, name, src_var):
        return self._wrap(self._zon[_lib.ZONAL_NAMES.index(name)], name, src_var)

    def _result(self, name, src_var):
        return self._wrap(self._res[_lib.RESULT_NAMES.index(name)], name, src_var)

    def _native(self, name, src_var):
        if self._eddy is None:                        # lazily materialised [ncol][plev][time] fields
            self._eddy = self.ZM._plan.tem_eddy(*self._dev_fields)
        return self._wrap(self._eddy[name], name, src_var, native=True)

    def iter_native(self, names=_lib.EDDY_NAMES, chunk_cols=65536):
        """Stream the native-grid attributes (``up vp thetap wapp upvp upwapp vptp``,
        tem_diagnostics.py:420-433) in blocks of columns instead of materialising ``[ncol][plev][time]``
        arrays whole: yields ``(col0, col1, {name: ndarray[col1 - col0, plev, time]})`` with the dtype the
        corresponding property would have.  Device memory: one block of seven arrays."""
        chunk = max(16, (int(chunk_cols) // 16) * 16)
        src = {"up": "ua", "vp": "va", "thetap": "ta", "wapp": "wap", "upvp": "ua", "upwapp": "ua", "vptp": "va"}
        for c0 in range(0, self.NCOL, chunk):
            c1 = min(self.NCOL, c0 + chunk)
            blk = self.ZM._plan.tem_eddy_rows(*self._dev_fields, c0, c1 - c0, names=tuple(names))
            out = {}
            for n, v in blk.items():
                a = v.cpu().numpy()
                dt = np.float64 if n in _F64_ALWAYS else self._np_dtype(src[n])
                out[n] = a.astype(np.float64 if dt == np.float64 else np.float32, copy=False)
            yield c0, c1, out

    # ---- getters (tem_diagnostics.py:412-487) ----
    ub = property(lambda s: s._zonal("ub", "ua"))
    vb = property(lambda s: s._zonal("vb", "va"))
    thetab = property(lambda s: s._zonal("thetab", "ta"))
    wapb = property(lambda s: s._zonal("wapb", "wap"))
    up = property(lambda s: s._native("up", "ua"))
    vp = property(lambda s: s._native("vp", "va"))
    thetap = property(lambda s: s._native("thetap", "ta"))
    wapp = property(lambda s: s._native("wapp", "wap"))
    upvp = property(lambda s: s._native("upvp", "ua"))
    upwapp = property(lambda s: s._native("upwapp", "ua"))
    vptp = property(lambda s: s._native("vptp", "va"))
    upvpb = property(lambda s: s._zonal("upvpb", "ua"))
    upwappb = property(lambda s: s._zonal("upwappb", "ua"))
    vptpb = property(lambda s: s._zonal("vptpb", "va"))
    dub_dp = property(lambda s: s._zonal("dub_dp", "ua"))
    dthetab_dp = property(lambda s: s._zonal("dthetab_dp", "ta"))
    ubcoslat = property(lambda s: s._zonal("ubcoslat", "ua"))
    dubcoslat_dlat = property(lambda s: s._zonal("dubcoslat_dlat", "ua"))
    psicoslat = property(lambda s: s._zonal("psicoslat", "ta"))
    dpsicoslat_dlat = property(lambda s: s._zonal("dpsicoslat_dlat", "ta"))
    int_vbdp = property(lambda s: s._zonal("int_vbdp", "va"))
    psi = property(lambda s: s._zonal("psi", "ta"))
    dpsi_dp = property(lambda s: s._zonal("dpsi_dp", "ta"))
    # ---- tracer getters: lists, one entry per tracer (tem_diagnostics.py:458-475) ----
    def _tzonal(self, name, src, force64=False):
        k = _lib.TRACER_ZONAL_NAMES.index(name)
        return [self._wrap(self._tzon[i][k], name, src(i), force64=force64) for i in range(self.ntrac)]

    def _tnative(self, name, src):
        out = []
        for i in range(self.ntrac):
            if self._teddy[i] is None:
                plan = self.ZM._plan
                if self._last_tracer != i:      # the plan holds the coefficients of one tracer at a time
                    plan.tracer_run(self._dev_q[i], self._dev_fields[1], self._dev_fields[3])
                    self._last_tracer = i
                self._teddy[i] = plan.tracer_eddy(self._dev_q[i], self._dev_fields[1], self._dev_fields[3])
            out.append(self._wrap(self._teddy[i][name], name, src(i), native=True))
        return out

    qb = property(lambda s: s._tzonal("qb", lambda i: "q%d" % i))
    qpvpb = property(lambda s: s._tzonal("qpvpb", lambda i: ("q%d" % i, "va")))
    qpwappb = property(lambda s: s._tzonal("qpwappb", lambda i: ("q%d" % i, "wap")))
    dqb_dp = property(lambda s: s._tzonal("dqb_dp", lambda i: "q%d" % i))
    qbcoslat = property(lambda s: s._tzonal("qbcoslat", lambda i: "q%d" % i, force64=True))
    dqbcoslat_dlat = property(lambda s: s._tzonal("dqbcoslat_dlat", lambda i: "q%d" % i, force64=True))
    qp = property(lambda s: s._tnative("qp", lambda i: "q%d" % i))
    qpvp = property(lambda s: s._tnative("qpvp", lambda i: ("q%d" % i, "va")))
    qpwapp = property(lambda s: s._tnative("qpwapp", lambda i: ("q%d" % i, "wap")))

    @property
    def theta(self):
        """theta = T (p0/p)^k (tem_diagnostics.py:498); recovered as thetap + its native zonal mean
        would cost a sweep, so it is formed from the same per-level scale the engine fuses."""
        import torch
        if self._theta is None:
            from .constants import k
            scale = torch.as_tensor((float(self.p0) / self._p_np) ** k, device=self._dev_fields[2].device)
            self._theta = self._dev_fields[2].to(torch.float64) * scale[None, :, None]
        return self._wrap(self._theta, "THETA", "ta", native=True, force64=True)

    @property
    def out_file(self):
        if self._out_file is None:
            warnings.warn("'out_file' is not set until to_netcdf() is called")
        return self._out_file

    @property
    def q_out_file(self):
        if len(self._q_out_file) == 0:
            warnings.warn("'q_out_file' is emtpy; no tracers currently present")
        if self._q_out_file.count(None) == self.ntrac:
            warnings.warn("'q_out_file' is not set until q_to_netcdf() is called")
        return self._q_out_file

    # ---- the ten diagnostics (tem_diagnostics.py:615-797), each cast to its input's dtype ----
    def vtem(self): return self._result("vtem", "va")                  # noqa: E704
    def omegatem(self): return self._result("omegatem", "wap")         # noqa: E704
    def wtem(self): return self._result("wtem", "wap")                 # noqa: E704
    def psitem(self): return self._result("psitem", "va")              # noqa: E704
    def epfy(self): return self._result("epfy", "ua")                  # noqa: E704
    def epfz(self): return self._result("epfz", "ua")                  # noqa: E704
    def epdiv(self): return self._result("epdiv", "ua")                # noqa: E704
    def utendepfd(self): return self._result("utendepfd", "ua")        # noqa: E704
    def utendvtem(self): return self._result("utendvtem", "ua")        # noqa: E704
    def utendwtem(self): return self._result("utendwtem", "ua")        # noqa: E704

    # ---- tracer TEM (Abalos+ 2017), tem_diagnostics.py:801-991 ----
    def _tracer_result(self, name, qi):
        if qi is None and self.ntrac == 1:
            qi = 0
        elif qi is None and self.ntrac > 1:                                   # :815-816 ...
            raise RuntimeError("qi must be passed to {}() when len(q) > 1!".format(name))
        if self.ntrac == 0:
            raise RuntimeError("no tracers present (argument `q` not passed at object construction)")
        return self._wrap(self._tres[qi][_lib.TRACER_RESULT_NAMES.index(name)], name, "q%d" % qi)

    def etfy(self, qi=None): return self._tracer_result("etfy", qi)               # noqa: E704
    def etfz(self, qi=None): return self._tracer_result("etfz", qi)               # noqa: E704
    def etdiv(self, qi=None): return self._tracer_result("etdiv", qi)             # noqa: E704
    def qtendetfd(self, qi=None): return self._tracer_result("qtendetfd", qi)     # noqa: E704
    def qtendvtem(self, qi=None): return self._tracer_result("qtendvtem", qi)     # noqa: E704
    def qtendwtem(self, qi=None): return self._tracer_result("qtendwtem", qi)     # noqa: E704

    def results(self):
        return {n: getattr(self, n)() for n in _lib.RESULT_NAMES}

    # ---- I/O (tem_diagnostics.py:995-1041): needs xarray + a NetCDF back end ----
    def to_netcdf(self, loc=os.getcwd(), prefix=None, include_attrs=False):
        prefix = "{}_".format(prefix) if prefix is not None else ""
        filename = "{}TEM_{}_{}_L{}.nc".format(prefix, self.ZM.grid_name, self.ZM.grid_out_name, self.L)
        self._out_file = "{}/{}".format(loc, filename)
        names = {}
        if include_attrs:   # (sic) key 'wawpp' as in tem_diagnostics.py:1011
            names = {"ub": "ub", "up": "up", "vb": "vb", "vp": "vp", "thetab": "thetab", "thetap": "thetap",
                     "wapb": "wapb", "wawpp": "wapp", "upvp": "upvp", "upvpb": "upvpb", "upwapp": "upwapp",
                     "upwappb": "upwappb", "vptp": "vptp", "vptpb": "vptpb", "dub_dp": "dub_dp",
                     "dthetab_dp": "dthetab_dp", "ubcoslat": "ubcoslat", "dubcoslat_dlat": "dubcoslat_dlat",
                     "psi": "psi", "psicoslat": "psicoslat", "dpsicoslat_dlat": "dpsicoslat_dlat",
                     "dpsi_dp": "dpsi_dp", "int_vbdp": "int_vbdp"}
        items = dict({k: getattr(self, v) for k, v in names.items()},
                     **{n: getattr(self, n)() for n in _lib.RESULT_NAMES})
        self._write_nc(self._out_file, items)
        return self._out_file

    def _write_nc(self, path, items):
        """xarray when importable (like the reference); otherwise NetCDF-3 through scipy (ncio.py)."""
        try:
            import xarray as xr
        except ImportError:
            xr = None
        if xr is not None:   # pragma: no cover - xarray is absent from this image
            xr.Dataset({k: self._as_xr(v) for k, v in items.items()}).to_netcdf(path)
            return
        from . import ncio
        variables = {}
        for k, v in items.items():
            arr = ncio._np(v)
            native = arr.shape[0] == self.NCOL and arr.shape[0] != len(self.lat)
            dims = v.dims if containers.is_labeled(v) else (("ncol" if native else "lat"), self.plevname, self.timename)
            variables[k] = (tuple(dims), arr, getattr(v, "attrs", None))
        coords = {"lat": np.asarray(self.lat, dtype=np.float64), self.plevname: np.asarray(self.plev, dtype=np.float64)}
        t = np.asarray(self.time)
        if t.dtype.kind in "fiu":
            coords[self.timename] = t.astype(np.float64)
        ncio.write_dataset(path, variables, coords)

    def _as_xr(self, x):
        import xarray as xr
        if isinstance(x, xr.DataArray):
            return x
        v = x.values if containers.is_labeled(x) else x
        if hasattr(v, "detach"):
            v = v.detach().cpu().numpy()
        dims = x.dims if containers.is_labeled(x) else ("lat", self.plevname, self.timename)
        return xr.DataArray(v, dims=dims)

    def q_to_netcdf(self, loc=os.getcwd(), qi=None, prefix=None, include_attrs=False):
        """tem_diagnostics.py:1045-1103 (file naming and variable set; needs xarray + NetCDF)."""
        assert self.ntrac > 0, "No tracers to output (argument `q` not passed at object construction)"
        prefix = "{}_".format(prefix) if prefix is not None else ""
        names = [n if n is not None else "q{}".format(i) for i, n in enumerate(self._tracer_names)]
        idx = range(self.ntrac) if qi is None else [qi]
        for i in idx:
            items = {n: getattr(self, n)(i) for n in _lib.TRACER_RESULT_NAMES}
            if include_attrs:   # (sic) key 'dqp_dp' as in tem_diagnostics.py:1081
                items = dict({"qpvp": self.qpvp[i], "qpwapp": self.qpwapp[i], "qpvpb": self.qpvpb[i],
                              "qpwappb": self.qpwappb[i], "dqp_dp": self.dqb_dp[i], "qbcoslat": self.qbcoslat[i],
                              "dqbcoslat_dlat": self.dqbcoslat_dlat[i]}, **items)
            self._q_out_file[i] = "{}/{}TEM_{}_{}_L{}_TRACER-{}.nc".format(
                loc, prefix, self.ZM.grid_name, self.ZM.grid_out_name, self.L, names[i])
            self._write_nc(self._q_out_file[i], items)
        return self._q_out_file
