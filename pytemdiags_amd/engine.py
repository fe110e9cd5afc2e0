"""Thin object wrapper over the C ABI (include/temx.h).  PyTorch-ROCm tensors are only the
device container: every number is produced by the HIP kernels in csrc/.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib
from ._lib import check

_DT = {torch.float64: _lib.F64, torch.float32: _lib.F32}


def _ptr(t):
    return C.c_void_p(t.data_ptr())


def _dbl(a):
    a = np.ascontiguousarray(a, dtype=np.float64)
    return a, a.ctypes.data_as(C.POINTER(C.c_double))


class Plan:
    """One plan per (device, native grid, output latitudes, L).  See include/temx.h."""

    def __init__(self, lat_deg, lat_out_deg, L, device=0, defer_finalize=False, symmetry=True, classes=True, qr=True,
                 form=None, fp32_fields=False):
        self._h = C.c_void_p()
        self.lib = _lib.load()
        if isinstance(device, torch.device):
            device = device.index or 0
        self.device_index = int(device)
        self.device = torch.device("cuda", self.device_index)
        lat, plat = _dbl(lat_deg)
        lat_out, plat_out = _dbl(lat_out_deg)
        self.N, self.M, self.L, self.K = int(lat.size), int(lat_out.size), int(L), int(L) + 1
        self.lat_out = lat_out
        self.nlev = self.nt = self.D = None
        check(self.lib.temx_plan_create(C.byref(self._h), self.device_index, self.N, self.L, self.M,
                                        plat, plat_out, (_lib.DEFER_FINALIZE if defer_finalize else 0)
                                        | (0 if symmetry else _lib.NO_SYMMETRY)
                                        | (0 if classes else _lib.NO_CLASSES)
                                        | (0 if qr else _lib.NO_QR)
                                        | (_lib.LAT_TOL_F32 if fp32_fields else 0)))
        if form is not None:
            self.configure(form=form)

    @property
    def paired(self):
        """True when the sweeps exploit the latitude structure of the grid (mirror-paired or
        latitude-class sweeps); see :attr:`sweep_mode`."""
        return bool(self.lib.temx_plan_is_paired(self._h))

    @property
    def sweep_mode(self):
        """0 generic sweeps, 1 mirror-paired sweeps, 2 latitude-class sweeps."""
        return int(self.lib.temx_plan_sweep_mode(self._h))

    @property
    def one_pass(self):
        """True when (after ``set_tem``) the class path reads the fields once (see include/temx.h)."""
        return bool(self.lib.temx_plan_one_pass(self._h))

    @property
    def single_sweep(self):
        """True when ``tem_run`` takes the single-sweep form: no class-sum stream (see include/temx.h)."""
        return bool(self.lib.temx_plan_single_sweep(self._h))

    @property
    def tracer_one_pass(self):
        """True when tracer runs should take the one-pass stages (``one_pass`` and the option / TEMX_TRACER_ONE_PASS=1;
        the two-pass tracer stages measured faster, see include/temx.h)."""
        return self.one_pass and self.option(_lib.OPT_TRACER_ONE_PASS) == 1

    def configure(self, form=None, os_map=None, op_map=None, os_subsample=None, tracer_one_pass=None,
                  single_sweep_min_groups=None, os_contract=None):
        """Path selection (temx_plan_configure); ``set_tem`` must follow.  ``form``: a key of ``_lib.FORMS``
        ("auto", "two-pass", "class-sums", "single-sweep", "no-single-sweep"); ``os_map`` / ``op_map``:
        "row" or "tile" (lane map of the loads of the single sweep / of sweep 1 of the class-sum form)."""
        def put(opt, val):
            check(self.lib.temx_plan_configure(self._h, opt, int(val)))
        if form is not None:
            put(_lib.OPT_FORM, _lib.FORMS[form] if isinstance(form, str) else form)
        for opt, val in ((_lib.OPT_OS_MAP, os_map), (_lib.OPT_OP_MAP, op_map)):
            if val is not None:
                put(opt, {"row": 0, "tile": 1}[val] if isinstance(val, str) else val)
        if os_subsample is not None:
            put(_lib.OPT_OS_SUBSAMPLE, os_subsample)
        if tracer_one_pass is not None:
            put(_lib.OPT_TRACER_ONE_PASS, 1 if tracer_one_pass else 0)
        if single_sweep_min_groups is not None:
            put(_lib.OPT_SINGLE_SWEEP_MIN_GROUPS, single_sweep_min_groups)
        if os_contract is not None:     # "mfma" (default) or "lds" (the round-3 form, A/B)
            put(_lib.OPT_OS_CONTRACT, {"mfma": 0, "lds": 1}[os_contract] if isinstance(os_contract, str) else os_contract)
        self.nlev = self.nt = self.D = None

    def option(self, opt):
        return int(self.lib.temx_plan_option(self._h, int(opt)))

    # ---- lifetime ----
    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self.lib.temx_plan_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _field(self, A, D=None):
        if not isinstance(A, torch.Tensor):
            raise TypeError("device tensors expected")
        if A.device != self.device:
            raise ValueError("tensor on %s, plan on %s" % (A.device, self.device))
        if A.dtype not in _DT:
            raise TypeError("dtype must be float64 or float32, got %s" % A.dtype)
        if A.shape[0] != self.N:
            raise ValueError("leading dimension must be ncol = %d" % self.N)
        A = A.contiguous()
        if D is not None and A.numel() != self.N * D:
            raise ValueError("field has %d elements, expected %d x %d" % (A.numel(), self.N, D))
        return A

    # ---- plan pieces ----
    def finalize(self, G=None):
        if G is None:
            check(self.lib.temx_plan_finalize(self._h, None))
        else:
            g, pg = _dbl(G)
            assert g.shape == (self.K, self.K)
            check(self.lib.temx_plan_finalize(self._h, pg))

    def refine(self, G2=None):
        """Second pass of the re-orthogonalisation (temx_plan_refine): ``G2`` the all-reduced
        ``matrix(MAT_GRAM2)`` of an ncol-sharded job, or None for this plan's own rows."""
        if G2 is None:
            check(self.lib.temx_plan_refine(self._h, None))
        else:
            g, pg = _dbl(G2)
            assert g.shape == (self.K, self.K)
            check(self.lib.temx_plan_refine(self._h, pg))

    def set_weights(self, weights):
        w, pw = _dbl(weights)
        assert w.size == self.N
        check(self.lib.temx_plan_set_weights(self._h, pw))

    def matrix(self, which):
        shape = {_lib.MAT_Y0: (self.N, self.K), _lib.MAT_Y0P: (self.M, self.K),
                 _lib.MAT_GRAM: (self.K, self.K), _lib.MAT_GINV: (self.K, self.K),
                 _lib.MAT_GRAM2: (self.K, self.K),
                 _lib.MAT_GX: (self.K, self.KX), _lib.MAT_GSUB: (self.KR, self.KR),
                 _lib.MAT_Y0INV: (self.K, self.N)}[which]
        out = torch.empty(shape, dtype=torch.float64, device=self.device)
        check(self.lib.temx_get_matrix(self._h, which, _ptr(out), self._stream()))
        return out

    # ---- operator API ----
    def project(self, A):
        A = self._field(A)
        D = A.numel() // self.N
        B = torch.empty((self.K, D), dtype=torch.float64, device=self.device)
        check(self.lib.temx_project(self._h, _ptr(A), _DT[A.dtype], D, _ptr(B), self._stream()))
        return B

    def zonal_mean(self, A, native=False):
        A = self._field(A)
        D = A.numel() // self.N
        out = torch.empty((self.N if native else self.M,) + tuple(A.shape[1:]), dtype=torch.float64,
                          device=self.device)
        check(self.lib.temx_zonal_mean(self._h, _ptr(A), _DT[A.dtype], D, _ptr(out), 1 if native else 0,
                                       self._stream()))
        return out

    def zonal_mean_from_sums(self, B, trailing_shape, native=False):
        D = int(B.shape[1])
        out = torch.empty((self.N if native else self.M,) + tuple(trailing_shape), dtype=torch.float64,
                          device=self.device)
        check(self.lib.temx_zonal_mean_from_sums(self._h, _ptr(B.contiguous()), D, _ptr(out),
                                                 1 if native else 0, self._stream()))
        return out

    # ---- TEM pipeline ----
    def set_tem(self, nlev, nt, p_pa, p0=101325.0):
        p, pp = _dbl(p_pa)
        assert p.size == nlev
        check(self.lib.temx_plan_set_tem(self._h, int(nlev), int(nt), pp, float(p0)))
        self.nlev, self.nt, self.D = int(nlev), int(nt), int(nlev) * int(nt)
        self.tem_args = (int(nlev), int(nt), p.copy(), float(p0))

    def _four(self, ua, va, ta, wap):
        if self.D is None:
            raise _lib.TemxError(-5, "temx_plan_set_tem has not been called")
        fs = [self._field(x, self.D) for x in (ua, va, ta, wap)]
        if len({f.dtype for f in fs}) != 1:
            raise TypeError("ua, va, ta, wap must share one dtype")
        return fs, _DT[fs[0].dtype]

    def _alloc_results(self, want_zonal):
        res = torch.empty((len(_lib.RESULT_NAMES), self.M, self.nlev, self.nt), dtype=torch.float64,
                          device=self.device)
        zon = None
        if want_zonal:
            zon = torch.empty((len(_lib.ZONAL_NAMES), self.M, self.nlev, self.nt), dtype=torch.float64,
                              device=self.device)
        return res, zon

    def tem_run(self, ua, va, ta, wap, want_zonal=False, out=None):
        (u, v, t, w), dt = self._four(ua, va, ta, wap)
        res, zon = out if out is not None else self._alloc_results(want_zonal)
        check(self.lib.temx_tem_run(self._h, _ptr(u), _ptr(v), _ptr(t), _ptr(w), dt, _ptr(res),
                                    _ptr(zon) if zon is not None else None, self._stream()))
        return res, zon

    # ---- the single sweep in three steps (ncol-sharded jobs exchange between them; include/temx.h) ----
    @property
    def KX(self):
        return 2 * self.L + 1

    @property
    def KR(self):
        return min(16, self.K)

    @property
    def os_rows(self):
        """Rows of the projections the single sweep hands over: four fields to degree 2L, three products to degree L."""
        return 4 * self.KX + 3 * self.K

    def set_os_matrices(self, Gx, Gsub):
        gx, pgx = _dbl(Gx)
        gs, pgs = _dbl(Gsub)
        assert gx.shape == (self.K, self.KX) and gs.shape == (self.KR, self.KR)
        check(self.lib.temx_plan_set_os_matrices(self._h, pgx, pgs))

    def _sliced(self, rows, nslices):
        ntmax = -(-self.nt // nslices)
        shape = (rows, self.nlev, self.nt) if nslices == 1 else (nslices, rows * self.nlev * ntmax)
        return torch.zeros(shape, dtype=torch.float64, device=self.device)

    def tem_os_prepass(self, ua, va, ta, wap, out=None):
        (u, v, t, w), dt = self._four(ua, va, ta, wap)
        As = out if out is not None else torch.empty((4, self.KR, self.D), dtype=torch.float64, device=self.device)
        check(self.lib.temx_tem_os_prepass(self._h, _ptr(u), _ptr(v), _ptr(t), _ptr(w), dt, _ptr(As), self._stream()))
        return As

    def tem_os_sweep(self, ua, va, ta, wap, As, nslices=1, out=None):
        """-> the projections, whole ([rows][nlev][nt]) or as ``nslices`` time slices ([nslices][chunk]: the input of
        ``reduce_scatter_tensor``; slice w holds rows of nlev x ntw(w) columns packed at its start)."""
        (u, v, t, w), dt = self._four(ua, va, ta, wap)
        proj = out if out is not None else self._sliced(self.os_rows, nslices)
        check(self.lib.temx_tem_os_sweep(self._h, _ptr(u), _ptr(v), _ptr(t), _ptr(w), dt, _ptr(As.contiguous()),
                                         int(nslices), _ptr(proj), self._stream()))
        return proj

    def tem_os_tail(self, proj_slice, t0=0, nts=None, want_zonal=False, out=None):
        nts = self.nt if nts is None else int(nts)
        res = out if out is not None else torch.empty((len(_lib.RESULT_NAMES), self.M, self.nlev, nts),
                                                      dtype=torch.float64, device=self.device)
        zon = None
        if want_zonal:
            zon = torch.empty((len(_lib.ZONAL_NAMES), self.M, self.nlev, nts), dtype=torch.float64, device=self.device)
        check(self.lib.temx_tem_os_tail(self._h, _ptr(proj_slice), int(t0), nts, _ptr(res),
                                        _ptr(zon) if zon is not None else None, self._stream()))
        return res, zon

    # tracers in the single-sweep form: one or two per sweep (two share one read of v and omega)
    def _tracers(self, qs, va, wap):
        if isinstance(qs, torch.Tensor):
            qs = [qs]
        if len(qs) not in (1, 2):
            raise ValueError("one or two tracers per sweep")
        fs, dt = self._three(qs[0], va, wap)
        qq = [fs[0]] + [self._field(x, self.D) for x in qs[1:]]
        if len({x.dtype for x in qq}) != 1:
            raise TypeError("the tracers must share the dtype of va, wap")
        ptrs = (C.c_void_p * len(qq))(*[x.data_ptr() for x in qq])
        return qq, ptrs, fs[1], fs[2], dt

    def tracers_os_prepass(self, qs, va, wap, out=None):
        qq, ptrs, v, w, dt = self._tracers(qs, va, wap)
        Asq = out if out is not None else torch.empty((len(qq), self.KR, self.D), dtype=torch.float64, device=self.device)
        check(self.lib.temx_tracers_os_prepass(self._h, len(qq), ptrs, _ptr(v), _ptr(w), dt, _ptr(Asq), self._stream()))
        return Asq

    def tracers_os_sweep(self, qs, va, wap, Asq, nslices=1, out=None):
        qq, ptrs, v, w, dt = self._tracers(qs, va, wap)
        projq = out if out is not None else self._sliced(len(qq) * (self.KX + 2 * self.K), nslices)
        check(self.lib.temx_tracers_os_sweep(self._h, len(qq), ptrs, _ptr(v), _ptr(w), dt, _ptr(Asq.contiguous()), int(nslices),
                                             _ptr(projq), self._stream()))
        return projq

    def _tracer_outs(self, nq, nts, want_zonal):
        tres = [torch.empty((len(_lib.TRACER_RESULT_NAMES), self.M, self.nlev, nts), dtype=torch.float64, device=self.device)
                for _ in range(nq)]
        tzon = [torch.empty((len(_lib.TRACER_ZONAL_NAMES), self.M, self.nlev, nts), dtype=torch.float64, device=self.device)
                for _ in range(nq)] if want_zonal else None
        pr = (C.c_void_p * nq)(*[t.data_ptr() for t in tres])
        pz = (C.c_void_p * nq)(*[t.data_ptr() for t in tzon]) if want_zonal else None
        return tres, tzon, pr, pz

    def tracers_os_tail(self, nq, projq_slice, nts, want_zonal=False):
        """-> [(tres, tzon)] per tracer, for the snapshots of the latest ``tem_os_tail``."""
        tres, tzon, pr, pz = self._tracer_outs(nq, nts, want_zonal)
        check(self.lib.temx_tracers_os_tail(self._h, int(nq), _ptr(projq_slice), pr, pz, self._stream()))
        return [(tres[i], tzon[i] if tzon else None) for i in range(nq)]

    def tracers_run(self, qs, va, wap, want_zonal=False):
        """All tracers of a TEM run (temx_tracers_run): in pairs on the single-sweep path.  -> [(tres, tzon)]."""
        qs = list(qs)
        fs, dt = self._three(qs[0], va, wap)
        qq = [fs[0]] + [self._field(x, self.D) for x in qs[1:]]
        if len({x.dtype for x in qq}) != 1:
            raise TypeError("the tracers must share the dtype of va, wap")
        ptrs = (C.c_void_p * len(qq))(*[x.data_ptr() for x in qq])
        tres, tzon, pr, pz = self._tracer_outs(len(qq), self.nt, want_zonal)
        check(self.lib.temx_tracers_run(self._h, len(qq), ptrs, _ptr(fs[1]), _ptr(fs[2]), dt, pr, pz, self._stream()))
        return [(tres[i], tzon[i] if tzon else None) for i in range(len(qq))]

    def time_slices(self, B, nslices):
        """Rows of [nlev][nt] columns -> the reduce-scatter layout [nslices][chunk] (temx_time_slices)."""
        B = B.contiguous()
        rows = B.numel() // self.D
        out = self._sliced(rows, nslices)
        check(self.lib.temx_time_slices(self._h, _ptr(B), rows, int(nslices), _ptr(out), self._stream()))
        return out

    def tem_tail_from_sums(self, B4s, B3s, t0, nts, want_zonal=False):
        res = torch.empty((len(_lib.RESULT_NAMES), self.M, self.nlev, nts), dtype=torch.float64, device=self.device)
        zon = None
        if want_zonal:
            zon = torch.empty((len(_lib.ZONAL_NAMES), self.M, self.nlev, nts), dtype=torch.float64, device=self.device)
        check(self.lib.temx_tem_tail_from_sums(self._h, _ptr(B4s), _ptr(B3s), int(t0), int(nts), _ptr(res),
                                               _ptr(zon) if zon is not None else None, self._stream()))
        return res, zon

    def tem_stage1(self, ua, va, ta, wap):
        (u, v, t, w), dt = self._four(ua, va, ta, wap)
        B4 = torch.empty((4, self.K, self.D), dtype=torch.float64, device=self.device)
        check(self.lib.temx_tem_stage1(self._h, _ptr(u), _ptr(v), _ptr(t), _ptr(w), dt, _ptr(B4),
                                       self._stream()))
        return B4

    def tem_stage2(self, ua, va, ta, wap, B4):
        (u, v, t, w), dt = self._four(ua, va, ta, wap)
        B3 = torch.empty((3, self.K, self.D), dtype=torch.float64, device=self.device)
        check(self.lib.temx_tem_stage2(self._h, _ptr(u), _ptr(v), _ptr(t), _ptr(w), dt,
                                       _ptr(B4.contiguous()), _ptr(B3), self._stream()))
        return B3

    def tem_stage2_from_sums(self, B4):
        """Stage 2 of the one-pass class path (``one_pass``): B3 from the class sums the latest
        ``tem_stage1`` on this plan stored -- describes the fields that call was given, reads none."""
        if self.D is None:
            raise _lib.TemxError(-5, "temx_plan_set_tem has not been called")
        B3 = torch.empty((3, self.K, self.D), dtype=torch.float64, device=self.device)
        check(self.lib.temx_tem_stage2_from_sums(self._h, _ptr(B4.contiguous()), _ptr(B3), self._stream()))
        return B3

    def tem_stage3(self, B3, want_zonal=False):
        res, zon = self._alloc_results(want_zonal)
        check(self.lib.temx_tem_stage3(self._h, _ptr(B3.contiguous()), _ptr(res),
                                       _ptr(zon) if zon is not None else None, self._stream()))
        return res, zon

    def tem_eddy(self, ua, va, ta, wap, names=_lib.EDDY_NAMES):
        (u, v, t, w), dt = self._four(ua, va, ta, wap)
        outs = {}
        ptrs = (C.c_void_p * len(_lib.EDDY_NAMES))()
        for i, n in enumerate(_lib.EDDY_NAMES):
            if n in names:
                outs[n] = torch.empty((self.N, self.nlev, self.nt), dtype=torch.float64, device=self.device)
                ptrs[i] = outs[n].data_ptr()
            else:
                ptrs[i] = None
        check(self.lib.temx_tem_eddy(self._h, _ptr(u), _ptr(v), _ptr(t), _ptr(w), dt, ptrs, self._stream()))
        return outs

    def tem_eddy_rows(self, ua, va, ta, wap, row0, nrows, names=_lib.EDDY_NAMES):
        """The native-grid eddies / products of rows [row0, row0 + nrows) only (row0 % 16 == 0), compact
        ``[nrows][nlev][nt]`` tensors: bounded device memory for streaming large runs."""
        (u, v, t, w), dt = self._four(ua, va, ta, wap)
        outs = {}
        ptrs = (C.c_void_p * len(_lib.EDDY_NAMES))()
        for i, n in enumerate(_lib.EDDY_NAMES):
            if n in names:
                outs[n] = torch.empty((int(nrows), self.nlev, self.nt), dtype=torch.float64, device=self.device)
                ptrs[i] = outs[n].data_ptr()
            else:
                ptrs[i] = None
        check(self.lib.temx_tem_eddy_rows(self._h, _ptr(u), _ptr(v), _ptr(t), _ptr(w), dt, int(row0), int(nrows),
                                          ptrs, self._stream()))
        return outs

    # ---- tracer TEM (one tracer at a time; needs a preceding tem_run on the same fields) ----
    def _three(self, q, va, wap):
        if self.D is None:
            raise _lib.TemxError(-5, "temx_plan_set_tem has not been called")
        fs = [self._field(x, self.D) for x in (q, va, wap)]
        if len({f.dtype for f in fs}) != 1:
            raise TypeError("q, va, wap must share one dtype")
        return fs, _DT[fs[0].dtype]

    def tracer_run(self, q, va, wap, want_zonal=False):
        (qq, v, w), dt = self._three(q, va, wap)
        tres = torch.empty((len(_lib.TRACER_RESULT_NAMES), self.M, self.nlev, self.nt), dtype=torch.float64,
                           device=self.device)
        tzon = None
        if want_zonal:
            tzon = torch.empty((len(_lib.TRACER_ZONAL_NAMES), self.M, self.nlev, self.nt), dtype=torch.float64,
                               device=self.device)
        check(self.lib.temx_tracer_run(self._h, _ptr(qq), _ptr(v), _ptr(w), dt, _ptr(tres),
                                       _ptr(tzon) if tzon is not None else None, self._stream()))
        return tres, tzon

    def _alloc_tracer(self, want_zonal):
        tres = torch.empty((len(_lib.TRACER_RESULT_NAMES), self.M, self.nlev, self.nt), dtype=torch.float64,
                           device=self.device)
        tzon = None
        if want_zonal:
            tzon = torch.empty((len(_lib.TRACER_ZONAL_NAMES), self.M, self.nlev, self.nt), dtype=torch.float64,
                               device=self.device)
        return tres, tzon

    def tem_tracer_run(self, ua, va, ta, wap, q, want_zonal=False):
        """TEM and one tracer in one call; on the one-pass class path the five arrays are swept ONCE
        (temx_tem_tracer_run).  Returns (results, zonal, tracer results, tracer zonal)."""
        (u, v, t, w), dt = self._four(ua, va, ta, wap)
        qq = self._field(q, self.D)
        if qq.dtype != u.dtype:
            raise TypeError("the tracer must have the dtype of the fields")
        res, zon = self._alloc_results(want_zonal)
        tres, tzon = self._alloc_tracer(want_zonal)
        check(self.lib.temx_tem_tracer_run(self._h, _ptr(u), _ptr(v), _ptr(t), _ptr(w), _ptr(qq), dt, _ptr(res),
                                           _ptr(zon) if zon is not None else None, _ptr(tres),
                                           _ptr(tzon) if tzon is not None else None, self._stream()))
        return res, zon, tres, tzon

    def tem_tracer_stage1(self, ua, va, ta, wap, q):
        """Stage 1 of the TEM run and of the tracer in one sweep (one-pass class path): (B4, Bq)."""
        (u, v, t, w), dt = self._four(ua, va, ta, wap)
        qq = self._field(q, self.D)
        if qq.dtype != u.dtype:
            raise TypeError("the tracer must have the dtype of the fields")
        B4 = torch.empty((4, self.K, self.D), dtype=torch.float64, device=self.device)
        Bq = torch.empty((self.K, self.D), dtype=torch.float64, device=self.device)
        check(self.lib.temx_tem_tracer_stage1(self._h, _ptr(u), _ptr(v), _ptr(t), _ptr(w), _ptr(qq), dt, _ptr(B4),
                                              _ptr(Bq), self._stream()))
        return B4, Bq

    def tracer_stage1(self, q):
        qq = self._field(q, self.D)
        Bq = torch.empty((self.K, self.D), dtype=torch.float64, device=self.device)
        check(self.lib.temx_tracer_stage1(self._h, _ptr(qq), _DT[qq.dtype], _ptr(Bq), self._stream()))
        return Bq

    def tracer_stage2(self, q, va, wap, Bq):
        (qq, v, w), dt = self._three(q, va, wap)
        Bq2 = torch.empty((2, self.K, self.D), dtype=torch.float64, device=self.device)
        check(self.lib.temx_tracer_stage2(self._h, _ptr(qq), _ptr(v), _ptr(w), dt, _ptr(Bq.contiguous()),
                                          _ptr(Bq2), self._stream()))
        return Bq2

    def tracer_stage1_sums(self, q, va, wap):
        """One-pass stage 1 (needs ``one_pass`` and a preceding TEM stage 1 / run on the same va, wap)."""
        (qq, v, w), dt = self._three(q, va, wap)
        Bq = torch.empty((self.K, self.D), dtype=torch.float64, device=self.device)
        check(self.lib.temx_tracer_stage1_sums(self._h, _ptr(qq), _ptr(v), _ptr(w), dt, _ptr(Bq), self._stream()))
        return Bq

    def tracer_stage2_from_sums(self, Bq):
        Bq2 = torch.empty((2, self.K, self.D), dtype=torch.float64, device=self.device)
        check(self.lib.temx_tracer_stage2_from_sums(self._h, _ptr(Bq.contiguous()), _ptr(Bq2), self._stream()))
        return Bq2

    def tracer_stage3(self, Bq2, want_zonal=False):
        tres = torch.empty((len(_lib.TRACER_RESULT_NAMES), self.M, self.nlev, self.nt), dtype=torch.float64,
                           device=self.device)
        tzon = None
        if want_zonal:
            tzon = torch.empty((len(_lib.TRACER_ZONAL_NAMES), self.M, self.nlev, self.nt), dtype=torch.float64,
                               device=self.device)
        check(self.lib.temx_tracer_stage3(self._h, _ptr(Bq2.contiguous()), _ptr(tres),
                                          _ptr(tzon) if tzon is not None else None, self._stream()))
        return tres, tzon

    def tracer_eddy(self, q, va, wap):
        (qq, v, w), dt = self._three(q, va, wap)
        outs = {n: torch.empty((self.N, self.nlev, self.nt), dtype=torch.float64, device=self.device)
                for n in _lib.TRACER_EDDY_NAMES}
        ptrs = (C.c_void_p * 3)(*[outs[n].data_ptr() for n in _lib.TRACER_EDDY_NAMES])
        check(self.lib.temx_tracer_eddy(self._h, _ptr(qq), _ptr(v), _ptr(w), dt, ptrs, self._stream()))
        return outs

    def status(self):
        """Synchronise; True when a non-finite value reached the zonal sums (NaN input)."""
        f = C.c_int(0)
        check(self.lib.temx_status(self._h, C.byref(f), self._stream()))
        return bool(f.value)

    # ---- measurement helpers ----
    def kernel_timing(self, enable=True):
        check(self.lib.temx_kernel_timing(self._h, 1 if enable else 0))

    def kernel_timing_read(self, which):
        ms, n = C.c_double(0), C.c_int(0)
        check(self.lib.temx_kernel_timing_read(self._h, which, C.byref(ms), C.byref(n)))
        return ms.value, n.value


def synth_fields(device, lat_deg, lon_deg, plev_hpa, nt, t0=0, dtype=torch.float64, seed=0):
    """SURVEY 8(d) synthetic fields generated in place on the device (bench / tests)."""
    lib = _lib.load()
    dev = torch.device("cuda", int(device))
    lat = torch.as_tensor(np.asarray(lat_deg, dtype=np.float64), device=dev)
    lon = torch.as_tensor(np.asarray(lon_deg, dtype=np.float64), device=dev)
    pl = torch.as_tensor(np.asarray(plev_hpa, dtype=np.float64), device=dev)
    N, nlev = lat.numel(), pl.numel()
    outs = [torch.empty((N, nlev, nt), dtype=dtype, device=dev) for _ in range(4)]
    st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    check(lib.temx_synth_fields(int(device), N, nlev, nt, t0, _ptr(lat), _ptr(lon), _ptr(pl), _DT[dtype],
                                seed, *[_ptr(o) for o in outs], st))
    torch.cuda.current_stream(dev).synchronize()   # lat/lon/pl temporaries stay alive until done
    return outs


def mfma_f64_peak(device=0, iters=20000):
    lib = _lib.load()
    t = C.c_double(0)
    check(lib.temx_mfma_f64_peak(int(device), iters, C.byref(t)))
    return t.value
