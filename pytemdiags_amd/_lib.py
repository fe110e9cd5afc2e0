"""ctypes binding of libtemx.so (C ABI declared in include/temx.h).

The library is the only compute back end: if it is missing or fails to load, importing the
engine raises -- there is no CPU fallback anywhere in this package.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("TEMX_LIB") or os.path.join(_HERE, "libtemx.so")   # TEMX_LIB: A/B builds

ABI_VERSION = 401           # temx_version() of the library these bindings were written for (include/temx.h)
F64, F32 = 0, 1
DEFER_FINALIZE = 1
NO_SYMMETRY = 2
NO_CLASSES = 4
NO_QR = 8
LAT_TOL_F32 = 16
MAT_Y0, MAT_Y0P, MAT_GRAM, MAT_GINV, MAT_Y0INV, MAT_GRAM2, MAT_GX, MAT_GSUB = 0, 1, 2, 3, 4, 5, 6, 7
# temx_plan_configure options and the forms of the latitude-class sweeps (include/temx.h)
OPT_FORM, OPT_OS_MAP, OPT_OP_MAP, OPT_OS_SUBSAMPLE, OPT_TRACER_ONE_PASS, OPT_SINGLE_SWEEP_MIN_GROUPS, OPT_OS_CONTRACT = 1, 2, 3, 4, 5, 6, 7
FORM_AUTO, FORM_TWO_PASS, FORM_CLASS_SUMS, FORM_SINGLE_SWEEP, FORM_NO_SINGLE_SWEEP = -1, 0, 1, 2, 3
FORMS = {"auto": FORM_AUTO, "two-pass": FORM_TWO_PASS, "class-sums": FORM_CLASS_SUMS,
         "single-sweep": FORM_SINGLE_SWEEP, "no-single-sweep": FORM_NO_SINGLE_SWEEP}

RESULT_NAMES = ("vtem", "omegatem", "wtem", "psitem", "epfy", "epfz", "epdiv",
                "utendepfd", "utendvtem", "utendwtem")
ZONAL_NAMES = ("ub", "vb", "thetab", "wapb", "upvpb", "upwappb", "vptpb", "dub_dp", "dthetab_dp",
               "ubcoslat", "dubcoslat_dlat", "psi", "psicoslat", "dpsicoslat_dlat", "dpsi_dp",
               "int_vbdp")
EDDY_NAMES = ("up", "vp", "thetap", "wapp", "upvp", "upwapp", "vptp")
TRACER_RESULT_NAMES = ("etfy", "etfz", "etdiv", "qtendetfd", "qtendvtem", "qtendwtem")
TRACER_ZONAL_NAMES = ("qb", "qpvpb", "qpwappb", "dqb_dp", "qbcoslat", "dqbcoslat_dlat")
TRACER_EDDY_NAMES = ("qp", "qpvp", "qpwapp")

ERRORS = {0: "TEMX_OK", -1: "TEMX_EINVAL", -2: "TEMX_EHIP", -3: "TEMX_ENOMEM", -4: "TEMX_ERANK",
          -5: "TEMX_ESTATE", -6: "TEMX_EUNSUPPORTED", -7: "TEMX_EINTERNAL"}

# every symbol include/temx.h declares: (name, restype, argtypes)
_vp, _i, _i64, _dp, _u64 = C.c_void_p, C.c_int, C.c_int64, C.POINTER(C.c_double), C.c_uint64
SIGNATURES = [
    ("temx_version", _i, []),
    ("temx_last_error", C.c_char_p, []),
    ("temx_device_count", _i, []),
    ("temx_plan_create", _i, [C.POINTER(_vp), _i, _i64, _i, _i, _dp, _dp, _i]),
    ("temx_plan_finalize", _i, [_vp, _dp]),
    ("temx_plan_refine", _i, [_vp, _dp]),
    ("temx_plan_set_weights", _i, [_vp, _dp]),
    ("temx_plan_destroy", None, [_vp]),
    ("temx_plan_is_paired", _i, [_vp]),
    ("temx_plan_sweep_mode", _i, [_vp]),
    ("temx_plan_one_pass", _i, [_vp]),
    ("temx_plan_single_sweep", _i, [_vp]),
    ("temx_get_matrix", _i, [_vp, _i, _vp, _vp]),
    ("temx_plan_configure", _i, [_vp, _i, _i]),
    ("temx_plan_option", _i, [_vp, _i]),
    ("temx_plan_set_os_matrices", _i, [_vp, _dp, _dp]),
    ("temx_project", _i, [_vp, _vp, _i, _i64, _vp, _vp]),
    ("temx_zonal_mean", _i, [_vp, _vp, _i, _i64, _vp, _i, _vp]),
    ("temx_zonal_mean_from_sums", _i, [_vp, _vp, _i64, _vp, _i, _vp]),
    ("temx_plan_set_tem", _i, [_vp, _i, _i64, _dp, C.c_double]),
    ("temx_tem_stage1", _i, [_vp, _vp, _vp, _vp, _vp, _i, _vp, _vp]),
    ("temx_tem_stage2", _i, [_vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp]),
    ("temx_tem_stage2_from_sums", _i, [_vp, _vp, _vp, _vp]),
    ("temx_tem_stage3", _i, [_vp, _vp, _vp, _vp, _vp]),
    ("temx_tem_run", _i, [_vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp]),
    ("temx_tem_os_prepass", _i, [_vp, _vp, _vp, _vp, _vp, _i, _vp, _vp]),
    ("temx_tem_os_sweep", _i, [_vp, _vp, _vp, _vp, _vp, _i, _vp, _i, _vp, _vp]),
    ("temx_tem_os_tail", _i, [_vp, _vp, _i64, _i64, _vp, _vp, _vp]),
    ("temx_tracers_os_prepass", _i, [_vp, _i, C.POINTER(_vp), _vp, _vp, _i, _vp, _vp]),
    ("temx_tracers_os_sweep", _i, [_vp, _i, C.POINTER(_vp), _vp, _vp, _i, _vp, _i, _vp, _vp]),
    ("temx_tracers_os_tail", _i, [_vp, _i, _vp, C.POINTER(_vp), C.POINTER(_vp), _vp]),
    ("temx_tracers_run", _i, [_vp, _i, C.POINTER(_vp), _vp, _vp, _i, C.POINTER(_vp), C.POINTER(_vp), _vp]),
    ("temx_tem_tail_from_sums", _i, [_vp, _vp, _vp, _i64, _i64, _vp, _vp, _vp]),
    ("temx_time_slices", _i, [_vp, _vp, _i64, _i, _vp, _vp]),
    ("temx_tem_eddy", _i, [_vp, _vp, _vp, _vp, _vp, _i, C.POINTER(_vp), _vp]),
    ("temx_tem_eddy_rows", _i, [_vp, _vp, _vp, _vp, _vp, _i, _i64, _i64, C.POINTER(_vp), _vp]),
    ("temx_tracer_stage1", _i, [_vp, _vp, _i, _vp, _vp]),
    ("temx_tracer_stage2", _i, [_vp, _vp, _vp, _vp, _i, _vp, _vp, _vp]),
    ("temx_tracer_stage3", _i, [_vp, _vp, _vp, _vp, _vp]),
    ("temx_tracer_stage1_sums", _i, [_vp, _vp, _vp, _vp, _i, _vp, _vp]),
    ("temx_tracer_stage2_from_sums", _i, [_vp, _vp, _vp, _vp]),
    ("temx_tracer_run", _i, [_vp, _vp, _vp, _vp, _i, _vp, _vp, _vp]),
    ("temx_tem_tracer_stage1", _i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp]),
    ("temx_tem_tracer_run", _i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp]),
    ("temx_tracer_eddy", _i, [_vp, _vp, _vp, _vp, _i, C.POINTER(_vp), _vp]),
    ("temx_status", _i, [_vp, C.POINTER(_i), _vp]),
    ("temx_synth_fields", _i, [_i, _i64, _i, _i64, _i64, _vp, _vp, _vp, _i, _u64, _vp, _vp, _vp, _vp, _vp]),
    ("temx_mfma_f64_peak", _i, [_i, _i, _dp]),
    ("temx_kernel_timing", _i, [_vp, _i]),
    ("temx_kernel_timing_read", _i, [_vp, _i, _dp, C.POINTER(_i)]),
    ("temx_selftest_exception", _i, [_i]),
]


class TemxError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("%s (%d): %s" % (ERRORS.get(code, "TEMX_E?"), code, msg))
        self.code = code


_lib = None


def load():
    """Load libtemx.so (once).  Raises RuntimeError when the HIP extension is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            "pytemdiags_amd: HIP extension %s is missing. Build it with `make -C pytemdiags_amd/csrc` "
            "or `python -c 'import __graft_entry__ as g; g.build()'`. There is no CPU fallback." % LIB_PATH)
    # PyTorch-ROCm ships its own HIP runtime.  Two runtimes in one process do not both see the GPU, so
    # let torch load first: libtemx.so's libamdhip64 dependency then binds to the runtime already mapped.
    try:
        import torch  # noqa: F401
    except ImportError:      # plain-C / ctypes-only consumers use the system runtime
        pass
    lib = C.CDLL(LIB_PATH)
    ab_build = "TEMX_LIB" in os.environ      # A/B builds of another source tree (development only)
    lib.temx_version.restype = _i
    have = int(lib.temx_version())
    if have != ABI_VERSION:
        msg = "pytemdiags_amd: %s reports ABI version %d, these bindings expect %d" % (LIB_PATH, have, ABI_VERSION)
        if not ab_build:
            raise RuntimeError(msg + "; rebuild it (make -C pytemdiags_amd/csrc)")
        import warnings
        warnings.warn(msg + " (TEMX_LIB build: continuing)")
    for name, res, args in SIGNATURES:
        if ab_build and not hasattr(lib, name):
            # fail at the call, with the same message on every rank, instead of an AttributeError somewhere
            # inside a collective sequence
            def _missing(*_a, _n=name):
                raise TemxError(-6, "symbol %s is missing in the TEMX_LIB build %s" % (_n, LIB_PATH))
            setattr(lib, name, _missing)
            continue
        fn = getattr(lib, name)          # AttributeError here = header/library mismatch
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc):
    if rc != 0:
        raise TemxError(rc, load().temx_last_error().decode("utf-8", "replace"))
