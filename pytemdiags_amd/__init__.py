"""pytemdiags_amd -- MI355X-native drop-in for PyTEMDiags' TEM hot path.

    from pytemdiags_amd import TEMDiagnostics, sph_zonal_averager

Same classes, signatures and semantics as ``PyTEMDiags`` (jhollowed/PyTEMDiags); the numerics run
in hand-written HIP kernels for gfx950 behind a C ABI (include/temx.h, libtemx.so).  There is no
CPU fallback: the engine raises if the extension is not built or no GPU is visible.
"""
from .sph_zonal_mean import sph_zonal_averager
from .tem_diagnostics import TEMDiagnostics
from .containers import LabeledArray

__all__ = ["TEMDiagnostics", "sph_zonal_averager", "LabeledArray"]
__version__ = "0.1"
