"""MI355X-native GCC-PHAT TDOA engine behind PyAudioLocalization's function signatures.

    from pyaudiolocalization_amd import main, utils, signal_processing, materials, calibration

mirrors the reference's flat modules for the hot path named in BASELINE.json: all-pairs PHAT
cross-correlation / peak selection, the image-source multipath simulator, the Butterworth
prefilter and the synchronisation cross-correlation run as hand-written HIP kernels (gfx950)
behind the C ABI of include/pal_hip.h.  Importing the package needs libpal_hip.so to be built;
running any hot-path function needs an AMD GPU - there is no CPU fallback.
"""
from . import _ffi

_ffi.load()   # fail loudly at import time when the HIP library has not been built

from .engine import Engine, default_engine, make_params, pair_list  # noqa: E402
from ._ffi import RECORD, PalError  # noqa: E402

__all__ = ["Engine", "default_engine", "make_params", "pair_list", "RECORD", "PalError"]
