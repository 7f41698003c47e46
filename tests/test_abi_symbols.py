"""The C-ABI library loads without a GPU and exports every symbol include/pal_hip.h declares."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "pal_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(pal_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_are_exported():
    names = declared_symbols()
    assert len(names) >= 30
    lib = ctypes.CDLL(os.path.join(ROOT, "pyaudiolocalization_amd", "libpal_hip.so"))
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing


def test_ctypes_binding_covers_the_header():
    from pyaudiolocalization_amd import _ffi
    assert sorted(_ffi.SIGNATURES) == declared_symbols()
    lib = _ffi.load()
    assert lib.pal_abi_version() == 1


def test_record_layout_matches_header():
    from pyaudiolocalization_amd import _ffi
    assert _ffi.RECORD.itemsize == 48
    assert [_ffi.RECORD.fields[k][1] for k in ("k_sel", "branch", "k_argmax", "n_sel", "cmax", "cmin", "snr", "sel_height")] \
        == [0, 4, 8, 12, 16, 24, 32, 40]
    assert ctypes.sizeof(_ffi.PhatParams) == 40


def test_engine_creation_fails_loudly_without_gpu():
    """No silent CPU fallback: on a box without a HIP device the engine refuses to start."""
    import pytest
    from pyaudiolocalization_amd import Engine, PalError, _ffi
    h = ctypes.c_void_p()
    count_ok = _ffi.load().pal_create(0, ctypes.byref(h)) == 0
    if count_ok:
        _ffi.load().pal_destroy(h)
        pytest.skip("a GPU is visible here")
    with pytest.raises(PalError):
        Engine(0)
