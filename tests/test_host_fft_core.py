"""Runs the workgroup FFT stage code of csrc/fft_core.h lane by lane on the host (tests/host)."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("hipcc") is None, reason="hipcc not on PATH")
def test_stage_code_matches_naive_dft(tmp_path):
    exe = tmp_path / "test_fft_core"
    subprocess.run(["hipcc", "-O2", "--offload-arch=gfx950", "-I", os.path.join(ROOT, "pyaudiolocalization_amd", "csrc"),
                    os.path.join(ROOT, "tests", "host", "test_fft_core.cpp"), "-o", str(exe)], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout
    assert "ALL OK" in out, out


@pytest.mark.skipif(shutil.which("hipcc") is None, reason="hipcc not on PATH")
def test_rader89_column_transform_matches_naive_dft(tmp_path):
    """csrc/pfa_rader89.h: the 89-point column DFT as Rader's 8 x 11 convolution over four wavefronts, exchanges emulated."""
    exe = tmp_path / "test_rader89"
    subprocess.run(["hipcc", "-O2", "--offload-arch=gfx950", "-w", "-I", os.path.join(ROOT, "pyaudiolocalization_amd", "csrc"),
                    os.path.join(ROOT, "tests", "host", "test_rader89.cpp"), "-o", str(exe)], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout
    assert "ALL OK" in out, out
