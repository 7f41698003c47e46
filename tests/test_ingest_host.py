"""Host-only pieces of the real-audio ingest: the resampler that stands in for the absent resampy package
and the standard-library WAV decoder.  No GPU: neither function touches the engine."""
import wave

import numpy as np


def test_resample_fallback_reproduces_a_band_limited_signal():
    from pyaudiolocalization_amd.signal_processing import resample_audio
    fs0, fs1 = 8000.0, 16000.0
    t0 = np.arange(4000) / fs0
    x = np.sin(2 * np.pi * 440.0 * t0) + 0.3 * np.sin(2 * np.pi * 1234.0 * t0)
    y = resample_audio(x, fs0, fs1)
    assert y.shape == (8000,)
    t1 = np.arange(8000) / fs1
    want = np.sin(2 * np.pi * 440.0 * t1) + 0.3 * np.sin(2 * np.pi * 1234.0 * t1)
    assert np.max(np.abs(y[400:-400] - want[400:-400])) < 2e-3       # (not resampy's kaiser_best: parity unpinned)
    z = resample_audio(x, 44100.0, 16000.0)                           # a non-trivial ratio: 160 / 441
    assert abs(len(z) - round(4000 * 160 / 441)) <= 1


def test_wav_decoder_scales_like_soundfile(tmp_path):
    from pyaudiolocalization_amd.utils import _read_wav_pcm
    q = np.array([[0, -32768], [32767, 1], [-1, 12345]], dtype="<i2")
    path = tmp_path / "s.wav"
    with wave.open(str(path), "wb") as w:
        w.setnchannels(2); w.setsampwidth(2); w.setframerate(22050); w.writeframes(q.tobytes())
    data, fs = _read_wav_pcm(str(path))
    assert fs == 22050 and data.shape == (3, 2)
    assert np.array_equal(data, q.astype(np.float64) / 32768.0)
