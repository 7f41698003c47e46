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
    assert np.max(np.abs(y[400:-400] - want[400:-400])) < 2e-5       # (the published kaiser_best design: parity unpinned, see the docstring)
    z = resample_audio(x, 44100.0, 16000.0)                           # a non-trivial ratio: 160 / 441
    assert len(z) == int(4000 * 16000.0 / 44100.0)                    # resampy's length rule: int(n x ratio)


def test_kaiser_best_resampler_conventions():
    """What the published algorithm fixes besides the filter: the output length int(n x ratio), output sample t at input time
    t / ratio (so ratio 1 is the identity up to the filter's pass-band ripple, and the first sample stays the first sample), a
    pass band of 0.9476 x the lower Nyquist rate with ~ -100 dB beyond it when downsampling, linearity, batches along the last
    axis, and agreement with SciPy's polyphase resampler in the interior of a band-limited signal."""
    from scipy.signal import resample_poly
    from pyaudiolocalization_amd.signal_processing import resample_kaiser_best
    rng = np.random.default_rng(3)
    fs0 = 48000.0
    t = np.arange(9600) / fs0
    x = np.sin(2 * np.pi * 1000.0 * t) + 0.5 * np.cos(2 * np.pi * 5200.0 * t + 0.3)
    same = resample_kaiser_best(x, fs0, fs0)
    assert same.shape == x.shape and np.max(np.abs(same[300:-300] - x[300:-300])) < 1e-5
    down = resample_kaiser_best(x, fs0, 44100.0)
    assert down.shape == (int(9600 * 44100.0 / 48000.0),)
    td = np.arange(down.size) / 44100.0
    want = np.sin(2 * np.pi * 1000.0 * td) + 0.5 * np.cos(2 * np.pi * 5200.0 * td + 0.3)
    # (downsampling walks the table in steps of int(scale x 512) = 470 entries where 470.4 would be exact: the published
    #  algorithm's own quantisation, 5e-4 here; upsampling steps by exactly 512 and is good to 2e-5 - first test)
    assert np.max(np.abs(down[300:-300] - want[300:-300])) < 1e-3
    ref = resample_poly(x, 147, 160, window=("kaiser", 14.769656459379492))
    assert np.max(np.abs(down[300:-300] - ref[300:down.size - 300])) < 2e-3
    # a tone above the new Nyquist rate is removed (anti-aliasing: the filter is scaled to the lower rate)
    alias = resample_kaiser_best(np.sin(2 * np.pi * 15000.0 * t), fs0, 16000.0)
    assert np.max(np.abs(alias[200:-200])) < 5e-4                    # (-66 dB or better: the same table quantisation)
    # linear, and rows of a batch are resampled independently
    a, b = rng.standard_normal(2000), rng.standard_normal(2000)
    both = resample_kaiser_best(np.stack([a, b]), 16000.0, 22050.0)
    assert both.shape == (2, int(2000 * 22050.0 / 16000.0))
    assert np.allclose(both[0], resample_kaiser_best(a, 16000.0, 22050.0), rtol=0, atol=1e-14)
    assert np.allclose(resample_kaiser_best(2.0 * a - b, 16000.0, 22050.0), 2.0 * both[0] - both[1], rtol=0, atol=1e-12)


def test_wav_decoder_scales_like_soundfile(tmp_path):
    from pyaudiolocalization_amd.utils import _read_wav_pcm
    q = np.array([[0, -32768], [32767, 1], [-1, 12345]], dtype="<i2")
    path = tmp_path / "s.wav"
    with wave.open(str(path), "wb") as w:
        w.setnchannels(2); w.setsampwidth(2); w.setframerate(22050); w.writeframes(q.tobytes())
    data, fs = _read_wav_pcm(str(path))
    assert fs == 22050 and data.shape == (3, 2)
    assert np.array_equal(data, q.astype(np.float64) / 32768.0)
