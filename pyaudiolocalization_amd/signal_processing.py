"""Drop-in for the reference's signal_processing.py on the hot path.

fractional_delay, normalize_signal, dynamic_range_compression and noise_reduction run on the
HIP engine; the one-off signal generators stay on the host (one base signal per run,
SURVEY.md section 2) and are restated here so that scripts written against the reference keep
working.  Citations are file:line into the reference.
"""
from __future__ import annotations

import logging

import numpy as np

from .engine import default_engine


# ---------------------------------------------------------------- generators (host, not hot path)
def generate_pink_noise(fs: float, duration: float) -> np.ndarray:
    """1/sqrt(f)-shaped white noise, normalised and compressed (signal_processing.py:11-23)."""
    count = int(fs * duration)
    spectrum = np.fft.rfft(np.random.randn(count))
    f = np.fft.rfftfreq(count, d=1.0 / fs)
    shape = np.zeros_like(f)
    shape[1:] = 1.0 / np.sqrt(f[1:])
    return dynamic_range_compression(normalize_signal(np.fft.irfft(spectrum * shape, n=count)))


def generate_realistic_speech(fs: float, duration: float) -> np.ndarray:
    """Three Hann-windowed formants + random transients + 5 % pink noise (signal_processing.py:38-64)."""
    from scipy.signal import get_window
    t = np.linspace(0, duration, int(fs * duration), endpoint=False)
    voiced = np.zeros_like(t)
    for freq, amp, phase in ((800, 1.0, 0.0), (1150, 0.8, np.pi / 4), (2900, 0.5, np.pi / 2)):
        voiced += amp * np.sin(2 * np.pi * freq * t + phase)
    voiced *= get_window("hann", len(t))
    burst_len = int(0.01 * fs)
    bursts = np.zeros_like(t)
    for _ in range(int(duration * 5)):
        at = np.random.randint(0, len(t) - burst_len)
        bursts[at:at + burst_len] += np.random.normal(0, 1, burst_len) * np.hanning(burst_len)
    mix = voiced + bursts + generate_pink_noise(fs, duration) * 0.05
    return dynamic_range_compression(normalize_signal(mix))


def generate_signal(signal_type: str, fs: float, duration: float, freq: float) -> np.ndarray:
    """sine / noise / linear chirp f..5f / speech (signal_processing.py:25-36)."""
    t = np.linspace(0, duration, int(fs * duration), endpoint=False)
    if signal_type == "sine":
        return np.sin(2 * np.pi * freq * t)
    if signal_type == "noise":
        return np.random.normal(0, 1, size=t.shape)
    if signal_type == "chirp":
        from scipy.signal import chirp
        return chirp(t, f0=freq, f1=freq * 5, t1=duration, method="linear")
    if signal_type == "speech":
        return generate_realistic_speech(fs, duration)
    raise ValueError("Unknown signal type. Available types: 'sine', 'noise', 'chirp', 'speech'")


# ---------------------------------------------------------------- hot path (HIP engine)
def fractional_delay(signal: np.ndarray, delay: float, fs: float) -> np.ndarray:
    """FFT(2N) phase-ramp delay + 1 % linear fades (signal_processing.py:66-80) on the GPU."""
    return default_engine().fractional_delay(np.asarray(signal, dtype=np.float64), float(delay), fs)


def normalize_signal(signal: np.ndarray) -> np.ndarray:
    """x / max|x|, identity for an all-zero signal (signal_processing.py:82-86)."""
    return default_engine().normalize_compress(signal, normalize_only=True)


def dynamic_range_compression(signal: np.ndarray, threshold: float = 0.8, epsilon: float = 1e-8) -> np.ndarray:
    """sign(x) log1p(|x|/threshold + epsilon), renormalised (signal_processing.py:88-94)."""
    return default_engine().normalize_compress(signal, normalize_only=False, threshold=threshold, epsilon=epsilon)


def dynamic_range_compression_soft_clip(signal: np.ndarray, threshold: float = 0.8) -> np.ndarray:
    """Unused by the reference (imported at main.py:12, never called; signal_processing.py:96-103)."""
    x = normalize_signal(signal)
    mag = np.abs(x)
    return np.where(mag > threshold, np.sign(x) * (threshold + (mag - threshold) * 0.5), x)


_KAISER_BEST = {}


def _kaiser_best_filter():
    """Right wing of the `kaiser_best` interpolation filter from its published design parameters (resampy's documentation of
    its shipped filters): a Kaiser-windowed sinc with 64 zero crossings, 2^9 table entries per crossing, beta =
    14.769656459379492 and roll-off 0.9475937167399596.  resampy ships the table as a data file; this one is generated."""
    if not _KAISER_BEST:
        from scipy.signal.windows import kaiser  # noqa: PLC0415
        num_zeros, precision, beta, rolloff = 64, 9, 14.769656459379492, 0.9475937167399596
        num_table = 2 ** precision
        n = num_table * num_zeros
        sinc_win = rolloff * np.sinc(rolloff * np.linspace(0, num_zeros, num=n + 1, endpoint=True))
        taper = kaiser(2 * n + 1, beta)[n:]                       # right half of the symmetric window
        _KAISER_BEST["win"] = taper * sinc_win
        _KAISER_BEST["num_table"] = num_table
    return _KAISER_BEST["win"], _KAISER_BEST["num_table"]


def resample_kaiser_best(data: np.ndarray, original_fs: float, target_fs: float) -> np.ndarray:
    """Band-limited sinc interpolation along the last axis in the form resampy publishes for `resample(..., filter=
    'kaiser_best')` (Smith's algorithm): output sample t sits at time t / ratio of the input; both wings of the
    windowed sinc are walked in steps of `scale x table entries` with linear interpolation between table entries, the
    filter is scaled by the ratio when downsampling, and the output has int(n x ratio) samples.  Vectorised over the
    output samples (one pass per filter tap).  PARITY UNPINNED: resampy is not importable here, there is no output of it
    to compare with - tests/test_ingest_host.py checks band-limited reconstruction and the published conventions."""
    x = np.asarray(data, dtype=np.float64)
    ratio = float(target_fs) / float(original_fs)
    if ratio <= 0:
        raise ValueError("Invalid sample rates")
    n_orig = x.shape[-1]
    n_out = int(n_orig * ratio)
    if n_out < 1:
        raise ValueError(f"Input signal length={n_orig} is too small to resample from {original_fs}->{target_fs}")
    win, num_table = _kaiser_best_filter()
    interp_win = win * ratio if ratio < 1 else win
    interp_delta = np.diff(interp_win, append=interp_win[-1])
    scale = min(1.0, ratio)
    index_step = int(scale * num_table)
    nwin = interp_win.shape[0]
    t_reg = np.arange(n_out) * (1.0 / ratio)
    n = t_reg.astype(np.int64)
    lead = x.shape[:-1]
    x2 = x.reshape(-1, n_orig)
    y = np.zeros((x2.shape[0], n_out))

    def wing(frac, count, sign, first):
        index_frac = frac * num_table
        offset = index_frac.astype(np.int64)
        eta = index_frac - offset
        taps = np.minimum(count, (nwin - offset) // index_step)
        for i in range(int(taps.max()) if taps.size else 0):
            live = i < taps
            idx = np.where(live, offset + i * index_step, 0)
            weight = np.where(live, interp_win[idx] + eta * interp_delta[idx], 0.0)
            src = np.where(live, first + sign * i, 0)
            y[:, :] += weight[None, :] * x2[:, src]

    frac = scale * (t_reg - n)
    wing(frac, n + 1, -1, n)                                      # left wing: x[n], x[n - 1], ...
    wing(scale - frac, n_orig - n - 1, +1, n + 1)                 # right wing: x[n + 1], x[n + 2], ...
    return y.reshape(lead + (n_out,))


def resample_audio(data: np.ndarray, original_fs: float, target_fs: float) -> np.ndarray:
    """signal_processing.py:105-107 (outside the hot path): resampy's kaiser_best when that optional package
    is installed; without it (the build image) the own implementation of the same published algorithm and filter design
    (`resample_kaiser_best`: parity unpinned, no resampy output exists here to compare with)."""
    try:
        import resampy  # noqa: PLC0415 - optional dependency
        return resampy.resample(data, original_fs, target_fs, filter="kaiser_best")
    except ImportError:
        return resample_kaiser_best(data, original_fs, target_fs)


def _filter_design(fs: float, method: str, lowcut: float, highcut: float, filter_order: int):
    """11-tap Butterworth band-pass or FIR design + lfilter_zi state: host scalar work."""
    from scipy.signal import butter, firwin, lfilter_zi
    nyquist = 0.5 * fs
    band = [lowcut / nyquist, highcut / nyquist]
    if method == "butterworth":
        b, a = butter(5, band, btype="band")                                   # signal_processing.py:127
    else:
        b, a = firwin(filter_order, band, pass_zero=False), np.array([1.0])   # signal_processing.py:132
    return b, a, lfilter_zi(b, a)


def noise_reduction_rows(rows: np.ndarray, fs: float, method: str = "butterworth", lowcut: float = 300,
                         highcut: float = 3400, filter_order: int = 101) -> np.ndarray:
    """Batched form: every row of rows[R][N] through the same filter in one launch."""
    if method in ("butterworth", "fir"):
        b, a, zi = _filter_design(fs, method, lowcut, highcut, filter_order)
        return default_engine().filtfilt(b, a, zi, rows)
    if method == "wiener":
        return default_engine().wiener3(rows)
    raise ValueError("Unknown filter method. Available methods: 'butterworth', 'fir', 'wiener'")


def noise_reduction(signal: np.ndarray, fs: float, method: str = "butterworth", lowcut: float = 300,
                    highcut: float = 3400, filter_order: int = 101) -> np.ndarray:
    """Zero-phase Butterworth / FIR (filtfilt) or Wiener-3 (signal_processing.py:109-138) on the GPU."""
    return noise_reduction_rows(np.asarray(signal, dtype=np.float64), fs, method, lowcut, highcut, filter_order)
