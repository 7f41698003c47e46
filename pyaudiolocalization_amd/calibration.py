"""Drop-in for the reference's calibration.py on the HIP engine.

Same function names, argument order, defaults and result dicts as the reference; citations are file:line
into it.  The per-microphone fractional delays run as one batched ``pal_fractional_delay`` call, the M full
cross-correlations against the calibration signal as one ``pal_xcorr_vs_ref`` call (the synchronisation
kernel of utils.py:418-427: the calibration signal rides along as the reference row), the
normalise / compress pair on the device as in signal_processing.py.  The chirp itself and the Gaussian noise
are host one-offs (SciPy / NumPy, as in the reference).
"""
from __future__ import annotations

import logging
from typing import Dict, List, Optional, Sequence

import numpy as np

from .engine import default_engine
from .materials import material_properties as _default_materials
from .signal_processing import dynamic_range_compression, normalize_signal
from .utils import calculate_attenuation, speed_of_sound


def generate_calibration_signal(fs, duration=1.0, signal_type="chirp", freq_start=500, freq_end=5000) -> np.ndarray:
    """calibration.py:10-21."""
    from scipy.signal import chirp
    t = np.linspace(0, duration, int(fs * duration), endpoint=False)
    if signal_type == "chirp":
        calib_signal = chirp(t, f0=freq_start, f1=freq_end, t1=duration, method="linear")
    elif signal_type == "impulse":
        calib_signal = np.zeros_like(t)
        calib_signal[0] = 1.0
    else:
        raise ValueError("Unsupported calibration signal type. Use 'chirp' or 'impulse'.")
    return dynamic_range_compression(normalize_signal(calib_signal))


def simulate_calibration_recording(calib_signal, mic_positions, source_position, fs, c, attenuation_factor=1.0,
                                   noise_level=0.01, freq=None, material_properties=None, noise=None) -> List[np.ndarray]:
    """calibration.py:23-41: delayed, attenuated copies of the calibration signal plus Gaussian noise.

    ``noise`` (extension, optional ``[M][N]`` array) replaces the reference's unseeded ``np.random.normal`` draws
    - one draw of N samples per microphone, in microphone order - so that a run can be reproduced."""
    if freq is None:
        freq = 1000
    if material_properties is None:
        material_properties = _default_materials
    x = np.ascontiguousarray(calib_signal, dtype=np.float64)
    mics = [np.array(m, dtype=float) for m in mic_positions]
    dist = np.array([np.linalg.norm(np.array(source_position, dtype=float) - m) for m in mics])
    gains = np.array([attenuation_factor * calculate_attenuation(d, "air", freq, material_properties) for d in dist])
    delayed = default_engine().fractional_delay(np.tile(x, (len(mics), 1)), dist / c, fs)      # [M][N], one launch group
    recordings = []
    for m in range(len(mics)):
        rec = delayed[m] * gains[m]
        rec += np.random.normal(0, noise_level, size=rec.shape) if noise is None else np.asarray(noise[m], dtype=float)
        recordings.append(rec)
    return recordings


def analyze_calibration(recorded_signals: Sequence[np.ndarray], calib_signal, fs) -> List[Dict[str, float]]:
    """calibration.py:43-52: lag of the largest |cross-correlation| against the calibration signal and its height."""
    ref = np.ascontiguousarray(calib_signal, dtype=np.float64)
    recs = [np.ascontiguousarray(r, dtype=np.float64) for r in recorded_signals]
    if not recs:
        return []
    if any(r.shape != ref.shape for r in recs):
        raise ValueError("recordings and calibration signal must have the same length")   # (the reference's recordings always do)
    rows = np.vstack(recs + [ref])
    kpk, _win, pk, _refpk = default_engine().xcorr_vs_ref(rows, len(recs))
    n_ref = ref.shape[0]
    # np.float64 like the reference's lags[...] / fs and np.max(np.abs(corr))
    return [{"delay": np.float64(int(kpk[m]) - (n_ref - 1)) / fs, "amplitude": np.float64(pk[m])} for m in range(len(recs))]


def plot_calibration_results(results) -> None:
    """calibration.py:54-72 (host plotting, unchanged behaviour)."""
    import matplotlib.pyplot as plt
    delays = [res["delay"] for res in results]
    amplitudes = [res["amplitude"] for res in results]
    fig, ax1 = plt.subplots(figsize=(8, 5))
    indices = np.arange(len(results))
    ax1.bar(indices, delays, color="skyblue", alpha=0.7, label="Delay (s)")
    ax1.set_xlabel("Microphone Index")
    ax1.set_ylabel("Delay (s)", color="b")
    ax1.tick_params(axis="y", labelcolor="b")
    ax2 = ax1.twinx()
    ax2.plot(indices, amplitudes, "r-o", label="Amplitude")
    ax2.set_ylabel("Cross-correlation Amplitude", color="r")
    ax2.tick_params(axis="y", labelcolor="r")
    plt.title("Calibration Results per Microphone")
    fig.tight_layout()
    plt.show()


def run_calibration(config, noise: Optional[np.ndarray] = None):
    """calibration.py:74-104: signal -> simulated recordings -> per-microphone delay / amplitude."""
    fs = config["fs"]
    duration = config["duration"]
    c = speed_of_sound(config["celsius"], config["humidity"])
    cal = config["calibration"]
    calib_signal = generate_calibration_signal(fs, duration, signal_type=cal.get("signal_type", "chirp"),
                                               freq_start=cal.get("freq_start", 500), freq_end=cal.get("freq_end", 5000))
    logging.info("Calibration signal generated.")
    recorded = simulate_calibration_recording(calib_signal, config["mic_positions"], config["source_position"], fs, c,
                                              attenuation_factor=cal.get("attenuation_factor", 1.0),
                                              noise_level=cal.get("noise_level", 0.01), noise=noise)
    logging.info("Simulated calibration recordings created.")
    results = analyze_calibration(recorded, calib_signal, fs)
    logging.info("Calibration analysis completed.")
    return results, calib_signal, recorded
