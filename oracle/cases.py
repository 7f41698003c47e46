"""Synthetic inputs of the BASELINE.json configs  --  TEST INFRASTRUCTURE ONLY.

Pure NumPy (plus the oracle's own signal helpers); no reference code is needed to rebuild an
input, so the same inputs exist on the GPU box.  ``make_golden.py`` feeds them to the imported
reference in the build container; tests feed them to the oracle and to the HIP engine.
"""
from __future__ import annotations

import numpy as np

from . import pal_oracle as O

DEFAULT_PLANES = [                                   # values of main.py:41-43
    {"plane": [1, 0, 0, -5], "material": "wood"},
    {"plane": [0, 1, 0, -5], "material": "metal"},
    {"plane": [0, 0, 1, -5], "material": "wood"},
]
LOW_LOSS = {
    "air": {"absorption": 0.01, "freq": 0.0},
    "wood": {"absorption": 0.05, "freq": 1e-6},
    "metal": {"absorption": 0.1, "freq": 1e-6},
}
C_SOUND = O.speed_of_sound(20, 50)                   # 343.62 m/s


def c1_config() -> dict:
    """Values of the reference's module-level ``config`` (main.py:26-64), analysis/plots off."""
    return {
        "fs": 44100, "duration": 1.0, "celsius": 20, "humidity": 50,
        "mic_positions": [[0.0, 0.0, 0.0], [1.0, 0.0, 0.0], [0.0, 1.0, 0.0], [0.0, 0.0, 1.0]],
        "source_position": [0.5, 0.5, 0.5], "signal_type": "sine", "freq": 1000,
        "reflective_planes": [dict(p) for p in DEFAULT_PLANES],
        "calibration": {"signal_type": "chirp", "freq_start": 500, "freq_end": 5000,
                        "attenuation_factor": 1.0, "noise_level": 0.01},
        "localization": {"max_reflections": 3, "filter_method": "butterworth", "absorption_threshold": 0.01,
                         "analyze_correlation": False, "visualize_correlation": False,
                         "clustering_method": "kmeans", "clustering_eps": 0.001, "clustering_min_samples": 2,
                         "max_expected_delay": 0.05},
    }


def c2_config() -> dict:
    cfg = c1_config()
    cfg.update({"fs": 48000, "mic_positions": np.random.default_rng(2).uniform(-0.5, 0.5, (8, 3)).tolist(),
                "source_position": [1.0, 2.0, 0.5], "signal_type": "chirp", "freq": 500})
    return cfg


def grid_array_64() -> np.ndarray:
    ax = (np.arange(8) - 3.5) * 0.1
    gx, gy = np.meshgrid(ax, ax, indexing="ij")
    return np.stack([gx.ravel(), gy.ravel(), np.zeros(64)], axis=1)


def c3_sources() -> np.ndarray:
    return np.random.default_rng(3).uniform([-3, -3, 0.5], [3, 3, 3], (16, 3))


def c3_base(trial: int) -> np.ndarray:
    return np.random.default_rng(100 + trial).standard_normal(24000)


def c3_config(trial: int) -> dict:
    cfg = c1_config()
    cfg.update({"fs": 48000, "duration": 0.5, "mic_positions": grid_array_64().tolist(),
                "source_position": c3_sources()[trial].tolist(), "signal_type": "noise", "freq": 1000,
                "reflective_planes": []})
    return cfg


def fibonacci_sphere(count: int, radius: float) -> np.ndarray:
    i = np.arange(count) + 0.5
    phi = np.arccos(1 - 2 * i / count)
    theta = np.pi * (1 + 5 ** 0.5) * i
    return radius * np.stack([np.cos(theta) * np.sin(phi), np.sin(theta) * np.sin(phi), np.cos(phi)], axis=1)


def c4_frames(mics: int = 256) -> np.ndarray:
    """First ``mics`` microphones of C4: delayed chirp 500->2500 Hz + white noise at -20 dB, 96 kHz, 1 s."""
    fs, n = 96000, 96000
    pos = fibonacci_sphere(256, 0.5)[:mics]
    chirp = O.generate_signal("chirp", fs, 1.0, 500)
    noise = np.random.default_rng(4)
    out = np.empty((mics, n))
    for m in range(mics):
        d = np.linalg.norm(np.array([2.0, 1.0, 0.5]) - pos[m])
        out[m] = O.fractional_delay(chirp, d / C_SOUND, fs) + 0.1 * noise.standard_normal(n)
    return out


def c5_source(frame: int) -> np.ndarray:
    steps = np.random.default_rng(5).normal(0.0, 0.02, (1024, 3))
    return np.array([1.0, 2.0, 0.5]) + steps[: frame + 1].sum(axis=0)


def c5_base(frame: int) -> np.ndarray:
    return np.random.default_rng(1000 + frame).standard_normal(12000)


def metric_frames(batch: int, mics: int = 64, n: int = 44100, first: int = 0) -> np.ndarray:
    """Frames of the metric run (defined next to bench.py's generator so both use one recipe)."""
    from pyaudiolocalization_amd.synthetic import metric_frames as make
    return make(batch, mics, n, first)


SHOEBOX = [{"plane": [1, 0, 0, -5], "material": "wood"}, {"plane": [1, 0, 0, 4], "material": "wood"},
           {"plane": [0, 1, 0, -5], "material": "metal"}, {"plane": [0, 1, 0, 3], "material": "metal"},
           {"plane": [0, 0, 1, -3], "material": "wood"}, {"plane": [0, 0, 1, 1], "material": "air"}]


def selection_edge_cases(count: int = 160):
    """Tiny signal pairs and parameter mixes that force every branch of the peak-selection fallback chain
    (SURVEY 8c item 3): unequal lengths, sines, 'adaptive' / unknown methods, tight and zero windows."""
    rng = np.random.default_rng(11)
    out = []
    t = 0
    while len(out) < count:
        t += 1
        n1 = int(rng.integers(24, 200))
        n2 = n1 if t % 3 else int(rng.integers(24, 200))
        fs = float(rng.choice([1000.0, 2000.0, 8000.0, 48000.0]))
        a, b = rng.standard_normal(n1), rng.standard_normal(n2)
        if t % 5 == 0:
            a = np.sin(0.31 * np.arange(n1))
            b = np.sin(0.31 * np.arange(n2) + 0.4)
        out.append({"t": t, "a": a, "b": b, "fs": fs, "med": [None, 0.05, 0.01, 0.001, 0.0][t % 5],
                    "method": ["median", "adaptive", "other"][t % 3], "mult": [1.0, 3.0, 25.0, 0.2][t % 4]})
    # degenerate inputs: no interior peak at all (delta / all-zero / one-sided sequences), even n
    ramp = np.arange(1.0, 41.0)
    extra = [(np.ones(40), np.ones(40), None), (np.zeros(50), np.zeros(50), 0.01), (ramp, ramp[::-1].copy(), None),
             (np.ones(33), np.ones(32), 0.005), (rng.standard_normal(64), np.zeros(64), None),
             (rng.standard_normal(65), rng.standard_normal(64), 0.002)]
    for k, (a, b, med) in enumerate(extra):
        out.append({"t": 10000 + k, "a": a, "b": b, "fs": 2000.0, "med": med, "method": "median", "mult": 1.0})
    return out


def waveform_digest(x: np.ndarray) -> np.ndarray:
    """Small fingerprint of a waveform: length, sum, sum of squares, max |x| and 64 strided samples."""
    x = np.asarray(x, dtype=np.float64)
    idx = np.linspace(0, x.shape[0] - 1, 64).astype(np.int64)
    return np.concatenate(([x.shape[0], x.sum(), (x * x).sum(), np.abs(x).max()], x[idx]))


def calibration_cases():
    """(tag, config, seed) for calibration.py's run_calibration: a chirp over six scattered microphones and an impulse
    over four; `seed` goes to np.random.seed right before the call (the reference draws its noise from the global RNG)."""
    rng = np.random.default_rng(77)
    mics6 = [list(map(float, m)) for m in rng.uniform(-1.5, 1.5, (6, 3))]
    mics4 = [[0.0, 0.0, 0.0], [1.0, 0.0, 0.0], [0.0, 1.2, 0.0], [0.3, 0.4, 0.9]]
    base = {"celsius": 20.0, "humidity": 50.0}
    chirp = dict(base, fs=16000, duration=0.25, source_position=[2.5, -1.0, 0.8], mic_positions=mics6,
                 calibration={"signal_type": "chirp", "freq_start": 500, "freq_end": 5000, "attenuation_factor": 1.0,
                              "noise_level": 0.01})
    impulse = dict(base, fs=8000, duration=0.2, source_position=[1.7, 2.2, -0.4], mic_positions=mics4,
                   calibration={"signal_type": "impulse", "attenuation_factor": 0.8, "noise_level": 0.002})
    return [("chirp6", chirp, 4321), ("impulse4", impulse, 99)]


def calibration_noise(cfg, seed):
    """The noise the reference draws for `cfg` after np.random.seed(seed): one normal(0, level, N) per microphone."""
    n = int(cfg["fs"] * cfg["duration"])
    state = np.random.get_state()
    np.random.seed(seed)
    noise = np.array([np.random.normal(0, cfg["calibration"].get("noise_level", 0.01), size=n) for _ in cfg["mic_positions"]])
    np.random.set_state(state)
    return noise


def loc_config(analyze: bool) -> dict:
    """Small localisation case for the calibration correction and the per-pair metrics of main.py:147-157,209-222:
    5 scattered microphones, chirp 300 -> 1500 Hz, 8 kHz x 0.25 s, no reflections (fast enough for the reference's
    1000-shuffle bootstrap: 10 pairs x 1000 PHAT correlations of 4 k points)."""
    rng = np.random.default_rng(31)
    cfg = c1_config()
    cfg.update({"fs": 8000, "duration": 0.25, "mic_positions": rng.uniform(-0.6, 0.6, (5, 3)).tolist(),
                "source_position": [1.4, -0.8, 0.9], "signal_type": "chirp", "freq": 300, "reflective_planes": []})
    cfg["localization"] = dict(cfg["localization"], analyze_correlation=bool(analyze), visualize_correlation=False)
    return cfg


LOC_CALIBRATION = [{"delay": d, "amplitude": 1.0} for d in (0.0, 2.5e-4, -1.25e-4, 3.75e-4, -5.0e-4)]
LOC_SEED = 20240917          # np.random.seed right before localize_sound_source: the bootstrap shuffles use the global RNG


def unequal_sync_signals():
    """Recordings whose lengths differ by a few samples (what read_audio_files returns for real files): a common
    noise burst at different offsets, 8 kHz."""
    rng = np.random.default_rng(47)
    y = rng.standard_normal(3300)
    lens, offs = (3000, 2993, 3011, 2987, 3004), (120, 131, 117, 126, 139)
    return [y[o: o + n] + 0.05 * rng.standard_normal(n) for n, o in zip(lens, offs)], 8000
