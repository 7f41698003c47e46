"""CPU oracle for the GCC-PHAT / multipath hot path  --  TEST INFRASTRUCTURE ONLY.

This file is a NumPy restatement of the reference's algorithm for the hot path
named in BASELINE.json.  It is the checker
the HIP engine is compared against.  Only ``tests/``, ``__graft_entry__.smoke``
and ``bench.py``'s ``cpu_baseline`` leg may import it; the product package
``pyaudiolocalization_amd`` never does (it fails loudly without the HIP library).

Pinning: the reference ships no tests and no golden vectors,
so the pin is (a) ``oracle/make_golden.py`` which imports the unmodified
reference in the build container and writes ``tests/golden/*.npz`` and (b)
``tests/test_oracle_golden.py`` which replays those fixtures everywhere.
Third-party numerics the reference calls (``numpy.fft`` = pocketfft,
``scipy.signal.butter``, ``scipy.interpolate.CubicSpline``, ``scipy.signal.chirp``,
``scipy.signal.firwin``) are called here as well; everything the GPU engine
implements itself (PHAT whitening, peak selection incl. ``find_peaks`` semantics,
SNR, image sources, fractional delay, normalise/compress, ``filtfilt``/``lfilter``,
Wiener-3, plain cross-correlation sync) is restated index by index.

Every function cites the reference lines (``/root/reference``) it follows.
"""
from __future__ import annotations

import math
from typing import Any, Dict, List, Optional, Sequence, Tuple

import numpy as np

# --------------------------------------------------------------------------
# branch codes of the peak-selection fallback chain (utils.py:153-172)
# --------------------------------------------------------------------------
BR_ALT_THRESHOLD = 1      # primary threshold gave no peaks, mean(|corr|) did        (utils.py:153-156)
BR_ARGMAX_NO_PEAKS = 2    # no peaks with either threshold -> argmax(corr)            (utils.py:157-160)
BR_WINDOW_RETRY = 4       # no peak inside the window -> retry with mean threshold    (utils.py:164-168)
BR_ARGMAX_WINDOW = 8      # still none inside the window -> argmax(corr), un-windowed (utils.py:169-172)

MATERIALS_DEFAULT = {      # values of materials.py:3-15 (data, not code)
    "air": {"absorption": 0.01, "freq": 0.1},
    "wood": {"absorption": 0.05, "freq": 0.8},
    "metal": {"absorption": 0.1, "freq": 0.6},
}


# --------------------------------------------------------------------------
# a1  phat_correlation                                   utils.py:108-119
# --------------------------------------------------------------------------
def phat_correlation(sig1: np.ndarray, sig2: np.ndarray) -> np.ndarray:
    """Unshifted PHAT sequence on the exact n = n1+n2-1 grid (utils.py:112-118)."""
    a = np.asarray(sig1, dtype=np.float64)
    b = np.asarray(sig2, dtype=np.float64)
    n = a.shape[0] + b.shape[0] - 1
    fa = np.fft.fft(a, n=n)
    fb = np.fft.fft(b, n=n)
    cross = fa * np.conj(fb)
    cross /= np.abs(cross) + 1e-10
    return np.fft.ifft(cross).real


# --------------------------------------------------------------------------
# scipy.signal.find_peaks(x, height=h, distance=d) semantics, restated
#   (call sites utils.py:152,156,167)
# --------------------------------------------------------------------------
def local_maxima(x: np.ndarray) -> np.ndarray:
    """Strict interior local maxima; a flat top counts once at its (floor) midpoint.

    End points never qualify.  Vectorised: a rising edge at i (x[i-1] < x[i]) opens a
    plateau that closes at the first r > i with x[r] != x[i]; it is a peak iff x[r] < x[i].
    """
    x = np.asarray(x, dtype=np.float64)
    n = x.shape[0]
    if n < 3:
        return np.zeros(0, dtype=np.int64)
    d = np.diff(x)
    rising = np.flatnonzero(d[:-1] > 0) + 1          # i in [1, n-2] with x[i-1] < x[i]
    if rising.size == 0:
        return np.zeros(0, dtype=np.int64)
    change = np.flatnonzero(d != 0)                   # j with x[j] != x[j+1]
    # first change position >= i gives the plateau's right edge r-1
    pos = np.searchsorted(change, rising, side="left")
    ok = pos < change.size
    rising = rising[ok]
    right = change[pos[ok]]                           # last index of the plateau
    falling = d[right] < 0
    left = rising[falling]
    right = right[falling]
    return ((left + right) // 2).astype(np.int64)


def select_by_distance(peaks: np.ndarray, heights: np.ndarray, distance: int) -> np.ndarray:
    """Greedy suppression in descending height: a kept peak removes every peak closer than
    ``distance`` samples (scipy ``_select_by_peak_distance``).  Equal heights: the later
    position wins (stable ascending sort walked from the end) - exact ties are unpinned."""
    m = peaks.shape[0]
    keep = np.ones(m, dtype=bool)
    order = np.argsort(heights, kind="stable")
    for idx in order[::-1]:
        if not keep[idx]:
            continue
        k = idx - 1
        while k >= 0 and peaks[idx] - peaks[k] < distance:
            keep[k] = False
            k -= 1
        k = idx + 1
        while k < m and peaks[k] - peaks[idx] < distance:
            keep[k] = False
            k += 1
    return keep


def find_peaks_height_distance(x: np.ndarray, height: float, distance: int) -> Tuple[np.ndarray, np.ndarray]:
    if distance < 1:
        raise ValueError("`distance` must be greater or equal to 1")
    pk = local_maxima(x)
    h = x[pk]
    sel = h >= height
    pk, h = pk[sel], h[sel]
    keep = select_by_distance(pk, h, distance)
    return pk[keep], h[keep]


# --------------------------------------------------------------------------
# a2  get_time_delays_phat                               utils.py:121-181
# --------------------------------------------------------------------------
def primary_threshold(corr: np.ndarray, method: str, multiplier: float) -> float:
    mag = np.abs(corr)
    if method == "adaptive":                          # utils.py:146-147
        return multiplier * (np.mean(mag) + np.std(mag))
    return multiplier * np.median(mag)                # 'median' and every other string (utils.py:144-149)


def select_peaks(corr: np.ndarray, n2: int, fs: float, num_peaks: int = 1,
                 threshold_method: str = "median", threshold_multiplier: float = 1.0,
                 max_expected_delay: Optional[float] = None) -> Tuple[np.ndarray, int]:
    """Index-level restatement of utils.py:144-179.  Returns (selected array indices k, branch)."""
    n = corr.shape[0]
    lag_t = (np.arange(n, dtype=np.int64) - (n2 - 1)) / fs       # correlation_lags(..)/fs  (utils.py:141-142)
    dist = int(fs * 0.001)                                        # utils.py:151
    branch = 0
    pk, h = find_peaks_height_distance(corr, primary_threshold(corr, threshold_method, threshold_multiplier), dist)
    if pk.size == 0:
        branch |= BR_ALT_THRESHOLD
        pk, h = find_peaks_height_distance(corr, np.mean(np.abs(corr)), dist)
        if pk.size == 0:
            return np.array([int(np.argmax(corr))], dtype=np.int64), branch | BR_ARGMAX_NO_PEAKS
    if max_expected_delay is not None:
        inside = np.abs(lag_t[pk]) <= max_expected_delay
        if not inside.any():
            branch |= BR_WINDOW_RETRY
            pk, h = find_peaks_height_distance(corr, np.mean(np.abs(corr)), dist)
            inside = np.abs(lag_t[pk]) <= max_expected_delay
            if not inside.any():
                return np.array([int(np.argmax(corr))], dtype=np.int64), branch | BR_ARGMAX_WINDOW
        pk, h = pk[inside], h[inside]
    order = np.argsort(h, kind="stable")[::-1]                    # utils.py:176
    return pk[order][:num_peaks].astype(np.int64), branch


def get_time_delays_phat(sig1, sig2, fs, num_peaks=1, threshold_method="median",
                         threshold_multiplier=1.0, max_expected_delay=None):
    """Same return contract as utils.py:121-181 plus nothing else."""
    corr = phat_correlation(sig1, sig2)
    n2 = len(sig2)
    ks, _ = select_peaks(corr, n2, fs, num_peaks, threshold_method, threshold_multiplier, max_expected_delay)
    lag_t = (np.arange(corr.shape[0], dtype=np.int64) - (n2 - 1)) / fs
    return list(lag_t[ks]), corr, lag_t


# --------------------------------------------------------------------------
# a3  epilogue metrics            utils.py:228-250, main.py:223
# --------------------------------------------------------------------------
def compute_peak_to_peak_ratio(corr: np.ndarray) -> float:
    lo = np.min(corr)
    return np.inf if lo == 0 else np.max(corr) / abs(lo)


def compute_snr(corr: np.ndarray) -> float:
    n = corr.shape[0]
    pk = int(np.argmax(corr))
    w = max(1, int(0.01 * n))
    lo, hi = max(0, pk - w), min(n, pk + w)
    noise = np.std(np.concatenate((corr[:lo], corr[hi:])))
    return np.inf if noise == 0 else corr[pk] / noise


def pair_record(corr: np.ndarray, n2: int, fs: float, threshold_method="median", threshold_multiplier=1.0,
                max_expected_delay=None) -> Dict[str, Any]:
    """Everything the batched engine reports per pair (k_sel, branch, max, min, argmax, snr)."""
    ks, br = select_peaks(corr, n2, fs, 1, threshold_method, threshold_multiplier, max_expected_delay)
    return {"k_sel": int(ks[0]), "branch": int(br), "cmax": float(np.max(corr)), "cmin": float(np.min(corr)),
            "k_argmax": int(np.argmax(corr)), "snr": float(compute_snr(corr))}


def all_pairs(frames: np.ndarray, fs: float, threshold_method="median", threshold_multiplier=1.0,
              max_expected_delay=None) -> Dict[str, np.ndarray]:
    """Row-major i<j pair loop of main.py:202-228 over frames[M][L] -> table of P records."""
    m = frames.shape[0]
    recs = []
    for i in range(m):
        for j in range(i + 1, m):
            corr = phat_correlation(frames[i], frames[j])
            recs.append(pair_record(corr, frames.shape[1], fs, threshold_method, threshold_multiplier,
                                    max_expected_delay))
    out = {}
    for key in ("k_sel", "branch", "k_argmax"):
        out[key] = np.array([r[key] for r in recs], dtype=np.int32)
    for key in ("cmax", "cmin", "snr"):
        out[key] = np.array([r[key] for r in recs], dtype=np.float64)
    return out


# --------------------------------------------------------------------------
# a4  image sources            utils.py:29-106
# --------------------------------------------------------------------------
def speed_of_sound(temperature: float, humidity: float, pressure: float = 101.325) -> float:
    if not (-50 <= temperature <= 50):                 # utils.py:20-22
        temperature = 20
    if not (0 <= humidity <= 100):                     # utils.py:23-25
        humidity = 50
    return 331 + 0.6 * temperature + 0.0124 * humidity + 0.0006 * (pressure - 101.325)


def reflect_point(point, plane) -> np.ndarray:
    a, b, c, d = plane
    den = a ** 2 + b ** 2 + c ** 2
    if den == 0:
        raise ValueError("invalid plane: a^2+b^2+c^2 == 0")        # utils.py:36-37
    x, y, z = point
    f = 2 * (a * x + b * y + c * z + d) / den
    return np.array([x - a * f, y - b * f, z - c * f])


def attenuation(dist: float, material: str, frequency: float, table: Dict[str, Any]) -> float:
    dist = max(dist, 0.1)                              # utils.py:54-55
    if material not in table:                          # utils.py:57-59 (silent fallback)
        material = "air"
    return (1 / dist) * np.exp(-table[material]["freq"] * frequency * dist) * np.exp(-table[material]["absorption"] * dist)


def image_sources(source, planes, max_order, frequency, table, mics, thr=0.01, decimals=6):
    """BFS over reflection orders with rounding de-dup and mean/min attenuation pruning (utils.py:79-106)."""
    mics = np.asarray(mics, dtype=np.float64)
    found: List[Dict[str, Any]] = []
    frontier = [np.asarray(source, dtype=np.float64)]
    seen = {tuple(np.round(np.asarray(source, dtype=np.float64), decimals=decimals))}
    for _ in range(max_order):
        nxt = []
        for src in frontier:
            for pl in planes:
                img = reflect_point(src, pl["plane"])
                key = tuple(np.round(img, decimals=decimals))
                if key in seen:
                    continue
                mat = pl.get("material", "air")
                if mat not in table:
                    raise ValueError(f"material '{mat}' undefined")                 # utils.py:93-94
                if "absorption" not in table[mat] or "freq" not in table[mat]:
                    raise ValueError(f"material '{mat}' incomplete")                # utils.py:95-96
                att = [attenuation(np.linalg.norm(img - mp), mat, frequency, table) for mp in mics]
                if np.mean(att) > thr and np.min(att) > thr / 2:                      # utils.py:99
                    seen.add(key)
                    found.append({"source": img, "material": mat})
                    nxt.append(img)
        frontier = nxt
        if not frontier:
            break
    return found


# --------------------------------------------------------------------------
# a6/a7  fractional delay, normalise, compress     signal_processing.py:66-94
# --------------------------------------------------------------------------
def fade_window(n: int) -> np.ndarray:
    """Linear fade-in/out of int(0.01 n) samples (signal_processing.py:75-78).  n < 100 breaks
    the reference (``[-0:]`` slice); restated literally so the same ValueError surfaces."""
    fl = int(0.01 * n)
    w = np.ones(n)
    w[:fl] *= np.linspace(0, 1, fl)
    w[-fl:] *= np.linspace(1, 0, fl)
    return w


def fractional_delay(signal: np.ndarray, delay: float, fs: float) -> np.ndarray:
    x = np.asarray(signal, dtype=np.float64)
    n = x.shape[0]
    spec = np.fft.fft(x, n=2 * n)
    f = np.fft.fftfreq(2 * n, d=1.0 / fs)
    # both factors are named arrays as in signal_processing.py:71-72: with a temporary on the right NumPy
    # multiplies in place in the other operand order, and its FMA complex product is not bitwise commutative
    phase = np.exp(-1j * 2 * np.pi * f * delay)
    product = spec * phase
    y = np.fft.ifft(product).real[:n]
    return y * fade_window(n)


def normalize_signal(x: np.ndarray) -> np.ndarray:
    m = np.max(np.abs(x))
    return x if m == 0 else x / m


def dynamic_range_compression(x: np.ndarray, threshold: float = 0.8, epsilon: float = 1e-8) -> np.ndarray:
    y = normalize_signal(x)
    y = np.sign(y) * np.log1p(np.abs(y) / threshold + epsilon)
    m = np.max(np.abs(y))
    return y / m if m > 0 else y


def generate_signal(kind: str, fs: float, duration: float, freq: float) -> np.ndarray:
    """Deterministic generators only (sine, chirp) - signal_processing.py:25-32."""
    t = np.linspace(0, duration, int(fs * duration), endpoint=False)
    if kind == "sine":
        return np.sin(2 * np.pi * freq * t)
    if kind == "chirp":
        from scipy.signal import chirp
        return chirp(t, f0=freq, f1=freq * 5, t1=duration, method="linear")
    raise ValueError("oracle generates only 'sine' and 'chirp'")


# --------------------------------------------------------------------------
# a5  multipath simulation           main.py:66-124
# --------------------------------------------------------------------------
def multipath_paths(source, mics, c, freq, planes, table, max_reflections, thr):
    """Per-mic path lists (delay seconds, gain) in the reference's summation order:
    direct path first with material 'air' (main.py:106-110), then images in discovery order."""
    mics = np.asarray(mics, dtype=np.float64)
    src = np.asarray(source, dtype=np.float64)
    imgs = image_sources(source, planes, max_reflections, freq, table, mics, thr)
    delays = np.zeros((mics.shape[0], 1 + len(imgs)))
    gains = np.zeros_like(delays)
    max_delay = 0.0
    for m, mp in enumerate(mics):
        d0 = np.linalg.norm(src - mp)
        delays[m, 0] = d0 / c
        gains[m, 0] = attenuation(d0, "air", freq, table)
        far = d0
        for p, im in enumerate(imgs):
            d = np.linalg.norm(im["source"] - mp)
            delays[m, 1 + p] = d / c
            gains[m, 1 + p] = attenuation(d, im["material"], freq, table)
            far = max(far, d)
        max_delay = max(max_delay, far / c)
    return delays, gains, max_delay, imgs


def simulate_from_base(base: np.ndarray, delays: np.ndarray, gains: np.ndarray, fs: float,
                       total_samples: int, trim_len: Optional[int]) -> np.ndarray:
    """Fused form of the per-path loop: one forward FFT of the padded base signal,
    spectrum times sum_p g_p exp(-j 2 pi f tau_p), one inverse FFT per mic."""
    padded = np.zeros(total_samples)
    padded[: base.shape[0]] = base
    spec = np.fft.fft(padded, n=2 * total_samples)
    f = np.fft.fftfreq(2 * total_samples, d=1.0 / fs)
    fade = fade_window(total_samples)
    out = []
    for m in range(delays.shape[0]):
        acc = np.zeros(2 * total_samples, dtype=np.complex128)
        for p in range(delays.shape[1]):
            acc += gains[m, p] * np.exp(-1j * 2 * np.pi * f * delays[m, p])
        y = np.fft.ifft(spec * acc).real[:total_samples] * fade
        if trim_len is not None:
            y = y[:trim_len]
        out.append(dynamic_range_compression(normalize_signal(y)))
    return np.array(out)


def simulate_literal(base: np.ndarray, delays: np.ndarray, gains: np.ndarray, fs: float,
                     total_samples: int, trim_len: Optional[int]) -> np.ndarray:
    """Path-by-path loop exactly as main.py:103-123 (slow; validates the fused form)."""
    padded = np.zeros(total_samples)
    padded[: base.shape[0]] = base
    out = []
    for m in range(delays.shape[0]):
        tot = np.zeros(total_samples)
        for p in range(delays.shape[1]):
            tot += fractional_delay(padded, delays[m, p], fs) * gains[m, p]
        if trim_len is not None:
            tot = tot[:trim_len]
        out.append(dynamic_range_compression(normalize_signal(tot)))
    return np.array(out)


def simulate_signals_with_multipath(source_pos, mic_positions, fs, c, duration=1.0, signal_type="sine", freq=1000,
                                    reflective_planes=None, material_properties=None, max_reflections=2,
                                    absorption_threshold=0.01, trim_to_duration=True, base_signal=None):
    base = generate_signal(signal_type, fs, duration, freq) if base_signal is None else np.asarray(base_signal)
    delays, gains, max_delay, _ = multipath_paths(source_pos, mic_positions, c, freq, reflective_planes,
                                                  material_properties, max_reflections, absorption_threshold)
    total = int((duration + max_delay) * fs)                                   # main.py:102
    trim = int(duration * fs) if trim_to_duration else None                    # main.py:119-120
    return list(simulate_literal(base, delays, gains, fs, total, trim))


# --------------------------------------------------------------------------
# a8  noise_reduction               signal_processing.py:109-138
# --------------------------------------------------------------------------
def lfilter_df2t(b: np.ndarray, a: np.ndarray, x: np.ndarray, zi: np.ndarray) -> np.ndarray:
    """Direct-form-II-transposed recurrence in scipy's operation order (no fused multiply-add)."""
    b = np.asarray(b, dtype=np.float64) / a[0]
    a = np.asarray(a, dtype=np.float64) / a[0]
    k = max(b.shape[0], a.shape[0])
    bb = np.zeros(k); bb[: b.shape[0]] = b
    aa = np.zeros(k); aa[: a.shape[0]] = a
    z = np.array(zi, dtype=np.float64).copy()
    y = np.empty_like(x)
    if k == 1:
        return bb[0] * x
    for n in range(x.shape[0]):
        xn = x[n]
        yn = z[0] + bb[0] * xn
        for i in range(k - 2):
            z[i] = z[i + 1] + xn * bb[i + 1] - yn * aa[i + 1]
        z[k - 2] = xn * bb[k - 1] - yn * aa[k - 1]
        y[n] = yn
    return y


def lfilter_zi(b: np.ndarray, a: np.ndarray) -> np.ndarray:
    """Steady-state DF2T state for a unit step (scipy.signal.lfilter_zi), host-side scalar algebra."""
    b = np.asarray(b, dtype=np.float64) / a[0]
    a = np.asarray(a, dtype=np.float64) / a[0]
    k = max(b.shape[0], a.shape[0])
    bb = np.zeros(k); bb[: b.shape[0]] = b
    aa = np.zeros(k); aa[: a.shape[0]] = a
    comp = np.zeros((k - 1, k - 1))
    comp[0, :] = -aa[1:]
    comp[1:, :-1] = np.eye(k - 2)
    return np.linalg.solve(np.eye(k - 1) - comp.T, bb[1:] - aa[1:] * bb[0])


def filtfilt(b: np.ndarray, a: np.ndarray, x: np.ndarray) -> np.ndarray:
    """scipy.signal.filtfilt defaults: odd extension by 3*max(len(a),len(b)), zi*x0 both ways."""
    x = np.asarray(x, dtype=np.float64)
    edge = 3 * max(len(a), len(b))
    if x.shape[0] <= edge:
        raise ValueError("The length of the input vector x must be greater than padlen, which is %d." % edge)
    ext = np.concatenate((2 * x[0] - x[edge:0:-1], x, 2 * x[-1] - x[-2:-(edge + 2):-1]))
    zi = lfilter_zi(b, a)
    y = lfilter_df2t(b, a, ext, zi * ext[0])
    y = lfilter_df2t(b, a, y[::-1].copy(), zi * y[-1])
    return y[::-1][edge:-edge].copy()


def wiener3(x: np.ndarray) -> np.ndarray:
    """scipy.signal.wiener(x) defaults: 3-tap local mean/variance, noise = mean(local variance)."""
    x = np.asarray(x, dtype=np.float64)
    p = np.pad(x, 1)
    mean = (p[:-2] + p[1:-1] + p[2:]) / 3
    var = (p[:-2] ** 2 + p[1:-1] ** 2 + p[2:] ** 2) / 3 - mean ** 2
    noise = np.mean(var)
    res = (x - mean) * (1 - noise / var) + mean
    return np.where(var < noise, mean, res)


def butter_bandpass(fs: float, lowcut: float = 300, highcut: float = 3400):
    from scipy.signal import butter
    nyq = 0.5 * fs
    return butter(5, [lowcut / nyq, highcut / nyq], btype="band")           # signal_processing.py:125-127


def noise_reduction(x, fs, method="butterworth", lowcut=300, highcut=3400, filter_order=101):
    if method == "butterworth":
        b, a = butter_bandpass(fs, lowcut, highcut)
        return filtfilt(b, a, x)
    if method == "fir":
        from scipy.signal import firwin
        nyq = 0.5 * fs
        taps = firwin(filter_order, [lowcut / nyq, highcut / nyq], pass_zero=False)
        return filtfilt(taps, np.array([1.0]), x)
    if method == "wiener":
        return wiener3(x)
    raise ValueError("Unknown filter method. Available methods: 'butterworth', 'fir', 'wiener'")


# --------------------------------------------------------------------------
# a9  synchronisation               utils.py:407-457
# --------------------------------------------------------------------------
def xcorr_full(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    """scipy.signal.correlate(a, b, 'full') for real input: c[k] = sum_t a[t + k - (nb-1)] b[t]."""
    na, nb = a.shape[0], b.shape[0]
    size = 1
    while size < na + nb - 1:
        size *= 2
    spec = np.fft.rfft(a, size) * np.conj(np.fft.rfft(b, size))
    full = np.fft.irfft(spec, size)
    return np.concatenate((full[size - (nb - 1):], full[:na]))


def sync_shifts(signals: Sequence[np.ndarray], fs: float, use_interpolation: bool = True):
    """Shifts (float) and reference index of utils.py:415-446, quirk Q7 kept."""
    from scipy.interpolate import CubicSpline
    energies = [np.sum(np.asarray(s) ** 2) for s in signals]
    ref_idx = int(np.argmax(energies))
    ref = np.asarray(signals[ref_idx], dtype=np.float64)
    ref_peak = np.max(np.abs(xcorr_full(ref, ref)))
    limit = int(fs * 0.05)
    shifts: List[float] = []
    for idx, sig in enumerate(signals):
        if idx == ref_idx:
            shifts.append(0)
            continue
        cc = xcorr_full(np.asarray(sig, dtype=np.float64), ref)
        pk = int(np.argmax(np.abs(cc)))
        refined = pk
        if not (abs(cc[pk]) < 0.3 * ref_peak) and use_interpolation and 1 < pk < cc.shape[0] - 2:
            xs = np.arange(pk - 2, pk + 3)
            fine = np.linspace(pk - 2, pk + 2, 100)
            refined = fine[np.argmax(np.abs(CubicSpline(xs, cc[pk - 2: pk + 3])(fine)))]
        shift = refined - (ref.shape[0] - 1)
        if abs(shift) > limit:
            shift = 0
        shifts.append(shift)
    return shifts, ref_idx


def synchronize_signals(signals: Sequence[np.ndarray], fs: float, use_interpolation: bool = True) -> List[np.ndarray]:
    shifts, _ = sync_shifts(signals, fs, use_interpolation)
    lo = min(shifts)
    padded = [np.concatenate((np.zeros(max(0, int(round(s - lo)))), np.asarray(x, dtype=np.float64)))
              for x, s in zip(signals, shifts)]
    length = max(p.shape[0] for p in padded)
    return [np.concatenate((p, np.zeros(length - p.shape[0]))) for p in padded]


# --------------------------------------------------------------------------
# a10  TDOA stage of localize_sound_source      main.py:188-228
# --------------------------------------------------------------------------
def tdoa_stage(signals: Sequence[np.ndarray], fs: float, filter_method: str = "butterworth",
               max_expected_delay: Optional[float] = None) -> Dict[str, Any]:
    synced = synchronize_signals(signals, fs)
    filt = np.array([noise_reduction(s, fs, method=filter_method) for s in synced])
    table = all_pairs(filt, fs, max_expected_delay=max_expected_delay)
    table["L"] = filt.shape[1]
    table["filtered"] = filt
    return table


# ---------------------------------------------------------------- calibration path
def generate_calibration_signal(fs, duration=1.0, signal_type="chirp", freq_start=500, freq_end=5000) -> np.ndarray:
    """calibration.py:10-21: linear chirp (or unit impulse), normalised and compressed."""
    from scipy.signal import chirp
    t = np.linspace(0, duration, int(fs * duration), endpoint=False)
    if signal_type == "chirp":
        sig = chirp(t, f0=freq_start, f1=freq_end, t1=duration, method="linear")
    elif signal_type == "impulse":
        sig = np.zeros_like(t)
        sig[0] = 1.0
    else:
        raise ValueError("Unsupported calibration signal type. Use 'chirp' or 'impulse'.")
    return dynamic_range_compression(normalize_signal(sig))


def simulate_calibration_recording(calib_signal, mic_positions, source_position, fs, c, attenuation_factor=1.0,
                                   noise_level=0.01, freq=None, material_properties=None, noise=None) -> List[np.ndarray]:
    """calibration.py:23-41: per mic one fractional delay of the calibration signal, air attenuation, additive
    Gaussian noise.  `noise[M][N]` replaces the reference's unseeded np.random.normal draws (same order: one
    draw of N samples per mic) so that tests are deterministic."""
    if freq is None:
        freq = 1000
    if material_properties is None:
        raise ValueError("pass the material table (the reference defaults to materials.material_properties)")
    out = []
    for m, mic in enumerate(mic_positions):
        dist = float(np.linalg.norm(np.array(source_position) - np.array(mic)))
        rec = fractional_delay(np.asarray(calib_signal, dtype=float), dist / c, fs) * (
            attenuation_factor * attenuation(dist, "air", freq, material_properties))
        rec = rec + (np.random.normal(0, noise_level, size=rec.shape) if noise is None else noise[m])
        out.append(rec)
    return out


def analyze_calibration(recorded_signals, calib_signal, fs) -> List[Dict[str, float]]:
    """calibration.py:43-52: full cross-correlation against the calibration signal, lag of max |corr|."""
    results = []
    n_ref = len(calib_signal)
    for rec in recorded_signals:
        corr = xcorr_full(np.asarray(rec, dtype=float), np.asarray(calib_signal, dtype=float))
        k = int(np.argmax(np.abs(corr)))
        results.append({"delay": (k - (n_ref - 1)) / fs, "amplitude": float(np.max(np.abs(corr)))})
    return results
