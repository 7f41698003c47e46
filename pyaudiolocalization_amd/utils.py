"""Drop-in for the reference's utils.py on the hot path (citations are file:line into the reference).

GPU (HIP engine): phat_correlation, get_time_delays_phat, compute_snr, compute_peak_to_peak_ratio,
synchronize_signals_improved (the M cross-correlations), the bootstrap's PHAT calls.
Host C++: generate_image_sources_iterative.  Host Python (3-unknown solve, scalars): the rest.
"""
from __future__ import annotations

import logging
import os
from typing import Any, Dict, List, Optional, Sequence, Tuple

import numpy as np

from . import _ffi
from .engine import default_engine, image_sources
from .materials import material_properties  # noqa: F401  (re-exported like the reference does)

log = logging.getLogger(__name__)


# ---------------------------------------------------------------- scalars and geometry (host)
def speed_of_sound(temperature: float, humidity: float, pressure: float = 101.325) -> float:
    """Linear c(T, H, P); out-of-range T / H fall back to 20 C / 50 % (utils.py:15-27)."""
    if not -50 <= temperature <= 50:
        log.warning("unusual temperature, using 20 C")
        temperature = 20
    if not 0 <= humidity <= 100:
        log.warning("unusual humidity, using 50 %")
        humidity = 50
    return 331 + 0.6 * temperature + 0.0124 * humidity + 0.0006 * (pressure - 101.325)


def reflect_point_across_plane(point: Sequence[float], plane: Sequence[float]) -> np.ndarray:
    """Mirror image of a point in the plane a x + b y + c z + d = 0 (utils.py:29-42)."""
    a, b, c, d = plane
    norm2 = a ** 2 + b ** 2 + c ** 2
    if norm2 == 0:
        raise ValueError("invalid plane: a^2 + b^2 + c^2 is 0")
    x, y, z = point
    k = 2 * (a * x + b * y + c * z + d) / norm2
    return np.array([x - a * k, y - b * k, z - c * k])


def distance(point1: Sequence[float], point2: Sequence[float]) -> float:
    return np.linalg.norm(np.array(point1) - np.array(point2))               # utils.py:44-48


def calculate_attenuation(distance_val: float, material: str, frequency: float, material_properties: Dict[str, Any]) -> float:
    """1/max(d, 0.1) * exp(-freq_coeff f d) * exp(-absorption d); unknown material -> 'air' (utils.py:50-65)."""
    d = max(distance_val, 0.1)
    if material not in material_properties:
        log.warning("material '%s' undefined, using 'air'", material)
        material = "air"
    row = material_properties[material]
    return (1 / d) * np.exp(-row["freq"] * frequency * d) * np.exp(-row["absorption"] * d)


def generate_image_sources_iterative(source, planes, max_order, frequency, material_properties, mic_positions,
                                     absorption_threshold: float = 0.01, round_decimals: int = 6) -> List[Dict[str, Any]]:
    """Breadth-first image-source search with rounding de-dup and attenuation pruning
    (utils.py:67-106) in the engine's host C++ (pal_image_sources)."""
    names = list(material_properties.keys())
    ids = []
    for pl in planes:
        mat = pl.get("material", "air")
        complete = mat in material_properties and "absorption" in material_properties[mat] and "freq" in material_properties[mat]
        ids.append(names.index(mat) if complete else -1)
    absorption = [material_properties[n].get("absorption", np.nan) for n in names]
    freq_coeff = [material_properties[n].get("freq", np.nan) for n in names]
    plane_rows = np.array([pl["plane"] for pl in planes], dtype=np.float64).reshape(-1, 4)
    try:
        img, mat = image_sources(source, plane_rows, ids, absorption, freq_coeff, max_order, frequency,
                                 np.asarray(mic_positions, dtype=np.float64), absorption_threshold, round_decimals)
    except KeyError as exc:                                                   # utils.py:93-96
        name = planes[int(exc.args[0])].get("material", "air")
        if name not in material_properties:
            raise ValueError(f"material '{name}' is not defined; add it to the material table") from None
        raise ValueError(f"absorption or frequency property missing for material '{name}'") from None
    return [{"source": img[k].copy(), "material": names[mat[k]]} for k in range(img.shape[0])]


# ---------------------------------------------------------------- PHAT correlation and TDOA (GPU)
def phat_correlation(sig1: np.ndarray, sig2: np.ndarray) -> np.ndarray:
    """Unshifted PHAT sequence on the exact n = n1 + n2 - 1 grid (utils.py:108-119)."""
    return default_engine().phat_correlation(sig1, sig2)


_WARNINGS = (
    (_ffi.BR_ALT_THRESHOLD, "no peaks with the primary threshold, trying mean(|corr|)"),
    (_ffi.BR_ARGMAX_NO_PEAKS, "no peaks with the alternative threshold either, using the correlation maximum"),
    (_ffi.BR_WINDOW_RETRY, "no peaks inside the expected delay range, trying mean(|corr|)"),
    (_ffi.BR_ARGMAX_WINDOW, "no valid peaks after the alternative filtering, using the correlation maximum"),
)


def get_time_delays_phat(sig1: np.ndarray, sig2: np.ndarray, fs: float, num_peaks: int = 1,
                         threshold_method: str = "median", threshold_multiplier: float = 1.0,
                         max_expected_delay: Optional[float] = None) -> Tuple[List[float], np.ndarray, np.ndarray]:
    """(time delays of the selected peaks, corr, lags in seconds) exactly like utils.py:121-181,
    including the un-shifted lag mapping and the whole fallback chain (Q4)."""
    n1, n2 = len(sig1), len(sig2)
    ks, rec, corr = default_engine().get_time_delays_phat(sig1, sig2, fs, num_peaks, threshold_method,
                                                          threshold_multiplier, max_expected_delay, want_corr=True)
    for bit, text in _WARNINGS:
        if int(rec["branch"]) & bit:
            log.warning(text)
    time_lags = np.arange(-(n2 - 1), n1) / fs                                 # correlation_lags(...)/fs (utils.py:141-142)
    return list(time_lags[ks]), corr, time_lags


def compute_peak_to_peak_ratio(corr: np.ndarray) -> float:
    rec = default_engine().corr_metrics(corr)                                 # utils.py:228-236
    return np.inf if rec["cmin"] == 0 else rec["cmax"] / abs(rec["cmin"])


def compute_snr(corr: np.ndarray) -> float:
    return float(default_engine().corr_metrics(corr)["snr"])                  # utils.py:238-250


# ---------------------------------------------------------------- significance (next row N1)
def bootstrap_significance(sig1: np.ndarray, sig2: np.ndarray, fs: float, num_bootstrap: int = 1000, alpha: float = 0.05,
                           bootstrap_mode: str = "permutation", block_size: int = 50, batch: int = 250) -> float:
    """(1 - alpha) percentile of max(PHAT(sig1, shuffled sig2)) (utils.py:183-216).  The shuffles come from
    the global NumPy RNG like the reference (so the result is statistically, not bitwise, comparable); the
    PHAT correlations of a batch of shuffles run as ONE one-vs-many engine call."""
    if bootstrap_mode not in ("permutation", "block", "circular"):
        raise ValueError("unknown bootstrap_mode; use 'permutation', 'block' or 'circular'")
    a, b = np.asarray(sig1, dtype=np.float64), np.asarray(sig2, dtype=np.float64)
    if a.shape != b.shape:            # unequal lengths: fall back to one engine call per shuffle
        batch = 1
    eng = default_engine()
    peaks: List[float] = []
    while len(peaks) < num_bootstrap:
        count = min(batch, num_bootstrap - len(peaks))
        rows = [a]
        for _ in range(count):
            if bootstrap_mode == "permutation":
                rows.append(np.random.permutation(b))
            elif bootstrap_mode == "block":
                blocks = [b[i:i + block_size] for i in range(0, len(b), block_size)]
                np.random.shuffle(blocks)
                rows.append(np.concatenate(blocks)[: len(b)])
            else:
                rows.append(np.roll(b, np.random.randint(0, len(b))))
        if a.shape == b.shape:
            pairs = np.stack([np.zeros(count, dtype=np.int32), np.arange(1, count + 1, dtype=np.int32)], axis=1)
            peaks.extend(eng.gcc_phat_pairs(np.array(rows), pairs, fs)["cmax"].tolist())
        else:
            peaks.append(float(eng.get_time_delays_phat(a, rows[1], fs, want_corr=False)[1]["cmax"]))
    return np.percentile(peaks, 100 * (1 - alpha))


def perform_significance_test_bootstrap(sig1, sig2, fs, alpha: float = 0.05) -> Tuple[float, bool]:
    _, rec, _ = default_engine().get_time_delays_phat(sig1, sig2, fs, want_corr=False)
    peak = rec["cmax"]
    return peak, peak > bootstrap_significance(sig1, sig2, fs, alpha=alpha)    # utils.py:218-226


def perform_significance_test(corr, sig1, sig2, fs, alpha: float = 0.05, snr_threshold: float = 2.0) -> Tuple[float, bool]:
    snr = compute_snr(corr)
    _, ok = perform_significance_test_bootstrap(sig1, sig2, fs, alpha=alpha)
    return snr, ok and snr > snr_threshold                                      # utils.py:252-259


def compute_cross_correlation_metrics(corr, sig1, sig2, fs, alpha: float = 0.05) -> Dict[str, Any]:
    ratio = compute_peak_to_peak_ratio(corr)
    snr, significant = perform_significance_test(corr, sig1, sig2, fs, alpha=alpha)
    return {"peak_to_peak_ratio": ratio, "snr": snr, "significant": significant}   # utils.py:261-271


# ---------------------------------------------------------------- TDOA -> position (host, 3 unknowns)
SILHOUETTE_EXACT_MAX = 4096      # points up to which determine_optimal_number_of_clusters scores every pairwise distance

def determine_optimal_number_of_clusters(data, max_clusters: int = 5, method: str = "kmeans", eps: float = 0.001,
                                         min_samples: int = 2) -> int:
    """Silhouette-best k for KMeans, or DBSCAN's cluster count when its silhouette is positive (utils.py:273-302)."""
    from sklearn.cluster import DBSCAN, KMeans
    from sklearn.metrics import silhouette_score
    pts = np.array(data)
    if len(pts) < 2:
        return 1
    if method == "kmeans":
        best, best_k = -1, 1
        # The silhouette needs all pairwise distances: O(P^2) time and memory (8.5 GB and minutes at the 32 640 pairs of a
        # 256-microphone array).  Up to SILHOUETTE_EXACT_MAX points it is the reference's exact score; above, the score of a
        # fixed random subset of that many points (sklearn's own sample_size option, seeded) - the choice of k it is used
        # for is a property of the point cloud's shape, which the subset keeps.
        sample = None if len(pts) <= SILHOUETTE_EXACT_MAX else SILHOUETTE_EXACT_MAX
        for k in range(2, min(max_clusters, len(pts)) + 1):
            score = silhouette_score(pts, KMeans(n_clusters=k, random_state=0).fit(pts).labels_, sample_size=sample,
                                     random_state=0)
            if score > best:
                best, best_k = score, k
        return best_k
    if method == "dbscan":
        labels = DBSCAN(eps=eps, min_samples=min_samples).fit(pts).labels_
        core = labels != -1
        if core.sum() < 2:
            return 1
        return len(set(labels[core])) if silhouette_score(pts[core], labels[core]) > 0 else 1
    raise ValueError("unknown clustering method; available: 'kmeans', 'dbscan'")


def heuristic_initialization_adaptive(mic_positions, mic_pairs, tdoas, c, clustering_method: str = "kmeans",
                                      eps: float = 0.001, min_samples: int = 2) -> List[List[float]]:
    """Per-pair points on the mic axis offset by c|td|/2 from the midpoint, clustered into start
    positions; the array centroid is always among the guesses (utils.py:304-362)."""
    from sklearn.cluster import DBSCAN, KMeans
    mics = np.array(mic_positions)
    centroid = np.mean(mics, axis=0)
    if np.size(tdoas) == 0:
        return [centroid.tolist()]
    if len(mic_pairs) > SILHOUETTE_EXACT_MAX:          # same points, all pairs at once (the loop below costs seconds here)
        pr = np.asarray(mic_pairs, dtype=np.int64).reshape(-1, 2)
        td_all = np.asarray(tdoas, dtype=np.float64)[: pr.shape[0]]
        a, b = mics[pr[:, 0]], mics[pr[:, 1]]
        axis = b - a
        length = np.sqrt(np.sum(axis * axis, axis=1))
        keep = length != 0
        shift = ((c * np.abs(td_all[keep])) / 2)[:, None] * (axis[keep] / length[keep, None])
        mid = (a[keep] + b[keep]) / 2
        points = np.where((td_all[keep] > 0)[:, None], mid - shift, mid + shift).tolist()
    else:
        points = []
        for (i, j), td in zip(mic_pairs, np.array(tdoas)):
            a, b = np.array(mic_positions[i]), np.array(mic_positions[j])
            axis = b - a
            length = np.linalg.norm(axis)
            if length == 0:
                continue
            shift = (c * abs(td)) / 2 * (axis / length)
            points.append(((a + b) / 2 - shift if td > 0 else (a + b) / 2 + shift).tolist())
    if not points:
        return [centroid.tolist()]
    if clustering_method == "kmeans":
        k = determine_optimal_number_of_clusters(points, method=clustering_method, eps=eps, min_samples=min_samples)
        guesses = KMeans(n_clusters=k, random_state=0).fit(points).cluster_centers_.tolist()
    elif clustering_method == "dbscan":
        labels = DBSCAN(eps=eps, min_samples=min_samples).fit(points).labels_
        guesses = [np.mean([p for p, l in zip(points, labels) if l == lab], axis=0).tolist()
                   for lab in set(labels) - {-1}]
        if not guesses:
            guesses = [centroid.tolist()]
    else:
        guesses = [centroid.tolist()]
    if not any(np.allclose(centroid, g, atol=1e-6) for g in guesses):
        guesses.append(centroid.tolist())
    return guesses


def dynamic_bounds_extended(mic_positions, tdoas, c, buffer: float = 5.0) -> List[Tuple[float, float]]:
    """Array bounding box grown by buffer + max(1, 75th percentile of c|td|) (utils.py:364-382)."""
    mics = np.array(mic_positions)
    extra = max(np.percentile(c * np.abs(np.array(tdoas)), 75), 1.0) if np.size(tdoas) > 0 else 0.0
    lo = np.min(mics, axis=0) - (buffer + extra)
    hi = np.max(mics, axis=0) + (buffer + extra)
    dims = mics.shape[1] if mics.ndim > 1 else 1
    return [(lo[i], hi[i]) for i in range(dims)]


def equations(vars, mic_positions, mic_pairs, tdoas, c, weights: Optional[np.ndarray] = None) -> List[float]:
    """Weighted range-difference residuals (d_j - d_i) - c td (utils.py:384-405), evaluated for all pairs at
    once (the reference loops in Python: 12.5 ms per call at 2016 pairs)."""
    return list(residuals(vars, mic_positions, mic_pairs, tdoas, c, weights))


def residuals(vars, mic_positions, mic_pairs, tdoas, c, weights: Optional[np.ndarray] = None) -> np.ndarray:
    """`equations` as an array (what the solver iterates on)."""
    if weights is not None and len(weights) != len(mic_pairs):
        raise ValueError("length of weights must equal the number of mic pairs")
    mics = np.asarray(mic_positions, dtype=np.float64)
    pairs = np.asarray(mic_pairs, dtype=np.int64).reshape(-1, 2)
    ranges = np.sqrt(np.sum((np.asarray(vars, dtype=np.float64) - mics) ** 2, axis=1))
    res = (ranges[pairs[:, 1]] - ranges[pairs[:, 0]]) - c * np.asarray(tdoas, dtype=np.float64)[: pairs.shape[0]]
    if weights is not None:
        res = res * np.asarray(weights)
    return res


def equations_jacobian(vars, mic_positions, mic_pairs, tdoas, c, weights: Optional[np.ndarray] = None) -> np.ndarray:
    """Analytic Jacobian of `equations`: d/dx (|x - m_j| - |x - m_i|) w = ((x - m_j) / d_j - (x - m_i) / d_i) w, [P][3].
    The reference lets least_squares difference the residuals (three extra evaluations of its Python loop per step)."""
    mics = np.asarray(mic_positions, dtype=np.float64)
    pairs = np.asarray(mic_pairs, dtype=np.int64).reshape(-1, 2)
    diff = np.asarray(vars, dtype=np.float64) - mics
    ranges = np.sqrt(np.sum(diff ** 2, axis=1))
    unit = diff / np.where(ranges > 0, ranges, 1.0)[:, None]
    jac = unit[pairs[:, 1]] - unit[pairs[:, 0]]
    if weights is not None:
        jac = jac * np.asarray(weights)[:, None]
    return jac


def compute_weights(correlation_metrics, mic_pairs) -> np.ndarray:
    """SNR per pair (1.0 when missing), normalised to mean 1 (utils.py:484-497)."""
    w = np.array([1.0 if correlation_metrics.get(p) is None else correlation_metrics[p].get("snr", 1.0) for p in mic_pairs])
    return w / np.mean(w) if np.mean(w) != 0 else w


# ---------------------------------------------------------------- synchronisation (GPU correlations)
def sync_shifts_from_measurements(kpk, win, pkabs, ref_peak, ref_idx, lens, lmax, fs, use_interpolation=True) -> List[float]:
    """Host half of utils.py:420-446: the engine's per-row measurements (argmax |corr| index into the padded
    2 lmax - 1 sequence, the five samples around it, |peak|, the reference's autocorrelation peak) -> float shifts,
    with the reference's quirks kept (low peak: shift NOT zeroed; |shift| > 50 ms: zeroed)."""
    from scipy.interpolate import CubicSpline
    nref = lens[ref_idx]
    limit = int(fs * 0.05)
    shifts: List[float] = []
    for idx in range(len(lens)):
        if idx == ref_idx:
            shifts.append(0)
            continue
        pk = int(kpk[idx]) - (lmax - nref)              # index into the reference's own len(sig) + len(ref) - 1 sequence
        refined = pk
        if pkabs[idx] < 0.3 * ref_peak:
            log.warning("low correlation peak for signal %d during synchronisation", idx)   # shift is NOT zeroed
        elif use_interpolation and 1 < pk < lens[idx] + nref - 3:
            fine = np.linspace(pk - 2, pk + 2, 100)
            refined = fine[np.argmax(np.abs(CubicSpline(np.arange(pk - 2, pk + 3), win[idx])(fine)))]
        shift = refined - (nref - 1)
        if abs(shift) > limit:
            log.warning("shift of %s samples for signal %d is implausible, using 0", shift, idx)
            shift = 0
        shifts.append(shift)
    return shifts


def sync_shifts_batch(kpk, win, pkabs, ref_peak, ref_idx, length: int, fs: float, use_interpolation: bool = True) -> np.ndarray:
    """sync_shifts_from_measurements for B frames of M rows of one length at once (the streaming chain, stream.py):
    kpk / pkabs [B][M], win [B][M][5], ref_peak / ref_idx [B] -> shifts [B][M] (float64).

    The 5-point spline of utils.py:428-437 is the same call - scipy's CubicSpline over np.arange(pk - 2, pk + 3), sampled
    on np.linspace(pk - 2, pk + 2, 100) - made once per DISTINCT peak index with the windows of all its rows as the columns
    of a 2-D ``y``: a spline per column, the same arithmetic per column as a call per row (ties of symmetric windows
    included; tests/test_host_tail.py compares the two bit for bit).  A call per row costs 50 us: 0.43 s for the 8192 rows
    of 128 frames, two thirds of the whole device-resident chain."""
    from scipy.interpolate import CubicSpline
    kpk = np.asarray(kpk, dtype=np.int64)
    win = np.asarray(win, dtype=np.float64)
    pkabs = np.asarray(pkabs, dtype=np.float64)
    ref_peak = np.asarray(ref_peak, dtype=np.float64)
    ref_idx = np.asarray(ref_idx, dtype=np.int64)
    b, m = kpk.shape
    limit = int(fs * 0.05)
    refined = kpk.astype(np.float64)
    is_ref = np.arange(m)[None, :] == ref_idx[:, None]
    low = (pkabs < 0.3 * ref_peak[:, None]) & ~is_ref
    if low.any():
        log.warning("low correlation peak for %d signal(s) during synchronisation", int(low.sum()))   # shifts NOT zeroed
    if use_interpolation:
        fit = ~is_ref & ~low & (kpk > 1) & (kpk < 2 * length - 3)
        for pk in np.unique(kpk[fit]):
            rows = fit & (kpk == pk)
            fine = np.linspace(pk - 2, pk + 2, 100)
            values = CubicSpline(np.arange(pk - 2, pk + 3), win[rows].T)(fine)          # [100][rows]
            refined[rows] = fine[np.argmax(np.abs(values), axis=0)]
    shifts = refined - (length - 1)
    wild = np.abs(shifts) > limit
    wild &= ~is_ref
    if wild.any():
        log.warning("implausible shift for %d signal(s), using 0", int(wild.sum()))
        shifts[wild] = 0.0
    shifts[is_ref] = 0.0
    return shifts


def synchronize_signals_improved(signals: List[np.ndarray], fs: float, use_interpolation: bool = True) -> List[np.ndarray]:
    """Align every signal to the highest-energy one (utils.py:407-457).  The M full cross-correlations
    and their argmax run on the engine; the 5-point spline refinement and zero padding are host work.

    Signals of different lengths are accepted like in the reference (recordings read by read_audio_files seldom
    agree to the sample): for the engine call only, every row is zero-padded at its end to the longest length
    Lmax.  Trailing zeros leave correlate(sig, ref, 'full') unchanged lag by lag - index k of the reference's
    sequence (length len(sig) + len(ref) - 1) sits at k + (Lmax - len(ref)) of the padded one - so the shift is
    k' - (Lmax - 1), and the pads are applied to the ORIGINAL rows as utils.py:448-456 does."""
    sigs = [np.asarray(s, dtype=np.float64) for s in signals]
    lens = [len(s) for s in sigs]
    lmax = max(lens)
    if len(set(lens)) == 1:
        rows = np.asarray(sigs, dtype=np.float64)
    else:
        rows = np.zeros((len(sigs), lmax))
        for r, s in zip(rows, sigs):
            r[: len(s)] = s
    energies = [np.sum(s ** 2) for s in sigs]
    ref_idx = int(np.argmax(energies))
    kpk, win, pkabs, ref_peak = default_engine().xcorr_vs_ref(rows, ref_idx)
    shifts = sync_shifts_from_measurements(kpk, win, pkabs, ref_peak, ref_idx, lens, lmax, fs, use_interpolation)
    lowest = min(shifts)
    padded = [np.pad(s, (max(0, int(round(sh - lowest))), 0)) for s, sh in zip(sigs, shifts)]
    length = max(len(p) for p in padded)
    return [np.pad(p, (0, length - len(p))) for p in padded]


def _read_wav_pcm(path: str) -> Tuple[np.ndarray, int]:
    """PCM WAV through the standard library, scaled like ``soundfile.read`` scales integer PCM (x / 2^(bits-1); 8-bit is
    unsigned): the fallback of read_audio_files when the optional soundfile package is absent.  [frames] or [frames][ch]."""
    import wave  # noqa: PLC0415
    with wave.open(path, "rb") as w:
        channels, width, fs, frames = w.getnchannels(), w.getsampwidth(), w.getframerate(), w.getnframes()
        raw = w.readframes(frames)
    if width == 1:
        x = (np.frombuffer(raw, dtype=np.uint8).astype(np.float64) - 128.0) / 128.0
    elif width == 2:
        x = np.frombuffer(raw, dtype="<i2").astype(np.float64) / 32768.0
    elif width == 3:
        b = np.frombuffer(raw, dtype=np.uint8).reshape(-1, 3).astype(np.int32)
        v = b[:, 0] | (b[:, 1] << 8) | (b[:, 2] << 16)
        x = np.where(v >= 1 << 23, v - (1 << 24), v).astype(np.float64) / float(1 << 23)
    elif width == 4:
        x = np.frombuffer(raw, dtype="<i4").astype(np.float64) / float(1 << 31)
    else:
        raise ValueError(f"unsupported PCM sample width {width}")
    return (x.reshape(-1, channels) if channels > 1 else x), fs


def read_audio_files(audio_files: List[str], expected_fs: float) -> List[np.ndarray]:
    """Real-audio ingest (utils.py:459-482; SURVEY 8f N4, outside the hot path): mono mix, resample to `expected_fs`,
    normalise, compress.  Decoding uses the optional soundfile package when it is installed and the standard library's
    PCM WAV reader otherwise (the build image has neither soundfile nor resampy)."""
    from .signal_processing import dynamic_range_compression, normalize_signal, resample_audio
    out = []
    for path in audio_files:
        if not os.path.isfile(path):
            log.error("audio file not found: %s", path)
            raise FileNotFoundError(f"audio file not found: {path}")
        try:
            try:
                import soundfile  # noqa: PLC0415 - optional dependency
                data, fs = soundfile.read(path)
            except ImportError:
                data, fs = _read_wav_pcm(path)
            if data.ndim > 1:
                data = np.mean(data, axis=1)
            if fs != expected_fs:
                data = resample_audio(data, fs, expected_fs)
            out.append(dynamic_range_compression(normalize_signal(data)))
        except Exception as exc:
            log.error("error reading audio file '%s': %s", path, exc)
            raise RuntimeError(f"error reading audio file '{path}': {exc}")
    return out
