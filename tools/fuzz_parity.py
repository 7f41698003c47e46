"""Randomised parity sweep on a GPU box: all-pairs tables of the engine against the oracle for random frame
lengths, mic counts, sampling rates, windows, threshold methods and multipliers.  Diagnostics, not a test:
prints every mismatch.  Reported apart: index differences at rounding-level ties of the oracle's own sequence, and rows
whose median |corr| (the threshold) is itself FFT rounding noise (sparse / tonal inputs: a few spikes over 1e-17)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import pal_oracle as O
from pyaudiolocalization_amd import Engine

eng = Engine(0)
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
cases = int(sys.argv[2]) if len(sys.argv) > 2 else 120
bad = ties = rows = 0
t0 = time.time()
for case in range(cases):
    L = int(rng.choice([rng.integers(2, 64), rng.integers(64, 700), rng.integers(700, 3000), rng.integers(3000, 9000),
                        rng.integers(9000, 25000), rng.integers(25000, 49000),   # register row tiles, register-row four-step geometries
                        rng.choice([496, 11962, 15525, 22651, 44100])]))   # (Rader rows at 991 = 2 x 496 - 1, fused column pass where the plan has N1 <= 89)
    mics = int(rng.integers(2, 7))
    fs = float(rng.choice([8000.0, 16000.0, 44100.0, 48000.0]))
    med = None if rng.random() < 0.4 else float(rng.choice([0.0005, 0.002, 0.01, 0.05]))
    method = "median" if rng.random() < 0.7 else "adaptive"
    mult = float(rng.choice([1.0, 0.5, 2.0, 6.0]))
    kind = rng.integers(0, 4)
    base = rng.standard_normal(L + 64)
    if kind == 0:                                   # independent noise
        frames = rng.standard_normal((mics, L))
    elif kind == 1:                                 # delayed copies + noise
        frames = np.stack([base[d:d + L] for d in rng.integers(0, 64, mics)]) + 0.3 * rng.standard_normal((mics, L))
    elif kind == 2:                                 # tones (ill-conditioned PHAT)
        t = np.arange(L) / fs
        frames = np.stack([np.sin(2 * np.pi * 440 * (t + d / fs)) for d in rng.integers(0, 20, mics)]) + 1e-3 * rng.standard_normal((mics, L))
    else:                                           # sparse / partly silent
        frames = rng.standard_normal((mics, L)) * (rng.random((mics, L)) < 0.05)
    if rng.random() < 0.15:                         # a silent microphone: exactly zero rows in the reference
        frames[int(rng.integers(0, mics))] = 0.0
    if int(fs * 0.001) < 1:
        continue
    try:
        got = eng.gcc_phat_all_pairs(frames, fs, threshold_method=method, threshold_multiplier=mult, max_expected_delay=med)
    except Exception as exc:
        print(f"case {case}: L={L} mics={mics} fs={fs} med={med} {method} x{mult}: engine raised {exc}")
        bad += 1
        continue
    want = O.all_pairs(frames, fs, threshold_method=method, threshold_multiplier=mult, max_expected_delay=med)
    for p in range(len(want["k_sel"])):
        rows += 1
        if got["k_sel"][p] == want["k_sel"][p] and got["branch"][p] == want["branch"][p] and got["k_argmax"][p] == want["k_argmax"][p]:
            continue
        i, j = np.triu_indices(mics, k=1)
        c = O.phat_correlation(frames[i[p]], frames[j[p]])
        gap = max(abs(c[got["k_sel"][p]] - c[want["k_sel"][p]]), abs(c[got["k_argmax"][p]] - c[want["k_argmax"][p]]))
        noise_floor = np.median(np.abs(c)) <= 1e-12 * np.max(np.abs(c))    # the threshold itself is FFT rounding noise
        if kind >= 2 and (gap <= 1e-12 * max(1.0, np.max(np.abs(c))) or noise_floor):
            ties += 1
        else:
            bad += 1
            print(f"case {case} pair {p}: L={L} mics={mics} fs={fs} med={med} {method} x{mult} kind={kind}: "
                  f"k_sel {got['k_sel'][p]} vs {want['k_sel'][p]}, branch {got['branch'][p]} vs {want['branch'][p]}, "
                  f"argmax {got['k_argmax'][p]} vs {want['k_argmax'][p]}, gap {gap:.2e}")
print(f"{rows} rows in {cases} cases, {bad} mismatches, {ties} rounding-level ties (ill-conditioned inputs), {time.time() - t0:.0f} s")
