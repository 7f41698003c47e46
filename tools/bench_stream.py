#!/usr/bin/env python3
"""Streaming configuration C5 end to end on ONE GPU: frames/s of the device-resident stage chain simulate -> synchronise
-> prefilter -> all pairs (pyaudiolocalization_amd/stream.py) for a slice of the 1024-frame stream (64 microphones,
48 kHz x 0.25 s, multipath simulation on: 3 planes, order 3, low-loss materials).

    python tools/bench_stream.py [frames=128] [repeats=3]

Prints one JSON line: frames/s and pair-correlations/s of the whole chain, the per-kernel HIP-event times, and the stage
times of the second path on their own (multipath synthesis at C2b's size, the reference needs 3.29 s there; filtfilt)."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyaudiolocalization_amd import Engine  # noqa: E402
from pyaudiolocalization_amd.stream import tdoa_stream  # noqa: E402
from pyaudiolocalization_amd.synthetic import C5_FS, C5_SAMPLES, c5_stream_inputs  # noqa: E402
from pyaudiolocalization_amd.utils import speed_of_sound  # noqa: E402

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 128
repeats = int(sys.argv[2]) if len(sys.argv) > 2 else 3
eng = Engine(0)
c = speed_of_sound(20, 50)
t0 = time.perf_counter()
bases, delays, gains, totals = c5_stream_inputs(0, frames, c)
geometry_s = time.perf_counter() - t0
tdoa_stream(bases[:8], delays[:8], gains[:8], C5_FS, totals[:8], C5_SAMPLES, "butterworth", 0.05, engine=eng)     # plans, scratch
best = None
for _ in range(repeats):
    eng.profile_begin(every=1)
    t0 = time.perf_counter()
    tables, lengths = tdoa_stream(bases, delays, gains, C5_FS, totals, C5_SAMPLES, "butterworth", 0.05, engine=eng)
    el = time.perf_counter() - t0
    eng.profile_end()
    if best is None or el < best[0]:
        best = (el, {k: {"ms": round(v[0], 3), "launches": v[1]} for k, v in eng.profile_entries().items() if v[1] > 0})
el, kernels = best
# the second path's stages on their own
from pyaudiolocalization_amd.main import multipath_geometry  # noqa: E402
rng = np.random.default_rng(2)
mics8 = rng.uniform(-0.5, 0.5, (8, 3))
from pyaudiolocalization_amd.synthetic import C5_LOW_LOSS, C5_PLANES  # noqa: E402
d8, g8, longest = multipath_geometry([1.0, 2.0, 0.5], mics8, c, 500, C5_PLANES, C5_LOW_LOSS, 3, 0.01)
from scipy.signal import chirp  # noqa: E402
t = np.linspace(0, 1.0, 48000, endpoint=False)
base = chirp(t, f0=500, f1=2500, t1=1.0, method="linear")
total = int((1.0 + longest) * 48000)
eng.simulate_multipath(base[None], 48000.0, total, d8[None], g8[None], 48000)
t0 = time.perf_counter()
for _ in range(5):
    eng.simulate_multipath(base[None], 48000.0, total, d8[None], g8[None], 48000)
sim_c2b = (time.perf_counter() - t0) / 5
from pyaudiolocalization_amd.signal_processing import noise_reduction_rows  # noqa: E402
rows = rng.standard_normal((64, 24000))
noise_reduction_rows(rows, 48000.0)
t0 = time.perf_counter()
for _ in range(5):
    noise_reduction_rows(rows, 48000.0)
filt64 = (time.perf_counter() - t0) / 5
print(json.dumps({
    "workload": f"C5 stream slice: {frames} frames x 64 mics x {C5_SAMPLES} samples @ 48 kHz, 8 paths per microphone, butterworth, "
                "max_expected_delay 0.05; device-resident chain (stream.tdoa_stream)",
    "frames_per_s": round(frames / el, 2), "pair_correlations_per_s": round(frames * 2016 / el, 1), "elapsed_s": round(el, 4),
    "distinct_simulated_lengths": len(set(totals)), "distinct_synchronised_lengths": len(set(int(v) for v in lengths)),
    "plans_built_evicted": list(eng.plan_stats()),
    "host_geometry_s": round(geometry_s, 3), "kernels_ms": kernels,
    "simulate_c2b_s": round(sim_c2b, 5), "simulate_c2b_note": "8 mics x (1 direct + 7 images), 48 kHz x 1 s, host arrays in and out; the reference: 3.29 s (BASELINE.md)",
    "filtfilt_64x24000_s": round(filt64, 5), "filtfilt_note": "Butterworth-5 band-pass, 64 rows x 24000 samples, host arrays in and out"}))
eng.close()
