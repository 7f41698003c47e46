#!/usr/bin/env python3
"""Pair-correlations/s of the device-resident all-pairs call for EVERY frame length of a range (default: the 200 lengths
L = 44100 ... 44299 the synchronisation padding of the metric configuration can produce, utils.py:448-456), with the route
each length takes.  One engine, 4 frames x 64 microphones of Gaussian noise per length, inputs resident in HBM.

    python tools/length_sweep.py [first=44100] [count=200] [frames=4] [stride=1] > profiles/<tag>_length_sweep.csv"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyaudiolocalization_amd import Engine  # noqa: E402
from pyaudiolocalization_amd._ffi import RECORD  # noqa: E402
from pyaudiolocalization_amd.engine import make_params  # noqa: E402

first = int(sys.argv[1]) if len(sys.argv) > 1 else 44100
count = int(sys.argv[2]) if len(sys.argv) > 2 else 200
frames = int(sys.argv[3]) if len(sys.argv) > 3 else 4
stride = int(sys.argv[4]) if len(sys.argv) > 4 else 1          # every stride-th length
mics, fs = 64, 44100.0
pairs = mics * (mics - 1) // 2
eng = Engine(0)
prm = make_params(fs, 1, "median", 1.0, 0.05)
rng = np.random.default_rng(5)
longest = first + (count - 1) * stride
host = rng.standard_normal((frames, mics, longest))
d_frames = eng.alloc(host.nbytes)
d_table = eng.alloc(frames * pairs * RECORD.itemsize)
print("L,n,route,n1,n2,tile,conv_m1,conv_m2,pairs_per_s")
for length in range(first, first + count * stride, stride):
    eng.upload(d_frames, np.ascontiguousarray(host[:, :, :length]))
    info = eng.plan_info(length)
    eng.gcc_phat_all_pairs_dev(d_frames, frames, mics, length, prm, d_table)      # plan, scratch
    eng.synchronize()
    best = None
    for _ in range(2):
        t0 = time.perf_counter()
        eng.gcc_phat_all_pairs_dev(d_frames, frames, mics, length, prm, d_table)
        eng.synchronize()
        el = time.perf_counter() - t0
        best = el if best is None or el < best else best
    route = "prime-factor" if info["n1"] else "four-step"
    print(f"{length},{info['n']},{route},{info['n1']},{info['n2']},{info['tile_len']},{info['m1']},{info['m2']},{frames * pairs / best:.0f}", flush=True)
    eng.clear_plans()
eng.close()
