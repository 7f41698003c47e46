#!/usr/bin/env python3
"""Phase times of k_pfa_cols_fin on the metric geometry: PAL_DEBUG_STAMPS=1 PAL_OVERLAP=0 python tools/stamps_fin.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if os.environ.get("STAMPS", "1") == "1":
    os.environ["PAL_DEBUG_STAMPS"] = "1"
os.environ["PAL_OVERLAP"] = "0"
import numpy as np
from pyaudiolocalization_amd import Engine, make_params, RECORD
from pyaudiolocalization_amd.synthetic import metric_frames
m = 32
L = int(os.environ.get("STAMPS_L", "44100"))
fr = metric_frames(1, m, L)
e = Engine(0)
d = e.alloc(fr.nbytes); e.upload(d, fr)
t = e.alloc(m * (m - 1) // 2 * RECORD.itemsize)
prm = make_params(44100.0, 1, "median", 1.0, 0.05)
for _ in range(3):
    e.gcc_phat_all_pairs_dev(d, 1, m, L, prm, t); e.synchronize()
    print("----", file=sys.stderr)
e.close()
