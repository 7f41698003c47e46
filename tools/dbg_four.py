import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pyaudiolocalization_amd import Engine
e = Engine(0)
rng = np.random.default_rng(1)
L = 44101
for m in (6, 12, 16, 23, 24, 32, 48, 64):
    fr = rng.standard_normal((1, m, L))
    try:
        t = e.gcc_phat_all_pairs(fr, 44100.0, 1, "median", 1.0, 0.05)
        print(m, "mics", m * (m - 1) // 2, "pairs ok", flush=True)
    except Exception as ex:
        print(m, "mics", m * (m - 1) // 2, "pairs FAILED:", str(ex)[:100], flush=True)
e.close()
