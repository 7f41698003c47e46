import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
os.environ["PAL_DEBUG_FALLBACK"] = "1"
import numpy as np
from pyaudiolocalization_amd import Engine
from pyaudiolocalization_amd.synthetic import metric_frames
e = Engine(0)
for (L, fs) in ((24000, 48000.0), (48000, 48000.0), (44100, 44100.0)):
    m = 64 if L != 48000 else 8
    fr = metric_frames(2, m, L)
    t = e.gcc_phat_all_pairs(fr, fs, 1, "median", 1.0, 0.05)
    off = t["k_sel"].ravel().astype(int) - (L - 1)
    w = int(0.05 * fs); d = int(fs * 1e-3)
    print(L, "rows", off.size, "near edge", int(np.sum(np.abs(off) > w - (d - 1))), "max |off|", np.abs(off).max(), "plan", e.plan_info(L))
e.close()
