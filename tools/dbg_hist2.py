import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
def eng(**env):
    for k, v in env.items(): os.environ[k] = v
    from pyaudiolocalization_amd import Engine
    e = Engine(0)
    for k in env: os.environ.pop(k)
    return e
rng = np.random.default_rng(5)
fr = rng.standard_normal((4, 16, 44100))
ref = eng(PAL_FIN="0")
bad_total = 0
e = eng(PAL_FIN="1")
for mult in (4.2, 1.0, 3.0):
    want = ref.gcc_phat_all_pairs(fr, 44100.0, 1, "median", mult, None)
    for rep in range(6):
        t = e.gcc_phat_all_pairs(fr, 44100.0, 1, "median", mult, None)
        bad = np.flatnonzero(t["cmax"].ravel() != want["cmax"].ravel())
        bad_total += bad.size
        if bad.size: print("mult", mult, "rep", rep, "cmax differs in rows", bad.tolist()[:8], [float((t['cmax'].ravel()[i] - want['cmax'].ravel()[i]) / want['cmax'].ravel()[i]) for i in bad[:4]])
print("TOTAL cmax mismatches", bad_total, "lib", os.environ.get("PAL_LIB_PATH", "in-tree"))
