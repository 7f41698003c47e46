import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["PAL_DEBUG_FALLBACK"] = "1"
import numpy as np
def eng(**env):
    for k, v in env.items(): os.environ[k] = v
    from pyaudiolocalization_amd import Engine
    e = Engine(0)
    for k in env: os.environ.pop(k)
    return e
rng = np.random.default_rng(5)
from pyaudiolocalization_amd.synthetic import metric_frames
metric_frames(2, 8, 44100)
fr = rng.standard_normal((2, 8, 44100))
ref = eng(PAL_FIN="0")
want = ref.gcc_phat_all_pairs(fr, 44100.0, 1, "median", 4.2, None)
want2 = ref.gcc_phat_all_pairs(fr, 44100.0, 1, "median", 4.2, None)
print("stored path reproducible:", want.tobytes() == want2.tobytes())
e = eng(PAL_FIN="1")
for rep in range(4):
    print("--- rep", rep, flush=True)
    sys.stderr.flush()
    t = e.gcc_phat_all_pairs(fr, 44100.0, 1, "median", 4.2, None)
    bad = np.flatnonzero(t.ravel() != want.ravel())
    for i in bad:
        print("  row", i, "\n    fin   ", t.ravel()[i], "\n    stored", want.ravel()[i])
e.close()
