import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from pyaudiolocalization_amd import Engine
from pyaudiolocalization_amd.stream import tdoa_stream
from pyaudiolocalization_amd.synthetic import C5_FS, C5_SAMPLES, c5_stream_inputs
from pyaudiolocalization_amd.utils import speed_of_sound
eng = Engine(0)
c = speed_of_sound(20, 50)
bases, delays, gains, totals = c5_stream_inputs(0, 128, c)
tdoa_stream(bases[:8], delays[:8], gains[:8], C5_FS, totals[:8], C5_SAMPLES, "butterworth", 0.05, engine=eng)
for rep in range(2):
    tm = {}
    t0 = time.perf_counter()
    tables, lengths = tdoa_stream(bases, delays, gains, C5_FS, totals, C5_SAMPLES, "butterworth", 0.05, engine=eng, timings=tm)
    print("elapsed", time.perf_counter() - t0, {k: round(v, 4) for k, v in tm.items()}, file=sys.stderr)
t0 = time.perf_counter()
tables, lengths = tdoa_stream(bases, delays, gains, C5_FS, totals, C5_SAMPLES, "butterworth", 0.05, engine=eng)
print("elapsed untimed", time.perf_counter() - t0, sorted(set(int(v) for v in lengths)), file=sys.stderr)
import collections
print(collections.Counter(int(v) for v in lengths), collections.Counter(int(t) for t in totals).most_common(5), file=sys.stderr)
for L in sorted(set(int(v) for v in lengths)):
    print(L, eng.plan_info(L), file=sys.stderr)
eng.close()
