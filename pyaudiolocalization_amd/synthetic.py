"""Synthetic mic-array frames of the metric run (last row): independent white
noise per mic plus one common component with an integer per-mic delay.  NumPy only; every frame
depends only on its global index so that each rank can build exactly its shard."""
from __future__ import annotations

import numpy as np


def metric_frames(batch: int, mics: int = 64, n: int = 44100, first: int = 0) -> np.ndarray:
    delays = np.random.default_rng(8).integers(-60, 60, size=max(64, mics))[:mics]   # (the first 64 do not depend on the count)
    out = np.empty((batch, mics, n))
    for b in range(batch):
        g = np.random.default_rng([7, first + b])
        common = g.standard_normal(n + 200)
        out[b] = g.standard_normal((mics, n))
        for m in range(mics):
            out[b, m] += common[100 + delays[m]: 100 + delays[m] + n]
    return out


# ---------------------------------------------------------------- streaming configuration C5
C5_FS, C5_SAMPLES = 48000.0, 12000
C5_PLANES = [{"plane": [1, 0, 0, -5], "material": "wood"}, {"plane": [0, 1, 0, -5], "material": "metal"},
             {"plane": [0, 0, 1, -5], "material": "wood"}]                      # the reference's three planes (main.py:41-43)
C5_LOW_LOSS = {"air": {"absorption": 0.01, "freq": 0.0}, "wood": {"absorption": 0.05, "freq": 1e-6},
               "metal": {"absorption": 0.1, "freq": 1e-6}}                       # a table under which images survive


def grid_array_64() -> np.ndarray:
    """8 x 8 planar array, pitch 0.1 m, z = 0, centred (configurations C3 / C5)."""
    ax = (np.arange(8) - 3.5) * 0.1
    gx, gy = np.meshgrid(ax, ax, indexing="ij")
    return np.stack([gx.ravel(), gy.ravel(), np.zeros(64)], axis=1)


def c5_stream_inputs(first: int, count: int, c: float):
    """Frames first .. first+count-1 of the 1024-frame stream: white-noise base signals (seed 1000 + f), a source that
    random-walks from [1, 2, 0.5] (sigma 2 cm per frame, seed 5), and the per-frame path tables of the multipath
    simulator (3 planes, order 3, low-loss materials): (bases, delays[M][K], gains[M][K], totals) - what
    stream.tdoa_stream takes.  Geometry is host work (main.multipath_geometry: C++ image sources)."""
    from .main import multipath_geometry
    steps = np.random.default_rng(5).normal(0.0, 0.02, (1024, 3))
    mics = grid_array_64()
    bases, delays, gains, totals = [], [], [], []
    for f in range(first, first + count):
        src = np.array([1.0, 2.0, 0.5]) + steps[: f + 1].sum(axis=0)
        d, g, longest = multipath_geometry(src, mics, c, 1000, C5_PLANES, C5_LOW_LOSS, 3, 0.01)
        bases.append(np.random.default_rng(1000 + f).standard_normal(C5_SAMPLES))
        delays.append(d)
        gains.append(g)
        totals.append(int((C5_SAMPLES / C5_FS + longest) * C5_FS))
    return bases, delays, gains, totals
