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
