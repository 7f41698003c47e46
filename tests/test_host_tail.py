"""The host-side TDOA -> position solve (main.py:233-298 restated in pyaudiolocalization_amd.main.solve_position)
against the reference's fixtures: the reference's own selected indices in, the reference's position out.  CPU only."""
import numpy as np

from oracle import cases
from pyaudiolocalization_amd import pair_list
from pyaudiolocalization_amd.main import solve_position
from pyaudiolocalization_amd.utils import equations


def _tdoas(gold, prefix, fs):
    n2 = int(gold[prefix + "L"][0])
    return [(np.int64(k) - (n2 - 1)) / fs for k in gold[prefix + "k_sel_0p05"]]


def test_positions_from_reference_tables(golden):
    g1 = golden("c1_example1.npz")
    cfg = cases.c1_config()
    mics = np.array(cfg["mic_positions"])
    pos = solve_position(mics, [tuple(p) for p in pair_list(4)], _tdoas(g1, "", 44100), cases.C_SOUND)
    assert np.max(np.abs(pos - g1["position"])) <= 1e-3                       # metres
    g2 = golden("c2_chirp8.npz")
    mics = np.array(cases.c2_config()["mic_positions"])
    for tag in ("a_", "b_"):
        pos = solve_position(mics, [tuple(p) for p in pair_list(8)], _tdoas(g2, tag, 48000), cases.C_SOUND)
        assert np.max(np.abs(pos - g2[tag + "position"])) <= 1e-3, tag


def test_vectorised_residuals_match_the_loop():
    rng = np.random.default_rng(0)
    mics = rng.uniform(-1, 1, (9, 3))
    pairs = [tuple(p) for p in pair_list(9)]
    td = rng.normal(0, 1e-3, len(pairs))
    w = rng.uniform(0.5, 2, len(pairs))
    x = np.array([0.3, -1.2, 2.0])
    want = []
    for k, ((i, j), t) in enumerate(zip(pairs, td)):
        want.append(((np.linalg.norm(x - mics[j]) - np.linalg.norm(x - mics[i])) - 343.62 * t) * w[k])
    assert np.allclose(equations(x, mics, pairs, td, 343.62, w), want, rtol=0, atol=1e-14)
    import pytest
    with pytest.raises(ValueError):
        equations(x, mics, pairs, td, 343.62, w[:-1])


def test_position_c3_grid64(golden):
    """2016 pairs: the clustering + trust-region tail at the size where it dominates the reference's runtime."""
    g = golden("c3_grid64_trial0.npz")
    mics = cases.grid_array_64()
    pos = solve_position(mics, [tuple(p) for p in pair_list(64)], _tdoas(g, "", 48000), cases.C_SOUND)
    assert np.max(np.abs(pos - g["position"])) <= 1e-3
