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


def test_positions_with_calibration_correction_and_snr_weights(golden):
    """main.py:209-212 (td - (delay_j - delay_i)) and main.py:254-257 (SNR weights from the per-pair metrics): the
    reference's own selected indices and SNRs in, the reference's positions out."""
    from pyaudiolocalization_amd.utils import compute_weights
    g = golden("localize_extras.npz")
    cfg = cases.loc_config(False)
    mics = np.array(cfg["mic_positions"])
    pairs = [tuple(p) for p in pair_list(5)]
    td = np.array(_tdoas(g, "loc_", cfg["fs"]))
    assert np.max(np.abs(solve_position(mics, pairs, list(td), cases.C_SOUND) - g["loc_position_plain"])) <= 1e-3
    assert np.max(np.abs(g["loc_position_badcalib"] - g["loc_position_plain"])) == 0      # main.py:148-150: ignored
    delays = np.array([d["delay"] for d in cases.LOC_CALIBRATION])
    tdc = [t - (delays[j] - delays[i]) for t, (i, j) in zip(td, pairs)]
    assert np.max(np.abs(solve_position(mics, pairs, tdc, cases.C_SOUND) - g["loc_position_calib"])) <= 1e-3
    assert [tuple(p) for p in g["loc_metric_pairs"]] == pairs
    metrics = {p: {"snr": s} for p, s in zip(pairs, g["loc_snr"])}
    w = compute_weights(metrics, pairs)
    assert np.max(np.abs(solve_position(mics, pairs, tdc, cases.C_SOUND, w) - g["loc_position_metrics"])) <= 1e-3
    assert np.max(np.abs(g["loc_position_metrics"] - g["loc_position_calib"])) > 0        # the weights do matter here
