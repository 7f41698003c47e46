"""localize_sound_source end to end on the HIP engine against the reference's fixtures: positions for C1 / C2a / C2b /
C3 (1e-3 m, north_star), the calibration correction and the per-pair correlation metrics of main.py:147-157,209-222, and
synchronize_signals_improved on recordings of unequal length (utils.py:407-457).

The reference's Butterworth prefilter amplifies a 1e-16 perturbation of its input to 1e-6..5e-3 of its output
(tests/stages.py docstring), so - like every other position test - the stages in front of the TDOA table are fed the
bit-identical rows the reference saw (the oracle's, pinned to the reference's digests); the table comes from the engine
and the host solve is the drop-in's own."""
import numpy as np
import pytest

from oracle import cases

import stages

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def _engine(engine):
    import pyaudiolocalization_amd.engine as E
    E._default = engine
    yield
    E._default = None


def _forced(monkeypatch, case):
    """Teacher-force simulate -> synchronise -> prefilter with the oracle's rows of `case`."""
    from pyaudiolocalization_amd import main as M
    base, delays, gains, fs, total, trim = case
    o = stages.OracleImpl()
    filt = o.prefilter(o.synchronize(o.simulate(base, delays, gains, fs, total, trim), fs), fs)
    monkeypatch.setattr(M, "simulate_signals_with_multipath", lambda **kw: [r for r in filt])
    monkeypatch.setattr(M, "synchronize_signals_improved", lambda s, fs_: s)
    monkeypatch.setattr(M, "noise_reduction_rows", lambda rows, fs_, method="butterworth": np.asarray(rows))
    return M, filt


def _quiet(cfg):
    cfg["localization"] = dict(cfg["localization"], analyze_correlation=False, visualize_correlation=False)
    return cfg


def test_position_c1(golden, tmp_path, monkeypatch):
    """Example 1: four identical signals, every PHAT sequence exactly symmetric, and the reference's choice
    between the mirror peaks 44098 / 44101 is the last bit of pocketfft.  Where the engine's table equals the reference's
    the drop-in's position must; rows that differ must be exact ties of the oracle's own sequence (1e-12), and the
    drop-in's host solve is then checked on the reference's indices."""
    from oracle import pal_oracle as O
    from pyaudiolocalization_amd import pair_list
    monkeypatch.chdir(tmp_path)
    g = golden("c1_example1.npz")
    M, filt = _forced(monkeypatch, stages.c1_case())
    res = M.localize_sound_source(_quiet(cases.c1_config()), use_simulation=True, show_plots=False)
    table = M.tdoa_table(filt, 44100, 0.05)
    differ = np.flatnonzero(table["k_sel"] != g["k_sel_0p05"])
    if differ.size == 0:
        assert np.max(np.abs(res["estimated_position"] - g["position"])) <= 1e-3
        return
    plist = pair_list(4)
    for row in differ:
        c = O.phat_correlation(filt[plist[row][0]], filt[plist[row][1]])
        assert abs(c[int(table["k_sel"][row])] - c[int(g["k_sel_0p05"][row])]) <= 1e-12
    print(f"[parity] C1: {differ.size}/6 selected indices differ at exact ties; position solved from the reference's indices")
    td = [(np.int64(k) - (filt.shape[1] - 1)) / 44100 for k in g["k_sel_0p05"]]
    pos = M.solve_position(np.array(cases.c1_config()["mic_positions"]), [tuple(p) for p in plist], td, cases.C_SOUND)
    assert np.max(np.abs(pos - g["position"])) <= 1e-3


@pytest.mark.parametrize("low,tag", [(False, "a_"), (True, "b_")])
def test_position_c2(golden, tmp_path, monkeypatch, low, tag):
    monkeypatch.chdir(tmp_path)
    g = golden("c2_chirp8.npz")
    M, _ = _forced(monkeypatch, stages.c2_case(low))
    res = M.localize_sound_source(_quiet(cases.c2_config()), use_simulation=True, show_plots=False)
    assert np.max(np.abs(res["estimated_position"] - g[tag + "position"])) <= 1e-3


def test_position_c3(golden, tmp_path, monkeypatch):
    monkeypatch.chdir(tmp_path)
    g = golden("c3_grid64_trial0.npz")
    M, _ = _forced(monkeypatch, stages.c3_case(0))
    res = M.localize_sound_source(_quiet(cases.c3_config(0)), use_simulation=True, show_plots=False)
    assert np.max(np.abs(res["estimated_position"] - g["position"])) <= 1e-3


# ---- the call a user makes: nothing substituted ---------------------------------------------------------
# tests/golden/sensitivity.npz (oracle/make_golden.py golden_sensitivity) holds the reference against ITSELF with its
# simulated signals moved by one unit in the last place: rows of the table that change, metres the position moves.
#     C1  6/6 rows, 13.2 m  (identical signals: every sequence symmetric, the mirror peak is chosen by rounding)
#     C2a 0/28, 0 m         C2b 0/28, 0 m          C3 9/2016 rows, 0.283 m (planar array: the solve is ill-conditioned)
# The engine's simulated signals differ from the reference's by up to 1e-11 (another FFT), i.e. by far more than one ulp,
# so its un-forced table can only be held to the reference's own spread: at most max(3 x the reference's changed rows, 1 % of
# the rows) may differ, and the position may move at most max(1e-3 m, 3 x the reference's own movement).  Where the
# reference is stable (C2a, C2b) that is bit-identical indices and 1e-3 m.
UNFORCED = {"c1": ("c1_example1.npz", "", stages.c1_case, cases.c1_config, None, None),
            "c2a": ("c2_chirp8.npz", "a_", lambda: stages.c2_case(False), cases.c2_config, None, None),
            "c2b": ("c2_chirp8.npz", "b_", lambda: stages.c2_case(True), cases.c2_config, cases.LOW_LOSS, None),
            "c3": ("c3_grid64_trial0.npz", "", lambda: stages.c3_case(0), lambda: cases.c3_config(0), None, lambda: cases.c3_base(0))}


@pytest.mark.parametrize("name", sorted(UNFORCED))
def test_unforced_end_to_end(golden, tmp_path, monkeypatch, name):
    """main.py:165-298 entirely on the engine - simulate -> synchronise -> prefilter -> pair table -> host solve, no stage
    replaced - against the reference's fixture; the differences are printed and held to the reference's own spread."""
    from pyaudiolocalization_amd import main as M
    monkeypatch.chdir(tmp_path)
    fixture, prefix, case, config, materials, base = UNFORCED[name]
    g, sens = golden(fixture), golden("sensitivity.npz")
    if materials is not None:
        monkeypatch.setattr(M, "material_properties", materials)
    if base is not None:
        monkeypatch.setattr(M, "generate_signal", lambda *a, **k: base().copy())   # the trial's noise burst IS the input (as in the fixture)
    e = stages.EngineImpl()
    sig, delays, gains, fs, total, trim = case()
    rows = e.prefilter(e.synchronize(e.simulate(sig, delays, gains, fs, total, trim), fs), fs)
    assert rows.shape[1] == int(g[prefix + "L"][0])
    table = e.pair_table(rows, fs, 0.05)
    want = g[prefix + "k_sel_0p05"]
    differ = int(np.count_nonzero(table["k_sel"] != want))
    res = M.localize_sound_source(_quiet(config()), use_simulation=True, show_plots=False)
    moved = float(np.linalg.norm(res["estimated_position"] - g[prefix + "position"]))
    ref_rows, ref_moved = int(sens[name + "_rows_differ"][0]), float(sens[name + "_position_delta_m"][0])
    print(f"[un-forced] {name}: {differ}/{want.size} selected indices differ from the fixture (reference against itself at one ulp: "
          f"{ref_rows}); position {np.round(res['estimated_position'], 6)} is {moved:.3e} m from the fixture's (reference: {ref_moved:.3e} m)")
    assert differ <= max(3 * ref_rows, want.size // 100), (differ, ref_rows)
    assert moved <= max(1e-3, 3 * ref_moved), (moved, ref_moved)


def test_calibration_correction_and_metrics(golden, tmp_path, monkeypatch):
    """main.py:147-157 (calibration delays), :209-212 (td - (delay_j - delay_i)), :219-222 (per-pair metrics with the
    1000-shuffle bootstrap drawn from the global NumPy RNG, seeded like the fixture), :254-257 (SNR weights)."""
    monkeypatch.chdir(tmp_path)
    g = golden("localize_extras.npz")
    M, _ = _forced(monkeypatch, stages.loc_case())
    plain = M.localize_sound_source(cases.loc_config(False), use_simulation=True, show_plots=False)
    assert np.max(np.abs(plain["estimated_position"] - g["loc_position_plain"])) <= 1e-3
    assert plain["correlation_metrics"] is None
    bad = M.localize_sound_source(cases.loc_config(False), calibration_data=cases.LOC_CALIBRATION[:3], use_simulation=True,
                                  show_plots=False)                              # main.py:148-150: ignored with a warning
    assert np.array_equal(bad["estimated_position"], plain["estimated_position"])
    cal = M.localize_sound_source(cases.loc_config(False), calibration_data=cases.LOC_CALIBRATION, use_simulation=True,
                                  show_plots=False)
    assert np.max(np.abs(cal["estimated_position"] - g["loc_position_calib"])) <= 1e-3
    assert cal["calibration_data"] is cases.LOC_CALIBRATION
    np.random.seed(cases.LOC_SEED)
    full = M.localize_sound_source(cases.loc_config(True), calibration_data=cases.LOC_CALIBRATION, use_simulation=True,
                                   show_plots=False)
    assert np.max(np.abs(full["estimated_position"] - g["loc_position_metrics"])) <= 1e-3
    metrics = full["correlation_metrics"]
    pairs = [tuple(int(v) for v in p) for p in g["loc_metric_pairs"]]
    assert sorted(metrics) == pairs
    snr = np.array([metrics[p]["snr"] for p in pairs])
    ptp = np.array([metrics[p]["peak_to_peak_ratio"] for p in pairs])
    assert np.allclose(snr, g["loc_snr"], rtol=1e-9) and np.allclose(ptp, g["loc_ptp"], rtol=1e-9)
    assert [bool(metrics[p]["significant"]) for p in pairs] == [bool(v) for v in g["loc_significant"]]


def test_synchronise_unequal_lengths(golden):
    """Recordings that differ in length by a few samples (VERDICT r1 'missing' 7): same pads and rows as the
    reference."""
    from pyaudiolocalization_amd.utils import synchronize_signals_improved
    g = golden("localize_extras.npz")
    sigs, fs = cases.unequal_sync_signals()
    got = synchronize_signals_improved(sigs, fs)
    assert np.array_equal([len(s) for s in got], g["uneq_len"])
    assert np.array_equal([int(np.flatnonzero(s)[0]) for s in got], g["uneq_first_nonzero"])
    stages.digest_close(got, g["uneq_digest"], 1e-13)
    want = stages.OracleImpl().synchronize(sigs, fs)
    assert all(np.array_equal(a, b) for a, b in zip(got, want))
