"""HIP engine (through the drop-in modules and the C ABI) against the oracle and the reference
fixtures.  Integer outputs bit-exact; float outputs to the tolerance written at each assertion."""
import os

import numpy as np
import pytest

from oracle import cases
from oracle import pal_oracle as O

import stages
from stages import EngineImpl, check_table, digest_close, tag_of

pytestmark = pytest.mark.gpu
IMPL = EngineImpl()


@pytest.fixture(scope="module", autouse=True)
def _engine(engine):
    """The drop-in modules use the process-wide default engine; make it the session engine."""
    import pyaudiolocalization_amd.engine as E
    E._default = engine
    yield
    E._default = None


# ---------------------------------------------------------------- exact-length DFT / PHAT sequence
@pytest.mark.parametrize("n1,n2", [(50, 50), (97, 64), (1000, 1000), (2049, 2047), (4096, 4096), (12000, 12000),
                                   (24000, 24000), (44100, 44100)])
def test_phat_correlation_matches_numpy(engine, n1, n2):
    rng = np.random.default_rng(n1 + n2)
    a, b = rng.standard_normal(n1), rng.standard_normal(n2)
    got = engine.phat_correlation(a, b)
    want = O.phat_correlation(a, b)
    assert got.shape == want.shape
    # PHAT values are O(1/sqrt(n)) with a unit-ish peak; fp64 chirp-z vs pocketfft agree to ~1e-15 absolute
    assert np.max(np.abs(got - want)) <= 5e-14, float(np.max(np.abs(got - want)))


# frame lengths whose n = 2L-1 exercises every shape of the prime-factor route (pfa.hip): one row tile
# (99 = 1 x 99, 999 = 1 x 999; 991 = 1 x 991 and 88199 = 89 x 991 take the Rader row pass of pfa_rader.h), dense column DFTs with N1 = 9 / 7 / 3 / 11 / 89, tiles of 1024 / 2048 / 4096 points,
# and lengths that have no usable split (1999 prime) and stay on the four-step chirp convolution
PFA_LENGTHS = [(50, 1, 99, 512), (496, 1, 991, 990), (500, 1, 999, 2048), (1000, 0, 0, 0), (2048, 9, 455, 1024), (2999, 3, 1999, 4096),
               (3000, 7, 857, 2048), (5000, 11, 909, 2048), (44100, 89, 991, 990),
               # register-resident row tiles (pfa_big.h): 8192 points at 20465 = 5 x 4093, 16384 points at C3's 47999 = 7 x 6857
               (10233, 5, 4093, 8192), (24000, 7, 6857, 16384),
               # 512-point LDS tiles (N2 <= 256): C5's 23999 = 103 x 233
               (12000, 103, 233, 512), (11170, 89, 251, 512)]


@pytest.mark.parametrize("length,n1,n2,tile", PFA_LENGTHS)
def test_prime_factor_route_matches_four_step_and_numpy(engine, length, n1, n2, tile, monkeypatch):
    """The two inverse-transform routes are independent code (CRT maps + in-LDS chirp convolutions + dense column
    DFTs against the four-step workspace passes): both must reproduce numpy's exact-length ifft to rounding."""
    from pyaudiolocalization_amd import Engine
    info = engine.plan_info(length)
    assert (info["n1"], info["n2"], info["tile_len"]) == (n1, n2, tile), info
    if n1:
        assert n1 * n2 == info["n"]
    rng = np.random.default_rng(length)
    a, b = rng.standard_normal(length), rng.standard_normal(length)
    want = O.phat_correlation(a, b)
    got = engine.phat_correlation(a, b)
    monkeypatch.setenv("PAL_PFA", "0")
    plain = Engine(engine.device)
    try:
        assert plain.plan_info(length)["n1"] == 0
        ref = plain.phat_correlation(a, b)
    finally:
        plain.close()
    scale = np.max(np.abs(want))
    assert np.max(np.abs(got - want)) <= 1e-12 * scale          # fp64 rounding of a length-n transform: ~1e-14
    assert np.max(np.abs(ref - want)) <= 1e-12 * scale
    assert int(np.argmax(got)) == int(np.argmax(want)) == int(np.argmax(ref))


def test_rader_rows_match_chirp_convolution_rows(engine, monkeypatch):
    """n2 = 991: the Rader row pass (pfa_rader.h) against the chirp-convolution tile (PAL_RADER=0) on the same plan."""
    from pyaudiolocalization_amd import Engine
    assert engine.plan_info(496)["tile_len"] == 990
    rng = np.random.default_rng(991)
    frames = rng.standard_normal((2, 5, 496))
    frames[:, 1:] += 0.5 * frames[:, :1]
    t1, c1 = engine.gcc_phat_all_pairs(frames, 16000.0, max_expected_delay=0.01, want_corr=True)
    monkeypatch.setenv("PAL_RADER", "0")
    plain = Engine(engine.device)
    try:
        assert plain.plan_info(496)["tile_len"] == 2048
        t0, c0 = plain.gcc_phat_all_pairs(frames, 16000.0, max_expected_delay=0.01, want_corr=True)
    finally:
        plain.close()
    assert np.max(np.abs(c1 - c0)) <= 1e-13
    for name in ("k_sel", "branch", "k_argmax", "n_sel"):
        assert np.array_equal(t1[name], t0[name]), name
    for b in range(2):
        want = O.all_pairs(frames[b], 16000.0, max_expected_delay=0.01)
        for name in ("k_sel", "branch", "k_argmax"):
            assert np.array_equal(t1[b][name], want[name]), name


def test_forward_spectra_through_the_prime_factor_cut(engine, monkeypatch):
    """Plans with Rader rows take the forward transform through the same cut (pfa_forward.h: two real frames per transform);
    PAL_PFA_FWD=0 keeps the four-step chirp convolution.  Odd frame counts, unequal lengths, a silent frame."""
    from pyaudiolocalization_amd import Engine
    rng = np.random.default_rng(4)
    engine.profile_begin()
    engine.gcc_phat_all_pairs(rng.standard_normal((1, 3, 496)), 16000.0)
    engine.profile_end()
    assert engine.profile_entries()["k_pfa_fwd_cols"][1] >= 1
    monkeypatch.setenv("PAL_PFA_FWD", "0")
    plain = Engine(engine.device)
    try:
        for mics, length, fs, med in ((3, 496, 16000.0, None), (5, 496, 8000.0, 0.01), (3, 44100, 44100.0, 0.05)):
            frames = rng.standard_normal((2, mics, length))
            frames[:, 1:] += 0.5 * frames[:, :1]
            frames[1, 0] = 0.0
            t1, c1 = engine.gcc_phat_all_pairs(frames, fs, max_expected_delay=med, want_corr=True)
            plain.profile_begin()
            t0, c0 = plain.gcc_phat_all_pairs(frames, fs, max_expected_delay=med, want_corr=True)
            plain.profile_end()
            assert "k_pfa_fwd_cols" not in plain.profile_entries()
            assert np.max(np.abs(c1 - c0)) <= 1e-13
            for name in ("k_sel", "branch", "k_argmax", "n_sel"):
                assert np.array_equal(t1[name], t0[name]), name
            if length < 1000:
                for b in range(2):
                    want = O.all_pairs(frames[b], fs, max_expected_delay=med)
                    for name in ("k_sel", "branch", "k_argmax"):
                        assert np.array_equal(t1[b][name], want[name]), name
        a, b = rng.standard_normal(600), rng.standard_normal(392)       # n = 991 with unequal lengths
        want = O.phat_correlation(a, b)
        assert np.max(np.abs(engine.phat_correlation(a, b) - want)) <= 5e-14
        assert np.max(np.abs(plain.phat_correlation(a, b) - want)) <= 5e-14
    finally:
        plain.close()


def test_prime_factor_route_odd_pair_counts_and_tables(engine, monkeypatch):
    """5 mics = 10 pairs = 5 packed transforms, 3 mics = 3 pairs (one half-empty transform): tables and
    sequences of the prime-factor route equal the four-step route's."""
    from pyaudiolocalization_amd import Engine
    rng = np.random.default_rng(77)
    monkeypatch.setenv("PAL_PFA", "0")
    plain = Engine(engine.device)
    try:
        for mics, length, fs, med in ((5, 3000, 16000.0, 0.004), (3, 2048, 8000.0, None), (6, 500, 8000.0, None)):
            frames = rng.standard_normal((2, mics, length))
            frames[:, 1:] += 0.6 * frames[:, :1]
            frames[1, mics - 1] = 0.0          # a silent microphone: its pairs' rows are exactly zero in the reference
            t1, c1 = engine.gcc_phat_all_pairs(frames, fs, max_expected_delay=med, want_corr=True)
            t0, c0 = plain.gcc_phat_all_pairs(frames, fs, max_expected_delay=med, want_corr=True)
            assert np.max(np.abs(c1 - c0)) <= 1e-13
            for name in ("k_sel", "branch", "k_argmax", "n_sel"):
                assert np.array_equal(t1[name], t0[name]), name
            for b in range(2):
                want = O.all_pairs(frames[b], fs, max_expected_delay=med)
                for name in ("k_sel", "branch", "k_argmax"):
                    assert np.array_equal(t1[b][name], want[name]), name        # bit-exact integer outputs
                    assert np.array_equal(t0[b][name], want[name]), name
                assert np.allclose(t1[b]["cmax"], want["cmax"], rtol=1e-11, atol=0)
                assert np.allclose(t1[b]["snr"], want["snr"], rtol=1e-9, atol=0)
    finally:
        plain.close()


# frame lengths whose split has N1 <= 89: there the column pass of the prime-factor route also does the streaming pass of
# the peak selection, every column block with its own pivots (pfa_cols_stats.h; PAL_FUSED=0 keeps the separate launches)
# convolution geometries of the four-step route with register-resident rows (conv_kernels.h k_colsreg_* / k_rowsreg): frame
# length -> (columns, row length) of the PHAT inverse's convolution; the LDS-tile passes (PAL_FOUR_REG=0) are the independent route
REG_GEOMETRIES = [(12000, 12, 4096), (16000, 16, 4096), (18000, 18, 4096), (20000, 20, 4096), (22000, 22, 4096), (24000, 24, 4096),
                  (30000, 16, 8192), (36000, 18, 8192), (40500, 20, 8192), (44102, 22, 8192), (48001, 24, 8192),
                  # two lanes per column
                  (60000, 32, 8192), (90000, 48, 8192)]


@pytest.mark.parametrize("length,m1,m2", REG_GEOMETRIES)
def test_register_row_four_step_matches_lds_tiles_and_numpy(length, m1, m2, monkeypatch):
    """Both four-step cuts must reproduce numpy's exact-length transforms to rounding and select the same peaks; the
    prime-factor route is off in both engines so that the PHAT inverse really runs the convolution under test."""
    from pyaudiolocalization_amd import Engine
    monkeypatch.setenv("PAL_PFA", "0")
    reg = Engine(0)
    monkeypatch.setenv("PAL_FOUR_REG", "0")
    lds = Engine(0)
    try:
        info = reg.plan_info(length)
        assert (info["m1"], info["m2"]) == (m1, m2) and info["n1"] == 0, info
        assert lds.plan_info(length)["m2"] <= 2048
        rng = np.random.default_rng(length)
        frames = rng.standard_normal((1, 4, length))
        frames[0, 1:] += 0.6 * np.roll(frames[0, :1], 37, axis=1)
        t1, c1 = reg.gcc_phat_all_pairs(frames, 16000.0, want_corr=True)
        t0, c0 = lds.gcc_phat_all_pairs(frames, 16000.0, want_corr=True)
        assert np.array_equal(t1["k_sel"], t0["k_sel"]) and np.array_equal(t1["branch"], t0["branch"])
        assert np.max(np.abs(c1 - c0)) <= 1e-13
        want = O.phat_correlation(frames[0, 0], frames[0, 1])
        assert np.max(np.abs(c1[0, 0] - want)) <= 1e-13
    finally:
        reg.close()
        lds.close()


FUSED_LENGTHS = [(11962, 47, 509), (15525, 61, 509), (22651, 89, 509), (44100, 89, 991), (7890, 31, 509), (1008, 5, 403),
                 (10233, 5, 4093), (24000, 7, 6857),
                 # 512-point row tiles under the fused column pass: 22339 = 89 x 251
                 (11170, 89, 251)]


def _plan_of(engine, length):
    info = engine.plan_info(length)
    return info["n1"], info["n2"]


@pytest.mark.parametrize("length,n1,n2", FUSED_LENGTHS)
@pytest.mark.parametrize("method", ["median", "adaptive"])
def test_fused_column_pass_matches_separate_launches_and_oracle(engine, length, n1, n2, method, monkeypatch):
    from pyaudiolocalization_amd import Engine
    assert _plan_of(engine, length) == (n1, n2)
    rng = np.random.default_rng(length)
    frames = rng.standard_normal((2, 5, length))                # 10 pairs = 5 packed transforms per frame
    frames[:, 1:] += 0.5 * frames[:, :1]
    frames[1, 3] = 0.0                                          # a silent microphone: four all-zero correlation rows (plateaus)
    fs, med = 16000.0, 0.02
    engine.profile_begin()
    t1, c1 = engine.gcc_phat_all_pairs(frames, fs, max_expected_delay=med, threshold_method=method, want_corr=True)
    engine.profile_end()
    ent = engine.profile_entries()
    assert ent["k_pfa_cols_stats"][1] >= 1 and "k_peak_stream" not in ent and "k_peak_pivots" not in ent, ent
    monkeypatch.setenv("PAL_FUSED", "0")
    plain = Engine(engine.device)
    try:
        plain.profile_begin()
        t0, c0 = plain.gcc_phat_all_pairs(frames, fs, max_expected_delay=med, threshold_method=method, want_corr=True)
        plain.profile_end()
        ent = plain.profile_entries()
        assert "k_pfa_cols_stats" not in ent and ent["k_peak_stream"][1] >= 1, ent
    finally:
        plain.close()
    if n1 == 89 and os.environ.get("PAL_R89", "1") != "0":
        # 89-point columns: the fused pass runs Rader's 8 x 11 convolution (csrc/pfa_rader89.h), the separate column pass the
        # dense form - other arithmetic, the same sequence to rounding (samples are below 1 in magnitude)
        assert np.max(np.abs(c1 - c0)) <= 4e-15
        for name in ("k_sel", "branch", "k_argmax", "n_sel"):
            assert np.array_equal(t1[name], t0[name]), name
        for name in ("cmax", "cmin", "sel_height"):
            assert np.allclose(t1[name], t0[name], rtol=1e-12, atol=4e-15), name
    else:
        assert np.array_equal(c1, c0)                           # the same FMAs in the same order
        for name in ("k_sel", "branch", "k_argmax", "n_sel", "cmax", "cmin", "sel_height"):
            assert np.array_equal(t1[name], t0[name]), name
    assert np.allclose(t1["snr"], t0["snr"], rtol=1e-11, atol=0)   # sums with other shifts, added in another order
    for b in range(2):
        want = O.all_pairs(frames[b], fs, max_expected_delay=med, threshold_method=method)
        for name in ("k_sel", "branch", "k_argmax"):
            assert np.array_equal(t1[b][name], want[name]), name
        assert np.allclose(t1[b]["snr"], want["snr"], rtol=1e-9, atol=0)


# the finishing column pass (csrc/pfa_cols_fin.h, pfa_fin_lean.h) in each of its forms: (length, n1, n2, environment)
FIN_CASES = [(44100, 89, 991, {}),                              # Rader-89 columns, four wavefronts
             (44113, 25, 3529, {}),                             # dense columns, two chunks = two wavefronts
             (44110, 47, 1877, {}),                             # dense columns, three chunks = three wavefronts, unused output indices
             (44254, 67, 1321, {}),                             # dense columns, four chunks
             (44103, 23, 3835, {}),                             # strips (one chunk, four strips per block)
             (44101, 0, 0, {"PAL_FIN_FOUR": "1"})]              # four-step last pass (opt-in): the last grid row is partial


@pytest.mark.parametrize("length,n1,n2,env", FIN_CASES, ids=[f"L{c[0]}" for c in FIN_CASES])
def test_finishing_column_pass_equals_stored_rows(engine, length, n1, n2, env, monkeypatch):
    """Records of the pass that never stores the correlation rows against the stored-row path (PAL_FIN=0), field by field, for
    noise, delayed copies, a tone (fallback branches) and a silent microphone; windowed and not; 'median' through the bound
    (multiplier 1), through histogram windows (4.2) and 'adaptive'.  Integer fields bit-exact; float fields to 1e-11 (sums in
    another order).  The pass must really have run (profile entries)."""
    from pyaudiolocalization_amd import Engine
    assert _plan_of(engine, length) == (n1, n2)
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    rng = np.random.default_rng(length)
    mics, fs = 5, 44100.0
    base = rng.standard_normal(length + 64)
    cases = {"noise": rng.standard_normal((1, mics, length)),
             "delayed": (np.stack([base[d:d + length] for d in rng.integers(0, 64, mics)]) + 0.3 * rng.standard_normal((mics, length)))[None],
             "tone": (np.sin(0.05 * np.arange(length))[None, :] + 0.3 * rng.standard_normal((mics, length)))[None]}
    silent = rng.standard_normal((1, mics, length))
    silent[0, 1] = 0.0
    cases["silent"] = silent
    monkeypatch.setenv("PAL_FIN", "1")
    fin = Engine(engine.device)
    monkeypatch.setenv("PAL_FIN", "0")
    stored = Engine(engine.device)
    try:
        for name, fr in cases.items():
            for med in (0.05, None):
                for method, mult in (("median", 1.0), ("median", 4.2), ("adaptive", 1.0)):
                    if n1 == 0 and mult > 2.0:
                        continue                                # (the four-step pass has no histogram form: stored rows both ways)
                    fin.profile_begin()
                    ta = fin.gcc_phat_all_pairs(fr, fs, 1, method, mult, med)
                    fin.profile_end()
                    ent = fin.profile_entries()
                    assert any(k.startswith("k_pfa_cols_fin") or k.startswith("k_colsreg_fin") for k in ent), (name, med, method, mult, sorted(ent))
                    tb = stored.gcc_phat_all_pairs(fr, fs, 1, method, mult, med)
                    tag = (name, med, method, mult)
                    for f in ("k_sel", "branch", "k_argmax", "n_sel"):
                        assert np.array_equal(ta[f], tb[f]), (tag, f)
                    for f in ("cmax", "cmin", "snr", "sel_height"):
                        assert np.allclose(ta[f], tb[f], rtol=1e-11, atol=1e-300), (tag, f)
    finally:
        fin.close()
        stored.close()


@pytest.mark.parametrize("length,fs", [(24000, 48000.0), (44100, 44100.0)], ids=["strips-7x6857", "rader89-89x991"])
def test_stored_row_pass_with_wavefront_statistics_equals_round2_statistics(engine, length, fs, monkeypatch):
    """Where the rows are stored (the caller wants corr, or the plan has no finishing form) the column pass carries the
    per-wavefront statistics and its finisher reads the SNR window from the stored row (k_pfa_cols_lean, PAL_LEAN_STORE default):
    records and rows against pfa_cols_stats.h + k_peak_finish (PAL_LEAN_STORE=0)."""
    from pyaudiolocalization_amd import Engine
    rng = np.random.default_rng(length + 1)
    mics = 5
    base = rng.standard_normal(length + 64)
    cases = {"noise": rng.standard_normal((2, mics, length)),
             "delayed": (np.stack([base[d:d + length] for d in rng.integers(0, 64, mics)]) + 0.3 * rng.standard_normal((mics, length)))[None],
             "tone": (np.sin(0.05 * np.arange(length))[None, :] + 0.3 * rng.standard_normal((mics, length)))[None]}
    monkeypatch.setenv("PAL_LEAN_STORE", "1")
    lean = Engine(engine.device)
    monkeypatch.setenv("PAL_LEAN_STORE", "0")
    old = Engine(engine.device)
    try:
        for name, fr in cases.items():
            for med in (0.05, None):
                for method in ("median", "adaptive"):
                    lean.profile_begin()
                    ta, ca = lean.gcc_phat_all_pairs(fr, fs, 1, method, 1.0, med, want_corr=True)
                    lean.profile_end()
                    assert "k_pfa_cols_lean" in lean.profile_entries(), sorted(lean.profile_entries())
                    tb, cb = old.gcc_phat_all_pairs(fr, fs, 1, method, 1.0, med, want_corr=True)
                    tag = (name, med, method)
                    assert np.allclose(ca, cb, rtol=0, atol=4e-15), tag
                    for f in ("k_sel", "branch", "k_argmax", "n_sel"):
                        assert np.array_equal(ta[f], tb[f]), (tag, f)
                    for f in ("cmax", "cmin", "snr", "sel_height"):
                        assert np.allclose(ta[f], tb[f], rtol=1e-11, atol=1e-300), (tag, f)
    finally:
        lean.close()
        old.close()


@pytest.mark.parametrize("length,fs", [(12000, 48000.0), (6007, 16000.0), (2500, 8000.0)], ids=["pfa-103x233", "four-step-12013", "four-step-4999"])
def test_row_major_statistics_pass_equals_the_three_launches(engine, length, fs, monkeypatch):
    """k_rows_lean (per-wavefront statistics over rows already in HBM: 22 overlapping 64-sample chunks per wavefront, the row's last
    chunk partial) against k_peak_pivots + k_peak_stream + k_peak_finish.  The engine takes it for calls of 200 000 pairs or more;
    PAL_ROWS_LEAN_MIN=1 lets a small call through."""
    from pyaudiolocalization_amd import Engine
    rng = np.random.default_rng(length + 2)
    mics = 6
    base = rng.standard_normal(length + 64)
    cases = {"noise": rng.standard_normal((2, mics, length)),
             "delayed": (np.stack([base[d:d + length] for d in rng.integers(0, 64, mics)]) + 0.3 * rng.standard_normal((mics, length)))[None],
             "tone": (np.sin(0.05 * np.arange(length))[None, :] + 0.3 * rng.standard_normal((mics, length)))[None]}
    silent = rng.standard_normal((1, mics, length))
    silent[0, 2] = 0.0
    cases["silent"] = silent
    monkeypatch.setenv("PAL_ROWS_LEAN_MIN", "1")
    monkeypatch.setenv("PAL_ROWS_LEAN", "1")
    lean = Engine(engine.device)
    monkeypatch.setenv("PAL_ROWS_LEAN", "0")
    old = Engine(engine.device)
    try:
        for name, fr in cases.items():
            for med in (0.05, None, 0.001):
                for method, mult in (("median", 1.0), ("adaptive", 1.0), ("median", 2.0)):
                    lean.profile_begin()
                    ta = lean.gcc_phat_all_pairs(fr, fs, 1, method, mult, med)
                    lean.profile_end()
                    assert "k_rows_lean" in lean.profile_entries(), sorted(lean.profile_entries())
                    tb = old.gcc_phat_all_pairs(fr, fs, 1, method, mult, med)
                    tag = (name, med, method, mult)
                    for f in ("k_sel", "branch", "k_argmax", "n_sel"):
                        assert np.array_equal(ta[f], tb[f]), (tag, f)
                    for f in ("cmax", "cmin", "snr", "sel_height"):
                        assert np.allclose(ta[f], tb[f], rtol=1e-11, atol=1e-300), (tag, f)
    finally:
        lean.close()
        old.close()


def test_pair_blocking_of_large_arrays_returns_the_row_major_table(engine, monkeypatch):
    """From 96 microphones up the pairs are processed in 16 x 16 blocks of microphones (a launch group then re-uses its spectra,
    csrc/pal_api.hip build_quads_blocked) and the records are scattered back: the table must be the row-major i < j table.  Another
    pair shares each packed transform, so float fields may differ by the partner's rounding noise; the indices may not."""
    from pyaudiolocalization_amd import Engine
    rng = np.random.default_rng(4)
    monkeypatch.setenv("PAL_PAIR_BLOCK", "16")
    blocked = Engine(engine.device)
    monkeypatch.setenv("PAL_PAIR_BLOCK", "0")
    plain = Engine(engine.device)
    try:
        for mics, length, frames in ((100, 3000, 2), (97, 2047, 1)):
            fr = rng.standard_normal((frames, mics, length))
            fr[0, 5] = 0.0                                          # a silent microphone
            for med in (0.05, None):
                ta = blocked.gcc_phat_all_pairs(fr, 16000.0, 1, "median", 1.0, med)
                tb = plain.gcc_phat_all_pairs(fr, 16000.0, 1, "median", 1.0, med)
                assert ta.shape == tb.shape == (frames, mics * (mics - 1) // 2)
                for f in ("k_sel", "branch", "k_argmax", "n_sel"):
                    assert np.array_equal(ta[f], tb[f]), (mics, med, f)
                for f in ("cmax", "cmin", "snr", "sel_height"):
                    assert np.allclose(ta[f], tb[f], rtol=1e-10, atol=1e-15), (mics, med, f)
        want = O.all_pairs(fr[0][:12], 16000.0, max_expected_delay=None)   # (and against the oracle: the first 12 microphones' pairs)
        got = blocked.gcc_phat_all_pairs(fr[:, :12], 16000.0, 1, "median", 1.0, None)[0]
        assert np.array_equal(got["k_sel"], want["k_sel"])
    finally:
        blocked.close()
        plain.close()


def test_fused_column_pass_plateaus_and_grid_edges(engine, monkeypatch):
    """Samples with equal neighbours and peaks in the grid's first / last column (lags m = 0 or N2 - 1 mod N2) take the
    finish launch's own tests: quantised inputs give exact ties, and the window is the whole row."""
    from pyaudiolocalization_amd import Engine
    length = 11962
    rng = np.random.default_rng(5)
    frames = np.round(rng.standard_normal((3, 3, length)) * 2) / 2     # 3 pairs: the second packed transform is half empty
    frames[:, 1] = np.roll(frames[:, 0], 509, axis=-1)          # true lag = a multiple of N2 = 509: column 0 of the grid
    frames[:, 2] = np.roll(frames[:, 0], -508, axis=-1)         # column N2 - 1, one output index lower
    t1 = engine.gcc_phat_all_pairs(frames, 8000.0)
    monkeypatch.setenv("PAL_FUSED", "0")
    plain = Engine(engine.device)
    try:
        t0 = plain.gcc_phat_all_pairs(frames, 8000.0)
    finally:
        plain.close()
    for name in ("k_sel", "branch", "k_argmax", "n_sel"):
        assert np.array_equal(t1[name], t0[name]), name
    for name in ("cmax", "cmin", "sel_height"):
        # (the column pass that finishes its rows itself, PAL_FIN=1, sends rows with ties through the stored-row path at the end of
        #  the call, packed with another partner pair: last-bit differences)
        assert np.allclose(t1[name], t0[name], rtol=1e-12, atol=4e-15), name
    for b in range(3):
        want = O.all_pairs(frames[b], 8000.0)
        for name in ("k_sel", "branch", "k_argmax"):
            assert np.array_equal(t1[b][name], want[name]), name


def test_fused_column_pass_hard_medians(engine):
    """Rows whose |corr| distribution is far from the noise-like case the per-block pivots are sized for: a delta
    sequence (identical signals), sparse inputs, a tone, and a row of exact zeros.  The median bracket may miss there -
    the exact radix select over the row then decides - and the selected indices must still equal the oracle's."""
    length = 11962
    rng = np.random.default_rng(9)
    base = rng.standard_normal(length)
    t = np.arange(length) / 16000.0
    frames = np.stack([base, base.copy(), np.roll(base, 7) + 1e-3 * rng.standard_normal(length),
                       rng.standard_normal(length) * (rng.random(length) < 0.02), np.zeros(length),
                       np.sin(2 * np.pi * 440 * t) + 1e-3 * rng.standard_normal(length)])
    for med in (None, 0.01):
        got = engine.gcc_phat_all_pairs(frames, 16000.0, max_expected_delay=med)
        want = O.all_pairs(frames, 16000.0, max_expected_delay=med)
        i, j = np.triu_indices(6, k=1)
        for p in range(15):
            if got["k_sel"][p] == want["k_sel"][p] and got["k_argmax"][p] == want["k_argmax"][p]:
                continue
            c = O.phat_correlation(frames[i[p]], frames[j[p]])             # ill-conditioned rows: exact ties only
            gap = max(abs(c[got["k_sel"][p]] - c[want["k_sel"][p]]), abs(c[got["k_argmax"][p]] - c[want["k_argmax"][p]]))
            assert gap <= 1e-12 * max(1.0, np.max(np.abs(c))) or np.median(np.abs(c)) <= 1e-12 * np.max(np.abs(c)), (p, med, gap)



def test_phat_of_identical_and_silent_signals(engine):
    x = np.random.default_rng(1).standard_normal(500)
    assert np.argmax(engine.phat_correlation(x, x)) == 0
    z = engine.phat_correlation(np.zeros(300), np.zeros(300))
    assert np.array_equal(z, np.zeros(599))                       # R = 0 / (0 + 1e-10)


def test_selection_edge_cases(engine, golden):
    """Every branch of the fallback chain (utils.py:153-172) on tiny inputs, unequal lengths, even n,
    'adaptive' and unknown threshold methods, zero-width windows; fixture rows come from the reference."""
    rows = golden("selection_edges.npz")["rows"]
    todo = cases.selection_edge_cases()
    seen = set()
    for case, want in zip(todo, rows):
        ks, rec, corr = engine.get_time_delays_phat(case["a"], case["b"], case["fs"], 1, case["method"], case["mult"], case["med"])
        ref_corr = O.phat_correlation(case["a"], case["b"])
        # pure tones and delta-like / all-zero sequences leave most cross-spectrum bins at rounding level; the
        # whitening R / (|R| + 1e-10) then amplifies the rounding noise of whichever FFT is in use (1e-6 here)
        ill_conditioned = case["t"] >= 10000 or case["t"] % 5 == 0
        assert np.max(np.abs(corr - ref_corr)) <= (2e-5 if ill_conditioned else 1e-13)
        # the selection kernel against the restated logic on the SAME sequence: exact, whatever the input
        own, own_br = O.select_peaks(corr, len(case["b"]), case["fs"], 1, case["method"], case["mult"], case["med"])
        assert int(ks[0]) == int(own[0]) and int(rec["branch"]) == own_br, (case["t"], int(ks[0]), int(own[0]), own_br)
        assert int(rec["k_argmax"]) == int(np.argmax(corr))
        seen.add(own_br)
        if ill_conditioned:
            continue
        _, br = O.select_peaks(ref_corr, len(case["b"]), case["fs"], 1, case["method"], case["mult"], case["med"])
        assert int(ks[0]) == int(want[1]) and int(rec["k_sel"]) == int(want[1]), (case["t"], int(ks[0]), int(want[1]), br)
        assert int(rec["branch"]) == br, (case["t"], int(rec["branch"]), br)
        assert int(rec["k_argmax"]) == int(want[4])
        assert np.isclose(rec["cmax"], want[2], rtol=1e-10, atol=1e-14) and np.isclose(rec["cmin"], want[3], rtol=1e-10, atol=1e-14)
        if np.isfinite(want[5]):
            assert np.isclose(rec["snr"], want[5], rtol=1e-8), (case["t"], float(rec["snr"]), want[5])
    assert {0, 1, 4, 12, 13}.issubset(seen), seen


def test_num_peaks_and_methods(engine):
    rng = np.random.default_rng(3)
    for trial in range(40):
        n = int(rng.integers(200, 3000))
        a = rng.standard_normal(n)
        b = np.roll(a, int(rng.integers(-20, 20))) + 0.3 * rng.standard_normal(n)
        fs = float(rng.choice([8000.0, 16000.0, 48000.0]))
        med = [None, 0.01, 0.002][trial % 3]
        meth = ["median", "adaptive"][trial % 2]
        npk = [1, 3, 5, 16][trial % 4]
        ks, rec, corr = engine.get_time_delays_phat(a, b, fs, npk, meth, 1.0, med)
        want, br = O.select_peaks(O.phat_correlation(a, b), n, fs, npk, meth, 1.0, med)
        assert np.array_equal(ks, want), (trial, ks, want)
        assert int(rec["branch"]) == br


def test_many_peaks_and_the_exact_slow_path_of_the_distance_rule(monkeypatch):
    """utils.py:152,176-179 put no limit on num_peaks or on chains of the distance rule.  64 peaks at fs = 192 kHz (distance 192)
    against the oracle; then the same selections with the on-chip memo / stack shrunk to two entries (PAL_DEBUG_MEMO=2), which
    sends every suppression chain through the exact slow path (scipy's greedy pass over bitmaps in global memory)."""
    from pyaudiolocalization_amd import Engine
    rng = np.random.default_rng(13)
    cases = []
    for trial in range(6):
        n = int(rng.integers(6000, 20000))
        a = rng.standard_normal(n)
        b = np.roll(a, int(rng.integers(-50, 50))) + [0.3, 1.0, 3.0][trial % 3] * rng.standard_normal(n)
        cases.append((a, b, [192000.0, 48000.0][trial % 2], [64, 200, 7][trial % 3], [None, 0.01][trial % 2], ["median", "adaptive"][trial // 3]))
    for memo in ("", "2"):
        if memo:
            monkeypatch.setenv("PAL_DEBUG_MEMO", memo)
        eng = Engine(0)
        try:
            for a, b, fs, npk, med, meth in cases:
                ks, rec, corr = eng.get_time_delays_phat(a, b, fs, npk, meth, 0.5, med)
                want, br = O.select_peaks(O.phat_correlation(a, b), a.size, fs, npk, meth, 0.5, med)
                assert np.array_equal(ks, want), (memo, npk, ks[:8], want[:8])
                assert int(rec["branch"]) == br
        finally:
            eng.close()


def test_argument_errors(engine):
    x = np.ones(64)
    with pytest.raises(ValueError):
        engine.get_time_delays_phat(x, x, 500.0)                  # int(fs*0.001) == 0 -> find_peaks raises
    with pytest.raises(ValueError):
        engine.get_time_delays_phat(x, x, 48000.0, num_peaks=257)
    with pytest.raises(ValueError):
        engine.gcc_phat_all_pairs(np.ones((1, 64)), 48000.0)      # needs two mics
    with pytest.raises(ValueError):
        engine.filtfilt([1.0, 0.5], [1.0, -0.2], [0.1], np.ones(6))   # shorter than padlen


# ---------------------------------------------------------------- batched table vs oracle (small) and fixtures (full size)
def test_all_pairs_small_batches(engine):
    rng = np.random.default_rng(9)
    for (b, m, length) in ((1, 2, 300), (3, 5, 777), (2, 9, 2048), (1, 7, 5000)):
        frames = rng.standard_normal((b, m, length))
        frames[:, 1:] += 0.7 * frames[:, :1]
        for med in (None, 0.004):
            table, corr = engine.gcc_phat_all_pairs(frames, 16000.0, max_expected_delay=med, want_corr=True)
            for t in range(b):
                want = O.all_pairs(frames[t], 16000.0, max_expected_delay=med)
                for key in ("k_sel", "branch", "k_argmax"):
                    assert np.array_equal(table[t][key], want[key]), (b, m, length, med, key)
                assert np.allclose(table[t]["cmax"], want["cmax"], rtol=1e-10, atol=1e-14)
                assert np.allclose(table[t]["cmin"], want["cmin"], rtol=1e-10, atol=1e-14)
                assert np.allclose(table[t]["snr"], want["snr"], rtol=1e-8)
            p = 0
            for i in range(m):
                for j in range(i + 1, m):
                    assert np.max(np.abs(corr[0, p] - O.phat_correlation(frames[0, i], frames[0, j]))) <= 1e-13
                    p += 1


def test_pair_list_and_bootstrap(engine):
    """Explicit pair lists (pal_gcc_phat_pairs) and the one-vs-many bootstrap built on them (utils.py:183-216)."""
    from pyaudiolocalization_amd import pair_list
    from pyaudiolocalization_amd import utils as U
    rng = np.random.default_rng(17)
    rows = rng.standard_normal((7, 1800))
    rows[1:] += 0.6 * rows[:1]
    full = engine.gcc_phat_all_pairs(rows, 16000.0, max_expected_delay=0.004)
    pick = np.array([0, 5, 6, 11, 20, 3, 3])
    sub = engine.gcc_phat_pairs(rows, pair_list(7)[pick], 16000.0, max_expected_delay=0.004)
    for key in ("k_sel", "branch", "k_argmax"):
        assert np.array_equal(sub[key], full[key][pick]), key
    assert np.allclose(sub["cmax"], full["cmax"][pick], rtol=1e-12)
    with pytest.raises(ValueError):
        engine.gcc_phat_pairs(rows, [[0, 7]], 16000.0)
    a, b = rows[0], rows[1]
    for mode in ("permutation", "block", "circular"):
        np.random.seed(5)
        got = U.bootstrap_significance(a, b, 16000.0, num_bootstrap=40, bootstrap_mode=mode, batch=16)
        np.random.seed(5)
        peaks = []
        for _ in range(40):                      # the reference's loop, with the oracle's PHAT
            if mode == "permutation":
                other = np.random.permutation(b)
            elif mode == "block":
                blocks = [b[i:i + 50] for i in range(0, len(b), 50)]
                np.random.shuffle(blocks)
                other = np.concatenate(blocks)[: len(b)]
            else:
                other = np.roll(b, np.random.randint(0, len(b)))
            peaks.append(np.max(O.phat_correlation(a, other)))
        assert np.isclose(got, np.percentile(peaks, 95), rtol=1e-12), mode
    with pytest.raises(ValueError):
        U.bootstrap_significance(a, b, 16000.0, bootstrap_mode="nope")
    m = U.compute_cross_correlation_metrics(O.phat_correlation(a, b), a, b, 16000.0)
    assert set(m) == {"peak_to_peak_ratio", "snr", "significant"} and bool(m["significant"]) is True


def test_pair_group_size_rule(engine):
    """Launch groups of the pair pipeline: 240 packed transforms where a workspace slot stays under 1 GiB, at least 32;
    pal_set_chunk (or PAL_CHUNK) fixes the size."""
    from pyaudiolocalization_amd import Engine
    fresh = Engine(engine.device)
    try:
        assert fresh.pair_group_size(44100) == 240
        assert fresh.pair_group_size(200000) == (1 << 30) // (16 * 399999)
        assert fresh.pair_group_size(1 << 20) == 32
        fresh.set_chunk(7)
        assert fresh.pair_group_size(44100) == 7
    finally:
        fresh.close()


def test_chunk_size_does_not_change_results(engine):
    frames = np.random.default_rng(4).standard_normal((2, 6, 1500))
    engine.set_chunk(32)
    a = engine.gcc_phat_all_pairs(frames, 16000.0, max_expected_delay=0.003)
    engine.set_chunk(3)
    b = engine.gcc_phat_all_pairs(frames, 16000.0, max_expected_delay=0.003)
    engine.set_chunk(32)
    assert a.tobytes() == b.tobytes()


def test_metric_frames_full_table(engine, golden):
    """BASELINE metric workload at full size: 64 mics x 44100 samples, 2016 pairs; fixture rows for the
    first 24 mics (276 pairs: the sample bench.py checks in every run) come from the reference; idempotence and
    pair-order properties cover the rest."""
    g = golden("metric_44k1_first24.npz")
    frames = cases.metric_frames(1, 64)[0]
    first8 = np.array([k for k, (i, j) in enumerate((i, j) for i in range(64) for j in range(i + 1, 64)) if j < 24])
    for med in (0.05, None):
        t64 = IMPL.pair_table(frames, 44100, med)
        t8 = IMPL.pair_table(frames[:24], 44100, med)
        check_table(t8, g, tag_of(med))
        for key in ("k_sel", "branch", "k_argmax"):                       # a pair's row does not depend on the batch around it
            assert np.array_equal(t64[key][first8], t8[key]), key
        for key in ("cmax", "cmin", "snr"):                               # (its transform partner changes: last-bit differences)
            assert np.allclose(t64[key][first8], t8[key], rtol=1e-11, atol=1e-15), key
        again = IMPL.pair_table(frames, 44100, med)
        assert all(np.array_equal(again[k], t64[k]) for k in t64)              # bitwise reproducible
    # unwindowed: the injected common component puts the peak at the integer delay difference (circular index)
    delays = np.random.default_rng(8).integers(-60, 60, size=64)
    t = IMPL.pair_table(frames, 44100, None)
    p = 0
    hits = 0
    for i in range(64):
        for j in range(i + 1, 64):
            hits += int(t["k_argmax"][p]) == int((delays[j] - delays[i]) % 88199)
            p += 1
    assert hits >= 2000, hits


# ---------------------------------------------------------------- BASELINE configs, stage by stage (teacher-forced)
def test_c1_example1(golden):
    # Example 1 places the source equidistant from all mics: the four signals are identical, every
    # PHAT sequence is exactly symmetric (corr[k] == corr[n-k]) and the reference's choice between the mirror
    # peaks 44098 / 44101 is decided by the last bit of its FFT.  Ties are checked to 1e-12 and reported.
    stages.run_chain(IMPL, golden("c1_example1.npz"), "", *stages.c1_case(), (0.05, None), allow_ties=True)


def test_c2_chirp8(golden):
    g = golden("c2_chirp8.npz")
    for tag, low in (("a_", False), ("b_", True)):
        stages.run_chain(IMPL, g, tag, *stages.c2_case(low), (0.05, None))


def test_c3_grid64_full_trial(golden):
    stages.run_chain(IMPL, golden("c3_grid64_trial0.npz"), "", *stages.c3_case(0), (0.05, None))


def test_c4_sphere_first12(golden):
    g = golden("c4_sphere_first12.npz")
    frames = cases.c4_frames(12)
    digest_close(frames, g["frames_digest"], 1e-12)
    for med in (0.05, None):
        check_table(IMPL.pair_table(frames, 96000, med), g, tag_of(med))


def test_c5_stream_frames(golden):
    g = golden("c5_stream_frames01.npz")
    for f in (0, 1):
        stages.run_chain(IMPL, g, f"f{f}_", *stages.c5_case(f), (0.05,))


# ---------------------------------------------------------------- second path: simulation, filters, sync
def test_fractional_delay_and_compression(engine, golden):
    g = golden("filters.npz")
    x = np.random.default_rng(21).standard_normal(4000)
    assert np.max(np.abs(engine.fractional_delay(x, 0.00123, 48000) - g["fracdelay"])) <= 1e-12
    assert np.max(np.abs(engine.normalize_compress(x) - g["compress"])) <= 1e-14
    assert np.max(np.abs(engine.normalize_compress(x, normalize_only=True) - O.normalize_signal(x))) == 0
    assert np.array_equal(engine.normalize_compress(np.zeros(10)), np.zeros(10))
    with pytest.raises(ValueError):
        engine.fractional_delay(np.ones(50), 0.001, 8000)            # N < 100 breaks the reference's fade slice too
    rows = np.random.default_rng(2).standard_normal((5, 1501))
    d = np.array([0.0, 1e-4, 0.0031, 0.02, 0.0007])
    got = engine.fractional_delay(rows, d, 16000.0)
    for r in range(5):
        assert np.max(np.abs(got[r] - O.fractional_delay(rows[r], d[r], 16000.0))) <= 1e-12


def test_prefilters(engine, golden):
    g = golden("filters.npz")
    from pyaudiolocalization_amd.signal_processing import noise_reduction
    x = np.random.default_rng(21).standard_normal(4000)
    for fs in (44100, 48000, 96000):
        assert np.array_equal(noise_reduction(x, fs), g[f"butter_{fs}"]), fs       # same operation order: bit-identical
    assert np.max(np.abs(noise_reduction(x, 48000, "fir") - g["fir_48000"])) <= 1e-13
    assert np.max(np.abs(noise_reduction(x, 48000, "wiener") - g["wiener"])) <= 1e-13
    with pytest.raises(ValueError):
        noise_reduction(x, 48000, "nope")


def test_image_sources_and_simulation_dropin(golden):
    from pyaudiolocalization_amd import main as M
    from pyaudiolocalization_amd import utils as U
    g = golden("c2_chirp8.npz")
    cfg = cases.c2_config()
    mics = np.array(cfg["mic_positions"])
    imgs = U.generate_image_sources_iterative(cfg["source_position"], cfg["reflective_planes"], 3, 500, cases.LOW_LOSS, mics, 0.01)
    assert np.array_equal(np.array([i["source"] for i in imgs]), g["b_images"])
    assert [i["material"] for i in imgs] == list(g["b_image_materials"])
    sig = M.simulate_signals_with_multipath(cfg["source_position"], mics, 48000, cases.C_SOUND, 1.0, "chirp", 500,
                                            cfg["reflective_planes"], cases.LOW_LOSS, 3, 0.01)
    digest_close(sig, g["b_sim_digest"], 1e-11)


def test_batched_simulation_and_filter(engine):
    """B trials in one call (C3 / C5 style): every trial has its own base signal and path table; gains spanning
    60 decades inside one complex transform must not leak between the two mics packed together."""
    rng = np.random.default_rng(31)
    b, m, k, nbase, total, trim = 3, 5, 4, 1500, 1700, 1500
    base = rng.standard_normal((b, nbase))
    delays = rng.uniform(0.0, 0.02, (b, m, k))
    gains = 10.0 ** rng.uniform(-60, 0, (b, m, k))
    got = engine.simulate_multipath(base, 8000.0, total, delays, gains, trim)
    assert got.shape == (b, m, trim)
    for t in range(b):
        want = O.simulate_from_base(base[t], delays[t], gains[t], 8000.0, total, trim)
        assert np.max(np.abs(got[t] - want)) <= 1e-11, (t, float(np.max(np.abs(got[t] - want))))
    rows = rng.standard_normal((b * m, 2500))
    from pyaudiolocalization_amd.signal_processing import noise_reduction_rows
    filt = noise_reduction_rows(rows, 48000.0)
    assert all(np.array_equal(filt[r], O.noise_reduction(rows[r], 48000.0)) for r in range(b * m))


def test_synchronise_with_real_shifts(engine):
    from pyaudiolocalization_amd.utils import synchronize_signals_improved
    rng = np.random.default_rng(12)
    y = rng.standard_normal(3000)
    sig = [np.roll(y, k) + 0.05 * rng.standard_normal(3000) for k in (0, 3, -5, 11, 400)]
    got = synchronize_signals_improved(sig, 8000)
    want = O.synchronize_signals(sig, 8000)
    assert len(got) == len(want) and all(np.array_equal(a, b) for a, b in zip(got, want))


def test_localize_sound_source_position(golden, tmp_path, monkeypatch):
    """End to end through the drop-in: the estimated position is checked against the reference's fixture
    with the reference's own stage inputs substituted where the reference is ill-conditioned
    (tests/stages.py docstring): the TDOA table from the engine feeds the unchanged host solve."""
    from pyaudiolocalization_amd import main as M
    g = golden("c2_chirp8.npz")
    monkeypatch.chdir(tmp_path)
    cfg = cases.c2_config()
    base, delays, gains, fs, total, trim = stages.c2_case(False)
    oracle = stages.OracleImpl()
    filt = oracle.prefilter(oracle.synchronize(oracle.simulate(base, delays, gains, fs, total, trim), fs), fs)
    monkeypatch.setattr(M, "simulate_signals_with_multipath", lambda **kw: [r for r in filt])
    monkeypatch.setattr(M, "synchronize_signals_improved", lambda s, fs_: s)
    monkeypatch.setattr(M, "noise_reduction_rows", lambda rows, fs_, method="butterworth": np.asarray(rows))
    res = M.localize_sound_source(cfg, use_simulation=True, show_plots=False)
    assert np.max(np.abs(res["estimated_position"] - g["a_position"])) <= 1e-3      # metres, north_star tolerance
    assert res["correlation_metrics"] is None and res["correlation_matrix"] is None
    # and the untouched end-to-end path runs and returns the documented result dict
    monkeypatch.undo()
    monkeypatch.chdir(tmp_path)
    cfg1 = cases.c1_config()
    cfg1["localization"]["visualize_correlation"] = True
    out = M.localize_sound_source(cfg1, use_simulation=True, show_plots=False)
    assert set(out) == {"estimated_position", "actual_position", "mic_positions", "correlation_metrics", "correlation_matrix",
                        "calibration_data"}
    assert out["estimated_position"].shape == (3,) and out["correlation_matrix"].shape == (4, 4)
    assert (tmp_path / "localization_result.png").exists() and (tmp_path / "heatmap.png").exists()
    with pytest.raises(KeyError):
        M.localize_sound_source({"fs": 48000}, show_plots=False)


def test_real_audio_ingest_without_soundfile(engine, tmp_path):
    """read_audio_files (utils.py:459-482) with the standard-library WAV decoder that stands in for the
    absent soundfile package: PCM scaling x / 2^(bits-1), mono mix, normalise + compress like the oracle; a missing file
    and an undecodable one raise what the reference raises."""
    import wave
    from pyaudiolocalization_amd import utils as U
    rng = np.random.default_rng(12)
    fs, n = 16000, 4000
    paths, want = [], []
    for k, (width, channels) in enumerate(((2, 1), (2, 2), (3, 1), (1, 1))):
        x = rng.uniform(-0.9, 0.9, (n, channels))
        if width == 2:
            q = np.round(x * 32767).astype("<i2")
            raw, dec = q.tobytes(), q.astype(np.float64) / 32768.0
        elif width == 3:
            q = np.round(x * (2 ** 23 - 1)).astype(np.int32)
            b = (q & 0xFFFFFF).astype(np.uint32)
            raw = np.stack([(b & 255), (b >> 8) & 255, (b >> 16) & 255], axis=-1).astype(np.uint8).tobytes()
            dec = q.astype(np.float64) / float(2 ** 23)
        else:
            q = np.round(x * 127 + 128).astype(np.uint8)
            raw, dec = q.tobytes(), (q.astype(np.float64) - 128.0) / 128.0
        path = tmp_path / f"mic{k}.wav"
        with wave.open(str(path), "wb") as w:
            w.setnchannels(channels); w.setsampwidth(width); w.setframerate(fs); w.writeframes(raw)
        paths.append(str(path))
        mono = dec.mean(axis=1) if channels > 1 else dec[:, 0]
        want.append(O.dynamic_range_compression(O.normalize_signal(mono)))
    got = U.read_audio_files(paths, fs)
    for g, w_ in zip(got, want):
        assert g.shape == w_.shape and np.max(np.abs(g - w_)) <= 1e-15
    with pytest.raises(FileNotFoundError):
        U.read_audio_files([str(tmp_path / "absent.wav")], fs)
    bad = tmp_path / "bad.wav"
    bad.write_bytes(b"not a wave file")
    with pytest.raises(RuntimeError):
        U.read_audio_files([str(bad)], fs)


def test_localize_from_audio_files(engine, tmp_path):
    """main.py:176-186: the real-recordings branch (use_simulation=False) end to end on WAV files of one noise burst with
    geometric delays: the result dict has the reference's keys and a finite position inside the solver's bounds.  (No
    accuracy claim: like the reference, the pipeline synchronises the recordings first, utils.py:407-457, which removes
    delays below 50 ms - the stage-by-stage tests pin every stage against the oracle instead.)"""
    import wave
    import pyaudiolocalization_amd.main as M
    fs, n = 16000, 8000
    mics = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [0, 0, 1], [1, 1, 0.5]], dtype=float)
    src = np.array([0.4, 0.7, 0.3])
    c = M.speed_of_sound(20.0, 50.0)
    rng = np.random.default_rng(33)
    base = rng.standard_normal(n + 400)
    paths = []
    for k, mpos in enumerate(mics):
        d = int(round(np.linalg.norm(src - mpos) / c * fs))
        x = base[200 - d: 200 - d + n] * 0.2
        q = np.round(x / np.max(np.abs(x)) * 30000).astype("<i2")
        path = tmp_path / f"m{k}.wav"
        with wave.open(str(path), "wb") as w:
            w.setnchannels(1); w.setsampwidth(2); w.setframerate(fs); w.writeframes(q.tobytes())
        paths.append(str(path))
    cfg = dict(M.config)
    cfg.update({"fs": fs, "mic_positions": mics.tolist(), "source_position": None, "analyze_correlation": False,
                "use_3d_plot": False, "max_expected_delay": 0.01})
    res = M.localize_sound_source(cfg, audio_files=paths, use_simulation=False, show_plots=False)
    assert set(res) >= {"estimated_position", "actual_position", "mic_positions", "correlation_metrics", "correlation_matrix",
                        "calibration_data"}
    assert res["actual_position"] is None
    pos = np.asarray(res["estimated_position"], dtype=float)
    assert pos.shape == (3,) and np.all(np.isfinite(pos)) and np.all(np.abs(pos) < 10.0)
    assert res["correlation_matrix"].shape == (5, 5)


def test_profile_counters_and_plan(engine):
    info = engine.plan_info(44100)
    assert info["n"] == 88199 and info["conv_len"] in (180224, 196608, 262144) and info["m1"] * info["m2"] == info["conv_len"]
    frames = np.random.default_rng(0).standard_normal((1, 4, 2000))
    engine.profile_begin()
    engine.gcc_phat_all_pairs(frames, 16000.0)
    engine.profile_end()
    ent = engine.profile_entries()
    assert any(k.startswith("k_rows<") and v[1] > 0 for k, v in ent.items()), ent
    assert ent["k_peak_stream"][1] >= 1 and ent["k_peak_finish"][1] >= 1


def test_rccl_single_rank_gather(engine):
    """One-rank communicator: the all-gather is a device copy; exercises the RCCL binding itself."""
    from pyaudiolocalization_amd import RECORD
    table = np.zeros(10, dtype=RECORD)
    table["k_sel"] = np.arange(10)
    d_a, d_b = engine.alloc(table.nbytes), engine.alloc(table.nbytes)
    engine.upload(d_a, table)
    engine.comm_init(1, 0, engine.comm_unique_id())
    engine.all_gather_dev(d_a, d_b, table.nbytes)
    engine.synchronize()
    back = np.zeros(10, dtype=RECORD)
    engine.download(back, d_b)
    engine.comm_destroy()
    engine.free(d_a)
    engine.free(d_b)
    assert np.array_equal(back["k_sel"], np.arange(10))


def test_calibration_dropin(golden):
    """calibration.py drop-in on the engine: batched fractional delays + batched correlations against
    the calibration signal reproduce the reference's run_calibration (same noise draws): lags exact, amplitudes and
    waveforms to fp64 rounding of the exact-length transforms."""
    from pyaudiolocalization_amd import calibration as cal_mod
    g = golden("calibration.npz")
    for tag, cfg, seed in cases.calibration_cases():
        results, calib, recs = cal_mod.run_calibration(cfg, noise=cases.calibration_noise(cfg, seed))
        digest_close([calib], g[f"{tag}_calib_digest"][None], 1e-12)
        digest_close(recs, g[f"{tag}_rec_digest"], 1e-11)
        assert np.array_equal(np.array([r["delay"] for r in results]), g[f"{tag}_delay"])      # integer lag / fs: exact
        assert np.allclose([r["amplitude"] for r in results], g[f"{tag}_amplitude"], rtol=1e-10, atol=0)
        assert all(isinstance(r["delay"], np.floating) for r in results)
    with pytest.raises(ValueError):
        cal_mod.generate_calibration_signal(8000, 0.1, signal_type="noise")


def test_small_and_odd_lengths_both_routes(engine):
    """Every frame length 1..24 and a spread of odd / power-of-two neighbours: n = 2L-1 walks through one-tile
    prime-factor plans (n < 1024, n prime or not), multi-row plans (2045 = 5 x 409, 2049 = 3 x 683) and lengths that
    stay on the four-step route (n = 1, 2047); unequal lengths make n even (four-step)."""
    rng = np.random.default_rng(3)
    for length in list(range(1, 25)) + [45, 64, 100, 255, 256, 257, 496, 512, 1023, 1024, 1025]:
        a, b = rng.standard_normal(length), rng.standard_normal(length)
        assert np.max(np.abs(engine.phat_correlation(a, b) - O.phat_correlation(a, b))) < 1e-13, length
    for n1, n2 in ((3, 7), (10, 4), (33, 32), (100, 41), (500, 496)):
        a, b = rng.standard_normal(n1), rng.standard_normal(n2)
        assert np.max(np.abs(engine.phat_correlation(a, b) - O.phat_correlation(a, b))) < 1e-13, (n1, n2)
    for length in (8, 16, 33):
        frames = rng.standard_normal((2, 4, length))
        table = engine.gcc_phat_all_pairs(frames, 8000.0)
        for t in range(2):
            want = O.all_pairs(frames[t], 8000.0)
            assert np.array_equal(table[t]["k_sel"], want["k_sel"]) and np.array_equal(table[t]["branch"], want["branch"])
