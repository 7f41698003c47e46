"""BASELINE configurations at their FULL sizes on the HIP engine, checked through size-independent properties and the
reference fixtures that cover part of them: C4 (256 microphones, 96 kHz x 1 s, 32 640 pairs in one call), a 16-trial C3
batch (64 microphones, 2016 pairs per trial) and a 64-frame slice of the C5 stream."""
import numpy as np
import pytest

from oracle import cases

import stages
from stages import check_table, tag_of

pytestmark = pytest.mark.gpu


def _sub_table_rows(mics, first):
    """Rows of the row-major i<j table of `mics` microphones that belong to the first `first` microphones."""
    return np.array([k for k, (i, j) in enumerate((i, j) for i in range(mics) for j in range(i + 1, mics)) if j < first])


def test_c4_all_256_microphones(engine, golden):
    """32 640 pairs at n = 191 999 (prime: the four-step chirp convolution carries the inverse).  The rows of the first
    twelve microphones equal the reference's fixture; a pair's row does not depend on the batch around it; the whole
    table is bitwise reproducible."""
    g = golden("c4_sphere_first12.npz")
    frames = cases.c4_frames(256)
    rows12 = _sub_table_rows(256, 12)
    assert rows12.size == 66
    for med in (0.05, None):
        t = engine.gcc_phat_all_pairs(frames, 96000, max_expected_delay=med)
        assert t.shape == (32640,)
        sub = {k: t[k][rows12] for k in ("k_sel", "branch", "k_argmax", "cmax", "cmin", "snr")}
        check_table(sub, g, tag_of(med))
        if med is not None:
            again = engine.gcc_phat_all_pairs(frames, 96000, max_expected_delay=med)
            assert again.tobytes() == t.tobytes()
        # the 12-microphone call on its own (other transform partners) selects the same indices
        t12 = engine.gcc_phat_all_pairs(frames[:12], 96000, max_expected_delay=med)
        for key in ("k_sel", "branch", "k_argmax"):
            assert np.array_equal(t12[key], t[key][rows12]), key
        # table symmetry property: the lag of (i, j) found without a window is a circular index; all are valid indices
        assert t["k_sel"].min() >= 0 and t["k_sel"].max() < 191999


def test_c3_sixteen_trial_batch(engine, golden):
    """16 trials x 2016 pairs in ONE engine call: trial 0 holds the prefiltered rows of the reference's trial 0 (oracle
    chain, pinned by the fixture's digests), the other trials hold noise of the same length.  Trial 0's rows of the
    batched table equal the reference's fixture and the single-trial call bit for bit."""
    g = golden("c3_grid64_trial0.npz")
    base, delays, gains, fs, total, trim = stages.c3_case(0)
    o = stages.OracleImpl()
    filt = o.prefilter(o.synchronize(o.simulate(base, delays, gains, fs, total, trim), fs), fs)
    stages.digest_close(filt, g["filt_digest"], 1e-13)
    length = filt.shape[1]
    batch = np.empty((16, 64, length))
    batch[0] = filt
    rng = np.random.default_rng(303)
    offsets = {}
    for t in range(1, 16):
        common = rng.standard_normal(length + 64)
        offs = offsets[t] = rng.integers(0, 64, 64)
        batch[t] = np.stack([common[o_: o_ + length] for o_ in offs]) + 0.3 * rng.standard_normal((64, length))
    for med in (0.05, None):
        tb = engine.gcc_phat_all_pairs(batch, fs, max_expected_delay=med)
        assert tb.shape == (16, 2016)
        t0 = engine.gcc_phat_all_pairs(filt, fs, max_expected_delay=med)
        check_table({k: tb[0][k] for k in ("k_sel", "branch", "k_argmax", "cmax", "cmin", "snr")}, g, tag_of(med))
        for key in ("k_sel", "branch", "k_argmax"):
            assert np.array_equal(tb[0][key], t0[key]), key
        for key in ("cmax", "cmin", "snr"):
            assert np.allclose(tb[0][key], t0[key], rtol=1e-11, atol=1e-15), key
        if med is None:                           # unwindowed: the common component puts the maximum at the offset difference
            n = 2 * length - 1
            i, j = np.triu_indices(64, k=1)
            for t in (1, 7, 15):
                want = (offsets[t][j] - offsets[t][i]) % n
                assert np.count_nonzero(tb[t]["k_argmax"] == want) >= 2000, t
    again = engine.gcc_phat_all_pairs(batch, fs, max_expected_delay=0.05)
    assert again.tobytes() == engine.gcc_phat_all_pairs(batch, fs, max_expected_delay=0.05).tobytes()


def test_c5_stream_slice(engine, golden):
    """64 frames of the streaming configuration (64 microphones x 12 000 samples, n = 23 999 = 103 x 233) in one call:
    frames 0 and 1 are the reference's prefiltered rows (fixture), the rest synthetic; rows of a frame do not depend on
    the frames batched with it."""
    g = golden("c5_stream_frames01.npz")
    o = stages.OracleImpl()
    rows, lens = [], []
    for f in (0, 1):
        base, delays, gains, fs, total, trim = stages.c5_case(f)
        filt = o.prefilter(o.synchronize(o.simulate(base, delays, gains, fs, total, trim), fs), fs)
        rows.append(filt)
        lens.append(filt.shape[1])
    rng = np.random.default_rng(505)
    for f, filt in enumerate(rows):               # frames of a batch share one length: each reference frame gets its own batch
        length = lens[f]
        batch = rng.standard_normal((64, 64, length))
        batch[5] = filt
        tb = engine.gcc_phat_all_pairs(batch, 48000, max_expected_delay=0.05)
        single = engine.gcc_phat_all_pairs(filt, 48000, max_expected_delay=0.05)
        pre = f"f{f}_"
        gg = {k[len(pre):]: v for k, v in g.items() if k.startswith(pre)}
        check_table({k: tb[5][k] for k in ("k_sel", "branch", "k_argmax", "cmax", "cmin", "snr")}, gg, "0p05")
        for key in ("k_sel", "branch", "k_argmax"):
            assert np.array_equal(tb[5][key], single[key]), key
