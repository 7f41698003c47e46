"""Round-2 engine surface on the GPU: the device-resident stage chain (pyaudiolocalization_amd.stream) against the staged
host path, explicit pair lists in HBM (pal_gcc_phat_pairs_dev), the N > 1 shard + gather paths with the ENGINE as the
per-rank compute (two processes on one GPU, gloo gather), input checks (non-finite samples), the bounded plan cache."""
import os
import socket

import numpy as np
import pytest

from oracle import cases
from oracle import pal_oracle as O

import stages

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def _engine(engine):
    import pyaudiolocalization_amd.engine as E
    E._default = engine
    yield
    E._default = None


def _c5_like_frames(count, mics=8, nbase=3000, fs=48000.0):
    """Path tables of `count` frames of a C5-shaped stream (3 planes, low-loss table, random-walk source) on a small
    array: the simulated lengths and the synchronised lengths differ between frames, like in the real configuration."""
    rng = np.random.default_rng(55)
    pos = rng.uniform(-0.4, 0.4, (mics, 3))
    src = np.array([1.0, 2.0, 0.5])
    bases, delays, gains, totals = [], [], [], []
    duration = nbase / fs
    for f in range(count):
        src = src + rng.normal(0.0, 0.05, 3)
        d, g, longest, _ = O.multipath_paths(src, pos, cases.C_SOUND, 1000, cases.DEFAULT_PLANES, cases.LOW_LOSS, 3, 0.01)
        bases.append(np.random.default_rng(2000 + f).standard_normal(nbase))
        delays.append(d)
        gains.append(g)
        totals.append(int((duration + longest) * fs))
    return bases, delays, gains, totals, nbase, fs


def test_stream_chain_equals_staged_host_path(engine):
    """simulate -> synchronise -> prefilter -> pairs with the waveforms resident in HBM gives the tables of the staged
    drop-in calls (host arrays between the stages), bit for bit, frame by frame, for frames of different lengths."""
    from pyaudiolocalization_amd.main import tdoa_table
    from pyaudiolocalization_amd.signal_processing import noise_reduction_rows
    from pyaudiolocalization_amd.stream import tdoa_stream
    from pyaudiolocalization_amd.utils import synchronize_signals_improved
    bases, delays, gains, totals, trim, fs = _c5_like_frames(7)
    tables, lengths = tdoa_stream(bases, delays, gains, fs, totals, trim, "butterworth", 0.05, engine=engine, frames_per_batch=4)
    assert tables.shape == (7, 28)
    seen = set()
    for f in range(7):
        sim = engine.simulate_multipath(bases[f][None], fs, totals[f], delays[f][None], gains[f][None], trim)[0]
        synced = np.array(synchronize_signals_improved(list(sim), fs))
        filt = noise_reduction_rows(synced, fs, "butterworth")
        want = tdoa_table(filt, fs, 0.05)
        assert lengths[f] == filt.shape[1]
        assert tables[f].tobytes() == want.tobytes(), f
        seen.add((totals[f], int(lengths[f])))
    assert len(seen) > 1                                                       # the case does exercise the grouping
    # and against the oracle's chain on one frame (teacher-forced per stage in test_gpu_parity; here end to end on
    # device data: the simulate stage differs from the oracle by <= 1e-11, which the chaotic prefilter may amplify -
    # the synchronised LENGTH and the pads are integers and must agree)
    f = 0
    sim_o = O.simulate_literal(bases[f], delays[f], gains[f], fs, totals[f], trim)
    assert np.array(O.synchronize_signals(list(sim_o), fs)).shape[1] == lengths[f]
    # Wiener prefilter through the same chain
    tw, lw = tdoa_stream(bases[:2], delays[:2], gains[:2], fs, totals[:2], trim, "wiener", None, engine=engine)
    sim = engine.simulate_multipath(bases[1][None], fs, totals[1], delays[1][None], gains[1][None], trim)[0]
    want = tdoa_table(noise_reduction_rows(np.array(synchronize_signals_improved(list(sim), fs)), fs, "wiener"), fs, None)
    assert tw[1].tobytes() == want.tobytes()
    with pytest.raises(ValueError):
        tdoa_stream(bases[:1], delays[:1], gains[:1], fs, totals[:1], trim, "median-filter", engine=engine)


def test_pairs_dev_equals_host_pairs_and_all_pairs(engine):
    """pal_gcc_phat_pairs_dev: rows, pair list and table in HBM.  A contiguous block of the ordered pair list gives that
    block of the all-pairs table (what a rank computes when one large frame is split over the GPUs)."""
    from pyaudiolocalization_amd import RECORD, make_params, pair_list
    from pyaudiolocalization_amd.distributed import shard_pairs
    rng = np.random.default_rng(61)
    rows = rng.standard_normal((9, 2500))
    rows[4] = 0.0                                                              # a silent microphone inside the block
    full = engine.gcc_phat_all_pairs(rows, 16000.0, max_expected_delay=0.004)
    plist = pair_list(9)
    prm = make_params(16000.0, 1, "median", 1.0, 0.004)
    d_rows = engine.alloc(rows.nbytes)
    engine.upload(d_rows, rows)
    try:
        pieces = []
        for rank in range(3):
            lo, hi = shard_pairs(9, rank, 3)
            block = np.ascontiguousarray(plist[lo:hi])
            d_pairs, d_tab = engine.alloc(block.nbytes), engine.alloc((hi - lo) * RECORD.itemsize)
            engine.upload(d_pairs, block)
            engine.gcc_phat_pairs_dev(d_rows, 9, 2500, d_pairs, hi - lo, prm, d_tab)
            engine.synchronize()
            got = np.zeros(hi - lo, dtype=RECORD)
            engine.download(got, d_tab)
            engine.free(d_pairs); engine.free(d_tab)
            host = engine.gcc_phat_pairs(rows, block, 16000.0, max_expected_delay=0.004)
            assert got.tobytes() == host.tobytes()
            pieces.append(got)
        got = np.concatenate(pieces)
        for key in ("k_sel", "branch", "k_argmax"):
            assert np.array_equal(got[key], full[key]), key
        for key in ("cmax", "cmin", "snr"):                                    # (the transform partner differs at block borders)
            assert np.allclose(got[key], full[key], rtol=1e-11, atol=1e-15, equal_nan=True), key
        bad = np.array([[0, 1], [2, 9]], dtype=np.int32)                       # row 9 does not exist
        d_pairs, d_tab = engine.alloc(bad.nbytes), engine.alloc(2 * RECORD.itemsize)
        engine.upload(d_pairs, bad)
        engine.gcc_phat_pairs_dev(d_rows, 9, 2500, d_pairs, 2, prm, d_tab)
        with pytest.raises(ValueError):
            engine.synchronize()
        engine.synchronize()                                                   # the status is reported once
        engine.free(d_pairs); engine.free(d_tab)
    finally:
        engine.free(d_rows)


def test_non_finite_samples_are_rejected(engine):
    """A NaN would poison the pair packed into the same complex transform (the reference confines it to the microphone's
    own pairs): the batched calls fail with ValueError (PAL_ERR_INVALID) instead of returning different rows."""
    rng = np.random.default_rng(62)
    rows = rng.standard_normal((5, 1200))
    rows[2, 77] = np.nan
    with pytest.raises(ValueError):
        engine.gcc_phat_all_pairs(rows, 8000.0)
    rows[2, 77] = np.inf
    with pytest.raises(ValueError):
        engine.gcc_phat_pairs(rows, [[0, 1], [2, 3]], 8000.0)
    rows[2, 77] = 0.5
    t = engine.gcc_phat_all_pairs(rows, 8000.0)                                 # the engine is usable afterwards
    want = O.all_pairs(rows, 8000.0)
    assert np.array_equal(t["k_sel"], want["k_sel"])
    # single-pair calls have no partner pair: NaN in, NaN out like numpy
    a = rng.standard_normal(300)
    b = a.copy()
    b[5] = np.nan
    assert np.all(np.isnan(engine.phat_correlation(a, b)))


def test_plan_cache_is_bounded(monkeypatch):
    """PAL_MAX_PLANS bounds the per-length plans: many distinct lengths in turn keep giving right answers
    while old plans are evicted; pal_clear_plans drops the rest."""
    from pyaudiolocalization_amd import Engine
    monkeypatch.setenv("PAL_MAX_PLANS", "3")
    eng = Engine(0)
    try:
        rng = np.random.default_rng(63)
        for n in (300, 301, 302, 303, 304, 300, 305, 301):
            a, b = rng.standard_normal(n), rng.standard_normal(n)
            want = O.phat_correlation(a, b)
            assert np.max(np.abs(eng.phat_correlation(a, b) - want)) <= 5e-14
        rows = [rng.standard_normal(n) for n in (800, 790, 805, 797)]          # unequal-length sync: one cached convolution
        for _ in range(2):
            from pyaudiolocalization_amd.utils import synchronize_signals_improved
            import pyaudiolocalization_amd.engine as E
            keep = E._default
            E._default = eng
            try:
                out = synchronize_signals_improved(rows, 8000)
            finally:
                E._default = keep
            assert len({len(r) for r in out}) == 1
        eng.clear_plans()
        a, b = rng.standard_normal(300), rng.standard_normal(300)
        assert np.max(np.abs(eng.phat_correlation(a, b) - O.phat_correlation(a, b))) <= 5e-14
    finally:
        eng.close()


def test_xcorr_row_counts_multiple_of_four(engine):
    """the result block of pal_xcorr_vs_ref was 4 bytes short when R % 4 == 0 (the default 4-microphone
    configuration).  Sizes around the alignment edge, values against the oracle."""
    rng = np.random.default_rng(64)
    for r in (3, 4, 5, 8, 12):
        rows = rng.standard_normal((r, 700))
        kpk, win, pk, ref = engine.xcorr_vs_ref(rows, r - 1)
        for i in range(r):
            cc = O.xcorr_full(rows[i], rows[r - 1])
            assert int(kpk[i]) == int(np.argmax(np.abs(cc)))
            assert abs(pk[i] - np.max(np.abs(cc))) <= 1e-10 * max(1.0, np.max(np.abs(cc)))
        assert abs(ref - pk[r - 1]) == 0


# ---------------------------------------------------------------- two ranks, engine compute, gloo gather
def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _rank_frames(first, count):
    return np.stack([np.random.default_rng([13, first + k]).standard_normal((6, 1500)) for k in range(count)])


def _rank_worker(rank, world, port, mode, queue):
    import torch.distributed as dist
    from pyaudiolocalization_amd import Engine
    from pyaudiolocalization_amd.distributed import sharded_pair_table, sharded_tdoa
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    eng = Engine(0)                                           # both ranks on the box's one GPU: the engine is the compute
    try:
        if mode == "frames":
            full = sharded_tdoa(_rank_frames, 5, rank, world, lambda fr: eng.gcc_phat_all_pairs(fr, 16000.0, max_expected_delay=0.003))
        else:
            frame = _rank_frames(0, 1)[0]
            full = sharded_pair_table(frame, rank, world, lambda fr, pr: eng.gcc_phat_pairs(fr, pr, 16000.0, max_expected_delay=0.003))
        queue.put((rank, full.tobytes()))
    finally:
        eng.close()
        dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["frames", "pairs"])
def test_two_ranks_engine_compute_gloo_gather(engine, mode):
    """The N > 1 path with the HIP engine as the per-rank compute: two processes (one engine each, both on this box's
    GPU), block partition of the frames - or of one frame's pair list - and one gather; every rank ends with the
    single-process table.  (RCCL itself needs one device per rank: its multi-rank gather runs on the driver's 8-GPU node.)"""
    import multiprocessing as mp
    import conftest
    conftest.require_forkserver()
    ctx = mp.get_context("forkserver")          # started in conftest.pytest_sessionstart, before this process touched the GPU
    queue = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rank_worker, args=(r, 2, port, mode, queue)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(queue.get(timeout=300) for _ in range(2))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    if mode == "frames":
        want = engine.gcc_phat_all_pairs(_rank_frames(0, 5), 16000.0, max_expected_delay=0.003)
        ref = np.frombuffer(got[0], dtype=want.dtype).reshape(want.shape)
    else:
        want = engine.gcc_phat_all_pairs(_rank_frames(0, 1)[0], 16000.0, max_expected_delay=0.003)
        ref = np.frombuffer(got[0], dtype=want.dtype)
    assert got[0] == got[1]
    for key in ("k_sel", "branch", "k_argmax", "n_sel"):
        assert np.array_equal(ref[key], want[key]), key
    for key in ("cmax", "cmin", "snr", "sel_height"):                          # (block borders change a pair's transform partner)
        assert np.allclose(ref[key], want[key], rtol=1e-11, atol=1e-15), key


def test_sync_reference_microphone_with_equal_energies(engine):
    """utils.py:413-414: the reference microphone is np.argmax of np.sum(sig**2).  Mirrored microphones give rows of equal or
    nearly equal energy; the device's summation order must not decide then - near-ties are settled with numpy's own sum."""
    rng = np.random.default_rng(41)
    b, m, n = 3, 8, 6000
    rows = rng.standard_normal((b, m, n)) * 0.5
    rows[0, 3] = rng.standard_normal(n) * 2.0                 # frame 0: rows 1, 3, 5 share one energy exactly (1 = -3, 5 = 3): the first wins
    rows[0, 1] = -rows[0, 3]
    rows[0, 5] = rows[0, 3]
    rows[1, 6] = rng.standard_normal(n) * 2.0                 # frame 1: rows 2 and 6 hold the same samples in another order: equal up to rounding
    rows[1, 2] = rows[1, 6][rng.permutation(n)]
    rows[2, 4] = rng.standard_normal(n) * 2.0                 # frame 2: a clear winner
    want = [int(np.argmax([np.sum(r ** 2) for r in rows[f]])) for f in range(b)]
    d = engine.alloc(rows.nbytes)
    try:
        engine.upload(d, np.ascontiguousarray(rows))
        ref, kpk, win, pk, refpk = engine.sync_measure_dev(d, b, m, n)
    finally:
        engine.free(d)
    assert ref.tolist() == want, (ref.tolist(), want)
    assert want[0] == 1 and want[2] == 4
