"""Sharding + single gather of the TDOA table over 2 (and 3) ranks on CPU (gloo).  The per-rank
compute is the oracle on tiny frames - this test covers the N>1 control path, not the kernels."""
import os
import socket

import numpy as np
import pytest

from pyaudiolocalization_amd import RECORD
from pyaudiolocalization_amd.distributed import shard_range, shard_sizes


def _frames(first, count):
    return np.stack([np.random.default_rng([3, first + k]).standard_normal((4, 400)) for k in range(count)])


def _table(frames):
    from oracle import pal_oracle as O
    out = np.zeros((frames.shape[0], 6), dtype=RECORD)
    for b in range(frames.shape[0]):
        rec = O.all_pairs(frames[b], 8000.0, max_expected_delay=0.01)
        for key in ("k_sel", "branch", "k_argmax", "cmax", "cmin", "snr"):
            out[b][key] = rec[key]
    return out


def _worker(rank, world, port, total, queue):
    import torch.distributed as dist
    from pyaudiolocalization_amd.distributed import sharded_tdoa
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        full = sharded_tdoa(_frames, total, rank, world, _table)
        queue.put((rank, full.tobytes()))
    finally:
        dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_shard_ranges_cover_everything():
    for total in (1, 2, 7, 8, 1024):
        for world in (1, 2, 3, 8):
            if world > total:
                continue
            spans = [shard_range(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert max(shard_sizes(total, world)) - min(shard_sizes(total, world)) <= 1
    with pytest.raises(ValueError):
        shard_range(4, 4, 4)


@pytest.mark.parametrize("world,total", [(2, 4), (2, 5), (3, 7)])
def test_gathered_table_equals_single_process_table(world, total):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    queue = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, total, queue)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(queue.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want = _table(_frames(0, total)).tobytes()
    assert all(got[r] == want for r in range(world))      # every rank holds the whole job's table, bit for bit


# ---------------------------------------------------------------- one large frame split by pair blocks
def _big_frame():
    return np.random.default_rng(91).standard_normal((7, 300))            # 21 pairs: blocks of 11 + 10, or 7 + 7 + 7


def _block(frame, pairs):
    """Rows of the pair table for an explicit block of the ordered pair list (the oracle on tiny rows)."""
    from oracle import pal_oracle as O
    out = np.zeros(len(pairs), dtype=RECORD)
    for k, (i, j) in enumerate(pairs):
        rec = O.pair_record(O.phat_correlation(frame[i], frame[j]), frame.shape[1], 8000.0, max_expected_delay=0.01)
        for key in ("k_sel", "branch", "k_argmax", "cmax", "cmin", "snr"):
            out[k][key] = rec[key]
    return out


def _pair_worker(rank, world, port, queue):
    import torch.distributed as dist
    from pyaudiolocalization_amd.distributed import sharded_pair_table
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        queue.put((rank, sharded_pair_table(_big_frame(), rank, world, _block).tobytes()))
    finally:
        dist.destroy_process_group()


def test_pair_blocks_cover_the_ordered_list():
    from pyaudiolocalization_amd.distributed import shard_pairs
    for mics, world in ((256, 8), (64, 3), (7, 2), (4, 6)):
        spans = [shard_pairs(mics, r, world) for r in range(world)]
        assert spans[0][0] == 0 and spans[-1][1] == mics * (mics - 1) // 2
        assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
    assert shard_pairs(256, 0, 8) == (0, 4080) and shard_pairs(256, 7, 8) == (28560, 32640)


@pytest.mark.parametrize("world", [2, 3])
def test_gathered_pair_block_table_equals_single_process_table(world):
    import torch.multiprocessing as mp
    from pyaudiolocalization_amd import pair_list
    ctx = mp.get_context("spawn")
    queue = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_pair_worker, args=(r, world, port, queue)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(queue.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want = _block(_big_frame(), pair_list(7)).tobytes()
    assert all(got[r] == want for r in range(world))
