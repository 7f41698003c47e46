#!/usr/bin/env python3
"""Headline benchmark: mic-pair GCC-PHAT correlations per second on 44.1 kHz x 1 s frames.

    python bench.py --gpus 1 --steps K --warmup W                        # the metric workload (BASELINE.json)
    python bench.py --config {c2,c3,c4,c5} ...                           # the other BASELINE configurations, same path
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

One step = one pass of the hot path (forward spectra of every microphone, all-pairs PHAT whitening + exact-length
inverse DFT, peak selection with the reference's full fallback chain, SNR / max / min) over one batch of synthetic
frames that already sits in HBM; float64 like the reference (selected indices are bit-identical).  Frames are
independent, so N ranks (one process per GPU) each own their frames - weak scaling, no data-path collective - and
every step ends with ONE all-gather of the 48-byte-per-pair TDOA tables (RCCL over xGMI through the engine's own
communicator; torch.distributed / gloo only carries the barrier, the unique id and the max-over-ranks of the time).

Rank 0 prints one JSON line.  `roofline` is the HBM roofline BASELINE.json asks for: `frac` is the WHOLE JOB
(pairs/s x algorithmic bytes per pair / 8 TB/s); the dominant kernel is priced beside it with its own compulsory
bytes and its duration from a single-stream calibration pass.  `roofline_fp64` prices the same run against the
fp64 vector rate, which is what actually binds this path.  `cpu_baseline` times the NumPy
oracle (the reference's pocketfft calls) on one core and on all cores of the box, BEFORE the GPU is touched.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec peak (MI355X_MICROARCH.md, chip-level parameters)
FP64_PEAK_TFLOPS = 59.0    # v_fma_f64 rate MEASURED on this part (tools/mfma_f64_rate.hip: the clock sits near 1.8 GHz under
                           # fp64 load; the data-sheet vector figure is 78.6); the guide lists no fp64 vector peak

# BASELINE.json configurations as workloads of THIS path (the pair table); sizes per GPU and step
CONFIGS = {
    "metric": dict(fs=44100, mics=64, length=44100, frames=64, label="metric run"),
    "c2": dict(fs=48000, mics=8, length=48000, frames=256, label="C2 (8-mic array, 48 kHz x 1 s; 256 independent arrays per step)"),
    "c3": dict(fs=48000, mics=64, length=24000, frames=16, label="C3 (64-mic planar array, 16 trials, 48 kHz x 0.5 s)"),
    "c4": dict(fs=96000, mics=256, length=96000, frames=1, label="C4 (256-mic sphere, 96 kHz x 1 s, 32 640 pairs)"),
    "c5": dict(fs=48000, mics=64, length=12000, frames=128, label="C5 (64 mics x 128 of 1024 streaming frames, 48 kHz x 0.25 s)"),
}


def algorithmic_bytes_per_pair(mics: int, length: int, real_bytes: int = 8) -> float:
    """each pair reads two half spectra and writes one record; each mic frame is
    read once and its half spectrum written once, amortised over the P pairs."""
    h = length                      # n = 2L-1 is odd: H = (n+1)/2 = L bins
    pairs = mics * (mics - 1) // 2
    return 4 * h * real_bytes + (mics / pairs) * (length * real_bytes + 2 * h * real_bytes) + 64


def fp64_flops_per_pair(info: dict, mics: int, length: int):
    """fp64 operations per pair-correlation of the route the plan takes (an FMA counts 2), from the kernels'
    structure - DESIGN.md section 5 derives every term.  None for routes without a count."""
    n, n1, n2, tile = info["n"], info.get("n1", 0), info.get("n2", 0), info.get("tile_len", 0)
    pairs = mics * (mics - 1) // 2
    stats = 7.0 * n                                             # streaming statistics: min, two shifted sums of squares
    if n1 == 89 and tile == n2 - 1 and n2 == 991 and os.environ.get("PAL_R89", "1") != "0":
        # Rader rows (9 x 10 x 11) and Rader columns (8 x 11 over four wavefronts, csrc/pfa_rader89.h)
        nr = (n1 + 1) // 2
        whiten = 66.0 * nr * n2
        rader = 2 * nr * (2 * (90 * 250 + 110 * 168 + 99 * 92) + 990 * 6)
        epilogue = 8.0 * n
        cols = n2 * 4 * (2 * (2 * 250 + 11 * 10) + 3 * 112)       # per column: four wavefronts x (stages A and C, three frequencies of stage B)
        per_transform = whiten + rader + epilogue + cols
        forward = per_transform * (mics / 2.0) / pairs
        return (per_transform / 2.0) + forward + 12.0 * n
    if n1 and tile == n2 - 1 and n2 == 991:                     # prime-factor cut with Rader rows (9 x 10 x 11)
        nr = (n1 + 1) // 2
        whiten = 66.0 * nr * n2                                 # two whitened bins + the packed combinations per position
        rader = 2 * nr * (2 * (90 * 250 + 110 * 168 + 99 * 92) + 990 * 6)    # two tiles per workgroup: forward + inverse stages, product
        epilogue = 8.0 * n
        h = (n1 - 1) // 2
        cols = n2 * (h * (6 + 8 * h) + 8 * h)                   # dense symmetric N1-point DFT per column, both pairs
        per_transform = whiten + rader + epilogue + cols        # one packed transform = two pairs
        forward = per_transform * (mics / 2.0) / pairs          # forward spectra through the same cut, amortised
        return (per_transform / 2.0) + forward + stats
    if n1:                                                      # prime-factor cut with chirp-convolution row tiles of `tile` points
        nr, h = (n1 + 1) // 2, (n1 - 1) // 2
        tiles = n1 if tile >= 8192 else 2 * nr                  # one register tile per row, or two LDS tiles per row pair
        conv = 2 * 5.0 * tile * np.log2(tile) + 6.0 * tile      # forward FFT, product with the chirp spectrum, inverse FFT
        whiten = 66.0 * n2 * (n1 if tile >= 8192 else nr)
        cols = n2 * (h * (6 + 8 * h) + 8 * h) if h else 0.0
        per_transform = tiles * conv + whiten + 12.0 * n + cols
        forward = (2 * 5.0 * info["conv_len"] * np.log2(info["conv_len"])) * mics / pairs     # forward spectra: four-step
        return per_transform / 2.0 + forward + stats
    if not n1:                                                  # four-step chirp convolution: 5 M log2 M per FFT, two FFTs + products
        m = info["conv_len"]
        fft = 5.0 * m * np.log2(m)
        per_transform = 2 * fft + 6.0 * m + 66.0 * (n // 2 + 1) + 12.0 * n
        forward = (2 * 5.0 * info["conv_len"] * np.log2(info["conv_len"])) * mics / pairs
        return per_transform / 2.0 + forward + stats
    return None


def cpu_info():
    """(CPU model, cores this process may run on, cores worth using): the affinity mask of a GPU box lists every thread
    of the host (256) while the box's share of it is 16 cores per GPU; a cgroup CPU quota, when set, is the hard bound."""
    model = "unknown"
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    model = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    affinity = len(os.sched_getaffinity(0))
    quota = None
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            q, period = f.read().split()[:2]
            if q != "max":
                quota = max(1, int(float(q) / float(period) + 0.5))
    except (OSError, ValueError):
        pass
    share = int(os.environ.get("PAL_BENCH_CPU_CORES", "0")) or min(affinity, quota if quota else 16)
    return model, affinity, share


_SHARED = {}          # frame 0 for the forked pool workers (copy-on-write: nothing is pickled per task)


def _cpu_pairs(args):
    """Worker of the all-cores leg: the oracle's get_time_delays_phat-equivalent for a block of pairs of one frame."""
    pairs, fs, med = args
    rows = _SHARED["frame0"]
    from oracle import pal_oracle as O
    out = []
    for i, j in pairs:
        rec = O.pair_record(O.phat_correlation(rows[i], rows[j]), rows.shape[1], fs, max_expected_delay=med)
        out.append((rec["k_sel"], rec["branch"], rec["cmax"]))
    return out


def cpu_baseline(frame0: np.ndarray, fs: float, med, cpu_mics: int, budget_s: float = 12.0):
    """Reference CPU path (NumPy oracle = the reference's pocketfft calls) on this box's host cores: one core (the
    reference is single-threaded), then a process pool over pairs on every core.  Runs before any HIP call."""
    import multiprocessing as mp
    from oracle import pal_oracle as O
    m = frame0.shape[0]
    cm = min(cpu_mics, m)
    t1 = time.perf_counter()
    want = O.all_pairs(frame0[:cm], fs, max_expected_delay=med)
    one_s = time.perf_counter() - t1
    cpairs = cm * (cm - 1) // 2
    model, affinity, cores = cpu_info()
    rate1 = cpairs / one_s
    # all cores: blocks of four pairs of the same frame handed to a process pool until the time budget is spent
    full = [(i, j) for i in range(m) for j in range(i + 1, m)]
    blocks = [full[k: k + 4] for k in range(0, len(full), 4)]
    all_rate, all_s, npool = None, None, 0
    try:
        _SHARED["frame0"] = frame0
        ctx = mp.get_context("fork")                              # (no GPU state exists yet in this process)
        pool = ctx.Pool(cores)
        try:
            pool.map(_cpu_pairs, [(b[:1], fs, med) for b in blocks[:cores]])      # imports + FFT plans
            t2 = time.perf_counter()
            for done in pool.imap_unordered(_cpu_pairs, [(b, fs, med) for b in blocks]):
                npool += len(done)
                if time.perf_counter() - t2 > budget_s:
                    break
            all_s = time.perf_counter() - t2
        finally:
            pool.terminate()
            pool.join()
        all_rate = npool / all_s
    except Exception as exc:                                      # reported, never silent
        print(f"[bench] all-cores CPU leg failed: {exc}", file=sys.stderr)
    cpu = {"value": round(rate1, 2), "unit": "pair-correlations/s", "cores": 1, "kind": "port",
           "sample": f"all {cpairs} pairs of the first {cm} mics of frame 0 ({one_s:.1f} s, NumPy/pocketfft oracle, "
                     "3 exact-length FFTs per pair like utils.py:114-118)",
           "all_cores": {"value": round(all_rate, 2) if all_rate else None, "cores": cores,
                         "sample": f"{npool} pairs of frame 0 over a pool of {cores} processes ({all_s:.1f} s)" if all_s else "failed"},
           "cpu_model": model, "host_cores_visible": affinity,
           "cores_note": "pool size = the box's CPU share per GPU (16) or its cgroup quota; the affinity mask lists the whole host"}
    return cpu, want, cm


def traffic_lookup(table, name):
    """A kernel's bytes per launch from a tools/pmc_summary.py table.  The profiler's names carry template arguments the
    engine's labels drop (k_pfa_rows_big<14,32> for k_pfa_rows_big<14>, k_colsreg2_fwd<48,13,PairLoader> for
    k_colsreg_fwd<48,PairLoader>): both sides are reduced to base name + first number + the loader / storer."""
    import re

    def norm(k):
        k = k.replace("colsreg2_", "colsreg_")
        m = re.match(r"([A-Za-z_0-9]+)(?:<(\d+)((?:,[^>]*)?)>)?", k)
        if not m:
            return k
        words = [w for w in (m.group(3) or "").split(",") if w and not w.isdigit()]
        return (m.group(1), m.group(2), tuple(words))

    if name in table:
        return table[name]
    want = norm(name)
    for k, v in table.items():
        if norm(k) == want:
            return v
    return None


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", choices=sorted(CONFIGS), default="metric",
                    help="metric = BASELINE.json's headline workload; c2..c5 = the other configurations on the same path")
    ap.add_argument("--frames", type=int, default=0, help="frames per step per GPU (0 = the configuration's default)")
    ap.add_argument("--mics", type=int, default=0)
    ap.add_argument("--length", type=int, default=0)
    ap.add_argument("--fs", type=float, default=0)
    ap.add_argument("--max-expected-delay", type=float, default=0.05, help="seconds; negative = None")
    ap.add_argument("--chunk", type=int, default=0, help="transforms per launch group (0 = engine default)")
    ap.add_argument("--cpu-mics", type=int, default=24, help="mics of frame 0 in the single-core CPU baseline / parity sample")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-events", action="store_true", help="diagnostic: no HIP events around the kernels (roofline kernel block = null)")
    ap.add_argument("--event-every", type=int, default=5, help="HIP events around every n-th launch group (1 = all)")
    ap.add_argument("--split", choices=["frames", "pairs"], default="frames",
                    help="frames: every rank owns its own frames (weak scaling); pairs: ONE frame's ordered pair list is block-partitioned "
                         "over the ranks, spectra recomputed on every rank (strong scaling of a single large frame, SURVEY 8e: C4)")
    ap.add_argument("--require-rccl", action="store_true", help="exit non-zero when the per-step gather cannot run on RCCL (N > 1)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    cfg = CONFIGS[args.config]
    fs = float(args.fs or cfg["fs"])
    b, m, length = args.frames or cfg["frames"], args.mics or cfg["mics"], args.length or cfg["length"]
    pairs = m * (m - 1) // 2
    med = None if args.max_expected_delay < 0 else args.max_expected_delay

    from pyaudiolocalization_amd.synthetic import metric_frames
    split_pairs = args.split == "pairs"
    if split_pairs:
        b = 1                                                     # ONE frame, the same on every rank; its pair list is block-partitioned
    frames = metric_frames(b, m, length, first=0 if split_pairs else rank * b)   # this rank's own frames (weak scaling) / the shared frame

    # ---- CPU baseline first: host cores only, before this process initialises the GPU -------------------------------
    cpu, want, cm = None, None, 0
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu, want, cm = cpu_baseline(frames[0], fs, med, args.cpu_mics if args.config == "metric" else min(args.cpu_mics, 12))

    from pyaudiolocalization_amd import Engine, RECORD, make_params, pair_list

    dist = None
    if world > 1:
        import torch.distributed as dist_mod
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist_mod.init_process_group("gloo", rank=rank, world_size=world)
        dist = dist_mod

    # one process per GPU; PAL_BENCH_SHARE_GPU=1 lets a rehearsal on a one-GPU box put every rank on device 0
    share = os.environ.get("PAL_BENCH_SHARE_GPU") == "1"
    eng = Engine(0 if share else local_rank)
    if args.chunk > 0:
        eng.set_chunk(args.chunk)
    prm = make_params(fs, 1, "median", 1.0, med)

    d_frames = eng.alloc(frames.nbytes)
    eng.upload(d_frames, frames)
    lo, hi = 0, pairs
    d_pairs = 0
    if split_pairs:
        from pyaudiolocalization_amd.distributed import shard_pairs
        lo, hi = shard_pairs(m, rank, world)
        if min(shard_pairs(m, r, world)[1] - shard_pairs(m, r, world)[0] for r in range(world)) <= 0:
            raise SystemExit("more ranks than pairs")
        # equal blocks for the all-gather: every rank's block is padded to the largest one (repeating its last pair)
        blk = max(shard_pairs(m, r, world)[1] - shard_pairs(m, r, world)[0] for r in range(world))
        mine = pair_list(m)[lo:hi]
        mine = np.concatenate([mine, np.repeat(mine[-1:], blk - len(mine), axis=0)]).astype(np.int32)
        d_pairs = eng.alloc(mine.nbytes)
        eng.upload(d_pairs, np.ascontiguousarray(mine))
        tbytes = blk * RECORD.itemsize
    else:
        tbytes = b * pairs * RECORD.itemsize
    d_table = eng.alloc(tbytes)
    d_all = eng.alloc(tbytes * world) if world > 1 else 0

    gather = "none"
    if world > 1:
        if share:
            # RCCL refuses two ranks on one device (ncclCommInitRank: "unhandled cuda error" - duplicate GPU); a one-GPU
            # rehearsal therefore gathers through gloo and says so
            gather = "gloo-host (ranks share one GPU: RCCL needs one device per rank)"
        else:
            try:                                                      # engine-native RCCL communicator
                ident = [Engine.comm_unique_id() if rank == 0 else None]
                dist.broadcast_object_list(ident, src=0)
                eng.comm_init(world, rank, ident[0])
                gather = "rccl-allgather"
            except Exception as exc:                                  # reported, never silent
                print(f"[rank {rank}] RCCL init failed ({exc}); gathering through gloo host tensors", file=sys.stderr)
                gather = "gloo-host"
        flags = [None] * world
        dist.all_gather_object(flags, gather)
        if any(f != "rccl-allgather" for f in flags) and gather == "rccl-allgather":
            gather = "gloo-host"
        if args.require_rccl and gather != "rccl-allgather":
            print(f"[rank {rank}] --require-rccl: the gather would run on '{gather}'", file=sys.stderr)
            dist.destroy_process_group()
            sys.exit(3)

    host_table = np.zeros((blk,) if split_pairs else (b, pairs), dtype=RECORD)

    # The frames are independent: the pair tables of a step need no exchange to be computed.  Where RCCL is up,
    # every step still ends with ONE all-gather of the 48-byte records on the engine's communicator (6 MB per rank: the
    # consumer of a step sees the whole table).  Without RCCL the shards stay on their ranks during the timed region and are
    # gathered ONCE through gloo host tensors after it (said in config.gather): a host round trip per step would measure the
    # loopback socket, not the path.
    def step() -> None:
        if split_pairs:
            eng.gcc_phat_pairs_dev(d_frames, m, length, d_pairs, blk, prm, d_table)
        else:
            eng.gcc_phat_all_pairs_dev(d_frames, b, m, length, prm, d_table)
        if gather == "rccl-allgather":
            eng.all_gather_dev(d_table, d_all, tbytes)

    def barrier() -> None:
        eng.synchronize()
        if dist is not None:
            dist.barrier()
        eng.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    if not args.no_kernel_events:
        eng.profile_begin(every=args.event_every)   # HIP events around a sample of the launch groups (they cost idle stream time)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    if not args.no_kernel_events:
        eng.profile_end()
    if dist is not None:
        import torch
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    eng.download(host_table, d_table)
    gather_timed = gather == "rccl-allgather"                      # the per-step exchange ran inside the timed region
    if gather.startswith("gloo-host"):
        if split_pairs:
            from pyaudiolocalization_amd.distributed import gather_blocks_torch
            full = gather_blocks_torch(host_table, [blk] * world)
            if len(full) != blk * world:
                raise RuntimeError("gathered table has the wrong number of pairs")
        else:
            from pyaudiolocalization_amd.distributed import gather_tables_torch
            full = gather_tables_torch(host_table, b * world, rank, world)
            if full is not None and len(full) != b * world:
                raise RuntimeError("gathered table has the wrong number of frames")
        gather += " - once, after the timed region"
    total_pairs = args.steps * (pairs if split_pairs else b * pairs * world)     # pairs: the ONE frame's list, whatever the rank count
    value = total_pairs / elapsed

    # ---- per-kernel durations: live (HIP events inside the timed region, three streams share the CUs) and alone
    #      (one untimed calibration frame on a second, single-stream engine: PAL_OVERLAP=0, events around every launch)
    entries = eng.profile_entries()
    kernels = {k: {"ms": round(v[0], 3), "launches": v[1]} for k, v in entries.items() if v[1] > 0}
    alone, alone_avg = {}, {}
    cal_pairs_per_launch = 0.0
    if rank == 0 and entries:
        saved = os.environ.get("PAL_OVERLAP")
        os.environ["PAL_OVERLAP"] = "0"
        cal = None
        try:
            cal = Engine(0 if share else local_rank)
            cb = 1 if split_pairs else min(b, max(1, -(-2 * eng.pair_group_size(length) // pairs)))       # frames that fill one launch group
            one = frames[:cb]
            d_one, d_tab = cal.alloc(one.nbytes), cal.alloc(cb * pairs * RECORD.itemsize)
            cal.upload(d_one, one)
            cal.gcc_phat_all_pairs_dev(d_one, cb, m, length, prm, d_tab)      # plans, tables, scratch
            cal.synchronize()
            cal.profile_begin(every=1)
            cal.gcc_phat_all_pairs_dev(d_one, cb, m, length, prm, d_tab)
            cal.synchronize()
            cal.profile_end()
            alone = {k: v[0] for k, v in cal.profile_entries().items() if v[1] > 0}
            # per FULL launch: the calibration frames end in a partial launch group, so the plain average per launch would
            # mix a short launch in; total time / units processed x units of a full launch instead (pair kernels: pairs of a
            # launch group; forward-spectrum kernels: the 256 frames of a launch of 128 packed transforms)
            full_pairs = 2 * eng.pair_group_size(length)
            for k, v in cal.profile_entries().items():
                if v[1] <= 0:
                    continue
                forward = "k_pfa_fwd" in k or "FrameLoader" in k or "SpectrumStorer" in k
                alone_avg[k] = v[0] / (cb * m) * min(256, cb * m) if forward else v[0] / (cb * pairs) * min(full_pairs, cb * pairs)
            cal_pairs_per_launch = min(full_pairs, cb * pairs)
        except Exception as exc:                                  # reported, never silent: the ranking falls back to elapsed totals
            print(f"[bench] calibration frame failed ({exc}); ranking kernels by elapsed time", file=sys.stderr)
            alone, alone_avg = {}, {}
        finally:
            if saved is None:
                os.environ.pop("PAL_OVERLAP", None)
            else:
                os.environ["PAL_OVERLAP"] = saved
            if cal is not None:
                cal.close()
    ranked = [k for k in sorted(alone, key=alone.get, reverse=True) if k in entries and entries[k][1] > 0]
    dom = (ranked[0], entries[ranked[0]]) if ranked else (max(entries.items(), key=lambda kv: kv[1][0]) if entries else ("none", (0.0, 0)))
    dom_name, (dom_ms, dom_launches) = dom
    b_alg = algorithmic_bytes_per_pair(m, length)
    chunk = eng.pair_group_size(length)
    groups_per_step = -(-((b * pairs + 1) // 2) // chunk)
    pairs_per_launch = b * pairs / groups_per_step
    info = eng.plan_info(length)
    h_bins = length
    # compulsory bytes of ONE pair inside each kernel class of the prime-factor / four-step pipelines (what that kernel
    # must move for a pair even with perfect caching): spectra in, workspace out / in, correlation row out / in
    n = info["n"]
    spec_b, corr_b = 4 * h_bins * 8, n * 8                        # two half spectra in, one correlation row
    y_b = n * 16 / 2                                              # prime-factor grid Y: one complex transform per two pairs
    w_b = info["conv_len"] * 16 / 2                               # four-step workspace W per pair

    def own(name):
        if "k_pfa_cols_fin" in name: return y_b                   # the column pass that finishes its rows: Y in, 48-byte records out
        if "k_pfa_cols" in name: return y_b + corr_b              # (also the fused k_pfa_cols_stats)
        if "k_pfa_rows" in name: return spec_b + y_b
        if "k_peak_stream" in name: return corr_b
        if name.startswith("k_cols") and "_fwd" in name: return spec_b + w_b
        if name.startswith("k_rows"): return 2 * w_b
        if name.startswith("k_cols") and "_inv" in name: return w_b + corr_b
        return None

    achieved_job = value / world * b_alg / 1e9                     # per GPU
    roofline = {"bound": "hbm", "achieved": round(achieved_job, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved_job / HBM_PEAK_GBS, 5),
                "frac_is": "whole job per GPU: pairs/s x algorithmic bytes per pair / peak",
                "algorithmic_bytes_per_pair": round(b_alg, 1), "traffic": None, "kernel": None}
    if dom_launches:
        live_s = dom_ms * 1e-3 / dom_launches
        traffic, traffic_src = None, None
        # counter passes are kept per configuration (tools/profile_round.sh): profiles/pmc_traffic.json for the metric run,
        # profiles/pmc_traffic_<config>.json for C2 ... C5; a workload given by --mics / --length has none
        tname = "pmc_traffic.json" if args.config == "metric" else f"pmc_traffic_{args.config}.json"
        tpath = os.path.join(ROOT, "profiles", tname)
        default_shape = not args.mics and not args.length
        if os.path.exists(tpath) and default_shape:
            try:
                traffic = traffic_lookup(json.load(open(tpath)), dom_name)
                traffic_src = f"profiles/{tname} (rocprofv3 --pmc passes of an earlier run of this command, NOT this run)"
            except Exception:
                traffic = None
        kb = own(dom_name)
        roofline.update({
            "traffic": traffic, "traffic_source": traffic_src, "kernel": dom_name,
            "dominant_by": "largest time alone (single-stream calibration pass)" if ranked else "largest elapsed time",
            "kernel_alone_launch_us": round(alone_avg[dom_name] * 1e3, 2) if dom_name in alone_avg else None,
            "kernel_live_launch_us": round(live_s * 1e6, 2),
            "kernel_live_note": "HIP events inside the timed region; three launch groups share the CUs, so this duration is stretched by concurrency",
            "kernel_own_bytes_per_pair": kb,
            "kernel_frac_alone": (round(kb * cal_pairs_per_launch / (alone_avg[dom_name] * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)
                                  if kb and dom_name in alone_avg else None),
            "launches_timed": dom_launches, "event_sampling": f"every {args.event_every}th launch group",
            "pairs_per_launch": round(pairs_per_launch, 2)})
    flops = fp64_flops_per_pair(info, m, length)
    flops_source = "operation count of the kernels' structure (DESIGN.md section 5), FMA = 2"
    mpath = os.path.join(ROOT, "profiles", "fp64_flops_per_pair.json")
    if not args.mics and not args.length and os.path.exists(mpath):
        try:                                                       # measured instruction counts of an earlier PMC pass of this workload
            rec = json.load(open(mpath))[args.config]
            flops = float(rec["fp64_flops_per_pair"])
            flops_source = "measured: fp64 instruction counters of an earlier PMC pass of this command, NOT this run (profiles/fp64_flops_per_pair.json)"
        except Exception:
            pass
    roofline_fp64 = None
    if flops:
        tf = value / world * flops / 1e12
        roofline_fp64 = {"bound": "fp64-valu", "achieved": round(tf, 2), "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(tf / FP64_PEAK_TFLOPS, 4), "flops_per_pair": round(flops),
                         "peak_source": "v_fma_f64 rate measured on MI355X (tools/mfma_f64_rate.hip); data sheet 78.6",
                         "flops_source": flops_source}
    binding = None
    if roofline_fp64:
        binding = "fp64-valu" if roofline_fp64["frac"] > roofline["frac"] else "hbm"

    # ---- parity sample against the CPU oracle's rows computed above ------------------------------------------------
    parity = None
    if want is not None and not split_pairs:
        full = pair_list(m)
        pick = np.flatnonzero((full[:, 0] < cm) & (full[:, 1] < cm))
        got = host_table[0][pick]
        parity = {"pairs_checked": int(pick.size), "k_sel_equal": int(np.count_nonzero(got["k_sel"] == want["k_sel"])),
                  "branch_equal": int(np.count_nonzero(got["branch"] == want["branch"])),
                  "max_rel_err_cmax": float(np.max(np.abs(got["cmax"] - want["cmax"]) / np.abs(want["cmax"])))}

    if rank == 0:
        if info.get("n1"):
            rader = info["tile_len"] == info["n2"] - 1
            route = (f"n={info['n']} = {info['n1']} x {info['n2']}: prime-factor inverse, "
                     + (f"Rader row DFTs (cyclic convolutions of {info['n2'] - 1} points in LDS)" if rader
                        else f"in-LDS chirp convolutions of {info['tile_len']} points")
                     + " + dense column DFTs")
        else:
            route = f"n={info['n']}: four-step chirp convolution {info['m1']}x{info['m2']}"
        line = {
            "metric": "mic-pair GCC-PHAT correlations/s @44.1kHz·1s" if args.config == "metric" and length == 44100
                      else f"mic-pair GCC-PHAT correlations/s @{fs / 1000:g}kHz·{length / fs:g}s",
            "value": round(value, 1), "unit": "pair-correlations/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True,
            "scaling": "strong" if split_pairs else "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"{cfg['label']}: {b} frame(s)/step/GPU x {m} mics x {length} samples @ {fs:g} Hz, "
                                   f"{pairs} pairs/frame, max_expected_delay={med}; {route}",
                       "name": args.config, "frames_per_step_per_gpu": b, "mics": m, "samples": length, "pairs_per_frame": pairs,
                       "fused_column_statistics": os.environ.get("PAL_FUSED", "default"),
                       "finishing_column_pass": os.environ.get("PAL_FIN", "default"), "rader_columns": os.environ.get("PAL_R89", "default"),
                       "gather": gather, "gather_in_timed_region": gather_timed if world > 1 else None, "split": args.split,
                       "parallelism": (f"pair list of one frame block-partitioned over {world} GPU(s), spectra recomputed per rank" if split_pairs
                                       else f"frames sharded over {world} GPU(s)")},
            "roofline": roofline, "roofline_fp64": roofline_fp64, "binding_roofline": binding,
            "cpu_baseline": cpu, "parity": parity, "kernels_ms": kernels,
            "kernels_alone_us": {k: round(v * 1e3, 2) for k, v in alone_avg.items()},
            "kernels_alone_note": "single-stream calibration pass, per full launch (group of 2 x %d pairs; forward kernels per 256 frames)" % chunk,
            "timed_region_s": round(elapsed, 3),
        }
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    eng.close()


if __name__ == "__main__":
    main()
