#!/usr/bin/env python3
"""Headline benchmark: mic-pair GCC-PHAT correlations per second on 44.1 kHz x 1 s frames.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

One step = one pass of the hot path (forward spectra of every mic, all-pairs PHAT whitening +
exact-length inverse DFT, peak selection with the reference's full fallback chain, SNR/max/min)
over one batch of synthetic frames that already sits in HBM: 64 mics x 44100 samples per frame,
2016 pairs per frame, float64 (the reference's precision; selected indices are bit-identical).
Frames are independent, so N ranks (one process per GPU) each own their frames - weak scaling, no
data-path collective - and every step ends with ONE all-gather of the 48-byte-per-pair TDOA tables
(RCCL over xGMI through the engine's own communicator; torch.distributed/gloo only carries the
barrier, the unique id and the max-over-ranks of the elapsed time).

Rank 0 prints one JSON line.  ``roofline`` prices the dominant kernel, measured live with HIP
events on the engine's stream inside the timed region; ``cpu_baseline`` times the NumPy oracle
(same pocketfft calls as the reference) on one host core over a bounded sample of the same frame.
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

FS = 44100
HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md, chip-level parameters)


def algorithmic_bytes_per_pair(mics: int, length: int, real_bytes: int = 8) -> float:
    """each pair reads two half spectra and writes one record; each mic frame is
    read once and its half spectrum written once, amortised over the P pairs."""
    h = length                      # n = 2L-1 is odd: H = (n+1)/2 = L bins
    pairs = mics * (mics - 1) // 2
    return 4 * h * real_bytes + (mics / pairs) * (length * real_bytes + 2 * h * real_bytes) + 64


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--frames", type=int, default=16, help="frames per step per GPU (one batch = one step)")
    ap.add_argument("--mics", type=int, default=64)
    ap.add_argument("--length", type=int, default=44100)
    ap.add_argument("--max-expected-delay", type=float, default=0.05, help="seconds; negative = None")
    ap.add_argument("--chunk", type=int, default=0, help="transforms per launch group (0 = engine default)")
    ap.add_argument("--cpu-mics", type=int, default=24, help="mics of frame 0 in the CPU baseline / parity sample")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-events", action="store_true", help="diagnostic: no HIP events around the kernels (roofline = null)")
    ap.add_argument("--event-every", type=int, default=5, help="HIP events around every n-th launch group (1 = all; 5 rotates through the 8 groups of a 64-mic frame)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")

    from pyaudiolocalization_amd import Engine, RECORD, make_params, pair_list
    from pyaudiolocalization_amd.synthetic import metric_frames

    dist = None
    if world > 1:
        import torch.distributed as dist_mod
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist_mod.init_process_group("gloo", rank=rank, world_size=world)
        dist = dist_mod

    # one process per GPU; PAL_BENCH_SHARE_GPU=1 lets a rehearsal on a one-GPU box put every rank on device 0
    eng = Engine(0 if os.environ.get("PAL_BENCH_SHARE_GPU") == "1" else local_rank)
    if args.chunk > 0:
        eng.set_chunk(args.chunk)
    b, m, length = args.frames, args.mics, args.length
    pairs = m * (m - 1) // 2
    med = None if args.max_expected_delay < 0 else args.max_expected_delay
    prm = make_params(FS, 1, "median", 1.0, med)

    frames = metric_frames(b, m, length, first=rank * b)          # this rank's own frames (weak scaling)
    d_frames = eng.alloc(frames.nbytes)
    eng.upload(d_frames, frames)
    tbytes = b * pairs * RECORD.itemsize
    d_table = eng.alloc(tbytes)
    d_all = eng.alloc(tbytes * world) if world > 1 else 0

    gather = "none"
    if world > 1:
        import torch
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
        if any(f != "rccl-allgather" for f in flags):
            gather = "gloo-host"

    host_table = np.zeros((b, pairs), dtype=RECORD)

    def step() -> None:
        eng.gcc_phat_all_pairs_dev(d_frames, b, m, length, prm, d_table)
        if gather == "rccl-allgather":
            eng.all_gather_dev(d_table, d_all, tbytes)
        elif gather == "gloo-host":
            from pyaudiolocalization_amd.distributed import gather_tables_torch
            eng.synchronize()
            eng.download(host_table, d_table)
            gather_tables_torch(host_table, b * world, rank, world)

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
    total_pairs = args.steps * b * pairs * world
    value = total_pairs / elapsed

    # ---- dominant kernel, HIP events on the engine's stream --------------------------------------
    entries = eng.profile_entries()
    kernels = {k: {"ms": round(v[0], 3), "launches": v[1]} for k, v in entries.items() if v[1] > 0}
    # Which kernel is "dominant": in the two-stream run a latency-bound launch (one workgroup per row) waits for slots
    # beside the other stream's FFT passes and its elapsed time stretches severalfold, so elapsed totals do not rank the
    # kernels by work.  One untimed calibration frame on a second, single-stream engine (PAL_OVERLAP=0, events around
    # every launch) ranks them by their time alone; the roofline then uses THAT kernel's live duration from the timed run.
    alone, alone_avg = {}, {}
    if rank == 0 and entries:
        saved = os.environ.get("PAL_OVERLAP")
        os.environ["PAL_OVERLAP"] = "0"
        cal = None
        try:
            cal = Engine(0 if os.environ.get("PAL_BENCH_SHARE_GPU") == "1" else local_rank)
            one = frames[:1]
            d_one, d_tab = cal.alloc(one.nbytes), cal.alloc(pairs * RECORD.itemsize)
            cal.upload(d_one, one)
            cal.gcc_phat_all_pairs_dev(d_one, 1, m, length, prm, d_tab)      # plans, tables, scratch
            cal.synchronize()
            cal.profile_begin(every=1)
            cal.gcc_phat_all_pairs_dev(d_one, 1, m, length, prm, d_tab)
            cal.synchronize()
            cal.profile_end()
            alone = {k: v[0] for k, v in cal.profile_entries().items() if v[1] > 0}
            alone_avg = {k: v[0] / v[1] for k, v in cal.profile_entries().items() if v[1] > 0}
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
    # pairs one launch of the pair pipeline processes: a step is cut into launch groups of `chunk` packed transforms
    # (two pairs each); the events sample every args.event_every-th group, so the launch COUNT says nothing here
    chunk = eng.pair_group_size(length)
    groups_per_step = -(-((b * pairs + 1) // 2) // chunk)
    pairs_per_launch = b * pairs / groups_per_step
    roofline = None
    if dom_launches:
        avg_s = dom_ms * 1e-3 / dom_launches
        achieved = b_alg * pairs_per_launch / avg_s / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath)).get(dom_name)
            except Exception:
                traffic = None
        roofline = {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic, "kernel": dom_name,
                    "avg_launch_us": round(avg_s * 1e6, 2), "launches_timed": dom_launches,
                    "event_sampling": f"every {args.event_every}th launch group",
                    "dominant_by": "largest time alone (one serial calibration frame)" if ranked else "largest elapsed time",
                    # the same launch alone on the GPU (calibration frame, launch groups of 256 transforms): what the kernel
                    # itself reaches; `frac` above is measured while two other launch groups share the CUs
                    "alone_launch_us": round(alone_avg[dom_name] * 1e3, 2) if dom_name in alone_avg else None,
                    "frac_alone": (round(b_alg * min(pairs_per_launch, pairs / max(1, -(-((pairs + 1) // 2) // chunk)))
                                         / (alone_avg[dom_name] * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)
                                   if dom_name in alone_avg else None),
                    "algorithmic_bytes_per_pair": round(b_alg, 1),
                    "pairs_per_launch": round(pairs_per_launch, 2)}

    # ---- CPU baseline + parity sample (rank 0, N = 1 only) ---------------------------------------------
    cpu = None
    parity = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import pal_oracle as O                       # checker + baseline only, never the product path
        cm = min(args.cpu_mics, m)
        t1 = time.perf_counter()
        want = O.all_pairs(frames[0, :cm], FS, max_expected_delay=med)
        cpu_s = time.perf_counter() - t1
        cpairs = cm * (cm - 1) // 2
        cpu = {"value": round(cpairs / cpu_s, 2), "unit": "pair-correlations/s", "cores": 1, "kind": "port",
               "sample": f"all {cpairs} pairs of the first {cm} mics of frame 0 ({cpu_s:.1f} s, NumPy/pocketfft oracle, "
                         "3 exact-length FFTs per pair like utils.py:114-118)"}
        full = pair_list(m)
        pick = np.flatnonzero((full[:, 0] < cm) & (full[:, 1] < cm))
        got = host_table[0][pick]
        parity = {"pairs_checked": int(cpairs), "k_sel_equal": int(np.count_nonzero(got["k_sel"] == want["k_sel"])),
                  "branch_equal": int(np.count_nonzero(got["branch"] == want["branch"])),
                  "max_rel_err_cmax": float(np.max(np.abs(got["cmax"] - want["cmax"]) / np.abs(want["cmax"])))}

    if rank == 0:
        info = eng.plan_info(length)
        line = {
            "metric": "mic-pair GCC-PHAT correlations/s @44.1kHz·1s",
            "value": round(value, 1), "unit": "pair-correlations/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"metric run: {b} frame(s)/step/GPU x {m} mics x {length} samples @ {FS} Hz, "
                                   f"{pairs} pairs/frame, max_expected_delay={med}, exact DFT length n={info['n']}"
                                   + (f" = {info['n1']} x {info['n2']} (prime-factor inverse: "
                                      + (f"Rader row DFTs, cyclic convolutions of {info['n2'] - 1} points in LDS"
                                         if info['tile_len'] & (info['tile_len'] - 1) else
                                         f"in-LDS chirp convolutions of {info['tile_len']} points")
                                      + " + dense column DFTs; forward spectra "
                                      + ("through the same cut, two real frames per transform)"
                                         if info['tile_len'] & (info['tile_len'] - 1) and os.environ.get("PAL_PFA_FWD", "1") != "0" else
                                         f"via chirp convolution {info['m1']}x{info['m2']})") if info.get("n1") else
                                      f" via chirp convolution {info['m1']}x{info['m2']}"),
                       "frames_per_step_per_gpu": b, "mics": m, "samples": length, "pairs_per_frame": pairs,
                       "gather": gather, "parallelism": f"frames sharded over {world} GPU(s)"},
            "roofline": roofline, "cpu_baseline": cpu, "parity": parity, "kernels_ms": kernels,
        }
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    eng.close()


if __name__ == "__main__":
    main()
