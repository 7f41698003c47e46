"""bench.py's host side (CPU) and its one-line JSON contract (GPU)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def test_configurations_match_baseline_json():
    """The workloads bench.py runs are the configurations BASELINE.json names: microphones, sampling rate, frame length."""
    import bench
    with open(os.path.join(ROOT, "BASELINE.json")) as f:
        base = json.load(f)
    assert bench.CONFIGS["metric"]["mics"] == 64 and bench.CONFIGS["metric"]["fs"] == 44100
    assert bench.CONFIGS["metric"]["length"] == 44100                     # 64 mics x 44.1 kHz x 1 s: 2016 pairs per frame
    assert "GCC-PHAT" in base["metric"] and "44.1kHz" in base["metric"]
    assert len(base["configs"]) >= 5                                      # C1 (the reference's own CPU case) ... C5
    for name in ("c2", "c3", "c4", "c5"):
        cfg = bench.CONFIGS[name]
        assert cfg["mics"] >= 2 and cfg["length"] > 0 and cfg["frames"] >= 1, name
    assert (bench.CONFIGS["c2"]["mics"], bench.CONFIGS["c2"]["length"], bench.CONFIGS["c2"]["fs"]) == (8, 48000, 48000)
    assert (bench.CONFIGS["c3"]["mics"], bench.CONFIGS["c3"]["length"]) == (64, 24000)
    assert (bench.CONFIGS["c4"]["mics"], bench.CONFIGS["c4"]["length"], bench.CONFIGS["c4"]["fs"]) == (256, 96000, 96000)
    assert (bench.CONFIGS["c5"]["mics"], bench.CONFIGS["c5"]["length"]) == (64, 12000)


def test_operation_counts_per_route():
    """fp64 operation counts of the roofline block: positive, finite, and ordered as the routes' work is."""
    import bench
    rader = bench.fp64_flops_per_pair({"n": 88199, "n1": 89, "n2": 991, "tile_len": 990, "conv_len": 180224}, 64, 44100)
    tiles = bench.fp64_flops_per_pair({"n": 47999, "n1": 7, "n2": 6857, "tile_len": 16384, "conv_len": 98304}, 64, 24000)
    four = bench.fp64_flops_per_pair({"n": 191999, "n1": 0, "n2": 0, "tile_len": 0, "conv_len": 393216}, 256, 96000)
    for v in (rader, tiles, four):
        assert v is not None and np.isfinite(v) and v > 0
    assert 1.0e7 < rader < 2.5e7                      # 17.6 M measured by the instruction counters
    assert four > rader                               # a 393 216-point convolution against an 88 199-point prime-factor cut


def test_cpu_share_is_sane():
    import bench
    model, visible, use = bench.cpu_info()
    assert isinstance(model, str) and visible >= 1 and 1 <= use <= max(visible, 1)


CONTRACT = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data", "config", "roofline", "cpu_baseline"}


def _run_bench(args, queue):
    """In a fork-server child (no HIP state): bench.py as a subprocess, its last stdout line back through the queue."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, timeout=600, cwd=ROOT)
    queue.put((out.returncode, out.stdout.strip().splitlines()[-1] if out.stdout.strip() else "", out.stderr[-2000:]))


@pytest.mark.gpu
def test_bench_line_keeps_the_contract():
    """`python bench.py` (reduced frames / steps) prints ONE JSON line with every contract key, the roofline block and the
    CPU baseline; the pairs it samples select the oracle's indices."""
    import multiprocessing as mp
    import conftest
    conftest.require_forkserver()
    ctx = mp.get_context("forkserver")          # started in conftest.pytest_sessionstart, before this process touched the GPU
    queue = ctx.Queue()
    p = ctx.Process(target=_run_bench, args=(["--steps", "2", "--warmup", "1", "--frames", "2", "--cpu-mics", "6"], queue))
    p.start()
    rc, line, err = queue.get(timeout=900)
    p.join(timeout=60)
    assert rc == 0, err
    d = json.loads(line)
    assert CONTRACT <= set(d), sorted(CONTRACT - set(d))
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["dtype"] == "f64" and d["data"] == "synthetic" and d["vs_baseline"] is None and d["scaling"] == "weak"
    assert d["value"] > 0 and abs(d["value"] - 2 * 2016 / (d["ms_per_step"] * 1e-3)) <= 0.01 * d["value"]   # (ms_per_step is rounded)
    roof = d["roofline"]
    assert roof["bound"] == "hbm" and roof["unit"] == "GB/s" and roof["peak"] == 8000.0 and 0 < roof["frac"] < 1
    assert abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-3
    cpu = d["cpu_baseline"]
    assert cpu["kind"] == "port" and cpu["cores"] == 1 and cpu["value"] > 0 and "sample" in cpu
    assert d["parity"]["k_sel_equal"] == d["parity"]["pairs_checked"] == 15
    assert "workload" in d["config"] and "model" not in d["config"]


def _run_bench_ranks(world, args, queue, env_extra):
    """In a fork-server child (no HIP state): bench.py under torch.distributed.run with `world` ranks on this box."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, **env_extra)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", str(world)] + args
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    lines = [l for l in out.stdout.strip().splitlines() if l.startswith("{")]
    queue.put((out.returncode, lines[-1] if lines else "", out.stderr[-3000:]))


@pytest.mark.gpu
@pytest.mark.parametrize("split", ["frames", "pairs"])
def test_bench_two_ranks_share_the_gpu(split):
    """The N = 2 line of bench.py, rehearsed on the one-GPU box (PAL_BENCH_SHARE_GPU=1: both ranks on device 0, gather through gloo
    because RCCL refuses two ranks on one device): rank count, partition, gather transport, and a rate of the right order.  The
    children come from the fork server that predates every HIP call."""
    import multiprocessing as mp
    import conftest
    conftest.require_forkserver()
    ctx = mp.get_context("forkserver")
    queue = ctx.Queue()
    args = ["--steps", "2", "--warmup", "1", "--mics", "16", "--length", "6000", "--no-cpu-baseline", "--no-kernel-events", "--split", split]
    if split == "frames":
        args += ["--frames", "2"]
    p = ctx.Process(target=_run_bench_ranks, args=(2, args, queue, {"PAL_BENCH_SHARE_GPU": "1"}))
    p.start()
    rc, line, err = queue.get(timeout=900)
    p.join(timeout=60)
    assert rc == 0, err
    d = json.loads(line)
    assert CONTRACT <= set(d) | {"cpu_baseline"}
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["value"] > 0
    assert d["config"]["gather"].startswith("gloo-host") and d["config"]["gather_in_timed_region"] is False
    assert d["config"]["split"] == split and d["scaling"] == ("strong" if split == "pairs" else "weak")
    pairs = 16 * 15 // 2
    units = pairs if split == "pairs" else 2 * pairs * 2                     # per step: one frame's list, or 2 frames on each of 2 ranks
    assert abs(d["value"] - units / (d["ms_per_step"] * 1e-3)) <= 0.01 * d["value"]


@pytest.mark.gpu
def test_bench_require_rccl_refuses_the_fallback():
    """--require-rccl: a scaling run must not report the gloo fallback silently (the shared-GPU rehearsal cannot bring RCCL up)."""
    import multiprocessing as mp
    import conftest
    conftest.require_forkserver()
    ctx = mp.get_context("forkserver")
    queue = ctx.Queue()
    args = ["--steps", "1", "--warmup", "0", "--mics", "8", "--length", "3000", "--frames", "1", "--no-cpu-baseline", "--no-kernel-events", "--require-rccl"]
    p = ctx.Process(target=_run_bench_ranks, args=(2, args, queue, {"PAL_BENCH_SHARE_GPU": "1"}))
    p.start()
    rc, line, err = queue.get(timeout=900)
    p.join(timeout=60)
    assert rc != 0 and "--require-rccl" in err, (rc, err[-500:])


def test_counter_traffic_is_found_for_every_configuration():
    """roofline.traffic comes from the committed counter passes of the SAME configuration (profiles/pmc_traffic.json for the
    metric run, profiles/pmc_traffic_<config>.json for C2 ... C5); the profiler's kernel names carry more template arguments
    than the engine's labels, the lookup bridges that (never None for a committed configuration)."""
    import bench
    table = {"k_pfa_rows_big<14,32>": 7, "k_colsreg2_fwd<48,13,PairLoader>": 9, "k_colsreg2_fwd<48,13,ChirpLoader>": 1, "k_pfa_cols_fin": 3}
    assert bench.traffic_lookup(table, "k_pfa_rows_big<14>") == 7
    assert bench.traffic_lookup(table, "k_colsreg_fwd<48,PairLoader>") == 9
    assert bench.traffic_lookup(table, "k_pfa_cols_fin") == 3
    assert bench.traffic_lookup(table, "k_rows<9,conv>") is None
    for cfg in ("metric", "c2", "c3", "c4", "c5"):
        tname = "pmc_traffic.json" if cfg == "metric" else f"pmc_traffic_{cfg}.json"
        line = "r03_c_bench.json" if cfg == "metric" else f"r03_c_bench_{cfg}.json"
        with open(os.path.join(ROOT, "profiles", tname)) as f:
            counters = json.load(f)
        with open(os.path.join(ROOT, "profiles", line)) as f:
            dominant = json.loads(f.read().strip().splitlines()[-1])["roofline"]["kernel"]
        assert bench.traffic_lookup(counters, dominant), (cfg, dominant)


def test_measured_operation_counts_exist_for_every_configuration():
    """roofline_fp64 uses instruction counters of the SAME configuration, not the structural count."""
    with open(os.path.join(ROOT, "profiles", "fp64_flops_per_pair.json")) as f:
        rec = json.load(f)
    for cfg in ("metric", "c2", "c3", "c4", "c5"):
        assert rec[cfg]["fp64_flops_per_pair"] > 1e6 and rec[cfg]["memory_side_bytes_per_pair"] > 1e5, cfg
    assert rec["metric"]["fp64_flops_per_pair"] <= 12.0e6 and rec["metric"]["memory_side_bytes_per_pair"] <= 1.8e6   # VERDICT r2 item 1's budgets
    assert rec["c4"]["memory_side_bytes_per_pair"] > 10 * rec["metric"]["memory_side_bytes_per_pair"]                   # the four-step route's bytes
