#!/usr/bin/env python3
"""fp64 operations and memory-side bytes per pair of the metric run from the per-dispatch averages of tools/pmc_summary.py:
    python tools/flops_per_pair.py gpurun_out/<tag>/pmc_flops.json gpurun_out/<tag>/pmc_traffic.json [frames=16] > profiles/fp64_flops_per_pair.json
The counter passes run `bench.py --frames F`: F x 2016 pairs in launch groups of 480 (the last one partial), the forward
transforms in dispatches of four frames (256 microphone spectra)."""
import json
import math
import sys

flops = json.load(open(sys.argv[1]))
traffic = json.load(open(sys.argv[2])) if len(sys.argv) > 2 else {}
frames = int(sys.argv[3]) if len(sys.argv) > 3 else 16
pairs = frames * 2016
per_group = pairs / math.ceil(pairs / 480)          # average pairs of a pair-pipeline dispatch
per_fwd = min(frames, 4) * 2016                     # pairs served by one forward dispatch
rows, fin = "k_pfa_rows_rader<11,9,10>", "k_pfa_cols_fin"
fwd = ["k_pfa_fwd_cols<11,4>", "k_pfa_fwd_rows_rader<11,9,10>"]
out = {"metric": {
    "fp64_flops_per_pair": round(flops[rows] / per_group + flops[fin] / per_group + sum(flops[k] for k in fwd) / per_fwd),
    "memory_side_bytes_per_pair": (round(traffic[rows] / per_group + traffic[fin] / per_group + sum(traffic[k] for k in fwd) / per_fwd)
                                   if traffic else None),
    "source": ("rocprofv3 --pmc passes (SQ_INSTS_VALU_{FMA,ADD,MUL,TRANS}_F64; FETCH_SIZE, WRITE_SIZE) over `bench.py --steps 2 --warmup 1 "
               f"--frames {frames}`: (2 FMA + ADD + MUL + TRANS) x 64 lanes and 2 x FETCH + WRITE KiB per dispatch; a pair-pipeline dispatch "
               f"holds {per_group:.1f} pairs on average, a forward dispatch serves {per_fwd}; tools/profile_round.sh, tools/flops_per_pair.py"),
    "route": "n=88199 = 89 x 991, Rader rows + Rader columns finishing their rows (pfa_cols_fin.h, pfa_fin_lean.h)",
    "per_kernel_per_pair": {rows: round(flops[rows] / per_group), fin: round(flops[fin] / per_group),
                            "forward": round(sum(flops[k] for k in fwd) / per_fwd)},
    "bytes_per_kernel_per_pair": ({rows: round(traffic[rows] / per_group), fin: round(traffic[fin] / per_group),
                                   "forward": round(sum(traffic[k] for k in fwd) / per_fwd)} if traffic else None)}}
json.dump(out, sys.stdout, indent=1)
