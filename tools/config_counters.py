#!/usr/bin/env python3
"""fp64 operations and memory-side bytes per pair of a BASELINE configuration C2 ... C5 from its counter passes
(tools/profile_round.sh: profiles/pmc_traffic_<cfg>.json, profiles/<tag>_pmc_flops_<cfg>.json - per-dispatch averages) and its bench
line (pairs per launch group):   python tools/config_counters.py r03_c   -> adds the configurations to profiles/fp64_flops_per_pair.json
Only the kernels of the pair pipeline are counted (row / column / last passes, statistics launches); the forward spectra (64 - 256
transforms per frame against 28 - 32 640 pairs) are left out and named."""
import json
import os
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r03_c"
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "profiles")
path = os.path.join(root, "fp64_flops_per_pair.json")
out = json.load(open(path))
PIPE = ("k_pfa_rows", "k_pfa_cols", "k_rows_lean", "k_peak_", "k_flag_", "PairLoader", "CorrStorer")
for cfg in ("c2", "c3", "c4", "c5"):
    line = json.loads(open(os.path.join(root, f"{tag}_bench_{cfg}.json")).read().strip().splitlines()[-1])
    per_launch = line["roofline"]["pairs_per_launch"]
    traffic = json.load(open(os.path.join(root, f"pmc_traffic_{cfg}.json")))
    flops = json.load(open(os.path.join(root, f"{tag}_pmc_flops_{cfg}.json")))
    names = [k for k in traffic if any(p in k for p in PIPE) and "hhat" not in k]
    if any("PairLoader" in k for k in names):                       # four-step pair pipeline: its row pass is the 8192-point convolution
        names += [k for k in traffic if k.startswith("k_rowsreg<13,conv>") or k.startswith("k_rows<") and "conv" in k]
    if any(k.startswith(("k_rows_lean", "k_pfa_cols_fin")) for k in traffic):
        # the statistics launches of round 2 appear only in the end-of-call pass over the flagged rows (a few small dispatches per call:
        # their per-dispatch averages are not per launch group) - left out, as are the k_flag_* helpers
        names = [k for k in names if not k.startswith(("k_peak_", "k_flag_", "k_pfa_cols_stats"))]
    names = sorted(set(names))
    out[cfg] = {
        "fp64_flops_per_pair": round(sum(flops.get(k, 0) for k in names) / per_launch),
        "memory_side_bytes_per_pair": round(sum(traffic[k] for k in names) / per_launch),
        "pairs_per_launch": per_launch,
        "kernels": {k: {"flops": flops.get(k, 0), "bytes": traffic[k]} for k in names},
        "source": f"rocprofv3 --pmc passes over `bench.py --config {cfg} --steps 1 --warmup 1` (tools/profile_round.sh), per-dispatch averages of the "
                  f"pair pipeline's kernels / pairs per launch group; forward spectra not included; profiles/pmc_traffic_{cfg}.json, profiles/{tag}_pmc_flops_{cfg}.json"}
json.dump(out, open(path, "w"), indent=1)
print({k: (v["fp64_flops_per_pair"], v.get("memory_side_bytes_per_pair")) for k, v in out.items()})
