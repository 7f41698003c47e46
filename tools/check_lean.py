#!/usr/bin/env python3
"""A/B of the stored-row column pass with per-wavefront statistics (PAL_LEAN_STORE=1, default) against the round-2 statistics
(pfa_cols_stats.h / three launches + k_peak_finish; PAL_LEAN_STORE=0) on one box: every record field and, where asked for, the
correlation rows.     python tools/check_lean.py [mics=6] [PAL_LEAN_STORE | PAL_ROWS_LEAN]
(PAL_ROWS_LEAN: k_rows_lean over the stored rows of the other routes against pivots + stream + finish.)"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("PAL_DEBUG_FALLBACK", "1")


SWITCH = sys.argv[2] if len(sys.argv) > 2 else "PAL_LEAN_STORE"


def engine(on):
    os.environ[SWITCH] = "1" if on else "0"
    os.environ["PAL_ROWS_LEAN_MIN"] = "1"                          # (k_rows_lean also for these small calls)
    from pyaudiolocalization_amd import Engine
    return Engine(0)


def main():
    mics = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    a, b = engine(True), engine(False)
    rng = np.random.default_rng(17)
    bad = 0
    lengths = ((12000, 48000.0), (24000, 48000.0), (48000, 48000.0), (44100, 44100.0), (44110, 44100.0), (11170, 16000.0), (3000, 16000.0))
    if SWITCH == "PAL_ROWS_LEAN":
        lengths = ((12000, 48000.0), (44101, 44100.0), (44184, 44100.0), (12007, 16000.0), (6007, 16000.0), (2100, 8000.0), (30011, 44100.0))
    for L, fs in lengths:
        info = a.plan_info(L)
        base = rng.standard_normal(L + 64)
        cases = {"noise": rng.standard_normal((2, mics, L)),
                 "delayed": np.stack([np.stack([base[d:d + L] for d in rng.integers(0, 64, mics)]) for _ in range(2)]) + 0.3 * rng.standard_normal((2, mics, L)),
                 "tone": np.sin(0.05 * np.arange(L))[None, None, :] + 0.3 * rng.standard_normal((1, mics, L))}
        z = rng.standard_normal((1, mics, L)); z[0, 1] = 0.0
        cases["silent mic"] = z
        for name, fr in cases.items():
            for med in (0.05, None, 0.001):
                for method, mult in (("median", 1.0), ("adaptive", 1.0), ("median", 2.0)):
                    for want_corr in (False, True):
                        ra = a.gcc_phat_all_pairs(fr, fs, 1, method, mult, med, want_corr=want_corr)
                        rb = b.gcc_phat_all_pairs(fr, fs, 1, method, mult, med, want_corr=want_corr)
                        ta, tb = (ra[0], rb[0]) if want_corr else (ra, rb)
                        ok = all(np.array_equal(ta[k], tb[k]) for k in ("k_sel", "branch", "k_argmax", "n_sel"))
                        ok = ok and all(np.allclose(ta[k], tb[k], rtol=1e-11, atol=1e-300) for k in ("cmax", "cmin", "snr", "sel_height"))
                        if want_corr:
                            ok = ok and np.allclose(ra[1], rb[1], rtol=0, atol=4e-15)
                        if not ok:
                            bad += 1
                            w = np.flatnonzero((ta["k_sel"] != tb["k_sel"]) | (ta["branch"] != tb["branch"]) | (ta["k_argmax"] != tb["k_argmax"])
                                               | ~np.isclose(ta["snr"], tb["snr"], rtol=1e-11) | ~np.isclose(ta["cmin"], tb["cmin"], rtol=1e-11))
                            print(f"MISMATCH L={L} {name} med={med} {method} x{mult} corr={want_corr}: {w.size} rows, first {w[:4]}")
                            for i in w.ravel()[:2]:
                                print("   lean  ", ta.ravel()[i]); print("   stored", tb.ravel()[i])
        print(f"L={L} n={info['n']} n1={info['n1']} n2={info['n2']}: done, mismatching cases so far {bad}", flush=True)
    print("FAILED" if bad else "ALL EQUAL", bad)
    a.close(); b.close()
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
