#!/usr/bin/env python3
"""A/B of the finishing column pass (PAL_FIN=1, pfa_cols_fin.h) against the stored-row path (PAL_FIN=0) on one box:
every record field of random and structured frames, both window modes, both threshold methods.
    python tools/check_fin.py [mics] [frames] [length=44100]     (PAL_FIN_DENSE=1 / PAL_FIN_STRIPS=1: the pass on dense column DFTs)"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("PAL_DEBUG_FALLBACK", "1")


def engine(fin):
    os.environ["PAL_FIN"] = "1" if fin else "0"
    from pyaudiolocalization_amd import Engine
    return Engine(0)


def main():
    mics = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    nfr = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    L = int(sys.argv[3]) if len(sys.argv) > 3 else 44100
    from pyaudiolocalization_amd.synthetic import metric_frames
    a, b = engine(True), engine(False)
    rng = np.random.default_rng(5)
    cases = {"metric": metric_frames(nfr, mics, L),
             "noise": rng.standard_normal((nfr, mics, L)),
             "tone+noise": np.sin(0.05 * np.arange(L))[None, None, :] + 0.3 * rng.standard_normal((nfr, mics, L))}
    z = rng.standard_normal((1, mics, L)); z[0, 1] = 0.0
    cases["silent mic"] = z
    bad = 0
    for name, fr in cases.items():
        for med in (0.05, None, 0.0005):
            for method, mult in (("median", 1.0), ("median", 4.2), ("adaptive", 1.0), ("median", 60.0)):
                ta = a.gcc_phat_all_pairs(fr, 44100.0, 1, method, mult, med)
                tb = b.gcc_phat_all_pairs(fr, 44100.0, 1, method, mult, med)
                ok = all(np.array_equal(ta[k], tb[k]) for k in ("k_sel", "branch", "k_argmax", "n_sel"))
                ok = ok and all(np.allclose(ta[k], tb[k], rtol=1e-11, atol=1e-300) for k in ("cmax", "cmin", "snr", "sel_height"))
                if not ok:
                    bad += 1
                    w = np.flatnonzero((ta["k_sel"] != tb["k_sel"]) | (ta["branch"] != tb["branch"]) | (ta["k_argmax"] != tb["k_argmax"])
                                       | ~np.isclose(ta["snr"], tb["snr"], rtol=1e-11) | ~np.isclose(ta["cmin"], tb["cmin"], rtol=1e-11))
                    print(f"MISMATCH {name} med={med} {method} x{mult}: {w.size} rows, first {w[:4]}")
                    for i in w.ravel()[:3]:
                        print("   fin   ", ta.ravel()[i]); print("   stored", tb.ravel()[i])
                else:
                    print(f"ok {name} med={med} {method} x{mult}  branches {np.unique(ta['branch']).tolist()}")
    print("FAILED" if bad else "ALL EQUAL", bad)
    a.close(); b.close()
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
