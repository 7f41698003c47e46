#!/usr/bin/env python3
"""A/B of the four-step last pass that finishes its rows (PAL_FIN_FOUR default) against the stored rows + statistics launches
(PAL_FIN_FOUR=0) on lengths that take the four-step route with register rows."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("PAL_DEBUG_FALLBACK", "1")
def engine(on):
    os.environ["PAL_FIN_FOUR"] = "1" if on else "0"
    from pyaudiolocalization_amd import Engine
    return Engine(0)
a, b = engine(True), engine(False)
rng = np.random.default_rng(9)
bad = 0
for L in (44101, 44102, 44106, 12007, 24001, 30011, 6007):
    info = a.plan_info(L)
    m = 6
    base = rng.standard_normal(L + 64)
    cases = {"noise": rng.standard_normal((2, m, L)),
             "delayed": np.stack([np.stack([base[d:d + L] for d in rng.integers(0, 64, m)]) for _ in range(2)]) + 0.3 * rng.standard_normal((2, m, L))}
    z = rng.standard_normal((1, m, L)); z[0, 2] = 0.0
    cases["silent mic"] = z
    for name, fr in cases.items():
        for med in (0.05, None, 0.001):
            for method, mult in (("median", 1.0), ("adaptive", 1.0), ("median", 0.0), ("median", 2.0)):
                ta = a.gcc_phat_all_pairs(fr, 44100.0, 1, method, mult, med)
                tb = b.gcc_phat_all_pairs(fr, 44100.0, 1, method, mult, med)
                ok = all(np.array_equal(ta[k], tb[k]) for k in ("k_sel", "branch", "k_argmax", "n_sel"))
                ok = ok and all(np.allclose(ta[k], tb[k], rtol=1e-11, atol=1e-300) for k in ("cmax", "cmin", "snr", "sel_height"))
                if not ok:
                    bad += 1
                    w = np.flatnonzero((ta["k_sel"] != tb["k_sel"]) | (ta["branch"] != tb["branch"]) | (ta["k_argmax"] != tb["k_argmax"]) | ~np.isclose(ta["snr"], tb["snr"], rtol=1e-11) | ~np.isclose(ta["cmax"], tb["cmax"], rtol=1e-11) | ~np.isclose(ta["cmin"], tb["cmin"], rtol=1e-11))
                    print(f"MISMATCH L={L} {name} med={med} {method} x{mult}: rows {w[:6]}")
                    for i in w.ravel()[:2]:
                        print("   fin   ", ta.ravel()[i]); print("   stored", tb.ravel()[i])
    print(f"L={L} n={info['n']} route m1={info['m1']} m2={info['m2']} n1={info['n1']}: done, mismatching cases so far {bad}", flush=True)
print("FAILED" if bad else "ALL EQUAL", bad)
sys.exit(1 if bad else 0)
