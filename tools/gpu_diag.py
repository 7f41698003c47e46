"""First-contact diagnostics on a GPU box: per-stage errors of the HIP engine against the oracle,
printed instead of asserted so that one run shows everything.  Not part of the product or the tests."""
import sys, time, os, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import pal_oracle as O
from pyaudiolocalization_amd import Engine

eng = Engine(0)
rng = np.random.default_rng(0)


def section(name, fn):
    t = time.time()
    try:
        fn()
    except Exception:
        print(f"[{name}] EXCEPTION\n{traceback.format_exc()}", flush=True)
    print(f"[{name}] {time.time() - t:.2f} s", flush=True)


def phat():
    for n1, n2 in ((50, 50), (97, 64), (1000, 1000), (2049, 2047), (5000, 5000), (12000, 12000), (44100, 44100), (96000, 96000)):
        a, b = rng.standard_normal(n1), rng.standard_normal(n2)
        got = eng.phat_correlation(a, b)
        want = O.phat_correlation(a, b)
        print(f"  phat n1={n1} n2={n2}: max abs err {np.max(np.abs(got - want)):.3e} argmax {np.argmax(got)} vs {np.argmax(want)} plan {eng.plan_info(max(n1,n2))}", flush=True)


def select():
    bad = 0
    for t in range(60):
        n = int(rng.integers(100, 4000))
        a = rng.standard_normal(n); b = np.roll(a, 5) + 0.5 * rng.standard_normal(n)
        med = [None, 0.01, 0.001][t % 3]; meth = ["median", "adaptive"][t % 2]; npk = [1, 4][t % 2]
        ks, rec, corr = eng.get_time_delays_phat(a, b, 16000.0, npk, meth, 1.0, med)
        want, br = O.select_peaks(O.phat_correlation(a, b), n, 16000.0, npk, meth, 1.0, med)
        ok = np.array_equal(ks, want) and int(rec["branch"]) == br
        if not ok:
            bad += 1
            print(f"  select mismatch t={t} n={n} med={med} meth={meth}: got {ks} br {int(rec['branch'])} want {want} br {br}")
    print(f"  select mismatches: {bad}/60", flush=True)


def allpairs():
    frames = rng.standard_normal((2, 6, 3000)); frames[:, 1:] += 0.5 * frames[:, :1]
    for med in (None, 0.004):
        table = eng.gcc_phat_all_pairs(frames, 16000.0, max_expected_delay=med)
        for t in range(2):
            want = O.all_pairs(frames[t], 16000.0, max_expected_delay=med)
            print(f"  all_pairs med={med} trial {t}: k_sel eq {np.array_equal(table[t]['k_sel'], want['k_sel'])} branch eq {np.array_equal(table[t]['branch'], want['branch'])} "
                  f"argmax eq {np.array_equal(table[t]['k_argmax'], want['k_argmax'])} cmax err {np.max(np.abs(table[t]['cmax']-want['cmax'])):.2e} snr rel {np.max(np.abs(table[t]['snr']/want['snr']-1)):.2e}", flush=True)


def sim():
    base = rng.standard_normal(3000)
    delays = rng.uniform(0.0, 0.01, (3, 4)); gains = rng.uniform(0.1, 1.0, (3, 4))
    got = eng.simulate_multipath(base, 16000.0, 3200, delays, gains, 3000)[0]
    want = O.simulate_from_base(base, delays, gains, 16000.0, 3200, 3000)
    print(f"  simulate: max abs err {np.max(np.abs(got - want)):.3e}", flush=True)
    x = rng.standard_normal(4000)
    print(f"  fractional_delay err {np.max(np.abs(eng.fractional_delay(x, 0.00123, 48000) - O.fractional_delay(x, 0.00123, 48000))):.3e}")
    print(f"  compress err {np.max(np.abs(eng.normalize_compress(x) - O.dynamic_range_compression(x))):.3e}", flush=True)


def filters():
    from scipy.signal import lfilter_zi, firwin
    x = rng.standard_normal((5, 4000))
    for fs in (44100, 48000, 96000):
        b, a = O.butter_bandpass(fs)
        got = eng.filtfilt(b, a, lfilter_zi(b, a), x)
        want = np.array([O.filtfilt(b, a, r) for r in x])
        print(f"  filtfilt fs={fs}: max abs err {np.max(np.abs(got - want)):.3e} bit-identical {np.array_equal(got, want)}", flush=True)
    taps = firwin(101, [300 / 24000, 3400 / 24000], pass_zero=False)
    got = eng.filtfilt(taps, [1.0], lfilter_zi(taps, np.array([1.0])), x[:2])
    want = np.array([O.filtfilt(taps, np.array([1.0]), r) for r in x[:2]])
    print(f"  fir filtfilt: max abs err {np.max(np.abs(got - want)):.3e}")
    print(f"  wiener err {np.max(np.abs(eng.wiener3(x) - np.array([O.wiener3(r) for r in x]))):.3e}", flush=True)


def xcorr():
    y = rng.standard_normal(3000)
    rows = np.array([np.roll(y, k) + 0.05 * rng.standard_normal(3000) for k in (0, 3, -5, 11)])
    kpk, win, pk, ref = eng.xcorr_vs_ref(rows, 0)
    for r in range(4):
        cc = O.xcorr_full(rows[r], rows[0]); k = int(np.argmax(np.abs(cc)))
        print(f"  xcorr row {r}: kpk {kpk[r]} vs {k}; win err {np.max(np.abs(win[r] - cc[k-2:k+3])):.2e}; pk {pk[r]:.6f} vs {abs(cc[k]):.6f}", flush=True)


def speed():
    from pyaudiolocalization_amd import make_params, RECORD
    from pyaudiolocalization_amd.synthetic import metric_frames
    frames = metric_frames(1, 64)
    prm = make_params(44100, 1, "median", 1.0, 0.05)
    ref = None
    for overlap in (1, 0):
        for chunk in (16, 32, 64, 128):
            os.environ["PAL_OVERLAP"] = str(overlap)
            e2 = Engine(0)
            e2.set_chunk(chunk)
            d_f = e2.alloc(frames.nbytes); e2.upload(d_f, frames)
            d_t = e2.alloc(2016 * RECORD.itemsize)
            e2.gcc_phat_all_pairs_dev(d_f, 1, 64, 44100, prm, d_t); e2.synchronize()
            t = time.time()
            for _ in range(5):
                e2.gcc_phat_all_pairs_dev(d_f, 1, 64, 44100, prm, d_t)
            e2.synchronize()
            dt = (time.time() - t) / 5
            tab = np.zeros(2016, dtype=RECORD); e2.download(tab, d_t)
            if ref is None:
                ref = tab.copy()
            same = np.array_equal(tab["k_sel"], ref["k_sel"]) and np.array_equal(tab["branch"], ref["branch"])
            print(f"  overlap={overlap} chunk={chunk:3d}: {2016 / dt:9.0f} pairs/s ({dt * 1e3:6.2f} ms per frame) table same as first: {same}", flush=True)
            if chunk == 32:
                e2.profile_begin()
                for _ in range(3):
                    e2.gcc_phat_all_pairs_dev(d_f, 1, 64, 44100, prm, d_t)
                e2.synchronize()
                e2.profile_end()
                for k, v in sorted(e2.profile_entries().items(), key=lambda kv: -kv[1][0]):
                    if v[1]:
                        print(f"    {k:40s} {v[0]:9.3f} ms {v[1]:5d} launches {v[0] / v[1] * 1e3:9.1f} us avg")
            e2.free(d_f); e2.free(d_t); e2.close()


def peaksvar():
    """Where does k_peaks spend its time?  Same frame, four parameter mixes that switch phases off."""
    from pyaudiolocalization_amd import make_params, RECORD
    from pyaudiolocalization_amd.synthetic import metric_frames
    frames = metric_frames(1, 64)
    d_f = eng.alloc(frames.nbytes); eng.upload(d_f, frames)
    d_t = eng.alloc(2016 * RECORD.itemsize)
    for meth in ("median", "adaptive"):
        for med in (0.05, None):
            prm = make_params(44100, 1, meth, 1.0, med)
            eng.gcc_phat_all_pairs_dev(d_f, 1, 64, 44100, prm, d_t); eng.synchronize()
            eng.profile_begin()
            for _ in range(3):
                eng.gcc_phat_all_pairs_dev(d_f, 1, 64, 44100, prm, d_t)
            eng.synchronize(); eng.profile_end()
            ms, cnt = eng.profile_get("k_peaks")
            print(f"  method={meth:8s} window={med}: k_peaks {ms / cnt * 1e3:8.1f} us per launch of 64 rows", flush=True)


for name, fn in (("peaksvar", peaksvar), ("phat", phat), ("select", select), ("allpairs", allpairs), ("sim", sim), ("filters", filters), ("xcorr", xcorr), ("speed", speed)):
    if len(sys.argv) < 2 or name in sys.argv[1:]:
        section(name, fn)


def pfa():
    """Prime-factor route against the four-step route and the oracle, per length."""
    os.environ["PAL_PFA"] = "0"
    e4 = Engine(0)
    os.environ["PAL_PFA"] = "1"
    ep = Engine(0)
    for L in (50, 496, 500, 1000, 2048, 2999, 3000, 5000, 44100):
        a, b = rng.standard_normal(L), rng.standard_normal(L)
        info = ep.plan_info(L)
        got = ep.phat_correlation(a, b)
        ref = e4.phat_correlation(a, b)
        want = O.phat_correlation(a, b)
        print(f"  L={L} n={info['n']} = {info['n1']} x {info['n2']} tile {info['tile_len']}: pfa-oracle {np.max(np.abs(got - want)):.3e} "
              f"4step-oracle {np.max(np.abs(ref - want)):.3e} pfa-4step {np.max(np.abs(got - ref)):.3e} argmax {np.argmax(got)} {np.argmax(want)}", flush=True)
    frames = rng.standard_normal((1, 5, 3000)); frames[:, 1:] += 0.5 * frames[:, :1]
    tp, cp = ep.gcc_phat_all_pairs(frames, 16000.0, max_expected_delay=0.004, want_corr=True)
    t4, c4 = e4.gcc_phat_all_pairs(frames, 16000.0, max_expected_delay=0.004, want_corr=True)
    print(f"  all-pairs 5 mics (odd pair count): corr diff {np.max(np.abs(cp - c4)):.3e} k_sel eq {np.array_equal(tp['k_sel'], t4['k_sel'])}", flush=True)


if "pfa" in sys.argv[1:]:
    section("pfa", pfa)

