"""Golden-vector generator  --  runs ONLY in the build container (needs /root/reference).

Imports the unmodified reference modules, feeds them the synthetic inputs of ``oracle/cases.py``
and writes KB-scale ``tests/golden/*.npz`` fixtures (inputs are rebuilt from seeds by the tests,
only the reference's OUTPUTS are stored).  Nothing here travels to the GPU box as a dependency.

``soundfile`` and ``resampy`` are not installed in this image and are imported at the top of the
reference's ``utils.py:8`` / ``signal_processing.py:8``; they are used only by ``read_audio_files``
and ``resample_audio`` (out of scope, never called here), so two EMPTY module objects are
registered under those names to let the import statements pass.

    cd /tmp && MPLBACKEND=Agg python /root/repo/oracle/make_golden.py
"""
from __future__ import annotations

import logging
import os
import sys
import tempfile
import time
import types

sys.dont_write_bytecode = True
os.environ.setdefault("MPLBACKEND", "Agg")
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
for _name in ("soundfile", "resampy"):
    sys.modules.setdefault(_name, types.ModuleType(_name))
sys.path.insert(0, REF)
sys.path.insert(0, REPO)

import numpy as np  # noqa: E402
import scipy  # noqa: E402
import sklearn  # noqa: E402

import main as ref_main  # noqa: E402
import signal_processing as ref_sp  # noqa: E402
import utils as ref_utils  # noqa: E402

from oracle import cases  # noqa: E402

logging.disable(logging.CRITICAL)
OUT = os.path.join(REPO, "tests", "golden")
META = np.array([f"numpy {np.__version__}", f"scipy {scipy.__version__}", f"sklearn {sklearn.__version__}",
                 "reference zeynelacikgoez/PyAudioLocalization @ 2025-08-01"])


def pair_table(filtered, fs, med):
    """Reference pair loop (main.py:202-228) -> per-pair records."""
    m = len(filtered)
    n2 = len(filtered[0])
    rec = {k: [] for k in ("k_sel", "k_argmax", "cmax", "cmin", "snr", "ptp", "td")}
    for i in range(m):
        for j in range(i + 1, m):
            td, corr, lags = ref_utils.get_time_delays_phat(filtered[i], filtered[j], fs, num_peaks=1,
                                                            max_expected_delay=med)
            rec["td"].append(td[0])
            rec["k_sel"].append(int(round(td[0] * fs)) + n2 - 1)
            rec["k_argmax"].append(int(np.argmax(corr)))
            rec["cmax"].append(np.max(corr))
            rec["cmin"].append(np.min(corr))
            rec["snr"].append(ref_utils.compute_snr(corr))
            rec["ptp"].append(ref_utils.compute_peak_to_peak_ratio(corr))
    out = {k: np.array(v, dtype=np.float64) for k, v in rec.items()}
    out["k_sel"] = out["k_sel"].astype(np.int32)
    out["k_argmax"] = out["k_argmax"].astype(np.int32)
    return out


def stage_outputs(signals, fs, meds, prefix=""):
    """sync -> filter -> pairs with the reference's functions (main.py:188-228)."""
    synced = ref_utils.synchronize_signals_improved(signals, fs)
    filt = [ref_sp.noise_reduction(s, fs, method="butterworth") for s in synced]
    out = {prefix + "L": np.array([len(filt[0])]),
           prefix + "sync_digest": np.array([cases.waveform_digest(s) for s in synced]),
           prefix + "filt_digest": np.array([cases.waveform_digest(s) for s in filt])}
    for med in meds:
        tag = "none" if med is None else ("%g" % med).replace(".", "p")
        for k, v in pair_table(filt, fs, med).items():
            out[f"{prefix}{k}_{tag}"] = v
    return out


def run_localize(cfg, materials=None, base=None, calibration=None, seed=None, full=False):
    """localize_sound_source in a scratch dir (it writes PNGs)."""
    keep_mat, keep_gen = ref_main.material_properties, ref_sp.generate_signal
    cwd = os.getcwd()
    try:
        if materials is not None:
            ref_main.material_properties = materials
        if base is not None:
            ref_sp.generate_signal = lambda *a, **k: base.copy()
        with tempfile.TemporaryDirectory() as tmp:
            os.chdir(tmp)
            if seed is not None:
                np.random.seed(seed)
            res = ref_main.localize_sound_source(cfg, calibration_data=calibration, use_simulation=True, show_plots=False)
            os.chdir(cwd)
        return res if full else np.asarray(res["estimated_position"], dtype=np.float64)
    finally:
        os.chdir(cwd)
        ref_main.material_properties, ref_sp.generate_signal = keep_mat, keep_gen


def save(name, **arrays):
    os.makedirs(OUT, exist_ok=True)
    np.savez_compressed(os.path.join(OUT, name), meta=META, **arrays)
    print(f"  wrote {name}: {sum(np.asarray(v).nbytes for v in arrays.values())} B raw", flush=True)


def golden_c1():
    cfg = cases.c1_config()
    sig = ref_main.simulate_signals_with_multipath(cfg["source_position"], np.array(cfg["mic_positions"]), cfg["fs"],
                                                   cases.C_SOUND, 1.0, "sine", 1000, cfg["reflective_planes"],
                                                   ref_main.material_properties, 3, 0.01)
    out = stage_outputs(sig, cfg["fs"], (0.05, None))
    out["sim_digest"] = np.array([cases.waveform_digest(s) for s in sig])
    out["position"] = run_localize(cfg)
    save("c1_example1.npz", **out)


def golden_c2():
    cfg = cases.c2_config()
    mics = np.array(cfg["mic_positions"])
    out = {}
    for tag, table in (("a_", ref_main.material_properties), ("b_", cases.LOW_LOSS)):
        imgs = ref_utils.generate_image_sources_iterative(cfg["source_position"], cfg["reflective_planes"], 3, 500,
                                                          table, mics, 0.01)
        out[tag + "images"] = np.array([i["source"] for i in imgs]).reshape(-1, 3)
        out[tag + "image_materials"] = np.array([i["material"] for i in imgs], dtype="U8")
        sig = ref_main.simulate_signals_with_multipath(cfg["source_position"], mics, 48000, cases.C_SOUND, 1.0,
                                                       "chirp", 500, cfg["reflective_planes"], table, 3, 0.01)
        out[tag + "sim_digest"] = np.array([cases.waveform_digest(s) for s in sig])
        out.update(stage_outputs(sig, 48000, (0.05, None), prefix=tag))
        out[tag + "position"] = run_localize(cfg, materials=table)
    save("c2_chirp8.npz", **out)


def golden_c3(trial=0):
    cfg = cases.c3_config(trial)
    base = cases.c3_base(trial)
    keep = ref_sp.generate_signal
    try:
        ref_sp.generate_signal = lambda *a, **k: base.copy()
        sig = ref_main.simulate_signals_with_multipath(cfg["source_position"], np.array(cfg["mic_positions"]), 48000,
                                                       cases.C_SOUND, 0.5, "noise", 1000, [],
                                                       ref_main.material_properties, 3, 0.01)
    finally:
        ref_sp.generate_signal = keep
    out = stage_outputs(sig, 48000, (0.05, None))
    out["sim_digest"] = np.array([cases.waveform_digest(s) for s in sig])
    out["position"] = run_localize(cfg, base=base)
    save(f"c3_grid64_trial{trial}.npz", **out)


def golden_c4(mics=12):
    frames = cases.c4_frames(mics)
    out = {"frames_digest": np.array([cases.waveform_digest(s) for s in frames])}
    for med in (0.05, None):
        tag = "none" if med is None else "0p05"
        for k, v in pair_table(list(frames), 96000, med).items():
            out[f"{k}_{tag}"] = v
    save("c4_sphere_first12.npz", **out)


def golden_c5(frames=(0, 1)):
    mics = cases.grid_array_64()
    out = {}
    keep = ref_sp.generate_signal
    for f in frames:
        base = cases.c5_base(f)
        try:
            ref_sp.generate_signal = lambda *a, **k: base.copy()
            sig = ref_main.simulate_signals_with_multipath(cases.c5_source(f), mics, 48000, cases.C_SOUND, 0.25,
                                                           "noise", 1000, cases.DEFAULT_PLANES, cases.LOW_LOSS, 3, 0.01)
        finally:
            ref_sp.generate_signal = keep
        out[f"f{f}_sim_digest"] = np.array([cases.waveform_digest(s) for s in sig])
        out.update(stage_outputs(sig, 48000, (0.05,), prefix=f"f{f}_"))
    save("c5_stream_frames01.npz", **out)


def golden_metric(mics=24):
    """first 24 microphones of frame 0 of the metric workload: the 276 pairs bench.py checks inside every run"""
    frames = cases.metric_frames(1, mics)[0]
    out = {}
    for med in (0.05, None):
        tag = "none" if med is None else "0p05"
        for k, v in pair_table(list(frames), 44100, med).items():
            out[f"{k}_{tag}"] = v
    save("metric_44k1_first24.npz", **out)


def golden_selection_edges():
    """Reference outputs for oracle/cases.selection_edge_cases (every branch of the fallback chain)."""
    rows = []
    for case in cases.selection_edge_cases():
        td, corr, lags = ref_utils.get_time_delays_phat(case["a"], case["b"], case["fs"], num_peaks=1,
                                                        threshold_method=case["method"],
                                                        threshold_multiplier=case["mult"], max_expected_delay=case["med"])
        n2 = len(case["b"])
        rows.append([case["t"], int(round(td[0] * case["fs"])) + n2 - 1, np.max(corr), np.min(corr), int(np.argmax(corr)),
                     ref_utils.compute_snr(corr), ref_utils.compute_peak_to_peak_ratio(corr)])
    save("selection_edges.npz", rows=np.array(rows, dtype=np.float64))


def golden_filters():
    rng = np.random.default_rng(21)
    x = rng.standard_normal(4000)
    out = {}
    for fs in (44100, 48000, 96000):
        out[f"butter_{fs}"] = ref_sp.noise_reduction(x, fs, method="butterworth")
    out["fir_48000"] = ref_sp.noise_reduction(x, 48000, method="fir")
    out["wiener"] = ref_sp.noise_reduction(x, 48000, method="wiener")
    out["fracdelay"] = ref_sp.fractional_delay(x, 0.00123, 48000)
    out["compress"] = ref_sp.dynamic_range_compression(x)
    save("filters.npz", **out)


def golden_images():
    mics = np.random.default_rng(2).uniform(-0.5, 0.5, (8, 3))
    shoebox = cases.SHOEBOX
    out = {}
    for order in (1, 2, 3):
        imgs = ref_utils.generate_image_sources_iterative([1.0, 2.0, 0.5], shoebox, order, 500, cases.LOW_LOSS, mics, 0.01)
        out[f"shoebox_o{order}"] = np.array([i["source"] for i in imgs]).reshape(-1, 3)
        out[f"shoebox_o{order}_mat"] = np.array([i["material"] for i in imgs], dtype="U8")
    for f in (0.01, 0.1, 0.25, 1.0):
        imgs = ref_utils.generate_image_sources_iterative([1.0, 2.0, 0.5], cases.DEFAULT_PLANES, 3, f,
                                                          ref_main.material_properties, mics, 0.01)
        out["default_f%s" % str(f).replace(".", "p")] = np.array([i["source"] for i in imgs]).reshape(-1, 3)
    save("image_sources.npz", **out)


def golden_calibration():
    """calibration.py end to end (run_calibration) with the global NumPy RNG seeded in front of the noise draws."""
    import calibration as ref_cal
    out = {}
    for tag, cfg, seed in cases.calibration_cases():
        np.random.seed(seed)
        results, calib, recs = ref_cal.run_calibration(cfg)
        out[f"{tag}_delay"] = np.array([r["delay"] for r in results])
        out[f"{tag}_amplitude"] = np.array([r["amplitude"] for r in results])
        out[f"{tag}_calib_head"] = np.asarray(calib[:64])
        out[f"{tag}_calib_digest"] = cases.waveform_digest(np.asarray(calib))
        out[f"{tag}_rec_digest"] = np.array([cases.waveform_digest(np.asarray(r)) for r in recs])
    save("calibration.npz", **out)


def golden_localize_extras():
    """main.py:147-157,209-222: calibration correction and per-pair correlation metrics inside localize_sound_source
    (global NumPy RNG seeded right before the call), and synchronize_signals_improved on unequal-length signals."""
    out = {}
    cfg = cases.loc_config(False)
    sig = ref_main.simulate_signals_with_multipath(cfg["source_position"], np.array(cfg["mic_positions"]), cfg["fs"],
                                                   cases.C_SOUND, cfg["duration"], "chirp", cfg["freq"], [],
                                                   ref_main.material_properties, 3, 0.01)
    out.update(stage_outputs(sig, cfg["fs"], (0.05,), prefix="loc_"))
    out["loc_sim_digest"] = np.array([cases.waveform_digest(s) for s in sig])
    out["loc_position_plain"] = run_localize(cfg)
    out["loc_position_calib"] = run_localize(cfg, calibration=cases.LOC_CALIBRATION)
    res = run_localize(cases.loc_config(True), calibration=cases.LOC_CALIBRATION, seed=cases.LOC_SEED, full=True)
    out["loc_position_metrics"] = np.asarray(res["estimated_position"], dtype=np.float64)
    keys = sorted(res["correlation_metrics"])
    out["loc_metric_pairs"] = np.array(keys, dtype=np.int32)
    out["loc_ptp"] = np.array([res["correlation_metrics"][k]["peak_to_peak_ratio"] for k in keys])
    out["loc_snr"] = np.array([res["correlation_metrics"][k]["snr"] for k in keys])
    out["loc_significant"] = np.array([bool(res["correlation_metrics"][k]["significant"]) for k in keys])
    # mismatched calibration length: ignored with a warning (main.py:148-150) -> the plain position
    out["loc_position_badcalib"] = run_localize(cfg, calibration=cases.LOC_CALIBRATION[:3])
    sigs, fs = cases.unequal_sync_signals()
    synced = ref_utils.synchronize_signals_improved(sigs, fs)
    out["uneq_len"] = np.array([len(s) for s in synced])
    out["uneq_digest"] = np.array([cases.waveform_digest(s) for s in synced])
    out["uneq_first_nonzero"] = np.array([int(np.flatnonzero(s)[0]) for s in synced])
    save("localize_extras.npz", **out)


def golden_sensitivity():
    """The reference against ITSELF: main.py:165-298 with the simulated signals of C1 / C2a / C2b / C3 moved by one unit in the
    last place (every sample to its neighbouring double, direction drawn from default_rng(77)).  The Butterworth prefilter
    (signal_processing.py:127-128) amplifies that, PHAT whitening promotes it: the fraction of selected indices that
    change and the distance the estimated position moves are the reference's OWN spread - the bound an implementation
    whose simulated signals are not bit-identical to the reference's can be held to (tests/test_gpu_localize.py)."""
    out = {}
    todo = [("c1", cases.c1_config(), None, None, "c1_example1.npz", ""),
            ("c2a", cases.c2_config(), None, None, "c2_chirp8.npz", "a_"),
            ("c2b", cases.c2_config(), cases.LOW_LOSS, None, "c2_chirp8.npz", "b_"),
            ("c3", cases.c3_config(0), None, cases.c3_base(0), "c3_grid64_trial0.npz", "")]
    keep_sim = ref_main.simulate_signals_with_multipath
    for tag, cfg, table, base, fixture, prefix in todo:
        gold = np.load(os.path.join(OUT, fixture))
        rng = np.random.default_rng(77)

        def nudged(*a, **k):
            sig = keep_sim(*a, **k)
            return [np.nextafter(x, np.where(rng.integers(0, 2, x.size) > 0, np.inf, -np.inf)) for x in sig]

        keep_mat, keep_gen = ref_main.material_properties, ref_sp.generate_signal
        try:
            if table is not None:
                ref_main.material_properties = table
            if base is not None:
                ref_sp.generate_signal = lambda *a, **k: base.copy()
            sig = nudged(cfg["source_position"], np.array(cfg["mic_positions"]), cfg["fs"], cases.C_SOUND, cfg["duration"],
                         cfg["signal_type"], cfg["freq"], cfg["reflective_planes"], ref_main.material_properties, 3, 0.01)
        finally:
            ref_main.material_properties, ref_sp.generate_signal = keep_mat, keep_gen
        st = stage_outputs(sig, cfg["fs"], (0.05,))
        k0, k1 = gold[prefix + "k_sel_0p05"], st["k_sel_0p05"]
        out[tag + "_rows"] = np.array([k0.size])
        out[tag + "_rows_differ"] = np.array([int(np.count_nonzero(k0 != k1))])
        out[tag + "_k_sel_nudged"] = k1
        rng = np.random.default_rng(77)                     # the same nudge inside localize_sound_source
        ref_main.simulate_signals_with_multipath = nudged
        try:
            pos = run_localize(cfg, materials=table, base=base)
        finally:
            ref_main.simulate_signals_with_multipath = keep_sim
        out[tag + "_position_nudged"] = pos
        out[tag + "_position_delta_m"] = np.array([float(np.linalg.norm(pos - gold[prefix + "position"]))])
        print(f"  {tag}: {out[tag + '_rows_differ'][0]}/{k0.size} selected indices change, position moves {out[tag + '_position_delta_m'][0]:.3e} m", flush=True)
    save("sensitivity.npz", **out)


if __name__ == "__main__":
    todo = sys.argv[1:] or ["edges", "filters", "images", "c1", "c2", "metric", "c4", "c5", "c3", "calibration", "extras", "sensitivity"]
    table = {"sensitivity": golden_sensitivity, "extras": golden_localize_extras, "calibration": golden_calibration, "edges": golden_selection_edges, "filters": golden_filters, "images": golden_images, "c1": golden_c1,
             "c2": golden_c2, "c3": golden_c3, "c4": golden_c4, "c5": golden_c5, "metric": golden_metric}
    for key in todo:
        t0 = time.time()
        print(f"[{key}]", flush=True)
        table[key]()
        print(f"  {time.time() - t0:.1f} s", flush=True)
