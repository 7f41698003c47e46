"""Device-resident stage chain of localize_sound_source for many frames (main.py:165-204): simulate -> synchronise ->
prefilter -> all pairs with every waveform staying in HBM.  This is the streaming configuration of BASELINE.json
(64 microphones x 1024 frames of 0.25 s, multipath simulation on): the host supplies the base signals, the per-frame
path tables and the filter design, reads back five numbers per row for the synchronisation (the 5-point spline and the
integer pads of utils.py:428-451 are host scalar work) and receives the TDOA tables.

Frame lengths follow the reference: the simulated length int((duration + longest path delay) * fs) (main.py:102) and the
synchronised length N + (max shift - min shift) (utils.py:448-456) differ from frame to frame, and with them
the exact DFT lengths.  Frames are therefore grouped by those lengths and every group runs as one batched engine call;
the results are the staged host path's (simulate_signals_with_multipath -> synchronize_signals_improved ->
noise_reduction -> pair table), bit for bit - tests/test_gpu_stream.py."""
from __future__ import annotations

from collections import defaultdict
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

from ._ffi import RECORD
from .engine import Engine, default_engine, make_params
from .signal_processing import _filter_design
from .utils import sync_shifts_batch


def tdoa_stream(bases: Sequence[np.ndarray], delays: Sequence[np.ndarray], gains: Sequence[np.ndarray], fs: float,
                totals: Sequence[int], trim_len: int, filter_method: str = "butterworth",
                max_expected_delay: Optional[float] = None, engine: Optional[Engine] = None,
                frames_per_batch: int = 128, timings: Optional[Dict[str, float]] = None) -> Tuple[np.ndarray, np.ndarray]:
    """bases[F][nbase], delays / gains[F][M][K], totals[F] (main.py:102) -> (tables[F][P], lengths[F]).

    ``trim_len`` = int(duration * fs) (main.py:119-120).  ``frames_per_batch`` bounds the HBM held by one batch
    (waveforms of a batch: 3 buffers of frames x M x L doubles).  ``timings`` (diagnostics): a dict that receives the
    seconds spent per stage; the device is synchronised at every stage boundary while it is given."""
    import time
    eng = engine or default_engine()
    clock = [time.perf_counter()]

    def lap(stage: str) -> None:
        if timings is None:
            return
        eng.synchronize()
        now = time.perf_counter()
        timings[stage] = timings.get(stage, 0.0) + now - clock[0]
        clock[0] = now

    nf = len(bases)
    if not (len(delays) == len(gains) == len(totals) == nf) or nf == 0:
        raise ValueError("one base, path table and total length per frame")
    m, k = np.asarray(delays[0]).shape
    npairs = m * (m - 1) // 2
    prm = make_params(fs, 1, "median", 1.0, max_expected_delay)
    if filter_method in ("butterworth", "fir"):
        design = _filter_design(fs, filter_method, 300, 3400, 101)
    elif filter_method == "wiener":
        design = None
    else:
        raise ValueError("Unknown filter method. Available methods: 'butterworth', 'fir', 'wiener'")
    tables = np.zeros((nf, npairs), dtype=RECORD)
    lengths = np.zeros(nf, dtype=np.int64)

    def out_len_of(total: int) -> int:
        return trim_len if 0 < trim_len < total else total

    by_out: Dict[int, List[int]] = defaultdict(list)                       # frames whose simulated rows have one length
    for f in range(nf):
        by_out[out_len_of(int(totals[f]))].append(f)
    # Every distinct simulated length and every synchronised length is a transform plan (8-16 MB each), and the batches visit them in
    # the same order again and again - the worst case for a least-recently-used cache that is smaller than the working set.
    # The bound follows the workload: simulated lengths + as many synchronised lengths + the correlations of the
    # synchronisation, with room to spare.
    eng.set_max_plans(min(4096, max(64, 3 * len(set(int(t) for t in totals)) + 32)))
    for out_len, members in by_out.items():
        for at in range(0, len(members), frames_per_batch):
            group = members[at: at + frames_per_batch]
            b = len(group)
            d_sim = eng.alloc(b * m * out_len * 8)
            try:
                # ---- simulate (main.py:165): frames that share the transform length 2 * total go through one call
                by_total: Dict[Tuple[int, int], List[int]] = defaultdict(list)
                for q, f in enumerate(group):
                    by_total[(int(totals[f]), len(bases[f]))].append(q)
                for (total, nbase), local in by_total.items():
                    base = np.ascontiguousarray([bases[group[q]] for q in local], dtype=np.float64)
                    dl = np.ascontiguousarray([delays[group[q]] for q in local], dtype=np.float64)
                    gn = np.ascontiguousarray([gains[group[q]] for q in local], dtype=np.float64)
                    d_base, d_dl, d_gn = eng.alloc(base.nbytes), eng.alloc(dl.nbytes), eng.alloc(gn.nbytes)
                    contiguous = local == list(range(local[0], local[0] + len(local)))
                    d_part = d_sim + local[0] * m * out_len * 8 if contiguous else eng.alloc(len(local) * m * out_len * 8)
                    try:
                        eng.upload(d_base, base); eng.upload(d_dl, dl); eng.upload(d_gn, gn)
                        eng.simulate_multipath_dev(d_base, len(local), nbase, fs, total, d_dl, d_gn, m, k, trim_len, d_part)
                        if not contiguous:                    # frames of this length are scattered over the batch: move them home
                            zero = np.zeros(m, dtype=np.int32)
                            for i, q in enumerate(local):
                                eng.align_rows_dev(d_part + i * m * out_len * 8, m, out_len, zero, out_len, d_sim + q * m * out_len * 8)
                        eng.synchronize()
                    finally:
                        eng.free(d_base); eng.free(d_dl); eng.free(d_gn)
                        if not contiguous:
                            eng.free(d_part)
                lap("simulate")
                ref, kpk, win, pk, refpk = eng.sync_measure_dev(d_sim, b, m, out_len)                          # utils.py:413-427
                lap("sync_measure")
                shifts = sync_shifts_batch(kpk, win, pk, refpk, ref, out_len, fs)                             # utils.py:428-446
                pads = np.maximum(0, np.rint(shifts - shifts.min(axis=1, keepdims=True))).astype(np.int32)    # utils.py:448-451
                lap("host_spline")
                # one buffer for the whole batch: the frames of one synchronised length L sit together ([frames][M][L]) so
                # that each length is one pair-table call, and ALL rows go through the prefilter in ONE launch (one lane
                # per row: a launch takes as long for 64 rows as for 65 536)
                by_len: Dict[int, List[int]] = defaultdict(list)
                for q in range(b):
                    by_len[out_len + int(pads[q].max())].append(q)
                region, at_d = {}, 0
                for length, local in by_len.items():
                    region[length] = at_d
                    at_d += len(local) * m * length
                d_al, d_flt = eng.alloc(at_d * 8), eng.alloc(at_d * 8)
                d_tab = eng.alloc(b * npairs * RECORD.itemsize)
                try:
                    offs, lens = [], []
                    for length, local in by_len.items():
                        base_d = region[length]
                        nb = len(local)
                        if local == list(range(local[0], local[0] + nb)):
                            eng.align_rows_dev(d_sim + local[0] * m * out_len * 8, nb * m, out_len, pads[local].reshape(-1), length,
                                               d_al + base_d * 8)
                        else:
                            for i, q in enumerate(local):
                                eng.align_rows_dev(d_sim + q * m * out_len * 8, m, out_len, pads[q], length,
                                                   d_al + (base_d + i * m * length) * 8)
                        offs.extend(base_d + k * length for k in range(nb * m))
                        lens.extend([length] * (nb * m))
                    lap("align")
                    if design is None:                                                                         # main.py:191
                        for length, local in by_len.items():
                            eng.wiener3_dev(d_al + region[length] * 8, len(local) * m, length, d_flt + region[length] * 8)
                    else:
                        eng.filtfilt_ragged_dev(design[0], design[1], design[2], d_al, d_flt, offs, offs, lens)
                    lap("prefilter")
                    row0 = 0
                    for length, local in by_len.items():                                                       # main.py:202-228
                        eng.gcc_phat_all_pairs_dev(d_flt + region[length] * 8, len(local), m, length, prm, d_tab + row0 * npairs * RECORD.itemsize)
                        row0 += len(local)
                    eng.synchronize()
                    lap("pairs")
                    got = np.zeros((b, npairs), dtype=RECORD)
                    eng.download(got, d_tab)
                    row0 = 0
                    for length, local in by_len.items():
                        for i, q in enumerate(local):
                            tables[group[q]] = got[row0 + i]
                            lengths[group[q]] = length
                        row0 += len(local)
                    lap("download")
                finally:
                    eng.free(d_al); eng.free(d_flt); eng.free(d_tab)
            finally:
                eng.free(d_sim)
    return tables, lengths
