"""Python face of one libpal_hip engine (one HIP device, one stream).

Host arrays in, host arrays out for the drop-in functions; ``*_dev`` methods work on buffers
that stay resident in HBM (``alloc`` / ``upload`` / ``download``).  All arithmetic happens in the
HIP kernels behind the C ABI - nothing here computes a result on the CPU.
"""
from __future__ import annotations

import ctypes as C
import math
import os
from typing import Optional, Sequence, Tuple

import numpy as np

from . import _ffi
from ._ffi import RECORD, PalError, PhatParams, f64, ptr


def _method_code(threshold_method: str) -> int:
    # utils.py:144-149: 'adaptive' is the only alternative, every other string means 'median'
    return 1 if threshold_method == "adaptive" else 0


def make_params(fs, num_peaks=1, threshold_method="median", threshold_multiplier=1.0,
                max_expected_delay=None) -> PhatParams:
    p = PhatParams()
    p.fs = float(fs)
    p.threshold_multiplier = float(threshold_multiplier)
    p.max_expected_delay = math.nan if max_expected_delay is None else float(max_expected_delay)
    p.threshold_method = _method_code(threshold_method)
    p.peak_distance = int(fs * 0.001)                       # utils.py:151
    p.num_peaks = int(num_peaks)
    p.reserved = 0
    return p


def pair_list(mics: int) -> np.ndarray:
    """(P, 2) mic indices in the row-major i<j order of main.py:202-203."""
    i, j = np.triu_indices(mics, k=1)
    return np.stack([i, j], axis=1).astype(np.int32)


class Engine:
    def __init__(self, device: Optional[int] = None):
        self._lib = _ffi.load()
        if device is None:
            device = int(os.environ.get("PAL_DEVICE", "0"))
        h = C.c_void_p()
        rc = self._lib.pal_create(int(device), C.byref(h))
        if rc != 0:
            text = self._lib.pal_last_error(None).decode()
            raise PalError(rc, f"pal_create(device={device}): {text} - the HIP engine needs an AMD GPU; "
                               "there is no CPU fallback")
        self._h = h
        self.device = int(device)

    # ---- plumbing -------------------------------------------------------------------
    def close(self) -> None:
        if getattr(self, "_h", None):
            self._lib.pal_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc: int) -> None:
        if rc == 0:
            return
        text = self._lib.pal_last_error(self._h).decode()
        if rc == _ffi.ERR_INVALID:
            raise ValueError(text)
        if rc == _ffi.ERR_NOMEM:
            raise MemoryError(text)
        raise PalError(rc, text)

    def synchronize(self) -> None:
        self._check(self._lib.pal_synchronize(self._h))

    def set_chunk(self, chunk: int) -> None:
        self._check(self._lib.pal_set_chunk(self._h, int(chunk)))

    def clear_plans(self) -> None:
        """Drop every cached transform plan (they are rebuilt on next use; the cache is bounded anyway)."""
        self._check(self._lib.pal_clear_plans(self._h))

    def set_max_plans(self, max_plans: int) -> None:
        """Bound of the plan caches (least recently used out first); see include/pal_hip.h."""
        self._check(self._lib.pal_set_max_plans(self._h, int(max_plans)))

    def plan_stats(self):
        """(plans built, plans evicted) since the engine was created."""
        b, e = C.c_int64(), C.c_int64()
        self._check(self._lib.pal_plan_stats(self._h, C.byref(b), C.byref(e)))
        return b.value, e.value

    def pair_group_size(self, length: int) -> int:
        """Packed transforms (two pairs each) per launch group of the all-pairs pipeline for frames of `length` samples."""
        g = C.c_int32()
        self._check(self._lib.pal_pair_group_size(self._h, int(length), C.byref(g)))
        return g.value

    def alloc(self, nbytes: int) -> int:
        p = C.c_void_p()
        self._check(self._lib.pal_device_alloc(self._h, int(nbytes), C.byref(p)))
        return int(p.value)

    def free(self, dptr: int) -> None:
        self._check(self._lib.pal_device_free(self._h, C.c_void_p(dptr)))

    def upload(self, dptr: int, host: np.ndarray) -> None:
        host = np.ascontiguousarray(host)
        self._check(self._lib.pal_upload(self._h, C.c_void_p(dptr), host.ctypes.data, host.nbytes))

    def download(self, host: np.ndarray, dptr: int) -> np.ndarray:
        assert host.flags.c_contiguous
        self._check(self._lib.pal_download(self._h, host.ctypes.data, C.c_void_p(dptr), host.nbytes))
        return host

    def plan_info(self, frame_len: int) -> dict:
        v = [C.c_int32() for _ in range(4)]
        self._check(self._lib.pal_plan_info(self._h, int(frame_len), *[C.byref(x) for x in v]))
        f = [C.c_int32() for _ in range(3)]
        self._check(self._lib.pal_plan_factors(self._h, int(frame_len), *[C.byref(x) for x in f]))
        return {"n": v[0].value, "conv_len": v[1].value, "m1": v[2].value, "m2": v[3].value,
                "n1": f[0].value, "n2": f[1].value, "tile_len": f[2].value}

    # ---- hot path A --------------------------------------------------------------------
    def gcc_phat_all_pairs(self, frames, fs, num_peaks=1, threshold_method="median", threshold_multiplier=1.0,
                           max_expected_delay=None, want_corr=False):
        """frames[B][M][L] (or [M][L]) -> TDOA table[B][P] (structured, see _ffi.RECORD) [+ corr[B][P][2L-1]]."""
        x = f64(frames)
        squeeze = x.ndim == 2
        if squeeze:
            x = x[None]
        if x.ndim != 3:
            raise ValueError("frames must be [M][L] or [B][M][L]")
        if num_peaks != 1:
            raise ValueError("the batched table carries one peak per pair (main.py:204 calls num_peaks=1)")
        b, m, length = x.shape
        npairs = m * (m - 1) // 2
        table = np.zeros((b, npairs), dtype=RECORD)
        corr = np.empty((b, npairs, 2 * length - 1)) if want_corr else None
        prm = make_params(fs, 1, threshold_method, threshold_multiplier, max_expected_delay)
        self._check(self._lib.pal_gcc_phat_all_pairs(self._h, x.ctypes.data, b, m, length, C.byref(prm),
                                                     table.ctypes.data, ptr(corr)))
        if squeeze:
            table = table[0]
            corr = None if corr is None else corr[0]
        return (table, corr) if want_corr else table

    def gcc_phat_all_pairs_dev(self, d_frames: int, b: int, m: int, length: int, prm: PhatParams, d_table: int) -> None:
        """Asynchronous: frames and table stay in HBM; call synchronize() before reading the table."""
        self._check(self._lib.pal_gcc_phat_all_pairs_dev(self._h, C.c_void_p(d_frames), b, m, length, C.byref(prm),
                                                         C.c_void_p(d_table)))

    def gcc_phat_pairs(self, rows, pairs, fs, threshold_method="median", threshold_multiplier=1.0,
                       max_expected_delay=None) -> np.ndarray:
        """rows[R][L] and an explicit pair list pairs[P][2] (row indices) -> table[P]."""
        x = f64(rows)
        if x.ndim != 2:
            raise ValueError("rows must be [R][L]")
        pr = np.ascontiguousarray(pairs, dtype=np.int32).reshape(-1, 2)
        table = np.zeros(pr.shape[0], dtype=RECORD)
        prm = make_params(fs, 1, threshold_method, threshold_multiplier, max_expected_delay)
        self._check(self._lib.pal_gcc_phat_pairs(self._h, x.ctypes.data, x.shape[0], x.shape[1], pr.ctypes.data, pr.shape[0],
                                                 C.byref(prm), table.ctypes.data))
        return table

    def gcc_phat_pairs_dev(self, d_rows: int, r: int, length: int, d_pairs: int, p: int, prm: PhatParams, d_table: int) -> None:
        """Asynchronous: rows[R][L], pairs[P][2] (int32) and table[P] stay in HBM (a rank's block of a large frame's pair list)."""
        self._check(self._lib.pal_gcc_phat_pairs_dev(self._h, C.c_void_p(d_rows), int(r), int(length), C.c_void_p(d_pairs), int(p),
                                                     C.byref(prm), C.c_void_p(d_table)))

    def phat_correlation(self, sig1, sig2) -> np.ndarray:
        a, b = f64(sig1), f64(sig2)
        if a.ndim != 1 or b.ndim != 1:
            raise ValueError("signals must be one-dimensional")
        corr = np.empty(a.shape[0] + b.shape[0] - 1)
        self._check(self._lib.pal_phat_correlation(self._h, a.ctypes.data, a.shape[0], b.ctypes.data, b.shape[0],
                                                   corr.ctypes.data))
        return corr

    def get_time_delays_phat(self, sig1, sig2, fs, num_peaks=1, threshold_method="median", threshold_multiplier=1.0,
                             max_expected_delay=None, want_corr=True):
        """-> (selected array indices k[<=num_peaks], record, corr)."""
        a, b = f64(sig1), f64(sig2)
        if a.ndim != 1 or b.ndim != 1:
            raise ValueError("signals must be one-dimensional")
        prm = make_params(fs, num_peaks, threshold_method, threshold_multiplier, max_expected_delay)
        ks = np.full(max(1, int(num_peaks)), -1, dtype=np.int32)
        rec = np.zeros(1, dtype=RECORD)
        corr = np.empty(a.shape[0] + b.shape[0] - 1) if want_corr else None
        self._check(self._lib.pal_get_time_delays_phat(self._h, a.ctypes.data, a.shape[0], b.ctypes.data, b.shape[0],
                                                       C.byref(prm), ks.ctypes.data, rec.ctypes.data, ptr(corr)))
        return ks[: int(rec["n_sel"][0])], rec[0], corr

    def corr_metrics(self, corr) -> np.void:
        c = f64(corr)
        rec = np.zeros(1, dtype=RECORD)
        self._check(self._lib.pal_corr_metrics(self._h, c.ctypes.data, c.shape[0], rec.ctypes.data))
        return rec[0]

    # ---- hot path B --------------------------------------------------------------------
    def simulate_multipath(self, base, fs, total_samples, delays, gains, trim_len=0) -> np.ndarray:
        """base[B][nbase], delays/gains[B][M][K] -> out[B][M][out_len] (normalised + compressed)."""
        x = f64(base)
        if x.ndim == 1:
            x = x[None]
        d, g = f64(delays), f64(gains)
        if d.ndim == 2:
            d, g = d[None], g[None]
        if d.shape != g.shape or d.ndim != 3 or d.shape[0] != x.shape[0]:
            raise ValueError("delays/gains must be [B][M][K] matching base[B][nbase]")
        b, m, k = d.shape
        out_len = trim_len if 0 < trim_len < total_samples else total_samples
        out = np.empty((b, m, out_len))
        self._check(self._lib.pal_simulate_multipath(self._h, x.ctypes.data, b, x.shape[1], float(fs), int(total_samples),
                                                     d.ctypes.data, g.ctypes.data, m, k, int(trim_len), out.ctypes.data))
        return out

    def fractional_delay(self, rows, delays, fs) -> np.ndarray:
        x = f64(rows)
        one = x.ndim == 1
        if one:
            x = x[None]
        d = f64(np.atleast_1d(delays))
        if d.shape != (x.shape[0],):
            raise ValueError("one delay per row")
        out = np.empty_like(x)
        self._check(self._lib.pal_fractional_delay(self._h, x.ctypes.data, x.shape[0], x.shape[1], d.ctypes.data, float(fs),
                                                   out.ctypes.data))
        return out[0] if one else out

    def normalize_compress(self, rows, normalize_only=False, threshold=0.8, epsilon=1e-8) -> np.ndarray:
        x = f64(rows)
        one = x.ndim == 1
        if one:
            x = x[None]
        out = np.empty_like(x)
        self._check(self._lib.pal_normalize_compress(self._h, x.ctypes.data, x.shape[0], x.shape[1], int(normalize_only),
                                                     float(threshold), float(epsilon), out.ctypes.data))
        return out[0] if one else out

    def filtfilt(self, b, a, zi, rows) -> np.ndarray:
        x = f64(rows)
        one = x.ndim == 1
        if one:
            x = x[None]
        b, a, zi = f64(b), f64(a), f64(zi)
        out = np.empty_like(x)
        self._check(self._lib.pal_filtfilt(self._h, b.ctypes.data, b.shape[0], a.ctypes.data, a.shape[0], zi.ctypes.data,
                                           x.ctypes.data, x.shape[0], x.shape[1], out.ctypes.data))
        return out[0] if one else out

    def wiener3(self, rows) -> np.ndarray:
        x = f64(rows)
        one = x.ndim == 1
        if one:
            x = x[None]
        out = np.empty_like(x)
        self._check(self._lib.pal_wiener3(self._h, x.ctypes.data, x.shape[0], x.shape[1], out.ctypes.data))
        return out[0] if one else out

    def xcorr_vs_ref(self, rows, ref_idx: int):
        """Full cross-correlation of every row against row ref_idx -> (kpk[R], win5[R][5], pkabs[R], refpk)."""
        x = f64(rows)
        r = x.shape[0]
        kpk = np.zeros(r, dtype=np.int32)
        win = np.zeros((r, 5))
        pk = np.zeros(r)
        ref = C.c_double()
        self._check(self._lib.pal_xcorr_vs_ref(self._h, x.ctypes.data, r, x.shape[1], int(ref_idx), kpk.ctypes.data,
                                               win.ctypes.data, pk.ctypes.data, C.byref(ref)))
        return kpk, win, pk, ref.value

    # ---- device-resident stage chain (main.py:165-204 without host copies of the waveforms) ------
    def simulate_multipath_dev(self, d_base: int, b: int, nbase: int, fs: float, total_samples: int, d_delays: int, d_gains: int,
                               m: int, k: int, trim_len: int, d_out: int) -> None:
        self._check(self._lib.pal_simulate_multipath_dev(self._h, C.c_void_p(d_base), int(b), int(nbase), float(fs), int(total_samples),
                                                         C.c_void_p(d_delays), C.c_void_p(d_gains), int(m), int(k), int(trim_len),
                                                         C.c_void_p(d_out)))

    def sync_measure_dev(self, d_rows: int, b: int, m: int, n: int):
        """Per frame: reference microphone (highest energy) and the cross-correlation measurements of every row against it
        -> (ref_idx[B], kpk[B][M], win5[B][M][5], pkabs[B][M], refpk[B]) on the host (a few numbers per row)."""
        # The reference microphone is np.argmax of np.sum(sig**2) (utils.py:413-414).  The device sums in another order: where the two
        # highest energies of a frame agree to 1e-12 (mirrored geometries, equal rows) those rows come to the host and numpy
        # itself decides, so that the choice - and with it every shift, pad and length behind it - is the reference's.
        en = np.zeros(b * m)
        self._check(self._lib.pal_row_energies_dev(self._h, C.c_void_p(d_rows), int(b * m), int(n), en.ctypes.data))
        en = en.reshape(b, m)
        ref = np.full(b, -1, dtype=np.int32)
        for f in range(b):
            if not np.all(np.isfinite(en[f])):
                continue                                            # (NaN rows: the device's rule, as np.argmax)
            top = float(en[f].max())
            close = np.flatnonzero(en[f] >= top - 1e-12 * abs(top))
            if close.size > 1:
                exact = en[f].copy()
                row = np.empty(n)
                for q in close:
                    self.download(row, d_rows + (f * m + int(q)) * n * 8)
                    exact[q] = np.sum(row ** 2)
                exact[np.setdiff1d(np.arange(m), close)] = -np.inf
                ref[f] = int(np.argmax(exact))
        kpk = np.zeros((b, m), dtype=np.int32)
        win = np.zeros((b, m, 5))
        pk = np.zeros((b, m))
        refpk = np.zeros(b)
        self._check(self._lib.pal_sync_measure_dev(self._h, C.c_void_p(d_rows), int(b), int(m), int(n), ref.ctypes.data, kpk.ctypes.data,
                                                   win.ctypes.data, pk.ctypes.data, refpk.ctypes.data))
        return ref, kpk, win, pk, refpk

    def align_rows_dev(self, d_rows: int, r: int, n: int, pad_left, lout: int, d_out: int) -> None:
        pads = np.ascontiguousarray(pad_left, dtype=np.int32)
        if pads.shape != (r,):
            raise ValueError("one pad per row")
        self._check(self._lib.pal_align_rows_dev(self._h, C.c_void_p(d_rows), int(r), int(n), pads.ctypes.data, int(lout), C.c_void_p(d_out)))

    def filtfilt_dev(self, b, a, zi, d_rows: int, r: int, n: int, d_out: int) -> None:
        b, a, zi = f64(b), f64(a), f64(zi)
        self._check(self._lib.pal_filtfilt_dev(self._h, b.ctypes.data, b.shape[0], a.ctypes.data, a.shape[0], zi.ctypes.data,
                                               C.c_void_p(d_rows), int(r), int(n), C.c_void_p(d_out)))

    def filtfilt_ragged_dev(self, b, a, zi, d_in: int, d_out: int, in_off, out_off, lengths) -> None:
        """Rows of different lengths in one launch: row r = d_in[in_off[r] : + lengths[r]] -> d_out[out_off[r] : ...] (offsets in doubles)."""
        b, a, zi = f64(b), f64(a), f64(zi)
        io = np.ascontiguousarray(in_off, dtype=np.int64)
        oo = np.ascontiguousarray(out_off, dtype=np.int64)
        ln = np.ascontiguousarray(lengths, dtype=np.int32)
        if not (io.shape == oo.shape == ln.shape) or io.ndim != 1:
            raise ValueError("one offset pair and one length per row")
        self._check(self._lib.pal_filtfilt_ragged_dev(self._h, b.ctypes.data, b.shape[0], a.ctypes.data, a.shape[0], zi.ctypes.data,
                                                      C.c_void_p(d_in), C.c_void_p(d_out), io.shape[0], io.ctypes.data, oo.ctypes.data,
                                                      ln.ctypes.data))

    def wiener3_dev(self, d_rows: int, r: int, n: int, d_out: int) -> None:
        self._check(self._lib.pal_wiener3_dev(self._h, C.c_void_p(d_rows), int(r), int(n), C.c_void_p(d_out)))

    # ---- multi-GPU ---------------------------------------------------------------------
    @staticmethod
    def comm_unique_id() -> bytes:
        buf = C.create_string_buffer(128)
        rc = _ffi.load().pal_comm_unique_id(buf)
        if rc != 0:
            raise PalError(rc, "ncclGetUniqueId failed (librccl not loadable?)")
        return buf.raw

    def comm_init(self, nranks: int, rank: int, unique_id: bytes) -> None:
        buf = C.create_string_buffer(unique_id, 128)
        self._check(self._lib.pal_comm_init(self._h, int(nranks), int(rank), buf))

    def all_gather_dev(self, d_send: int, d_recv: int, nbytes_per_rank: int) -> None:
        self._check(self._lib.pal_comm_all_gather(self._h, C.c_void_p(d_send), C.c_void_p(d_recv), int(nbytes_per_rank)))

    def comm_destroy(self) -> None:
        self._check(self._lib.pal_comm_destroy(self._h))

    # ---- measurement ---------------------------------------------------------------------
    def profile_begin(self, every: int = 1) -> None:
        """Bracket the engine's launches with HIP events (every `every`-th launch group of the pair pipeline)."""
        self._check(self._lib.pal_profile_sampling(self._h, int(every)))
        self._check(self._lib.pal_profile_begin(self._h))

    def profile_end(self) -> None:
        self._check(self._lib.pal_profile_end(self._h))

    def profile_get(self, name: str) -> Tuple[float, int]:
        ms, cnt = C.c_double(), C.c_int64()
        self._check(self._lib.pal_profile_get(self._h, name.encode(), C.byref(ms), C.byref(cnt)))
        return ms.value, cnt.value


    def profile_entries(self) -> dict:
        """{kernel instance name: (total ms, launches)} accumulated since profile_begin()."""
        out = {}
        idx = 0
        while True:
            name = C.create_string_buffer(96)
            ms, cnt = C.c_double(), C.c_int64()
            if self._lib.pal_profile_entry(self._h, idx, name, 96, C.byref(ms), C.byref(cnt)) != 0:
                return out
            out[name.value.decode()] = (ms.value, cnt.value)
            idx += 1


def image_sources(source, planes, material_id, absorption, freq_coeff, max_order, frequency, mics, threshold=0.01,
                  round_decimals=6, cap=4096):
    """Host C++ breadth-first image-source search (utils.py:67-106) -> (images[count][3], material index[count])."""
    lib = _ffi.load()
    src = f64(source, (3,))
    pl = f64(planes).reshape(-1, 4) if len(planes) else np.zeros((0, 4))
    mid = np.ascontiguousarray(material_id, dtype=np.int32)
    ab, fc = f64(absorption), f64(freq_coeff)
    mic = f64(mics).reshape(-1, 3)
    img = np.zeros((cap, 3))
    mat = np.zeros(cap, dtype=np.int32)
    cnt = C.c_int()
    rc = lib.pal_image_sources(src.ctypes.data, pl.ctypes.data, mid.ctypes.data, pl.shape[0], ab.ctypes.data, fc.ctypes.data,
                               ab.shape[0], int(max_order), float(frequency), mic.ctypes.data, mic.shape[0],
                               float(threshold), int(round_decimals), img.ctypes.data, mat.ctypes.data, cap, C.byref(cnt))
    if rc == _ffi.ERR_INVALID:
        raise ValueError("invalid plane (a^2 + b^2 + c^2 == 0) or material index")
    if rc == _ffi.ERR_MATERIAL:
        raise KeyError(cnt.value)                              # plane index whose material is undefined
    if rc != 0:
        raise PalError(rc, f"image source capacity {cap} exceeded ({cnt.value} found)")
    return img[: cnt.value].copy(), mat[: cnt.value].copy()


_default: Optional[Engine] = None


def default_engine() -> Engine:
    """Process-wide engine used by the drop-in modules (device from PAL_DEVICE, default 0)."""
    global _default
    if _default is None:
        _default = Engine()
    return _default
