/* pal_hip.h - C ABI of the MI355X GCC-PHAT / multipath engine (libpal_hip.so).
 *
 * The reference (zeynelacikgoez/PyAudioLocalization) is pure Python and has no FFI of its own;
 * its boundary for this path is the set of module-level functions cited at each entry point
 * below (file:line into the reference).  A maintainer binds these symbols with ctypes - the
 * stub is shown in INTEGRATION.md and shipped as pyaudiolocalization_amd/_ffi.py.
 *
 * Conventions
 *   - every call returns 0 (PAL_OK) or a negative PAL_ERR_* code; pal_last_error() gives the text;
 *   - plain pointers and sizes only; arrays are C-contiguous, row-major, float64 unless noted;
 *   - "host" entry points take host pointers and return when the result is in the caller's
 *     buffers; "_dev" entry points take device (HBM) pointers obtained from pal_device_alloc(),
 *     enqueue on the engine's HIP stream and return immediately - pal_synchronize() waits;
 *   - one engine per HIP device; calls on one handle must be serialised by the caller;
 *   - the caller owns every buffer it passes; the engine owns plans and workspaces it allocates;
 *   - all arithmetic is float64/complex128 like the reference's NumPy path.
 */
#ifndef PAL_HIP_H
#define PAL_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PAL_ABI_VERSION 1

#define PAL_OK 0
#define PAL_ERR_INVALID (-1)     /* bad argument (the Python shim raises ValueError)          */
#define PAL_ERR_HIP (-2)         /* HIP runtime / launch failure                              */
#define PAL_ERR_NOMEM (-3)       /* device or host allocation failed                          */
#define PAL_ERR_UNSUPPORTED (-4) /* size outside the engine's range                           */
#define PAL_ERR_INTERNAL (-5)    /* a kernel reported an internal overflow (see last error)   */
#define PAL_ERR_COMM (-6)        /* RCCL failure                                              */
#define PAL_ERR_MATERIAL (-7)    /* pal_image_sources reached a plane whose material is undefined
                                    (utils.py:93-96); *count holds the plane index            */

#define PAL_MAX_PEAKS 256

/* branch bits of the peak-selection fallback chain (utils.py:153-172) */
#define PAL_BR_ALT_THRESHOLD 1
#define PAL_BR_ARGMAX_NO_PEAKS 2
#define PAL_BR_WINDOW_RETRY 4
#define PAL_BR_ARGMAX_WINDOW 8

typedef struct pal_engine* pal_handle;

/* arguments of get_time_delays_phat (utils.py:121-127) that reach the device */
typedef struct pal_phat_params {
  double fs;                   /* sampling rate                                              */
  double threshold_multiplier; /* utils.py:126                                               */
  double max_expected_delay;   /* seconds; NaN = None (utils.py:127,162)                     */
  int32_t threshold_method;    /* 0 = 'median' and any unknown string, 1 = 'adaptive'        */
  int32_t peak_distance;       /* int(fs * 0.001), computed by the caller (utils.py:151)     */
  int32_t num_peaks;           /* 1..PAL_MAX_PEAKS (utils.py:124)                            */
  int32_t reserved;
} pal_phat_params;

/* one row of the TDOA table: everything main.py:204-225 and utils.py:228-250 read off `corr` */
typedef struct pal_pair_record {
  int32_t k_sel;      /* selected array index k; reported lag = k - (n2-1)      */
  int32_t branch;     /* PAL_BR_* bits                                                      */
  int32_t k_argmax;   /* np.argmax(corr)                                                    */
  int32_t n_sel;      /* number of selected peaks (<= num_peaks)                            */
  double cmax;        /* np.max(corr)  (main.py:223)                                        */
  double cmin;        /* np.min(corr)  (utils.py:233)                                       */
  double snr;         /* compute_snr   (utils.py:238-250); +inf when the noise std is 0     */
  double sel_height;  /* corr[k_sel]                                                        */
} pal_pair_record;

/* ---- lifetime ------------------------------------------------------------------------ */
int pal_abi_version(void);
int pal_create(int device, pal_handle* out);
void pal_destroy(pal_handle h);
const char* pal_last_error(pal_handle h); /* h may be NULL: error of the last failed pal_create */
int pal_synchronize(pal_handle h);
/* transforms processed per launch group (workspace = chunk * M * 16 B); 0 keeps the default: 128, and for the pair
 * pipeline (two pairs per transform) 240 where one workspace slot stays below 1 GiB, 32 at least */
int pal_set_chunk(pal_handle h, int chunk);
/* Plans (chirps, chirp spectra, prime-factor tables: a few MB per transform length) are built on first use and cached
 * per length, at most 64 of them (PAL_MAX_PLANS, pal_set_max_plans), least recently used out first; pal_clear_plans drops them all now
 * (after draining the engine's streams).  The reference keeps no such state (numpy.fft plans are per call). */
int pal_clear_plans(pal_handle h);
/* Bound of the plan caches (2 .. 4096; 8-16 MB of HBM per plan).  A stream whose frames come in more distinct lengths than the
 * bound, visited cyclically, would rebuild a plan on every visit (least recently used out first is the worst case for a
 * cycle): stream.tdoa_stream raises the bound to the number of lengths of its batch.  pal_plan_stats: plans built and plans
 * evicted since the engine was created (a run that rebuilds plans shows it there). */
int pal_set_max_plans(pal_handle h, int max_plans);
int pal_plan_stats(pal_handle h, int64_t* built, int64_t* evicted);
/* packed transforms (pairs / 2) one launch group of the all-pairs pipeline carries for frames of L samples */
int pal_pair_group_size(pal_handle h, int L, int32_t* transforms);

/* ---- device buffers (so that a host language needs no HIP binding of its own) ---------- */
int pal_device_alloc(pal_handle h, size_t bytes, void** dptr);
int pal_device_free(pal_handle h, void* dptr);
int pal_upload(pal_handle h, void* dptr, const void* host, size_t bytes);
int pal_download(pal_handle h, void* host, const void* dptr, size_t bytes);

/* ---- hot path A: all-pairs GCC-PHAT + peak selection ----------------------------------
 * Replaces the pair loop main.py:202-228 with get_time_delays_phat (utils.py:121-181),
 * phat_correlation (utils.py:108-119), compute_snr / compute_peak_to_peak_ratio
 * (utils.py:228-250) and np.max(corr) (main.py:223) for every i<j of every trial.
 * frames[B][M][L] -> table[B][P], P = M(M-1)/2, pairs in row-major i<j order.
 * corr (optional, may be NULL) receives [B][P][2L-1]. */
int pal_gcc_phat_all_pairs(pal_handle h, const double* frames, int B, int M, int L, const pal_phat_params* prm,
                           pal_pair_record* table, double* corr);
int pal_gcc_phat_all_pairs_dev(pal_handle h, const double* d_frames, int B, int M, int L, const pal_phat_params* prm,
                               pal_pair_record* d_table);
/* Ordering of the _dev forms: everything is enqueued on the engine's streams and the results are complete after
 * pal_synchronize.  With one peak per row (num_peaks = 1) and no corr the column pass finishes the rows itself
 * (csrc/pfa_cols_fin.h); the call then reads ONE integer back at its end (the number of rows that need the stored-row
 * path - a tie, a peak at the lag window's edge: typically 0.1 % - which it resolves before returning), i.e. it blocks
 * the host for the duration of its own device work.  PAL_FIN=0 keeps the call fully asynchronous (stored rows + finish launch). */
/* Non-finite samples: the reference confines a NaN to the pairs of its own microphone.  Here two pairs share one complex
 * transform, so a frame with a NaN or an infinity makes the batched calls (all_pairs, pairs) fail with PAL_ERR_INVALID
 * (reported by the call itself, or by pal_synchronize for the _dev forms) instead of returning rows that differ from the
 * reference's.  The single-pair entry points below have no partner pair and behave like the reference. */

/* explicit pair list over rows[R][L]: pairs[P][2] row indices -> table[P].  Carries the 1000 shuffled
 * correlations per pair of bootstrap_significance (utils.py:183-216) as one call (rows = sig1 + shuffles of sig2,
 * pairs = (0, k)); only cmax of each record is used there. */
int pal_gcc_phat_pairs(pal_handle h, const double* rows, int R, int L, const int32_t* pairs, int64_t P,
                       const pal_phat_params* prm, pal_pair_record* table);

/* the same with the rows, the pair list (int32 [P][2]) and the table in HBM; asynchronous like every _dev entry point.
 * This is what a rank calls for its contiguous block of the ordered pair list when ONE large frame (256 microphones,
 * 32 640 pairs) is split over the GPUs of a node: every rank holds the frame and recomputes the spectra locally.
 * A row index outside 0..R-1 is reported as PAL_ERR_INVALID by pal_synchronize. */
int pal_gcc_phat_pairs_dev(pal_handle h, const double* d_rows, int R, int L, const int32_t* d_pairs, int64_t P,
                           const pal_phat_params* prm, pal_pair_record* d_table);

/* single-pair signatures: phat_correlation(sig1, sig2) (utils.py:108) -> corr[n1+n2-1] */
int pal_phat_correlation(pal_handle h, const double* sig1, int n1, const double* sig2, int n2, double* corr);
/* get_time_delays_phat (utils.py:121): corr[n1+n2-1] (may be NULL), k_out[num_peaks] array indices */
int pal_get_time_delays_phat(pal_handle h, const double* sig1, int n1, const double* sig2, int n2,
                             const pal_phat_params* prm, int32_t* k_out, pal_pair_record* rec, double* corr);
/* compute_snr / compute_peak_to_peak_ratio / max on an existing correlation row (utils.py:228-250) */
int pal_corr_metrics(pal_handle h, const double* corr, int n, pal_pair_record* rec);

/* ---- hot path B: image-source multipath simulation ------------------------------------
 * generate_image_sources_iterative (utils.py:67-106): host C++, discovery order preserved.
 * planes[K][4], material_id[K] index into absorption[]/freq_coeff[] (n_materials entries, -1 = undefined);
 * images[cap][3], image_material[cap]; *count receives the number found (PAL_ERR_UNSUPPORTED if > cap). */
int pal_image_sources(const double* source, const double* planes, const int32_t* material_id, int K,
                      const double* absorption, const double* freq_coeff, int n_materials, int max_order,
                      double frequency, const double* mics, int M, double threshold, int round_decimals,
                      double* images, int32_t* image_material, int cap, int* count);
/* simulate_signals_with_multipath (main.py:103-123) after the path geometry is known:
 * base[B][nbase] zero-padded to total_samples, delays/gains[B][M][K] (seconds, linear gain, fp64),
 * out[B][M][out_len] with out_len = trim_len > 0 ? trim_len : total_samples; fractional_delay
 * (signal_processing.py:66-80), normalize_signal + dynamic_range_compression (:82-94) fused. */
int pal_simulate_multipath(pal_handle h, const double* base, int B, int nbase, double fs, int total_samples,
                           const double* delays, const double* gains, int M, int K, int trim_len, double* out);
/* fractional_delay(signal, delay, fs) alone (signal_processing.py:66-80); rows[R][N], delay per row */
int pal_fractional_delay(pal_handle h, const double* rows, int R, int N, const double* delays, double fs, double* out);
/* normalize_signal (normalize_only != 0) or dynamic_range_compression (signal_processing.py:82-94) */
int pal_normalize_compress(pal_handle h, const double* rows, int R, int N, int normalize_only, double threshold,
                           double epsilon, double* out);

/* ---- prefilter and synchronisation ------------------------------------------------------
 * noise_reduction (signal_processing.py:109-138): filtfilt(b, a, x) with scipy's defaults (odd
 * extension 3*max(nb,na), lfilter_zi initial state `zi`, designed on the host); rows[R][N]. */
int pal_filtfilt(pal_handle h, const double* b, int nb, const double* a, int na, const double* zi,
                 const double* rows, int R, int N, double* out);
int pal_wiener3(pal_handle h, const double* rows, int R, int N, double* out);
/* synchronize_signals_improved (utils.py:415-427): full cross-correlation of every row against row
 * ref_idx; kpk = argmax|corr| (index into the 2N-1 sequence), win5 = corr[kpk-2..kpk+2] (NaN outside),
 * pkabs = |corr[kpk]|, *refpk = max|autocorrelation of the reference|. */
int pal_xcorr_vs_ref(pal_handle h, const double* rows, int R, int N, int ref_idx, int32_t* kpk, double* win5,
                     double* pkabs, double* refpk);

/* ---- device-resident stage chain (main.py:165-204 without host copies of the waveforms) ----------------------------
 * The streaming configuration (64 microphones x 1024 frames, multipath simulation on) keeps every waveform in HBM:
 * simulate -> measure the synchronisation shifts -> align -> prefilter -> all pairs.  The host supplies the path tables
 * and the filter coefficients and reads back five numbers per row (the 5-point spline refinement and the integer pads of
 * utils.py:428-451 stay host work, as in pal_xcorr_vs_ref).  All `d_` pointers come from pal_device_alloc.
 *
 * pal_simulate_multipath_dev: as pal_simulate_multipath with d_base[B][nbase], d_delays / d_gains[B][M][K], d_out[B][M][out_len].
 * pal_row_energies_dev: energy[r] = sum of squares of row r of d_rows[R][N] (device summation order: last-bit differences
 *   from numpy's np.sum(sig**2) - the caller settles near-ties of utils.py:413-414 with numpy itself, engine.sync_measure_dev).
 * pal_sync_measure_dev: per frame b of d_rows[B][M][N]: ref_idx[b] (IN / OUT: >= 0 on entry = use this reference microphone;
 *   < 0 = argmax of the device's row energies, utils.py:413-414), then
 *   pal_xcorr_vs_ref of the frame's rows against that row: kpk[B][M], win5[B][M][5], pkabs[B][M], refpk[B] (host arrays).
 * pal_align_rows_dev: utils.py:448-456 - d_out[r][pad_left[r] + i] = d_rows[r][i], zeros elsewhere (rows of Lout samples).
 * pal_filtfilt_dev / pal_wiener3_dev: as pal_filtfilt / pal_wiener3 on rows in HBM (b, a, zi stay host arrays). */
int pal_simulate_multipath_dev(pal_handle h, const double* d_base, int B, int nbase, double fs, int total_samples,
                               const double* d_delays, const double* d_gains, int M, int K, int trim_len, double* d_out);
int pal_row_energies_dev(pal_handle h, const double* d_rows, int R, int N, double* energy);
int pal_sync_measure_dev(pal_handle h, const double* d_rows, int B, int M, int N, int32_t* ref_idx, int32_t* kpk,
                         double* win5, double* pkabs, double* refpk);
int pal_align_rows_dev(pal_handle h, const double* d_rows, int R, int N, const int32_t* pad_left, int Lout, double* d_out);
int pal_filtfilt_dev(pal_handle h, const double* b, int nb, const double* a, int na, const double* zi, const double* d_rows,
                     int R, int N, double* d_out);
int pal_wiener3_dev(pal_handle h, const double* d_rows, int R, int N, double* d_out);
/* the same filter over R rows of DIFFERENT lengths in one launch (the frames of a stream have their own synchronised
 * lengths, utils.py:448-456): row r is d_in[in_off[r] .. + lengths[r]) -> d_out[out_off[r] .. + lengths[r]) (host arrays
 * of offsets in doubles and lengths).  One lane per row, 64 rows per wavefront: a launch takes the same time for 64 rows
 * as for 65 536, so the caller batches. */
int pal_filtfilt_ragged_dev(pal_handle h, const double* b, int nb, const double* a, int na, const double* zi, const double* d_in,
                            double* d_out, int R, const int64_t* in_off, const int64_t* out_off, const int32_t* lengths);

/* ---- multi-GPU: one gather of the TDOA table over RCCL/xGMI ------------------------------ */
int pal_comm_unique_id(void* id128);                       /* rank 0; 128-byte ncclUniqueId          */
int pal_comm_init(pal_handle h, int nranks, int rank, const void* id128);
int pal_comm_all_gather(pal_handle h, const void* d_send, void* d_recv, size_t bytes_per_rank);
int pal_comm_destroy(pal_handle h);

/* ---- measurement ---------------------------------------------------------------------------
 * HIP-event timing of the engine's own kernels on its own stream.  Between pal_profile_begin and
 * pal_profile_end every launch is bracketed by events; pal_profile_get returns the summed
 * duration (ms) and launch count of the kernel class `name`, spelled like the kernel's template
 * instance ("k_rows<10,conv>", "k_pfa_rows<11>", "k_peak_stream", ...).  The events cost a few microseconds of
 * idle stream each: pal_profile_sampling(h, every) brackets only every `every`-th launch group of the pair
 * pipeline (averages are unchanged, the run is barely perturbed); the default is 1 = every group. */
int pal_profile_begin(pal_handle h);
int pal_profile_end(pal_handle h);
int pal_profile_sampling(pal_handle h, int every);
int pal_profile_get(pal_handle h, const char* name, double* total_ms, int64_t* launches);
/* enumerate the kernel classes seen so far: index 0.. until PAL_ERR_INVALID */
int pal_profile_entry(pal_handle h, int index, char* name, int cap, double* total_ms, int64_t* launches);
/* transform geometry of the all-pairs plan for frames of L samples: n = 2L-1, conv length M, M1, M2 */
int pal_plan_info(pal_handle h, int L, int32_t* n, int32_t* conv_len, int32_t* m1, int32_t* m2);

/* Prime-factor route of the inverse transform for frame length L: n = n1 * n2 (coprime, odd) with the n2-point
 * DFTs as in-LDS convolutions of `tile_len` points: a chirp convolution (tile_len a power of two >= 2 n2 - 1) or,
 * for n2 = 991, Rader's cyclic convolution (tile_len = n2 - 1 = 990).  All three are 0 when n has no such split
 * (or PAL_PFA=0) and the four-step chirp convolution reported by pal_plan_info does the inverse too. */
int pal_plan_factors(pal_handle h, int L, int32_t* n1, int32_t* n2, int32_t* tile_len);

#ifdef __cplusplus
}
#endif
#endif /* PAL_HIP_H */
