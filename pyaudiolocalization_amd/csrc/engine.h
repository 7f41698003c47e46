// engine.h - internal C++ state behind the pal_handle of include/pal_hip.h.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <map>
#include <string>
#include <tuple>
#include <vector>

#include "../../include/pal_hip.h"
#include "fft_core.h"

namespace pal {

// One circular convolution of length M = M1 x M2 (four-step, both factors in LDS).
// M2 = 2^l2; M1 = 2^l1, or 3 * 2^l1 (r3) when the 3 * 2^k length is the smaller fit.
struct PeakArgs;    // peak_types.h
struct FinArgs;     // pfa_cols_fin.h

struct Conv {
  int l1 = 0, l2 = 0;
  bool r3 = false;
  bool reg = false;              // rows of 8192 points in registers, columns of m1r points in the registers of one lane (conv_kernels.h)
  int m1r = 0;
  size_t m = 0;                  // points per transform
  cd* chat = nullptr;            // FFT_M of the chirp kernel in [k1][k2] order, pre-scaled
  cd* twA = nullptr;             // exp(-2 pi i q / M1), q < M1
  cd* twB = nullptr;             // exp(-2 pi i r / M),  r < M2
  size_t M() const { return m; }
  int M1() const { return reg ? m1r : (r3 ? 3 : 1) << l1; }
  int M2() const { return 1 << l2; }
};

// Prime-factor cut n = N1 x N2 (coprime, both odd) of the inverse PHAT transform (pfa.hip): N2-point DFTs as
// chirp convolutions of length M = 2^lm that never leave LDS, then dense N1-point DFTs down the columns.
struct Pfa {
  int n1 = 0, n2 = 0;      // N1 <= 127 (dense), N2 <= M/2 (in-LDS Bluestein)
  int lm = 0;              // log2 M: 10..12 two tiles per workgroup in LDS (pfa_kernels.h), 13..14 one register-resident tile (pfa_big.h)
  int u1 = 0;              // N2^-1 mod N1
  long long e1 = 0, e2 = 0;   // CRT idempotents: k = (e1 k1 + e2 k2) mod n
  int nch = 1;             // accumulator chunks of the column pass
  cd* b = nullptr;         // chirp exp(i pi u2 j^2 / N2), j < N2 (u2 = N1^-1 mod N2, made even)
  cd* hhat = nullptr;      // FFT_M of the conjugate chirp kernel, scaled by 1 / (M n)
  cd* r1 = nullptr;        // exp(-2 pi i q / N1), q < N1
  double* T = nullptr;     // cos / sin (2 pi j t / N1) in the column pass's chunked order
  int2* rowtab = nullptr;  // per row of Y: (u1 row mod N1, its step between outputs of a last-stage butterfly)
  // Rader variant of the row pass (pfa_rader.h) when N2 is prime and N2 - 1 = 11 x 9 x 10
  bool rader = false;
  cd *rd_bhat_f = nullptr;   // forward direction (pfa_forward.h): FFT_L of exp(-2 pi i u2 g^s / N2) / L
  cd* rd_bhat = nullptr;     // 3-D spectrum of the Rader kernel sequence in prime-factor positions (mixed_radix.h)
  int *rd_qidx = nullptr, *rd_ridx = nullptr;
  void* r89 = nullptr;       // N1 = 89: tables of the Rader column transform (pfa_rader89.h: Rader89Tab)
  int rows() const { return (n1 + 1) / 2; }   // spectrum rows k1 <= (N1-1)/2 kept by the permuted layout
  bool on() const { return n1 > 0; }
};

// Exact-length-n DFT plan (Bluestein / chirp-z): forward real -> half spectrum, inverse pairs.
struct Plan {
  int n = 0;        // DFT length
  int H = 0;        // n/2 + 1 bins of the half spectrum
  int lin = 0;      // longest real input of a forward transform
  int nout = 0;     // outputs kept by an inverse transform (n for PHAT, N of 2N for fractional delay)
  cd* w = nullptr;  // chirp exp(i pi j^2 / n), j < n
  Conv fwd;         // forward:  lin inputs -> H outputs
  Conv inv;         // inverse:  n inputs  -> n outputs (two real sequences per complex transform)
  Pfa pfa;          // prime-factor route of the PHAT inverse when n splits (otherwise `inv` does it)
  long long used = 0;   // Engine::plan_clock at the last get_plan (the cache evicts the least recently used plan)
  // elements per row of the spectra that forward_spectra writes and pair_correlations reads: the half spectrum, or
  // the permuted rows k1 <= (N1-1)/2 of the prime-factor layout
  size_t spec_stride() const { return pfa.on() ? size_t(pfa.rows()) * size_t(pfa.n2) : size_t(H); }
};

struct ProfileSlot {
  double ms = 0;
  int64_t launches = 0;
};

struct Engine {
  int device = 0;
  int cu_count = 256;              // compute units of the device (persistent kernels size their grids with it)
  hipStream_t stream = nullptr;    // FFT passes, copies, everything a caller can order against
  hipStream_t stream2 = nullptr;   // second launch-group slot (or the peak selection in PAL_OVERLAP=2)
  hipStream_t stream3 = nullptr;   // third launch-group slot (PAL_OVERLAP=3)
  hipEvent_t ev_join3 = nullptr;
  hipEvent_t ev_fin = nullptr;     // end of the latest finishing column pass (pfa_cols_fin.h): they run one at a time, see fin_serialize
  int fin_serialize(hipStream_t on);   // before such a launch: wait for the previous one; fin_done(on) behind it
  int fin_done(hipStream_t on);
  bool fin_pending = false;
  bool fin_serial = false;         // PAL_FIN_SERIAL=1
  // diagnostic switches of the finishing pass, read when the engine is created (not once per process: tests build engines
  // under different settings)
  int fin_dense = -1;              // PAL_FIN_DENSE: 1 / 0 = the pass on every / no dense column DFT; -1 (unset): where it measured faster
  bool fin_strips = false;         // PAL_FIN_STRIPS=1: also on short columns beside 16384-point row tiles
  bool fin_four = false;           // PAL_FIN_FOUR=1: on the four-step last pass
  bool fin_wide = false;           // PAL_FIN_WIDE=1: on column DFTs of five or six chunks
  bool fin_hist = false;           // PAL_FIN_HIST=1: histogram windows for every threshold multiplier
  bool rows_lean = true;           // PAL_ROWS_LEAN=0: stored rows of the other routes keep the three statistics launches
  long long rows_lean_min = 200000;   // PAL_ROWS_LEAN_MIN=<pairs>: smallest call that takes k_rows_lean (tests lower it)
  bool lean_store = true;          // PAL_LEAN_STORE=0: stored rows keep the round-2 statistics (pfa_cols_stats.h / three launches) + k_peak_finish
  int debug_memo = 0;              // PAL_DEBUG_MEMO=<n>: shrinks the distance rule's on-chip memo / stack (tests of its slow path)
  hipEvent_t ev_corr[2] = {}, ev_peaks[2] = {};   // hand-offs of the two correlation buffers between the streams
  int overlap = 3;                 // PAL_OVERLAP: 0 one stream; 1 launch groups alternate between two streams; 2 transforms on
                                   // `stream`, peak selection on `stream2`; 3 (default) groups rotate over three streams
  bool allow_r3 = true;            // PAL_RADIX3=0 forces power-of-two convolution lengths
  bool allow_pfa = true;           // PAL_PFA=0 keeps the PHAT inverse on the four-step chirp convolution
  bool allow_rader = true;         // PAL_RADER=0 keeps the row pass on the in-LDS chirp convolution
  int four_reg = -1;               // PAL_FOUR_REG: register-resident rows of the four-step route: -1 choose, 12 / 13 force 2^12 / 2^13, 0 = LDS tiles only
  bool xcd_rows = true;            // PAL_XCD_ROWS=0: row passes in plain workgroup order (pfa_kernels.h: row_work_item)
  bool allow_big = true;           // PAL_PFA_BIG=0: no register-resident row tiles (N2 <= 2048 only, as in round 1)
  int pfa_sub = 0;                 // transforms per row/column pass of the prime-factor route (PAL_PFA_SUB; 0 = whole group)
  std::string err;
  int chunk = 128;                              // transforms per launch group (forward spectra, simulation, synchronisation)
  bool chunk_auto = true;                       // pair pipeline: 240 transforms per group where a workspace slot stays <= 1 GiB
                                                // (480 rows: the finish launch's workgroups, two per CU, leave slots free
                                                //  beside the other streams' launches - 256 measured 3 % less than 240)
  int pair_group(int n) const {                 // (PAL_CHUNK / pal_set_chunk fix both)
    if (!chunk_auto) return chunk;
    const long long g = (1ll << 30) / (16ll * (n > 0 ? n : 1));
    return int(g < 32 ? 32 : (g > 240 ? 240 : g));
  }
  std::map<std::tuple<int, int, int>, Plan> plans;   // (n, lin, nout) -> plan
  struct XConv { Conv conv; long long used = 0; };
  std::map<size_t, XConv> xconvs;                    // sequence length -> convolution of pal_xcorr_vs_ref
  int max_plans = 64;                                // PAL_MAX_PLANS / pal_set_max_plans: bound of both caches (least recently used out first)
  long long plans_built = 0, plans_evicted = 0;      // pal_plan_stats
  long long plan_clock = 0;
  cd* stage_tw[16] = {};                        // stage-major twiddles per log2 N (<= 14: the big row tiles of pfa_big.h)
  cd* stage_twc[16] = {};                       // the same with a compact last stage (fft_core.h stage_twc_size)
  // growable device scratch
  void* ws[24] = {};              // (16..18: per-stream scratch of the finishing column pass, pfa_cols_fin.h; 19..21: its flagged pairs)
  size_t ws_bytes[24] = {};
  // profiling
  bool profiling = false;
  int prof_every = 1;              // pair pipeline: events around every prof_every-th launch group only
  long long prof_tick = 0;
  bool prof_gate = true;           // false while an unsampled launch group is being enqueued
  std::vector<hipEvent_t> ev_pool;
  size_t ev_used = 0;
  struct Pending { int slot; hipEvent_t a, b; };
  std::vector<Pending> pending;
  std::vector<std::string> slot_names;
  std::vector<ProfileSlot> slots;
  // rccl
  void* comm = nullptr;
  // device copy of the (trial, i<j) pair table of the last all-pairs shape
  void* quads = nullptr;
  int quad_B = 0, quad_M = 0;
  // pair blocking (large arrays): pairs processed in 16 x 16 blocks of microphones, records scattered back to row-major order
  int4* quads_blk = nullptr;
  int* perm_blk = nullptr;
  int blk_B = 0, blk_M = 0;
  int pair_block = 16;             // PAL_PAIR_BLOCK=<microphones per block side>, 0 = off; used from 96 microphones up

  int fail(int code, const char* fmt, ...);
  int check(hipError_t e, const char* what);
  int scratch(int idx, size_t bytes, void** out);
  const cd* stage_table(int ln);
  const cd* stage_table_compact(int ln);
  int build_pfa(Plan& pl);                  // pfa.hip: choose the split and make the tables (leaves pl.pfa off if none fits)
  void free_pfa(Pfa& f);
  int build_rader(Pfa& f, long long n, long long u2);   // pfa.hip: Rader tables when N2 is 991
  int pfa_pair_group(const Plan& pl, const cd* permuted, const int4* quads, int G, cd* Y, double* corr, size_t stride,
                     const int* zero_rows, hipStream_t on);
  int get_plan(int n, int lin, int nout, Plan** out);
  void free_plan(Plan& pl);
  int clear_plans();
  int xcorr_conv(size_t len, Conv** out);
  int alloc_conv(Conv& c, size_t needed);   // geometry, tables and chirp-spectrum storage for >= `needed` points
  void free_conv(Conv& c);
  int build_conv(Conv& c, const cd* w, int n, int neg_count, int pos_count, bool conj_kernel, double extra_scale);
  // profiling helpers
  int prof_slot(const char* name);
  void prof_begin(int slot, hipEvent_t* a, hipStream_t on);
  void prof_end(int slot, hipEvent_t a, hipStream_t on);
  void prof_flush();

  // pipelines (all pointers are device pointers)
  int forward_spectra(Plan& pl, const double* frames, size_t frame_stride, int rows, int len, cd* spectra, int* nonzero = nullptr);
  int pair_correlations(Plan& pl, const cd* spectra, int nspec, const int4* quads, int64_t npairs, int n2,
                        const pal_phat_params& prm, pal_pair_record* table, int32_t* ksel_multi, double* corr_out,
                        const int* nonzero = nullptr);
  int pairs_dev(const double* d_rows, int R, int L, const int32_t* d_pairs, int64_t P, const pal_phat_params& prm,
                pal_pair_record* d_table);
  int peaks(const double* corr, size_t stride, int rows, int n, int n2, const pal_phat_params& prm,
            pal_pair_record* table, int32_t* ksel_multi, hipStream_t on);
  // the same in pieces, for the column pass that produces the streaming statistics itself (pfa_cols_stats.h):
  // arguments + scratch (segments = `blocks` column blocks of a grid with rows of grid_n2, each with its own pivots),
  // [the caller's fused column launch], the finish launch
  int peaks_setup(const double* corr, size_t stride, int rows, int n, int n2, const pal_phat_params& prm, int blocks, int grid_n2,
                  hipStream_t on, PeakArgs& a);
  int peaks_finish(PeakArgs& a, int rows, pal_pair_record* table, int32_t* ksel_multi, hipStream_t on);
  int pfa_rows(const Plan& pl, const cd* permuted, const int4* quads, int G, cd* Y, hipStream_t on);
  bool pfa_can_fuse(const Plan& pl) const;
  int fin_setup(const Plan& pl, int rows, int nblk, int grid_rows, int grid_cols, const pal_phat_params& prm, int n2, pal_pair_record* table,
                int* need, int slot, hipStream_t on, PeakArgs& a, struct FinArgs& fa, unsigned& nwg, int G);
  bool fourstep_can_finish(const Plan& pl, const pal_phat_params& prm) const;   // pfa_cols_fin.h applies to the four-step last pass
  int fourstep_pair_group_fin(const Plan& pl, const cd* W, int G, int rows, const int* zero_rows, const pal_phat_params& prm, int n2,
                              pal_pair_record* table, int* need, int slot, hipStream_t on);
  bool pfa_can_finish(const Plan& pl, const pal_phat_params& prm) const;   // pfa_cols_fin.h applies (one peak per row, N1 of 2..4 chunks)
  bool rows_can_lean(const Plan& pl, const pal_phat_params& prm) const;        // k_rows_lean instead of pivots + stream + finish
  int rows_lean_group(const Plan& pl, const double* corr, size_t stride, int G, int rows, const pal_phat_params& prm, int n2,
                      pal_pair_record* table, int* need, int slot, hipStream_t on);
  bool pfa_can_lean_store(const Plan& pl, const pal_phat_params& prm) const;   // the per-wavefront statistics beside STORED rows (pfa_fin_lean.h)
  int pfa_pair_group_fin(const Plan& pl, const cd* permuted, const int4* quads, int G, int rows, cd* Y, const int* zero_rows,
                         const pal_phat_params& prm, int n2, pal_pair_record* table, int* need, int slot, hipStream_t on,
                         double* corr = nullptr, size_t stride = 0);
  int pfa_pair_group_fused(const Plan& pl, const cd* permuted, const int4* quads, int G, int rows, cd* Y, double* corr, size_t stride,
                           const int* zero_rows, const pal_phat_params& prm, int n2, pal_pair_record* table, int32_t* ksel_multi,
                           hipStream_t on);
  bool pfa_forward_applies(const Plan& pl, int len) const;
  int pfa_forward_spectra(Plan& pl, const double* frames, size_t frame_stride, int rows, int len, cd* spectra);
  bool pfa_forward = true;    // PAL_PFA_FWD=0: forward spectra on the four-step route even where the prime-factor cut applies
  bool fin_cols = true;       // PAL_FIN=0: the fused column pass always stores the correlation rows for a finish launch (pfa_cols_stats.h) instead
                              // of finishing the rows itself without storing them (pfa_cols_fin.h: one peak per row, nobody asks for `corr`)
  unsigned fin_epoch[3] = {};   // launches of the finishing column pass per stream slot (pfa_cols_fin.h: validity tag of what its blocks exchange)
  size_t fin_bytes[3] = {};
  bool allow_r89 = true;      // PAL_R89=0: dense 89-point column DFTs instead of Rader's 8 x 11 convolution (pfa_rader89.h)
  bool fuse_peaks = true;     // PAL_FUSED=0: separate column pass + pivot / stream launches instead of the fused column pass +
                              // peak statistics (pfa_cols_stats.h) where that applies
};

struct ProfScope {
  Engine* e;
  int slot;
  hipEvent_t a = nullptr;
  hipStream_t on;
  ProfScope(Engine* eng, const char* name, hipStream_t s = nullptr) : e(eng), slot(-1), on(s ? s : eng->stream) {
    if (e->profiling && e->prof_gate) {
      slot = e->prof_slot(name);
      e->prof_begin(slot, &a, on);
    }
  }
  ~ProfScope() {
    if (slot >= 0) e->prof_end(slot, a, on);
  }
};

inline int ceil_log2(size_t v) {
  int l = 0;
  while ((size_t(1) << l) < v) ++l;
  return l;
}

#define PAL_TRY(expr)                \
  do {                               \
    int _rc = (expr);                \
    if (_rc != PAL_OK) return _rc;   \
  } while (0)

#define PAL_HIP(expr) PAL_TRY(check((expr), #expr))

}  // namespace pal
