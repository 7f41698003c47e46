// pfa_cols_fin.h - column pass of the prime-factor route that FINISHES the rows itself (gfx950, fp64): the correlation rows are
// never written to HBM (`corr` leaves the chip only when the caller asks for it) and there is no finish launch.
//
// k_pfa_cols_stats (pfa_cols_stats.h) still stores the 0.7 MB per row because k_peak_finish reads a few thousand of them back:
// the window around the maximum that compute_snr leaves out (utils.py:238-250), the grid's edge columns, and the
// neighbourhoods of the candidates it resolves.  Here everything the selection of ONE peak per row (num_peaks = 1,
// main.py:204) can need is taken from the samples while they are in the accumulators:
//
// Two bodies share this kernel.  Without histograms (threshold 'adaptive', or 'median' with a multiplier in 0 .. 2: the default of
// main.py:204) every WAVEFRONT runs the statistics of pfa_fin_lean.h on its own and one wavefront per transform finishes the rows:
// that is the path of the metric run, read that header first.  With histograms (multipliers above 2) the block-level body below runs:
//
//   phase 1   as in k_pfa_cols_stats: histogram window of |x| around the block's median, maximum / first argmax, minimum,
//             sums, highest strict peak, reported ties - plus, when the lag window max_expected_delay is set, the highest
//             strict peak inside the window (it is kept by the distance rule unless a higher peak lies just outside the
//             window, utils.py:152,163), and the four edge columns of the grid (their neighbours sit in another output
//             index: tested by the finishing block).  A block whose samples at or above 0.8 x its maximum hold no strict
//             peak tests ALL its samples again (registers, no memory), so every block reports its true best peak and
//             nothing ever has to be rescanned.
//   phase 2   the SNR window [argmax - w, argmax + w) needs the row's argmax, which only exists once all column blocks of
//             the transform have seen their samples.  Every wavefront therefore publishes its maximum and the first index
//             of it right behind pass A (one more look at its samples) and bumps the transform's counter; the blocks of one
//             transform are adjacent in dispatch order (siblings are resident together), so by the time a block has done
//             its histogram windows, pass B and the lag-window search, its siblings' maxima are there: one lane checks the
//             counter (a bounded spin), then every wavefront sums ITS samples inside the window - still in registers.
//   finish    the block that arrives last at the second counter merges the blocks' results of both rows: SNR, the
//             threshold interval from the histogram windows, the fallback chain of utils.py:152-179 for one peak, and
//             writes the 48-byte record.  A row whose selection needs what was not kept - a threshold comparison inside
//             the median's interval, a tie that may outrank the best peak, a window peak closer than `distance` to the
//             window's edge (a higher peak just outside could suppress it), a window that holds most of the row's
//             energy - is flagged instead (need[pair] = 1): at the end of the call the engine runs the flagged pairs, in
//             pair order, through the stored-row path (pfa_cols_stats.h + k_peak_finish) - exact for every input, and
//             nothing extra is launched per group.
//
// Deadlock freedom of the wait: workgroups are dispatched in index order and the grid index interleaves the eight XCD
// streams, so the siblings of a block lie within 8 x blocks-per-transform consecutive indices; a waiting block's
// siblings are resident or next in line, and blocks that do not wait finish on their own.  The spin is bounded anyway
// (status bit 4 -> PAL_ERR_INTERNAL) so that a violated assumption cannot hang the device.
#pragma once
#include <climits>

#include "pfa_cols_stats.h"
#include "reduce.h"
#include "wave_reduce.h"

namespace pal {

struct FinPartial {               // what one column block hands to the finishing block, per row (whole 8-byte words)
  double vmin, hb, plat;          // minimum; highest strict peak (mb = -1: none); highest sample with an equal neighbour (-inf: none)
  double s1, s2, a1;              // sum x, sum x^2, sum |x|
  double hw, platw;               // lag window: highest strict peak inside it (mw = -1: none), highest tie inside it or its margins
  double hm;                      // highest strict peak in the margins of distance - 1 samples beside the window (mm = -1: none)
  double w1, w2;                  // sum x, sum x^2 of the block's samples inside the SNR window around the row's maximum
  int mb, mw, mm, pad;
};
static_assert(sizeof(FinPartial) == 104, "thirteen words");

struct FinArgs {
  pal_pair_record* table;         // [rows] records of this launch group
  int* need;                      // [rows] of this launch group inside the call-wide flag array: 1 = the pair goes through the stored-row path
  unsigned* done;                 // [G][S] the launch number (`epoch`) once a block's results are complete
  double* edge;                   // [rows][4][N1] columns 0, 1, N2 - 2, N2 - 1 of the grid
  double* emax;                   // [rows][S][4][2] every wavefront's (maximum, 2^32 epoch + 1 + first index of it), ONE 16-byte store behind pass A
  unsigned epoch;                 // number of this launch on its stream (> 0; the scratch starts zeroed): stale entries fail the comparison, nothing is reset
  FinPartial* parts;              // [rows][S]
  int* status;                    // engine status words (bit 2 of word 0: a wait timed out; word 4: flagged rows)
  int win_lo, win_hi;             // lag window as sample indices, |m - (n2 - 1)| / fs <= max_expected_delay (win_lo > win_hi: empty)
  int windowed;
  int pw;                         // FinPartial entries and `done` words per column block: 1, or one per wavefront (pfa_fin_lean.h)
  double* corr;                   // null, or [rows][stride]: the pass ALSO stores the correlation rows (the caller wants them, or the plan
  size_t stride;                  //   has no finishing form): nobody polls siblings then, the finisher reads the SNR window from the stored row
  int store_rows;                 // 1: this pass writes the rows (column forms); 0 with corr set: they are there already (k_rows_lean reads them)
  int cheb;                       // 1: no histograms - the median of |corr| is bounded by sqrt(2 mean(corr^2)) (see fin_row)
  unsigned long long* stamps;     // diagnostics (PAL_DEBUG_STAMPS=1): [workgroup][8] 100 MHz clock reads of lane 0 per phase
};

struct FinWave {                  // one wavefront's share of the finishing block's merge
  double vmax, hb, hw, hm, plat, platw, nvmin, s1, s2, a1, w1, w2;
  int imax, mb, mw, mm;
};

struct FinShared {                // finishing block's scratch, laid over the histograms of phase 1
  unsigned hwin[2 * kWin];
  int swin0[144];
  FinWave wave[4];
};
static_assert(sizeof(FinShared) <= 2 * (kLogBins + 1) * sizeof(unsigned), "the finish scratch aliases the histograms");
static_assert(alignof(FinShared) <= 16, "(the histograms' storage is 16-byte aligned)");

// What the column blocks of a transform hand to each other travels in device-scope relaxed atomic stores and loads (sc1:
// through the caches to the coherence point) - no release / acquire fences, which on this part write back and invalidate the
// whole L2 (buffer_wbl2 / buffer_inv: eight of them per workgroup made this kernel five times slower than its arithmetic).
// A device-scope store takes microseconds to be acknowledged, so nothing waits for one in the middle of a block's life: a
// wavefront's (maximum, index) pair is ONE aligned 16-byte store whose second word carries the launch number (a reader polls
// the entry until the number is this launch's); only at its very end a block waits for its stores (s_waitcnt vmcnt(0)),
// passes a barrier and writes the launch number into its `done` word, which the transform's last block polls.
__device__ __forceinline__ unsigned ld_agent(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ int ld_agent(const int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ double ld_agent(const double* p) {
  return __longlong_as_double(__hip_atomic_load(reinterpret_cast<const long long*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
__device__ __forceinline__ void st_agent(unsigned* p, unsigned v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_agent(double* p, double v) {
  __hip_atomic_store(reinterpret_cast<long long*>(p), __double_as_longlong(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <class T> __device__ __forceinline__ void st_words(T* dst, const T& v) {           // a struct of whole 8-byte words
  static_assert(sizeof(T) % 8 == 0, "whole words");
  const long long* src = reinterpret_cast<const long long*>(&v);
  long long* d = reinterpret_cast<long long*>(dst);
#pragma unroll
  for (int k = 0; k < int(sizeof(T) / 8); ++k) __hip_atomic_store(d + k, src[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <class T> __device__ __forceinline__ T ld_words(const T* src) {
  T v;
  long long* d = reinterpret_cast<long long*>(&v);
  const long long* s = reinterpret_cast<const long long*>(src);
#pragma unroll
  for (int k = 0; k < int(sizeof(T) / 8); ++k) d[k] = __hip_atomic_load(s + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return v;
}
__device__ __forceinline__ void stores_done() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
typedef double pal_d2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void st_agent16(double* p, double a, double b) {                 // one aligned 16-byte store, device scope
  const pal_d2 v = {a, b};
  // (s_nop: a store of more than 8 bytes reads its upper data registers a cycle late - the compiler knows that hazard for
  //  its own stores, not for inline assembly, and the registers are dead for it behind this statement)
  asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ void ld_agent16(const double* p, double& a, double& b) {
  pal_d2 v;
  asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  a = v.x; b = v.y;
}
constexpr double kEpochUnit = 4294967296.0;                    // 2^32: an entry's second word = epoch x 2^32 + (index + 1)

// Every wait is bounded, and giving up is not an error: a block that does not see its siblings' maxima in time publishes its results
// marked `abandoned` (the transform's rows are then flagged and resolved from stored rows at the end of the call), the
// finishing block that does not see its siblings' results flags the rows itself.  Siblings normally arrive within
// microseconds; the bound matters when finishing passes of several streams fill every workgroup slot of the device with
// blocks whose siblings then find no slot (possible in principle, never seen with 16 blocks per transform; seen with 34):
// the waiting blocks give up after a few milliseconds, leave, and the late siblings find the maxima they need.
constexpr int kSpinLimit = 1 << 11;

// ---- the finishing block: one row from the blocks' published results (all LANES lanes, uniform control flow) ----
template <int LANES>
__device__ __forceinline__ void fin_row(const PeakArgs& pa, const FinArgs& fa, int row, int N1, int N2, int nch, FinShared& fs, int tid) {
  constexpr int NW = LANES / 64;
  const int S = pa.splits, n = pa.n;
  const bool windowed = fa.windowed != 0;
  const bool want_median = pa.method == 0;
  // ---- every lane: some wavefronts' maxima, at most a few blocks' results and edge samples; merged locally, then across the workgroup
  double vmax = 0, vmin = INFINITY, hb = 0, plat = -INFINITY, s1 = 0, s2 = 0, a1 = 0, w1 = 0, w2 = 0;
  int imax = -1, mb = -1;
  double hw = 0, hm = 0, platw = -INFINITY;
  int mw = -1, mm = -1;
  bool abandoned = false;                                      // a block gave up waiting for the row's argmax: its window sums are missing
  {
    const double* em = fa.emax + size_t(row) * S * 8;
    for (int q = tid; q < S * 4; q += LANES) {
      if ((q & 3) >= nch) continue;
      const double v = ld_agent(em + 2 * q);
      const int i = int(ld_agent(em + 2 * q + 1) - double(fa.epoch) * kEpochUnit) - 1;      // (complete: every block of the transform is done)
      if (i >= 0 && i < n && (imax < 0 || arg_better<0>(v, i, vmax, imax))) { vmax = v; imax = i; }
    }
  }
  for (int q = tid; q < S * fa.pw; q += LANES) {
    const FinPartial pt = ld_words(fa.parts + size_t(row) * S * fa.pw + q);
    vmin = fmin(vmin, pt.vmin);
    if (pt.mb >= 0 && (mb < 0 || higher(pt.hb, pt.mb, hb, mb))) { hb = pt.hb; mb = pt.mb; }
    plat = fmax(plat, pt.plat);
    s1 += pt.s1; s2 += pt.s2; a1 += pt.a1;
    w1 += pt.w1; w2 += pt.w2;
    abandoned = abandoned || pt.pad != 0;
    if (windowed) {
      if (pt.mw >= 0 && (mw < 0 || higher(pt.hw, pt.mw, hw, mw))) { hw = pt.hw; mw = pt.mw; }
      if (pt.mm >= 0 && (mm < 0 || higher(pt.hm, pt.mm, hm, mm))) { hm = pt.hm; mm = pt.mm; }
      platw = fmax(platw, pt.platw);
    }
  }
  // the grid's first and last column: neighbours in another output index (m - 1 = (N2 - 1, t - 1), m + 1 = (0, t + 1))
  const double* E = fa.edge + size_t(row) * 4 * N1;
  for (int k = tid; k < 2 * N1; k += LANES) {
    const bool first = k < N1;
    const int t = first ? k : k - N1;
    const int m = first ? N2 * t : N2 * t + N2 - 1;
    if (m < 1 || m > n - 2) continue;                          // the row's end points are never peaks
    const double x = ld_agent(first ? E + t : E + 3 * N1 + t);
    const double xl = ld_agent(first ? E + 3 * N1 + t - 1 : E + 2 * N1 + t);
    const double xr = ld_agent(first ? E + N1 + t : E + t + 1);
    const bool tie = xl == x || xr == x;
    const bool pk = xl < x && xr < x;
    const bool inw = windowed && m >= fa.win_lo && m <= fa.win_hi;
    const bool inm = windowed && !inw && m >= fa.win_lo - (pa.dist - 1) && m <= fa.win_hi + (pa.dist - 1);
    if (tie) plat = fmax(plat, x);
    if (tie && (inw || inm)) platw = fmax(platw, x);
    if (pk && (mb < 0 || higher(x, m, hb, mb))) { hb = x; mb = m; }
    if (pk && inw && (mw < 0 || higher(x, m, hw, mw))) { hw = x; mw = m; }
    if (pk && inm && (mm < 0 || higher(x, m, hm, mm))) { hm = x; mm = m; }
  }
  {
    double nvmin = -vmin;
    for (int o = 32; o > 0; o >>= 1) {
      const double ov = shfl_down_d(vmax, o);
      const int oi = __shfl_down(imax, o, 64);
      if (oi >= 0 && (imax < 0 || arg_better<0>(ov, oi, vmax, imax))) { vmax = ov; imax = oi; }
      const double hv = shfl_down_d(hb, o);
      const int hi_ = __shfl_down(mb, o, 64);
      if (hi_ >= 0 && (mb < 0 || higher(hv, hi_, hb, mb))) { hb = hv; mb = hi_; }
      const double wv = shfl_down_d(hw, o);
      const int wi = __shfl_down(mw, o, 64);
      if (wi >= 0 && (mw < 0 || higher(wv, wi, hw, mw))) { hw = wv; mw = wi; }
      const double gv = shfl_down_d(hm, o);
      const int gi = __shfl_down(mm, o, 64);
      if (gi >= 0 && (mm < 0 || higher(gv, gi, hm, mm))) { hm = gv; mm = gi; }
      plat = fmax(plat, shfl_down_d(plat, o));
      platw = fmax(platw, shfl_down_d(platw, o));
      nvmin = fmax(nvmin, shfl_down_d(nvmin, o));
      s1 += shfl_down_d(s1, o); s2 += shfl_down_d(s2, o); a1 += shfl_down_d(a1, o);
      w1 += shfl_down_d(w1, o); w2 += shfl_down_d(w2, o);
    }
    __syncthreads();                                           // (the previous row's readers are done with the scratch)
    if ((tid & 63) == 0) {
      FinWave& w = fs.wave[tid >> 6];
      w.vmax = vmax; w.imax = imax; w.hb = hb; w.mb = mb; w.hw = hw; w.mw = mw; w.hm = hm; w.mm = mm; w.plat = plat; w.platw = platw; w.nvmin = nvmin;
      w.s1 = s1; w.s2 = s2; w.a1 = a1; w.w1 = w1; w.w2 = w2;
    }
    __syncthreads();
    const FinWave w0 = fs.wave[0];
    vmax = w0.vmax; imax = w0.imax; hb = w0.hb; mb = w0.mb; hw = w0.hw; mw = w0.mw; hm = w0.hm; mm = w0.mm; plat = w0.plat; platw = w0.platw; nvmin = w0.nvmin;
    s1 = w0.s1; s2 = w0.s2; a1 = w0.a1; w1 = w0.w1; w2 = w0.w2;
    for (int k = 1; k < NW; ++k) {
      const FinWave w = fs.wave[k];
      if (w.imax >= 0 && (imax < 0 || arg_better<0>(w.vmax, w.imax, vmax, imax))) { vmax = w.vmax; imax = w.imax; }
      if (w.mb >= 0 && (mb < 0 || higher(w.hb, w.mb, hb, mb))) { hb = w.hb; mb = w.mb; }
      if (w.mw >= 0 && (mw < 0 || higher(w.hw, w.mw, hw, mw))) { hw = w.hw; mw = w.mw; }
      if (w.mm >= 0 && (mm < 0 || higher(w.hm, w.mm, hm, mm))) { hm = w.hm; mm = w.mm; }
      plat = fmax(plat, w.plat); platw = fmax(platw, w.platw); nvmin = fmax(nvmin, w.nvmin);
      s1 += w.s1; s2 += w.s2; a1 += w.a1; w1 += w.w1; w2 += w.w2;
    }
    vmin = -nvmin;
  }
  bool flag = false;                                           // the row needs its samples: stored-row path at the end of the call
  int why = 0;                                                 // (diagnostics: which rule flagged it)
  if (imax < 0 || imax >= n) { imax = 0; flag = true; why |= 1; }
  if (__syncthreads_or(abandoned ? 1 : 0)) { flag = true; why |= 1; }
  // a tie that may outrank the best strict peak (plateaus are resolved from the stored row)
  if (plat > -INFINITY && (mb < 0 || plat >= hb)) { flag = true; why |= 2; }
  if (windowed && platw > -INFINITY && (mw < 0 || platw >= hw)) { flag = true; why |= 4; }

  // ---- SNR (utils.py:238-250): totals minus the window around the maximum
  const int wlo_s = imax - pa.snr_w > 0 ? imax - pa.snr_w : 0;
  const int whi_s = imax + pa.snr_w < n ? imax + pa.snr_w : n;
  const double nn = double(n - (whi_s - wlo_s));
  const double o1 = s1 - w1, o2 = s2 - w2;
  if (!(o2 >= 0.25 * s2)) { flag = true; why |= 8; }           // the window holds most of the energy: two-pass sum of the noise region
  double var = (o2 - o1 * o1 / nn) / nn;
  if (var < 0) var = 0;
  const double noise = sqrt(var);
  const double snr = noise == 0.0 ? INFINITY : vmax / noise;

  // ---- primary threshold (utils.py:144-149): exact ('adaptive'), or an interval from the blocks' histogram windows
  double tlo = 0, thi = 0;
  if (!want_median) {
    double va = (s2 - a1 * a1 / double(n)) / double(n);
    if (va < 0) va = 0;
    tlo = thi = pa.mult * (a1 / double(n) + sqrt(va));         // utils.py:147
  } else if (fa.cheb) {
    // No histograms: at most half of the samples can have x^2 >= 2 mean(x^2) (Markov), so median(|corr|) <= sqrt(2 s2 / n).
    // A peak at or above mult x that bound passes the threshold of utils.py:145 for certain - and the highest peak of a PHAT
    // row stands 3 to 4 sigma above a median of 0.67 sigma, the bound is 1.41 sigma.  A candidate below the bound is not
    // decided here: the row is flagged and the stored-row path computes its exact median.
    thi = pa.mult * sqrt(2.0 * s2 / double(n)) * (1.0 + 1e-12);
    tlo = -INFINITY;
  } else {
    const unsigned r1 = unsigned((n - 1) / 2), r2 = unsigned(n / 2);
    const BlockHist* bh = pa.bh + size_t(row) * S;
    bool have = S <= 144;
    if (tid < 2 * kWin) fs.hwin[tid] = 0;
    if (tid < S && tid < 144) fs.swin0[tid] = ld_agent(&bh[tid].win0);
    __syncthreads();
    int w0 = 0, wend = kLogBins;
    for (int q = 0; q < S && q < 144; ++q) {
      const int v = fs.swin0[q];
      w0 = v > w0 ? v : w0;
      wend = v + kWin < wend ? v + kWin : wend;
    }
    for (int q = tid; q < S && have; q += LANES) {
      atomicAdd(&fs.hwin[kWin], ld_agent(&bh[q].below));
      atomicAdd(&fs.hwin[kWin + 1], ld_agent(&bh[q].total));
    }
    for (int e = tid; e < S * kWin && have; e += LANES) {
      const int q = e / kWin, k = e - q * kWin;
      const int b = fs.swin0[q] + k;
      const unsigned v = ld_agent(&bh[q].h[k]);
      if (b < w0) atomicAdd(&fs.hwin[kWin], v);
      else if (b < wend) atomicAdd(&fs.hwin[b - w0], v);
    }
    __syncthreads();
    const unsigned total = fs.hwin[kWin + 1], base = fs.hwin[kWin];
    have = have && w0 < wend && total == unsigned(n) && r1 >= base;
    double b1lo = 0, b1hi = 0, b2lo = 0, b2hi = 0;
    if (have) {
      unsigned e = base;
      int f1 = -1, f2 = -1;
      for (int k = 0; k < wend - w0; ++k) {
        const unsigned v = fs.hwin[k];
        if (f1 < 0 && r1 < e + v) f1 = k;
        if (f2 < 0 && r2 < e + v) f2 = k;
        e += v;
      }
      have = f1 >= 0 && f2 >= 0;
      if (have) {
        b1lo = log_bin_floor(w0 + f1); b1hi = log_bin_floor(w0 + f1 + 1);
        b2lo = log_bin_floor(w0 + f2); b2hi = log_bin_floor(w0 + f2 + 1);
      }
    }
    __syncthreads();
    if (have) {                                                // np.median = the middle value, or the mean of the two middle ones
      const double ml_ = r2 != r1 ? (b1lo + b2lo) * 0.5 : b1lo, mh_ = r2 != r1 ? (b1hi + b2hi) * 0.5 : b1hi;
      tlo = pa.mult >= 0 ? pa.mult * ml_ : pa.mult * mh_;
      thi = pa.mult >= 0 ? pa.mult * mh_ : pa.mult * ml_;
    } else {
      flag = true;                                             // the windows missed the row's median: exact select over the stored row
      why |= 16;
    }
  }

  // ---- the fallback chain of utils.py:152-179 for ONE peak
  const double mean_abs = a1 / double(n);
  int branch = 0, sel = imax;
  double sel_h = vmax;
  bool argmax_fallback = false;
  if (!flag) {
    bool alt = false;
    if (mb >= 0 && hb >= thi) {
    } else if (mb >= 0 && hb >= tlo) {
      flag = true;                                             // inside the median's interval
      why |= 32;
    } else {
      branch |= PAL_BR_ALT_THRESHOLD;
      alt = true;
      if (!(mb >= 0 && hb >= mean_abs)) { branch |= PAL_BR_ARGMAX_NO_PEAKS; argmax_fallback = true; }
    }
    if (!flag && !argmax_fallback) {
      if (!windowed) {
        sel = mb; sel_h = hb;                                  // the highest peak of the row is kept by the distance rule
      } else {
        const double t_lo = alt ? mean_abs : tlo, t_hi = alt ? mean_abs : thi;
        bool found = false;
        if (mw >= 0 && hw >= t_hi) found = true;
        else if (mw >= 0 && hw >= t_lo) { flag = true; why |= 64; }
        if (!flag && !found) {                                 // no peak of the first search inside the window: mean(|corr|), then argmax
          branch |= PAL_BR_WINDOW_RETRY;
          if (mw >= 0 && hw >= mean_abs) found = true;
          else { branch |= PAL_BR_ARGMAX_WINDOW; argmax_fallback = true; }
        }
        if (found) {
          // the window's best peak is kept unless a HIGHER peak lies closer than `distance`; inside the window there is none,
          // and a sample outside it is that close only to peaks within distance - 2 of the window's edge.  The margins' best
          // peak stands for every peak there: only if IT is higher can the window's peak be suppressed (whether it is - the
          // margin peak may be suppressed itself - is a chain the stored row resolves)
          const bool near = mw - fa.win_lo < pa.dist - 1 || fa.win_hi - mw < pa.dist - 1;
          if (near && mm >= 0 && higher(hm, mm, hw, mw)) { flag = true; why |= 128; }
          else { sel = mw; sel_h = hw; }
        }
      }
    }
    if (argmax_fallback) { sel = imax; sel_h = vmax; }
  }
  if (tid == 0) {
    fa.need[row] = flag ? 1 : 0;
    if (flag) {
      atomicAdd(fa.status + 4, 1);
      for (int b = 0; b < 8; ++b)
        if (why >> b & 1) atomicAdd(fa.status + 5 + b, 1);
    } else {
      pal_pair_record r;
      r.k_sel = sel; r.branch = branch; r.k_argmax = imax; r.n_sel = 1;
      r.cmax = vmax; r.cmin = vmin; r.snr = snr; r.sel_height = sel_h;
      fa.table[row] = r;
    }
  }
}

}  // namespace pal
#include "pfa_fin_lean.h"
namespace pal {

// Where the samples of a column block come from
//   kColsDense     prime-factor grid, dense column DFT: the NW wavefronts are chunks of P1 = TC output indices of ONE 62-column strip
//   kColsRader89   prime-factor grid, N1 = 89: Rader's 8 x 11 convolution spread over four wavefronts (pfa_rader89.h)
//   kColsStrips    prime-factor grid, short dense column DFT (one chunk, N1 <= 23): the wavefronts are NW neighbouring strips
//   kColsFourStep  last pass of the four-step chirp convolution with register rows (conv_kernels.h k_colsreg_inv): a lane holds the
//                  P1 = M1 points of one column of the 2^P2-column workspace, sample m = r 2^P2 + c, the last row is partial
//                  (m < n); the wavefronts are NW neighbouring strips
#if defined(PAL_ABL_FIN) && PAL_ABL_FIN == 9
#define FIN_SYNC() ((void)0)                                    // timing experiment: no block barriers after the transform (invalid results)
#else
#define FIN_SYNC() __syncthreads()
#endif
enum { kColsDense = 0, kColsRader89 = 1, kColsStrips = 2, kColsFourStep = 3 };

struct FinSrc {
  const cd* Y;                    // prime-factor grid [G][N1][N2], or the four-step workspace [G][M1][2^LR]
  const double* T;                // dense column DFT: cos / sin table (pfa_kernels.h)
  const Rader89Tab* tab;          // kColsRader89
  const cd *twA, *twB, *w;        // kColsFourStep: root tables of the convolution, chirp exp(i pi j^2 / n)
};

// grid: 8 * ceil(G / 8) * nblk workgroups; index b -> XCD stream b & 7, slot b >> 3; the transforms g = x, x + 8, ... of stream x
// take nblk consecutive slots each (siblings adjacent, one XCD's L2 behind them when G is a multiple of 8)
// HIST: threshold method 'median' with a histogram window per block (a rigorous 0.5 % interval for the row's median); false:
//       'adaptive', or 'median' bounded without histograms (FinArgs.cheb: multipliers up to 2)
template <int MODE, int P1, int P2, bool HIST, bool FULL, int NW>
__global__ __launch_bounds__(64 * NW) __attribute__((amdgpu_waves_per_eu(MODE == kColsRader89 || (MODE == kColsFourStep && P1 <= 16) ? 3 : 2))) void k_pfa_cols_fin(FinSrc src, int N1, int N2, int G, int nch, int nblk,
                                                      const int* __restrict__ zero_rows, PeakArgs pa, FinArgs fa, int rows) {
  constexpr bool R89 = MODE == kColsRader89, FOUR = MODE == kColsFourStep;
  constexpr bool STRIPS = MODE == kColsStrips || MODE == kColsFourStep;
  constexpr int TC = FOUR ? 1 : P1, UNR = FOUR ? 1 : P2;
  static_assert(!R89 || (NW == 4 && FULL), "the Rader column transform is the four-wavefront N1 = 89 case");
  static_assert(!FOUR || !HIST, "the four-step last pass finishes its rows without histograms only");
  static_assert(!HIST || NW <= 4, "the histogram form merges at most four wavefronts per block");
  const cd* __restrict__ Y = src.Y;
  const double* __restrict__ T = src.T;
  const Rader89Tab* __restrict__ tab = src.tab;
  // histograms of |x| (+ one dump bin for the lanes that own nothing); with R89 the same memory is first the exchange plane of
  // the column transform (88 x 64 doubles), and in the finishing block its scratch at the end
  constexpr int kHistDoubles = (2 * (kLogBins + 1) * int(sizeof(unsigned)) + 7) / 8;
  __shared__ __attribute__((aligned(16))) double lds_big[R89 ? 88 * 64 : kHistDoubles];
  unsigned (*const hist)[kLogBins + 1] = reinterpret_cast<unsigned (*)[kLogBins + 1]>(lds_big);
  __shared__ FinPartial res[4][2];
  __shared__ double wmax[4][2];
  __shared__ unsigned wtot[4][2];
  __shared__ int medbin[2];
  __shared__ int s_imax[2];
  __shared__ int s_flag;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ch = STRIPS ? 0 : wave;
  const int xs = int(blockIdx.x & 7u), slot = int(blockIdx.x >> 3);
  const int cb = slot % nblk, g = xs + 8 * (slot / nblk);
  if (g >= G) return;                                          // (uniform: the grid is padded to whole XCD rounds)
  int stamp_at = 0;
  auto stamp = [&]() {
    if (fa.stamps && tid == 0) fa.stamps[size_t(blockIdx.x) * 8 + stamp_at] = __builtin_amdgcn_s_memrealtime();
    ++stamp_at;
  };
  stamp();
  // the grid's columns are dealt evenly to the strips (at most kColsOwn = 62 each: two border lanes per wavefront)
  const int strips = STRIPS ? nblk * NW : nblk, strip = STRIPS ? cb * NW + wave : cb;
  const int nact = STRIPS ? NW : nch;                          // wavefronts of a block that hold samples
  const int c_lo = int((long long)strip * N2 / strips), c_hi = int((long long)(strip + 1) * N2 / strips);   // owned columns [c_lo, c_hi)
  const bool active = STRIPS ? true : ch < nch;
  const int m2 = c_lo - 1 + lane;
  const bool live = m2 >= 0 && m2 < N2 && lane <= c_hi - c_lo + 1;
  const bool own = live && lane >= 1 && lane <= c_hi - c_lo;
  const bool inner = own && m2 >= 1 && m2 <= N2 - 2;
  const int m2c = m2 < 0 ? 0 : (m2 < N2 ? m2 : N2 - 1);
  const int h = (N1 - 1) / 2;
  const int n = pa.n;
  constexpr bool want_median = HIST;
  constexpr int LANES = 64 * NW;
  if (want_median && !R89) {
    unsigned* hz = &hist[0][0];
    for (int q = tid; q < 2 * (kLogBins + 1); q += LANES) hz[q] = 0;
  }
  constexpr int TCD = (R89 || FOUR) ? 1 : TC;                  // (the dense form's accumulators exist in the dense forms only)
  double cx[TCD], sy[TCD], cy[TCD], sx[TCD];
  double sumx = 0, sumy = 0;
  cd y0 = mk(0, 0);
  cd ro[R89 ? kR89Slots : (FOUR ? P1 : 1)];                    // Rader form: c[t] of the output indices tab->tmap[wave][.]; four-step: the column's P1 points
  cd c0 = mk(0, 0);                                            // Rader form: c[0] (wavefront 0)
  const double kp = zero_rows && zero_rows[2 * g] ? 0.0 : 1.0, kq = zero_rows && 2 * g + 1 < rows && zero_rows[2 * g + 1] ? 0.0 : 1.0;
  // four-step: does sample (t, this lane's column) exist?  (the last row of the workspace's grid is partial: m = t N2 + m2 < n)
  auto ok_at = [&](int t) { return !FOUR || t * N2 + m2 < n; };
  // Rader form: the output index of slot i sits in lane i of one register (loaded once; a scalar load per sample and pass made
  // every statistics loop a chain of scalar-memory latencies - 140 of them per wavefront)
  int slot_t = 0;
  if constexpr (R89) {
    slot_t = tab->tmap[wave][lane < kR89Slots ? lane : 0];
    r89_columns(Y + size_t(g) * N1 * N2 + m2c, N2, wave, lane, tab, lds_big, ro, c0);
    if (want_median) {                                         // (the exchange plane is free now: it becomes the histograms)
      unsigned* hz = &hist[0][0];
      for (int q = tid; q < 2 * (kLogBins + 1); q += LANES) hz[q] = 0;
    }
    if (kp == 0.0 || kq == 0.0) {                              // (uniform; a silent microphone's rows are exact zeros)
      c0.x *= kp; c0.y *= kq;
#pragma unroll
      for (int i = 0; i < kR89Slots; ++i) { ro[i].x *= kp; ro[i].y *= kq; }
    }
  } else if constexpr (FOUR) {
    // conv_kernels.h k_colsreg_inv: the column's M1 points, four-step twiddle, inverse M1-point DFT, then the chirp (bluestein.hip CorrStorer)
    const cd* in = Y + (size_t(g) * P1 << P2) + m2c;
#pragma unroll
    for (int k1 = 0; k1 < P1; ++k1) ro[k1] = in[size_t(k1) << P2];
    colsreg_twiddle<P1, P2, true>(ro, unsigned(m2c), src.twA, src.twB);
    reg_dft<P1, true>(ro, src.twA);
#pragma unroll
    for (int r = 0; r < P1; ++r) {
      const int j = (r << P2) + m2c;
      const cd z = j < n ? cmul(ro[r], src.w[j]) : mk(0, 0);
      ro[r] = mk(z.x * kp, z.y * kq);
    }
  } else {
    if (active) pfa_cols_accumulate<TC, UNR>(Y + size_t(g) * N1 * N2 + m2c, N1, N2, nch, ch, T, y0, cx, sy, cy, sx, sumx, sumy);
    if (kp == 0.0) { y0.x = sumx = 0.0; }
    if (kq == 0.0) { y0.y = sumy = 0.0; }
#pragma unroll
    for (int tt = 0; tt < TCD; ++tt) {
      cx[tt] *= kp; sy[tt] *= kp; cy[tt] *= kq; sx[tt] *= kq;
    }
  }
  stamp();                                                     // 1: accumulated
  if constexpr (!HIST) {
    // every wavefront on its own from here (pfa_fin_lean.h); ONE wavefront of the transform's last block then finishes both rows
    bool last;
    if constexpr (R89) {
      last = fin_lean_r89<true, false, kR89Slots>(ro, c0, slot_t, 0u, wave, lane, g, cb, nblk, N1, N2, rows, c_lo, m2, own, inner, cb == 0 || cb == nblk - 1,
                                                  -1, true, pa, fa, stamp);
    } else if constexpr (FOUR) {
      // four-step workspace: slot 1 + r = grid row r (t = r), no slot 0; the row ends inside grid row (n - 1) / N2
      unsigned emask = 0;
#pragma unroll
      for (int r = 0; r < P1; ++r)
        if (r * N2 < n) emask |= 1u << (1 + r);
      const int pr = (n - 1) / N2;
      last = fin_lean_r89<false, true, P1>(ro, mk(0, 0), lane, emask, wave, lane, g, cb, nblk, N1, N2, rows, c_lo, m2, own, inner, c_lo == 0 || c_hi == N2,
                                           1 + pr, pr * N2 + m2 < n, pa, fa, stamp);
    } else {
      // dense column DFT: the accumulators become samples once (slot 1 + tt: t = ch TC + tt + 1, slot 1 + TC + tt: N1 - t; slot 0: t = 0)
      static_assert(R89 || 2 * TCD == kR89Slots, "22 slots per wavefront");
      cd zs[kR89Slots];
      const cd z0 = mk(y0.x + sumx, y0.y + sumy);
      unsigned emask = active && ch == 0 ? 1u : 0u;
#pragma unroll
      for (int tt = 0; tt < TCD; ++tt) {
        zs[tt] = mk(y0.x + cx[tt] - sy[tt], y0.y + cy[tt] + sx[tt]);
        zs[TCD + tt] = mk(y0.x + cx[tt] + sy[tt], y0.y + cy[tt] - sx[tt]);
        if (active && (FULL || ch * TC + tt + 1 <= h)) emask |= (1u << (1 + tt)) | (1u << (1 + TCD + tt));
      }
      const int tl = ch * TC + (lane < TCD ? lane : lane - TCD) + 1;
      const int st = lane < TCD ? tl : N1 - tl;
      last = fin_lean_r89<false, false, kR89Slots>(zs, z0, st, emask, wave, lane, g, cb, nblk, N1, N2, rows, c_lo, m2, own, inner,
                                                   active && (c_lo == 0 || c_hi == N2), -1, true, pa, fa, stamp);
    }
    if (!last) return;
    stamp();                                                   // 5
    if (wave != 0) return;                                     // ONE wavefront finishes the transform's rows (fin_row_wave)
    bool late = false;
    for (int q = lane; q < nblk * fa.pw; q += 64) {
      int spins = 0;
      while (ld_agent(fa.done + size_t(g) * nblk * fa.pw + q) != fa.epoch) {
        if (++spins > kSpinLimit) { late = true; break; }
        __builtin_amdgcn_s_sleep(8);
      }
    }
    if (__ballot(late)) {                                      // the siblings' results are not there: both rows go through the stored-row path
      if (lane < 2 && 2 * g + lane < rows) { fa.need[2 * g + lane] = 1; atomicAdd(fa.status + 4, 1); atomicAdd(fa.status + 13, 1); }
      return;
    }
#pragma nounroll
    for (int r = 0; r < 2; ++r)
      if (2 * g + r < rows) fin_row_wave(pa, fa, 2 * g + r, N1, N2, lane);
    stamp();                                                   // 6: both rows finished (last block only)
    return;
  }
#if defined(PAL_ABL_FIN) && PAL_ABL_FIN == 1
  { double acc = c0.x + c0.y; for (int i = 0; i < int(sizeof(ro) / sizeof(ro[0])); ++i) acc += ro[i].x * ro[i].y; if (acc == 1.2345e300) fa.status[3] = 1; return; }
#endif
  // the samples of this lane, fn(value, t, exists): `value()` yields the sample (formed by two additions in the dense form, so
  // only where it is wanted); t and `exists` are wave-uniform.  Dense form: lag order (t = 0 for chunk 0 only, the chunk
  // ascending, the mirrors descending); Rader form: table order
  auto each_sample = [&](int r, auto&& fn) {
    if constexpr (FOUR) {
#pragma unroll
      for (int i = 0; i < P1; ++i) fn([&]() { return r ? ro[i].y : ro[i].x; }, i, i * N2 < n);
    } else if constexpr (R89) {
      fn([&]() { return r ? c0.y : c0.x; }, 0, wave == 0);
#pragma unroll
      for (int i = 0; i < kR89Slots; ++i) fn([&]() { return r ? ro[i].y : ro[i].x; }, __builtin_amdgcn_readlane(slot_t, i), true);
    } else {
      const double base = r ? y0.y : y0.x;
      fn([&]() { return base + (r ? sumy : sumx); }, 0, ch == 0);
#pragma unroll
      for (int tt = 0; tt < TCD; ++tt) {
        const int t = ch * TC + tt + 1;
        fn([&]() { return r ? base + cy[tt] + sx[tt] : base + cx[tt] - sy[tt]; }, t, FULL || t <= h);
      }
#pragma unroll
      for (int tt = TCD - 1; tt >= 0; --tt) {
        const int t = ch * TC + tt + 1;
        fn([&]() { return r ? base + cy[tt] - sx[tt] : base + cx[tt] + sy[tt]; }, N1 - t, FULL || t <= h);
      }
    }
  };
  FIN_SYNC();

  // ---- pass A: histogram of |x| (exact counts), maximum with its first index, minimum, sums
  const int nrow = 2 * g + 1 < rows ? 2 : 1;
  constexpr unsigned kBase4 = unsigned(1023 * 128 - kLogBins) * 4u, kTop4 = kBase4 + unsigned(kLogBins - 1) * 4u;
  const unsigned lift = own ? 0u : unsigned(kLogBins) * 4u;
  const unsigned one = 1u;
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    double vm = -INFINITY, vn = INFINITY, s1 = 0, s2 = 0, a1 = 0;
    int tm = 0;                                                // output index of the lane's maximum (samples come in lag order: the first one stays)
    if (active && r < nrow) {
      char* const hrow = reinterpret_cast<char*>(&hist[r][0]);
      each_sample(r, [&](auto&& value, int t, bool exists) {
        if (!exists) return;
        const double xv = value();
        const bool ok = ok_at(t);                              // (four-step: the last grid row is partial)
        const double x = ok ? xv : 0.0;
        const bool up = ok & ((x > vm) | (R89 & (x == vm) & (t < tm)));  // (Rader form: not in lag order - the smaller index of equal samples; no short cuts: one basic block)
        vm = up ? x : vm;
        tm = up ? t : tm;
        vn = ok ? min_raw(vn, x) : vn;
        s1 += x;
        s2 = __builtin_fma(x, x, s2);
        a1 += fabs(x);
        if (want_median) {
          const unsigned key4 = (unsigned(__double2hiint(x)) & 0x7fffe000u) >> 11;
          const unsigned off = max(min(max(key4, kBase4), kTop4) - kBase4, lift);
          atomicAdd(reinterpret_cast<unsigned*>(hrow + off), one);
        }
      });
    }
    int im = own && vm > -INFINITY ? m2 + N2 * tm : -1;
    if (!own) { vm = -INFINITY; vn = INFINITY; s1 = s2 = a1 = 0; }
#if defined(PAL_ABL_FIN) && PAL_ABL_FIN == 21
    if (vm + vn + s1 + s2 + a1 == 1.2345e300 && im == 77) fa.status[3] = 1;
    if (r == 1) return;
    continue;
#endif
    wave_arg63(vm, im, [](double v1, int i1, double v2, int i2) { return arg_better<0>(v1, i1, v2, i2); });
    vn = wave_min63(vn);
    s1 = wave_sum63(s1);
    s2 = wave_sum63(s2);
    a1 = wave_sum63(a1);
#if defined(PAL_ABL_FIN) && PAL_ABL_FIN == 22
    if (vm + vn + s1 + s2 + a1 == 1.2345e300 && im == 77) fa.status[3] = 1;
    if (r == 1) return;
    continue;
#endif
    if (lane == 63) {
      wmax[wave][r] = im >= 0 ? vm : -INFINITY;
      FinPartial& w = res[wave][r];
      w.vmin = vn; w.s1 = s1; w.s2 = s2; w.a1 = a1;
      // the wavefront's maximum and the first index of it go out NOW, in one 16-byte store that nobody waits for: the siblings
      // need the row's argmax for their SNR window sums, and it has arrived by the time they are through their own pass B
      if (active && r < nrow)
        st_agent16(fa.emax + ((size_t(2 * g + r) * pa.splits + cb) * 4 + wave) * 2, vm, double(fa.epoch) * kEpochUnit + double(im + 1));
    }
  }
  FIN_SYNC();
  // ---- the block's median bin per row, then the window around it is published
  if (want_median) {
    constexpr int PER = kLogBins / LANES;
    unsigned hv[2][PER], sum[2] = {0, 0};
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
      for (int q = 0; q < PER; ++q) { hv[r][q] = hist[r][PER * tid + q]; sum[r] += hv[r][q]; }
    unsigned inc[2] = {sum[0], sum[1]};
    for (int o = 1; o < 64; o <<= 1) {
      const unsigned t0 = __shfl_up(inc[0], o, 64), t1 = __shfl_up(inc[1], o, 64);
      if (lane >= o) { inc[0] += t0; inc[1] += t1; }
    }
    if (lane == 63) { wtot[wave][0] = inc[0]; wtot[wave][1] = inc[1]; }
    if (tid < 2) medbin[tid] = 0;
    __syncthreads();
    unsigned ex[2], cnt[2];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      unsigned before = 0;
      cnt[r] = 0;
      for (int w = 0; w < NW; ++w) { before += w < wave ? wtot[w][r] : 0; cnt[r] += wtot[w][r]; }
      ex[r] = before + inc[r] - sum[r];
      const unsigned mid = cnt[r] >> 1;
      unsigned e = ex[r];
#pragma unroll
      for (int q = 0; q < PER; ++q) {
        if (hv[r][q] && mid >= e && mid < e + hv[r][q]) medbin[r] = PER * tid + q;
        e += hv[r][q];
      }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      if (2 * g + r >= rows) continue;
      int w0 = medbin[r] - kWin / 2;
      w0 = w0 < 0 ? 0 : (w0 > kLogBins - kWin ? kLogBins - kWin : w0);
      BlockHist* bh = pa.bh + size_t(2 * g + r) * pa.splits + cb;
      if (w0 >= PER * tid && w0 < PER * tid + PER) {
        unsigned below = ex[r];
#pragma unroll
        for (int q = 0; q < PER; ++q) below += PER * tid + q < w0 ? hv[r][q] : 0u;
        st_agent(reinterpret_cast<unsigned*>(&bh->win0), unsigned(w0)); st_agent(&bh->below, below); st_agent(&bh->total, cnt[r]);
      }
      if (tid < kWin) st_agent(&bh->h[tid], hist[r][w0 + tid]);
    }
  }

  // ---- the grid's edge columns go to the finishing block as they are (blocks 0 and nblk - 1 only)
  if ((c_lo == 0 || c_hi == N2) && active) {                   // (uniform)
    const bool mine = own && (m2 <= 1 || m2 >= N2 - 2);
    const int e = m2 <= 1 ? m2 : 3 - (N2 - 1 - m2);
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      if (2 * g + r >= rows) continue;
      double* dst = fa.edge + (size_t(2 * g + r) * 4 + (mine ? e : 0)) * N1;
      each_sample(r, [&](auto&& value, int t, bool exists) {
        if (!exists) return;
        const double x = value();
        if (mine && ok_at(t)) st_agent(dst + t, x);
      });
    }
  }
  stamp();                                                     // 2: pass A, histogram windows, edge columns
#if defined(PAL_ABL_FIN) && PAL_ABL_FIN == 2
  return;
#endif

  // ---- pass B: the highest strict peak behind the block's exact bound; a block whose bounded search found no peak
  //      searches all its samples (second round), so its result is exact
  const bool windowed = fa.windowed != 0;
  double pfl[2];
  bool redo[2] = {false, false};
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    double vfloor = wmax[0][r];
#pragma unroll
    for (int w = 1; w < NW; ++w) vfloor = fmax(vfloor, wmax[w][r]);
    pfl[r] = vfloor > 0 ? 0.8 * vfloor : -INFINITY;
  }
#pragma nounroll
  for (int round = 0; round < 2; ++round) {
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const int row = 2 * g + r;
      if (row >= rows || (round == 1 && !redo[r])) continue;   // (uniform)
      const double pfloor = round == 0 ? pfl[r] : -INFINITY;
      const double myfloor = inner ? pfloor : INFINITY;        // (the grid's edge columns are the finishing block's)
      double hb = -INFINITY, plat = -INFINITY;
      int mb = -1;
      if (active) {
        each_sample(r, [&](auto&& value, int t, bool exists) {
          if (!exists) return;
          const double x = value();
          if (__ballot(x >= myfloor)) {
            const int m = m2 + N2 * t;
            const double left = from_lower_lane(x), right = from_upper_lane(x);
            const bool here = inner && (!FOUR || m <= n - 2);    // (four-step: the row ends inside the grid's last row)
            const bool cand = here && x >= pfloor && (R89 ? (x > hb || (x == hb && m > mb)) : x >= hb);
            const bool pk = cand && left < x && right < x;
            hb = pk ? x : hb;
            mb = pk ? m : mb;
            plat = here && x >= pfloor && left == x ? fmax(plat, x) : plat;
          }
        });
      }
      wave_arg63(hb, mb, [](double v1, int i1, double v2, int i2) { return higher(v1, i1, v2, i2); });
      plat = wave_max63(plat);
      if (lane == 63) {
        FinPartial& w = res[wave][r];
        w.hb = hb; w.plat = plat; w.mb = mb;
      }
      // the lag window and its margins of distance - 1 samples: their highest strict peaks, whatever their height (a handful of
      // output indices meet them)
      if (windowed && round == 0) {
        const int elo = fa.win_lo - (pa.dist - 1), ehi = fa.win_hi + (pa.dist - 1);
        const int lo1 = elo > 1 ? elo : 1, hi1 = ehi < n - 2 ? ehi : n - 2;     // (the row's end points are never peaks)
        const int t_lo = lo1 / N2, t_hi = hi1 / N2;
        double hq = -INFINITY, hg = -INFINITY, platw = -INFINITY;
        int mq = -1, mg = -1;
        if (active && lo1 <= hi1) {
          each_sample(r, [&](auto&& value, int t, bool exists) {
            if (!exists || t < t_lo || t > t_hi) return;       // (uniform)
            const double x = value();
            const int m = m2 + N2 * t;
            const double left = from_lower_lane(x), right = from_upper_lane(x);
            const bool in = inner && m >= lo1 && m <= hi1;
            platw = in && (left == x || right == x) ? fmax(platw, x) : platw;
            const bool pk = in && left < x && right < x;
            const bool inw = m >= fa.win_lo && m <= fa.win_hi;
            if (pk && inw && (R89 ? (x > hq || (x == hq && m > mq)) : x >= hq)) { hq = x; mq = m; }
            if (pk && !inw && (R89 ? (x > hg || (x == hg && m > mg)) : x >= hg)) { hg = x; mg = m; }
          });
        }
        wave_arg63(hq, mq, [](double v1, int i1, double v2, int i2) { return higher(v1, i1, v2, i2); });
        wave_arg63(hg, mg, [](double v1, int i1, double v2, int i2) { return higher(v1, i1, v2, i2); });
        platw = wave_max63(platw);
        if (lane == 63) {
          FinPartial& w = res[wave][r];
          w.hw = hq; w.mw = mq; w.hm = hg; w.mm = mg; w.platw = platw;
        }
      }
    }
    FIN_SYNC();
    if (round == 1) break;
    bool again = false;
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      bool any = false;
      for (int w = 0; w < NW; ++w) any = any || res[w][r].mb >= 0;
      redo[r] = 2 * g + r < rows && !any && pfl[r] > -INFINITY;
      again = again || redo[r];
    }
    if (!again) break;                                         // (uniform: every lane read the same LDS words)
    FIN_SYNC();
  }
  // ---- phase 2: the siblings' maxima (published long ago), then the SNR window sums of this block's samples
  stamp();                                                     // 3: pass B
#if defined(PAL_ABL_FIN) && PAL_ABL_FIN == 3
  return;
#endif
  if (tid == 0) s_flag = 0;                                    // (1: a wavefront gave up waiting for the siblings' maxima)
  FIN_SYNC();
  if (wave < 2 && 2 * g + wave < rows) {
    const int row = 2 * g + wave;
    double bv = 0;
    int bi = -1;
    const double* em = fa.emax + size_t(row) * pa.splits * 8;
    const double want = double(fa.epoch);
    bool late = false;
    for (int q = lane; q < pa.splits * 4; q += 64) {           // (entry = block * 4 + wavefront; idle wavefronts never write theirs)
      if ((q & 3) >= nact) continue;
      double v = 0, code = 0;
      int spins = 0;
      for (;;) {                                               // (the entry of this launch: its second word carries the launch number)
        ld_agent16(em + 2 * q, v, code);
        if (floor(code / kEpochUnit) == want) break;
        if (++spins > kSpinLimit) { late = true; break; }
        __builtin_amdgcn_s_sleep(8);
      }
      const int i = int(code - want * kEpochUnit) - 1;
      if (!late && i >= 0 && i < n && (bi < 0 || arg_better<0>(v, i, bv, bi))) { bv = v; bi = i; }
    }
    const bool gave_up = __ballot(late) != 0;
    if (gave_up && lane == 0) { s_flag = 1; atomicAdd(fa.status + 13, 1); }
    wave_arg63(bv, bi, [](double v1, int i1, double v2, int i2) { return arg_better<0>(v1, i1, v2, i2); });
    if (lane == 63) s_imax[wave] = bi;
  }
  FIN_SYNC();
  stamp();                                                     // 4: the row's argmax is known
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    if (2 * g + r >= rows) continue;
    int imax = s_imax[r];
    if (imax < 0 || imax >= n) imax = 0;
    const int A = imax - pa.snr_w > 0 ? imax - pa.snr_w : 0, B = imax + pa.snr_w < n ? imax + pa.snr_w : n;   // [A, B)
    const int tA = A / N2, tB = (B - 1) / N2;
    const bool inA = m2 >= A - tA * N2, inB = m2 < B - tB * N2;
    double w1 = 0, w2 = 0;
    if (active) {
      each_sample(r, [&](auto&& value, int t, bool exists) {
        if (!exists || t < tA || t > tB) return;               // (uniform)
        const double x = value();
        const bool in = own && (t > tA || inA) && (t < tB || inB) && ok_at(t);
        const double xm = in ? x : 0.0;
        w1 += xm;
        w2 = __builtin_fma(xm, xm, w2);
      });
    }
    w1 = wave_sum63(w1);
    w2 = wave_sum63(w2);
    if (lane == 63) { res[wave][r].w1 = w1; res[wave][r].w2 = w2; }
  }
  FIN_SYNC();
  // ---- publish: lanes 0 / 1 merge the wavefronts of row p / q; then the block is done
  if (tid < 2 && 2 * g + tid < rows) {
    const int r = tid, row = 2 * g + r;
    FinPartial pt;
    pt.hb = pt.plat = pt.platw = -INFINITY;
    pt.vmin = INFINITY;
    pt.mb = pt.mw = pt.mm = -1;
    pt.pad = s_flag;                                           // 1: the window sums are not valid (the block gave up waiting)
    pt.s1 = pt.s2 = pt.a1 = pt.hw = pt.hm = pt.w1 = pt.w2 = 0;
    for (int w = 0; w < NW; ++w) {
      const FinPartial x = res[w][r];
      pt.vmin = fmin(pt.vmin, x.vmin);
      if (x.mb >= 0 && (pt.mb < 0 || higher(x.hb, x.mb, pt.hb, pt.mb))) { pt.hb = x.hb; pt.mb = x.mb; }
      pt.s1 += x.s1; pt.s2 += x.s2; pt.a1 += x.a1; pt.w1 += x.w1; pt.w2 += x.w2;
      pt.plat = fmax(pt.plat, x.plat);
      if (windowed) {
        if (x.mw >= 0 && (pt.mw < 0 || higher(x.hw, x.mw, pt.hw, pt.mw))) { pt.hw = x.hw; pt.mw = x.mw; }
        if (x.mm >= 0 && (pt.mm < 0 || higher(x.hm, x.mm, pt.hm, pt.mm))) { pt.hm = x.hm; pt.mm = x.mm; }
        pt.platw = fmax(pt.platw, x.platw);
      }
    }
    st_words(fa.parts + size_t(row) * pa.splits + cb, pt);
  }
  stores_done();                                               // this wavefront's stores (results, histogram windows, edge columns) have landed ...
  FIN_SYNC();                                             // ... and so have the other wavefronts' before lane 0 announces the block
  if (tid == 0) st_agent(fa.done + size_t(g) * nblk + cb, fa.epoch);
  stamp();                                                     // 5: published
  if (cb != nblk - 1) return;                                  // (uniform)

  // ---- the last block of the transform waits for its siblings' results and finishes both rows
  {
    bool late = false;
    for (int q = tid; q < nblk - 1; q += LANES) {
      int spins = 0;
      while (ld_agent(fa.done + size_t(g) * nblk + q) != fa.epoch) {
        if (++spins > kSpinLimit) { late = true; break; }
        __builtin_amdgcn_s_sleep(8);
      }
    }
    if (__syncthreads_or(late ? 1 : 0)) {                     // the siblings' results are not there: both rows go through the stored-row path
      if (tid < 2 && 2 * g + tid < rows) { fa.need[2 * g + tid] = 1; atomicAdd(fa.status + 4, 1); atomicAdd(fa.status + 13, 1); }
      return;
    }
  }
  FinShared& fsh = *reinterpret_cast<FinShared*>(lds_big);
#pragma nounroll
  for (int r = 0; r < 2; ++r)
    if (2 * g + r < rows) fin_row<LANES>(pa, fa, 2 * g + r, N1, N2, nact, fsh, tid);
  stamp();                                                     // 6: both rows finished (last block only)
}


// ---- per-wavefront statistics over rows that are already in HBM (any route: the four-step last pass, column DFTs of five or six
//      chunks, small grids): the counterpart of k_peak_pivots + k_peak_stream + k_peak_finish for one peak per row without histograms.
// A wavefront reads NS overlapping chunks of 64 consecutive samples of both rows of a transform - chunk t = samples 62 t - 1 ...
// 62 t + 62: lanes 1 .. 62 own a sample, lanes 0 and 63 hold the neighbours - which is the (slot, lane) layout of pfa_fin_lean.h with
// N2 = 62 columns, m = 62 t + (lane - 1): the row's partial last chunk is the PART slot, chunks behind it do not exist.  No sibling
// polls (FinArgs.corr set, store_rows 0); the finishing wavefront of the transform reads the SNR window from the row.
template <int NS>
__global__ __launch_bounds__(256) void k_rows_lean(const double* __restrict__ corr, size_t stride, int G, int nblk, PeakArgs pa, FinArgs fa, int rows) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int xs = int(blockIdx.x & 7u), slot = int(blockIdx.x >> 3);
  const int cb = slot % nblk, g = xs + 8 * (slot / nblk);
  if (g >= G) return;                                          // (uniform)
  int stamp_at = 0;
  auto stamp = [&]() {
    if (fa.stamps && tid == 0) fa.stamps[size_t(blockIdx.x) * 8 + stamp_at] = __builtin_amdgcn_s_memrealtime();
    ++stamp_at;
  };
  stamp();
  constexpr int W = kColsOwn;                                  // 62 owned samples per chunk
  const int n = pa.n;
  const int t0 = (cb * 4 + wave) * NS;                         // first chunk of this wavefront
  const double* const r0 = corr + size_t(2 * g) * stride;
  const double* const r1 = 2 * g + 1 < rows ? r0 + stride : r0;
  const double keep1 = 2 * g + 1 < rows ? 1.0 : 0.0;
  cd ro[NS];
  unsigned emask = 0;
#pragma unroll
  for (int i = 0; i < NS; ++i) {
    const int m = W * (t0 + i) + lane - 1;
    const bool ok = m >= 0 && m < n;
    const int mc = ok ? m : 0;
    const double a = r0[mc], b = r1[mc];
    ro[i] = mk(ok ? a : 0.0, ok ? b * keep1 : 0.0);
    if (W * (t0 + i) < n) emask |= 1u << (1 + i);
  }
  stamp();                                                     // 1: loaded
  const int tp = (n - 1) / W;                                  // the row ends inside chunk tp
  const int pslot = tp >= t0 && tp < t0 + NS ? 1 + tp - t0 : -1;
  const bool own = lane >= 1 && lane <= W;
  const bool pvalid = W * tp + lane - 1 < n;
  const int st = t0 + (lane < NS ? lane : 0);
  const bool last = fin_lean_r89<false, true, NS>(ro, mk(0, 0), st, emask, wave, lane, g, cb, nblk, 0, W, rows, 0, lane - 1, own, own, false, pslot,
                                                  pvalid, pa, fa, stamp);
  if (!last) return;
  stamp();                                                     // 5
  if (wave != 0) return;                                       // ONE wavefront finishes the transform's rows
  bool late = false;
  for (int q = lane; q < nblk * fa.pw; q += 64) {
    int spins = 0;
    while (ld_agent(fa.done + size_t(g) * nblk * fa.pw + q) != fa.epoch) {
      if (++spins > kSpinLimit) { late = true; break; }
      __builtin_amdgcn_s_sleep(8);
    }
  }
  if (__ballot(late)) {
    if (lane < 2 && 2 * g + lane < rows) { fa.need[2 * g + lane] = 1; atomicAdd(fa.status + 4, 1); atomicAdd(fa.status + 13, 1); }
    return;
  }
#pragma nounroll
  for (int r = 0; r < 2; ++r)
    if (2 * g + r < rows) fin_row_wave(pa, fa, 2 * g + r, 0, W, lane);
  stamp();                                                     // 6
}

}  // namespace pal
