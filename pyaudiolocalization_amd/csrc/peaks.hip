// peaks.hip - peak selection and correlation metrics for one PHAT row per workgroup (gfx950).
//
// Replaces utils.py:140-181 (threshold, scipy.signal.find_peaks(height, distance), the whole
// fallback chain, window filter, top-num_peaks), utils.py:228-250 (compute_snr /
// compute_peak_to_peak_ratio inputs) and np.max(corr) of main.py:223.
//
// scipy's find_peaks is evaluated lazily and exactly instead of materialising peak lists:
//   - a sample m is a peak iff it is the floor-midpoint of a plateau whose two outer neighbours
//     are strictly lower (end points never qualify)                      (_local_maxima_1d)
//   - the greedy distance suppression keeps peak X iff no KEPT peak of higher priority (height,
//     then position) lies closer than `distance`; that recursion is resolved depth first from the
//     candidate, with a memo, because chains of rising peaks are short   (_select_by_peak_distance)
//   - candidates are visited in descending priority inside the lag window until num_peaks are kept.
// The exact median of |corr| is a radix select over the IEEE-754 bit pattern (11-bit digits, LDS
// histogram, then an in-LDS rank search once the surviving bin holds <= 2048 values).
// Reductions: wavefront (64-lane) shuffles, then one LDS hop across the 4 wavefronts.
#include <cfloat>
#include <climits>
#include <cmath>

#include "engine.h"
#include "reduce.h"

namespace pal {

namespace {

constexpr int kCap = 2048;      // in-LDS rank-search capacity
constexpr int kMemo = 1024;     // resolved peaks remembered per selection
constexpr int kStack = 64;      // depth of the suppression recursion

struct PeakArgs {
  const double* corr;
  size_t stride;
  int n, n2;
  double fs, mult, med;   // med: NaN = no window
  int method, dist, num_peaks, snr_w;
};

struct Shared {
  unsigned hist[2048];
  double list[kCap];
  double red_d[8];
  int red_i[8];
  unsigned scan[kLanes];
  int count;
  // selection state
  int memo_pos[kMemo];
  int memo_kept[kMemo];
  int memo_n;
  int stack_pos[kStack];
  double stack_h[kStack];
  int stack_n;
  int flag;
  double bc_d[4];
  int bc_i[4];
};

__device__ __forceinline__ bool higher(double h1, int m1, double h2, int m2) {   // priority(h1,m1) > priority(h2,m2)
  return h1 > h2 || (h1 == h2 && m1 > m2);
}

// ---- block reductions (result valid in every lane): reduce.h over this kernel's LDS slots ----
__device__ double block_sum(double v, Shared& s, int tid) { return pal::block_sum(v, s.red_d, tid); }
template <int MODE> __device__ void block_arg(double& v, int& i, Shared& s, int tid) {
  pal::block_arg<MODE>(v, i, s.red_d, s.red_i, tid);
}

// ---- scipy _local_maxima_1d, evaluated for one sample ----
__device__ bool peak_mid(const double* c, int n, int m, double& h) {
  if (m < 1 || m > n - 2) return false;
  const double x = c[m];
  int l = m, r = m;
  while (l > 0 && c[l - 1] == x) --l;
  while (r < n - 1 && c[r + 1] == x) ++r;
  if (l < 1 || r > n - 2) return false;
  if (!(c[l - 1] < x) || !(c[r + 1] < x)) return false;
  if (m != (l + r) / 2) return false;
  h = x;
  return true;
}

// ---- radix select digits over the 63 magnitude bits ----
__device__ __forceinline__ int digit_shift(int level) { return level < 5 ? 52 - 11 * level : 0; }
__device__ __forceinline__ unsigned digit_mask(int level) { return level < 5 ? 0x7FFu : 0xFFu; }
__device__ __forceinline__ unsigned long long mag_key(double x) {
  return (unsigned long long)__double_as_longlong(fabs(x));
}
__device__ __forceinline__ bool key_matches(unsigned long long key, int level, unsigned long long prefix) {
  return level == 0 || (key >> digit_shift(level - 1)) == prefix;
}

// find the histogram bin that holds 0-based rank `rank`; returns bin, rank inside it and its population
__device__ void find_bin(Shared& s, int tid, unsigned rank, unsigned& bin, unsigned& inner, unsigned& pop) {
  // 2048 bins = 256 lanes x 8 consecutive bins
  unsigned local = 0;
  for (int k = 0; k < 8; ++k) local += s.hist[tid * 8 + k];
  s.scan[tid] = local;
  __syncthreads();
  if (tid == 0) {
    unsigned acc = 0;
    int t = 0;
    while (t < kLanes - 1 && acc + s.scan[t] <= rank) { acc += s.scan[t]; ++t; }
    int b = t * 8;
    while (b < t * 8 + 7 && acc + s.hist[b] <= rank) { acc += s.hist[b]; ++b; }
    s.bc_i[0] = b;
    s.bc_i[1] = int(rank - acc);
    s.bc_i[2] = int(s.hist[b]);
  }
  __syncthreads();
  bin = unsigned(s.bc_i[0]);
  inner = unsigned(s.bc_i[1]);
  pop = unsigned(s.bc_i[2]);
  __syncthreads();
}

struct Extra {       // side reductions carried by the passes over the row
  int kind;          // 0 none, 1 sums outside [lo,hi) (+ sum (|x|-mabs)^2), 2 sum (x-mean)^2 outside [lo,hi)
  int lo, hi;
  double mean, mabs;
  double out0, out1;
};

// one streaming pass over the row: optional histogram of digit `level` among keys matching `prefix`,
// optional compaction of the matching magnitudes into s.list, optional side reductions
__device__ void stream_pass(const double* c, int n, int tid, Shared& s, bool do_hist, bool do_compact, int level,
                            unsigned long long prefix, Extra& ex) {
  if (do_hist) for (int k = tid; k < 2048; k += kLanes) s.hist[k] = 0;
  if (do_compact && tid == 0) s.count = 0;
  __syncthreads();
  double a0 = 0, a1 = 0;
  const int sh = digit_shift(level);
  const unsigned mk_ = digit_mask(level);
  for (int i = tid; i < n; i += kLanes) {
    const double x = c[i];
    if (do_hist || do_compact) {
      const unsigned long long key = mag_key(x);
      if (key_matches(key, level, prefix)) {
        if (do_hist) atomicAdd(&s.hist[unsigned(key >> sh) & mk_], 1u);
        if (do_compact) {
          const int p = atomicAdd(&s.count, 1);
          if (p < kCap) s.list[p] = fabs(x);
        }
      }
    }
    if (ex.kind == 1) {
      if (i < ex.lo || i >= ex.hi) a0 += x;
      const double d = fabs(x) - ex.mabs;
      a1 += d * d;
    } else if (ex.kind == 2) {
      if (i < ex.lo || i >= ex.hi) { const double d = x - ex.mean; a0 += d * d; }
    }
  }
  if (ex.kind != 0) {
    ex.out0 = block_sum(a0, s, tid);
    ex.out1 = block_sum(a1, s, tid);
  }
  __syncthreads();
}

// value of 0-based rank `inner` among the s.count magnitudes compacted in s.list
__device__ double list_rank(Shared& s, int tid, unsigned inner) {
  const int cnt = s.count;
  if (tid == 0) s.bc_d[0] = 0;
  __syncthreads();
  for (int e = tid; e < cnt; e += kLanes) {
    const double v = s.list[e];
    unsigned below = 0;
    for (int j = 0; j < cnt; ++j) {
      const double u = s.list[j];
      below += (u < v) || (u == v && j < e);
    }
    if (below == inner) s.bc_d[0] = v;
  }
  __syncthreads();
  const double r = s.bc_d[0];
  __syncthreads();
  return r;
}

// radix select of |corr| at 0-based `rank`.  `have_level0` says s.hist already holds the digit-0
// histogram.  Side reductions queued in exq[] ride along with the passes; any left over are run after.
__device__ double select_rank(const double* c, int n, int tid, Shared& s, unsigned rank, bool have_level0, Extra* exq,
                              int& ex_next, int ex_count) {
  Extra none{0, 0, 0, 0, 0, 0, 0};
  unsigned long long prefix = 0;
  int level = 0;
  if (!have_level0) stream_pass(c, n, tid, s, true, false, 0, 0, none);
  unsigned inner = rank, bin, pop;
  for (;;) {
    find_bin(s, tid, inner, bin, inner, pop);
    prefix = (prefix << (level < 5 ? 11 : 8)) | bin;
    ++level;
    if (pop <= unsigned(kCap) || level == 6) break;
    Extra& ex = ex_next < ex_count ? exq[ex_next] : none;
    stream_pass(c, n, tid, s, true, false, level, prefix, ex);
    if (&ex != &none) ++ex_next;
  }
  if (level == 6) return __longlong_as_double((long long)prefix);   // every magnitude bit fixed
  Extra& ex = ex_next < ex_count ? exq[ex_next] : none;
  stream_pass(c, n, tid, s, false, true, level, prefix, ex);
  if (&ex != &none) ++ex_next;
  return list_rank(s, tid, inner);
}

// ---- greedy distance suppression, resolved from one candidate ----
__device__ int memo_find(const Shared& s, int pos) {
  for (int k = 0; k < s.memo_n; ++k)
    if (s.memo_pos[k] == pos) return s.memo_kept[k];
  return -1;
}

// returns 1 kept, 0 suppressed, -1 overflow; block-uniform control flow
__device__ int resolve(const double* c, int n, int dist, int tid, Shared& s, int pos0, double h0) {
  if (tid == 0) { s.stack_n = 1; s.stack_pos[0] = pos0; s.stack_h[0] = h0; s.flag = 0; }
  __syncthreads();
  for (int guard = 0; guard < 100000; ++guard) {
    const int depth = s.stack_n;
    if (depth == 0) break;
    const int p = s.stack_pos[depth - 1];
    const double h = s.stack_h[depth - 1];
    __syncthreads();
    if (memo_find(s, p) >= 0) {                       // resolved while deeper frames ran
      if (tid == 0) s.stack_n = depth - 1;
      __syncthreads();
      continue;
    }
    // neighbours closer than dist with higher priority
    int any_kept = 0;
    double bh = 0;
    int bm = -1;
    for (int o = tid - (dist - 1); o <= dist - 1; o += kLanes) {
      if (o == 0) continue;
      const int m = p + o;
      double hm;
      if (peak_mid(c, n, m, hm) && higher(hm, m, h, p)) {
        const int st = memo_find(s, m);
        if (st == 1) any_kept = 1;
        else if (st < 0 && (bm < 0 || higher(hm, m, bh, bm))) { bh = hm; bm = m; }
      }
    }
    any_kept = __syncthreads_or(any_kept);
    block_arg<2>(bh, bm, s, tid);
    if (tid == 0) {
      if (any_kept || bm < 0) {
        if (s.memo_n < kMemo) {
          s.memo_pos[s.memo_n] = p;
          s.memo_kept[s.memo_n] = any_kept ? 0 : 1;
          ++s.memo_n;
        } else {
          s.flag = 1;
        }
        s.stack_n = depth - 1;
      } else if (depth < kStack) {
        s.stack_pos[depth] = bm;
        s.stack_h[depth] = bh;
        s.stack_n = depth + 1;
      } else {
        s.flag = 1;
      }
    }
    __syncthreads();
    if (s.flag) return -1;
  }
  __syncthreads();
  const int st = memo_find(s, pos0);
  __syncthreads();
  return st < 0 ? -1 : st;
}

__device__ __forceinline__ bool in_window(int m, int n2, double fs, double med) {
  return fabs(double(m - (n2 - 1)) / fs) <= med;       // abs(time_lags[k]) <= max_expected_delay (utils.py:163)
}

// highest-priority peak with height >= thr inside [wlo, whi] (exact window test when windowed) and
// priority below (bh, bm); returns false when none
__device__ bool next_candidate(const PeakArgs& a, const double* c, int tid, Shared& s, double thr, bool windowed, int wlo,
                               int whi, double bound_h, int bound_m, double& ch, int& cm) {
  double bh = 0;
  int bm = -1;
  for (int m = wlo + tid; m <= whi; m += kLanes) {
    double hm;
    if (!peak_mid(c, a.n, m, hm)) continue;
    if (!(hm >= thr)) continue;
    if (windowed && !in_window(m, a.n2, a.fs, a.med)) continue;
    if (!higher(bound_h, bound_m, hm, m)) continue;
    if (bm < 0 || higher(hm, m, bh, bm)) { bh = hm; bm = m; }
  }
  block_arg<2>(bh, bm, s, tid);
  ch = bh;
  cm = bm;
  return bm >= 0;
}

// top-num_peaks kept peaks >= thr in the window; returns count or -1 on overflow
__device__ int select_peaks(const PeakArgs& a, const double* c, int tid, Shared& s, double thr, bool windowed, int wlo,
                            int whi, double first_h, int first_m, int* sel, double* selh) {
  if (tid == 0) s.memo_n = 0;
  __syncthreads();
  double bound_h = INFINITY;
  int bound_m = INT_MAX;
  int count = 0;
  bool use_first = first_m >= 0;
  while (count < a.num_peaks) {
    double ch;
    int cm;
    if (use_first) {
      ch = first_h; cm = first_m; use_first = false;
    } else if (!next_candidate(a, c, tid, s, thr, windowed, wlo, whi, bound_h, bound_m, ch, cm)) {
      break;
    }
    const int st = resolve(c, a.n, a.dist, tid, s, cm, ch);
    if (st < 0) return -1;
    if (st == 1) { sel[count] = cm; selh[count] = ch; ++count; }
    bound_h = ch;
    bound_m = cm;
  }
  return count;
}

__global__ __launch_bounds__(256) void k_peaks(PeakArgs a, pal_pair_record* table, int32_t* ksel_multi, int* status) {
  __shared__ Shared s;
  const int tid = threadIdx.x;
  const int row = blockIdx.x;
  const double* c = a.corr + size_t(row) * a.stride;
  const int n = a.n;

  // ---- pass A: max/argmax, min, sum|x|, exponent histogram, highest local maximum ----
  for (int k = tid; k < 2048; k += kLanes) s.hist[k] = 0;
  __syncthreads();
  double vmax = -INFINITY, vmin = INFINITY, sabs = 0, hb = 0;
  int imax = -1, imin = -1, mb = -1;
  const bool want_median = a.method == 0;
  for (int i = tid; i < n; i += kLanes) {
    const double x = c[i];
    if (imax < 0 || x > vmax) { vmax = x; imax = i; }
    if (imin < 0 || x < vmin) { vmin = x; imin = i; }
    sabs += fabs(x);
    if (want_median) atomicAdd(&s.hist[unsigned(mag_key(x) >> 52) & 0x7FFu], 1u);
    if (i >= 1 && i <= n - 2 && c[i - 1] < x) {          // rising edge: owns the plateau that starts here
      int r = i + 1;
      while (r < n - 1 && c[r] == x) ++r;
      if (c[r] < x) {
        const int m = (i + r - 1) / 2;
        if (mb < 0 || higher(x, m, hb, mb)) { hb = x; mb = m; }
      }
    }
  }
  block_arg<0>(vmax, imax, s, tid);
  block_arg<1>(vmin, imin, s, tid);
  block_arg<2>(hb, mb, s, tid);
  sabs = block_sum(sabs, s, tid);
  const double mean_abs = sabs / double(n);               // np.mean(np.abs(corr)) (utils.py:155)

  // ---- SNR window (utils.py:244-247) and the side reductions that ride on later passes ----
  const int lo = imax - a.snr_w > 0 ? imax - a.snr_w : 0;
  const int hi = imax + a.snr_w < n ? imax + a.snr_w : n;
  const int noise_n = lo + (n - hi);
  Extra exq[2];
  exq[0] = Extra{1, lo, hi, 0.0, mean_abs, 0, 0};
  exq[1] = Extra{2, lo, hi, 0.0, 0.0, 0, 0};
  int ex_next = 0;
  double thr1;
  if (want_median) {
    double med;
    if (n & 1) {
      med = select_rank(c, n, tid, s, unsigned(n / 2), true, exq, ex_next, 1);
    } else {
      const double m0 = select_rank(c, n, tid, s, unsigned(n / 2 - 1), true, exq, ex_next, 1);
      const double m1 = select_rank(c, n, tid, s, unsigned(n / 2), false, exq, ex_next, 1);
      med = (m0 + m1) * 0.5;                              // np.median of an even count
    }
    if (ex_next < 1) { stream_pass(c, n, tid, s, false, false, 0, 0, exq[0]); ex_next = 1; }
    thr1 = a.mult * med;
  } else {
    stream_pass(c, n, tid, s, false, false, 0, 0, exq[0]);
    ex_next = 1;
    thr1 = a.mult * (mean_abs + sqrt(exq[0].out1 / double(n)));   // mean + std of |corr| (utils.py:147)
  }
  exq[1].mean = exq[0].out0 / double(noise_n);
  stream_pass(c, n, tid, s, false, false, 0, 0, exq[1]);
  const double noise = sqrt(exq[1].out0 / double(noise_n));
  const double snr = noise == 0.0 ? INFINITY : vmax / noise;

  // ---- fallback chain (utils.py:152-179) ----
  int branch = 0;
  int sel[PAL_MAX_PEAKS];
  double selh[PAL_MAX_PEAKS];
  int count = 0;
  bool overflow = false;
  const bool windowed = !isnan(a.med);
  int wlo = 1, whi = n - 2;
  if (windowed) {
    const double span = a.med * a.fs;
    const double c0 = double(a.n2 - 1);
    const double flo = c0 - span - 2.0, fhi = c0 + span + 2.0;
    wlo = flo > 1.0 ? (flo < double(n) ? int(flo) : n) : 1;
    whi = fhi < double(n - 2) ? (fhi > -1.0 ? int(fhi) : -1) : n - 2;
  }
  double thr = thr1;
  bool argmax_fallback = false;
  if (!(mb >= 0 && hb >= thr1)) {                          // no peak reaches the primary threshold
    branch |= PAL_BR_ALT_THRESHOLD;
    thr = mean_abs;
    if (!(mb >= 0 && hb >= mean_abs)) { branch |= PAL_BR_ARGMAX_NO_PEAKS; argmax_fallback = true; }
  }
  if (!argmax_fallback) {
    const bool first_ok = !windowed;                       // unwindowed: the best peak is already known
    count = select_peaks(a, c, tid, s, thr, windowed, wlo, whi, first_ok ? hb : 0.0, first_ok ? mb : -1, sel, selh);
    if (count < 0) overflow = true;
    if (count == 0 && windowed) {
      branch |= PAL_BR_WINDOW_RETRY;
      count = select_peaks(a, c, tid, s, mean_abs, true, wlo, whi, 0.0, -1, sel, selh);
      if (count < 0) overflow = true;
      if (count == 0) { branch |= PAL_BR_ARGMAX_WINDOW; argmax_fallback = true; }
    }
  }
  if (argmax_fallback || overflow) { sel[0] = imax; selh[0] = vmax; count = 1; }

  if (tid == 0) {
    pal_pair_record r;
    r.k_sel = sel[0];
    r.branch = branch;
    r.k_argmax = imax;
    r.n_sel = count;
    r.cmax = vmax;
    r.cmin = vmin;
    r.snr = snr;
    r.sel_height = selh[0];
    table[row] = r;
    if (ksel_multi)
      for (int k = 0; k < PAL_MAX_PEAKS; ++k) ksel_multi[size_t(row) * PAL_MAX_PEAKS + k] = k < count ? sel[k] : -1;
    if (overflow) atomicOr(status, 1);
  }
}

// metrics only (max, min, argmax, snr) for rows that are not PHAT sequences
__global__ __launch_bounds__(256) void k_metrics(PeakArgs a, pal_pair_record* table) {
  __shared__ Shared s;
  const int tid = threadIdx.x;
  const double* c = a.corr + size_t(blockIdx.x) * a.stride;
  const int n = a.n;
  double vmax = -INFINITY, vmin = INFINITY;
  int imax = -1, imin = -1;
  for (int i = tid; i < n; i += kLanes) {
    const double x = c[i];
    if (imax < 0 || x > vmax) { vmax = x; imax = i; }
    if (imin < 0 || x < vmin) { vmin = x; imin = i; }
  }
  block_arg<0>(vmax, imax, s, tid);
  block_arg<1>(vmin, imin, s, tid);
  const int lo = imax - a.snr_w > 0 ? imax - a.snr_w : 0;
  const int hi = imax + a.snr_w < n ? imax + a.snr_w : n;
  const int noise_n = lo + (n - hi);
  Extra e1{1, lo, hi, 0.0, 0.0, 0, 0};
  stream_pass(c, n, tid, s, false, false, 0, 0, e1);
  Extra e2{2, lo, hi, e1.out0 / double(noise_n), 0.0, 0, 0};
  stream_pass(c, n, tid, s, false, false, 0, 0, e2);
  const double noise = sqrt(e2.out0 / double(noise_n));
  if (tid == 0) {
    pal_pair_record r;
    r.k_sel = imax; r.branch = 0; r.k_argmax = imax; r.n_sel = 0;
    r.cmax = vmax; r.cmin = vmin; r.snr = noise == 0.0 ? INFINITY : vmax / noise; r.sel_height = vmax;
    table[blockIdx.x] = r;
  }
}

}  // namespace

static int* g_status_dev(Engine* e) {
  void* p = nullptr;
  if (e->scratch(7, 64, &p) != PAL_OK) return nullptr;
  return static_cast<int*>(p);
}

int Engine::peaks(const double* corr, size_t stride, int rows, int n, int n2, const pal_phat_params& prm,
                  pal_pair_record* table, int32_t* ksel_multi) {
  if (rows <= 0) return PAL_OK;
  if (prm.num_peaks < 1 || prm.num_peaks > PAL_MAX_PEAKS) return fail(PAL_ERR_INVALID, "num_peaks %d outside 1..%d", prm.num_peaks, PAL_MAX_PEAKS);
  if (prm.peak_distance < 1) return fail(PAL_ERR_INVALID, "`distance` must be greater or equal to 1");
  if (n < 1) return fail(PAL_ERR_INVALID, "empty correlation");
  int* status = g_status_dev(this);
  if (!status) return fail(PAL_ERR_NOMEM, "status word");
  PeakArgs a;
  a.corr = corr; a.stride = stride; a.n = n; a.n2 = n2;
  a.fs = prm.fs; a.mult = prm.threshold_multiplier; a.med = prm.max_expected_delay;
  a.method = prm.threshold_method; a.dist = prm.peak_distance; a.num_peaks = prm.num_peaks;
  const int w = int(0.01 * double(n));                       // utils.py:244
  a.snr_w = w > 1 ? w : 1;
  ProfScope ps(this, prm.threshold_method < 0 ? "k_metrics" : "k_peaks");
  if (prm.threshold_method < 0) {
    k_metrics<<<dim3(rows), dim3(kLanes), 0, stream>>>(a, table);
  } else {
    k_peaks<<<dim3(rows), dim3(kLanes), 0, stream>>>(a, table, ksel_multi, status);
  }
  return check(hipGetLastError(), "k_peaks");
}

}  // namespace pal
