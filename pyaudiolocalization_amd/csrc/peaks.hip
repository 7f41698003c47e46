// peaks.hip - peak selection and correlation metrics of the PHAT rows (gfx950), three launches per group:
//
//   k_peak_pivots  one workgroup per row: an 8192-point block sample gives the shifts of the one-pass variances
//                  and two pivots that bracket the median of |corr| (6 sigma of the sample's rank error)
//   k_peak_stream  the row (n doubles, 0.7 MB at 44.1 kHz x 1 s) is cut into segments of a few 2048-element
//                  tiles; every 256-lane workgroup streams ONE segment with 16-byte loads, four in flight per
//                  lane plus a register double buffer: max / first argmax, min, shifted sums for the SNR and
//                  the 'adaptive' threshold, the highest local maximum, the count below the lower pivot, and
//                  the values between the pivots (compacted in LDS, flushed to the row's global list with one
//                  atomic per workgroup).  Small LDS, no inter-workgroup waits: bandwidth-bound, and light
//                  enough to run beside the FFT passes of the next launch group.
//   k_peak_finish  one workgroup per row: merges the segment results, SNR window, exact median by a rank search
//                  inside the bracket list (radix select over the IEEE-754 bit pattern if the pivots missed or
//                  the list overflowed), then scipy's find_peaks and the reference's fallback chain.
//
// Replaces utils.py:140-181 (threshold, scipy.signal.find_peaks(height, distance), the whole fallback chain,
// window filter, top-num_peaks), utils.py:228-250 (compute_snr / compute_peak_to_peak_ratio inputs) and
// np.max(corr) of main.py:223.
//
// The tiles of the stream are branch-free (guarded loads serialise on s_waitcnt); lane 0 / lane 63 of every
// wavefront leave their outer element's peak test to a short edge pass.
//
// scipy's find_peaks is evaluated lazily and exactly instead of materialising peak lists:
//   - a sample m is a peak iff it is the floor-midpoint of a plateau whose two outer neighbours
//     are strictly lower (end points never qualify)                      (_local_maxima_1d)
//   - the greedy distance suppression keeps peak X iff no KEPT peak of higher priority (height,
//     then position) lies closer than `distance`; that recursion is resolved depth first from the
//     candidate, with a memo, because chains of rising peaks are short   (_select_by_peak_distance)
//   - candidates are visited in descending priority inside the lag window until num_peaks are kept.
// Reductions: wavefront (64-lane) shuffles, then one LDS hop across the wavefronts.
#include <cfloat>
#include <climits>
#include <cmath>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <type_traits>
#include <vector>

#include "engine.h"
#include "peak_types.h"
#include "pfa_sample.h"
#include "reduce.h"

namespace pal {

namespace {

constexpr int kT = 512;           // lanes of the pivot and finish kernels (one row per workgroup)
constexpr int kNW = kT / 64;
constexpr int kTS = 256;          // lanes of the stream kernel (one segment per workgroup)
constexpr int kNWS = kTS / 64;
constexpr int kLoc = 3072;        // LDS capacity of one segment's share of it (a segment of 11 tiles brackets ~1100 values)
constexpr int kMaxTilesPerSeg = 11;
constexpr int kSample = 8192;     // block sample that places the pivots
constexpr int kBins = 2048;       // histogram bins (sample pivots, list search, radix digits)
constexpr int kSmall = 1024;      // exact rank search capacity
constexpr int kMemo = 1024;       // resolved peaks remembered per selection
constexpr int kStack = 64;        // depth of the suppression recursion
constexpr int kMaxStaged = 136;   // segments whose results the finish launch stages through LDS (the fused column pass has at most ceil(8192 / 62) = 133 blocks)
constexpr int kUnroll = 4;        // 16-byte loads in flight per lane
constexpr int kTile = kTS * kUnroll;   // element pairs per tile of the stream

struct Shared {                          // pivot and finish kernels
  unsigned hist[kBins];
  double small[kSmall];
  double red_d[kNW];
  double many[kNW * 8];
  int red_i[kNW];
  long long red_l[kNW];
  unsigned wave_tot[kNW];
  int count, count2;
  int memo_pos[kMemo];
  int memo_kept[kMemo];
  int memo_n;
  int stack_pos[kStack];
  double stack_h[kStack];
  int stack_n;
  int flag;
  double bc_d[4];
  int bc_i[4];
  int sel_pos[PAL_MAX_PEAKS];            // selected peaks of the finish launch (written and read by lane 0)
  double sel_h[PAL_MAX_PEAKS];
};

struct StreamWave {                      // one wavefront's share of a segment's statistics
  double vmax, vmin, hb, sums[5];
  int imax, mb;
};

struct StreamShared {                    // stream kernel
  double list[kLoc + 1];                 // + one dump slot for the unconditional stores of values outside the bracket
  StreamWave wave[kNWS];
  int count;
  int bc_i;
};

__device__ double bsum(double v, Shared& s, int tid) { return block_sum<kNW>(v, s.red_d, tid); }
template <int N, int NW> __device__ void bsum_many(double (&v)[N], double* many, int tid) {   // N sums, one barrier pair
#pragma unroll
  for (int q = 0; q < N; ++q)
    for (int o = 32; o > 0; o >>= 1) v[q] += __shfl_down(v[q], o, 64);
  __syncthreads();
  if ((tid & 63) == 0) {
#pragma unroll
    for (int q = 0; q < N; ++q) many[(tid >> 6) * N + q] = v[q];
  }
  __syncthreads();
#pragma unroll
  for (int q = 0; q < N; ++q) {
    double r = 0;
    for (int k = 0; k < NW; ++k) r += many[k * N + q];
    v[q] = r;
  }
}
__device__ long long bsum_ll(long long v, Shared& s, int tid) { return block_sum_ll<kNW>(v, s.red_l, tid); }
template <int MODE> __device__ void barg(double& v, int& i, Shared& s, int tid) {
  block_arg<MODE, kNW>(v, i, s.red_d, s.red_i, tid);
}

// exclusive prefix sum over the workgroup (one value per lane)
__device__ unsigned block_excl_scan(unsigned v, Shared& s, int tid) {
  const int lane = tid & 63, wave = tid >> 6;
  unsigned inc = v;
  for (int o = 1; o < 64; o <<= 1) {
    const unsigned t = __shfl_up(inc, o, 64);
    if (lane >= o) inc += t;
  }
  __syncthreads();
  if (lane == 63) s.wave_tot[wave] = inc;
  __syncthreads();
  unsigned before = 0;
  for (int k = 0; k < wave; ++k) before += s.wave_tot[k];
  return before + inc - v;
}

// bin of s.hist that holds 0-based rank `rank` (2 bins per lane); rank inside the bin and its population
__device__ void find_bin(Shared& s, int tid, unsigned rank, unsigned& bin, unsigned& inner, unsigned& pop) {
  constexpr int kPer = kBins / kT;                  // consecutive bins owned by one lane
  unsigned h[kPer], own = 0;
#pragma unroll
  for (int q = 0; q < kPer; ++q) { h[q] = s.hist[kPer * tid + q]; own += h[q]; }
  unsigned ex = block_excl_scan(own, s, tid);
  if (tid == 0) s.bc_i[0] = -1;
  __syncthreads();
#pragma unroll
  for (int q = 0; q < kPer; ++q) {
    if (rank >= ex && rank < ex + h[q]) { s.bc_i[0] = kPer * tid + q; s.bc_i[1] = int(rank - ex); s.bc_i[2] = int(h[q]); }
    ex += h[q];
  }
  __syncthreads();
  if (s.bc_i[0] < 0) { bin = kBins - 1; inner = 0; pop = 0; }   // rank beyond the histogram's total
  else { bin = unsigned(s.bc_i[0]); inner = unsigned(s.bc_i[1]); pop = unsigned(s.bc_i[2]); }
  __syncthreads();
}

// bins of s.hist that hold the 0-based ranks `ra` <= `rb` (one scan, half the barriers of two find_bin calls)
__device__ void find_two_bins(Shared& s, int tid, unsigned ra, unsigned rb, unsigned& bin_a, unsigned& bin_b) {
  constexpr int kPer = kBins / kT;
  unsigned h[kPer], own = 0;
#pragma unroll
  for (int q = 0; q < kPer; ++q) { h[q] = s.hist[kPer * tid + q]; own += h[q]; }
  unsigned ex = block_excl_scan(own, s, tid);
  if (tid == 0) { s.bc_i[0] = -1; s.bc_i[1] = -1; }
  __syncthreads();
#pragma unroll
  for (int q = 0; q < kPer; ++q) {
    if (ra >= ex && ra < ex + h[q]) s.bc_i[0] = kPer * tid + q;
    if (rb >= ex && rb < ex + h[q]) s.bc_i[1] = kPer * tid + q;
    ex += h[q];
  }
  __syncthreads();
  bin_a = s.bc_i[0] < 0 ? unsigned(kBins - 1) : unsigned(s.bc_i[0]);   // rank beyond the histogram's total
  bin_b = s.bc_i[1] < 0 ? unsigned(kBins - 1) : unsigned(s.bc_i[1]);
  __syncthreads();
}

// wavefront-aggregated append to an LDS list (one LDS atomic per wavefront)
__device__ __forceinline__ void append(bool pred, double v, double* list, int* counter, int cap, int lane) {
  const unsigned long long mask = __ballot(pred);
  if (mask == 0) return;
  const int leader = __ffsll((long long)mask) - 1;
  int base = 0;
  if (lane == leader) base = atomicAdd(counter, __popcll(mask));
  base = __shfl(base, leader, 64);
  if (pred) {
    const int at = base + __popcll(mask & ((1ull << lane) - 1ull));
    if (at < cap) list[at] = v;
  }
}

// ---- scipy _local_maxima_1d, evaluated for one sample ----
__device__ __forceinline__ bool peak_mid(const double* c, int n, int m, double& h) {
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

__device__ bool small_rank(Shared& s, int tid, unsigned inner, double& out);

// ---- exact rank inside the row's bracket list (global memory, L2-resident) whose values lie in [lo, hi] ----
// linear bins spread the bracket over the histogram; the winning bin (a handful of values) is ranked by counting.
// returns false when that bin is too crowded for the exact search (caller falls back to the radix select)
__device__ bool list_select(Shared& s, int tid, const double* __restrict__ list, int cnt, unsigned rank, double lo, double hi,
                            double& out) {
  constexpr int kKeep = 16;                                    // list values a lane keeps in registers between the two passes
  for (int k = tid; k < kBins; k += kT) s.hist[k] = 0;
  if (tid == 0) s.count2 = 0;
  __syncthreads();
  const double span = hi - lo;
  const double inv = (span > 0 && isfinite(span)) ? double(kBins - 1) / span : 0.0;
  auto bin_of = [&](double v) {
    int b = int((v - lo) * inv);
    return b < 0 ? 0 : (b > kBins - 1 ? kBins - 1 : b);
  };
  double keep[kKeep];
#pragma unroll
  for (int q = 0; q < kKeep; ++q) {                            // all loads in flight together
    const int e = tid + q * kT;
    keep[q] = e < cnt ? list[e] : 0.0;
  }
#pragma unroll
  for (int q = 0; q < kKeep; ++q)
    if (tid + q * kT < cnt) atomicAdd(&s.hist[bin_of(keep[q])], 1u);
  for (int e = tid + kKeep * kT; e < cnt; e += kT) atomicAdd(&s.hist[bin_of(list[e])], 1u);
  __syncthreads();
  unsigned bin, inner, pop;
  find_bin(s, tid, rank, bin, inner, pop);
  if (pop == 0 || pop > unsigned(kSmall)) return false;
  // the winning bin holds a handful of values: a lane that owns one takes a slot with its own LDS atomic
#pragma unroll
  for (int q = 0; q < kKeep; ++q)
    if (tid + q * kT < cnt && unsigned(bin_of(keep[q])) == bin) {
      const int at = atomicAdd(&s.count2, 1);
      if (at < kSmall) s.small[at] = keep[q];
    }
  for (int e = tid + kKeep * kT; e < cnt; e += kT) {
    const double v = list[e];
    if (unsigned(bin_of(v)) == bin) {
      const int at = atomicAdd(&s.count2, 1);
      if (at < kSmall) s.small[at] = v;
    }
  }
  __syncthreads();
  return small_rank(s, tid, inner, out);
}

// exact value of 0-based rank `inner` among the s.count2 values collected in s.small (false: more than its capacity)
__device__ bool small_rank(Shared& s, int tid, unsigned inner, double& out) {
  if (s.count2 > kSmall) { __syncthreads(); return false; }
  const int m = s.count2;
  for (int e = tid; e < m; e += kT) {
    const double v = s.small[e];
    unsigned below = 0;
    for (int j = 0; j < m; ++j) {
      const double u = s.small[j];
      below += (u < v) || (u == v && j < e);
    }
    if (below == inner) s.bc_d[0] = v;
  }
  __syncthreads();
  out = s.bc_d[0];
  __syncthreads();
  return true;
}

// Exact order statistic `rank_in` (0-based, counted inside the interval) of |c| among the row's samples with
// lo <= |c| < hi: one pass over the stored row.  The fused column pass knows the interval from its histograms (about
// 0.2 % of the row); used only when a threshold comparison falls inside mult x that interval.
__device__ bool interval_select(Shared& s, int tid, const double* __restrict__ c, int n, double lo, double hi, unsigned rank_in,
                                double& out) {
  if (tid == 0) s.count2 = 0;
  __syncthreads();
  for (int i0 = 0; i0 < n; i0 += 4 * kT) {
    double v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = i0 + u * kT + tid;
      v[u] = i < n ? fabs(c[i]) : -1.0;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) append(v[u] >= lo && v[u] < hi, v[u], s.small, &s.count2, kSmall, tid & 63);
  }
  __syncthreads();
  if (unsigned(s.count2) <= rank_in) { __syncthreads(); return false; }
  return small_rank(s, tid, rank_in, out);
}

// ---- fallback: radix select over the 63 magnitude bits, re-reading the row ----
__device__ __forceinline__ int digit_shift(int level) { return level < 5 ? 52 - 11 * level : 0; }
__device__ __forceinline__ unsigned digit_mask(int level) { return level < 5 ? 0x7FFu : 0xFFu; }
__device__ __forceinline__ unsigned long long mag_key(double x) { return (unsigned long long)__double_as_longlong(fabs(x)); }

__device__ __forceinline__ double radix_select(const double* c, int n, int tid, Shared& s, unsigned rank) {
  unsigned long long prefix = 0;
  unsigned inner = rank, bin, pop;
  for (int level = 0; level < 6; ++level) {
    for (int k = tid; k < kBins; k += kT) s.hist[k] = 0;
    __syncthreads();
    const int sh = digit_shift(level);
    const unsigned mk_ = digit_mask(level);
    for (int i = tid; i < n; i += kT) {
      const unsigned long long key = mag_key(c[i]);
      if (level == 0 || (key >> digit_shift(level - 1)) == prefix) atomicAdd(&s.hist[unsigned(key >> sh) & mk_], 1u);
    }
    __syncthreads();
    find_bin(s, tid, inner, bin, inner, pop);
    prefix = (prefix << (level < 5 ? 11 : 8)) | bin;
    if (pop <= unsigned(kSmall) && level < 5) {
      // few survivors: rank them exactly
      if (tid == 0) s.count2 = 0;
      __syncthreads();
      for (int i0 = 0; i0 < n; i0 += kT) {
        const int i = i0 + tid;
        bool hit = false;
        double v = 0;
        if (i < n) {
          v = fabs(c[i]);
          hit = (mag_key(v) >> sh) == prefix;
        }
        append(hit, v, s.small, &s.count2, kSmall, tid & 63);
      }
      __syncthreads();
      const int m = s.count2;
      for (int e = tid; e < m; e += kT) {
        const double v = s.small[e];
        unsigned below = 0;
        for (int j = 0; j < m; ++j) {
          const double u = s.small[j];
          below += (u < v) || (u == v && j < e);
        }
        if (below == inner) s.bc_d[0] = v;
      }
      __syncthreads();
      const double r = s.bc_d[0];
      __syncthreads();
      return r;
    }
  }
  return __longlong_as_double((long long)prefix);   // every magnitude bit fixed: all survivors are equal
}

// ---- greedy distance suppression, resolved from one candidate ----
__device__ int memo_find(const Shared& s, int pos) {
  for (int k = 0; k < s.memo_n; ++k)
    if (s.memo_pos[k] == pos) return s.memo_kept[k];
  return -1;
}

// returns 1 kept, 0 suppressed, -1 overflow; workgroup-uniform control flow
__device__ __forceinline__ int resolve(const double* c, int n, int dist, int tid, Shared& s, int pos0, double h0, int memo_cap, int stack_cap) {
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
    int any_kept = 0;                                 // neighbours closer than dist with higher priority
    double bh = 0;
    int bm = -1;
    for (int o = tid - (dist - 1); o <= dist - 1; o += kT) {
      if (o == 0) continue;
      const int m = p + o;
      if (m < 1 || m > n - 2) continue;
      const double xl = c[m - 1], xr = c[m + 1];             // the three reads are independent: one round trip, not three
      double hm = c[m];
      if (!(xl < hm && xr < hm)) {                            // not a strict peak: a plateau midpoint, or nothing
        if (!(xl == hm || xr == hm) || !peak_mid(c, n, m, hm)) continue;
      }
      if (higher(hm, m, h, p)) {
        const int st = memo_find(s, m);
        if (st == 1) any_kept = 1;
        else if (st < 0 && (bm < 0 || higher(hm, m, bh, bm))) { bh = hm; bm = m; }
      }
    }
    any_kept = __syncthreads_or(any_kept);
    barg<2>(bh, bm, s, tid);
    if (tid == 0) {
      if (any_kept || bm < 0) {
        if (s.memo_n < memo_cap) {
          s.memo_pos[s.memo_n] = p;
          s.memo_kept[s.memo_n] = any_kept ? 0 : 1;
          ++s.memo_n;
        } else {
          s.flag = 1;
        }
        s.stack_n = depth - 1;
      } else if (depth < stack_cap) {
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

// ---- the same decision without capacity limits: scipy's greedy pass itself, over every peak above the candidate ----
// resolve() answers "is the candidate kept?" depth first with an on-chip memo (1024 resolved peaks, 64 frames): enough for any
// PHAT row met so far, but utils.py:152 has no such limit.  When it overflows, the row takes this path: a bitmap of the row's
// peaks (scipy's _local_maxima_1d, plateau midpoints included) and a bitmap of the positions already suppressed, both in
// global memory; then _select_by_peak_distance literally - the highest peak not yet suppressed is kept and suppresses every
// position closer than `distance` - until the candidate's priority is reached.  Milliseconds per row, exact for any input.
__device__ int resolve_slow(const double* c, int n, int dist, int tid, Shared& s, unsigned* bits, int pos0, double h0) {
  const int words = (n + 31) / 32;
  unsigned* peak = bits;                                       // bit m: sample m is a peak
  unsigned* gone = bits + words;                               // bit m: suppressed (or kept: not to be found again)
  for (int w = tid; w < words; w += kT) {
    unsigned pk = 0;
    for (int b = 0; b < 32; ++b) {
      const int m = 32 * w + b;
      if (m < 1 || m > n - 2) continue;
      const double xl = c[m - 1], x = c[m], xr = c[m + 1];
      double hm;
      if ((xl < x && xr < x) || ((xl == x || xr == x) && peak_mid(c, n, m, hm))) pk |= 1u << b;
    }
    peak[w] = pk;
    gone[w] = 0;
  }
  __syncthreads();
  for (int guard = 0; guard < n; ++guard) {
    // the highest peak above the candidate that is still standing
    double bh = 0;
    int bm = -1;
    for (int w = tid; w < words; w += kT) {
      unsigned live = peak[w] & ~gone[w];
      while (live) {
        const int b = __ffs(int(live)) - 1;
        live &= live - 1;
        const int m = 32 * w + b;
        const double hm = c[m];
        if (higher(hm, m, h0, pos0) && (bm < 0 || higher(hm, m, bh, bm))) { bh = hm; bm = m; }
      }
    }
    barg<2>(bh, bm, s, tid);
    if (bm < 0) break;                                         // nothing above the candidate is left: it is kept
    if (bm - pos0 < dist && pos0 - bm < dist) return 0;        // a kept higher peak closer than `distance`
    const int lo = bm - (dist - 1) > 0 ? bm - (dist - 1) : 0, hi = bm + (dist - 1) < n - 1 ? bm + (dist - 1) : n - 1;
    for (int m = lo + tid; m <= hi; m += kT) atomicOr(&gone[m >> 5], 1u << (m & 31));
    __syncthreads();
  }
  return 1;
}

__device__ __forceinline__ bool in_window(int m, int n2, double fs, double med) {
  return fabs(double(m - (n2 - 1)) / fs) <= med;       // abs(time_lags[k]) <= max_expected_delay (utils.py:163)
}

// highest-priority peak with height >= thr inside [wlo, whi] (exact window test when windowed) and
// priority below (bound_h, bound_m); returns false when none
struct SelArgs {            // by value: a reference to the kernel's argument struct would force it (and the address
  int n, n2, dist, num_peaks;   // space of every pointer in it) through private memory
  double fs, med;
  unsigned* bits;           // this row's bitmaps for resolve_slow
  int memo_cap, stack_cap;
};

__device__ __forceinline__ bool next_candidate(const SelArgs a, const double* c, int tid, Shared& s, double thr, bool windowed, int wlo,
                               int whi, double bound_h, int bound_m, double& ch, int& cm) {
  double bh = 0;
  int bm = -1;
  // four samples per lane and round with all twelve loads in flight (the scan is latency-bound: a dependent
  // round trip per sample made it the longest phase of the finish kernel); plateaus take the exact slow test
  for (int m0 = wlo + tid; m0 <= whi; m0 += 4 * kT) {
    double xl[4], xc[4], xr[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int m = m0 + u * kT;
      const int mm = m <= whi ? m : whi;                       // wlo >= 1 and whi <= n - 2: all three reads are in range
      xl[u] = c[mm - 1];
      xc[u] = c[mm];
      xr[u] = c[mm + 1];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int m = m0 + u * kT;
      if (m > whi) continue;
      double hm = xc[u];
      if (!(xl[u] < hm && xr[u] < hm)) {                       // not a strict peak: a plateau midpoint, or nothing
        if (!(xl[u] == hm || xr[u] == hm) || !peak_mid(c, a.n, m, hm)) continue;
      }
      if (!(hm >= thr)) continue;
      if (windowed && !in_window(m, a.n2, a.fs, a.med)) continue;
      if (!higher(bound_h, bound_m, hm, m)) continue;
      if (bm < 0 || higher(hm, m, bh, bm)) { bh = hm; bm = m; }
    }
  }
  barg<2>(bh, bm, s, tid);
  ch = bh;
  cm = bm;
  return bm >= 0;
}

// top-num_peaks kept peaks >= thr in the window; returns count or -1 on overflow.  The threshold may be known as an
// interval only (tlo <= thr <= thi, fused column pass): candidates arrive in descending height, one at or above thi
// passes, one below tlo ends the search, and for one in between the function returns -2: the caller makes the
// threshold exact (tlo == thi) and calls again.
__device__ __forceinline__ int select_peaks(const SelArgs a, const double* c, int tid, Shared& s, double tlo, double thi,
                            bool windowed, int wlo, int whi, double first_h, int first_m) {
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
    } else {
      if (!next_candidate(a, c, tid, s, tlo, windowed, wlo, whi, bound_h, bound_m, ch, cm)) break;
      if (!(ch >= thi)) return -2;                             // inside the threshold's interval (never when tlo == thi)
    }
    int st = resolve(c, a.n, a.dist, tid, s, cm, ch, a.memo_cap, a.stack_cap);
    if (st < 0 && a.bits) {                                              // memo or stack exhausted: the exact slow path, then a fresh memo
      st = resolve_slow(c, a.n, a.dist, tid, s, a.bits, cm, ch);
      __syncthreads();
      if (tid == 0) s.memo_n = 0;
      __syncthreads();
    }
    if (st < 0) return -1;
    if (st == 1) {                                             // (the list lives in LDS: 48 registers per lane otherwise)
      if (tid == 0) { s.sel_pos[count] = cm; s.sel_h[count] = ch; }
      ++count;
    }
    bound_h = ch;
    bound_m = cm;
  }
  return count;
}

struct Stream {            // per-lane accumulators of the single pass over a segment
  double vmax, vmin, hb;
  int imax, imin, mb;      // < 0: nothing recorded yet
  double s1, s2, a1, a2;   // sums of (x-K0), (x-K0)^2, (|x|-Ka), (|x|-Ka)^2
  int below;
};

// statistics of one sample: selects instead of branches (the loop is issue-bound, and a divergent branch costs
// more scalar bookkeeping than the handful of conditional moves it would skip).  First occurrence wins inside a
// lane because the lane meets its samples in increasing index order.  vmax / vmin / hb start at -inf / +inf / -inf.
template <bool FULL> __device__ __forceinline__ void visit_max(Stream& t, double x, int i, bool valid) {
  const bool up = (FULL || valid) && x > t.vmax;
  t.imax = up ? i : t.imax;
  t.vmax = up ? x : t.vmax;
}
template <bool FULL> __device__ __forceinline__ void visit(Stream& t, double x, bool valid, double k0, double ka) {
  t.vmin = (FULL || valid) ? fmin(t.vmin, x) : t.vmin;
  const double d = (FULL || valid) ? x - k0 : 0.0, e = (FULL || valid) ? fabs(x) - ka : 0.0;
  t.s1 += d;
  t.s2 = __builtin_fma(d, d, t.s2);
  t.a1 += e;
  t.a2 = __builtin_fma(e, e, t.a2);
}

// x at index i with neighbours l (i-1) and r (i+1): the highest local maximum so far (scipy's floor-midpoint of a
// plateau; inside a lane a later peak of equal height wins, as in the priority order).  `ok` carries the lane
// mask (the outer element of lane 0 / 63 belongs to the edge pass) and, outside full tiles, the range 1 <= i <= n-2.
// Returns true when the sample starts a plateau (r == x), which the caller resolves from memory in a rarely
// taken branch.
__device__ __forceinline__ bool peak_fast(Stream& t, int i, bool ok, double l, double x, double r) {
  const bool cand = ok && l < x && x >= t.hb;
  const bool pk = cand && r < x;
  t.hb = pk ? x : t.hb;
  t.mb = pk ? i : t.mb;
  return cand && r == x;
}

// plateau that starts at absolute index i: find its right edge in memory; indices are recorded relative to `lane_off`
__device__ __forceinline__ void peak_plateau(Stream& t, const double* c, int n, int i, double x, int lane_off) {
  int q = i + 1;
  while (q < n - 1 && c[q] == x) ++q;
  if (!(c[q] < x)) return;
  const int m = (i + q - 1) / 2 - lane_off;
  if (t.mb < 0 || higher(x, m, t.hb, t.mb)) { t.hb = x; t.mb = m; }
}

// edge-pass form: neighbours read from memory
__device__ __forceinline__ void peak_test(Stream& t, const double* c, int n, int i, double l, double x, double r, int lane_off) {
  if (peak_fast(t, i - lane_off, i >= 1 && i <= n - 2, l, x, r)) peak_plateau(t, c, n, i, x, lane_off);
}

// ------------------------------------------------------------------ 1. pivots
// a block sample of the row gives the shifts of the one-pass variances and the pivots that bracket the median.
// RUNS values per lane, `ns` of them real in the whole workgroup (the others are zero and `have` of this lane's
// values are real, the first ones).
template <int RUNS> __device__ __forceinline__ void pivot_search(const PeakArgs& a, Shared& s, int tid, int row, const double* sv,
                                                                  const bool* real, int ns, double vfloor = -INFINITY,
                                                                  double pfloor = -INFINITY) {
  const int n = a.n;
  double ssum = 0, sabs = 0;
#pragma unroll
  for (int q = 0; q < RUNS; ++q) {
    ssum += sv[q];                                             // (values that are not real are zero)
    sabs += fabs(sv[q]);
  }
  double both[2] = {ssum, sabs};
  bsum_many<2, kNW>(both, s.many, tid);
  const double k0 = both[0] / double(ns);                     // ~ mean(x)
  const double ka = both[1] / double(ns);                     // ~ mean(|x|)
  double lo = 0, hi = INFINITY;
  if (a.method == 0) {
    const unsigned r1 = unsigned((n - 1) / 2), r2 = unsigned(n / 2);   // ranks of the median's one or two order statistics
    for (int k = tid; k < kBins; k += kT) s.hist[k] = 0;
    __syncthreads();
    const double top = ka > 0 ? 4.0 * ka : 1.0;               // median <= 2 mean for non-negative data
    const double inv = double(kBins - 1) / top;
#pragma unroll
    for (int q = 0; q < RUNS; ++q) {
      if (real[q]) {
        int b = int(fabs(sv[q]) * inv);
        atomicAdd(&s.hist[b < kBins - 1 ? b : kBins - 1], 1u);
      }
    }
    __syncthreads();
    // 6 sigma of an independent sample median's rank.  The block sample is not independent (neighbouring lags of a
    // PHAT sequence are correlated): at 4 sigma 10 of 40320 metric rows missed and paid the radix select, whose
    // row alone then takes longer than the whole launch.
    const int margin = int(3.0 * sqrt(double(ns))) + 8;
    const long long c1 = (long long)r1 * ns / n, c2 = (long long)r2 * ns / n;
    const unsigned slo = unsigned(c1 - margin > 0 ? c1 - margin : 0);
    const unsigned shi = unsigned(c2 + margin < ns - 1 ? c2 + margin : ns - 1);
    unsigned b_lo, b_hi;
    find_two_bins(s, tid, slo, shi, b_lo, b_hi);
    lo = double(b_lo) / inv;
    hi = b_hi >= unsigned(kBins - 1) ? INFINITY : double(b_hi + 1) / inv;
    if (slo == 0) lo = 0;
    if (shi == unsigned(ns - 1)) hi = INFINITY;
  }
  if (tid == 0) {
    RowPre pre;
    pre.k0 = k0; pre.ka = ka; pre.lo = lo; pre.hi = hi;
    pre.vfloor = vfloor; pre.pfloor = pfloor;
    a.pre[row] = pre;
    a.gcount[row] = 0;      // the row's bracket list starts empty (the launch behind this one fills it)
  }
}

// sample = 16 coalesced runs of 512 values spread over the row
__global__ __launch_bounds__(kT) void k_peak_pivots(PeakArgs a) {
  __shared__ Shared s;
  const int tid = threadIdx.x;
  const int row = blockIdx.x;
  const double* c = a.corr + size_t(row) * a.stride;
  const int n = a.n;
  constexpr int kRuns = kSample / kT;
  const int ns = n < kSample ? n : kSample;
  double sv[kRuns];
  bool real[kRuns];
#pragma unroll
  for (int q = 0; q < kRuns; ++q) {
    const int si = tid + q * kT;
    const size_t at = n <= kSample ? size_t(si) : size_t((long long)q * (n - kT) / (kRuns - 1)) + tid;
    real[q] = si < ns;
    sv[q] = real[q] ? c[at] : 0.0;
  }
  // Lower bounds for the stream launch: the sample's maximum, and its highest strict peak (inside a run the lanes
  // tid -/+ 1 of the same wavefront hold the neighbours).  A sample below them can be neither the row's maximum nor its
  // highest local maximum, which lets the stream keep the index bookkeeping and the neighbour tests out of its loop.
  const int lane = tid & 63;
  double vm = -INFINITY, pm = -INFINITY;
#pragma unroll
  for (int q = 0; q < kRuns; ++q) {
    const double x = sv[q];
    const double left = from_lower_lane(x), right = from_upper_lane(x);
    vm = real[q] ? fmax(vm, x) : vm;
    const bool inner = real[q] && lane >= 1 && lane <= 62 && tid + 1 + q * kT < ns;
    pm = inner && left < x && right < x ? fmax(pm, x) : pm;
  }
  vm = block_max<kNW>(vm, s.red_d, tid);
  __syncthreads();
  pm = block_max<kNW>(pm, s.red_d, tid);
  __syncthreads();
  pivot_search<kRuns>(a, s, tid, row, sv, real, ns, vm, pm);
}

// ------------------------------------------------------------------ 2. stream
__global__ __launch_bounds__(kTS) void k_peak_stream(PeakArgs a) {
  __shared__ StreamShared s;
  const int tid = threadIdx.x, lane = tid & 63;
  const int S = a.splits;
  const int row = blockIdx.x / S, seg = blockIdx.x % S;
  const double* c = a.corr + size_t(row) * a.stride;
  const int n = a.n;
  const bool want_median = a.method == 0;
  const RowPre pre = load_pre(a.pre, row);
  const double k0 = pre.k0, ka = pre.ka, lo = pre.lo, hi = pre.hi;
  const double floor = fmin(pre.vfloor, pre.pfloor);           // (-inf when the sample held no strict peak: every tile is examined)
  if (tid == 0) s.count = 0;
  __syncthreads();

  // ---- branch-free tiles of kUnroll 16-byte loads per lane, then a short guarded tail ----
  Stream t;
  t.vmax = t.hb = -INFINITY;
  t.vmin = INFINITY;
  t.imax = t.imin = t.mb = -1;
  t.s1 = t.s2 = t.a1 = t.a2 = 0;
  t.below = 0;
  const bool aligned = (reinterpret_cast<size_t>(c) & 15) == 0;
  const int npair = (n + 1) / 2;
  // this workgroup's share: a whole number of tiles, so segment borders fall on multiples of 128 elements
  const int per_seg = a.tiles_per_seg * kTile;
  const int p_lo = seg * per_seg < npair ? seg * per_seg : npair;
  const int p_hi = p_lo + per_seg < npair ? p_lo + per_seg : npair;
  const int both = p_hi < (n - 1) / 2 ? p_hi : (n - 1) / 2;    // pairs below `both`: both elements valid and at most n - 2
  const int full = p_lo + (both > p_lo ? (both - p_lo) / kTile * kTile : 0);   // end of the branch-free tiles
  // Element e = 2p (+1) of pair p; lane = p % 64.  Neighbours come from the adjacent lanes; the first element
  // of lane 0 and the second of lane 63 (e % 128 == 0 / 127) are peak-tested by the edge pass below instead.
  // One tile = kUnroll element pairs per lane.  Statistics and peak tests per element; the bracket values of the
  // whole tile are appended with ONE LDS atomic per wavefront: per element a ballot gives the lane's rank
  // (mbcnt) and the wavefront's count (scalar popcount), the running scalar total is the tile's reservation.
  const bool not0 = lane != 0, not63 = lane != 63;
  // the edge elements e = 128 q and e = 128 q + 127 of this segment (at most two per lane): their values are requested
  // now and looked at behind the stream; the neighbours are only read for the few that reach the bounds
  static_assert(kMaxTilesPerSeg * kTile * 2 / 64 <= 2 * kTS, "two edge elements per lane");
  int edge_at[2];
  double edge_x[2];
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    const int j = tid + r * kTS;
    const int e = 2 * p_lo + (j >> 1) * 128 + ((j & 1) ? 127 : 0);
    const bool ok = 2 * p_lo + 64 * j < 2 * p_hi && e >= 1 && e <= n - 2 && e < 2 * p_hi;
    edge_at[r] = ok ? e : -1;
    edge_x[r] = ok ? c[e] : -INFINITY;
  }
  auto consume_tile = [&](const double* xa, const double* xb, int pair0u, auto full_tag) {
    constexpr bool FULL = decltype(full_tag)::value;           // full tiles: every element valid, none at index 0 or n - 1
    const int pair0 = pair0u + tid;                            // pair0u: the tile's first pair (wave-uniform)
    int off[2 * kUnroll];
    int run = 0;
    unsigned long long reach = 0;                              // lanes with a sample at or above the pivot launch's bounds
#pragma unroll
    for (int k = 0; k < kUnroll; ++k) {
      const int e0 = 2 * (pair0 + k * kTS);
      const bool va = FULL || e0 < n, vb = FULL || e0 + 1 < n;
      visit<FULL>(t, xa[k], va, k0, ka);
      visit<FULL>(t, xb[k], vb, k0, ka);
      reach |= __ballot((va && xa[k] >= floor) || (vb && xb[k] >= floor));
      if (want_median) {
        const double ma = fabs(xa[k]), mb_ = fabs(xb[k]);
        t.below += int(va && ma < lo) + int(vb && mb_ < lo);
        const bool ha = va && ma >= lo && ma <= hi, hb_ = vb && mb_ >= lo && mb_ <= hi;
        const unsigned long long m0 = __ballot(ha), m1 = __ballot(hb_);
        // slot inside the tile's reservation, or the dump slot kLoc for values outside the bracket: the LDS store
        // below is unconditional (a guarded store costs a divergent branch per sample)
        const int ra = run + int(__builtin_amdgcn_mbcnt_hi(unsigned(m0 >> 32), __builtin_amdgcn_mbcnt_lo(unsigned(m0), 0u)));
        run += __popcll(m0);
        const int rb = run + int(__builtin_amdgcn_mbcnt_hi(unsigned(m1 >> 32), __builtin_amdgcn_mbcnt_lo(unsigned(m1), 0u)));
        run += __popcll(m1);
        off[2 * k] = ha ? ra : kLoc;
        off[2 * k + 1] = hb_ ? rb : kLoc;
      }
    }
    if (reach) {
      // about one tile in ten holds a sample that can be the row's maximum or its highest local maximum (none below
      // the bounds can): only there the index bookkeeping and the neighbour tests run
      bool plateau = false;
#pragma unroll
      for (int k = 0; k < kUnroll; ++k) {
        const int e0 = 2 * (pair0 + k * kTS);
        // indices are recorded relative to the lane's own offset 2 tid: the recorded value is wave-uniform (a scalar
        // operand of the select instead of a vector add per sample); the lane offset is added back after the loop
        const int u0 = 2 * (pair0u + k * kTS);
        const bool va = FULL || e0 < n, vb = FULL || e0 + 1 < n;
        const double left = from_lower_lane(xb[k]);
        const double right = from_upper_lane(xa[k]);
        visit_max<FULL>(t, xa[k], u0, va);
        visit_max<FULL>(t, xb[k], u0 + 1, vb);
        plateau |= peak_fast(t, u0, FULL ? not0 : (not0 && va && e0 >= 1 && e0 <= n - 2), left, xa[k], xb[k]);
        plateau |= peak_fast(t, u0 + 1, FULL ? not63 : (not63 && vb && e0 + 1 <= n - 2), xa[k], xb[k], right);
      }
      if (__ballot(plateau)) {                                 // equal neighbours somewhere in the tile: the exact, slow test
#pragma unroll
        for (int k = 0; k < kUnroll; ++k) {
          const int e0 = 2 * (pair0 + k * kTS);
          const bool va = FULL || e0 < n, vb = FULL || e0 + 1 < n;
          const double left = from_lower_lane(xb[k]);
          if (va && lane != 0 && e0 >= 1 && e0 <= n - 2 && left < xa[k] && xb[k] == xa[k]) peak_plateau(t, c, n, e0, xa[k], 2 * tid);
          const double right = from_upper_lane(xa[k]);
          if (vb && lane != 63 && e0 + 1 <= n - 2 && xa[k] < xb[k] && right == xb[k]) peak_plateau(t, c, n, e0 + 1, xb[k], 2 * tid);
        }
      }
    }
    if (want_median && run > 0) {                              // wavefront-uniform
      int base = 0;
      if (lane == 0) base = atomicAdd(&s.count, run);
      base = __builtin_amdgcn_readfirstlane(base);
#pragma unroll
      for (int k = 0; k < kUnroll; ++k) {
        const int ia = off[2 * k] < kLoc ? min(base + off[2 * k], kLoc) : kLoc;
        const int ib = off[2 * k + 1] < kLoc ? min(base + off[2 * k + 1], kLoc) : kLoc;
        s.list[ia] = fabs(xa[k]);
        s.list[ib] = fabs(xb[k]);
      }
    }
  };
  // register double buffer: the next tile's loads are in flight while this one is consumed
  if (aligned) {
    double2 cur[kUnroll], nxt[kUnroll];
    if (p_lo < full) {
#pragma unroll
      for (int k = 0; k < kUnroll; ++k) cur[k] = *reinterpret_cast<const double2*>(c + 2 * (p_lo + k * kTS + tid));
    }
    for (int base = p_lo; base < full; base += kTile) {
      const int ahead = base + kTile < full ? base + kTile : base;    // last round re-reads its own tile: no branch
#pragma unroll
      for (int k = 0; k < kUnroll; ++k) nxt[k] = *reinterpret_cast<const double2*>(c + 2 * (ahead + k * kTS + tid));
      double xa[kUnroll], xb[kUnroll];
#pragma unroll
      for (int k = 0; k < kUnroll; ++k) { xa[k] = cur[k].x; xb[k] = cur[k].y; }
      consume_tile(xa, xb, base, std::true_type{});
#pragma unroll
      for (int k = 0; k < kUnroll; ++k) cur[k] = nxt[k];
    }
  } else {
    for (int base = p_lo; base < full; base += kTile) {
      double xa[kUnroll], xb[kUnroll];
#pragma unroll
      for (int k = 0; k < kUnroll; ++k) {
        xa[k] = c[2 * (base + k * kTS + tid)];
        xb[k] = c[2 * (base + k * kTS + tid) + 1];
      }
      consume_tile(xa, xb, base, std::true_type{});
    }
  }
  if (full < p_hi) {                                           // tail: one guarded tile
    double xa[kUnroll], xb[kUnroll];
#pragma unroll
    for (int k = 0; k < kUnroll; ++k) {
      const int e0 = 2 * (full + k * kTS + tid);
      xa[k] = e0 < n ? c[e0] : 0.0;
      xb[k] = e0 + 1 < n ? c[e0 + 1] : 0.0;
    }
    consume_tile(xa, xb, full, std::false_type{});
  }
#pragma unroll
  for (int r = 0; r < 2; ++r) {                                // edge pass
    const int e = edge_at[r];
    if (e >= 0 && edge_x[r] >= floor) peak_test(t, c, n, e, c[e - 1], edge_x[r], c[e + 1], 2 * tid);
  }
  // the list's reservation in the row's global list: requested now, used behind the reductions
  __syncthreads();
  const int cnt = s.count;
  if (want_median && tid == 0) s.bc_i = atomicAdd(&a.gcount[row], cnt <= kLoc ? cnt : kList + 1);   // LDS overflow poisons the list
  // ---- one reduction for everything: wavefront shuffles, one LDS hop, lane 0 merges the four wavefronts ----
  double vmax = t.vmax, vmin = t.vmin, hb = t.hb;
  int imax = t.imax < 0 ? -1 : t.imax + 2 * tid, mb = t.mb < 0 ? -1 : t.mb + 2 * tid;
  double sums[5] = {t.s1, t.s2, t.a1, t.a2, double(t.below)};   // (counts < 2^31 are exact in fp64)
  for (int o = 32; o > 0; o >>= 1) {
    const double ov = __shfl_down(vmax, o, 64);
    const int oi = __shfl_down(imax, o, 64);
    if (oi >= 0 && (imax < 0 || arg_better<0>(ov, oi, vmax, imax))) { vmax = ov; imax = oi; }
    const double hv = __shfl_down(hb, o, 64);
    const int hi_ = __shfl_down(mb, o, 64);
    if (hi_ >= 0 && (mb < 0 || arg_better<2>(hv, hi_, hb, mb))) { hb = hv; mb = hi_; }
    vmin = fmin(vmin, __shfl_down(vmin, o, 64));
#pragma unroll
    for (int q = 0; q < 5; ++q) sums[q] += __shfl_down(sums[q], o, 64);
  }
  if (lane == 0) {
    StreamWave& w = s.wave[tid >> 6];
    w.vmax = vmax; w.vmin = vmin; w.hb = hb; w.imax = imax; w.mb = mb;
#pragma unroll
    for (int q = 0; q < 5; ++q) w.sums[q] = sums[q];
  }
  __syncthreads();
  if (tid == 0) {
    for (int k = 1; k < kNWS; ++k) {
      const StreamWave& w = s.wave[k];
      if (w.imax >= 0 && (imax < 0 || arg_better<0>(w.vmax, w.imax, vmax, imax))) { vmax = w.vmax; imax = w.imax; }
      if (w.mb >= 0 && (mb < 0 || arg_better<2>(w.hb, w.mb, hb, mb))) { hb = w.hb; mb = w.mb; }
      vmin = fmin(vmin, w.vmin);
#pragma unroll
      for (int q = 0; q < 5; ++q) sums[q] += w.sums[q];
    }
  }
  const int imin = vmin < INFINITY ? 0 : -1;                   // (the finish launch only asks whether the segment has a minimum)

  // ---- publish: the segment's bracket values join the row's list (one global atomic), its statistics its slot ----
  if (want_median) {
    const int at = s.bc_i;                                     // (written before the reductions' barriers)
    double* dst = a.glist + size_t(row) * kList;
    if (cnt <= kLoc && at >= 0 && at + cnt <= kList)
      for (int k = tid; k < cnt; k += kTS) dst[at + k] = s.list[k];
  }
  if (tid == 0) {
    Partial pt;
    pt.vmax = vmax; pt.vmin = vmin; pt.hb = hb; pt.s1 = sums[0]; pt.s2 = sums[1]; pt.a1 = sums[2]; pt.a2 = sums[3];
    pt.below = (long long)sums[4]; pt.imax = imax; pt.imin = imin; pt.mb = mb; pt.pad = 0;
    pt.plat = -INFINITY;
    pt.pfloor = -INFINITY;                                     // (the row's bounds come from the pivot launch on this path)
    a.parts[size_t(row) * S + seg] = pt;
  }
}

// ------------------------------------------------------------------ 3. finish
template <bool LOCAL>   // LOCAL: the segments are column blocks of the fused column pass (histogram windows, no pivots)
__device__ __forceinline__ void finish_row(const PeakArgs& a, pal_pair_record* table, int32_t* ksel_multi, int* status, const int row) {
  __shared__ Shared s;
  const int tid = threadIdx.x;
  const int S = a.splits;
  const double* c = a.corr + size_t(row) * a.stride;
  const int n = a.n;
  const bool want_median = a.method == 0;
  constexpr bool local = LOCAL;                               // fused column pass: zero shifts, no row parameters
  RowPre pre;
  if (local) { pre.k0 = pre.ka = 0; pre.lo = 0; pre.hi = INFINITY; pre.vfloor = pre.pfloor = -INFINITY; }
  else pre = load_pre(a.pre, row);
  const double k0 = pre.k0, ka = pre.ka, lo = pre.lo, hi = pre.hi;
  const unsigned r1 = unsigned((n - 1) / 2), r2 = unsigned(n / 2);   // ranks of the median's one or two order statistics
  int stamp_at = 0;
  auto stamp = [&]() {
    if (a.stamps && tid == 0) a.stamps[size_t(row) * 8 + stamp_at] = __builtin_amdgcn_s_memrealtime();
    ++stamp_at;
  };
  stamp();

  // ---- merge the segments (every lane the same loop: no broadcast needed) ----
  int imax = -1, imin = -1, mb = -1;
  double vmax = 0, vmin = 0, hb = 0, s1 = 0, s2 = 0, a1 = 0, a2 = 0;
  long long below = 0;
  double plat = -INFINITY;
  // (the segments' results are staged through LDS in one round of loads: a loop of global loads over sixteen column
  //  blocks costs sixteen dependent round trips in this latency-bound launch)
  __shared__ Partial sparts[kMaxStaged];
  const bool staged = S <= kMaxStaged;
  if (staged) {
    const double* src = reinterpret_cast<const double*>(a.parts + size_t(row) * S);
    double* dst = reinterpret_cast<double*>(sparts);
    for (int i = tid; i < S * int(sizeof(Partial) / sizeof(double)); i += kT) dst[i] = src[i];
    __syncthreads();
  }
  if (local && staged && S > 16) {
    // many column blocks (row lengths N2 up to 8192: 133 of them): one segment per lane, workgroup reductions
    Partial pt;
    pt.imax = pt.imin = pt.mb = -1;
    pt.vmax = pt.hb = 0; pt.vmin = INFINITY; pt.plat = pt.pfloor = -INFINITY;
    pt.s1 = pt.s2 = pt.a1 = pt.a2 = 0; pt.below = 0;
    if (tid < S) pt = sparts[tid];
    vmax = pt.vmax; imax = pt.imax;
    barg<0>(vmax, imax, s, tid);
    __syncthreads();
    hb = pt.hb; mb = pt.mb;
    barg<2>(hb, mb, s, tid);
    __syncthreads();
    vmin = -block_max<kNW>(tid < S && pt.imin >= 0 ? -pt.vmin : -INFINITY, s.red_d, tid);
    imin = vmin < INFINITY ? 0 : -1;
    __syncthreads();
    plat = block_max<kNW>(pt.plat, s.red_d, tid);
    __syncthreads();
    pre.pfloor = block_max<kNW>(pt.pfloor, s.red_d, tid);
    __syncthreads();
    double sums[4] = {pt.s1, pt.s2, pt.a1, pt.a2};
    bsum_many<4, kNW>(sums, s.many, tid);
    s1 = sums[0]; s2 = sums[1]; a1 = sums[2]; a2 = sums[3];
    __syncthreads();
  } else
  for (int q = 0; q < S; ++q) {
    const Partial pt = staged ? sparts[q] : a.parts[size_t(row) * S + q];
    plat = fmax(plat, pt.plat);
    if (pt.imax >= 0 && (imax < 0 || arg_better<0>(pt.vmax, pt.imax, vmax, imax))) { vmax = pt.vmax; imax = pt.imax; }
    if (pt.imin >= 0 && (imin < 0 || arg_better<1>(pt.vmin, pt.imin, vmin, imin))) { vmin = pt.vmin; imin = pt.imin; }
    if (pt.mb >= 0 && (mb < 0 || higher(pt.hb, pt.mb, hb, mb))) { hb = pt.hb; mb = pt.mb; }
    s1 += pt.s1; s2 += pt.s2; a1 += pt.a1; a2 += pt.a2;
    below += pt.below;
    if (local) pre.pfloor = fmax(pre.pfloor, pt.pfloor);       // highest bound under which a segment left samples untested
  }
  stamp();
  if (imax < 0) {                                              // no sample reached the pivot launch's bound for the maximum (insurance): scan
    double bv = 0;
    int bi = -1;
    for (int i = tid; i < n; i += kT) {
      const double x = c[i];
      if (x == x && (bi < 0 || x > bv)) { bv = x; bi = i; }
    }
    barg<0>(bv, bi, s, tid);
    if (bi >= 0) { vmax = bv; imax = bi; }
  }
  if (imax < 0 || imax >= n) imax = 0;                         // all-NaN row (and a guard for every index used below)
  if (mb >= n) mb = -1;
  {
    // Strict peaks and plateau STARTS, tested from memory (a start walks to its plateau's end once).
    double tie = -INFINITY;                                    // highest tested sample with an equal neighbour
    auto test = [&](int m, double& bh, int& bm) {
      if (m < 1 || m > n - 2) return;
      const double xl = c[m - 1], x = c[m], xr = c[m + 1];
      if (xl == x || xr == x) tie = fmax(tie, x);
      int at = -1;
      if (xl < x && xr < x) at = m;
      else if (xl < x && xr == x) {
        int q = m + 1;
        while (q < n - 1 && c[q] == x) ++q;
        if (c[q] < x) at = (m + q - 1) / 2;
      }
      if (at >= 0 && (bm < 0 || higher(x, at, bh, bm))) { bh = x; bm = at; }
    };
    double eh = 0;
    int em = -1;
    if (a.edge_n2 > 0) {
      // The segments were column blocks of the prime-factor grid (pfa_cols_stats.h): the samples of the grid's first
      // and last column (neighbours in another output row) had no peak test there, and samples with an equal neighbour
      // were only reported (`plat`; a plateau start that is not in an edge column has been reported).
      const int N2 = a.edge_n2, N1 = n / N2;
      for (int k = tid; k < 2 * N1; k += kT) test(k < N1 ? N2 * k : N2 * (k - N1) + N2 - 1, eh, em);
      barg<2>(eh, em, s, tid);
      if (em >= 0 && (mb < 0 || higher(eh, em, hb, mb))) { hb = eh; mb = em; }
      plat = fmax(plat, block_max<kNW>(tie, s.red_d, tid));
    }
    if constexpr (local) {
      // A column block tested only its samples at or above 0.8 x ITS maximum.  Where that maximum is no peak at all (an
      // end point of the row, or a sample beside a higher one in the next block) and the row's best peak so far is
      // lower than the block's bound, the block's untested samples may hold a higher peak: test that block again, all of
      // it (62 columns x N1 output indices; about one row in a hundred - e.g. the zero-lag maximum of two microphones
      // with equal delays sits at index 0, which is never a peak).
      const int N2 = a.edge_n2, N1 = N2 > 0 ? n / N2 : 0;
      for (int q = 0; q < S && N2 > 0; ++q) {
        const double pf = staged ? sparts[q].pfloor : a.parts[size_t(row) * S + q].pfloor;
        if (!(pf > -INFINITY) || (mb >= 0 && hb >= pf)) continue;          // (uniform)
        const int c0 = int((long long)q * N2 / S), cols = int((long long)(q + 1) * N2 / S) - c0;   // (the columns are dealt evenly)
        eh = 0;
        em = -1;
        for (int k = tid; k < cols * N1; k += kT) test(c0 + k % cols + N2 * (k / cols), eh, em);
        barg<2>(eh, em, s, tid);
        if (em >= 0 && (mb < 0 || higher(eh, em, hb, mb))) { hb = eh; mb = em; }
        plat = fmax(plat, block_max<kNW>(tie, s.red_d, tid));
      }
    }
    // Rescan the row when a reported plateau may outrank the best strict peak, or (separate launches) when the best peak
    // ends up below the pivot launch's bound (the segments only tested samples above it: then it was not a bound -
    // cannot happen while the bound is a value of the row itself, kept as insurance)
    if ((plat > -INFINITY && (mb < 0 || plat >= hb)) || (!local && pre.pfloor > -INFINITY && (mb < 0 || hb < pre.pfloor))) {
      eh = 0;
      em = -1;
      for (int m = 1 + tid; m <= n - 2; m += kT) test(m, eh, em);
      barg<2>(eh, em, s, tid);
      hb = eh;
      mb = em;
    }
  }

  // ---- SNR (utils.py:238-250): totals minus the window around the maximum ----
  const int wlo_s = imax - a.snr_w > 0 ? imax - a.snr_w : 0;
  const int whi_s = imax + a.snr_w < n ? imax + a.snr_w : n;
  double w1 = 0, w2 = 0;
  for (int i = wlo_s + tid; i < whi_s; i += kT) {
    const double d = c[i] - k0;
    w1 += d;
    w2 += d * d;
  }
  double wsum[2] = {w1, w2};
  bsum_many<2, kNW>(wsum, s.many, tid);
  w1 = wsum[0];
  w2 = wsum[1];
  const double nn = double(n - (whi_s - wlo_s));
  double o1 = s1 - w1, o2 = s2 - w2;
  if (!(o2 >= 0.25 * s2)) {
    // the window holds most of the row's energy (strongly correlated signals: a near-delta sequence), so
    // "total minus window" would cancel: sum the noise region itself in one more pass
    // (two passes like np.std: the sample mean k0 may sit 1e3 noise sigmas away when it caught the peak)
    double q1 = 0, q2 = 0;
    for (int i = tid; i < n; i += kT)
      if (i < wlo_s || i >= whi_s) q1 += c[i];
    const double mu = bsum(q1, s, tid) / nn;
    for (int i = tid; i < n; i += kT)
      if (i < wlo_s || i >= whi_s) { const double d = c[i] - mu; q2 += d * d; }
    o1 = 0;
    o2 = bsum(q2, s, tid);
  }
  double var = (o2 - o1 * o1 / nn) / nn;
  if (var < 0) var = 0;
  const double noise = sqrt(var);
  const double snr = noise == 0.0 ? INFINITY : vmax / noise;
  stamp();

  if (a.method < 0) {                                          // metrics only (pal_corr_metrics)
    if (tid == 0) {
      pal_pair_record r;
      r.k_sel = imax; r.branch = 0; r.k_argmax = imax; r.n_sel = 0;
      r.cmax = vmax; r.cmin = vmin; r.snr = snr; r.sel_height = vmax;
      table[row] = r;
    }
    return;
  }

  // ---- primary threshold (utils.py:144-149): exact, or (fused column pass) an interval [tlo, thi] that holds it ----
  double tlo, thi;
  unsigned rin1 = 0, rin2 = 0;                                 // ranks of the median's order statistics inside their bins
  double b1lo = 0, b1hi = 0, b2lo = 0, b2hi = 0;               // those bins
  bool have_bins = false;
  if (want_median) {
    if constexpr (local) {
      // Every segment published the counts of 48 histogram bins around its own median and the count below them.  Where
      // the windows overlap the sums are the row's exact counts: the bin that holds rank r is a rigorous interval for
      // that order statistic (relative width 2^(1/128) - 1 = 0.5 %).
      const BlockHist* bh = a.bh + size_t(row) * S;
      // headers first (one lane per segment), then every (segment, bin) entry by its own lane: two rounds of loads
      __shared__ int swin0[kMaxStaged];
      if (tid < 2 * kWin) s.hist[tid] = 0;                     // [0, kWin): merged bins of the common window; [kWin]: count below it; [kWin + 1]: total
      if (tid < S && tid < kMaxStaged) swin0[tid] = bh[tid].win0;
      __syncthreads();
      int w0 = 0, w1 = kLogBins;
      for (int q = 0; q < S && q < kMaxStaged; ++q) {
        const int v = swin0[q];
        w0 = v > w0 ? v : w0;
        w1 = v + kWin < w1 ? v + kWin : w1;
      }
      if (S > kMaxStaged) w1 = w0;                             // (guard: more segments than staged - the exact select decides)
      if (tid < S && tid < kMaxStaged) {
        atomicAdd(&s.hist[kWin], bh[tid].below);
        atomicAdd(&s.hist[kWin + 1], bh[tid].total);
      }
      for (int e = tid; e < S * kWin && S <= kMaxStaged; e += kT) {
        const int q = e / kWin, k = e - q * kWin;
        const int b = swin0[q] + k;
        const unsigned v = bh[q].h[k];
        if (b < w0) atomicAdd(&s.hist[kWin], v);
        else if (b < w1) atomicAdd(&s.hist[b - w0], v);
      }
      __syncthreads();
      const unsigned total = s.hist[kWin + 1];
      const unsigned base = s.hist[kWin];
      have_bins = w0 < w1 && total == unsigned(n) && r1 >= base;
      if (have_bins) {
        unsigned e = base;
        int f1 = -1, f2 = -1;
        for (int k = 0; k < w1 - w0; ++k) {
          const unsigned v = s.hist[k];
          if (f1 < 0 && r1 < e + v) { f1 = k; rin1 = r1 - e; }
          if (f2 < 0 && r2 < e + v) { f2 = k; rin2 = r2 - e; }
          e += v;
        }
        have_bins = f1 >= 0 && f2 >= 0;
        if (have_bins) {
          b1lo = log_bin_floor(w0 + f1); b1hi = log_bin_floor(w0 + f1 + 1);
          b2lo = log_bin_floor(w0 + f2); b2hi = log_bin_floor(w0 + f2 + 1);
        }
      }
      __syncthreads();
    }
    if (have_bins) {                                           // np.median = the middle value, or the mean of the two middle ones
      const double ml = r2 != r1 ? (b1lo + b2lo) * 0.5 : b1lo, mh = r2 != r1 ? (b1hi + b2hi) * 0.5 : b1hi;
      tlo = a.mult >= 0 ? a.mult * ml : a.mult * mh;
      thi = a.mult >= 0 ? a.mult * mh : a.mult * ml;
    }
  }
  stamp();
  const bool windowed = !isnan(a.med);
  int wlo = 1, whi = n - 2;
  if (windowed) {
    const double span = a.med * a.fs;
    const double c0 = double(a.n2 - 1);
    const double flo = c0 - span - 2.0, fhi = c0 + span + 2.0;
    wlo = flo > 1.0 ? (flo < double(n) ? int(flo) : n) : 1;
    whi = fhi < double(n - 2) ? (fhi > -1.0 ? int(fhi) : -1) : n - 2;
  }
  const SelArgs sa{n, a.n2, a.dist, a.num_peaks, a.fs, a.med, a.bits ? a.bits + size_t(row) * 2 * ((n + 31) / 32) : nullptr, a.memo_cap, a.stack_cap};
  // np.mean(np.abs(corr)) (utils.py:155): the alternative threshold of the fallback chain.  The fused column pass sums
  // |x| only for the 'adaptive' method; the (rare) fallback branches sum it from the stored row.
  double mean_abs = ka + a1 / double(n);
  bool have_mean_abs = !(local && want_median);
  int branch = 0, count = 0;
  bool overflow = false, argmax_fallback = false;
  // The fallback chain (utils.py:152-179) runs on the interval [tlo, thi] first; a comparison inside it (`ambiguous`)
  // sends the row through the loop a second time with the exact primary threshold.
  bool thr_exact = !(local && want_median && have_bins);
  for (int attempt = 0; attempt < 2; ++attempt) {
    if (attempt == 1 || thr_exact) {
      if (!thr_exact || attempt == 0) {
        // the exact primary threshold: mean + std of |corr|, or the median from the old launches' bracket list, or -
        // fused column pass - one pass over the stored row for the bin that holds it
        if (!want_median) {
          double va = (a2 - a1 * a1 / double(n)) / double(n);
          if (va < 0) va = 0;
          tlo = thi = a.mult * ((ka + a1 / double(n)) + sqrt(va));      // utils.py:147
        } else {
          double m0 = 0, m1 = 0;
          bool ok = false;
          if constexpr (local) {
            if (have_bins) {
              ok = interval_select(s, tid, c, n, b1lo, b1hi, rin1, m0);
              if (ok) { m1 = m0; if (r2 != r1) ok = interval_select(s, tid, c, n, b2lo, b2hi, rin2, m1); }
            }
            if (tid == 0) atomicAdd(status + (ok ? 3 : 1), 1);       // diagnostics: rows that needed the exact median / the radix select
          } else {
            const int cnt = a.gcount[row];
            const double* list = a.glist + size_t(row) * kList;
            ok = cnt >= 0 && cnt <= kList && (long long)r1 >= below && (long long)r2 < below + cnt;
            if (ok) ok = list_select(s, tid, list, cnt, unsigned(r1 - below), lo, hi, m0);
            if (ok) { m1 = m0; if (r2 != r1) ok = list_select(s, tid, list, cnt, unsigned(r2 - below), lo, hi, m1); }
            if (!ok && tid == 0) atomicAdd(status + 1, 1);           // diagnostics: rows that needed the slow exact select
          }
          if (!ok) {                                                 // pivots / windows missed or a list overflowed: exact radix select
            m0 = radix_select(c, n, tid, s, r1);
            m1 = r2 != r1 ? radix_select(c, n, tid, s, r2) : m0;
          }
          tlo = thi = a.mult * (r2 != r1 ? (m0 + m1) * 0.5 : m0);    // np.median
        }
        thr_exact = true;
      }
    }
    branch = 0;
    count = 0;
    overflow = argmax_fallback = false;
    bool alt = false;                                        // the search runs with mean(|corr|) instead of the primary threshold
    bool reach = false;
    if (mb >= 0) {
      if (hb >= thi) reach = true;
      else if (hb >= tlo) continue;                          // inside the interval: exact threshold, second round
    }
    if (!reach) {                                            // no peak reaches the primary threshold
      branch |= PAL_BR_ALT_THRESHOLD;
      alt = true;
    }
    if (alt || windowed) {                                   // (the window retry below may need it: one place for the row pass)
      if (alt && !have_mean_abs) {
        double acc = 0;
        for (int i = tid; i < n; i += kT) acc += fabs(c[i]);
        mean_abs = bsum(acc, s, tid) / double(n);
        __syncthreads();
        have_mean_abs = true;
      }
    }
    if (alt && !(mb >= 0 && hb >= mean_abs)) { branch |= PAL_BR_ARGMAX_NO_PEAKS; argmax_fallback = true; }
    if (!argmax_fallback) {
      const bool first_ok = !windowed;                       // unwindowed: the best peak is already known
      count = select_peaks(sa, c, tid, s, alt ? mean_abs : tlo, alt ? mean_abs : thi, windowed, wlo, whi, first_ok ? hb : 0.0,
                           first_ok ? mb : -1);
      if (count == -2) continue;                             // a candidate inside the interval: exact threshold, second round
      if (count < 0) overflow = true;
      if (count == 0 && windowed) {
        branch |= PAL_BR_WINDOW_RETRY;
        if (!have_mean_abs) {
          double acc = 0;
          for (int i = tid; i < n; i += kT) acc += fabs(c[i]);
          mean_abs = bsum(acc, s, tid) / double(n);
          __syncthreads();
          have_mean_abs = true;
        }
        count = select_peaks(sa, c, tid, s, mean_abs, mean_abs, true, wlo, whi, 0.0, -1);
        if (count < 0) overflow = true;
        if (count == 0) { branch |= PAL_BR_ARGMAX_WINDOW; argmax_fallback = true; }
      }
    }
    break;
  }
  if (argmax_fallback || overflow) {
    if (tid == 0) { s.sel_pos[0] = imax; s.sel_h[0] = vmax; }
    count = 1;
  }
  stamp();

  if (tid == 0) {
    pal_pair_record r;
    r.k_sel = s.sel_pos[0];
    r.branch = branch;
    r.k_argmax = imax;
    r.n_sel = count;
    r.cmax = vmax;
    r.cmin = vmin;
    r.snr = snr;
    r.sel_height = s.sel_h[0];
    table[row] = r;
    if (ksel_multi)
      for (int k = 0; k < PAL_MAX_PEAKS; ++k) ksel_multi[size_t(row) * PAL_MAX_PEAKS + k] = k < count ? s.sel_pos[k] : -1;
    if (overflow) atomicOr(status, 1);
  }
}

template <bool LOCAL>
__global__ __launch_bounds__(kT, 4) void k_peak_finish(PeakArgs a, pal_pair_record* table, int32_t* ksel_multi, int* status) {   // (4 waves per SIMD = two workgroups per CU: the launch is latency-bound)
  finish_row<LOCAL>(a, table, ksel_multi, status, blockIdx.x);
}

}  // namespace

static int* g_status_dev(Engine* e) {
  void* p = nullptr;
  if (e->scratch(7, 64, &p) != PAL_OK) return nullptr;
  return static_cast<int*>(p);
}

int Engine::peaks_setup(const double* corr, size_t stride, int rows, int n, int n2, const pal_phat_params& prm, int blocks, int grid_n2,
                        hipStream_t on, PeakArgs& a) {
  const bool metrics_only = prm.threshold_method < 0;
  if (!metrics_only) {
    if (prm.num_peaks < 1 || prm.num_peaks > PAL_MAX_PEAKS) return fail(PAL_ERR_INVALID, "num_peaks %d outside 1..%d", prm.num_peaks, PAL_MAX_PEAKS);
    if (prm.peak_distance < 1) return fail(PAL_ERR_INVALID, "`distance` must be greater or equal to 1");
  }
  if (n < 1) return fail(PAL_ERR_INVALID, "empty correlation");
  a.corr = corr; a.stride = stride; a.n = n; a.n2 = n2;
  a.fs = prm.fs; a.mult = prm.threshold_multiplier; a.med = prm.max_expected_delay;
  a.method = prm.threshold_method; a.dist = prm.peak_distance; a.num_peaks = prm.num_peaks;
  const int w = int(0.01 * double(n));                       // utils.py:244
  a.snr_w = w > 1 ? w : 1;
  a.edge_n2 = blocks > 0 ? grid_n2 : 0;
  a.local_pivots = blocks > 0 ? 1 : 0;
  a.stamps = nullptr;
  if (blocks > 0) {
    a.splits = blocks;
    a.tiles_per_seg = 0;
  } else {
    // segments: about 1024 workgroups per launch (four resident per CU, all in flight at once), so that a workgroup
    // streams as many tiles as possible behind one set of fixed costs (row parameters, reductions, publish)
    const int ntiles = ((n + 1) / 2 + kTile - 1) / kTile;
    int want = 1024 / rows;                                     // (rounded down: one resident round rather than a second, short one)
    want = want < 1 ? 1 : (want > ntiles ? ntiles : want);
    a.tiles_per_seg = (ntiles + want - 1) / want;
    if (a.tiles_per_seg > kMaxTilesPerSeg) a.tiles_per_seg = kMaxTilesPerSeg;   // the segment's bracket values must fit its LDS list
    a.splits = (ntiles + a.tiles_per_seg - 1) / a.tiles_per_seg;
  }
  // per-stream scratch: [gcount | pre | parts | glist]; the pivot launch zeroes the list fills
  size_t off_pre = (size_t(rows) * sizeof(int) + 127) & ~size_t(127);
  size_t off_parts = (off_pre + size_t(rows) * sizeof(RowPre) + 127) & ~size_t(127);
  size_t off_list = (off_parts + size_t(rows) * a.splits * sizeof(Partial) + 127) & ~size_t(127);
  // (fused column pass: the segments' histogram windows take the place of the bracket lists)
  const size_t list_bytes = a.local_pivots ? size_t(rows) * a.splits * sizeof(BlockHist) : (a.method == 0 ? size_t(rows) * kList * sizeof(double) : 0);
  const size_t off_bits = (off_list + list_bytes + 127) & ~size_t(127);
  const size_t total = off_bits + (corr ? size_t(rows) * 2 * ((size_t(n) + 31) / 32) * sizeof(unsigned) : 0);   // (bitmaps only where rows are stored)
  void* sp = nullptr;
  PAL_TRY(scratch(on == stream2 ? 9 : (on == stream3 ? 12 : 8), total, &sp));
  char* base = static_cast<char*>(sp);
  a.gcount = reinterpret_cast<int*>(base);
  a.pre = reinterpret_cast<RowPre*>(base + off_pre);
  a.parts = reinterpret_cast<Partial*>(base + off_parts);
  a.glist = reinterpret_cast<double*>(base + off_list);
  a.bh = reinterpret_cast<BlockHist*>(base + off_list);
  a.bits = corr ? reinterpret_cast<unsigned*>(base + off_bits) : nullptr;
  {
    const int dbg = debug_memo;                                // (PAL_DEBUG_MEMO, tests: a tiny memo sends every chain to the slow path)
    a.memo_cap = dbg > 0 && dbg < kMemo ? dbg : kMemo;
    a.stack_cap = dbg > 0 && dbg < kStack ? dbg : kStack;
  }
  return PAL_OK;
}

int Engine::peaks_finish(PeakArgs& a, int rows, pal_pair_record* table, int32_t* ksel_multi, hipStream_t on) {
  const bool metrics_only = a.method < 0;
  int* status = g_status_dev(this);
  if (!status) return fail(PAL_ERR_NOMEM, "status word");
  a.stamps = nullptr;
  static const bool want_stamps = getenv("PAL_DEBUG_STAMPS") != nullptr;
  if (want_stamps) {
    void* st = nullptr;
    PAL_TRY(scratch(13, size_t(rows) * 8 * sizeof(unsigned long long), &st));
    a.stamps = static_cast<unsigned long long*>(st);
  }
  {
    ProfScope ps(this, metrics_only ? "k_peak_finish(metrics)" : "k_peak_finish", on);
    if (a.local_pivots) k_peak_finish<true><<<dim3(rows), dim3(kT), 0, on>>>(a, table, ksel_multi, status);
    else k_peak_finish<false><<<dim3(rows), dim3(kT), 0, on>>>(a, table, ksel_multi, status);
    PAL_HIP(hipGetLastError());
  }
  if (want_stamps && !metrics_only) {      // diagnostics: median phase times of this launch (synchronises)
    std::vector<unsigned long long> h(size_t(rows) * 8);
    PAL_HIP(hipStreamSynchronize(on));
    PAL_HIP(hipMemcpy(h.data(), a.stamps, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    const char* names[4] = {"merge", "snr window", "threshold", "select"};
    std::vector<double> d(rows);
    for (int ph = 0; ph < 4; ++ph) {
      for (int r = 0; r < rows; ++r) d[r] = double(h[size_t(r) * 8 + ph + 1] - h[size_t(r) * 8 + ph]) / 100.0;
      std::sort(d.begin(), d.end());
      fprintf(stderr, "[pal] k_peak_finish %-10s median %6.2f us  p90 %6.2f us  max %6.2f us\n", names[ph], d[rows / 2], d[rows * 9 / 10], d[rows - 1]);
    }
  }
  return PAL_OK;
}

int Engine::peaks(const double* corr, size_t stride, int rows, int n, int n2, const pal_phat_params& prm,
                  pal_pair_record* table, int32_t* ksel_multi, hipStream_t on) {
  if (rows <= 0) return PAL_OK;
  PeakArgs a;
  PAL_TRY(peaks_setup(corr, stride, rows, n, n2, prm, 0, 0, on, a));
  {
    ProfScope ps(this, "k_peak_pivots", on);
    k_peak_pivots<<<dim3(rows), dim3(kT), 0, on>>>(a);
    PAL_HIP(hipGetLastError());
  }
  {
    ProfScope ps(this, "k_peak_stream", on);
    k_peak_stream<<<dim3(unsigned(rows) * unsigned(a.splits)), dim3(kTS), 0, on>>>(a);
    PAL_HIP(hipGetLastError());
  }
  return peaks_finish(a, rows, table, ksel_multi, on);
}

}  // namespace pal
