// pfa_cols_stats.h - column pass of the prime-factor route that also does the streaming pass of the peak
// selection (gfx950, fp64): no pivot launch, no second read of the correlation rows.
//
// k_pfa_cols (pfa_kernels.h) writes the correlation rows and k_peak_stream (peaks.hip) reads them back once for
// the statistics the finish launch needs: max / argmax, min, the sums behind SNR and thresholds, the highest local
// maximum, the count below the lower pivot and the values between the pivots that bracket the median of |corr|
// (utils.py:145).  Here the values meet those statistics in the registers they were accumulated in, and the 0.7 MB per
// row are only written (the finish launch still reads the few thousand samples around the peaks it resolves).
//
//   - a workgroup owns up to 62 columns m2 (the grid's columns dealt evenly to the blocks) of one packed transform (both rows: pair p = real parts, pair q = imaginary
//     parts) and all N1 output indices t, its four wavefronts four chunks of kPfaTC indices and their mirrors;
//     lanes 0 and 63 compute the neighbouring blocks' border columns again, so that every owned sample has both
//     neighbours m -/+ 1 = (m2 -/+ 1, t) one lane away (DPP wave shifts)
//   - the median of |corr| (utils.py:145) without pivots, lists or a second pass: pass A puts |x| of the block's ~5500
//     samples per row into a logarithmic histogram in LDS (128 bins per octave over the 16 octaves below 1.0 - a PHAT
//     sequence never exceeds 1; EXACT counts), finds the bin of the block's own median and publishes the 48 bins around
//     it plus the count below them (BlockHist, 208 bytes).  The blocks' medians differ by a few bins, so their windows
//     overlap around the ROW's median: the finish launch adds them up and gets the bin that holds it - a rigorous
//     interval of relative width 0.5 %.  A threshold comparison whose peak height lies outside mult x that interval is
//     decided; the exact median is computed from the stored row only for a comparison inside it (rare: the candidates
//     the selection examines are the highest peaks, the median sits at 0.67 sigma).  Pass A also takes the exact
//     maximum, the minimum and the sums (and sum |x| for the 'adaptive' threshold only).
//   - pass B (same registers): stores; the index bookkeeping of the maximum and of the highest strict peak sits behind
//     a wave-uniform test against the block's exact maximum and 0.8 of it (a sample below them can be neither the
//     block's maximum nor, unless none of the samples above is a strict peak, its highest peak - the finish launch
//     rescans a row whose best peak ends up below a block's bound)
//   - the first and last column of the grid have their neighbours in another output index: their peak test is left
//     to the finish launch (2 N1 samples per row), like the samples with an equal neighbour (plateaus), which are
//     only reported
//   - a lane meets its samples in increasing lag order (t = 0, the chunk ascending, the mirrors descending), so the
//     first maximum / last peak of equal height win without index comparisons, as in the stream kernel
#pragma once
#include <cmath>

#include "peak_types.h"
#include "pfa_kernels.h"
#include "pfa_rader89.h"

namespace pal {

constexpr int kColsOwn = 62;      // columns a workgroup of the fused column pass owns (64 lanes - two border lanes)

struct ColsWaveResult {           // one wavefront's share of a row segment
  double vmax, vmin, hb, s1, s2, a1, plat;
  int imax, mb, pad0, pad1;
};

__device__ __forceinline__ double shfl_down_d(double v, int o) { return __shfl_down(v, o, 64); }

// max / min without the canonicalising v_max_f64 x, x, x the compiler puts in front of every fmax / fmin (the inputs are
// sums of finite products: there is no signalling NaN to quiet)
__device__ __forceinline__ double max_raw(double a, double b) {
  double r;
  asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ double min_raw(double a, double b) {
  double r;
  asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}

// ADAPTIVE: threshold method 'adaptive' (mean + std of |corr|: sum |x| instead of the histogram).
// FULL: every chunk index exists ((N1 - 1) / 2 == nch * TC, e.g. N1 = 89): no per-sample existence tests.
// NW: wavefronts per workgroup (2 or 4).  STRIPS = false: they are the chunks of output indices of ONE 62-column strip.
// STRIPS = true (a short column DFT with a single chunk, N1 <= 23, e.g. 47 999 = 7 x 6857): they are NW neighbouring strips,
// so that the block's histogram and window (fixed costs of a workgroup) are shared by NW x 62 columns instead of being
// paid for 62 x N1 samples.
// R89 (N1 = 89, four wavefronts): the column DFT is Rader's 8 x 11 convolution spread over the wavefronts (pfa_rader89.h)
// instead of the dense form - half the arithmetic, a quarter of the loads; a wavefront then holds the output indices
// tab->tmap[wave][.] (not a chunk in lag order: ties between equal samples are broken by their indices explicitly).
template <int TC, int UNR, bool ADAPTIVE, bool FULL, int NW, bool STRIPS, bool R89 = false>
__global__ __launch_bounds__(64 * NW) void k_pfa_cols_stats(const cd* __restrict__ Y, double* __restrict__ corr, size_t stride, int N1, int N2,
                                                        int G, int nch, const double* __restrict__ T, const int* __restrict__ zero_rows, PeakArgs pa,
                                                        int rows, const Rader89Tab* __restrict__ tab = nullptr) {
  static_assert(!R89 || (NW == 4 && FULL && !STRIPS), "the Rader column transform is the four-wavefront N1 = 89 case");
  // histograms of |x| (+ one dump bin for the lanes that own nothing); with R89 the same memory is first the exchange plane
  // of the column transform (88 x 64 doubles)
  constexpr int kHistDoubles = (2 * (kLogBins + 1) * int(sizeof(unsigned)) + 7) / 8;
  __shared__ __attribute__((aligned(16))) double lds_big[R89 ? 88 * 64 : kHistDoubles];
  unsigned (*const hist)[kLogBins + 1] = reinterpret_cast<unsigned (*)[kLogBins + 1]>(lds_big);
  __shared__ ColsWaveResult res[4][2];
  __shared__ double wmax[4][2];
  __shared__ unsigned wtot[4][2];
  __shared__ int medbin[2];                                   // bin of the block's own median per row
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ch = STRIPS ? 0 : wave;                           // nch <= 4: one workgroup covers every output index
  const int g = blockIdx.x % G, cb = blockIdx.x / G;
  const int strip = STRIPS ? cb * NW + wave : cb;             // column strip of this wavefront
  // The grid's columns are dealt EVENLY to the strips (widths differ by one, at most kColsOwn): with fixed 62-column strips the
  // last one can be a few columns wide (4219 = 68 x 62 + 3), and the median of so small a block lies so far from the row's
  // that its 48-bin window misses it - the finish launch then needs the exact median of the row (18 % of the rows at 21 x 4219).
  const int strips = int(gridDim.x / unsigned(G)) * (STRIPS ? NW : 1);
  const int c_lo = int((long long)strip * N2 / strips), c_hi = int((long long)(strip + 1) * N2 / strips);   // owned columns [c_lo, c_hi)
  const bool active = STRIPS ? true : ch < nch;
  const int m2 = c_lo - 1 + lane;
  const bool live = m2 >= 0 && m2 < N2 && lane <= c_hi - c_lo + 1;   // the owned columns and one border lane on each side
  const bool own = live && lane >= 1 && lane <= c_hi - c_lo;
  const bool inner = own && m2 >= 1 && m2 <= N2 - 2;          // both neighbours are samples of the same output index
  const int m2c = m2 < 0 ? 0 : (m2 < N2 ? m2 : N2 - 1);       // border lanes outside the grid repeat its first / last column
  const cd* Yg = Y + size_t(g) * N1 * N2 + m2c;
  const int h = (N1 - 1) / 2;
  constexpr bool want_median = !ADAPTIVE;
  constexpr int LANES = 64 * NW;
  if (want_median && !R89) {                                  // histograms of both rows start empty (the loads below are in flight meanwhile)
    unsigned* hz = &hist[0][0];
    for (int q = tid; q < 2 * (kLogBins + 1); q += LANES) hz[q] = 0;
  }
  constexpr int TCD = R89 ? 1 : TC;                            // (the dense form's accumulators do not exist in the Rader form)
  double cx[TCD], sy[TCD], cy[TCD], sx[TCD];
  double sumx = 0, sumy = 0;
  cd y0 = mk(0, 0);
  cd ro[R89 ? kR89Slots : 1];                                  // Rader form: c[t] of the output indices tab->tmap[wave][.] (real part: pair p, imaginary part: pair q)
  cd c0 = mk(0, 0);                                            //             and c[0] (wavefront 0)
  const double kp = zero_rows && zero_rows[2 * g] ? 0.0 : 1.0, kq = zero_rows && 2 * g + 1 < rows && zero_rows[2 * g + 1] ? 0.0 : 1.0;
  if constexpr (R89) {
    r89_columns(Yg, N2, wave, lane, tab, lds_big, ro, c0);
    if (want_median) {                                         // (the exchange plane is free now: it becomes the histograms)
      unsigned* hz = &hist[0][0];
      for (int q = tid; q < 2 * (kLogBins + 1); q += LANES) hz[q] = 0;
    }
    // a pair with a silent microphone: the row is exactly zero in the reference (see k_pfa_cols)
    c0.x *= kp; c0.y *= kq;
#pragma unroll
    for (int i = 0; i < kR89Slots; ++i) { ro[i].x *= kp; ro[i].y *= kq; }
  } else {
    if (active) pfa_cols_accumulate<TC, UNR>(Yg, N1, N2, nch, ch, T, y0, cx, sy, cy, sx, sumx, sumy);
    if (kp == 0.0) { y0.x = sumx = 0.0; }
    if (kq == 0.0) { y0.y = sumy = 0.0; }
#pragma unroll
    for (int tt = 0; tt < TCD; ++tt) {
      cx[tt] *= kp; sy[tt] *= kp; cy[tt] *= kq; sx[tt] *= kq;
    }
  }
  // the samples of this lane, fn(x, t, exists) with wave-uniform t and `exists`: in lag order for the dense form (t = 0 for
  // chunk 0 only, the chunk ascending, the mirrors descending), in table order for the Rader form
  auto each_sample = [&](int r, auto&& fn) {
    if constexpr (R89) {
      const auto* tm = reinterpret_cast<const __attribute__((address_space(4))) int*>(reinterpret_cast<uintptr_t>(&tab->tmap[wave][0]));
      fn(r ? c0.y : c0.x, 0, wave == 0);
#pragma unroll
      for (int i = 0; i < kR89Slots; ++i) fn(r ? ro[i].y : ro[i].x, tm[i], true);
    } else {
      const double base = r ? y0.y : y0.x;
      fn(base + (r ? sumy : sumx), 0, ch == 0);
#pragma unroll
      for (int tt = 0; tt < TCD; ++tt) {
        const int t = ch * TC + tt + 1;
        fn(r ? base + cy[tt] + sx[tt] : base + cx[tt] - sy[tt], t, FULL || t <= h);
      }
#pragma unroll
      for (int tt = TCD - 1; tt >= 0; --tt) {
        const int t = ch * TC + tt + 1;
        fn(r ? base + cy[tt] - sx[tt] : base + cx[tt] + sy[tt], N1 - t, FULL || t <= h);
      }
    }
  };
  __syncthreads();                                             // the zeroed histograms are visible

  // ---- pass A: histogram of |x| (exact counts), maximum, minimum, sums - everything that needs no index
  const int nrow = 2 * g + 1 < rows ? 2 : 1;                   // odd tail: the last transform carries one pair (uniform)
  // byte offset of a sample's bin inside its row's histogram: (key - kBase) * 4 clamped to the table, where key = the
  // exponent and seven mantissa bits of |x| (log_bin); lanes that own nothing are lifted to the dump bin behind the table
  constexpr unsigned kBase4 = unsigned(1023 * 128 - kLogBins) * 4u, kTop4 = kBase4 + unsigned(kLogBins - 1) * 4u;
  const unsigned lift = own ? 0u : unsigned(kLogBins) * 4u;
  const unsigned one = 1u;
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    double vm = -INFINITY, vn = INFINITY, s1 = 0, s2 = 0, a1 = 0;
    if (active && r < nrow) {
      char* const hrow = reinterpret_cast<char*>(&hist[r][0]);
      each_sample(r, [&](double x, int, bool exists) {
        if (!exists) return;
        vm = max_raw(vm, x);
        vn = min_raw(vn, x);
        s1 += x;
        s2 = __builtin_fma(x, x, s2);
        if (ADAPTIVE) a1 += fabs(x);
        if (want_median) {
          const unsigned key4 = (unsigned(__double2hiint(x)) & 0x7fffe000u) >> 11;         // key * 4
          const unsigned off = max(min(max(key4, kBase4), kTop4) - kBase4, lift);                    // (v_med3_u32)
          atomicAdd(reinterpret_cast<unsigned*>(hrow + off), one);
        }
      });
    }
    if (!own) { vm = -INFINITY; vn = INFINITY; s1 = s2 = a1 = 0; }
    for (int o = 32; o > 0; o >>= 1) {
      vm = fmax(vm, shfl_down_d(vm, o));
      vn = fmin(vn, shfl_down_d(vn, o));
      s1 += shfl_down_d(s1, o);
      s2 += shfl_down_d(s2, o);
      if (ADAPTIVE) a1 += shfl_down_d(a1, o);
    }
    if (lane == 0) {
      wmax[wave][r] = vm;
      ColsWaveResult& w = res[wave][r];
      w.vmin = vn; w.s1 = s1; w.s2 = s2; w.a1 = a1;
    }
  }
  __syncthreads();
  // ---- the block's median bin per row (every lane scans PER bins of both rows), then the window around it is published
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
      ex[r] = before + inc[r] - sum[r];                        // samples in the bins under this lane's first bin
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
      // the lane whose bins straddle w0 knows the count under the window; lanes 0 .. kWin-1 copy the window's bins
      if (w0 >= PER * tid && w0 < PER * tid + PER) {
        unsigned below = ex[r];
#pragma unroll
        for (int q = 0; q < PER; ++q) below += PER * tid + q < w0 ? hv[r][q] : 0u;
        bh->win0 = w0; bh->below = below; bh->total = cnt[r]; bh->pad = 0;
      }
      if (tid < kWin) bh->h[tid] = hist[r][w0 + tid];
    }
  }

  // ---- pass B: stores, and the index bookkeeping of maximum / highest strict peak behind the block's exact bounds
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    const int row = 2 * g + r;
    if (row >= rows) continue;
    double vfloor = wmax[0][r];                                // the block's exact maximum
#pragma unroll
    for (int w = 1; w < NW; ++w) vfloor = fmax(vfloor, wmax[w][r]);
    const double pfloor = vfloor > 0 ? 0.8 * vfloor : -INFINITY;
    // one bound per lane: samples with both neighbours in their row are tested from 0.8 of the maximum on, the grid's
    // edge columns only for the maximum itself, border lanes never
    const double myfloor = inner ? pfloor : (own ? vfloor : INFINITY);
    double* const out = corr + size_t(row) * stride + m2c;
    // A lane meets its samples in increasing lag order: the first maximum and the last peak of equal height win
    // inside the lane, the merges compare indices.
    double vmax = -INFINITY, hb = -INFINITY, plat = -INFINITY;
    int imax = -1, mb = -1;
    if (active) {
      each_sample(r, [&](double x, int t, bool exists) {
        if (!exists) return;
        out[N2 * t] = x;                                       // (border lanes store the value their column's owner stores)
        if (__ballot(x >= myfloor)) {
          const int m = m2 + N2 * t;
          // (dense form: a lane meets its samples in lag order; Rader form: equal samples are ordered by their indices here)
          const bool up = own && (x > vmax || (R89 && x == vmax && m < imax));
          vmax = up ? x : vmax;
          imax = up ? m : imax;
          const double left = from_lower_lane(x), right = from_upper_lane(x);
          const bool cand = inner && x >= pfloor && (R89 ? (x > hb || (x == hb && m > mb)) : x >= hb);
          const bool pk = cand && left < x && right < x;
          hb = pk ? x : hb;
          mb = pk ? m : mb;
          // an equal pair (m - 1, m) is reported by its right element (here, or by the finish launch for the grid's edge columns)
          plat = inner && x >= pfloor && left == x ? fmax(plat, x) : plat;
        }
      });
    }
    for (int o = 32; o > 0; o >>= 1) {                         // wavefront reduction (lane 0 holds the result)
      const double ov = shfl_down_d(vmax, o);
      const int oi = __shfl_down(imax, o, 64);
      if (oi >= 0 && (imax < 0 || ov > vmax || (ov == vmax && oi < imax))) { vmax = ov; imax = oi; }
      const double hv = shfl_down_d(hb, o);
      const int hi_ = __shfl_down(mb, o, 64);
      if (hi_ >= 0 && (mb < 0 || higher(hv, hi_, hb, mb))) { hb = hv; mb = hi_; }
      plat = fmax(plat, shfl_down_d(plat, o));
    }
    if (lane == 0) {
      ColsWaveResult& w = res[wave][r];
      w.vmax = vmax; w.hb = hb; w.plat = plat; w.imax = imax; w.mb = mb;
    }
  }
  __syncthreads();
  // ---- publish: lanes 0 / 1 merge the four wavefronts of row p / q
  if (tid < 2 && 2 * g + tid < rows) {
    const int r = tid, row = 2 * g + r;
    Partial pt;
    pt.vmax = pt.hb = pt.plat = -INFINITY;
    pt.vmin = INFINITY;
    pt.imax = pt.imin = pt.mb = -1;
    pt.s1 = pt.s2 = pt.a1 = pt.a2 = 0;
    pt.below = 0;
    pt.pad = 0;
    for (int w = 0; w < NW; ++w) {
      const ColsWaveResult x = res[w][r];
      if (x.imax >= 0 && (pt.imax < 0 || x.vmax > pt.vmax || (x.vmax == pt.vmax && x.imax < pt.imax))) { pt.vmax = x.vmax; pt.imax = x.imax; }
      pt.vmin = fmin(pt.vmin, x.vmin);
      if (x.mb >= 0 && (pt.mb < 0 || higher(x.hb, x.mb, pt.hb, pt.mb))) { pt.hb = x.hb; pt.mb = x.mb; }
      pt.s1 += x.s1; pt.s2 += x.s2; pt.a1 += x.a1;
      pt.plat = fmax(pt.plat, x.plat);
    }
    pt.a2 = pt.s2;                                              // sum |x|^2 = sum x^2 (both shifts are zero on this path)
    pt.imin = 0;                                                // (the finish launch only asks whether the segment has a minimum)
    double bmax = wmax[0][r];
    for (int w = 1; w < NW; ++w) bmax = fmax(bmax, wmax[w][r]);
    pt.pfloor = bmax > 0 ? 0.8 * bmax : -INFINITY;
    pa.parts[size_t(row) * pa.splits + cb] = pt;
  }
}

}  // namespace pal
