// pfa_cols_stats.h - column pass of the prime-factor route that also does the streaming pass of the peak
// selection (gfx950, fp64).
//
// k_pfa_cols (pfa_kernels.h) writes the correlation rows and k_peak_stream (peaks.hip) reads them back once for
// the statistics the finish launch needs: max / argmax, min, the shifted sums, the highest local maximum, the
// count below the lower pivot and the values between the pivots.  Here the values meet those statistics in the
// registers they were accumulated in, and the 0.7 MB per row are only written (the finish launch still reads the
// few hundred samples around the peaks it resolves).
//
//   - a workgroup owns 62 columns m2 of one packed transform (both rows: pair p = real parts, pair q = imaginary
//     parts) and all N1 output indices t, its four wavefronts four chunks of kPfaTC indices and their mirrors;
//     lanes 0 and 63 compute the neighbouring blocks' border columns again, so that every owned sample has both
//     neighbours m -/+ 1 = (m2 -/+ 1, t) one lane away (DPP wave shifts)
//   - the first and last column of the grid have their neighbours in another output index: their peak test is left
//     to the finish launch (2 N1 samples per row), like the samples with an equal neighbour (plateaus), which are
//     only reported
//   - a lane meets its samples in increasing lag order (t = 0, the chunk ascending, the mirrors descending), so the
//     first maximum / last peak of equal height win without index comparisons, as in the stream kernel
//   - bracket values: a private LDS list per wavefront and row (ballot + mbcnt, no atomics), one global atomic per
//     workgroup and row
//   - the pivots come from k_peak_pivots_grid (a block sample computed from the grid, pfa_sample.h) in front of
//     this launch
#pragma once
#include <cmath>

#include "peak_types.h"
#include "pfa_kernels.h"

namespace pal {

constexpr int kColsOwn = 62;      // columns a workgroup of the fused column pass owns (64 lanes - two border lanes)
constexpr int kColsList = 256;    // bracket values per wavefront and row (about 95 expected at 6 sigma pivots)

struct ColsWaveResult {           // one wavefront's share of a row segment
  double vmax, vmin, hb, s1, s2, a1, a2, plat;
  int imax, mb, below, pad;
};

__device__ __forceinline__ double shfl_down_d(double v, int o) { return __shfl_down(v, o, 64); }

template <int TC, int UNR>
__global__ __launch_bounds__(256) void k_pfa_cols_stats(const cd* __restrict__ Y, double* __restrict__ corr, size_t stride, int N1, int N2,
                                                        int G, int nch, const double* __restrict__ T, const int* __restrict__ zero_rows, PeakArgs pa,
                                                        int rows) {
  __shared__ double list[4][2][kColsList + 1];                // + one dump slot for the unconditional stores
  __shared__ ColsWaveResult res[4][2];
  __shared__ int lcount[4][2];
  __shared__ int gbase[2];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ch = wave;                                        // nch <= 4: one workgroup covers every output index
  const bool active = ch < nch;
  const int g = blockIdx.x % G, cb = blockIdx.x / G;
  const int m2 = cb * kColsOwn - 1 + lane;
  const bool live = m2 >= 0 && m2 < N2;
  const bool own = live && lane >= 1 && lane <= kColsOwn;
  const bool inner = own && m2 >= 1 && m2 <= N2 - 2;          // both neighbours are samples of the same output index
  const int m2c = m2 < 0 ? 0 : (m2 < N2 ? m2 : N2 - 1);       // border lanes outside the grid repeat its first / last column
  const cd* Yg = Y + size_t(g) * N1 * N2 + m2c;
  const int h = (N1 - 1) / 2;
  double cx[TC], sy[TC], cy[TC], sx[TC];
  double sumx = 0, sumy = 0;
  cd y0 = mk(0, 0);
  if (active) pfa_cols_accumulate<TC, UNR>(Yg, N1, N2, nch, ch, T, y0, cx, sy, cy, sx, sumx, sumy);
  const bool want_median = pa.method == 0;
  {   // a pair with a silent microphone: the row is exactly zero in the reference (see k_pfa_cols)
    const double kp = zero_rows && zero_rows[2 * g] ? 0.0 : 1.0, kq = zero_rows && 2 * g + 1 < rows && zero_rows[2 * g + 1] ? 0.0 : 1.0;
    if (kp == 0.0) { y0.x = sumx = 0.0; }
    if (kq == 0.0) { y0.y = sumy = 0.0; }
#pragma unroll
    for (int tt = 0; tt < TC; ++tt) {
      cx[tt] *= kp; sy[tt] *= kp; cy[tt] *= kq; sx[tt] *= kq;
    }
  }
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    const int row = 2 * g + r;
    if (row >= rows) { if (lane == 0) lcount[wave][r] = 0; continue; }     // odd tail: the last transform has one pair (uniform)
    const RowPre pre = load_pre(pa.pre, row);
    const double k0 = pre.k0, ka = pre.ka, lo = pre.lo, hi = pre.hi, vfloor = pre.vfloor, pfloor = pre.pfloor;
    double* const out = corr + size_t(row) * stride + m2c;
    // Per-lane state.  A sample below the pivot launch's bounds (vfloor <= the row's maximum, pfloor <= its highest
    // strict peak) can be neither, so the index bookkeeping and the neighbour exchange sit behind a wave-uniform branch
    // that about one step in fifty takes.  A lane meets its samples in increasing lag order: the first maximum and the
    // last peak of equal height win inside the lane, the merges compare indices.
    double vmax = -INFINITY, vmin = INFINITY, hb = -INFINITY, plat = -INFINITY;
    int imax = -1, mb = -1;
    double s1 = 0, s2 = 0, a1 = 0, a2 = 0;
    int below = 0, run = 0;                                    // wave-uniform counts
    double* mylist = list[wave][r];
    auto sample = [&](double x, int t, bool exists) {          // `exists` is wave-uniform
      if (!exists) return;
      out[N2 * t] = x;                                         // (border lanes store the value their column's owner stores)
      vmin = fmin(vmin, x);                                    // (lanes that own nothing are reset below)
      const double mag = fabs(x);
      const double d = x - k0, e = mag - ka;
      s1 += d;
      s2 = __builtin_fma(d, d, s2);
      a1 += e;
      a2 = __builtin_fma(e, e, a2);
      if (__ballot((own && x >= vfloor) || (inner && x >= pfloor))) {
        const int m = m2 + N2 * t;
        const bool up = own && x > vmax;
        vmax = up ? x : vmax;
        imax = up ? m : imax;
        const double left = from_lower_lane(x), right = from_upper_lane(x);
        const bool cand = inner && x >= pfloor && x >= hb;
        const bool pk = cand && left < x && right < x;
        hb = pk ? x : hb;
        mb = pk ? m : mb;
        // an equal pair (m - 1, m) is reported by its right element (here, or by the finish launch for the grid's edge columns)
        plat = inner && x >= pfloor && left == x ? fmax(plat, x) : plat;
      }
      if (want_median) {
        below += __popcll(__ballot(own && mag < lo));
        const bool in = own && mag >= lo && mag <= hi;
        const unsigned long long mask = __ballot(in);
        const int at = run + int(__builtin_amdgcn_mbcnt_hi(unsigned(mask >> 32), __builtin_amdgcn_mbcnt_lo(unsigned(mask), 0u)));
        mylist[in ? min(at, kColsList) : kColsList] = mag;     // unconditional store, one dump slot
        run += __popcll(mask);
      }
    };
    if (active) {
      const double base = r ? y0.y : y0.x;
      sample(base + (r ? sumy : sumx), 0, ch == 0);
#pragma unroll
      for (int tt = 0; tt < TC; ++tt) {
        const int t = ch * TC + tt + 1;
        sample(r ? base + cy[tt] + sx[tt] : base + cx[tt] - sy[tt], t, t <= h);
      }
#pragma unroll
      for (int tt = TC - 1; tt >= 0; --tt) {
        const int t = ch * TC + tt + 1;
        sample(r ? base + cy[tt] - sx[tt] : base + cx[tt] + sy[tt], N1 - t, t <= h);
      }
    }
    if (!own) { vmin = INFINITY; s1 = s2 = a1 = a2 = 0; }
    // wavefront reduction (lane 0 holds the result)
    for (int o = 32; o > 0; o >>= 1) {
      const double ov = shfl_down_d(vmax, o);
      const int oi = __shfl_down(imax, o, 64);
      if (oi >= 0 && (imax < 0 || ov > vmax || (ov == vmax && oi < imax))) { vmax = ov; imax = oi; }
      const double hv = shfl_down_d(hb, o);
      const int hi_ = __shfl_down(mb, o, 64);
      if (hi_ >= 0 && (mb < 0 || higher(hv, hi_, hb, mb))) { hb = hv; mb = hi_; }
      vmin = fmin(vmin, shfl_down_d(vmin, o));
      s1 += shfl_down_d(s1, o);
      s2 += shfl_down_d(s2, o);
      a1 += shfl_down_d(a1, o);
      a2 += shfl_down_d(a2, o);
      plat = fmax(plat, shfl_down_d(plat, o));
    }
    if (lane == 0) {
      ColsWaveResult w;
      w.vmax = vmax; w.vmin = vmin; w.hb = hb; w.s1 = s1; w.s2 = s2; w.a1 = a1; w.a2 = a2; w.plat = plat;
      w.imax = imax; w.mb = mb; w.below = below; w.pad = 0;
      res[wave][r] = w;
      lcount[wave][r] = run;                                   // (> kColsList: the private list overflowed)
    }
  }
  __syncthreads();
  // ---- publish: lanes 0 / 1 merge the four wavefronts of row p / q; the bracket values join the rows' global lists
  if (tid < 2 && 2 * g + tid < rows) {
    const int r = tid, row = 2 * g + r;
    Partial pt;
    pt.vmax = pt.hb = pt.plat = -INFINITY;
    pt.vmin = INFINITY;
    pt.imax = pt.imin = pt.mb = -1;
    pt.s1 = pt.s2 = pt.a1 = pt.a2 = 0;
    pt.below = 0;
    pt.pad = 0;
    int total = 0;
    bool overflow = false;
    for (int w = 0; w < 4; ++w) {
      const ColsWaveResult x = res[w][r];
      if (x.imax >= 0 && (pt.imax < 0 || x.vmax > pt.vmax || (x.vmax == pt.vmax && x.imax < pt.imax))) { pt.vmax = x.vmax; pt.imax = x.imax; }
      pt.vmin = fmin(pt.vmin, x.vmin);
      if (x.mb >= 0 && (pt.mb < 0 || higher(x.hb, x.mb, pt.hb, pt.mb))) { pt.hb = x.hb; pt.mb = x.mb; }
      pt.s1 += x.s1; pt.s2 += x.s2; pt.a1 += x.a1; pt.a2 += x.a2;
      pt.below += x.below;
      pt.plat = fmax(pt.plat, x.plat);
      overflow = overflow || lcount[w][r] > kColsList;
      total += lcount[w][r];
    }
    pt.imin = 0;                                                // (the finish launch only asks whether the segment has a minimum)
    pa.parts[size_t(row) * pa.splits + cb] = pt;
    gbase[r] = want_median ? atomicAdd(&pa.gcount[row], overflow ? kList + 1 : total) : -1;   // an overflow poisons the list
  }
  __syncthreads();
  if (want_median) {
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      if (2 * g + r >= rows) continue;
      int before = 0, total = 0;
      bool overflow = false;
      for (int w = 0; w < 4; ++w) {
        const int c = lcount[w][r];
        overflow = overflow || c > kColsList;
        before += w < wave ? c : 0;
        total += c;
      }
      const int at = gbase[r];
      if (overflow || at < 0 || at + total > kList) continue;
      double* dst = pa.glist + size_t(2 * g + r) * kList + at + before;
      const int mine = lcount[wave][r];
      for (int k = lane; k < mine; k += 64) dst[k] = list[wave][r][k];
    }
  }
}

}  // namespace pal
