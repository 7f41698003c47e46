// peak_types.h - what the peak-selection launches (peaks.hip) share with the column pass that feeds them
// directly (pfa_cols_stats.h): per-row parameters, per-segment partial results, the launch arguments.
#pragma once
#include <cstdint>
#include <hip/hip_runtime.h>

namespace pal {

constexpr int kList = 16384;      // capacity of a row's bracket list in global memory (doubles)

struct RowPre {                          // pivot launch -> the other two
  double k0, ka;                         // ~ mean(x), ~ mean(|x|): shifts of the one-pass sums
  double lo, hi;                         // pivots around the median of |x| (0 / inf when no median is needed)
  double vfloor, pfloor;                 // fused column pass: lower bounds of the row's maximum and of its highest strict
                                         // peak, taken from the block sample (-inf: none); samples below cannot be either
};

struct Partial {                         // one segment's share of the streaming pass
  double vmax, vmin, hb, s1, s2, a1, a2;
  double plat;                           // fused column pass only: highest sample with an equal neighbour (-inf: none)
  double pfloor;                         // fused column pass only: samples of this segment below it had no peak test
  long long below;
  int imax, imin, mb, pad;
};

// Fused column pass: a column block's histogram of |x| around ITS median, 128 logarithmic bins per octave (seven mantissa
// bits) over the sixteen octaves below 1.0 (a PHAT sequence never exceeds 1).  Counts are exact; the finish launch adds
// the blocks' windows where they overlap and reads off the bin that holds the row's median: a rigorous interval of
// relative width 0.5 % - enough to decide every threshold comparison that is not inside it (the exact median is computed
// from the row only when one is).
constexpr int kLogBins = 2048;           // 128 bins per octave x 16 octaves
constexpr int kWin = 48;                 // bins a block publishes (its median bin -24 .. +23; the blocks' medians differ by ~3 bins)
struct BlockHist {
  int win0;                              // first bin of the window
  unsigned below;                        // samples of the block in bins under win0
  unsigned total;                        // samples of the block
  unsigned pad;
  unsigned h[kWin];
};

__host__ __device__ inline int log_bin(double mag) {                    // bin 2047 = [2^(-1/128), 1) and everything above, bin 0 = everything below 2^-16
  const long long bits = __builtin_bit_cast(long long, mag) & 0x7fffffffffffffffll;
  const int key = int(bits >> 45);                                      // 11 exponent bits + 7 mantissa bits
  const int b = key - (1023 * 128 - kLogBins);
  return b < 0 ? 0 : (b > kLogBins - 1 ? kLogBins - 1 : b);
}
__host__ __device__ inline double log_bin_floor(int b) {                // smallest magnitude of bin b (0 for bin 0, infinity past the last)
  if (b <= 0) return 0.0;
  if (b > kLogBins - 1) return __builtin_huge_val();
  return __builtin_bit_cast(double, (long long)(b + (1023 * 128 - kLogBins)) << 45);
}

struct PeakArgs {
  const double* corr;
  size_t stride;
  int n, n2;
  double fs, mult, med;   // med: NaN = no window
  int method, dist, num_peaks, snr_w;   // method: 0 median, 1 adaptive, < 0 metrics only
  int splits, tiles_per_seg;             // segments per row, tiles per segment
  int local_pivots;                      // 1: fused column pass - no RowPre, no bracket lists: every segment brought a histogram
                                         //    window (BlockHist) and its own bound for untested samples (Partial.pfloor)
  BlockHist* bh;                         // [rows][splits] (local_pivots only)
  int edge_n2;                           // > 0: segments are column blocks of the prime-factor grid (row length edge_n2; block q holds the
                                         //      columns [q n2 / splits, (q + 1) n2 / splits));
                                         //      the finish launch tests the samples of columns 0 and edge_n2 - 1 itself
  RowPre* pre;                           // [rows]
  Partial* parts;                        // [rows][splits]
  double* glist;                         // [rows][kList] bracket values
  int* gcount;                           // [rows] fill of glist (> kList: overflow, the finish kernel re-reads the row)
  unsigned long long* stamps;            // diagnostics (PAL_DEBUG_STAMPS=1): [rows][8] 100 MHz clock reads of the finish launch
  int memo_cap, stack_cap;               // capacities of the on-chip memo / stack of the distance rule (PAL_DEBUG_MEMO shrinks them: tests of the slow path)
  unsigned* bits;                        // [rows][2][(n + 31) / 32] scratch bitmaps of the exact slow path of the distance rule (peaks.hip resolve_slow)
};

__device__ __forceinline__ bool higher(double h1, int m1, double h2, int m2) {   // priority(h1,m1) > priority(h2,m2)
  return h1 > h2 || (h1 == h2 && m1 > m2);
}

// one row's parameters through the scalar cache (uniform address, written by the previous launch)
__device__ __forceinline__ RowPre load_pre(const RowPre* pre, int row) {
  const auto* p = reinterpret_cast<const __attribute__((address_space(4))) double*>(reinterpret_cast<uintptr_t>(pre)) + 6 * size_t(row);
  RowPre r;
  r.k0 = p[0]; r.ka = p[1]; r.lo = p[2]; r.hi = p[3]; r.vfloor = p[4]; r.pfloor = p[5];
  return r;
}

// neighbour lanes through DPP wave shifts (one VALU move per dword, no LDS crossbar)
__device__ __forceinline__ double from_lower_lane(double v) {   // lane i receives lane i - 1 (lane 0: zero, never used)
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_mov_dpp(lo, 0x138, 0xf, 0xf, true);           // wave_shr:1
  hi = __builtin_amdgcn_mov_dpp(hi, 0x138, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double from_upper_lane(double v) {   // lane i receives lane i + 1 (lane 63: zero, never used)
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_mov_dpp(lo, 0x130, 0xf, 0xf, true);           // wave_shl:1
  hi = __builtin_amdgcn_mov_dpp(hi, 0x130, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}

}  // namespace pal
