// pfa_sample.h - samples of a PHAT row straight from the prime-factor grid Y (before the column pass has written
// the row): the pivot launch of the fused column pass + peak statistics (pfa_cols_stats.h) draws its block sample
// here.  128 columns (16 clusters of 8 neighbours) x 2 TC output indices t per lane, computed like k_pfa_cols does.
#pragma once
#include "fft_core.h"

namespace pal {

constexpr int kPfaTC = 11;   // output indices t per wavefront of the column pass (x 4 accumulators each)
constexpr int kPfaUnr = 4;   // steps j per loop iteration of the column pass (the table is padded with kPfaUnr zero rows)

struct PfaSample {
  const cd* Y;          // [G][N1][N2]
  const double* T;      // cos / sin rows of the column pass: [(j-1) nch 2 TC + ch 2 TC + tt], + TC for the sine
  const int* zero_rows; // per correlation row: 1 = a microphone of the pair is silent, the row is exactly zero (or null)
  int N1, N2, nch;
};

constexpr int kSampleCols = 128;   // columns per row sample; x 4 chunks x 2 TC values = lanes x values per lane

constexpr int kSampleTabMax = 44 * 4 * 2 * kPfaTC;   // table doubles of the largest grid the fused column pass takes (h <= 44, nch <= 4)

// lane `tid` of a 512-lane workgroup: chunk ch = tid / 128 (wave-uniform), column from tid % 128.
// out[tt] = c[t], out[TC + tt] = c[N1 - t] for t = ch TC + tt + 1 (zero where t > h).
// `tab` (LDS, kSampleTabMax doubles) receives the cos / sin rows first: with one workgroup per CU the table (31 KB at
// N1 = 89) does not stay in the scalar cache, and a scalar load per step that goes to L2 made this loop 20 us long.
template <int TC> __device__ __forceinline__ void pfa_sample_row(const PfaSample& sp, int row, int tid, double* tab, double* out) {
  const int g = row >> 1, part = row & 1;
  const int N1 = sp.N1, N2 = sp.N2, h = (N1 - 1) / 2;
  const int ch = __builtin_amdgcn_readfirstlane(tid >> 7), ci = tid & (kSampleCols - 1);
  const int cluster = ci >> 3, within = ci & 7;
  int m2 = int((long long)cluster * (N2 - 8) / 15) + within;
  m2 = m2 < 0 ? 0 : (m2 > N2 - 1 ? N2 - 1 : m2);
  const cd* Yg = sp.Y + size_t(g) * N1 * N2 + m2;
  const int tstep = sp.nch * 2 * TC;
  for (int k = tid; k < h * tstep; k += 512) tab[k] = sp.T[k];
  double ca[TC], sb[TC];
#pragma unroll
  for (int tt = 0; tt < TC; ++tt) ca[tt] = sb[tt] = 0.0;
  const cd y0 = Yg[0];
  const bool chunk_ok = ch < sp.nch;
  constexpr int U = 4;                                         // steps per batch, the next batch's loads in flight (as in the column pass)
  cd yj[U], ym[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const int jj = 1 + u <= h ? 1 + u : h;
    yj[u] = Yg[size_t(jj) * N2];
    ym[u] = Yg[size_t(h > 0 ? N1 - jj : 0) * N2];
  }
  __syncthreads();
  if (chunk_ok) {
    const double* Tj = tab + ch * 2 * TC;
    for (int j = 1; j <= h; j += U) {
      cd nj[U], nm[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int jn = j + U + u <= h ? j + U + u : h;
        nj[u] = Yg[size_t(jn) * N2];
        nm[u] = Yg[size_t(N1 - jn) * N2];
      }
#pragma unroll
      for (int u = 0; u < U; ++u, Tj += tstep) {
        if (j + u > h) continue;                               // (uniform; the LDS copy has no zero rows behind step h)
        // pair p (real parts): cos (y_j.x + y_-j.x) -/+ sin (y_j.y - y_-j.y);  pair q: cos (y_j.y + y_-j.y) +/- sin (y_j.x - y_-j.x)
        const double a = part ? yj[u].y + ym[u].y : yj[u].x + ym[u].x;
        const double b = part ? yj[u].x - ym[u].x : yj[u].y - ym[u].y;
#pragma unroll
        for (int tt = 0; tt < TC; ++tt) {
          ca[tt] = __builtin_fma(Tj[tt], a, ca[tt]);
          sb[tt] = __builtin_fma(Tj[TC + tt], b, sb[tt]);
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) { yj[u] = nj[u]; ym[u] = nm[u]; }
    }
  }
  const double base = part ? y0.y : y0.x;
  const bool zero = sp.zero_rows && sp.zero_rows[row];
#pragma unroll
  for (int tt = 0; tt < TC; ++tt) {
    const bool ok = chunk_ok && ch * TC + tt + 1 <= h;
    const double d = part ? -sb[tt] : sb[tt];
    out[tt] = ok && !zero ? base + ca[tt] - d : 0.0;
    out[TC + tt] = ok && !zero ? base + ca[tt] + d : 0.0;
  }
}

}  // namespace pal
