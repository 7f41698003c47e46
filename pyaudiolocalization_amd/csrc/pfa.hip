// pfa.hip - prime-factor route of the inverse PHAT transform (gfx950, fp64).
//
// numpy.fft.ifft at the exact length n = n1 + n2 - 1 (utils.py:118) is where the reference spends its time.
// The four-step chirp convolution of bluestein.hip works for every n but moves a 2n-point workspace through
// HBM three times.  When n = N1 x N2 with coprime odd factors, N1 <= 127 and N2 <= 2048 (the headline length
// 2 * 44100 - 1 = 88199 = 89 x 991 is such an n), the Chinese-remainder map k <-> (k mod N1, k mod N2) on the
// frequency side turns the length-n DFT into independent short DFTs:
//
//   c[m2 + N2 t] = 1/n  sum_k1 e^(2 pi i k1 t / N1) * e^(2 pi i u1 k1 m2 / N1) * sum_k2 e^(2 pi i u2 k2 m2 / N2) x[k1, k2]
//                       '------ k_pfa_cols: dense N1-point DFT ------'   '------ k_pfa_rows: N2-point DFT + twiddle ------'
//
// (u1 = N2^-1 mod N1, u2 = N1^-1 mod N2).  The N2-point DFTs are convolutions that live entirely in LDS - Rader's cyclic
// convolution of N2 - 1 points where N2 = 991 (pfa_rader.h: 990 = 9 x 10 x 11, twiddle-free prime-factor stages), else a
// chirp convolution of length M = 2^lm >= 2 N2 - 1 (pfa_kernels.h: forward FFT, multiply by the chirp spectrum, inverse
// FFT, two tiles per workgroup) - and the N1-point DFTs are real cos / sin contractions whose coefficients are wave-uniform and
// come through the scalar cache.  A transform touches HBM twice (16 B/point each way) instead of three times
// at twice the points, and the output index m = m2 + N2 t is the natural one, so the correlation rows are
// written in coalesced runs.
//
// As in bluestein.hip one complex transform carries two mic pairs (real part = pair p, imaginary part = pair q)
// and the whitened cross spectrum R = S_a conj(S_b) / (|.| + 1e-10) (utils.py:116-117) is built inside the
// first FFT stage.  The mic spectra arrive in the permuted layout SP[mic][k1][k2], k1 <= (N1-1)/2 (written by the
// forward transform: pfa_forward.h for Rader rows - the same cut, transposed - or the storer of the four-step route,
// bluestein.hip PermSpectrumStorer): row N1-k1
// is row k1 reversed and conjugated (Hermitian symmetry), so one workgroup serves both rows from the same
// loads - tile 0 transforms x[k1, e], tile 1 the reversed row z[e] = x[N1-k1, -e], whose DFT is the reversed DFT.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <utility>
#include <vector>

#include "pfa_kernels.h"
#include "pfa_rader.h"
#include "pfa_cols_stats.h"
#include "pfa_cols_fin.h"
#include "pfa_forward.h"
#include "pfa_big.h"

namespace pal {

// points per lane of the register-resident row tiles (pfa_big.h); the build can override them for A/B runs
#ifndef PAL_BIG13_PTS
#define PAL_BIG13_PTS 32
#endif
#ifndef PAL_BIG14_PTS
#define PAL_BIG14_PTS 32
#endif
constexpr int kBig13 = PAL_BIG13_PTS, kBig14 = PAL_BIG14_PTS;

// ------------------------------------------------------------------ plan
static long long inv_mod(long long a, long long m) {   // a^-1 mod m (gcd = 1), 0 when m == 1
  if (m == 1) return 0;
  long long g = m, x = 0, y = 1, aa = a % m;
  while (aa) {
    const long long q = g / aa;
    long long t = g - q * aa; g = aa; aa = t;
    t = x - q * y; x = y; y = t;
  }
  return g == 1 ? ((x % m) + m) % m : -1;
}

static long long gcd_ll(long long a, long long b) { while (b) { long long t = a % b; a = b; b = t; } return a; }

void Engine::free_pfa(Pfa& f) {
  if (f.b) (void)hipFree(f.b);
  if (f.hhat) (void)hipFree(f.hhat);
  if (f.r1) (void)hipFree(f.r1);
  if (f.T) (void)hipFree(f.T);
  if (f.rowtab) (void)hipFree(f.rowtab);
  if (f.r89) (void)hipFree(f.r89);
  for (void* p : {(void*)f.rd_bhat, (void*)f.rd_bhat_f, (void*)f.rd_qidx,
                  (void*)f.rd_ridx})
    if (p) (void)hipFree(p);
  f = Pfa();
}

#define PAL_SWITCH_LM(lm, ...)                                     \
  switch (lm) {                                                    \
    case 9: { constexpr int LM = 9; __VA_ARGS__; } break;          \
    case 10: { constexpr int LM = 10; __VA_ARGS__; } break;        \
    case 11: { constexpr int LM = 11; __VA_ARGS__; } break;        \
    case 12: { constexpr int LM = 12; __VA_ARGS__; } break;        \
    default: return fail(PAL_ERR_INTERNAL, "prime-factor plan with log2 M = %d", lm); \
  }

int Engine::build_pfa(Plan& pl) {
  pl.pfa = Pfa();
  const long long n = pl.n;
  if (!allow_pfa || pl.nout != pl.n || n < 3 || (n & 1) == 0) return PAL_OK;
  // Best coprime split by a time model fitted to this part (microseconds per packed transform inside a launch group of
  // 240): a row tile of 2^lm points costs w[lm] - 0.005 / 0.011 / 0.026 for the LDS-resident tiles of
  // 1024 / 2048 / 4096 points (two per workgroup), 0.050 / 0.145 for the register-resident tiles of 8192 / 16384 (pfa_big.h:
  // the 16384-point tile runs one workgroup per CU) - and the column pass 1.15e-5 n for its
  // traffic plus 4e-8 N1 n for the dense N1-point DFTs.  The four-step route: 1.6e-5 per point of its convolution for the
  // first two passes, 2.2e-5 n for the last pass and the statistics launches, x 0.86 with register-resident rows.  A split that cannot take the fused column
  // pass (more than four chunks of output indices: N1 > 89) must beat the four-step route by 15 %.
  static const double kTile[15] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0.003, 0.005, 0.0113, 0.026, 0.050, 0.145};
  double best = -1;
  int bn1 = 0, bn2 = 0, blm = 0;
  for (long long d = 1; d <= 127 && d <= n; d += 2) {
    if (n % d) continue;
    const long long r = n / d;
    if (r > (allow_big ? 8192 : 2048) || gcd_ll(d, r) != 1) continue;
    int lm = 9;
    while ((1ll << lm) < 2 * r - 1) ++lm;
    // (16384-point tiles run one workgroup per CU for 45 us each: with few rows per transform the rounds of workgroups and the
    //  per-group fixed costs weigh more than the model says - 24 051 = 3 x 8017 measured 1.41 M pairs/s against 1.84 M on the
    //  four-step route, 48 k-point lengths break even: profiles/r02_g_length_sweep_*.csv)
    if (lm == 14 && n < 40000) continue;
    const int chunks = d > 1 ? int(((d - 1) / 2 + kPfaTC - 1) / kPfaTC) : 1;
    double cost = double(d) * kTile[lm] + double(n) * (1.15e-5 + 4e-8 * double(d));
    if (chunks > 4) cost *= 1.15;
    if (best < 0 || cost < best) { best = cost; bn1 = int(d); bn2 = int(r); blm = lm; }
  }
  const double four_step = (1.6e-5 * double(pl.inv.M()) + 2.2e-5 * double(n)) * (pl.inv.reg ? 0.86 : 1.0);
  if (best < 0 || best > four_step) return PAL_OK;             // the four-step route is no worse
  // (short transforms are launch-bound, the model does not apply: there the split must also stay within 1.5 x the
  //  four-step route's points, round 1's rule)
  if (n < 16384 && size_t(((bn1 + 1) / 2) * (2ll << blm)) > pl.inv.M() + pl.inv.M() / 2) return PAL_OK;
  Pfa f;
  f.n1 = bn1; f.n2 = bn2; f.lm = blm;
  f.u1 = int(inv_mod(bn2, bn1));
  long long u2 = inv_mod(bn1, bn2);
  if (bn2 == 1) u2 = 0;
  f.e1 = (long long)bn2 * f.u1 % n;
  f.e2 = (long long)bn1 * u2 % n;
  if (u2 & 1) u2 += bn2;                       // even multiplier: the chirp is periodic mod N2 and symmetric
  const int h = (bn1 - 1) / 2;
  f.nch = h > 0 ? (h + kPfaTC - 1) / kPfaTC : 1;
  const cd* tws = blm >= 13 ? stage_table(blm) : (blm >= 11 ? stage_table_compact(blm) : stage_table(blm));
  if (!tws || !stage_table(blm)) return fail(PAL_ERR_NOMEM, "twiddle tables");   // (the full table feeds the register twiddles)
  PAL_HIP(hipMalloc(&f.b, sizeof(cd) * bn2));
  PAL_HIP(hipMalloc(&f.hhat, sizeof(cd) << blm));
  PAL_HIP(hipMalloc(&f.r1, sizeof(cd) * bn1));
  k_make_chirp<<<dim3((bn2 + 255) / 256), dim3(256), 0, stream>>>(f.b, bn2, int(u2));
  k_make_roots<<<dim3((bn1 + 255) / 256), dim3(256), 0, stream>>>(f.r1, bn1, double(bn1));
  const double scale = 1.0 / (double(1 << blm) * double(n));   // inverse FFT_M and numpy.fft.ifft's 1/n
  if (blm == 13) k_pfa_hhat_big<13, kBig13><<<dim3(1), dim3(BigTile<13, kBig13>::kLanes), 0, stream>>>(f.b, bn2, f.hhat, scale, tws);
  else if (blm == 14) k_pfa_hhat_big<14, kBig14><<<dim3(1), dim3(BigTile<14, kBig14>::kLanes), 0, stream>>>(f.b, bn2, f.hhat, scale, tws);
  else PAL_SWITCH_LM(blm, k_pfa_hhat<LM><<<dim3(1), dim3(PfaLds<LM>::kLanes), 0, stream>>>(f.b, bn2, f.hhat, scale, tws));
  PAL_HIP(hipGetLastError());
  // cos / sin of 2 pi j t / N1 with the argument reduced exactly (j t mod N1) before the long-double evaluation
  std::vector<double> tab(size_t(h + kPfaUnr) * f.nch * 2 * kPfaTC, 0.0);   // kPfaUnr zero rows behind the last step
  const long double two_pi = 6.283185307179586476925286766559005768L;
  for (int j = 1; j <= h; ++j)
    for (int t = 1; t <= h; ++t) {
      const int ch = (t - 1) / kPfaTC, tt = (t - 1) % kPfaTC;
      const long double ang = two_pi * (long double)((long long)j * t % bn1) / (long double)bn1;
      double* row = &tab[(size_t(j - 1) * f.nch + ch) * 2 * kPfaTC];
      row[tt] = double(cosl(ang));
      row[kPfaTC + tt] = double(sinl(ang));
    }
  // twiddle bookkeeping of the row pass's last stage: per row of Y the index multiplier and its step (pfa_kernels.h)
  {
    const int ll = stage_tw_last(blm);
    const long long P = 1ll << ll;
    std::vector<int2> rt;
    rt.resize(size_t(bn1));
    for (int row = 0; row < bn1; ++row) {
      const long long uk = (long long)f.u1 * row % bn1;
      rt[size_t(row)] = make_int2(int(uk), int(uk * P % bn1));
    }
    PAL_HIP(hipMalloc(&f.rowtab, rt.size() * sizeof(int2)));
    PAL_HIP(hipMemcpy(f.rowtab, rt.data(), rt.size() * sizeof(int2), hipMemcpyHostToDevice));
  }
  PAL_HIP(hipMalloc(&f.T, tab.size() * sizeof(double)));
  PAL_HIP(hipMemcpyAsync(f.T, tab.data(), tab.size() * sizeof(double), hipMemcpyHostToDevice, stream));
  PAL_HIP(hipStreamSynchronize(stream));       // `tab` is host memory
  if (allow_rader && bn2 == 991) PAL_TRY(build_rader(f, n, u2 % bn2));   // 991 is prime and 990 = 11 x 9 x 10
  if (allow_r89 && bn1 == kR89 && f.nch == 4) {                // the 89-point column DFT as Rader's 8 x 11 convolution (pfa_rader89.h)
    Rader89Tab tab;
    make_rader89_tab(tab);
    PAL_HIP(hipMalloc(&f.r89, sizeof tab));
    PAL_HIP(hipMemcpy(f.r89, &tab, sizeof tab, hipMemcpyHostToDevice));
  }
  pl.pfa = f;
  return PAL_OK;
}

template <class T> static int upload_table(Engine* e, const std::vector<T>& host, T** dev) {
  if (hipMalloc(dev, host.size() * sizeof(T)) != hipSuccess) return e->fail(PAL_ERR_NOMEM, "rader tables");
  return e->check(hipMemcpy(*dev, host.data(), host.size() * sizeof(T), hipMemcpyHostToDevice), "rader tables");
}

// Rader tables of the prime N2 = f.n2 (pfa_rader.h): generator re-indexing into prime-factor positions and the 3-D
// spectra of the kernel sequences w^(g^s), w = exp(+/- 2 pi i u2 / N2)
int Engine::build_rader(Pfa& f, long long n, long long u2) {
  constexpr int R1 = 11, R2 = 9, R3 = 10;
  const int p = f.n2, L = p - 1;
  if (L != R1 * R2 * R3) return PAL_OK;
  int g = 0;                                                   // smallest primitive root: order L, checked on L's prime factors
  for (int c = 2; c < p && !g; ++c) {
    bool ok = true;
    for (int q : {2, 3, 5, 11}) {
      long long x = 1, b = c;
      for (int ex = L / q; ex; ex >>= 1, b = b * b % p)
        if (ex & 1) x = x * b % p;
      ok = ok && x != 1;
    }
    if (ok) g = c;
  }
  if (!g) return PAL_OK;
  using AX = Axes<R1, R2, R3>;                                 // prime-factor positions of the convolution index (mixed_radix.h)
  std::vector<int> gpow(L), qidx(p, 0), ridx(p, 0);
  long long x = 1;
  for (int s = 0; s < L; ++s, x = x * g % p) gpow[s] = int(x);
  for (int s = 0; s < L; ++s) {
    ridx[gpow[s]] = AX::pos(s);                                // X[g^s] = x[0] + C[s]
    qidx[gpow[(L - s) % L]] = AX::pos(s);                      // a[s] = x[g^-s]
  }
  qidx[0] = L;                                                 // the rows of SP keep this order, bin 0 last
  const long double two_pi = 6.283185307179586476925286766559005768L;
  std::vector<long double> br(L), bi(L);                       // kernel sequence b[s] = w^(g^s), w = exp(2 pi i u2 / N2)
  for (int s = 0; s < L; ++s) {
    const long double ang = two_pi * (long double)(u2 * gpow[s] % p) / (long double)p;
    br[s] = cosl(ang);
    bi[s] = sinl(ang);
  }
  std::vector<long double> cr(L), ci(L);                       // exp(-2 pi i m / L)
  for (int s = 0; s < L; ++s) {
    const long double ang = -two_pi * (long double)s / (long double)L;
    cr[s] = cosl(ang);
    ci[s] = sinl(ang);
  }
  // 3-D spectrum at position k2 + R2 k3 + R2 R3 k1: sum_s b[s] exp(-2 pi i ((s mod R2) k2 / R2 + (s mod R3) k3 / R3 + (s mod R1) k1 / R1));
  // inverse direction scaled by 1 / (L n) (the unnormalised inverse stages and numpy's 1 / n), forward direction (the
  // conjugate sequence) by 1 / L
  std::vector<cd> bhat(L), bhat_f(L);
  const long double scale = 1.0L / ((long double)L * (long double)n);
  for (int k1 = 0; k1 < R1; ++k1)
    for (int k3 = 0; k3 < R3; ++k3)
      for (int k2 = 0; k2 < R2; ++k2) {
        long double ar = 0, ai = 0, fr = 0, fi = 0;
        for (int s = 0; s < L; ++s) {
          const int m = int(((long long)(s % R2) * k2 * (L / R2) + (long long)(s % R3) * k3 * (L / R3) + (long long)(s % R1) * k1 * (L / R1)) % L);
          ar += br[s] * cr[m] - bi[s] * ci[m];
          ai += br[s] * ci[m] + bi[s] * cr[m];
          fr += br[s] * cr[m] + bi[s] * ci[m];                 // (br - i bi)(cr + i ci)
          fi += br[s] * ci[m] - bi[s] * cr[m];
        }
        const size_t at = size_t(k2) + size_t(R2) * k3 + size_t(R2) * R3 * k1;
        bhat[at] = mk(double(ar * scale), double(ai * scale));
        bhat_f[at] = mk(double(fr / (long double)L), double(fi / (long double)L));
      }
  PAL_TRY(upload_table(this, bhat, &f.rd_bhat));
  PAL_TRY(upload_table(this, bhat_f, &f.rd_bhat_f));
  PAL_TRY(upload_table(this, qidx, &f.rd_qidx));
  PAL_TRY(upload_table(this, ridx, &f.rd_ridx));
  f.rader = true;
  return PAL_OK;
}

int Engine::pfa_rows(const Plan& pl, const cd* permuted, const int4* quads, int G, cd* Y, hipStream_t on) {
  const Pfa& f = pl.pfa;
  const cd* tws = f.lm >= 11 ? stage_table_compact(f.lm) : stage_table(f.lm);
  if (f.rader) {
    ProfScope ps(this, "k_pfa_rows_rader<11,9,10>", on);
    PfaRaderArgs a{permuted, quads, Y, f.rd_bhat, f.r1, f.rd_ridx, f.rowtab,
                   f.n1, f.n2, f.rows(), G, 1.0f / float(f.n1), 1.0 / double(pl.n), nullptr, xcd_rows};
    k_pfa_rows_rader<11, 9, 10><<<dim3(row_work_grid(G, f.rows(), xcd_rows)), dim3(256), 0, on>>>(a);
    PAL_HIP(hipGetLastError());
  } else if (f.lm >= 13) {
    char name[48];
    snprintf(name, sizeof name, "k_pfa_rows_big<%d>", f.lm);
    ProfScope ps(this, name, on);
    PfaRowsArgs a{permuted, quads, Y, f.b, f.hhat, f.r1, stage_table(f.lm), stage_table(f.lm), f.rowtab, f.n1, f.n2, f.rows(), G, f.u1, 1.0f / float(f.n1), nullptr, xcd_rows};
    const unsigned grid = row_work_grid(G, f.n1, xcd_rows);    // one workgroup per row of Y
    if (f.lm == 13) k_pfa_rows_big<13, kBig13><<<dim3(grid), dim3(BigTile<13, kBig13>::kLanes), 0, on>>>(a);
    else k_pfa_rows_big<14, kBig14><<<dim3(grid), dim3(BigTile<14, kBig14>::kLanes), 0, on>>>(a);
    PAL_HIP(hipGetLastError());
  } else {
    char name[48];
    snprintf(name, sizeof name, "k_pfa_rows<%d>", f.lm);
    ProfScope ps(this, name, on);
    PfaRowsArgs a{permuted, quads, Y, f.b, f.hhat, f.r1, tws, stage_table(f.lm), f.rowtab, f.n1, f.n2, f.rows(), G, f.u1, 1.0f / float(f.n1), nullptr, xcd_rows};
    const unsigned grid = row_work_grid(G, f.rows(), xcd_rows);
    PAL_SWITCH_LM(f.lm, k_pfa_rows<LM><<<dim3(grid), dim3(PfaLds<LM>::kLanes), 0, on>>>(a));
    PAL_HIP(hipGetLastError());
  }
  return PAL_OK;
}

int Engine::pfa_pair_group(const Plan& pl, const cd* permuted, const int4* quads, int G, cd* Y, double* corr, size_t stride,
                           const int* zero_rows, hipStream_t on) {
  const Pfa& f = pl.pfa;
  PAL_TRY(pfa_rows(pl, permuted, quads, G, Y, on));
  {
    ProfScope ps(this, "k_pfa_cols", on);
    const unsigned nblk = unsigned(f.n2 + 63) / 64;
    k_pfa_cols<kPfaTC, kPfaUnr><<<dim3(unsigned(G) * nblk, unsigned(f.nch + 3) / 4), dim3(256), 0, on>>>(Y, corr, stride, f.n1,
                                                                                                     f.n2, G, f.nch, f.T, zero_rows);
    PAL_HIP(hipGetLastError());
  }
  return PAL_OK;
}

// Forward spectra of `rows` real frames through the prime-factor cut (pfa_forward.h): two frames per transform, column
// DFTs into the workspace, Rader rows into the SP layout.  Applies to plans with Rader rows and frames that leave the
// upper half of the n points empty (len <= N2 (h + 1)).
bool Engine::pfa_forward_applies(const Plan& pl, int len) const {
  const Pfa& f = pl.pfa;
  return pfa_forward && f.on() && f.rader && f.nch >= 1 && (len - 1) / f.n2 <= (f.n1 - 1) / 2;
}

int Engine::pfa_forward_spectra(Plan& pl, const double* frames, size_t frame_stride, int rows, int len, cd* spectra) {
  const Pfa& f = pl.pfa;
  void* wsp = nullptr;
  PAL_TRY(scratch(0, size_t(chunk) * size_t(pl.n) * sizeof(cd), &wsp));
  cd* Y = static_cast<cd*>(wsp);
  for (int r0 = 0; r0 < rows; r0 += 2 * chunk) {
    const int R = rows - r0 < 2 * chunk ? rows - r0 : 2 * chunk;   // frames of this group, two per transform
    const int G = (R + 1) / 2;
    {
      ProfScope ps(this, "k_pfa_fwd_cols", stream);
      const PfaFwdColsArgs a{frames + size_t(r0) * frame_stride, frame_stride, len, R, Y, f.T, f.n1, f.n2, G, f.nch};
      const unsigned nblk = unsigned(f.n2 + 63) / 64;
      k_pfa_fwd_cols<kPfaTC, kPfaUnr><<<dim3(unsigned(G) * nblk, unsigned(f.nch + 3) / 4), dim3(256), 0, stream>>>(a);
      PAL_HIP(hipGetLastError());
    }
    {
      ProfScope ps(this, "k_pfa_fwd_rows_rader<11,9,10>", stream);
      const PfaFwdRowsArgs a{Y, spectra + size_t(r0) * pl.spec_stride(), f.rd_bhat_f, f.r1,
                             f.rd_qidx, f.rowtab, f.n1, f.n2, f.rows(), G, R, 1.0f / float(f.n1)};
      k_pfa_fwd_rows_rader<11, 9, 10><<<dim3(unsigned(G) * unsigned(f.rows())), dim3(256), 0, stream>>>(a);
      PAL_HIP(hipGetLastError());
    }
  }
  return PAL_OK;
}

// the fused column pass applies when one workgroup covers every output index (nch <= 4 chunks of kPfaTC: N1 <= 89)
bool Engine::pfa_can_fuse(const Plan& pl) const {
  const Pfa& f = pl.pfa;
  return fuse_peaks && f.on() && f.nch >= 1 && f.nch <= 4 && f.n2 >= 3;
}

// row pass, column pass + streaming statistics (every column block publishes a histogram window around its median),
// finish: the peak selection of one launch group without a pivot launch, without bracket lists and without the separate
// read of its correlation rows (pfa_cols_stats.h)
int Engine::pfa_pair_group_fused(const Plan& pl, const cd* permuted, const int4* quads, int G, int rows, cd* Y, double* corr, size_t stride,
                                 const int* zero_rows, const pal_phat_params& prm, int n2, pal_pair_record* table, int32_t* ksel_multi,
                                 hipStream_t on) {
  const Pfa& f = pl.pfa;
  // short column DFTs (one chunk: N1 <= 23): the four wavefronts of a workgroup take four neighbouring strips
  const bool shortcols = f.nch <= 1;
  const int per_blk = shortcols ? kColsOwn * 4 : kColsOwn;
  const int nblk = (f.n2 + per_blk - 1) / per_blk;
  PeakArgs a;
  PAL_TRY(peaks_setup(corr, stride, rows, pl.n, n2, prm, nblk, f.n2, on, a));
  PAL_TRY(pfa_rows(pl, permuted, quads, G, Y, on));
  {
    ProfScope ps(this, "k_pfa_cols_stats", on);
    const dim3 grid(unsigned(G) * unsigned(nblk));
    const bool full = (f.n1 - 1) / 2 == f.nch * kPfaTC;          // every chunk index exists (N1 = 89: 44 = 4 x 11)
    const bool adaptive = a.method > 0;
    const int nw = f.nch == 2 ? 2 : 4;                         // wavefronts per workgroup = chunks (three chunks: the fourth wavefront idles) or strips
#define PAL_COLS_STATS(AD, FU, NW, ST) k_pfa_cols_stats<kPfaTC, kPfaUnr, AD, FU, NW, ST><<<grid, dim3(64 * NW), 0, on>>>(Y, corr, stride, f.n1, f.n2, G, f.nch, f.T, zero_rows, a, rows)
#define PAL_COLS_STATS_NW(AD, FU) do { if (shortcols) PAL_COLS_STATS(AD, false, 4, true); else if (nw == 2) PAL_COLS_STATS(AD, FU, 2, false); else PAL_COLS_STATS(AD, FU, 4, false); } while (0)
    if (f.r89 && full && nw == 4 && !shortcols) {
      const Rader89Tab* tab = static_cast<const Rader89Tab*>(f.r89);
      if (adaptive) k_pfa_cols_stats<kPfaTC, kPfaUnr, true, true, 4, false, true><<<grid, dim3(256), 0, on>>>(Y, corr, stride, f.n1, f.n2, G, f.nch, f.T, zero_rows, a, rows, tab);
      else k_pfa_cols_stats<kPfaTC, kPfaUnr, false, true, 4, false, true><<<grid, dim3(256), 0, on>>>(Y, corr, stride, f.n1, f.n2, G, f.nch, f.T, zero_rows, a, rows, tab);
    } else if (adaptive) { if (full) PAL_COLS_STATS_NW(true, true); else PAL_COLS_STATS_NW(true, false); }
    else { if (full) PAL_COLS_STATS_NW(false, true); else PAL_COLS_STATS_NW(false, false); }
#undef PAL_COLS_STATS_NW
#undef PAL_COLS_STATS
    PAL_HIP(hipGetLastError());
  }
  return peaks_finish(a, rows, table, ksel_multi, on);
}


// ---- the column pass that finishes the rows itself (pfa_cols_fin.h, pfa_fin_lean.h): no correlation rows in HBM, no finish launch ----
// One peak per row (main.py:204), the caller does not ask for `corr`, and the grid's rows have at least 256 columns.  Which column
// forms take it is decided by measurement over the sync-padded lengths (see below); the other plans keep their rows in HBM - with
// the same per-wavefront statistics where the grid is large enough (pfa_can_lean_store), else with round 2's statistics.
bool Engine::pfa_can_finish(const Plan& pl, const pal_phat_params& prm) const {
  const Pfa& f = pl.pfa;
  const bool strips = fin_strips;
  // Measured over L = 44100 ... 44299 with the per-wavefront statistics of pfa_fin_lean.h (profiles/r03_c_length_sweep_dense_fin.csv
  // against ..._default.csv): the pass wins with Rader-89 columns (+10 %), with two or four chunks of output indices (+3 ... +12 %, +4 %)
  // and with short columns beside row tiles of up to 8192 points (+8 %); three chunks leave the fourth wavefront idle (-2 %), and
  // beside the 16384-point row tiles it is a wash
  bool cols_ok = f.r89 != nullptr || (f.nch >= 2 && f.nch <= 4) || (f.nch <= 1 && f.lm <= 13);
  // ... on grids of at least twelve column blocks per transform (N2 >= 683; strips: N2 >= 2729): with fewer, a launch group is one or
  // two rounds of blocks and the pass is the sum of one block's latencies - the stream chain's lengths (n = 24 000 ... 24 500:
  // 59 x 407, seven blocks) ran 534 frames/s with the pass and 579 without
  const int nblk = (f.n2 + (f.nch <= 1 ? 4 : 1) * kColsOwn - 1) / ((f.nch <= 1 ? 4 : 1) * kColsOwn);
  if (!f.r89 && nblk < 12) cols_ok = false;
  if (fin_dense >= 0) cols_ok = f.r89 != nullptr || fin_dense != 0;      // PAL_FIN_DENSE=1 / 0: every / no dense column DFT
  if (f.nch <= 1 && strips) cols_ok = true;
  // five and six chunks (N1 up to 133: C5's 103 x 233; five- / six-wavefront blocks, no histogram form): opt-in, PAL_FIN_WIDE=1.
  // Correct, but C5 runs 2.53 against 2.58 M pairs/s with it: 960 blocks are one round of the machine, every wavefront is in the
  // same phase at the same time and the pass (160 us) is the sum of its latencies, where the separate launches (70 + 18 + 31 + 32) overlap
  const bool nohist = prm.threshold_method > 0 || (prm.threshold_multiplier >= 0 && prm.threshold_multiplier <= 2.0 && !fin_hist);
  if (f.nch >= 5) {
    return fin_cols && fin_wide && fuse_peaks && f.on() && f.nch <= 6 && nohist && prm.num_peaks == 1 && f.n2 >= 124;
  }
  return fin_cols && pfa_can_fuse(pl) && prm.num_peaks == 1 && f.n2 >= 256 && cols_ok;
}

// The blocks of a finishing pass wait for their siblings (pfa_cols_fin.h).  ONE such launch is deadlock-free (its workgroups are
// dispatched in order and siblings are adjacent), but two of them on different streams can fill every workgroup slot of the
// device with blocks that wait for siblings which then find no slot (seen with the four-step source: 34 blocks per transform,
// two per CU, three streams).  The kernel's waits are bounded and a block that gives up only sends its rows through the
// stored-row path, so this costs time, never correctness.  PAL_FIN_SERIAL=1 runs the finishing launches one at a time
// instead (each waits for the previous one's end, wherever that ran): no such stall can happen, 10 % slower on the metric run.
int Engine::fin_serialize(hipStream_t on) {
  if (fin_serial && fin_pending) PAL_HIP(hipStreamWaitEvent(on, ev_fin, 0));
  return PAL_OK;
}
int Engine::fin_done(hipStream_t on) {
  if (!fin_serial) return PAL_OK;
  PAL_HIP(hipEventRecord(ev_fin, on));
  fin_pending = true;
  return PAL_OK;
}

// arguments and scratch of one launch of the finishing pass: a grid of `grid_rows` x `grid_cols` samples per row, nblk blocks per transform
int Engine::fin_setup(const Plan& pl, int rows, int nblk, int grid_rows, int grid_cols, const pal_phat_params& prm, int n2, pal_pair_record* table,
                      int* need, int slot, hipStream_t on, PeakArgs& a, FinArgs& fa, unsigned& nwg, int G) {
  const int n = pl.n;
  PAL_TRY(peaks_setup(nullptr, 0, rows, n, n2, prm, nblk, grid_cols, on, a));
  // per-stream scratch of the finishing pass: [done words G x blocks | emax | parts | edge]
  const int Gmax = pair_group(n);
  // (`done` words and FinPartial entries: room for one per WAVEFRONT of a block, pfa_fin_lean.h)
  const size_t off_emax = (size_t(Gmax) * nblk * 6 * sizeof(unsigned) + 127) & ~size_t(127);
  const size_t off_parts = (off_emax + size_t(2 * Gmax) * nblk * 12 * sizeof(double) + 127) & ~size_t(127);
  const size_t off_edge = (off_parts + size_t(2 * Gmax) * nblk * 6 * sizeof(FinPartial) + 127) & ~size_t(127);
  const size_t total = off_edge + size_t(2 * Gmax) * 4 * grid_rows * sizeof(double);
  void* sp = nullptr;
  PAL_TRY(scratch(16 + slot, total, &sp));
  char* base = static_cast<char*>(sp);
  int* status = nullptr;
  {
    void* stp = nullptr;
    PAL_TRY(scratch(7, 64, &stp));
    status = static_cast<int*>(stp);
  }
  fa.table = table;
  fa.need = need;
  fa.done = reinterpret_cast<unsigned*>(base);
  // launch number of this stream's scratch: entries of earlier launches fail the comparison (no resets, no counters)
  if (ws_bytes[16 + slot] != fin_bytes[slot] || fin_epoch[slot] >= (1u << 20)) {     // new (zeroed) scratch, or the number would outgrow a double's integers
    PAL_HIP(hipMemsetAsync(sp, 0, total, on));
    fin_bytes[slot] = ws_bytes[16 + slot];
    fin_epoch[slot] = 0;
  }
  fa.epoch = ++fin_epoch[slot];
  fa.emax = reinterpret_cast<double*>(base + off_emax);
  fa.parts = reinterpret_cast<FinPartial*>(base + off_parts);
  fa.edge = reinterpret_cast<double*>(base + off_edge);
  fa.status = status;
  // the lag window |m - (n2 - 1)| / fs <= max_expected_delay (utils.py:163) as sample indices, with the reference's arithmetic
  fa.pw = 1;
  fa.corr = nullptr;
  fa.stride = 0;
  fa.store_rows = 0;
  fa.windowed = std::isnan(prm.max_expected_delay) ? 0 : 1;
  fa.win_lo = 1;
  fa.win_hi = n - 2;
  if (fa.windowed) {
    const double med = prm.max_expected_delay, fs = prm.fs;
    long long k = -1;
    if (med >= 0) {
      const double est = med * fs;
      k = est < double(n) ? (long long)est + 2 : (long long)n;
      while (k >= 0 && !(std::fabs(double(k) / fs) <= med)) --k;
    }
    if (k < 0) { fa.win_lo = 1; fa.win_hi = 0; }
    else {
      const long long lo = (long long)(n2 - 1) - k, hi = (long long)(n2 - 1) + k;
      fa.win_lo = int(lo < 1 ? 1 : lo);
      fa.win_hi = int(hi > n - 2 ? n - 2 : hi);
    }
  }
  const bool no_cheb = fin_hist;                               // diagnostics: histograms for every multiplier
  fa.cheb = a.method == 0 && prm.threshold_multiplier >= 0 && prm.threshold_multiplier <= 2.0 && !no_cheb ? 1 : 0;
  fa.stamps = nullptr;
  static const bool want_stamps = getenv("PAL_DEBUG_STAMPS") != nullptr;
  nwg = 8u * unsigned((G + 7) / 8) * unsigned(nblk);
  if (want_stamps) {
    void* st = nullptr;
    PAL_TRY(scratch(13, size_t(nwg) * 8 * sizeof(unsigned long long), &st));
    PAL_HIP(hipMemsetAsync(st, 0, size_t(nwg) * 8 * sizeof(unsigned long long), on));
    fa.stamps = static_cast<unsigned long long*>(st);
  }
  return PAL_OK;
}

// the last pass of the four-step chirp convolution finishes its rows itself where its columns fit one lane (register rows,
// M1 <= 24), one peak per row is asked for and the threshold needs no histograms ('adaptive', or 'median' with a multiplier in 0 .. 2)
bool Engine::fourstep_can_finish(const Plan& pl, const pal_phat_params& prm) const {
  const Conv& c = pl.inv;
  const bool off = !fin_four;                                  // opt-in (PAL_FIN_FOUR=1): measured 0.36 against 0.50 M pairs/s for the stored rows + statistics launches
  const bool nohist = prm.threshold_method > 0 || (prm.threshold_multiplier >= 0 && prm.threshold_multiplier <= 2.0 && !fin_hist);
  return fin_cols && !off && c.reg && c.M1() <= 24 && prm.num_peaks == 1 && nohist && pl.nout == pl.n;
}

int Engine::fourstep_pair_group_fin(const Plan& pl, const cd* W, int G, int rows, const int* zero_rows, const pal_phat_params& prm, int n2,
                                    pal_pair_record* table, int* need, int slot, hipStream_t on) {
  Engine* e = this;
  const Conv& c = pl.inv;
  const int N2 = 1 << c.l2, M1 = c.M1();
  const int nblk = ((N2 + kColsOwn - 1) / kColsOwn + 3) / 4;
  PeakArgs a;
  FinArgs fa;
  unsigned nwg = 0;
  PAL_TRY(fin_setup(pl, rows, nblk, M1, N2, prm, n2, table, need, slot, on, a, fa, nwg, G));
  {
    char name[48];
    snprintf(name, sizeof name, "k_colsreg_fin<%d>", M1);
    ProfScope ps(this, name, on);
    FinSrc src{W, nullptr, nullptr, c.twA, c.twB, pl.w};
    fa.pw = 4;                                                 // one FinPartial / `done` word per wavefront (pfa_fin_lean.h)
    PAL_TRY(fin_serialize(on));
    PAL_SWITCH_M1(M1, c.l2, k_pfa_cols_fin<kColsFourStep, MM, LR, false, false, 4><<<dim3(nwg), dim3(256), 0, on>>>(src, M1, N2, G, 1, nblk, zero_rows, a, fa, rows));
    PAL_HIP(hipGetLastError());
    PAL_TRY(fin_done(on));
  }
  return PAL_OK;
}

// Statistics of rows that are already in HBM, any route (k_rows_lean): one launch instead of pivots + stream + finish where one peak per
// row is asked for and the threshold needs no histograms; flagged rows are resolved at the end of the call.
bool Engine::rows_can_lean(const Plan& pl, const pal_phat_params& prm) const {
  const bool nohist = prm.threshold_method > 0 || (prm.threshold_multiplier >= 0 && prm.threshold_multiplier <= 2.0 && !fin_hist);
  // Measured: rows of 12 013 ... 24 013 samples +2 ... +8 % (C5 2.58 -> 2.87 M pairs/s: 66 us per group against 18 + 31 + 32), rows of
  // 88 201 ... 88 367 -3 ... +1 %, C4's 191 999 the same: long rows keep the three launches (their stream pass runs at the HBM rate)
  return rows_lean && fin_cols && prm.num_peaks == 1 && nohist && pl.nout == pl.n && pl.n >= 4096 && pl.n <= 50000;
}

int Engine::rows_lean_group(const Plan& pl, const double* corr, size_t stride, int G, int rows, const pal_phat_params& prm, int n2,
                            pal_pair_record* table, int* need, int slot, hipStream_t on) {
  constexpr int NS = 22;
  const int chunks = (pl.n + kColsOwn - 1) / kColsOwn;
  const int nblk = (chunks + 4 * NS - 1) / (4 * NS);
  PeakArgs a;
  FinArgs fa;
  unsigned nwg = 0;
  PAL_TRY(fin_setup(pl, rows, nblk, 1, kColsOwn, prm, n2, table, need, slot, on, a, fa, nwg, G));
  fa.pw = 4;
  fa.corr = const_cast<double*>(corr);
  fa.stride = stride;
  fa.store_rows = 0;
  ProfScope ps(this, "k_rows_lean", on);
  k_rows_lean<NS><<<dim3(nwg), dim3(256), 0, on>>>(corr, stride, G, nblk, a, fa, rows);
  PAL_HIP(hipGetLastError());
  return PAL_OK;
}

// The same pass with the correlation rows stored as well (the caller wants them, or the plan has no finishing form that pays):
// per-wavefront statistics, no sibling polls, the finisher reads the SNR window from the stored row; flagged rows are resolved at the
// end of the call like the finishing pass's.  Replaces pfa_cols_stats.h / the three statistics launches + k_peak_finish where one
// peak per row is asked for and the threshold needs no histograms.
bool Engine::pfa_can_lean_store(const Plan& pl, const pal_phat_params& prm) const {
  const Pfa& f = pl.pfa;
  const bool nohist = prm.threshold_method > 0 || (prm.threshold_multiplier >= 0 && prm.threshold_multiplier <= 2.0 && !fin_hist);
  // Several rounds of column blocks per launch group, or the pass is the sum of one block's latencies: C5 (103 x 233: four blocks per
  // transform, 960 per group, one round) measured 2.32 against 2.60 M pairs/s with it; C3 (7 x 6857: 28 blocks) 1.16 against 1.08,
  // C2 (17 x 5647: 23 blocks) 0.422 against 0.417
  const int nblk = (f.n2 + (f.nch <= 1 ? 4 : 1) * kColsOwn - 1) / ((f.nch <= 1 ? 4 : 1) * kColsOwn);
  // (five and six chunks, N1 = 91 ... 127 beside 700 - 970 columns: 0.63 - 0.66 against 0.69 - 0.71 M with it: they keep the three statistics launches)
  return lean_store && fin_cols && fuse_peaks && f.on() && f.nch >= 1 && f.nch <= 4 && nblk >= 12 && prm.num_peaks == 1 && nohist;
}

int Engine::pfa_pair_group_fin(const Plan& pl, const cd* permuted, const int4* quads, int G, int rows, cd* Y, const int* zero_rows,
                               const pal_phat_params& prm, int n2, pal_pair_record* table, int* need, int slot, hipStream_t on,
                               double* corr, size_t stride) {
  const Pfa& f = pl.pfa;
  const bool shortcols = f.nch <= 1;                           // short column DFTs (N1 <= 23): the four wavefronts of a block are four strips
  const int nblk = (f.n2 + (shortcols ? 4 : 1) * kColsOwn - 1) / ((shortcols ? 4 : 1) * kColsOwn);
  PeakArgs a;
  FinArgs fa;
  unsigned nwg = 0;
  PAL_TRY(fin_setup(pl, rows, nblk, f.n1, f.n2, prm, n2, table, need, slot, on, a, fa, nwg, G));
  fa.corr = corr;
  fa.stride = stride;
  fa.store_rows = corr ? 1 : 0;
  static const bool want_stamps = getenv("PAL_DEBUG_STAMPS") != nullptr;
  PAL_TRY(pfa_rows(pl, permuted, quads, G, Y, on));
  {
    ProfScope ps(this, corr ? "k_pfa_cols_lean" : "k_pfa_cols_fin", on);
    const dim3 grid(nwg);
    const bool full = (f.n1 - 1) / 2 == f.nch * kPfaTC;
    // histograms only where the bound sqrt(2 mean(x^2)) on the median cannot decide: multipliers above 2 (or negative)
    const bool hist = !(a.method > 0 || fa.cheb);
    const int nw = f.nch == 2 ? 2 : (f.nch == 3 && !hist ? 3 : (f.nch >= 5 ? f.nch : 4));   // (three chunks: three wavefronts where the statistics are per wavefront)
    if (f.nch >= 5 && hist) return fail(PAL_ERR_INTERNAL, "finishing column pass with %d chunks and histograms", f.nch);
    FinSrc src{Y, f.T, static_cast<const Rader89Tab*>(f.r89), nullptr, nullptr, nullptr};
    PAL_TRY(fin_serialize(on));
#define PAL_COLS_FIN(MODE, HI, FU, NW) k_pfa_cols_fin<MODE, kPfaTC, kPfaUnr, HI, FU, NW><<<grid, dim3(64 * NW), 0, on>>>(src, f.n1, f.n2, G, f.nch, nblk, zero_rows, a, fa, rows)
    if (!hist) fa.pw = shortcols ? 4 : nw;                    // one FinPartial / `done` word per wavefront (pfa_fin_lean.h)
    if (shortcols) { if (hist) PAL_COLS_FIN(kColsStrips, true, false, 4); else PAL_COLS_FIN(kColsStrips, false, false, 4); }
    else if (f.r89 && full && nw == 4) { if (hist) PAL_COLS_FIN(kColsRader89, true, true, 4); else PAL_COLS_FIN(kColsRader89, false, true, 4); }
    else if (nw == 2) {
      if (hist) { if (full) PAL_COLS_FIN(kColsDense, true, true, 2); else PAL_COLS_FIN(kColsDense, true, false, 2); }
      else { if (full) PAL_COLS_FIN(kColsDense, false, true, 2); else PAL_COLS_FIN(kColsDense, false, false, 2); }
    } else if (nw == 3) {
      if (full) PAL_COLS_FIN(kColsDense, false, true, 3); else PAL_COLS_FIN(kColsDense, false, false, 3);
    } else if (nw == 5) {
      if (full) PAL_COLS_FIN(kColsDense, false, true, 5); else PAL_COLS_FIN(kColsDense, false, false, 5);
    } else if (nw == 6) {
      if (full) PAL_COLS_FIN(kColsDense, false, true, 6); else PAL_COLS_FIN(kColsDense, false, false, 6);
    } else {
      if (hist) { if (full) PAL_COLS_FIN(kColsDense, true, true, 4); else PAL_COLS_FIN(kColsDense, true, false, 4); }
      else { if (full) PAL_COLS_FIN(kColsDense, false, true, 4); else PAL_COLS_FIN(kColsDense, false, false, 4); }
    }
#undef PAL_COLS_FIN
    PAL_HIP(hipGetLastError());
    PAL_TRY(fin_done(on));
  }
  if (want_stamps) {                                           // diagnostics: phase times of this launch (synchronises)
    std::vector<unsigned long long> hst(size_t(nwg) * 8);
    PAL_HIP(hipStreamSynchronize(on));
    PAL_HIP(hipMemcpy(hst.data(), fa.stamps, hst.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    const char* names[6] = {"accumulate", "pass A + windows", "pass B + publish", "wait for siblings", "window sums", "finish (last block)"};
    unsigned long long t0 = ~0ull, t1 = 0;
    for (unsigned b = 0; b < nwg; ++b)
      if (hst[size_t(b) * 8]) { t0 = std::min(t0, hst[size_t(b) * 8]); for (int k = 0; k < 7; ++k) t1 = std::max(t1, hst[size_t(b) * 8 + k]); }
    fprintf(stderr, "[pal] k_pfa_cols_fin: %u workgroups, first start to last stamp %.1f us\n", nwg, double(t1 - t0) / 100.0);
    {
      double life = 0;
      std::vector<std::pair<unsigned long long, int>> ev;
      for (unsigned b = 0; b < nwg; ++b) {
        if (!hst[size_t(b) * 8]) continue;
        unsigned long long e = 0;
        for (int k = 0; k < 7; ++k) e = std::max(e, hst[size_t(b) * 8 + k]);
        life += double(e - hst[size_t(b) * 8]) / 100.0;
        ev.push_back({hst[size_t(b) * 8], 1});
        ev.push_back({e, -1});
      }
      std::sort(ev.begin(), ev.end());
      int cur = 0, peak = 0;
      for (auto& x : ev) { cur += x.second; peak = std::max(peak, cur); }
      fprintf(stderr, "[pal]   resident workgroups: average %.0f, peak %d; mean lifetime %.1f us\n", life / (double(t1 - t0) / 100.0), peak, life / nwg);
      // when did the workgroups start (index order)?
      for (unsigned b : {0u, 1u, 8u, 127u, 128u, 767u, 768u, 769u, 1535u, 1536u, 3071u, 3839u})
        if (b < nwg) fprintf(stderr, "[pal]   workgroup %4u started at %7.1f us, ended at %7.1f\n", b, double(hst[size_t(b) * 8] - t0) / 100.0,
                             double(std::max(hst[size_t(b) * 8 + 5], hst[size_t(b) * 8 + 6]) - t0) / 100.0);
    }
    std::vector<double> d;
    for (int ph = 0; ph < 6; ++ph) {
      d.clear();
      for (unsigned b = 0; b < nwg; ++b)
        if (hst[size_t(b) * 8 + ph] && hst[size_t(b) * 8 + ph + 1]) d.push_back(double(hst[size_t(b) * 8 + ph + 1] - hst[size_t(b) * 8 + ph]) / 100.0);
      if (d.empty()) continue;
      std::sort(d.begin(), d.end());
      fprintf(stderr, "[pal]   %-20s n %5zu  median %7.2f us  p90 %7.2f us  max %7.2f us\n", names[ph], d.size(), d[d.size() / 2], d[d.size() * 9 / 10], d.back());
    }
  }
  return PAL_OK;
}

}  // namespace pal
