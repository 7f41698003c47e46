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
// (u1 = N2^-1 mod N1, u2 = N1^-1 mod N2).  The N2-point DFTs are chirp convolutions of length M = 2^lm >= 2 N2 - 1
// that live entirely in LDS (forward FFT, multiply by the chirp spectrum, inverse FFT - two 2048-point tiles per
// workgroup), and the N1-point DFTs are real cos / sin contractions whose coefficients are wave-uniform and
// come through the scalar cache.  A transform touches HBM twice (16 B/point each way) instead of three times
// at twice the points, and the output index m = m2 + N2 t is the natural one, so the correlation rows are
// written in coalesced runs.
//
// As in bluestein.hip one complex transform carries two mic pairs (real part = pair p, imaginary part = pair q)
// and the whitened cross spectrum R = S_a conj(S_b) / (|.| + 1e-10) (utils.py:116-117) is built inside the
// first FFT stage.  The mic spectra are kept in the permuted layout SP[mic][k1][k2], k1 <= (N1-1)/2: row N1-k1
// is row k1 reversed and conjugated (Hermitian symmetry), so one workgroup serves both rows from the same
// loads - tile 0 transforms x[k1, e], tile 1 the reversed row z[e] = x[N1-k1, -e], whose DFT is the reversed DFT.
#include <cmath>
#include <vector>

#include "conv_kernels.h"

namespace pal {

constexpr int kPfaTC = 22;   // accumulator pairs per lane of the column pass

// ------------------------------------------------------------------ permuted spectra
// SP[row][k1][k2] = full Hermitian-extended spectrum at k = (e1 k1 + e2 k2) mod n, for k1 < NR
__global__ void k_pfa_permute(const cd* __restrict__ S, cd* __restrict__ SP, int n, int H, int NR, int N2,
                              long long e1, long long e2) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= NR * N2) return;
  const int k1 = idx / N2, k2 = idx - k1 * N2;
  const long long k = (e1 * k1 + e2 * k2) % n;
  const cd* row = S + size_t(blockIdx.y) * H;
  cd v;
  if (k < H) v = row[k];
  else v = cconj(row[n - k]);
  SP[(size_t(blockIdx.y) * NR + k1) * N2 + k2] = v;
}

// ------------------------------------------------------------------ stage functors of the row pass
template <int LM> struct PfaIn {          // first stage of the forward FFT: whitened pair values times the chirp
  static constexpr bool kLds = false;
  const cd *sa, *sb, *sc, *sd;            // row k1 of the four mic spectra (sc = sa / sd = sb when unused)
  const cd* b;
  int N2;
  bool second, share;
  __device__ cd operator()(int t, int e) const {
    constexpr int HALF = 1 << (LM - 1);   // N2 <= M/2: the upper half of every tile is zero padding
    if (e >= HALF) return mk(0, 0);
    const int ee = e < N2 ? e : N2 - 1;
    const cd a = sa[ee];
    const cd r1 = whiten(a, sb[ee]);
    cd r2 = mk(0, 0);
    if (second) r2 = whiten(share ? a : sc[ee], sd[ee]);
    // tile 0: R^p + i R^q at (k1, e);  tile 1: conj(R^p) + i conj(R^q) = the reversed row N1 - k1
    const cd x = t == 0 ? mk(r1.x - r2.y, r1.y + r2.x) : mk(r1.x + r2.y, r2.x - r1.y);
    const cd y = cmul(x, b[ee]);
    return e < N2 ? y : mk(0, 0);
  }
};

template <int LM> struct PfaHhatToLds {   // last stage of the forward FFT: times the chirp spectrum, into LDS
  static constexpr bool kLds = true;
  cd* data;
  const cd* hh;
  __device__ void operator()(int t, int e, cd v) const { data[lds_addr<LM, false, 2>(t, e)] = cmul(v, hh[e]); }
};

template <int LM> struct PfaOut {         // last stage of the inverse FFT: chirp, column twiddle, store Y[row][m2]
  static constexpr bool kLds = false;
  cd* Yg;                                 // Y of this transform: [N1][N2]
  const cd *b, *r1;
  int N1, N2, k1, uk0, uk1;               // uk_t = u1 * row_t mod N1
  float inv;
  __device__ void operator()(int t, int e, cd v) const {
    constexpr int HALF = 1 << (LM - 1);
    if (e >= HALF) return;
    const int ee = e < N2 ? e : N2 - 1;
    const int m2 = t == 0 ? ee : (ee ? N2 - ee : 0);
    const int row = t == 0 ? k1 : N1 - k1;
    const unsigned x = unsigned(t == 0 ? uk0 : uk1) * unsigned(m2);     // < 2^24: exact in float
    const unsigned q = unsigned(float(x) * inv);
    int r = int(x) - int(q) * N1;
    if (r < 0) r += N1;
    if (r >= N1) r -= N1;
    const cd z = cmulc(cmul(v, b[ee]), r1[r]);                           // r1 holds exp(-2 pi i q / N1)
    if (e < N2 && !(t == 1 && k1 == 0)) Yg[size_t(row) * N2 + m2] = z;
  }
};

template <int LM> struct PfaChirpIn {     // chirp kernel of the convolution: h[d mod M] = conj(b[|d|]), |d| < N2
  static constexpr bool kLds = false;
  const cd* b;
  int N2;
  __device__ cd operator()(int, int e) const {
    constexpr int M = 1 << LM;
    if (e < N2) return cconj(b[e]);
    if (M - e < N2) return cconj(b[M - e]);
    return mk(0, 0);
  }
};

template <int LM> struct PfaScaledOut {
  static constexpr bool kLds = false;
  cd* out;
  double scale;
  __device__ void operator()(int t, int e, cd v) const { if (t == 0) out[e] = cscale(v, scale); }
};

template <int LM> struct PfaLds {         // LDS sizes of the row pass
  static constexpr bool kCompact = LM >= 11;
  static constexpr int kM = 1 << LM, kLanes = 2 * kM / 16;
  static constexpr int kTw = kCompact ? stage_twc_size(LM) : stage_tw_size(LM);
};

template <int LM>
__global__ __launch_bounds__(PfaLds<LM>::kLanes) void k_pfa_hhat(const cd* __restrict__ b, int N2, cd* __restrict__ hhat,
                                                                 double scale, const cd* __restrict__ tws) {
  using L = PfaLds<LM>;
  __shared__ cd data[2 * L::kM];
  __shared__ cd tw[L::kTw];
  const int tid = threadIdx.x;
  for (int i = tid; i < L::kTw; i += L::kLanes) tw[i] = tws[i];
  wg_fft<LM, false, false, 2, L::kCompact>(data, tw, tid, PfaChirpIn<LM>{b, N2}, PfaScaledOut<LM>{hhat, scale});
}

struct PfaRowsArgs {
  const cd* SP;        // permuted spectra [mic][NR][N2]
  const int4* quad;    // mic rows (a, b) of pair p and (c, d) of pair q; c < 0: no second pair
  cd* Y;               // [G][N1][N2]
  const cd *b, *hhat, *r1, *tws;
  int N1, N2, NR, G, u1;
  float inv;
};

// grid = G * NR workgroups, transform fastest so that neighbours share the tables
template <int LM>
__global__ __launch_bounds__(PfaLds<LM>::kLanes) void k_pfa_rows(PfaRowsArgs a) {
  using L = PfaLds<LM>;
  __shared__ cd data[2 * L::kM];
  __shared__ cd tw[L::kTw];
  const int tid = threadIdx.x;
  const int g = blockIdx.x % a.G, k1 = blockIdx.x / a.G;
  for (int i = tid; i < L::kTw; i += L::kLanes) tw[i] = a.tws[i];
  const int4 q = a.quad[g];
  const size_t mic = size_t(a.NR) * a.N2, off = size_t(k1) * a.N2;
  const bool second = q.z >= 0;
  const cd* sa = a.SP + size_t(q.x) * mic + off;
  const cd* sb = a.SP + size_t(q.y) * mic + off;
  const PfaIn<LM> in{sa, sb, second ? a.SP + size_t(q.z) * mic + off : sa, second ? a.SP + size_t(q.w) * mic + off : sb,
                     a.b, a.N2, second, second && q.z == q.x};
  wg_fft<LM, false, false, 2, L::kCompact>(data, tw, tid, in, PfaHhatToLds<LM>{data, a.hhat});
  const int kr = k1 ? a.N1 - k1 : 0;
  const PfaOut<LM> out{a.Y + size_t(g) * a.N1 * a.N2, a.b, a.r1, a.N1, a.N2, k1, (a.u1 * k1) % a.N1, (a.u1 * kr) % a.N1, a.inv};
  wg_fft<LM, false, true, 2, L::kCompact>(data, tw, tid, LdsTile<LM, false, 2>{data}, out);
}

// ------------------------------------------------------------------ column pass
// One lane per column m2 (64 consecutive columns per wavefront: coalesced 1 KB loads of Y, 512 B stores of a
// correlation row).  The four wavefronts of a workgroup split the work by role (pair p = real parts, pair q =
// imaginary parts) and by chunk of kPfaTC output indices t; with E_j = Y_j + Y_{N1-j}, O_j = Y_j - Y_{N1-j}:
//   p:  c[t] = Re Y_0 + sum_j cos(j t) Re E_j - sum_j sin(j t) Im O_j,   c[N1-t] = the same with + sin
//   q:  c[t] = Im Y_0 + sum_j cos(j t) Im E_j + sum_j sin(j t) Re O_j,   c[N1-t] = the same with - sin
// The cos / sin rows are wave-uniform: T[(j-1)][chunk][cos | sin][tt] is read through scalar loads.
template <int TC>
__global__ __launch_bounds__(256) void k_pfa_cols(const cd* __restrict__ Y, double* __restrict__ corr, size_t stride,
                                                  int N1, int N2, int G, int nch, const double* __restrict__ T) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(int(threadIdx.x) >> 6);
  const int role = wave & 1, ch = int(blockIdx.y) * 2 + (wave >> 1);
  if (ch >= nch) return;
  const int g = blockIdx.x % G, cb = blockIdx.x / G;
  const int m2 = cb * 64 + lane;
  const bool live = m2 < N2;
  const cd* Yg = Y + size_t(g) * N1 * N2 + (live ? m2 : N2 - 1);
  const int h = (N1 - 1) / 2;
  double accC[TC], accS[TC];
#pragma unroll
  for (int tt = 0; tt < TC; ++tt) accC[tt] = accS[tt] = 0.0;
  double sumE = 0.0;
  const double* Tj = T + size_t(ch) * 2 * TC;
  const size_t tstep = size_t(nch) * 2 * TC;
  for (int j = 1; j <= h; ++j, Tj += tstep) {
    const cd yj = Yg[size_t(j) * N2], ym = Yg[size_t(N1 - j) * N2];
    const double a = role ? yj.y + ym.y : yj.x + ym.x;
    const double b = role ? yj.x - ym.x : yj.y - ym.y;
    sumE += a;
#pragma unroll
    for (int tt = 0; tt < TC; ++tt) {
      accC[tt] = __builtin_fma(Tj[tt], a, accC[tt]);
      accS[tt] = __builtin_fma(Tj[TC + tt], b, accS[tt]);
    }
  }
  const cd y0 = Yg[0];
  const double base = role ? y0.y : y0.x;
  if (!live) return;
  double* out = corr + size_t(2 * g + role) * stride + m2;
  if (ch == 0) out[0] = base + sumE;
#pragma unroll
  for (int tt = 0; tt < TC; ++tt) {
    const int t = ch * TC + tt + 1;
    if (t <= h) {
      const double s = role ? accS[tt] : -accS[tt];
      out[size_t(N2) * t] = base + accC[tt] + s;
      out[size_t(N2) * (N1 - t)] = base + accC[tt] - s;
    }
  }
}

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
  f = Pfa();
}

#define PAL_SWITCH_LM(lm, ...)                                     \
  switch (lm) {                                                    \
    case 10: { constexpr int LM = 10; __VA_ARGS__; } break;        \
    case 11: { constexpr int LM = 11; __VA_ARGS__; } break;        \
    case 12: { constexpr int LM = 12; __VA_ARGS__; } break;        \
    default: return fail(PAL_ERR_INTERNAL, "prime-factor plan with log2 M = %d", lm); \
  }

int Engine::build_pfa(Plan& pl) {
  pl.pfa = Pfa();
  const long long n = pl.n;
  if (!allow_pfa || pl.nout != pl.n || n < 3 || (n & 1) == 0) return PAL_OK;
  // best coprime split: fewest tile points NR * 2M, then the smaller N1
  long long best = -1;
  int bn1 = 0, bn2 = 0, blm = 0;
  for (long long d = 1; d <= 127 && d <= n; d += 2) {
    if (n % d) continue;
    const long long r = n / d;
    if (r > 2048 || gcd_ll(d, r) != 1) continue;
    int lm = 10;
    while ((1ll << lm) < 2 * r - 1) ++lm;
    const long long cost = ((d + 1) / 2) * (2ll << lm);
    if (best < 0 || cost < best) { best = cost; bn1 = int(d); bn2 = int(r); blm = lm; }
  }
  if (best < 0 || size_t(best) > pl.inv.M() + pl.inv.M() / 2) return PAL_OK;   // the four-step route is no worse
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
  const cd* tws = blm >= 11 ? stage_table_compact(blm) : stage_table(blm);
  if (!tws) return fail(PAL_ERR_NOMEM, "twiddle tables");
  PAL_HIP(hipMalloc(&f.b, sizeof(cd) * bn2));
  PAL_HIP(hipMalloc(&f.hhat, sizeof(cd) << blm));
  PAL_HIP(hipMalloc(&f.r1, sizeof(cd) * bn1));
  k_make_chirp<<<dim3((bn2 + 255) / 256), dim3(256), 0, stream>>>(f.b, bn2, int(u2));
  k_make_roots<<<dim3((bn1 + 255) / 256), dim3(256), 0, stream>>>(f.r1, bn1, double(bn1));
  const double scale = 1.0 / (double(1 << blm) * double(n));   // inverse FFT_M and numpy.fft.ifft's 1/n
  PAL_SWITCH_LM(blm, k_pfa_hhat<LM><<<dim3(1), dim3(PfaLds<LM>::kLanes), 0, stream>>>(f.b, bn2, f.hhat, scale, tws));
  PAL_HIP(hipGetLastError());
  // cos / sin of 2 pi j t / N1 with the argument reduced exactly (j t mod N1) before the long-double evaluation
  std::vector<double> tab(size_t(h > 0 ? h : 1) * f.nch * 2 * kPfaTC, 0.0);
  const long double two_pi = 6.283185307179586476925286766559005768L;
  for (int j = 1; j <= h; ++j)
    for (int t = 1; t <= h; ++t) {
      const int ch = (t - 1) / kPfaTC, tt = (t - 1) % kPfaTC;
      const long double ang = two_pi * (long double)((long long)j * t % bn1) / (long double)bn1;
      double* row = &tab[(size_t(j - 1) * f.nch + ch) * 2 * kPfaTC];
      row[tt] = double(cosl(ang));
      row[kPfaTC + tt] = double(sinl(ang));
    }
  PAL_HIP(hipMalloc(&f.T, tab.size() * sizeof(double)));
  PAL_HIP(hipMemcpyAsync(f.T, tab.data(), tab.size() * sizeof(double), hipMemcpyHostToDevice, stream));
  PAL_HIP(hipStreamSynchronize(stream));       // `tab` is host memory
  pl.pfa = f;
  return PAL_OK;
}

int Engine::pfa_permute(const Plan& pl, const cd* spectra, int nspec, cd* out, hipStream_t on) {
  const Pfa& f = pl.pfa;
  ProfScope ps(this, "k_pfa_permute", on);
  const int count = f.rows() * f.n2;
  k_pfa_permute<<<dim3((count + 255) / 256, nspec), dim3(256), 0, on>>>(spectra, out, pl.n, pl.H, f.rows(), f.n2, f.e1, f.e2);
  return check(hipGetLastError(), "k_pfa_permute");
}

int Engine::pfa_pair_group(const Plan& pl, const cd* permuted, const int4* quads, int G, cd* Y, double* corr, size_t stride,
                           hipStream_t on) {
  const Pfa& f = pl.pfa;
  const cd* tws = f.lm >= 11 ? stage_table_compact(f.lm) : stage_table(f.lm);
  {
    char name[48];
    snprintf(name, sizeof name, "k_pfa_rows<%d>", f.lm);
    ProfScope ps(this, name, on);
    PfaRowsArgs a{permuted, quads, Y, f.b, f.hhat, f.r1, tws, f.n1, f.n2, f.rows(), G, f.u1, 1.0f / float(f.n1)};
    const unsigned grid = unsigned(G) * unsigned(f.rows());
    PAL_SWITCH_LM(f.lm, k_pfa_rows<LM><<<dim3(grid), dim3(PfaLds<LM>::kLanes), 0, on>>>(a));
    PAL_HIP(hipGetLastError());
  }
  {
    ProfScope ps(this, "k_pfa_cols", on);
    const unsigned nblk = unsigned(f.n2 + 63) / 64;
    k_pfa_cols<kPfaTC><<<dim3(unsigned(G) * nblk, unsigned(f.nch + 1) / 2), dim3(256), 0, on>>>(Y, corr, stride, f.n1, f.n2, G,
                                                                                           f.nch, f.T);
    PAL_HIP(hipGetLastError());
  }
  return PAL_OK;
}

}  // namespace pal
