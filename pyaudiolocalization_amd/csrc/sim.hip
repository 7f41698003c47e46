// sim.hip - second batched HIP path: image-source multipath synthesis, prefilter, synchronisation.
//
//   simulate  : main.py:103-123 (sum over paths of att * fractional_delay(base)), fused per mic as
//               X(f) * sum_p g_p exp(-j 2 pi f tau_p) -> one exact-length inverse DFT(2N) per mic
//               (SURVEY Q10), two mics per complex transform; fade window of
//               signal_processing.py:75-79, trim (main.py:119-120), normalize_signal +
//               dynamic_range_compression (signal_processing.py:82-94).
//   filtfilt  : scipy.signal.filtfilt defaults as called by noise_reduction
//               (signal_processing.py:127-134): odd extension, lfilter_zi state, DF2T recurrence in
//               scipy's multiply-then-add order.  This translation unit is built with
//               -ffp-contract=off so the recurrence is bit-compatible with the CPU path.
//   wiener3   : scipy.signal.wiener(x) defaults (signal_processing.py:136).
//   xcorr     : scipy.signal.correlate(sig, reference, 'full') + argmax|.| of
//               synchronize_signals_improved (utils.py:418-427) as a power-of-two circular
//               convolution with the reversed reference (any length >= 2N-1 is exact here).
#include <cmath>

#include "conv_kernels.h"
#include "reduce.h"

namespace pal {

namespace {

// ------------------------------------------------------------------ multipath synthesis
struct SimLoader {
  static constexpr const char* kName = "SimLoader";
  const cd* X;            // base spectra [bases][H], H = N + 1, transform length n = 2N
  const double* delays;   // [rows][K] seconds
  const double* gains;    // [rows][K]
  int K, rows, rows_per_base, N, row0;   // row0: first row of this launch group
  double val;             // 1.0 / (2N * (1/fs)): numpy.fft.fftfreq spacing (signal_processing.py:70)
  const cd* w;
  __device__ cd term(int r, unsigned kk, double f, bool nyquist, bool mirror) const {
    r += row0;
    if (r >= rows) return mk(0, 0);
    const cd x = X[size_t(r / rows_per_base) * (N + 1) + kk];
    double hx = 0, hy = 0;
    const double a = -6.283185307179586 * f;               // (-1j * 2 * np.pi * freqs) ...
    for (int p = 0; p < K; ++p) {
      double s, c;
      sincos(a * delays[size_t(r) * K + p], &s, &c);       // ... * delay
      const double g = gains[size_t(r) * K + p];
      hx += g * c;
      hy += g * s;
    }
    cd z = cmul(x, mk(hx, hy));
    if (nyquist) z.y = 0;                                   // only the real part of bin N survives .real
    if (mirror) z.y = -z.y;
    return z;
  }
  __device__ cd operator()(int g, unsigned j) const {
    const unsigned n = 2u * unsigned(N);
    if (j >= n) return mk(0, 0);
    const bool mirror = j > unsigned(N), nyq = j == unsigned(N);
    const unsigned kk = mirror ? n - j : j;
    const double f = nyq ? -double(N) * val : double(kk) * val;   // fftfreq: bin N carries -fs/2
    const cd z1 = term(2 * g, kk, f, nyq, mirror);
    const cd z2 = term(2 * g + 1, kk, f, nyq, mirror);
    return cmul(mk(z1.x - z2.y, z1.y + z2.x), w[j]);
  }
};

struct SimStorer {
  static constexpr const char* kName = "SimStorer";
  double* out;
  size_t stride;
  int rows, N, out_len, fl, row0;
  double step_in, step_out;   // 1/(fl-1), -1/(fl-1)  (np.linspace steps, signal_processing.py:77-78)
  const cd* w;
  __device__ double fade(int j) const {
    if (j < fl) return fl == 1 ? 0.0 : (j == fl - 1 ? 1.0 : double(j) * step_in);
    if (j >= N - fl) {
      const int i = j - (N - fl);
      return fl == 1 ? 1.0 : (i == fl - 1 ? 0.0 : double(i) * step_out + 1.0);
    }
    return 1.0;
  }
  __device__ void operator()(int g, unsigned j, cd y) const {
    if (j >= unsigned(out_len)) return;
    const cd z = cmul(y, w[j]);
    const double f = fade(int(j));
    const int r = row0 + 2 * g;
    if (r < rows) out[size_t(r) * stride + j] = z.x * f;
    if (r + 1 < rows) out[size_t(r + 1) * stride + j] = z.y * f;
  }
};

// normalize_signal / dynamic_range_compression, one workgroup per row (signal_processing.py:82-94).
// max|compressed| is the compressed value of the row maximum (|x/max| = 1 exactly, log1p monotone),
// so one max reduction serves both normalisations.
__global__ __launch_bounds__(256) void k_norm_compress(const double* in, size_t istride, double* out, size_t ostride,
                                                       int len, int normalize_only, double thr, double eps) {
  __shared__ double rd[4];
  const int tid = threadIdx.x;
  const double* x = in + size_t(blockIdx.x) * istride;
  double* y = out + size_t(blockIdx.x) * ostride;
  double m = 0;
  for (int i = tid; i < len; i += kLanes) m = fmax(m, fabs(x[i]));
  m = block_max(m, rd, tid);
  const double top = m > 0 ? log1p(1.0 / thr + eps) : 0.0;
  for (int i = tid; i < len; i += kLanes) {
    const double v = m == 0 ? x[i] : x[i] / m;
    if (normalize_only) { y[i] = v; continue; }
    const double sg = v > 0 ? 1.0 : (v < 0 ? -1.0 : 0.0);
    const double c = sg * log1p(fabs(v) / thr + eps);
    y[i] = top > 0 ? c / top : c;
  }
}

// Per-mic power-of-two rescale of the path gains: gains reach 1e-63 and two mics share one complex
// transform, so a mic 1e16 times weaker than its partner would drown in the partner's rounding error.  Scaling a row by
// 2^-e is exact and cancels bit for bit in normalize_signal (x / max|x|).  One lane per row (K is a handful of paths).
__global__ __launch_bounds__(256) void k_gain_rescale(const double* in, double* out, int rows, int K) {   // (in place when in == out)
  const int r = blockIdx.x * 256 + threadIdx.x;
  if (r >= rows) return;
  double top = 0;
  for (int p = 0; p < K; ++p) top = fmax(top, fabs(in[size_t(r) * K + p]));
  const bool scale = top > 0 && isfinite(top);
  const int ex = scale ? ilogb(top) : 0;
  for (int p = 0; p < K; ++p) out[size_t(r) * K + p] = scale ? ldexp(in[size_t(r) * K + p], -ex) : in[size_t(r) * K + p];
}

// np.sum(sig ** 2) per row (utils.py:413): only the ORDER of the rows' energies matters (argmax picks the reference
// microphone); identical rows give identical sums, so np.argmax's first-of-equals rule carries over.
__global__ __launch_bounds__(256) void k_row_energy(const double* __restrict__ x, size_t stride, int N, double* __restrict__ energy) {
  __shared__ double rd[4];
  const double* r = x + size_t(blockIdx.x) * stride;
  double acc = 0;
  for (int i = threadIdx.x; i < N; i += kLanes) acc = __builtin_fma(r[i], r[i], acc);
  acc = block_sum(acc, rd, threadIdx.x);
  if (threadIdx.x == 0) energy[blockIdx.x] = acc;
}

// utils.py:448-456: np.pad(sig, (pad_left, 0)) then right-pad to the common length: out[row][pad + i] = in[row][i]
__global__ __launch_bounds__(256) void k_align_rows(const double* __restrict__ in, size_t istride, int N, const int32_t* __restrict__ pad,
                                                    double* __restrict__ out, size_t ostride, int Lout) {
  const int row = blockIdx.y;
  const int p = pad[row];
  const double* src = in + size_t(row) * istride;
  double* dst = out + size_t(row) * ostride;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < Lout; i += gridDim.x * 256) {
    const int j = i - p;
    dst[i] = j >= 0 && j < N ? src[j] : 0.0;
  }
}

// ------------------------------------------------------------------ filtfilt
__device__ __forceinline__ double odd_ext(const double* x, int N, int edge, int i) {
  if (i < edge) return 2 * x[0] - x[edge - i];
  if (i < edge + N) return x[i - edge];
  return 2 * x[N - 1] - x[N - 2 - (i - edge - N)];
}

// One lane per row (the recurrence is sequential along time and scipy's operation order is kept bit for bit), 64 rows
// per wavefront.  The samples travel through LDS in tiles of 64: for every row of the tile the 64 lanes load 64
// CONSECUTIVE samples (one coalesced 512-byte request instead of 64 lanes touching 64 cache lines per step, which left
// the loop bound by one memory round trip per sample: 8 ms per launch), lane r then walks row r of the tile in LDS
// (pitch 65: conflict-free), overwrites it with its outputs, and the tile goes back to memory coalesced the same way.
// KT > 0: compile-time tap count with the state in registers, KT == 0: run-time (state in scratch memory).
constexpr int kFiltTile = 64, kFiltPitch = kFiltTile + 1;

// Rows may differ in length (`desc`: offsets of a row in x / out and its length; null: R rows of N samples back to back):
// the frames of a stream have their own synchronised lengths (utils.py:448-456) and still share one launch.
struct FiltRow { long long in_off, out_off; int n, pad; };

template <int KT>
__global__ __launch_bounds__(64) void k_filtfilt(const double* __restrict__ x, int R, int Nmax, const FiltRow* __restrict__ desc,
                                                 const double* __restrict__ b, const double* __restrict__ a,
                                                 const double* __restrict__ zi, int K, double* __restrict__ tmp,
                                                 double* __restrict__ out) {
  __shared__ double tile[64 * kFiltPitch];
  __shared__ FiltRow rows[64];
  const int lane = threadIdx.x;
  const int r0 = blockIdx.x * 64;
  const int r = r0 + lane;
  const int nrow = R - r0 < 64 ? R - r0 : 64;                  // rows of this wavefront
  const bool live = r < R;
  const int k = KT > 0 ? KT : K;
  const int edge = 3 * k;
  {
    FiltRow d;
    if (desc) d = desc[live ? r : r0];
    else { d.in_off = d.out_off = (long long)(live ? r : r0) * Nmax; d.n = Nmax; d.pad = 0; }
    rows[lane] = d;
  }
  __syncthreads();
  const int N = rows[lane].n, len = N + 2 * edge;              // this lane's own row
  const size_t tstride = size_t(Nmax) + 2 * size_t(edge);      // rows of tmp
  int lenmax = 0;
  for (int q = 0; q < nrow; ++q) lenmax = rows[q].n + 2 * edge > lenmax ? rows[q].n + 2 * edge : lenmax;
  const double* xr = x + rows[lane].in_off;
  constexpr int ZN = KT > 0 ? KT : 512;
  double z[ZN];
  double bb[KT > 0 ? KT : 1], aa[KT > 0 ? KT : 1];
  if (KT > 0) {
#pragma unroll
    for (int q = 0; q < KT; ++q) { bb[q] = b[q]; aa[q] = a[q]; }
  }
  auto step = [&](double xn) -> double {                       // scipy's DF2T lfilter step: multiply, then add
    double yn;
    if constexpr (KT > 0) {
      yn = z[0] + bb[0] * xn;
#pragma unroll
      for (int q = 0; q < KT - 2; ++q) z[q] = z[q + 1] + xn * bb[q + 1] - yn * aa[q + 1];
      z[KT - 2] = xn * bb[KT - 1] - yn * aa[KT - 1];
    } else {
      yn = z[0] + b[0] * xn;
      for (int q = 0; q < k - 2; ++q) z[q] = z[q + 1] + xn * b[q + 1] - yn * a[q + 1];
      z[k - 2] = xn * b[k - 1] - yn * a[k - 1];
    }
    return yn;
  };
  // ---- forward over the odd extension: tmp[i] for i < len
  const double x0 = odd_ext(xr, N, edge, 0);
  for (int q = 0; q < k - 1; ++q) z[q] = zi[q] * x0;
  for (int i0 = 0; i0 < lenmax; i0 += kFiltTile) {
    for (int q0 = 0; q0 < nrow; q0 += 8) {                     // row q of the tile: 64 consecutive samples, one per lane;
      double v[8];                                             // eight rows' loads in flight before the first LDS store
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int q = q0 + u < nrow ? q0 + u : nrow - 1;
        const int nq = rows[q].n;
        v[u] = (q0 + u < nrow && i0 + lane < nq + 2 * edge) ? odd_ext(x + rows[q].in_off, nq, edge, i0 + lane) : 0.0;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (q0 + u < nrow) tile[(q0 + u) * kFiltPitch + lane] = v[u];
    }
    __syncthreads();
    const int cnt = len - i0 < kFiltTile ? len - i0 : kFiltTile;   // (this lane's row; <= 0 behind its end)
    if (live)
      for (int j = 0; j < cnt; ++j) tile[lane * kFiltPitch + j] = step(tile[lane * kFiltPitch + j]);
    __syncthreads();
    for (int q = 0; q < nrow; ++q)
      if (i0 + lane < rows[q].n + 2 * edge) tmp[size_t(r0 + q) * tstride + i0 + lane] = tile[q * kFiltPitch + lane];
    __syncthreads();
  }
  // ---- backward over tmp (reversed): y[j] for the N samples inside the extension
  const double y0 = live ? tmp[size_t(r) * tstride + len - 1] : 0.0;
  for (int q = 0; q < k - 1; ++q) z[q] = zi[q] * y0;
  for (int i0 = 0; i0 < lenmax; i0 += kFiltTile) {             // position p = len - 1 - i runs backwards through tmp
    // Row q's tile holds tmp[len_q - 1 - (i0 + j)] at column j (cnt_q columns): lane l loads the ascending address
    // len_q - cnt_q - i0 + l and puts it at column cnt_q - 1 - l
    for (int q0 = 0; q0 < nrow; q0 += 8) {
      double v[8];
      int col[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int q = q0 + u < nrow ? q0 + u : nrow - 1;
        const int lq = rows[q].n + 2 * edge;
        const int cq = lq - i0 < kFiltTile ? lq - i0 : kFiltTile;
        const bool ok = q0 + u < nrow && lane < cq;
        col[u] = ok ? cq - 1 - lane : -1;
        v[u] = ok ? tmp[size_t(r0 + q) * tstride + (lq - cq - i0) + lane] : 0.0;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (col[u] >= 0) tile[(q0 + u) * kFiltPitch + col[u]] = v[u];
    }
    __syncthreads();
    const int cnt = len - i0 < kFiltTile ? len - i0 : kFiltTile;
    if (live)
      for (int j = 0; j < cnt; ++j) tile[lane * kFiltPitch + j] = step(tile[lane * kFiltPitch + j]);
    __syncthreads();
    for (int q = 0; q < nrow; ++q) {
      const int nq = rows[q].n, lq = nq + 2 * edge;
      const int cq = lq - i0 < kFiltTile ? lq - i0 : kFiltTile;
      const int p = (lq - cq - i0) + lane;                     // position of column cq - 1 - lane in the extended sequence
      const int j = p - edge;
      if (lane < cq && j >= 0 && j < nq) out[rows[q].out_off + j] = tile[q * kFiltPitch + (cq - 1 - lane)];
    }
    __syncthreads();
  }
}

// ------------------------------------------------------------------ Wiener-3
__device__ __forceinline__ void wiener_local(const double* x, int N, int i, double& mean, double& var) {
  const double l = i > 0 ? x[i - 1] : 0.0, c = x[i], r = i < N - 1 ? x[i + 1] : 0.0;
  mean = ((l + c) + r) / 3;
  var = ((l * l + c * c) + r * r) / 3 - mean * mean;
}

__global__ __launch_bounds__(256) void k_wiener3(const double* in, double* out, int N) {
  __shared__ double rd[4];
  const int tid = threadIdx.x;
  const double* x = in + size_t(blockIdx.x) * N;
  double* y = out + size_t(blockIdx.x) * N;
  double acc = 0;
  for (int i = tid; i < N; i += kLanes) {
    double m, v;
    wiener_local(x, N, i, m, v);
    acc += v;
  }
  const double noise = block_sum(acc, rd, tid) / double(N);
  for (int i = tid; i < N; i += kLanes) {
    double m, v;
    wiener_local(x, N, i, m, v);
    double res = (x[i] - m) * (1 - noise / v) + m;
    y[i] = v < noise ? m : res;
  }
}

// ------------------------------------------------------------------ plain cross-correlation
struct RefLoader {
  static constexpr const char* kName = "RefLoader";        // reversed reference as the convolution kernel
  const double* ref;
  int N;
  __device__ cd operator()(int, unsigned j) const { return j < unsigned(N) ? mk(ref[N - 1 - j], 0.0) : mk(0, 0); }
};

struct RowPairLoader {
  static constexpr const char* kName = "RowPairLoader";    // two real rows per complex transform
  const double* x;
  int N, rows;
  __device__ cd operator()(int g, unsigned j) const {
    if (j >= unsigned(N)) return mk(0, 0);
    const double a = x[size_t(2 * g) * N + j];
    const double b = 2 * g + 1 < rows ? x[size_t(2 * g + 1) * N + j] : 0.0;
    return mk(a, b);
  }
};

struct PlainStorer {
  static constexpr const char* kName = "PlainStorer";
  double* corr;
  size_t stride;
  int len;
  __device__ void operator()(int g, unsigned j, cd y) const {
    if (j >= unsigned(len)) return;
    corr[size_t(2 * g) * stride + j] = y.x;
    corr[size_t(2 * g + 1) * stride + j] = y.y;
  }
};

__global__ __launch_bounds__(256) void k_xcorr_peak(const double* corr, size_t stride, int len, int32_t* kpk, double* win5,
                                                    double* pkabs) {
  __shared__ double rd[4];
  __shared__ int ri[4];
  const int tid = threadIdx.x;
  const double* c = corr + size_t(blockIdx.x) * stride;
  double best = 0;
  int bi = -1;
  for (int i = tid; i < len; i += kLanes) {
    const double v = fabs(c[i]);
    if (bi < 0 || v > best) { best = v; bi = i; }
  }
  block_arg<0>(best, bi, rd, ri, tid);
  if (tid == 0) {
    kpk[blockIdx.x] = bi;
    pkabs[blockIdx.x] = best;
    for (int q = -2; q <= 2; ++q) win5[size_t(blockIdx.x) * 5 + q + 2] = (bi + q >= 0 && bi + q < len) ? c[bi + q] : NAN;
  }
}

}  // namespace

// ------------------------------------------------------------------ host-side pipelines
static int simulate_dev(Engine* e, const double* d_base, int bases, int nbase, double fs, int N, const double* d_delays,
                        const double* d_gains, int rows, int rows_per_base, int K, int out_len, bool compress,
                        bool normalize, double* d_out) {
  if (N < 100) return e->fail(PAL_ERR_INVALID, "fractional_delay needs at least 100 samples (fade slice of signal_processing.py:75-78)");
  if (nbase > N) return e->fail(PAL_ERR_INVALID, "base signal longer than total_samples");
  if (N > (1 << 19)) return e->fail(PAL_ERR_UNSUPPORTED, "total_samples %d exceeds 2^19", N);
  Plan* pl = nullptr;
  PAL_TRY(e->get_plan(2 * N, nbase, N, &pl));
  void* sp = nullptr;
  PAL_TRY(e->scratch(2, size_t(bases) * pl->H * sizeof(cd), &sp));
  cd* X = static_cast<cd*>(sp);
  PAL_TRY(e->forward_spectra(*pl, d_base, size_t(nbase), bases, nbase, X));
  const Conv& c = pl->inv;
  void* wsp = nullptr;
  PAL_TRY(e->scratch(0, size_t(e->chunk) * c.M() * sizeof(cd), &wsp));
  cd* W = static_cast<cd*>(wsp);
  const int fl = int(0.01 * double(N));
  const double d = 1.0 / fs;
  const double val = 1.0 / (double(2 * N) * d);
  const int ntr = (rows + 1) / 2;
  for (int t0 = 0; t0 < ntr; t0 += e->chunk) {
    const int G = ntr - t0 < e->chunk ? ntr - t0 : e->chunk;
    const int r0 = 2 * t0;
    SimLoader ld{X, d_delays, d_gains, K, rows, rows_per_base, N, r0, val, pl->w};
    SimStorer st{d_out, size_t(out_len), rows, N, out_len, fl, r0,
                 fl > 1 ? 1.0 / double(fl - 1) : 0.0, fl > 1 ? -1.0 / double(fl - 1) : 0.0, pl->w};
    PAL_TRY(launch_cols_fwd(e, c, G, ld, W));
    PAL_TRY(launch_rows(e, c, G, W, true, 1.0));
    PAL_TRY(launch_cols_inv(e, c, G, W, st));
  }
  if (compress || normalize) {
    ProfScope ps(e, "k_norm_compress");
    k_norm_compress<<<dim3(rows), dim3(kLanes), 0, e->stream>>>(d_out, size_t(out_len), d_out, size_t(out_len), out_len,
                                                               compress ? 0 : 1, 0.8, 1e-8);
    PAL_TRY(e->check(hipGetLastError(), "k_norm_compress"));
  }
  return PAL_OK;
}


// filtfilt of rows[R][N] that already sit in HBM (b | a | zi normalised by a[0] are uploaded: 3 K doubles)
// `rows` (host, optional): per-row offsets and lengths for rows of different length (N = the longest then)
static int filtfilt_dev(Engine* e, const double* b, int nb, const double* a, int na, const double* zi, const double* d_x, int R,
                        int N, double* d_y, const FiltRow* rows = nullptr) {
  if (!b || !a || !zi || !d_x || !d_y || nb < 1 || na < 1 || R < 1) return e->fail(PAL_ERR_INVALID, "bad filtfilt arguments");
  const int K = nb > na ? nb : na;
  if (K < 2 || K > 512) return e->fail(PAL_ERR_UNSUPPORTED, "filter length %d outside 2..512", K);
  const int edge = 3 * K;
  if (N <= edge) return e->fail(PAL_ERR_INVALID, "The length of the input vector x must be greater than padlen, which is %d.", edge);
  if (a[0] == 0) return e->fail(PAL_ERR_INVALID, "a[0] must be non-zero");
  std::vector<double> coef(size_t(3 * K), 0.0);          // b | a | zi, normalised by a[0] like scipy's lfilter
  for (int q = 0; q < nb; ++q) coef[q] = b[q] / a[0];
  for (int q = 0; q < na; ++q) coef[K + q] = a[q] / a[0];
  for (int q = 0; q < K - 1; ++q) coef[2 * K + q] = zi[q];
  void *dc = nullptr, *dt = nullptr, *dd = nullptr;
  PAL_TRY(e->scratch(3, coef.size() * sizeof(double) + (rows ? size_t(R) * sizeof(FiltRow) : 0), &dc));
  PAL_TRY(e->scratch(1, size_t(R) * (N + 2 * edge) * sizeof(double), &dt));
  PAL_TRY(e->check(hipMemcpyAsync(dc, coef.data(), coef.size() * sizeof(double), hipMemcpyHostToDevice, e->stream), "upload"));
  if (rows) {
    dd = static_cast<char*>(dc) + coef.size() * sizeof(double);
    PAL_TRY(e->check(hipMemcpyAsync(dd, rows, size_t(R) * sizeof(FiltRow), hipMemcpyHostToDevice, e->stream), "upload"));
  }
  PAL_TRY(e->check(hipStreamSynchronize(e->stream), "upload sync"));   // `coef` / `rows` are host memory
  const double* cb = static_cast<double*>(dc);
  {
    ProfScope ps(e, "k_filtfilt");
    const dim3 grid((R + 63) / 64);
    const FiltRow* dr = static_cast<const FiltRow*>(dd);
    if (K == 11) k_filtfilt<11><<<grid, dim3(64), 0, e->stream>>>(d_x, R, N, dr, cb, cb + K, cb + 2 * K, K, static_cast<double*>(dt), d_y);
    else k_filtfilt<0><<<grid, dim3(64), 0, e->stream>>>(d_x, R, N, dr, cb, cb + K, cb + 2 * K, K, static_cast<double*>(dt), d_y);
  }
  return e->check(hipGetLastError(), "k_filtfilt");
}

// full cross-correlation of rows[R][N] against row ref_idx (all in HBM): kpk / win5 / pkabs land in device arrays of R entries
static int xcorr_dev(Engine* e, const double* x, int R, int N, int ref_idx, int32_t* dk, double* dw, double* dp) {
  if (N > (1 << 20)) return e->fail(PAL_ERR_UNSUPPORTED, "signal longer than 2^20 samples");
  const int len = 2 * N - 1;
  // geometry, tables and kernel-spectrum storage of the convolution are kept per length (sync runs once per frame in
  // the streaming configuration; hipMalloc / hipFree per call cost more than the correlations)
  Conv* cp = nullptr;
  PAL_TRY(e->xcorr_conv(size_t(len), &cp));
  const Conv& c = *cp;
  void *wsp = nullptr, *dcor = nullptr;
  RefLoader rl{x + size_t(ref_idx) * N, N};
  PAL_TRY(launch_cols_fwd(e, c, 1, rl, c.chat));
  PAL_TRY(launch_rows(e, c, 1, c.chat, false, 1.0 / double(c.M())));
  PAL_TRY(e->scratch(0, size_t(e->chunk) * c.M() * sizeof(cd), &wsp));
  const size_t stride = size_t(len) + 1;
  PAL_TRY(e->scratch(1, size_t(2 * e->chunk) * stride * sizeof(double), &dcor));
  const int ntr = (R + 1) / 2;
  for (int t0 = 0; t0 < ntr; t0 += e->chunk) {
    const int G = ntr - t0 < e->chunk ? ntr - t0 : e->chunk;
    const int r0 = 2 * t0, nrows = R - r0 < 2 * G ? R - r0 : 2 * G;
    RowPairLoader ld{x + size_t(r0) * N, N, R - r0};
    PlainStorer st{static_cast<double*>(dcor), stride, len};
    PAL_TRY(launch_cols_fwd(e, c, G, ld, static_cast<cd*>(wsp)));
    PAL_TRY(launch_rows(e, c, G, static_cast<cd*>(wsp), true, 1.0));
    PAL_TRY(launch_cols_inv(e, c, G, static_cast<cd*>(wsp), st));
    {
      ProfScope ps(e, "k_xcorr_peak");
      k_xcorr_peak<<<dim3(nrows), dim3(kLanes), 0, e->stream>>>(static_cast<double*>(dcor), stride, len, dk + r0, dw + size_t(r0) * 5, dp + r0);
    }
    PAL_TRY(e->check(hipGetLastError(), "k_xcorr_peak"));
  }
  return PAL_OK;
}

}  // namespace pal

using namespace pal;

#define ENGINE(h)                                   \
  if (!(h)) return PAL_ERR_INVALID;                 \
  Engine* e = reinterpret_cast<Engine*>(h);         \
  if (hipSetDevice(e->device) != hipSuccess) return e->fail(PAL_ERR_HIP, "hipSetDevice(%d) failed", e->device)

#define UP(dst, src, bytes) PAL_TRY(e->check(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, e->stream), "upload"))
#define DOWN(dst, src, bytes) PAL_TRY(e->check(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, e->stream), "download"))

extern "C" {

int pal_simulate_multipath(pal_handle h, const double* base, int B, int nbase, double fs, int total_samples,
                           const double* delays, const double* gains, int M, int K, int trim_len, double* out) {
  ENGINE(h);
  if (!base || !delays || !gains || !out) return e->fail(PAL_ERR_INVALID, "NULL buffer");
  if (B < 1 || M < 1 || K < 1 || nbase < 1 || !(fs > 0)) return e->fail(PAL_ERR_INVALID, "bad simulation geometry");
  const int N = total_samples;
  const int out_len = trim_len > 0 && trim_len < N ? trim_len : N;
  const int rows = B * M;
  void *db = nullptr, *dd = nullptr, *dg = nullptr, *dout = nullptr;
  int rc = PAL_OK;
  do {
    if ((rc = e->scratch(4, size_t(B) * nbase * sizeof(double), &db)) != PAL_OK) break;
    if ((rc = e->scratch(5, size_t(rows) * K * sizeof(double) * 2, &dd)) != PAL_OK) break;
    dg = static_cast<double*>(dd) + size_t(rows) * K;
    if ((rc = e->scratch(6, size_t(rows) * out_len * sizeof(double), &dout)) != PAL_OK) break;
    if ((rc = e->check(hipMemcpyAsync(db, base, size_t(B) * nbase * sizeof(double), hipMemcpyHostToDevice, e->stream), "upload")) != PAL_OK) break;
    if ((rc = e->check(hipMemcpyAsync(dd, delays, size_t(rows) * K * sizeof(double), hipMemcpyHostToDevice, e->stream), "upload")) != PAL_OK) break;
    // per-mic power-of-two rescale of the gains: k_gain_rescale, in place
    if ((rc = e->check(hipMemcpyAsync(dg, gains, size_t(rows) * K * sizeof(double), hipMemcpyHostToDevice, e->stream), "upload")) != PAL_OK) break;
    k_gain_rescale<<<dim3((rows + 255) / 256), dim3(256), 0, e->stream>>>(static_cast<double*>(dg), static_cast<double*>(dg), rows, K);
    if ((rc = e->check(hipGetLastError(), "k_gain_rescale")) != PAL_OK) break;
    if ((rc = e->check(hipStreamSynchronize(e->stream), "upload sync")) != PAL_OK) break;
    rc = simulate_dev(e, static_cast<double*>(db), B, nbase, fs, N, static_cast<double*>(dd), static_cast<double*>(dg), rows, M,
                      K, out_len, true, true, static_cast<double*>(dout));
    if (rc != PAL_OK) break;
    rc = e->check(hipMemcpyAsync(out, dout, size_t(rows) * out_len * sizeof(double), hipMemcpyDeviceToHost, e->stream), "download");
  } while (0);
  if (rc != PAL_OK) return rc;
  return pal_synchronize(h);
}

int pal_fractional_delay(pal_handle h, const double* rows_in, int R, int N, const double* delays, double fs, double* out) {
  ENGINE(h);
  if (!rows_in || !delays || !out || R < 1 || !(fs > 0)) return e->fail(PAL_ERR_INVALID, "bad fractional_delay arguments");
  void *db = nullptr, *dd = nullptr, *dout = nullptr;
  PAL_TRY(e->scratch(4, size_t(R) * N * sizeof(double), &db));
  PAL_TRY(e->scratch(5, size_t(R) * sizeof(double) * 2, &dd));
  PAL_TRY(e->scratch(6, size_t(R) * N * sizeof(double), &dout));
  std::vector<double> ones(size_t(R), 1.0);
  double* dg = static_cast<double*>(dd) + R;
  UP(db, rows_in, size_t(R) * N * sizeof(double));
  UP(dd, delays, size_t(R) * sizeof(double));
  UP(dg, ones.data(), size_t(R) * sizeof(double));
  PAL_TRY(e->check(hipStreamSynchronize(e->stream), "upload sync"));
  PAL_TRY(simulate_dev(e, static_cast<double*>(db), R, N, fs, N, static_cast<double*>(dd), dg, R, 1, 1, N, false, false,
                       static_cast<double*>(dout)));
  DOWN(out, dout, size_t(R) * N * sizeof(double));
  return pal_synchronize(h);
}

int pal_normalize_compress(pal_handle h, const double* rows_in, int R, int N, int normalize_only, double threshold,
                           double epsilon, double* out) {
  ENGINE(h);
  if (!rows_in || !out || R < 1 || N < 1) return e->fail(PAL_ERR_INVALID, "bad normalize arguments");
  void* db = nullptr;
  PAL_TRY(e->scratch(4, size_t(R) * N * sizeof(double), &db));
  UP(db, rows_in, size_t(R) * N * sizeof(double));
  {
    ProfScope ps(e, "k_norm_compress");
    k_norm_compress<<<dim3(R), dim3(kLanes), 0, e->stream>>>(static_cast<double*>(db), size_t(N), static_cast<double*>(db),
                                                            size_t(N), N, normalize_only, threshold, epsilon);
  }
  PAL_TRY(e->check(hipGetLastError(), "k_norm_compress"));
  DOWN(out, db, size_t(R) * N * sizeof(double));
  return pal_synchronize(h);
}

int pal_filtfilt(pal_handle h, const double* b, int nb, const double* a, int na, const double* zi, const double* rows_in,
                 int R, int N, double* out) {
  ENGINE(h);
  if (!rows_in || !out || R < 1 || N < 1) return e->fail(PAL_ERR_INVALID, "bad filtfilt arguments");
  void *dx = nullptr, *dy = nullptr;
  PAL_TRY(e->scratch(4, size_t(R) * N * sizeof(double), &dx));
  PAL_TRY(e->scratch(6, size_t(R) * N * sizeof(double), &dy));
  UP(dx, rows_in, size_t(R) * N * sizeof(double));
  PAL_TRY(filtfilt_dev(e, b, nb, a, na, zi, static_cast<double*>(dx), R, N, static_cast<double*>(dy)));
  DOWN(out, dy, size_t(R) * N * sizeof(double));
  return pal_synchronize(h);
}

int pal_wiener3(pal_handle h, const double* rows_in, int R, int N, double* out) {
  ENGINE(h);
  if (!rows_in || !out || R < 1 || N < 1) return e->fail(PAL_ERR_INVALID, "bad wiener arguments");
  void *dx = nullptr, *dy = nullptr;
  PAL_TRY(e->scratch(4, size_t(R) * N * sizeof(double), &dx));
  PAL_TRY(e->scratch(6, size_t(R) * N * sizeof(double), &dy));
  UP(dx, rows_in, size_t(R) * N * sizeof(double));
  {
    ProfScope ps(e, "k_wiener3");
    k_wiener3<<<dim3(R), dim3(kLanes), 0, e->stream>>>(static_cast<double*>(dx), static_cast<double*>(dy), N);
  }
  PAL_TRY(e->check(hipGetLastError(), "k_wiener3"));
  DOWN(out, dy, size_t(R) * N * sizeof(double));
  return pal_synchronize(h);
}

int pal_xcorr_vs_ref(pal_handle h, const double* rows_in, int R, int N, int ref_idx, int32_t* kpk, double* win5,
                     double* pkabs, double* refpk) {
  ENGINE(h);
  if (!rows_in || !kpk || !win5 || !pkabs || R < 1 || N < 1 || ref_idx < 0 || ref_idx >= R)
    return e->fail(PAL_ERR_INVALID, "bad xcorr arguments");
  void *dx = nullptr, *dres = nullptr;
  PAL_TRY(e->scratch(4, size_t(R) * N * sizeof(double), &dx));
  UP(dx, rows_in, size_t(R) * N * sizeof(double));
  // result block: kpk[R] (int32, padded to 16 bytes) | win5[R][5] | pkabs[R]
  const size_t off_d = (size_t(R) * sizeof(int32_t) + 15) & ~size_t(15);
  PAL_TRY(e->scratch(5, off_d + size_t(R) * 6 * sizeof(double), &dres));
  int32_t* dk = static_cast<int32_t*>(dres);
  double* dw = reinterpret_cast<double*>(static_cast<char*>(dres) + off_d);
  double* dp = dw + size_t(R) * 5;
  int rc = xcorr_dev(e, static_cast<const double*>(dx), R, N, ref_idx, dk, dw, dp);
  if (rc == PAL_OK) rc = e->check(hipMemcpyAsync(kpk, dk, size_t(R) * sizeof(int32_t), hipMemcpyDeviceToHost, e->stream), "download");
  if (rc == PAL_OK) rc = e->check(hipMemcpyAsync(win5, dw, size_t(R) * 5 * sizeof(double), hipMemcpyDeviceToHost, e->stream), "download");
  if (rc == PAL_OK) rc = e->check(hipMemcpyAsync(pkabs, dp, size_t(R) * sizeof(double), hipMemcpyDeviceToHost, e->stream), "download");
  if (rc != PAL_OK) { (void)hipStreamSynchronize(e->stream); return rc; }
  PAL_TRY(pal_synchronize(h));
  if (refpk) *refpk = pkabs[ref_idx];
  return PAL_OK;
}

/* ---- device-resident forms (streaming configuration: simulate -> synchronise -> prefilter -> pairs without host copies
 *      of the waveforms; the host supplies path tables and filter coefficients and reads back a few numbers per row) ---- */

int pal_simulate_multipath_dev(pal_handle h, const double* d_base, int B, int nbase, double fs, int total_samples,
                               const double* d_delays, const double* d_gains, int M, int K, int trim_len, double* d_out) {
  ENGINE(h);
  if (!d_base || !d_delays || !d_gains || !d_out) return e->fail(PAL_ERR_INVALID, "NULL buffer");
  if (B < 1 || M < 1 || K < 1 || nbase < 1 || !(fs > 0)) return e->fail(PAL_ERR_INVALID, "bad simulation geometry");
  const int N = total_samples;
  const int out_len = trim_len > 0 && trim_len < N ? trim_len : N;
  const int rows = B * M;
  void* dg = nullptr;
  PAL_TRY(e->scratch(5, size_t(rows) * K * sizeof(double), &dg));
  k_gain_rescale<<<dim3((rows + 255) / 256), dim3(256), 0, e->stream>>>(d_gains, static_cast<double*>(dg), rows, K);
  PAL_TRY(e->check(hipGetLastError(), "k_gain_rescale"));
  return simulate_dev(e, d_base, B, nbase, fs, N, d_delays, static_cast<double*>(dg), rows, M, K, out_len, true, true, d_out);
}

int pal_filtfilt_dev(pal_handle h, const double* b, int nb, const double* a, int na, const double* zi, const double* d_rows,
                     int R, int N, double* d_out) {
  ENGINE(h);
  return filtfilt_dev(e, b, nb, a, na, zi, d_rows, R, N, d_out);
}

int pal_filtfilt_ragged_dev(pal_handle h, const double* b, int nb, const double* a, int na, const double* zi, const double* d_in,
                            double* d_out, int R, const int64_t* in_off, const int64_t* out_off, const int32_t* lengths) {
  ENGINE(h);
  if (!in_off || !out_off || !lengths || R < 1) return e->fail(PAL_ERR_INVALID, "bad filtfilt arguments");
  const int K = nb > na ? nb : na;
  std::vector<FiltRow> rows(static_cast<size_t>(R));
  int nmax = 0;
  for (int r = 0; r < R; ++r) {
    if (lengths[r] <= 3 * K) return e->fail(PAL_ERR_INVALID, "The length of the input vector x must be greater than padlen, which is %d.", 3 * K);
    if (in_off[r] < 0 || out_off[r] < 0) return e->fail(PAL_ERR_INVALID, "negative row offset");
    rows[size_t(r)] = FiltRow{in_off[r], out_off[r], lengths[r], 0};
    nmax = lengths[r] > nmax ? lengths[r] : nmax;
  }
  return filtfilt_dev(e, b, nb, a, na, zi, d_in, R, nmax, d_out, rows.data());
}

int pal_wiener3_dev(pal_handle h, const double* d_rows, int R, int N, double* d_out) {
  ENGINE(h);
  if (!d_rows || !d_out || R < 1 || N < 1) return e->fail(PAL_ERR_INVALID, "bad wiener arguments");
  {
    ProfScope ps(e, "k_wiener3");
    k_wiener3<<<dim3(R), dim3(kLanes), 0, e->stream>>>(d_rows, d_out, N);
  }
  return e->check(hipGetLastError(), "k_wiener3");
}

int pal_row_energies_dev(pal_handle h, const double* d_rows, int R, int N, double* energy) {
  ENGINE(h);
  if (!d_rows || !energy || R < 1 || N < 1) return e->fail(PAL_ERR_INVALID, "bad energy arguments");
  void* de = nullptr;
  PAL_TRY(e->scratch(6, size_t(R) * sizeof(double), &de));
  k_row_energy<<<dim3(R), dim3(kLanes), 0, e->stream>>>(d_rows, size_t(N), N, static_cast<double*>(de));
  PAL_TRY(e->check(hipGetLastError(), "k_row_energy"));
  PAL_TRY(e->check(hipMemcpyAsync(energy, de, size_t(R) * sizeof(double), hipMemcpyDeviceToHost, e->stream), "download"));
  return e->check(hipStreamSynchronize(e->stream), "energy sync");
}

int pal_sync_measure_dev(pal_handle h, const double* d_rows, int B, int M, int N, int32_t* ref_idx, int32_t* kpk,
                         double* win5, double* pkabs, double* refpk) {
  ENGINE(h);
  if (!d_rows || !ref_idx || !kpk || !win5 || !pkabs || B < 1 || M < 1 || N < 1) return e->fail(PAL_ERR_INVALID, "bad sync arguments");
  const int R = B * M;
  // energies of every row -> the reference microphone of each frame (np.argmax: first of equals)
  void* de = nullptr;
  PAL_TRY(e->scratch(6, size_t(R) * sizeof(double), &de));
  k_row_energy<<<dim3(R), dim3(kLanes), 0, e->stream>>>(d_rows, size_t(N), N, static_cast<double*>(de));
  PAL_TRY(e->check(hipGetLastError(), "k_row_energy"));
  std::vector<double> en(size_t(R), 0.0);
  PAL_TRY(e->check(hipMemcpyAsync(en.data(), de, size_t(R) * sizeof(double), hipMemcpyDeviceToHost, e->stream), "download"));
  PAL_TRY(e->check(hipStreamSynchronize(e->stream), "energy sync"));
  for (int b = 0; b < B; ++b) {
    if (ref_idx[b] >= 0 && ref_idx[b] < M) continue;            // the caller has chosen (near-ties of the energies settled with numpy)
    int best = 0;
    for (int m = 1; m < M; ++m)
      if (en[size_t(b) * M + m] > en[size_t(b) * M + best]) best = m;      // (NaN never wins: np.argmax would return the NaN row -
    for (int m = 0; m < M; ++m)                                            //  checked here)
      if (en[size_t(b) * M + m] != en[size_t(b) * M + m]) { best = m; break; }
    ref_idx[b] = best;
  }
  // result block for all frames: kpk[R] | win5[R][5] | pkabs[R]
  void* dres = nullptr;
  const size_t off_d = (size_t(R) * sizeof(int32_t) + 15) & ~size_t(15);
  PAL_TRY(e->scratch(5, off_d + size_t(R) * 6 * sizeof(double), &dres));
  int32_t* dk = static_cast<int32_t*>(dres);
  double* dw = reinterpret_cast<double*>(static_cast<char*>(dres) + off_d);
  double* dp = dw + size_t(R) * 5;
  int rc = PAL_OK;
  for (int b = 0; b < B && rc == PAL_OK; ++b)     // (the kernel spectrum is the frame's own reference row: one convolution set-up per frame)
    rc = xcorr_dev(e, d_rows + size_t(b) * M * N, M, N, ref_idx[b], dk + size_t(b) * M, dw + size_t(b) * M * 5, dp + size_t(b) * M);
  if (rc == PAL_OK) rc = e->check(hipMemcpyAsync(kpk, dk, size_t(R) * sizeof(int32_t), hipMemcpyDeviceToHost, e->stream), "download");
  if (rc == PAL_OK) rc = e->check(hipMemcpyAsync(win5, dw, size_t(R) * 5 * sizeof(double), hipMemcpyDeviceToHost, e->stream), "download");
  if (rc == PAL_OK) rc = e->check(hipMemcpyAsync(pkabs, dp, size_t(R) * sizeof(double), hipMemcpyDeviceToHost, e->stream), "download");
  if (rc != PAL_OK) { (void)hipStreamSynchronize(e->stream); return rc; }
  PAL_TRY(pal_synchronize(h));
  if (refpk)
    for (int b = 0; b < B; ++b) refpk[b] = pkabs[size_t(b) * M + ref_idx[b]];
  return PAL_OK;
}

int pal_align_rows_dev(pal_handle h, const double* d_rows, int R, int N, const int32_t* pad_left, int Lout, double* d_out) {
  ENGINE(h);
  if (!d_rows || !pad_left || !d_out || R < 1 || N < 1 || Lout < N) return e->fail(PAL_ERR_INVALID, "bad align arguments");
  for (int r = 0; r < R; ++r)
    if (pad_left[r] < 0 || pad_left[r] + N > Lout) return e->fail(PAL_ERR_INVALID, "pad %d of row %d does not fit %d samples", pad_left[r], r, Lout);
  void* dp = nullptr;
  PAL_TRY(e->scratch(3, size_t(R) * sizeof(int32_t), &dp));
  PAL_TRY(e->check(hipMemcpyAsync(dp, pad_left, size_t(R) * sizeof(int32_t), hipMemcpyHostToDevice, e->stream), "upload"));
  PAL_TRY(e->check(hipStreamSynchronize(e->stream), "upload sync"));   // `pad_left` is host memory
  const unsigned gx = unsigned((Lout + 255) / 256 < 64 ? (Lout + 255) / 256 : 64);
  k_align_rows<<<dim3(gx, unsigned(R)), dim3(256), 0, e->stream>>>(d_rows, size_t(N), N, static_cast<const int32_t*>(dp), d_out, size_t(Lout), Lout);
  return e->check(hipGetLastError(), "k_align_rows");
}

}  // extern "C"
