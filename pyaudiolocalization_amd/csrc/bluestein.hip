// bluestein.hip - exact-length DFTs for GCC-PHAT on gfx950 (fp64, no MFMA: the work is
// butterflies and streaming, not a dense contraction).
//
// Replaces numpy.fft.fft(sig, n) / numpy.fft.ifft(R) of utils.py:114-118 at the exact length
// n = n1 + n2 - 1 (zero-padding to a smooth length changes the selected peaks).
//
//   DFT_n(x)[k] = conj(w_k) * sum_j (x_j conj(w_j)) w_{k-j},   w_j = exp(i pi j^2 / n)
//
// i.e. one circular convolution of power-of-two length M = M1 x M2 with a fixed chirp.  It runs as
// three launches over a resident HBM workspace W[g][M] (g = transform within a launch group):
//
//   cols_fwd : build the input on the fly (loader functor), M1-point column FFTs in LDS,
//              four-step twiddle, write W                         (read: source, write: 16 B/pt)
//   rows_conv: M2-point row FFT, multiply by the chirp spectrum (L2/Infinity-Cache resident,
//              shared by every transform), M2-point inverse row FFT, in place   (16 B r + 16 B w)
//   cols_inv : conj twiddle, M1-point inverse column FFTs, hand y[j] to a storer functor
//
// The inverse PHAT transform packs two mic pairs into one complex transform (real part = pair p,
// imaginary part = pair q) and whitens R = S_a conj(S_b) / (|.| + 1e-10) inside the loader, so the
// whitened cross spectrum never exists in memory.
#include <cmath>

#include "conv_kernels.h"

namespace pal {

// ------------------------------------------------------------------ table generators
// w[j] = exp(i pi mult j^2 / n)
__global__ void k_make_chirp(cd* w, int n, int mult) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  unsigned long long r = (unsigned long long)j * (unsigned long long)j % (2ull * (unsigned long long)n);
  r = r * (unsigned long long)mult % (2ull * (unsigned long long)n);
  double sgn = 1.0;
  if (r >= (unsigned long long)n) { r -= n; sgn = -1.0; }
  double s, c;
  sincospi(double(r) / double(n), &s, &c);
  w[j] = mk(sgn * c, sgn * s);
}

// out[q] = exp(-2 pi i q / denom)
__global__ void k_make_roots(cd* out, int count, double denom) {
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= count) return;
  double s, c;
  sincospi(-2.0 * double(q) / denom, &s, &c);
  out[q] = mk(c, s);
}

__global__ void k_make_stage_tw(cd* out, int ln, bool compact) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  for (int lp = stage_log2r(ln, 0); lp < ln; lp += stage_log2r(ln, lp)) {
    const int R = stage_radix(ln, lp), P = 1 << lp, off = stage_tw_offset(ln, lp);
    if (compact && stage_is_last(ln, lp)) {          // rows r = 1, 2, 4 (, 8) only
      if (idx >= off && idx < off + stage_log2r(ln, lp) * P) {
        const int r = 1 << ((idx - off) / P), k = (idx - off) % P;
        double s, c;
        sincospi(-2.0 * double(k * r) / double(P * R), &s, &c);
        out[idx] = mk(c, s);
      }
    } else if (idx >= off && idx < off + (R - 1) * P) {
      const int r = (idx - off) / P + 1, k = (idx - off) % P;
      double s, c;
      sincospi(-2.0 * double(k * r) / double(P * R), &s, &c);
      out[idx] = mk(c, s);
    }
  }
}

// ------------------------------------------------------------------ loader / storer functors
// chirp kernel of the convolution: c[m mod M] = w_|m| (or its conjugate) for -neg < m < pos
struct ChirpLoader {
  static constexpr const char* kName = "ChirpLoader";
  const cd* w;
  int neg, pos;
  unsigned M;
  bool conj_kernel;
  __device__ cd operator()(int, unsigned j) const {
    unsigned a;
    if (j < (unsigned)pos) a = j;
    else if (M - j < (unsigned)neg) a = M - j;
    else return mk(0, 0);
    cd v = w[a];
    return conj_kernel ? cconj(v) : v;
  }
};

// forward transform input: x_j conj(w_j)
struct FrameLoader {
  static constexpr const char* kName = "FrameLoader";
  const double* x;
  size_t stride;
  int len;
  const cd* w;
  __device__ cd operator()(int g, unsigned j) const {
    if (j >= (unsigned)len) return mk(0, 0);
    const double v = x[size_t(g) * stride + j];
    const cd c = w[j];
    return mk(v * c.x, -v * c.y);
  }
};

// forward transform output: X_k = conj(w_k) y_k, k < H
struct SpectrumStorer {
  static constexpr const char* kName = "SpectrumStorer";
  cd* S;
  int H;
  const cd* w;
  __device__ void operator()(int g, unsigned j, cd y) const {
    if (j < (unsigned)H) S[size_t(g) * H + j] = cmulc(y, w[j]);
  }
};

// forward transform output straight into the layout of the prime-factor route (pfa.hip): SP[row][k1][k2] holds the
// Hermitian-extended spectrum at k = CRT(k1, k2) for k1 < NR = (N1+1)/2.  Bin j < H lands at (j mod N1, j mod N2) when
// that row is kept; its mirror n - j has the residues (N1 - k1, N2 - k2) and takes the conjugate when THAT row is
// kept (row 0 pairs with itself and takes both).
struct PermSpectrumStorer {
  static constexpr const char* kName = "PermSpectrumStorer";
  cd* SP;
  int H, N1, N2, NR;
  float inv1, inv2;
  const cd* w;
  const int* slot;   // column k2 -> position inside a row (Rader rows keep generator order, pfa_rader.h); null: k2 itself
  __device__ void operator()(int g, unsigned j, cd y) const {
    if (j >= (unsigned)H) return;
    const cd v = cmulc(y, w[j]);
    int k1 = int(j) - int(unsigned(float(j) * inv1)) * N1;      // j < 2^24: the float quotient is off by at most one
    k1 = k1 < 0 ? k1 + N1 : (k1 >= N1 ? k1 - N1 : k1);
    int k2 = int(j) - int(unsigned(float(j) * inv2)) * N2;
    k2 = k2 < 0 ? k2 + N2 : (k2 >= N2 ? k2 - N2 : k2);
    cd* base = SP + size_t(g) * NR * N2;
    const int km = k2 ? N2 - k2 : 0;
    if (k1 < NR) base[size_t(k1) * N2 + (slot ? slot[k2] : k2)] = v;
    if (k1 == 0 ? j != 0 : k1 >= NR) base[size_t(k1 ? N1 - k1 : 0) * N2 + (slot ? slot[km] : km)] = cconj(v);
  }
};

// inverse transform input: (R^p_k + i R^q_k) w_k over the full Hermitian-extended grid k < n
struct PairLoader {
  static constexpr const char* kName = "PairLoader";
  const cd* S;       // spectra[row][H]
  const int4* quad;  // rows (a, b) of pair p and (c, d) of pair q; c < 0: no second pair
  int n, H;
  const cd* w;
  __device__ cd operator()(int g, unsigned j) const {
    if (j >= (unsigned)n) return mk(0, 0);
    const bool mirror = j >= (unsigned)H;
    const unsigned jj = mirror ? n - j : j;
    const int4 q = quad[g];
    const cd sa = S[size_t(q.x) * H + jj];
    cd r1 = whiten(sa, S[size_t(q.y) * H + jj]);
    cd r2 = mk(0, 0);
    // consecutive pairs of the i<j order share their first mic: reuse its spectrum bin (a quarter of the reads)
    if (q.z >= 0) r2 = whiten(q.z == q.x ? sa : S[size_t(q.z) * H + jj], S[size_t(q.w) * H + jj]);
    if (mirror) { r1.y = -r1.y; r2.y = -r2.y; }
    return cmul(mk(r1.x - r2.y, r1.y + r2.x), w[j]);
  }
};

// inverse transform output: z_m = w_m y_m; real part -> row 2g, imaginary part -> row 2g+1
struct CorrStorer {
  static constexpr const char* kName = "CorrStorer";
  double* corr;
  size_t stride;
  int n;
  const cd* w;
  const int* zero_rows;   // per row: 1 = a microphone of the pair is silent and the row is exactly zero (k_pair_zero); or null
  __device__ void operator()(int g, unsigned j, cd y) const {
    if (j >= (unsigned)n) return;
    const cd z = cmul(y, w[j]);
    corr[size_t(2 * g) * stride + j] = zero_rows && zero_rows[2 * g] ? 0.0 : z.x;
    corr[size_t(2 * g + 1) * stride + j] = zero_rows && zero_rows[2 * g + 1] ? 0.0 : z.y;
  }
};

// Silent channels.  A frame of zeros has a zero spectrum, R = 0 / (0 + 1e-10) = 0 in every bin, and the reference's
// correlation row is exactly zero (argmax 0, no peaks).  Here two pairs share one complex transform, whose rounding
// leaves 1e-17 of the OTHER pair in that row: the rows of pairs with a silent microphone are therefore forced to zero.
// The same pass looks for non-finite samples.  The reference confines a NaN to the pairs of its own microphone; here the
// pair packed into the same complex transform would be poisoned as well, so a non-finite frame sets bit 2 of the
// engine's status word and the call is reported as PAL_ERR_INVALID by pal_synchronize (never silently wrong rows).
__global__ __launch_bounds__(256) void k_row_nonzero(const double* __restrict__ frames, size_t frame_stride, int len, int* __restrict__ flags,
                                                     int* __restrict__ status) {
  const double* x = frames + size_t(blockIdx.x) * frame_stride;
  bool any = false, bad = false;
  for (int i = threadIdx.x; i < len; i += 256) {
    const double v = x[i];
    any = any || v != 0.0;
    bad = bad || !(v - v == 0.0);                             // NaN or infinity
  }
  const int all = __syncthreads_or(any ? 1 : 0);
  const int nonfinite = __syncthreads_or(bad ? 1 : 0);
  if (threadIdx.x == 0) {
    flags[blockIdx.x] = all;
    if (nonfinite && status) atomicOr(status + 2, 1);
  }
}

// explicit pair list (row indices, two per pair) -> the packed table of the pair pipeline: two pairs per complex
// transform, (c, d) = (-1, -1) when the count is odd.  A row index outside 0..R-1 sets bit 1 of status word 2 and the
// pair is redirected to row 0 (no out-of-range read; pal_synchronize reports PAL_ERR_INVALID).
__global__ __launch_bounds__(256) void k_pairs_to_quads(const int32_t* __restrict__ pairs, int64_t P, int R, int4* __restrict__ quads,
                                                        int* __restrict__ status) {
  const int64_t g = int64_t(blockIdx.x) * 256 + threadIdx.x;
  if (g >= (P + 1) / 2) return;
  int v[4];
  bool bad = false;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int64_t at = 4 * g + q;
    v[q] = at < 2 * P ? pairs[at] : -1;
    if (at < 2 * P && (v[q] < 0 || v[q] >= R)) { bad = true; v[q] = 0; }
  }
  quads[g] = make_int4(v[0], v[1], v[2], v[3]);
  if (bad && status) atomicOr(status + 2, 2);
}

__global__ __launch_bounds__(256) void k_pair_zero(const int4* __restrict__ quads, const int* __restrict__ nonzero, int64_t ntr,
                                                   int* __restrict__ zero_rows) {
  const int64_t g = int64_t(blockIdx.x) * 256 + threadIdx.x;
  if (g >= ntr) return;
  const int4 q = quads[g];
  zero_rows[2 * g] = !nonzero[q.x] || !nonzero[q.y];
  zero_rows[2 * g + 1] = q.z >= 0 && (!nonzero[q.z] || !nonzero[q.w]);
}

// ---- the pairs the finishing column pass flagged (pfa_cols_fin.h): their TRANSFORMS, in order, with their original packing
// (a flagged pair is resolved beside the same partner pair as in the first pass, so its record does not depend on which other
// pairs of the call were flagged - the last bits of cmax / cmin / snr depend on the partner) ----
// one workgroup walks the transforms in tiles of 1024 (a stable compaction)
__global__ __launch_bounds__(1024) void k_flag_list(const int* __restrict__ need, int64_t npairs, int* __restrict__ list, int* __restrict__ count) {
  __shared__ int wsum[16];
  __shared__ int base;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t ntr = (npairs + 1) / 2;
  if (tid == 0) base = 0;
  __syncthreads();
  for (int64_t g0 = 0; g0 < ntr; g0 += 1024) {
    const int64_t g = g0 + tid;
    const bool f = g < ntr && (need[2 * g] != 0 || (2 * g + 1 < npairs && need[2 * g + 1] != 0));
    const unsigned long long mask = __ballot(f);
    if (lane == 0) wsum[wave] = __popcll(mask);
    __syncthreads();
    int before = base;
    for (int w = 0; w < wave; ++w) before += wsum[w];
    if (f) list[before + __popcll(mask & ((1ull << lane) - 1ull))] = int(g);
    __syncthreads();
    if (tid == 0) { int t = 0; for (int w = 0; w < 16; ++w) t += wsum[w]; base += t; }
    __syncthreads();
  }
  if (tid == 0) *count = base;
}

__global__ __launch_bounds__(256) void k_flag_quads(const int4* __restrict__ quads, const int* __restrict__ list, int count, int4* __restrict__ out) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < count) out[i] = quads[list[i]];
}

__global__ __launch_bounds__(256) void k_flag_scatter(const pal_pair_record* __restrict__ src, const int* __restrict__ list, int count,
                                                      const int* __restrict__ need, int64_t npairs, pal_pair_record* __restrict__ table) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= 2 * count) return;
  const int64_t p = 2 * int64_t(list[i >> 1]) + (i & 1);
  if (p < npairs && need[p]) table[p] = src[i];
}

// ------------------------------------------------------------------ plans
const cd* Engine::stage_table(int ln) {
  if (!stage_tw[ln]) {
    cd* p = nullptr;
    if (hipMalloc(&p, sizeof(cd) << ln) != hipSuccess) return nullptr;
    (void)hipMemsetAsync(p, 0, sizeof(cd) << ln, stream);
    k_make_stage_tw<<<dim3(((1 << ln) + 255) / 256), dim3(256), 0, stream>>>(p, ln, false);
    stage_tw[ln] = p;
  }
  return stage_tw[ln];
}

const cd* Engine::stage_table_compact(int ln) {
  if (!stage_twc[ln]) {
    cd* p = nullptr;
    if (hipMalloc(&p, sizeof(cd) << ln) != hipSuccess) return nullptr;
    (void)hipMemsetAsync(p, 0, sizeof(cd) << ln, stream);
    k_make_stage_tw<<<dim3(((1 << ln) + 255) / 256), dim3(256), 0, stream>>>(p, ln, true);
    stage_twc[ln] = p;
  }
  return stage_twc[ln];
}

// Convolution geometry for at least `needed` points: the smaller of 2^k and 3 * 2^k (k >= 12 for the latter, so
// that a transform is a whole number of 4096-point row tiles).
int Engine::alloc_conv(Conv& c, size_t needed) {
  int lm = ceil_log2(needed);
  if (lm < 12) lm = 12;
  if (lm > 22) return fail(PAL_ERR_UNSUPPORTED, "convolution of %zu points exceeds 2^22", needed);
  c.r3 = false;
  c.m = size_t(1) << lm;
  c.l2 = lm <= 19 ? (lm - 6 < 10 ? lm - 6 : 10) : 11;
  c.l1 = lm - c.l2;
  const int k3 = lm - 2;                                   // 3 * 2^(lm-2) = 0.75 * 2^lm
  if (allow_r3 && k3 >= 12 && (size_t(3) << k3) >= needed) {
    int l2 = k3 - 4 < 10 ? k3 - 4 : 10;
    int l1 = k3 - l2;
    if (l1 > 8) { l2 = 11; l1 = k3 - l2; }
    if (l1 >= 4 && l1 <= 8) {
      c.r3 = true;
      c.m = size_t(3) << k3;
      c.l1 = l1;
      c.l2 = l2;
    }
  }
  // Register-resident rows (conv_kernels.h k_colsreg_* / k_rowsreg) where the columns then fit one lane comfortably: rows
  // of 4096 points with 12 ... 24-point columns, or rows of 8192 points with 6 ... 24-point columns (measured on C2 / C3 / C5
  // and at n = 88 203 with the prime-factor route off: +13 ... +19 % over the LDS-tile passes either way, the shorter rows
  // ahead where both fit; 32- and 48-point columns need 250 registers in one lane, so they are shared by two neighbouring
  // lanes (k_colsreg2_*, rows of 8192 points: C4's 393 216 = 48 x 8192, +5 %).  A column length only needs a DFT that fits one lane - 2^a, 3 * 2^a, or 2 x 9 / 10 / 11
  // (reg_fft.h reg_dft) - so the convolution length is the smallest M1 x 2^12 / 2^13 that holds the sequence, in steps
  // of about 10 % instead of the 2^k / 3 * 2^k ladder (88 203 points: 22 x 8192 = 180 224 instead of 196 608).
  // PAL_FOUR_REG = 12 / 13 forces a row length (columns up to 48 points then), 0 turns the route off.
  c.reg = false;
  if (four_reg != 0) {
    static const int kCols[] = {6, 8, 12, 16, 18, 20, 22, 24, 32, 48};
    size_t best_m = 0;
    int best_lr = 0, best_m1 = 0;
    for (int lr = 12; lr <= 13; ++lr) {
      if ((four_reg == 12 || four_reg == 13) && lr != four_reg) continue;
      for (int m1 : kCols) {
        if (lr == 12 && m1 < 12) continue;
        if (m1 > 24 && lr != 13) continue;                 // 32- / 48-point columns: two lanes per column, rows of 8192
        const size_t m = size_t(m1) << lr;
        if (m < needed || m > c.m) continue;               // (never longer than the 2^k / 3 * 2^k choice)
        if (!best_m || m < best_m) { best_m = m; best_lr = lr; best_m1 = m1; }   // ties: the shorter rows (lr = 12 comes first)
      }
    }
    if (best_m) {
      c.reg = true;
      c.m = best_m;
      c.m1r = best_m1;
      c.l2 = best_lr;
      c.l1 = 0;
      c.r3 = false;
      if (!stage_table(best_lr)) return fail(PAL_ERR_NOMEM, "twiddle tables");
    }
  }
  if (!c.reg && (!stage_table(c.l1) || !stage_table(c.l2))) return fail(PAL_ERR_NOMEM, "twiddle tables");
  PAL_HIP(hipMalloc(&c.chat, c.m * sizeof(cd)));
  PAL_HIP(hipMalloc(&c.twA, sizeof(cd) * c.M1()));
  PAL_HIP(hipMalloc(&c.twB, sizeof(cd) << c.l2));
  k_make_roots<<<dim3((c.M1() + 255) / 256), dim3(256), 0, stream>>>(c.twA, c.M1(), double(c.M1()));
  k_make_roots<<<dim3(((1 << c.l2) + 255) / 256), dim3(256), 0, stream>>>(c.twB, 1 << c.l2, double(c.m));
  return check(hipGetLastError(), "k_make_roots");
}

void Engine::free_conv(Conv& c) {
  if (c.chat) (void)hipFree(c.chat);
  if (c.twA) (void)hipFree(c.twA);
  if (c.twB) (void)hipFree(c.twB);
  c = Conv();
}

int Engine::build_conv(Conv& c, const cd* w, int n, int neg_count, int pos_count, bool conj_kernel,
                       double extra_scale) {
  Engine* e = this;
  PAL_TRY(alloc_conv(c, size_t(neg_count) + size_t(pos_count) - 1));
  // chirp spectrum: column FFTs + twiddle, then forward row FFTs, scaled, in [k1][k2] order
  ChirpLoader ld{w, neg_count, pos_count, unsigned(c.M()), conj_kernel};
  PAL_TRY(launch_cols_fwd(e, c, 1, ld, c.chat));
  PAL_TRY(launch_rows(e, c, 1, c.chat, false, extra_scale / double(c.M())));
  return PAL_OK;
}

void Engine::free_plan(Plan& pl) {
  if (pl.w) (void)hipFree(pl.w);
  free_conv(pl.fwd);
  free_conv(pl.inv);
  free_pfa(pl.pfa);
  pl = Plan();
}

// every plan and the cached synchronisation convolutions; the streams are drained first
int Engine::clear_plans() {
  PAL_HIP(hipStreamSynchronize(stream));
  PAL_HIP(hipStreamSynchronize(stream2));
  PAL_HIP(hipStreamSynchronize(stream3));
  for (auto& kv : plans) free_plan(kv.second);
  plans.clear();
  for (auto& kv : xconvs) free_conv(kv.second.conv);
  xconvs.clear();
  return PAL_OK;
}

// convolution geometry of the plain cross-correlation of pal_xcorr_vs_ref for sequences of `len` points, kept per
// length (at most max_plans of them, least recently used out first)
int Engine::xcorr_conv(size_t len, Conv** out) {
  auto it = xconvs.find(len);
  if (it == xconvs.end()) {
    if (int(xconvs.size()) >= max_plans) {
      auto old = xconvs.begin();
      for (auto k = xconvs.begin(); k != xconvs.end(); ++k)
        if (k->second.used < old->second.used) old = k;
      PAL_HIP(hipStreamSynchronize(stream));
      free_conv(old->second.conv);
      xconvs.erase(old);
    }
    XConv x;
    const int rc = alloc_conv(x.conv, len);
    if (rc != PAL_OK) { free_conv(x.conv); return rc; }
    it = xconvs.emplace(len, x).first;
  }
  it->second.used = ++plan_clock;
  *out = &it->second.conv;
  return PAL_OK;
}

int Engine::get_plan(int n, int lin, int nout, Plan** out) {
  auto key = std::make_tuple(n, lin, nout);
  auto it = plans.find(key);
  if (it != plans.end()) {
    it->second.used = ++plan_clock;
    *out = &it->second;
    return PAL_OK;
  }
  if (n < 1 || lin < 1 || lin > n || nout < 1 || nout > n)
    return fail(PAL_ERR_INVALID, "bad transform geometry n=%d lin=%d nout=%d", n, lin, nout);
  // The cache is bounded (recordings of many different lengths, ragged pairs, per-frame lengths behind the
  // synchronisation): the least recently used plan goes once every stream has drained (PAL_MAX_PLANS, default 32).
  while (int(plans.size()) >= max_plans) {
    auto old = plans.begin();
    for (auto k = plans.begin(); k != plans.end(); ++k)
      if (k->second.used < old->second.used) old = k;
    PAL_HIP(hipStreamSynchronize(stream));
    PAL_HIP(hipStreamSynchronize(stream2));
    PAL_HIP(hipStreamSynchronize(stream3));
    free_plan(old->second);
    plans.erase(old);
    ++plans_evicted;
  }
  Plan pl;
  pl.n = n;
  pl.H = n / 2 + 1;
  pl.lin = lin;
  pl.nout = nout;
  int rc = PAL_OK;
  do {
    if ((rc = check(hipMalloc(&pl.w, size_t(n) * sizeof(cd)), "chirp alloc")) != PAL_OK) break;
    k_make_chirp<<<dim3((n + 255) / 256), dim3(256), 0, stream>>>(pl.w, n, 1);
    if ((rc = check(hipGetLastError(), "k_make_chirp")) != PAL_OK) break;
    // forward: j < lin inputs, k < H outputs, kernel w_{k-j}
    if ((rc = build_conv(pl.fwd, pl.w, n, lin, pl.H, false, 1.0)) != PAL_OK) break;
    // inverse: k < n inputs, m < nout outputs, kernel conj(w_{m-k}); 1/n of numpy.fft.ifft folded in
    if ((rc = build_conv(pl.inv, pl.w, n, n, nout, true, 1.0 / double(n))) != PAL_OK) break;
    if ((rc = build_pfa(pl)) != PAL_OK) break;   // prime-factor route of the PHAT inverse, when n splits
    rc = check(hipStreamSynchronize(stream), "plan setup");
  } while (0);
  if (rc != PAL_OK) {                            // nothing of a half-built plan stays behind
    const std::string keep = err;
    (void)hipStreamSynchronize(stream);
    free_plan(pl);
    err = keep;
    return rc;
  }
  pl.used = ++plan_clock;
  ++plans_built;
  auto ins = plans.emplace(key, pl);
  *out = &ins.first->second;
  return PAL_OK;
}

// ------------------------------------------------------------------ pipelines
int Engine::forward_spectra(Plan& pl, const double* frames, size_t frame_stride, int rows, int len, cd* spectra, int* nonzero) {
  Engine* e = this;
  if (len > pl.lin) return fail(PAL_ERR_INVALID, "frame length %d exceeds plan input length %d", len, pl.lin);
  if (nonzero && rows > 0) {
    void* stp = nullptr;
    PAL_TRY(scratch(7, 64, &stp));
    k_row_nonzero<<<dim3(rows), dim3(256), 0, stream>>>(frames, frame_stride, len, nonzero, static_cast<int*>(stp));
    PAL_HIP(hipGetLastError());
  }
  if (pfa_forward_applies(pl, len)) return pfa_forward_spectra(pl, frames, frame_stride, rows, len, spectra);
  const Conv& c = pl.fwd;
  void* wsp = nullptr;
  PAL_TRY(scratch(0, size_t(chunk) * c.M() * sizeof(cd), &wsp));
  cd* W = static_cast<cd*>(wsp);
  for (int r0 = 0; r0 < rows; r0 += chunk) {
    const int G = rows - r0 < chunk ? rows - r0 : chunk;
    FrameLoader ld{frames + size_t(r0) * frame_stride, frame_stride, len, pl.w};
    PAL_TRY(launch_cols_fwd(e, c, G, ld, W));
    PAL_TRY(launch_rows(e, c, G, W, true, 1.0));
    if (pl.pfa.on()) {   // the prime-factor inverse reads the permuted layout (spec_stride() elements per row)
      const Pfa& f = pl.pfa;
      PermSpectrumStorer st{spectra + size_t(r0) * pl.spec_stride(), pl.H, f.n1, f.n2, f.rows(), 1.0f / float(f.n1),
                            1.0f / float(f.n2), pl.w, f.rader ? f.rd_qidx : nullptr};
      PAL_TRY(launch_cols_inv(e, c, G, W, st));
    } else {
      SpectrumStorer st{spectra + size_t(r0) * pl.H, pl.H, pl.w};
      PAL_TRY(launch_cols_inv(e, c, G, W, st));
    }
  }
  return PAL_OK;
}

// explicit pair list over R equal-length rows, everything in HBM: one-vs-many bootstrap batches, sparse pair sets, and
// the contiguous pair blocks a rank owns when ONE large frame is split over the GPUs (spectra recomputed locally)
int Engine::pairs_dev(const double* d_rows, int R, int L, const int32_t* d_pairs, int64_t P, const pal_phat_params& prm,
                      pal_pair_record* d_table) {
  if (R < 1 || L < 1 || P < 1) return fail(PAL_ERR_INVALID, "need R >= 1, L >= 1, P >= 1");
  if (L > (1 << 20)) return fail(PAL_ERR_UNSUPPORTED, "frame length %d exceeds 2^20", L);
  Plan* pl = nullptr;
  PAL_TRY(get_plan(2 * L - 1, L, 2 * L - 1, &pl));
  void *sp = nullptr, *dq = nullptr, *stp = nullptr;
  PAL_TRY(scratch(2, size_t(R) * pl->spec_stride() * sizeof(cd) + size_t(R) * sizeof(int), &sp));
  int* nonzero = reinterpret_cast<int*>(static_cast<cd*>(sp) + size_t(R) * pl->spec_stride());
  const int64_t ntr = (P + 1) / 2;
  PAL_TRY(scratch(3, size_t(ntr) * sizeof(int4), &dq));
  PAL_TRY(scratch(7, 64, &stp));
  k_pairs_to_quads<<<dim3(unsigned((ntr + 255) / 256)), dim3(256), 0, stream>>>(d_pairs, P, R, static_cast<int4*>(dq), static_cast<int*>(stp));
  PAL_HIP(hipGetLastError());
  PAL_TRY(forward_spectra(*pl, d_rows, size_t(L), R, L, static_cast<cd*>(sp), nonzero));
  return pair_correlations(*pl, static_cast<const cd*>(sp), R, static_cast<const int4*>(dq), P, L, prm, d_table, nullptr, nullptr, nonzero);
}

int Engine::pair_correlations(Plan& pl, const cd* spectra, int nspec, const int4* quads, int64_t npairs, int n2,
                              const pal_phat_params& prm, pal_pair_record* table, int32_t* ksel_multi,
                              double* corr_out, const int* nonzero) {
  Engine* e = this;
  const Conv& c = pl.inv;
  const int n = pl.n;
  const bool pfa = pl.pfa.on();
  const int chunk = pair_group(n);   // (shadows the engine-wide group size: larger groups amortise launch tails and the peak kernels' fixed costs)
  const cd* permuted = spectra;   // forward_spectra wrote the (k mod N1, k mod N2) layout when the plan has the split
  (void)nspec;
  // Launch groups alternate between two HIP streams, each with its own workspace and correlation buffer: the
  // memory-bound head and tail of one group's kernels overlap the LDS/VALU-bound middle of the other's, and the
  // peak kernel of group g runs beside the FFT passes of group g+1.  Same-slot reuse is ordered by the stream.
  const bool two = (overlap == 1 || overlap == 3) && table != nullptr;
  const int nslot = two ? (overlap == 3 ? 3 : 2) : 1;
  const bool split = overlap == 2 && table != nullptr;        // transforms on `stream`, peak selection on `stream2`
  const size_t wpoints = size_t(chunk) * (pfa ? size_t(n) : c.M());
  void* wsp = nullptr;
  PAL_TRY(scratch(0, size_t(nslot) * wpoints * sizeof(cd), &wsp));
  cd* W = static_cast<cd*>(wsp);
  const size_t stride = corr_out ? size_t(n) : (size_t(n) + 1) & ~size_t(1);
  const size_t buf_doubles = size_t(2 * chunk) * stride;
  void* p = nullptr;
  PAL_TRY(scratch(1, size_t(nslot > 2 ? nslot : 2) * buf_doubles * sizeof(double), &p));
  double* cbuf = static_cast<double*>(p);
  const int64_t ntr = (npairs + 1) / 2;
  int* zero_rows = nullptr;
  if (nonzero && ntr > 0) {
    void* zp = nullptr;
    PAL_TRY(scratch(14, size_t(2 * ntr) * sizeof(int), &zp));
    zero_rows = static_cast<int*>(zp);
    k_pair_zero<<<dim3(unsigned((ntr + 255) / 256)), dim3(256), 0, stream>>>(quads, nonzero, ntr, zero_rows);
    PAL_HIP(hipGetLastError());
  }
  // the finishing column pass (pfa_cols_fin.h) writes one flag per pair: 1 = resolved at the end of this call from stored rows
  const bool fin = table && !split && !corr_out && !ksel_multi &&
                   (pfa ? pfa_sub == 0 && pfa_can_finish(pl, prm) : fourstep_can_finish(pl, prm));
  // stored rows + per-wavefront statistics (pfa_fin_lean.h with FinArgs.corr): the caller wants corr, or the plan has no finishing form
  const bool lean = !fin && pfa && table && !split && !ksel_multi && pfa_sub == 0 && pfa_can_lean_store(pl, prm);
  // rows of any other route: their statistics in one launch over the stored rows (k_rows_lean)
  // (calls of at least 200 000 pairs: the end-of-call count and the flagged rows' second pass - 1.8 % of the rows at C5's lag window -
  //  cost the 30 000 - 80 000-pair calls of the stream chain more than the launch saves, and stall its host: 480 - 497 against 527 - 535 frames/s)
  const bool rlean = !fin && !lean && table && !split && !ksel_multi && npairs >= rows_lean_min && rows_can_lean(pl, prm) &&
                     !(pfa && pfa_sub == 0 && pfa_can_fuse(pl));
  int* need = nullptr;
  if (fin || lean || rlean) {
    void* np = nullptr;
    PAL_TRY(scratch(19, (size_t(2 * npairs) + 64) * sizeof(int), &np));     // [flags | list | count]
    need = static_cast<int*>(np);
  }
  if (two || split) {   // the other streams start after everything already queued on `stream` (spectra, pair table)
    PAL_HIP(hipEventRecord(ev_corr[0], stream));
    PAL_HIP(hipStreamWaitEvent(stream2, ev_corr[0], 0));
    if (nslot == 3) PAL_HIP(hipStreamWaitEvent(stream3, ev_corr[0], 0));
  }
  // one launch group; a failure leaves through the common exit below (side streams joined, sampling gate restored)
  auto run_group = [&](int64_t t0, int64_t group) -> int {
    const int G = int(ntr - t0 < chunk ? ntr - t0 : chunk);
    const int64_t p0 = 2 * t0;
    const int rows = int(npairs - p0 < 2 * G ? npairs - p0 : 2 * G);
    const int slot = two ? int(group % nslot) : (split ? int(group & 1) : 0);
    hipStream_t on = two ? (slot == 0 ? stream : (slot == 1 ? stream2 : stream3)) : stream;
    prof_gate = prof_every <= 1 || (prof_tick++ % prof_every) == 0;
    cd* Wg = W + (two ? size_t(slot) * wpoints : 0);
    // odd tail with a caller buffer: the imaginary half of the last transform has no destination row there
    const bool via_scratch = !corr_out || rows < 2 * G;
    double* crow = via_scratch ? cbuf + size_t(slot) * buf_doubles : corr_out + size_t(p0) * stride;
    if (split && group >= 2) PAL_HIP(hipStreamWaitEvent(stream, ev_peaks[slot], 0));   // group - 2 is done with this buffer
    const bool fused = pfa && table && !split && pfa_sub == 0 && pfa_can_fuse(pl);
    if (lean) {
      PAL_TRY(pfa_pair_group_fin(pl, permuted, quads + t0, G, rows, Wg, zero_rows ? zero_rows + p0 : nullptr, prm, n2, table + p0, need + p0,
                                 slot, on, crow, stride));
    } else if (fin && pfa) {
      // nobody reads the correlation rows: the column pass finishes them without storing them (pfa_cols_fin.h)
      PAL_TRY(pfa_pair_group_fin(pl, permuted, quads + t0, G, rows, Wg, zero_rows ? zero_rows + p0 : nullptr, prm, n2, table + p0, need + p0,
                                 slot, on));
    } else if (fused) {
      PAL_TRY(pfa_pair_group_fused(pl, permuted, quads + t0, G, rows, Wg, crow, stride, zero_rows ? zero_rows + p0 : nullptr, prm, n2, table + p0,
                                   ksel_multi ? ksel_multi + p0 * PAL_MAX_PEAKS : nullptr, on));
    } else if (pfa) {
      // sub-groups: the Y of a sub-group (1.4 MB per transform) is still in the Infinity Cache when the column
      // pass reads it back, while the peak kernels keep whole launch groups (their fixed costs want many rows)
      const int sub = pfa_sub > 0 && pfa_sub < G ? pfa_sub : G;
      for (int g0 = 0; g0 < G; g0 += sub) {
        const int Gs = G - g0 < sub ? G - g0 : sub;
        PAL_TRY(pfa_pair_group(pl, permuted, quads + t0 + g0, Gs, Wg, crow + size_t(2 * g0) * stride, stride,
                               zero_rows ? zero_rows + p0 + 2 * g0 : nullptr, on));
      }
    } else {
      PairLoader ld{spectra, quads + t0, n, pl.H, pl.w};
      PAL_TRY(launch_cols_fwd(e, c, G, ld, Wg, on));
      PAL_TRY(launch_rows(e, c, G, Wg, true, 1.0, on));
      if (fin) {
        // the last pass finishes its rows itself: no correlation rows in HBM, no statistics launches (pfa_cols_fin.h)
        PAL_TRY(fourstep_pair_group_fin(pl, Wg, G, rows, zero_rows ? zero_rows + p0 : nullptr, prm, n2, table + p0, need + p0, slot, on));
      } else {
        CorrStorer st{crow, stride, n, pl.w, zero_rows ? zero_rows + p0 : nullptr};
        PAL_TRY(launch_cols_inv(e, c, G, Wg, st, on));
      }
    }
    if (corr_out && via_scratch)
      PAL_HIP(hipMemcpyAsync(corr_out + size_t(p0) * stride, crow, size_t(rows) * stride * sizeof(double),
                             hipMemcpyDeviceToDevice, on));
    hipStream_t pon = on;
    if (split) {
      PAL_HIP(hipEventRecord(ev_corr[slot], stream));
      PAL_HIP(hipStreamWaitEvent(stream2, ev_corr[slot], 0));
      pon = stream2;
    }
    if (rlean)
      PAL_TRY(rows_lean_group(pl, crow, stride, G, rows, prm, n2, table + p0, need + p0, slot, pon));
    else if (table && !fused && !fin && !lean)
      PAL_TRY(peaks(crow, stride, rows, n, n2, prm, table + p0, ksel_multi ? ksel_multi + p0 * PAL_MAX_PEAKS : nullptr, pon));
    if (split) PAL_HIP(hipEventRecord(ev_peaks[slot], stream2));
    return PAL_OK;
  };
  int64_t group = 0;
  for (int64_t t0 = 0; t0 < ntr; t0 += chunk, ++group) {
    const int rc = run_group(t0, group);
    if (rc != PAL_OK) {
      // work may still be running on the side streams: nothing that follows (frees, reuse of the workspaces, the
      // caller's next call) may overtake it
      prof_gate = true;
      const std::string keep = err;
      (void)hipStreamSynchronize(stream);
      (void)hipStreamSynchronize(stream2);
      (void)hipStreamSynchronize(stream3);
      err = keep;
      return rc;
    }
  }
  prof_gate = true;
  if (two || split) {   // whatever follows on `stream` (downloads, the RCCL gather) sees the finished table
    PAL_HIP(hipEventRecord(ev_peaks[0], stream2));
    PAL_HIP(hipStreamWaitEvent(stream, ev_peaks[0], 0));
    if (nslot == 3) {
      PAL_HIP(hipEventRecord(ev_join3, stream3));
      PAL_HIP(hipStreamWaitEvent(stream, ev_join3, 0));
    }
  }
  if (fin || lean || rlean) {
    // The pairs the finishing blocks flagged (a threshold comparison inside the median's interval, a tie, a window peak next to
    // the window's edge, ...) go through the stored-row path now, packed in pair order.  The count comes to the host: this
    // is the one synchronisation of the call (its launch groups above never waited for the host).
    int* list = need + npairs;
    int* dcount = need + 2 * npairs;
    k_flag_list<<<dim3(1), dim3(1024), 0, stream>>>(need, npairs, list, dcount);
    PAL_HIP(hipGetLastError());
    int count = 0;
    PAL_HIP(hipMemcpyAsync(&count, dcount, sizeof count, hipMemcpyDeviceToHost, stream));
    PAL_HIP(hipStreamSynchronize(stream));
    if (count > 0) {                                           // `count` transforms hold a flagged pair
      void *qp = nullptr, *tp = nullptr;
      PAL_TRY(scratch(20, size_t(count) * sizeof(int4), &qp));
      PAL_TRY(scratch(21, size_t(2 * count) * sizeof(pal_pair_record), &tp));
      k_flag_quads<<<dim3(unsigned((count + 255) / 256)), dim3(256), 0, stream>>>(quads, list, count, static_cast<int4*>(qp));
      PAL_HIP(hipGetLastError());
      const bool keep = fin_cols;
      fin_cols = false;
      const int rc = pair_correlations(pl, spectra, nspec, static_cast<const int4*>(qp), int64_t(2) * count, n2, prm, static_cast<pal_pair_record*>(tp),
                                       nullptr, nullptr, nonzero);
      fin_cols = keep;
      PAL_TRY(rc);
      k_flag_scatter<<<dim3(unsigned((2 * count + 255) / 256)), dim3(256), 0, stream>>>(static_cast<const pal_pair_record*>(tp), list, count, need,
                                                                                      npairs, table);
      PAL_HIP(hipGetLastError());
    }
  }
  return PAL_OK;
}

}  // namespace pal
