// conv_kernels.h - the three workgroup passes of one circular convolution of length M = M1 x M2
// (see bluestein.hip for the algorithm) as templates over loader / storer functors, shared by the
// PHAT pipeline (bluestein.hip) and the multipath / synchronisation pipelines (sim.hip).
//
// M2 (rows) is always a power of two.  M1 (columns) is 2^l1, or 3 * 2^l1 when the convolution length
// is 3 * 2^k: all BASELINE frame lengths need 2n-1 = 0.73 * 2^k points, so the 3 * 2^(k-1) length cuts
// a quarter of the HBM traffic of every pass.  The radix-3 column kernels run 192 lanes (3 wavefronts)
// over 3072 points: one 3-point butterfly stage (fft_core.h radix3_item) around the same power-of-two
// sub-transforms.
//
// Every pass feeds the first FFT stage straight from global memory (or from the loader functor) and lets
// the last stage write global memory (or call the storer functor): the only LDS traffic is the exchange
// between stages.
#pragma once
#include "engine.h"
#include "reg_fft.h"

namespace pal {

__global__ void k_make_chirp(cd* w, int n, int mult);
__global__ void k_make_roots(cd* out, int count, double denom);
__global__ void k_make_stage_tw(cd* out, int ln, bool compact);

// Which (transform g, row or tile k1) a workgroup of a pass takes whose work list is row-major, transform-minor: all G
// workgroups of one row read the same shared data (a row of every microphone's spectrum in the prime-factor row passes,
// the spectrum elements of a column tile in the four-step loaders, a tile of the chirp spectrum in k_rows).  Workgroups
// are dealt to the 8 XCDs round-robin (blockIdx % 8), so in plain order every XCD's L2 ends up fetching all of it; with
// `xcd` the XCD x takes the x-th contiguous eighth of the list instead (5-6 rows of 45 at the metric length: each
// spectrum row crosses the fabric once per launch, not once per XCD).  The grid is 8 * ceil(total / 8); false = nothing
// to do (wave-uniform, before any barrier).
__device__ __forceinline__ bool row_work_item(unsigned b, int G, int rows, int xcd, int& g, int& k1) {
  const unsigned total = unsigned(G) * unsigned(rows);
  unsigned w = b;
  if (xcd) {
    const unsigned per = (total + 7u) >> 3;
    w = (b & 7u) * per + (b >> 3);
  }
  g = int(w % unsigned(G));
  k1 = int(w / unsigned(G));
  return w < total;
}
__host__ inline unsigned row_work_grid(int G, int rows, int xcd) {
  const unsigned total = unsigned(G) * unsigned(rows);
  return xcd ? 8u * ((total + 7u) >> 3) : total;
}

// exp(-2 pi i e / M) for e < M from the two root tables (e = q * M2 + r)
__device__ __forceinline__ cd four_step_twiddle(unsigned e, int l2, const cd* __restrict__ twA, const cd* __restrict__ twB) {
  return cmul(twA[e >> l2], twB[e & ((1u << l2) - 1)]);
}

// R = S_a conj(S_b);  R /= |R| + 1e-10        (utils.py:116-117)
// The magnitude and the reciprocal use the hardware estimates (v_rsq_f64 / v_rcp_f64) plus two Newton steps
// each (relative error ~1e-16) instead of the IEEE-exact sqrt and divide sequences: a third of the loader's
// instructions, and the result only has to be right to the rounding level of the transforms around it.
__device__ __forceinline__ cd whiten(cd a, cd b) {
  const cd r = cmulc(a, b);
  const double m2 = __builtin_fma(r.x, r.x, r.y * r.y);
  double y = __builtin_amdgcn_rsq(m2);
  y = y * __builtin_fma(-0.5 * m2 * y, y, 1.5);
  y = y * __builtin_fma(-0.5 * m2 * y, y, 1.5);
  const double mag = m2 > 0 ? m2 * y : 0.0;               // |R| (rsq(0) is infinite)
  const double d = mag + 1e-10;
  double inv = __builtin_amdgcn_rcp(d);
  inv = inv * __builtin_fma(-d, inv, 2.0);
  inv = inv * __builtin_fma(-d, inv, 2.0);
  return mk(r.x * inv, r.y * inv);
}

// ------------------------------------------------------------------ stage sources / sinks
template <int L2> struct RowsGlobal {             // tile of 4096 consecutive points = 4096 / 2^L2 rows
  static constexpr bool kLds = false;
  cd* base;
  __device__ cd operator()(int t, int e) const { return base[(t << L2) + e]; }
  __device__ void operator()(int t, int e, cd v) const { base[(t << L2) + e] = v; }
};

template <int L2> struct RowsChatToLds {          // pointwise product with the chirp spectrum on the way into LDS
  static constexpr bool kLds = true;
  cd* data;
  const cd* ch;
  __device__ void operator()(int t, int e, cd v) const { data[lds_addr<L2, false>(t, e)] = cmul(v, ch[(t << L2) + e]); }
};

template <int L2> struct RowsScaled {             // forward-only rows (chirp-spectrum setup)
  static constexpr bool kLds = false;
  cd* base;
  double scale;
  __device__ void operator()(int t, int e, cd v) const { base[(t << L2) + e] = cscale(v, scale); }
};

template <class Loader> struct ColsFromLoader {   // column c0 + t, row e of the transform's input
  static constexpr bool kLds = false;
  const Loader& ld;
  int g, l2;
  unsigned c0;
  __device__ cd operator()(int t, int e) const { return ld(g, (unsigned(e) << l2) + c0 + t); }
};

struct ColsToGlobal {                             // column FFT output (row k1 = e) times the four-step twiddle
  static constexpr bool kLds = false;
  cd* out;
  int l2;
  unsigned c0;
  const cd *twA, *twB;
  __device__ void operator()(int t, int e, cd v) const {
    out[(size_t(e) << l2) + c0 + t] = cmul(v, four_step_twiddle((c0 + t) * unsigned(e), l2, twA, twB));
  }
};

struct ColsFromGlobal {                           // inverse: row k1 = e times the conjugate twiddle
  static constexpr bool kLds = false;
  const cd* in;
  int l2;
  unsigned c0;
  const cd *twA, *twB;
  __device__ cd operator()(int t, int e) const {
    return cmulc(in[(size_t(e) << l2) + c0 + t], four_step_twiddle((c0 + t) * unsigned(e), l2, twA, twB));
  }
};

template <class Storer> struct ColsToStorer {
  static constexpr bool kLds = false;
  const Storer& st;
  int g, l2;
  unsigned c0;
  __device__ void operator()(int t, int e, cd v) const { st(g, (unsigned(e) << l2) + c0 + t, v); }
};

// radix-3 column passes: sub-transform t = q * T + c, element e  <->  frequency row k1 = 3e + q
template <int T> struct Cols3ToGlobal {
  static constexpr bool kLds = false;
  cd* out;
  int l2;
  unsigned c0;
  const cd *twA, *twB;
  __device__ void operator()(int t, int e, cd v) const {
    const unsigned k1 = 3u * unsigned(e) + unsigned(t / T), c = c0 + unsigned(t % T);
    out[(size_t(k1) << l2) + c] = cmul(v, four_step_twiddle(c * k1, l2, twA, twB));
  }
};

template <int T> struct Cols3FromGlobal {
  static constexpr bool kLds = false;
  const cd* in;
  int l2;
  unsigned c0;
  const cd *twA, *twB;
  __device__ cd operator()(int t, int e) const {
    const unsigned k1 = 3u * unsigned(e) + unsigned(t / T), c = c0 + unsigned(t % T);
    return cmulc(in[(size_t(k1) << l2) + c], four_step_twiddle(c * k1, l2, twA, twB));
  }
};

// stage-major twiddle table of a 2^LN-point sub-transform: global -> registers -> LDS with every load in flight
// before the first store (a guarded element-wise copy makes the compiler wait for ALL outstanding loads once per
// round).  The device table and the LDS array both hold 2^LN entries (>= stage_tw_size(LN)).
template <int LN, int LANES> __device__ __forceinline__ void copy_tw(cd* tw, const cd* __restrict__ tws, int tid) {
  constexpr int N = 1 << LN;
  if constexpr (N % LANES == 0) {
    cd r[N / LANES];
#pragma unroll
    for (int q = 0; q < N / LANES; ++q) r[q] = tws[tid + q * LANES];
#pragma unroll
    for (int q = 0; q < N / LANES; ++q) tw[tid + q * LANES] = r[q];
  } else {
    for (int i = tid; i < stage_tw_size(LN); i += LANES) tw[i] = tws[i];
  }
}

// ------------------------------------------------------------------ the three passes (M1 = 2^L1)
// Each workgroup (256 lanes) owns 4096 points.  Grid = G * M / 4096, transform index fastest so that
// the workgroups that share chirp-spectrum rows and twiddles run together.
template <int L1, class Loader>
__global__ __launch_bounds__(256) void k_cols_fwd(Loader ld, cd* __restrict__ W, int l2, int G,
                                                  const cd* __restrict__ tws, const cd* __restrict__ twA,
                                                  const cd* __restrict__ twB, int tiles, int xcd) {
  constexpr int N1 = 1 << L1, T = kPoints / N1;
  __shared__ cd data[kPoints];
  __shared__ cd tw[N1];
  const int tid = threadIdx.x;
  int g, tile;
  if (!row_work_item(blockIdx.x, G, tiles, xcd, g, tile)) return;
  const unsigned c0 = unsigned(tile) * T;
  copy_tw<L1, kLanes>(tw, tws, tid);
  wg_fft<L1, true, false, T>(data, tw, tid, ColsFromLoader<Loader>{ld, g, l2, c0},
                             ColsToGlobal{W + (size_t(g) << (L1 + l2)), l2, c0, twA, twB});
}

// rows of length 2^L2; `m` = points per transform (any multiple of 4096)
template <int L2, bool CONV>
__global__ __launch_bounds__(256) void k_rows(cd* __restrict__ W, const cd* __restrict__ chat, size_t m, int G,
                                              const cd* __restrict__ tws, double scale, int tiles, int xcd) {
  constexpr int N2 = 1 << L2, T = kPoints / N2;
  __shared__ cd data[kPoints];
  __shared__ cd tw[N2];
  const int tid = threadIdx.x;
  int g, tile_i;
  if (!row_work_item(blockIdx.x, G, tiles, xcd, g, tile_i)) return;
  const size_t tile = size_t(tile_i);                       // 4096 consecutive points = 4096/N2 rows
  copy_tw<L2, kLanes>(tw, tws, tid);
  cd* base = W + size_t(g) * m + tile * kPoints;
  if (CONV) {
    wg_fft<L2, false, false, T>(data, tw, tid, RowsGlobal<L2>{base}, RowsChatToLds<L2>{data, chat + tile * kPoints});
    wg_fft<L2, false, true, T>(data, tw, tid, LdsTile<L2, false, T>{data}, RowsGlobal<L2>{base});
  } else {
    wg_fft<L2, false, false, T>(data, tw, tid, RowsGlobal<L2>{base}, RowsScaled<L2>{base, scale});
  }
}

template <int L1, class Storer>
__global__ __launch_bounds__(256) void k_cols_inv(const cd* __restrict__ W, Storer st, int l2, int G,
                                                  const cd* __restrict__ tws, const cd* __restrict__ twA,
                                                  const cd* __restrict__ twB) {
  constexpr int N1 = 1 << L1, T = kPoints / N1;
  __shared__ cd data[kPoints];
  __shared__ cd tw[N1];
  const int tid = threadIdx.x;
  const int g = blockIdx.x % G;
  const unsigned c0 = (blockIdx.x / G) * T;
  copy_tw<L1, kLanes>(tw, tws, tid);
  wg_fft<L1, true, true, T>(data, tw, tid, ColsFromGlobal{W + (size_t(g) << (L1 + l2)), l2, c0, twA, twB},
                            ColsToStorer<Storer>{st, g, l2, c0});
}

// ------------------------------------------------------------------ column passes for M1 = 3 * 2^LN
// 192 lanes, T = 1024 / 2^LN columns per tile, 3072 points: sub-transform (q, c) of the LDS tile holds row
// q*N + e of column c on the time side and row 3e + q on the frequency side (fft_core.h radix3_item).
constexpr int kPoints3 = 3072, kLanes3 = 192;

template <int LN, class Loader>
__global__ __launch_bounds__(192) void k_cols3_fwd(Loader ld, cd* __restrict__ W, int l2, int G,
                                                   const cd* __restrict__ tws, const cd* __restrict__ twA,
                                                   const cd* __restrict__ twB, int tiles, int xcd) {
  constexpr int N = 1 << LN, T = 1024 / N, NSUB = 3 * T;
  __shared__ cd data[kPoints3];
  __shared__ cd tw[N < 16 ? 16 : N];
  const int tid = threadIdx.x;
  int g, tile;
  if (!row_work_item(blockIdx.x, G, tiles, xcd, g, tile)) return;
  const unsigned c0 = unsigned(tile) * T;
  for (int i = tid; i < stage_tw_size(LN); i += kLanes3) tw[i] = tws[i];
  const auto rows3 = [&](int q, int c, int e) { return ld(g, (unsigned(q * N + e) << l2) + c0 + c); };
  const LdsTile3<LN, T> tile3{data};
  for (int w = tid; w < N * T; w += kLanes3) radix3_item<T, false>(rows3, tile3, twA, w);
  __syncthreads();
  wg_fft<LN, true, false, NSUB>(data, tw, tid, LdsTile<LN, true, NSUB>{data},
                                Cols3ToGlobal<T>{W + size_t(g) * (size_t(3 * N) << l2), l2, c0, twA, twB});
}

template <int LN, class Storer>
__global__ __launch_bounds__(192) void k_cols3_inv(const cd* __restrict__ W, Storer st, int l2, int G,
                                                   const cd* __restrict__ tws, const cd* __restrict__ twA,
                                                   const cd* __restrict__ twB) {
  constexpr int N = 1 << LN, T = 1024 / N, NSUB = 3 * T;
  __shared__ cd data[kPoints3];
  __shared__ cd tw[N < 16 ? 16 : N];
  const int tid = threadIdx.x;
  const int g = blockIdx.x % G;
  const unsigned c0 = (blockIdx.x / G) * T;
  for (int i = tid; i < stage_tw_size(LN); i += kLanes3) tw[i] = tws[i];
  wg_fft<LN, true, true, NSUB>(data, tw, tid, Cols3FromGlobal<T>{W + size_t(g) * (size_t(3 * N) << l2), l2, c0, twA, twB},
                               LdsTile<LN, true, NSUB>{data});
  const LdsTile3<LN, T> tile3{data};
  const auto rows3 = [&](int q, int c, int e, cd v) { st(g, (unsigned(q * N + e) << l2) + c0 + c, v); };
  for (int w = tid; w < N * T; w += kLanes3) radix3_item<T, true>(tile3, rows3, twA, w);
}

// ------------------------------------------------------------------ the three passes with 8192-point rows (M = M1 x 8192)
// Convolution lengths from 49 152 to 393 216 points (frames of 12 000 ... 96 000 samples) can be cut with rows of 8192
// points, which leaves columns of only M1 = 6 ... 48 points:
//   - a column then fits the registers of ONE lane (reg_fft.h reg_dft): lane = column, 256 consecutive columns per
//     workgroup, every load and store of a wavefront is one contiguous 1 KB request, no LDS, no barrier - the column
//     passes are plain streaming kernels with M1 loads in flight per lane (against 16-column tiles of 192 rows moved
//     through LDS by 192-lane workgroups, three per CU, whose wavefronts waited on memory 65-70 % of the time);
//   - a row is one register-resident 8192-point transform (reg_fft.h big_fft: 256 lanes x 32 points, LDS only for the
//     stage exchanges, two workgroups per CU), forward x chirp spectrum x inverse without leaving the registers.
// The four-step twiddle exp(-/+ 2 pi i c k1 / M) of column c is looked up exactly at k1 = 1 and at every fourth k1 and
// completed by at most two multiplications (w^(4j+i) = w^(4j) w^i).
constexpr int kRegCols = 256;

template <int M1, int kRegL2, bool INV>
__device__ __forceinline__ void colsreg_twiddle(cd* v, unsigned c, const cd* __restrict__ twA, const cd* __restrict__ twB) {
  const cd w1 = twB[c];                                        // c < 2^L2: exp(-2 pi i c / M)
  const cd w2 = cmul(w1, w1), w3 = cmul(w2, w1);
#pragma unroll
  for (int k1 = 1; k1 < M1; ++k1) {
    cd w;
    if (k1 < 4) w = k1 == 1 ? w1 : (k1 == 2 ? w2 : w3);
    else {
      const cd anchor = four_step_twiddle(c * unsigned(k1 & ~3), kRegL2, twA, twB);
      w = (k1 & 3) == 0 ? anchor : cmul(anchor, (k1 & 3) == 1 ? w1 : ((k1 & 3) == 2 ? w2 : w3));
    }
    v[k1] = INV ? cmulc(v[k1], w) : cmul(v[k1], w);
  }
}

template <int M1, int kRegL2, class Loader>
__global__ __launch_bounds__(kRegCols) void k_colsreg_fwd(Loader ld, cd* __restrict__ W, int G, const cd* __restrict__ twA,
                                                          const cd* __restrict__ twB, int xcd) {
  int g, tile;
  if (!row_work_item(blockIdx.x, G, (1 << kRegL2) / kRegCols, xcd, g, tile)) return;
  const unsigned c = unsigned(tile) * kRegCols + threadIdx.x;
  cd v[M1];
#pragma unroll
  for (int r = 0; r < M1; ++r) v[r] = ld(g, (unsigned(r) << kRegL2) + c);
  reg_dft<M1, false>(v, twA);
  colsreg_twiddle<M1, kRegL2, false>(v, c, twA, twB);
  cd* out = W + (size_t(g) * M1 << kRegL2) + c;
#pragma unroll
  for (int k1 = 0; k1 < M1; ++k1) out[size_t(k1) << kRegL2] = v[k1];
}

template <int M1, int kRegL2, class Storer>
__global__ __launch_bounds__(kRegCols) void k_colsreg_inv(const cd* __restrict__ W, Storer st, int G, const cd* __restrict__ twA,
                                                          const cd* __restrict__ twB, int xcd) {
  int g, tile;
  if (!row_work_item(blockIdx.x, G, (1 << kRegL2) / kRegCols, xcd, g, tile)) return;
  const unsigned c = unsigned(tile) * kRegCols + threadIdx.x;
  const cd* in = W + (size_t(g) * M1 << kRegL2) + c;
  cd v[M1];
#pragma unroll
  for (int k1 = 0; k1 < M1; ++k1) v[k1] = in[size_t(k1) << kRegL2];
  colsreg_twiddle<M1, kRegL2, true>(v, c, twA, twB);
  reg_dft<M1, true>(v, twA);
#pragma unroll
  for (int r = 0; r < M1; ++r) st(g, (unsigned(r) << kRegL2) + c, v[r]);
}

// Columns of 32 or 48 points (C4: 393 216 = 48 x 8192) need 250 registers in one lane; here TWO neighbouring lanes share a
// column: lane `half` holds the rows 2 j + half on the time side and the rows k + half * M1 / 2 on the frequency side, each
// lane transforms its M1 / 2 points (reg_dft) and ONE exchange with the partner lane (DPP quad_perm [1,0,3,2], one move per
// dword) completes the radix-2 step:  X[k], X[k + M1/2] = E[k] +/- w^k O[k]  (forward);
// x[2 j + p] = IDFT_{M1/2}((X[k] +/- X[k + M1/2]) w^(-p k))[j]  (inverse).  128 columns per workgroup: a wavefront's load or
// store is two contiguous 512-byte requests.
__device__ __forceinline__ double from_partner_lane(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_mov_dpp(lo, 0xB1, 0xf, 0xf, true);            // quad_perm:[1,0,3,2]
  hi = __builtin_amdgcn_mov_dpp(hi, 0xB1, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ cd from_partner_lane(cd v) { return mk(from_partner_lane(v.x), from_partner_lane(v.y)); }

// four-step twiddle of the rows k0 + k, k < MH, of column c (k0 = 0 or MH, a multiple of 4): see colsreg_twiddle
template <int MH, int kRegL2, bool INV>
__device__ __forceinline__ void colsreg_twiddle_from(cd* v, unsigned c, int k0, const cd* __restrict__ twA, const cd* __restrict__ twB) {
  static_assert(MH % 4 == 0, "anchors every fourth row");
  const cd w1 = twB[c];
  const cd w2 = cmul(w1, w1), w3 = cmul(w2, w1);
#pragma unroll
  for (int k = 0; k < MH; ++k) {
    cd w;
    if ((k & 3) == 0) {
      w = four_step_twiddle(c * unsigned(k0 + k), kRegL2, twA, twB);        // (row 0: twA[0] twB[0] = 1 exactly)
    } else {
      const cd anchor = four_step_twiddle(c * unsigned(k0 + (k & ~3)), kRegL2, twA, twB);
      w = cmul(anchor, (k & 3) == 1 ? w1 : ((k & 3) == 2 ? w2 : w3));
    }
    v[k] = INV ? cmulc(v[k], w) : cmul(v[k], w);
  }
}

template <int M1, int kRegL2, class Loader>
__global__ __launch_bounds__(kRegCols) void k_colsreg2_fwd(Loader ld, cd* __restrict__ W, int G, const cd* __restrict__ twA,
                                                           const cd* __restrict__ twB, int xcd) {
  constexpr int MH = M1 / 2, COLS = kRegCols / 2;
  int g, tile;
  if (!row_work_item(blockIdx.x, G, (1 << kRegL2) / COLS, xcd, g, tile)) return;
  const int half = threadIdx.x & 1;
  const unsigned c = unsigned(tile) * COLS + (threadIdx.x >> 1);
  cd v[MH];
#pragma unroll
  for (int j = 0; j < MH; ++j) v[j] = ld(g, (unsigned(2 * j + half) << kRegL2) + c);
  reg_dft<MH, false, 2>(v, twA);                               // E (half 0) or O (half 1)
#pragma unroll
  for (int k = 0; k < MH; ++k) {
    const cd w = uniform_root(twA, k);
    const cd mine = (half && k) ? cmul(v[k], w) : v[k];        // O[k] w^k on the odd lane
    const cd theirs = from_partner_lane(mine);
    v[k] = half ? theirs - mine : mine + theirs;               // X[k] = E + P on the even lane, X[k + MH] = E - P on the odd one
  }
  colsreg_twiddle_from<MH, kRegL2, false>(v, c, half * MH, twA, twB);
  cd* out = W + (size_t(g) * M1 << kRegL2) + c + (size_t(half * MH) << kRegL2);
#pragma unroll
  for (int k = 0; k < MH; ++k) out[size_t(k) << kRegL2] = v[k];
}

template <int M1, int kRegL2, class Storer>
__global__ __launch_bounds__(kRegCols) void k_colsreg2_inv(const cd* __restrict__ W, Storer st, int G, const cd* __restrict__ twA,
                                                           const cd* __restrict__ twB, int xcd) {
  constexpr int MH = M1 / 2, COLS = kRegCols / 2;
  int g, tile;
  if (!row_work_item(blockIdx.x, G, (1 << kRegL2) / COLS, xcd, g, tile)) return;
  const int half = threadIdx.x & 1;
  const unsigned c = unsigned(tile) * COLS + (threadIdx.x >> 1);
  const cd* in = W + (size_t(g) * M1 << kRegL2) + c + (size_t(half * MH) << kRegL2);
  cd v[MH];
#pragma unroll
  for (int k = 0; k < MH; ++k) v[k] = in[size_t(k) << kRegL2];
  colsreg_twiddle_from<MH, kRegL2, true>(v, c, half * MH, twA, twB);
#pragma unroll
  for (int k = 0; k < MH; ++k) {
    const cd theirs = from_partner_lane(v[k]);
    const cd w = uniform_root(twA, k);
    const cd dif = theirs - v[k];                              // odd lane: X[k] - X[k + MH]
    v[k] = half ? (k ? cmulc(dif, w) : dif) : v[k] + theirs;
  }
  reg_dft<MH, true, 2>(v, twA);
#pragma unroll
  for (int j = 0; j < MH; ++j) st(g, (unsigned(2 * j + half) << kRegL2) + c, v[j]);
}

// rows of 8192 points: one workgroup per row.  CONV: forward, x chirp spectrum, inverse, in place; else forward x scale.
template <int kRegL2, bool CONV>
__global__ __launch_bounds__((BigTile<kRegL2, 32>::kLanes)) __attribute__((amdgpu_waves_per_eu(2)))
void k_rowsreg(cd* __restrict__ W, const cd* __restrict__ chat, int rows, int G, const cd* __restrict__ tws, double scale, int xcd) {
  using B = BigTile<kRegL2, 32>;
  constexpr int N2 = 1 << kRegL2, LANES = B::kLanes, PTS = 32;
  __shared__ double plane[N2];
  const int tid = threadIdx.x;
  int g, row;
  if (!row_work_item(blockIdx.x, G, rows, xcd, g, row)) return;
  cd* const base = W + ((size_t(g) * rows + row) << kRegL2);
  cd v[PTS];
#pragma unroll
  for (int s = 0; s < PTS; ++s) v[B::reg_of(s, B::kR0)] = base[tid + LANES * s];
  big_fft<kRegL2, PTS, false>(plane, tws, v, tid);
  if constexpr (CONV) {
    const cd* const ch = chat + (size_t(row) << kRegL2);
    {
      cd u[PTS];
#pragma unroll
      for (int reg = 0; reg < PTS; ++reg) {
        if (reg % 8 == 0) asm volatile("" ::: "memory");      // eight chirp-spectrum loads in flight, not all 32
        const int s = B::slot_of(reg, B::kRL);
        u[B::reg_of(s, B::kR0)] = cmul(v[reg], ch[tid + LANES * s]);
      }
#pragma unroll
      for (int r = 0; r < PTS; ++r) v[r] = u[r];
    }
    {
      // (opaque copies of the table offset and the lane index: seen through the same values the compiler keeps the forward
      //  transform's twiddles and LDS positions live for the inverse one and spills them - pfa_big.h)
      size_t again = 0;
      int tid2 = tid;
      asm volatile("" : "+s"(again), "+v"(tid2));
      big_fft<kRegL2, PTS, true>(plane, tws + again, v, tid2);
    }
#pragma unroll
    for (int reg = 0; reg < PTS; ++reg) base[tid + LANES * B::slot_of(reg, B::kRL)] = v[reg];
  } else {
#pragma unroll
    for (int reg = 0; reg < PTS; ++reg) base[tid + LANES * B::slot_of(reg, B::kRL)] = cscale(v[reg], scale);
  }
}

#define PAL_SWITCH_M1(m1, l2, ...)                                 \
  switch ((l2) * 100 + (m1)) {                                     \
    case 1306: { constexpr int MM = 6, LR = 13; __VA_ARGS__; } break;   \
    case 1308: { constexpr int MM = 8, LR = 13; __VA_ARGS__; } break;   \
    case 1312: { constexpr int MM = 12, LR = 13; __VA_ARGS__; } break;  \
    case 1316: { constexpr int MM = 16, LR = 13; __VA_ARGS__; } break;  \
    case 1318: { constexpr int MM = 18, LR = 13; __VA_ARGS__; } break;  \
    case 1320: { constexpr int MM = 20, LR = 13; __VA_ARGS__; } break;  \
    case 1322: { constexpr int MM = 22, LR = 13; __VA_ARGS__; } break;  \
    case 1324: { constexpr int MM = 24, LR = 13; __VA_ARGS__; } break;  \
    case 1212: { constexpr int MM = 12, LR = 12; __VA_ARGS__; } break;  \
    case 1216: { constexpr int MM = 16, LR = 12; __VA_ARGS__; } break;  \
    case 1218: { constexpr int MM = 18, LR = 12; __VA_ARGS__; } break;  \
    case 1220: { constexpr int MM = 20, LR = 12; __VA_ARGS__; } break;  \
    case 1222: { constexpr int MM = 22, LR = 12; __VA_ARGS__; } break;  \
    case 1224: { constexpr int MM = 24, LR = 12; __VA_ARGS__; } break;  \
    default: return e->fail(PAL_ERR_INTERNAL, "register column pass of %d points, rows of 2^%d", m1, l2); \
  }

// ------------------------------------------------------------------ launch helpers
#define PAL_SWITCH_L(l, ...)                                       \
  switch (l) {                                                     \
    case 6: { constexpr int LL = 6; __VA_ARGS__; } break;          \
    case 7: { constexpr int LL = 7; __VA_ARGS__; } break;          \
    case 8: { constexpr int LL = 8; __VA_ARGS__; } break;          \
    case 9: { constexpr int LL = 9; __VA_ARGS__; } break;          \
    case 10: { constexpr int LL = 10; __VA_ARGS__; } break;        \
    case 11: { constexpr int LL = 11; __VA_ARGS__; } break;        \
    default: return e->fail(PAL_ERR_UNSUPPORTED, "sub-transform log2 size %d outside 6..11", l); \
  }

#define PAL_SWITCH_L3(l, ...)                                      \
  switch (l) {                                                     \
    case 4: { constexpr int LL = 4; __VA_ARGS__; } break;          \
    case 5: { constexpr int LL = 5; __VA_ARGS__; } break;          \
    case 6: { constexpr int LL = 6; __VA_ARGS__; } break;          \
    case 7: { constexpr int LL = 7; __VA_ARGS__; } break;          \
    case 8: { constexpr int LL = 8; __VA_ARGS__; } break;          \
    default: return e->fail(PAL_ERR_UNSUPPORTED, "radix-3 column sub-transform log2 size %d outside 4..8", l); \
  }

template <class Loader>
static int launch_cols_fwd(Engine* e, const Conv& c, int G, Loader ld, cd* W, hipStream_t on = nullptr) {
  if (!on) on = e->stream;
  char name[64];
  if (c.reg) snprintf(name, sizeof name, "k_colsreg_fwd<%d,%s>", c.M1(), Loader::kName);
  else snprintf(name, sizeof name, "k_cols%s_fwd<%d,%s>", c.r3 ? "3" : "", c.l1, Loader::kName);
  ProfScope ps(e, name, on);
  if (c.reg) {
    const int xcd = e->xcd_rows && G > 1;
    const bool pair = c.M1() > 24;                             // two lanes per column
    const unsigned grid = row_work_grid(G, (1 << c.l2) / (pair ? kRegCols / 2 : kRegCols), xcd);
    if (pair) {
      if (c.M1() == 48) k_colsreg2_fwd<48, 13, Loader><<<dim3(grid), dim3(kRegCols), 0, on>>>(ld, W, G, c.twA, c.twB, xcd);
      else if (c.M1() == 32) k_colsreg2_fwd<32, 13, Loader><<<dim3(grid), dim3(kRegCols), 0, on>>>(ld, W, G, c.twA, c.twB, xcd);
      else return e->fail(PAL_ERR_INTERNAL, "two-lane column pass of %d points", c.M1());
    } else
    PAL_SWITCH_M1(c.M1(), c.l2, k_colsreg_fwd<MM, LR, Loader><<<dim3(grid), dim3(kRegCols), 0, on>>>(ld, W, G, c.twA, c.twB, xcd));
  } else if (c.r3) {
    const int tiles = int(c.M() / kPoints3), xcd = e->xcd_rows && G > 1;
    PAL_SWITCH_L3(c.l1, k_cols3_fwd<LL, Loader><<<dim3(row_work_grid(G, tiles, xcd)), dim3(kLanes3), 0, on>>>(
                            ld, W, c.l2, G, e->stage_table(LL), c.twA, c.twB, tiles, xcd));
  } else {
    const int tiles = int(c.M() / kPoints), xcd = e->xcd_rows && G > 1;
    PAL_SWITCH_L(c.l1, k_cols_fwd<LL, Loader><<<dim3(row_work_grid(G, tiles, xcd)), dim3(kLanes), 0, on>>>(
                           ld, W, c.l2, G, e->stage_table(LL), c.twA, c.twB, tiles, xcd));
  }
  return e->check(hipGetLastError(), "k_cols_fwd");
}

static int launch_rows(Engine* e, const Conv& c, int G, cd* W, bool conv, double scale, hipStream_t on = nullptr) {
  if (!on) on = e->stream;
  char name[64];
  snprintf(name, sizeof name, c.reg ? "k_rowsreg<%d,%s>" : "k_rows<%d,%s>", c.l2, conv ? "conv" : "fwd");
  ProfScope ps(e, name, on);
  const int tiles = int(c.M() / kPoints), xcd = e->xcd_rows && G > 1;
  const unsigned grid = row_work_grid(G, tiles, xcd);
  if (c.reg) {
    const cd* tws = e->stage_table(c.l2);
    if (!tws) return e->fail(PAL_ERR_NOMEM, "twiddle tables");
    const unsigned rgrid = row_work_grid(G, c.M1(), xcd);
    if (c.l2 == 13) {
      constexpr int lanes = BigTile<13, 32>::kLanes;
      if (conv) k_rowsreg<13, true><<<dim3(rgrid), dim3(lanes), 0, on>>>(W, c.chat, c.M1(), G, tws, scale, xcd);
      else k_rowsreg<13, false><<<dim3(rgrid), dim3(lanes), 0, on>>>(W, c.chat, c.M1(), G, tws, scale, xcd);
    } else {
      constexpr int lanes = BigTile<12, 32>::kLanes;
      if (conv) k_rowsreg<12, true><<<dim3(rgrid), dim3(lanes), 0, on>>>(W, c.chat, c.M1(), G, tws, scale, xcd);
      else k_rowsreg<12, false><<<dim3(rgrid), dim3(lanes), 0, on>>>(W, c.chat, c.M1(), G, tws, scale, xcd);
    }
  } else if (conv) {
    PAL_SWITCH_L(c.l2, k_rows<LL, true><<<dim3(grid), dim3(kLanes), 0, on>>>(W, c.chat, c.M(), G, e->stage_table(LL), scale, tiles, xcd));
  } else {
    PAL_SWITCH_L(c.l2, k_rows<LL, false><<<dim3(grid), dim3(kLanes), 0, on>>>(W, c.chat, c.M(), G, e->stage_table(LL), scale, tiles, xcd));
  }
  return e->check(hipGetLastError(), "k_rows");
}

template <class Storer>
static int launch_cols_inv(Engine* e, const Conv& c, int G, const cd* W, Storer st, hipStream_t on = nullptr) {
  if (!on) on = e->stream;
  char name[64];
  if (c.reg) snprintf(name, sizeof name, "k_colsreg_inv<%d,%s>", c.M1(), Storer::kName);
  else snprintf(name, sizeof name, "k_cols%s_inv<%d,%s>", c.r3 ? "3" : "", c.l1, Storer::kName);
  ProfScope ps(e, name, on);
  if (c.reg) {
    const int xcd = e->xcd_rows && G > 1;
    const bool pair = c.M1() > 24;                             // two lanes per column
    const unsigned grid = row_work_grid(G, (1 << c.l2) / (pair ? kRegCols / 2 : kRegCols), xcd);
    if (pair) {
      if (c.M1() == 48) k_colsreg2_inv<48, 13, Storer><<<dim3(grid), dim3(kRegCols), 0, on>>>(W, st, G, c.twA, c.twB, xcd);
      else if (c.M1() == 32) k_colsreg2_inv<32, 13, Storer><<<dim3(grid), dim3(kRegCols), 0, on>>>(W, st, G, c.twA, c.twB, xcd);
      else return e->fail(PAL_ERR_INTERNAL, "two-lane column pass of %d points", c.M1());
    } else
    PAL_SWITCH_M1(c.M1(), c.l2, k_colsreg_inv<MM, LR, Storer><<<dim3(grid), dim3(kRegCols), 0, on>>>(W, st, G, c.twA, c.twB, xcd));
  } else if (c.r3) {
    const unsigned grid = unsigned(size_t(G) * (c.M() / kPoints3));
    PAL_SWITCH_L3(c.l1, k_cols3_inv<LL, Storer><<<dim3(grid), dim3(kLanes3), 0, on>>>(W, st, c.l2, G, e->stage_table(LL),
                                                                                              c.twA, c.twB));
  } else {
    const unsigned grid = unsigned(size_t(G) * (c.M() / kPoints));
    PAL_SWITCH_L(c.l1, k_cols_inv<LL, Storer><<<dim3(grid), dim3(kLanes), 0, on>>>(W, st, c.l2, G, e->stage_table(LL),
                                                                                          c.twA, c.twB));
  }
  return e->check(hipGetLastError(), "k_cols_inv");
}

}  // namespace pal
