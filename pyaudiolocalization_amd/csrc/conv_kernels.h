// conv_kernels.h - the three workgroup passes of one circular convolution of length M = M1 x M2
// (see bluestein.hip for the algorithm) as templates over loader / storer functors, shared by the
// PHAT pipeline (bluestein.hip) and the multipath / synchronisation pipelines (sim.hip).
//
// M2 (rows) is always a power of two.  M1 (columns) is 2^l1, or 3 * 2^l1 when the convolution length
// is 3 * 2^k: all BASELINE frame lengths need 2n-1 = 0.73 * 2^k points, so the 3 * 2^(k-1) length cuts
// a quarter of the HBM traffic of every pass.  The radix-3 column kernels run 192 lanes (3 wavefronts)
// over 3072 points: one in-place 3-point butterfly stage in LDS (fft_core.h radix3_item) around the
// same power-of-two sub-transforms.
#pragma once
#include "engine.h"

namespace pal {

__global__ void k_make_chirp(cd* w, int n);
__global__ void k_make_roots(cd* out, int count, double denom);
__global__ void k_make_stage_tw(cd* out, int ln);

// ------------------------------------------------------------------ the three passes (M1 = 2^L1)
// Each workgroup (256 lanes) owns 4096 points.  Grid = G * M / 4096, transform index fastest so that
// the workgroups that share chirp-spectrum rows and twiddles run together.
template <int L1, class Loader>
__global__ __launch_bounds__(256) void k_cols_fwd(Loader ld, cd* __restrict__ W, int l2, int G,
                                                  const cd* __restrict__ tws, const cd* __restrict__ twA,
                                                  const cd* __restrict__ twB) {
  constexpr int N1 = 1 << L1, T = kPoints / N1;
  __shared__ cd data[kPoints];
  __shared__ cd tw[N1];
  const int tid = threadIdx.x;
  const int g = blockIdx.x % G;
  const unsigned c0 = (blockIdx.x / G) * T;
  for (int i = tid; i < stage_tw_size(L1); i += kLanes) tw[i] = tws[i];
#pragma unroll
  for (int q = 0; q < kPoints / kLanes; ++q) {
    const unsigned idx = tid + kLanes * q, c = idx % T, j1 = idx / T;
    data[idx] = ld(g, (j1 << l2) + c0 + c);
  }
  __syncthreads();
  wg_fft<L1, true, false>(data, tw, tid);
  cd* out = W + (size_t(g) << (L1 + l2));
  const unsigned mask = (1u << l2) - 1;
#pragma unroll
  for (int q = 0; q < kPoints / kLanes; ++q) {
    const unsigned idx = tid + kLanes * q, c = idx % T, k1 = idx / T;
    const unsigned e = (c0 + c) * k1;                       // < M
    const cd f = cmul(twA[e >> l2], twB[e & mask]);        // exp(-2 pi i e / M)
    out[(size_t(k1) << l2) + c0 + c] = cmul(data[idx], f);
  }
}

// rows of length 2^L2; `m` = points per transform (any multiple of 4096)
template <int L2, bool CONV>
__global__ __launch_bounds__(256) void k_rows(cd* __restrict__ W, const cd* __restrict__ chat, size_t m, int G,
                                              const cd* __restrict__ tws, double scale) {
  constexpr int N2 = 1 << L2;
  __shared__ cd data[kPoints];
  __shared__ cd tw[N2];
  const int tid = threadIdx.x;
  const int g = blockIdx.x % G;
  const size_t tile = blockIdx.x / G;                       // 4096 consecutive points = 4096/N2 rows
  for (int i = tid; i < stage_tw_size(L2); i += kLanes) tw[i] = tws[i];
  cd* base = W + size_t(g) * m + tile * kPoints;
#pragma unroll
  for (int q = 0; q < kPoints / kLanes; ++q) {
    const int idx = tid + kLanes * q;
    data[lds_addr<L2, false>(idx >> L2, idx & (N2 - 1))] = base[idx];
  }
  __syncthreads();
  wg_fft<L2, false, false>(data, tw, tid);
  if (CONV) {
    const cd* ch = chat + tile * kPoints;
#pragma unroll
    for (int q = 0; q < kPoints / kLanes; ++q) {
      const int idx = tid + kLanes * q;
      const int a = lds_addr<L2, false>(idx >> L2, idx & (N2 - 1));
      data[a] = cmul(data[a], ch[idx]);
    }
    __syncthreads();
    wg_fft<L2, false, true>(data, tw, tid);
#pragma unroll
    for (int q = 0; q < kPoints / kLanes; ++q) {
      const int idx = tid + kLanes * q;
      base[idx] = data[lds_addr<L2, false>(idx >> L2, idx & (N2 - 1))];
    }
  } else {
#pragma unroll
    for (int q = 0; q < kPoints / kLanes; ++q) {
      const int idx = tid + kLanes * q;
      base[idx] = cscale(data[lds_addr<L2, false>(idx >> L2, idx & (N2 - 1))], scale);
    }
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
  for (int i = tid; i < stage_tw_size(L1); i += kLanes) tw[i] = tws[i];
  const cd* in = W + (size_t(g) << (L1 + l2));
  const unsigned mask = (1u << l2) - 1;
#pragma unroll
  for (int q = 0; q < kPoints / kLanes; ++q) {
    const unsigned idx = tid + kLanes * q, c = idx % T, k1 = idx / T;
    const unsigned e = (c0 + c) * k1;
    const cd f = cmul(twA[e >> l2], twB[e & mask]);
    data[idx] = cmulc(in[(size_t(k1) << l2) + c0 + c], f);   // times exp(+2 pi i e / M)
  }
  __syncthreads();
  wg_fft<L1, true, true>(data, tw, tid);
#pragma unroll
  for (int q = 0; q < kPoints / kLanes; ++q) {
    const unsigned idx = tid + kLanes * q, c = idx % T, j1 = idx / T;
    st(g, (j1 << l2) + c0 + c, data[idx]);
  }
}

// ------------------------------------------------------------------ column passes for M1 = 3 * 2^LN
// 192 lanes, T = 1024 / 2^LN columns per tile, 3072 points: sub-transform (q, c) of the LDS tile holds row
// q*N + e of column c on the time side and row 3e + q on the frequency side (fft_core.h radix3_item).
constexpr int kPoints3 = 3072, kLanes3 = 192;

template <int LN, class Loader>
__global__ __launch_bounds__(192) void k_cols3_fwd(Loader ld, cd* __restrict__ W, int l2, int G,
                                                   const cd* __restrict__ tws, const cd* __restrict__ twA,
                                                   const cd* __restrict__ twB) {
  constexpr int N = 1 << LN, T = 1024 / N, NSUB = 3 * T;
  __shared__ cd data[kPoints3];
  __shared__ cd tw[N < 16 ? 16 : N];
  const int tid = threadIdx.x;
  const int g = blockIdx.x % G;
  const unsigned c0 = (blockIdx.x / G) * T;
  for (int i = tid; i < stage_tw_size(LN); i += kLanes3) tw[i] = tws[i];
#pragma unroll
  for (int q = 0; q < kPoints3 / kLanes3; ++q) {
    const unsigned idx = tid + kLanes3 * q, c = idx % T, j1 = idx / T;      // j1 < 3N
    data[lds_addr<LN, true, NSUB>((j1 >> LN) * T + c, j1 & (N - 1))] = ld(g, (j1 << l2) + c0 + c);
  }
  __syncthreads();
  for (int w = tid; w < N * T; w += kLanes3) radix3_item<LN, T, false>(data, twA, w);
  __syncthreads();
  wg_fft<LN, true, false, NSUB>(data, tw, tid);
  cd* out = W + size_t(g) * (size_t(3 * N) << l2);
  const unsigned mask = (1u << l2) - 1;
#pragma unroll
  for (int q = 0; q < kPoints3 / kLanes3; ++q) {
    const unsigned idx = tid + kLanes3 * q, c = idx % T, k1 = idx / T;      // k1 = 3e + r
    const unsigned e = (c0 + c) * k1;
    const cd f = cmul(twA[e >> l2], twB[e & mask]);
    out[(size_t(k1) << l2) + c0 + c] = cmul(data[lds_addr<LN, true, NSUB>((k1 % 3) * T + c, k1 / 3)], f);
  }
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
  const cd* in = W + size_t(g) * (size_t(3 * N) << l2);
  const unsigned mask = (1u << l2) - 1;
#pragma unroll
  for (int q = 0; q < kPoints3 / kLanes3; ++q) {
    const unsigned idx = tid + kLanes3 * q, c = idx % T, k1 = idx / T;
    const unsigned e = (c0 + c) * k1;
    const cd f = cmul(twA[e >> l2], twB[e & mask]);
    data[lds_addr<LN, true, NSUB>((k1 % 3) * T + c, k1 / 3)] = cmulc(in[(size_t(k1) << l2) + c0 + c], f);
  }
  __syncthreads();
  wg_fft<LN, true, true, NSUB>(data, tw, tid);
  for (int w = tid; w < N * T; w += kLanes3) radix3_item<LN, T, true>(data, twA, w);
  __syncthreads();
#pragma unroll
  for (int q = 0; q < kPoints3 / kLanes3; ++q) {
    const unsigned idx = tid + kLanes3 * q, c = idx % T, j1 = idx / T;
    st(g, (j1 << l2) + c0 + c, data[lds_addr<LN, true, NSUB>((j1 >> LN) * T + c, j1 & (N - 1))]);
  }
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
static int launch_cols_fwd(Engine* e, const Conv& c, int G, Loader ld, cd* W) {
  char name[64];
  snprintf(name, sizeof name, "k_cols%s_fwd<%d,%s>", c.r3 ? "3" : "", c.l1, Loader::kName);
  ProfScope ps(e, name);
  if (c.r3) {
    const unsigned grid = unsigned(size_t(G) * (c.M() / kPoints3));
    PAL_SWITCH_L3(c.l1, k_cols3_fwd<LL, Loader><<<dim3(grid), dim3(kLanes3), 0, e->stream>>>(ld, W, c.l2, G, e->stage_table(LL),
                                                                                              c.twA, c.twB));
  } else {
    const unsigned grid = unsigned(size_t(G) * (c.M() / kPoints));
    PAL_SWITCH_L(c.l1, k_cols_fwd<LL, Loader><<<dim3(grid), dim3(kLanes), 0, e->stream>>>(ld, W, c.l2, G, e->stage_table(LL),
                                                                                          c.twA, c.twB));
  }
  return e->check(hipGetLastError(), "k_cols_fwd");
}

static int launch_rows(Engine* e, const Conv& c, int G, cd* W, bool conv, double scale) {
  char name[64];
  snprintf(name, sizeof name, "k_rows<%d,%s>", c.l2, conv ? "conv" : "fwd");
  ProfScope ps(e, name);
  const unsigned grid = unsigned(size_t(G) * (c.M() / kPoints));
  if (conv) {
    PAL_SWITCH_L(c.l2, k_rows<LL, true><<<dim3(grid), dim3(kLanes), 0, e->stream>>>(W, c.chat, c.M(), G, e->stage_table(LL), scale));
  } else {
    PAL_SWITCH_L(c.l2, k_rows<LL, false><<<dim3(grid), dim3(kLanes), 0, e->stream>>>(W, c.chat, c.M(), G, e->stage_table(LL), scale));
  }
  return e->check(hipGetLastError(), "k_rows");
}

template <class Storer>
static int launch_cols_inv(Engine* e, const Conv& c, int G, const cd* W, Storer st) {
  char name[64];
  snprintf(name, sizeof name, "k_cols%s_inv<%d,%s>", c.r3 ? "3" : "", c.l1, Storer::kName);
  ProfScope ps(e, name);
  if (c.r3) {
    const unsigned grid = unsigned(size_t(G) * (c.M() / kPoints3));
    PAL_SWITCH_L3(c.l1, k_cols3_inv<LL, Storer><<<dim3(grid), dim3(kLanes3), 0, e->stream>>>(W, st, c.l2, G, e->stage_table(LL),
                                                                                              c.twA, c.twB));
  } else {
    const unsigned grid = unsigned(size_t(G) * (c.M() / kPoints));
    PAL_SWITCH_L(c.l1, k_cols_inv<LL, Storer><<<dim3(grid), dim3(kLanes), 0, e->stream>>>(W, st, c.l2, G, e->stage_table(LL),
                                                                                          c.twA, c.twB));
  }
  return e->check(hipGetLastError(), "k_cols_inv");
}

}  // namespace pal
