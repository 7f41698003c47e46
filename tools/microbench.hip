// Ablation timings of the row pass (k_rows<10, conv>) on the metric geometry: G transforms of M = 196608 points.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -I pyaudiolocalization_amd/csrc -I include tools/microbench.hip -o tools/microbench
// Variants: full | memory only (no FFT stages) | LDS/VALU only (no global traffic).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

#include "fft_core.h"

using namespace pal;

template <int L2> struct GlobalIO {
  static constexpr bool kLds = false;
  cd* base;
  __device__ cd operator()(int t, int e) const { return base[(t << L2) + e]; }
  __device__ void operator()(int t, int e, cd v) const { base[(t << L2) + e] = v; }
};
template <int L2> struct ChatToLds {
  static constexpr bool kLds = true;
  cd* data;
  const cd* ch;
  __device__ void operator()(int t, int e, cd v) const { data[lds_addr<L2, false>(t, e)] = cmul(v, ch[(t << L2) + e]); }
};
template <int L2> struct ConstIn {
  static constexpr bool kLds = false;
  __device__ cd operator()(int t, int e) const { return mk(double(e), double(t)); }
};
template <int L2> struct ConstChatToLds {
  static constexpr bool kLds = true;
  cd* data;
  __device__ void operator()(int t, int e, cd v) const { data[lds_addr<L2, false>(t, e)] = cmul(v, mk(0.5, 0.25)); }
};
template <int L2> struct Sink {
  static constexpr bool kLds = false;
  cd* base;
  __device__ void operator()(int t, int e, cd v) const { if (v.x == 123.456) base[0] = v; }
};

template <int L2, int MODE, int T = (kPoints >> L2), bool TWLDS = true>   // MODE 0 full, 1 memory only, 2 compute only
__global__ __launch_bounds__(T * (1 << L2) / 16) void rows(cd* __restrict__ W, const cd* __restrict__ chat, size_t m, int G,
                                            const cd* __restrict__ tws) {
  constexpr int N2 = 1 << L2, POINTS = T * N2, LANES = POINTS / 16;
  __shared__ cd data[POINTS];
  __shared__ cd twl[TWLDS ? N2 : 1];
  const int tid = threadIdx.x;
  const int g = blockIdx.x % G;
  const size_t tile = blockIdx.x / G;
  if (TWLDS) for (int i = tid; i < stage_tw_size(L2); i += LANES) twl[i] = tws[i];
  const cd* tw = TWLDS ? twl : tws;
  cd* base = W + size_t(g) * m + tile * POINTS;
  const cd* ch = chat + tile * POINTS;
  if (MODE == 0) {
    wg_fft<L2, false, false, T>(data, tw, tid, GlobalIO<L2>{base}, ChatToLds<L2>{data, ch});
    wg_fft<L2, false, true, T>(data, tw, tid, LdsTile<L2, false, T>{data}, GlobalIO<L2>{base});
  } else if (MODE == 1) {
#pragma unroll
    for (int q = 0; q < 16; ++q) { const int idx = tid + LANES * q; base[idx] = cmul(base[idx], ch[idx]); }
  } else {
    wg_fft<L2, false, false, T>(data, tw, tid, ConstIn<L2>{}, ConstChatToLds<L2>{data});
    wg_fft<L2, false, true, T>(data, tw, tid, LdsTile<L2, false, T>{data}, Sink<L2>{base});
  }
}

template <int MODE, int T = 4, bool TWLDS = true> static float run(cd* W, cd* chat, cd* tws, size_t m, int G, int reps) {
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  const unsigned grid = unsigned(size_t(G) * (m / (T * 1024)));
  rows<10, MODE, T, TWLDS><<<grid, T * 64>>>(W, chat, m, G, tws);
  hipDeviceSynchronize();
  hipEventRecord(a);
  for (int r = 0; r < reps; ++r) rows<10, MODE, T, TWLDS><<<grid, T * 64>>>(W, chat, m, G, tws);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms = 0;
  hipEventElapsedTime(&ms, a, b);
  return ms / reps * 1000.f;
}

int main() {
  const size_t m = 196608;
  for (int G : {16, 32, 64, 128}) {
    cd *W, *chat, *tws;
    hipMalloc(&W, G * m * sizeof(cd));
    hipMalloc(&chat, m * sizeof(cd));
    hipMalloc(&tws, 1024 * sizeof(cd));
    hipMemset(W, 0, G * m * sizeof(cd));
    hipMemset(chat, 0, m * sizeof(cd));
    hipMemset(tws, 0, 1024 * sizeof(cd));
    const float full = run<0>(W, chat, tws, m, G, 20), mem = run<1>(W, chat, tws, m, G, 20), cmp = run<2>(W, chat, tws, m, G, 20);
    const double bytes = 2.0 * G * m * 16;
    std::printf("G=%3d  full %7.1f us (%5.2f TB/s r+w)   memory-only %7.1f us (%5.2f TB/s)   compute-only %7.1f us\n", G, full,
                bytes / full * 1e-6, mem, bytes / mem * 1e-6, cmp);
    std::printf("       2 rows/WG (128 lanes): tw in LDS %7.1f us, tw from L1 %7.1f us | 1 row/WG (64 lanes): %7.1f / %7.1f us | 4 rows, tw from L1 %7.1f us\n",
                run<0, 2, true>(W, chat, tws, m, G, 20), run<0, 2, false>(W, chat, tws, m, G, 20), run<0, 1, true>(W, chat, tws, m, G, 20),
                run<0, 1, false>(W, chat, tws, m, G, 20), run<0, 4, false>(W, chat, tws, m, G, 20));
    hipFree(W); hipFree(chat); hipFree(tws);
  }
  return 0;
}
