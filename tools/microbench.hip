// Ablation timings of the row pass (k_rows<10, conv>) on the metric geometry: G transforms of M = 196608 points.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -I pyaudiolocalization_amd/csrc -I include tools/microbench.hip -o /tmp/microbench
// Variants: full | memory only (no FFT stages) | LDS/VALU only (no global traffic).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

#include "fft_core.h"

using namespace pal;

template <int L2, int MODE>   // MODE 0 full, 1 memory only, 2 compute only
__global__ __launch_bounds__(256) void rows(cd* __restrict__ W, const cd* __restrict__ chat, size_t m, int G,
                                            const cd* __restrict__ tws) {
  constexpr int N2 = 1 << L2;
  __shared__ cd data[kPoints];
  __shared__ cd tw[N2];
  const int tid = threadIdx.x;
  const int g = blockIdx.x % G;
  const size_t tile = blockIdx.x / G;
  for (int i = tid; i < stage_tw_size(L2); i += kLanes) tw[i] = tws[i];
  cd* base = W + size_t(g) * m + tile * kPoints;
#pragma unroll
  for (int q = 0; q < kPoints / kLanes; ++q) {
    const int idx = tid + kLanes * q;
    data[lds_addr<L2, false>(idx >> L2, idx & (N2 - 1))] = MODE == 2 ? mk(double(idx), 1.0) : base[idx];
  }
  __syncthreads();
  if (MODE != 1) wg_fft<L2, false, false>(data, tw, tid);
  const cd* ch = chat + tile * kPoints;
#pragma unroll
  for (int q = 0; q < kPoints / kLanes; ++q) {
    const int idx = tid + kLanes * q;
    const int a = lds_addr<L2, false>(idx >> L2, idx & (N2 - 1));
    data[a] = cmul(data[a], MODE == 2 ? mk(0.5, 0.25) : ch[idx]);
  }
  __syncthreads();
  if (MODE != 1) wg_fft<L2, false, true>(data, tw, tid);
  if (MODE == 2) {
    if (data[tid].x == 123.456) base[0] = data[tid];      // keep the work alive, never true in practice
    return;
  }
#pragma unroll
  for (int q = 0; q < kPoints / kLanes; ++q) {
    const int idx = tid + kLanes * q;
    base[idx] = data[lds_addr<L2, false>(idx >> L2, idx & (N2 - 1))];
  }
}

template <int MODE> static float run(cd* W, cd* chat, cd* tws, size_t m, int G, int reps) {
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  const unsigned grid = unsigned(size_t(G) * (m / kPoints));
  rows<10, MODE><<<grid, 256>>>(W, chat, m, G, tws);
  hipDeviceSynchronize();
  hipEventRecord(a);
  for (int r = 0; r < reps; ++r) rows<10, MODE><<<grid, 256>>>(W, chat, m, G, tws);
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
    hipFree(W); hipFree(chat); hipFree(tws);
  }
  return 0;
}
