// microbench_tile.hip - what the column passes' access pattern costs on its own (GPU box): a 192-lane workgroup moves a
// tile of 16 columns x 192 rows of complex doubles of a [192][1024] matrix per transform, 16 loads / stores of 16 B per
// lane, (a) in the natural row-major layout (256-B segments at a 16 KB stride), (b) in a tiled layout where the tile
// is one contiguous 48 KB block, (c) 64-column tiles [16 tiles][192][64] (256-B segments at a 1 KB stride).
//   hipcc -O3 --offload-arch=gfx950 tools/microbench_tile.hip -o tools/microbench_tile && ./tools/microbench_tile
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
constexpr int M1 = 192, M2 = 1024, T = 16;

template <int LAYOUT> __device__ __forceinline__ size_t addr(int k1, int c) {
  if (LAYOUT == 0) return size_t(k1) * M2 + c;
  if (LAYOUT == 1) return (size_t(c >> 4) * M1 + k1) * 16 + (c & 15);
  return (size_t(c >> 6) * M1 + k1) * 64 + (c & 63);
}

template <int LAYOUT, int MODE>   // MODE 0: read, 1: write, 2: read + write in place
__global__ __launch_bounds__(192) void k_tile(double2* __restrict__ W, int G, double* out) {
  const int g = blockIdx.x % G, c0 = (blockIdx.x / G) * T;
  double2* Wg = W + size_t(g) * M1 * M2;
  const int col = c0 + (threadIdx.x & 15), rg = threadIdx.x >> 4;   // 12 row groups
  double2 v[16];
  if (MODE != 1) {
#pragma unroll
    for (int r = 0; r < 16; ++r) v[r] = Wg[addr<LAYOUT>(rg + 12 * r, col)];
  } else {
#pragma unroll
    for (int r = 0; r < 16; ++r) v[r] = make_double2(double(threadIdx.x + r), 1.0);
  }
  if (MODE == 0) {
    double acc = 0;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc += v[r].x + v[r].y;
    if (acc == 12345.678) out[0] = acc;
  } else {
#pragma unroll
    for (int r = 0; r < 16; ++r) Wg[addr<LAYOUT>(rg + 12 * r, col)] = make_double2(v[r].y, v[r].x);
  }
}

// the row pass: 256 lanes, 4 rows of 1024 points, read + write in place; lane l of wave w: row 4 b + w, columns l + 64 r
template <int LAYOUT> __global__ __launch_bounds__(256) void k_rows4(double2* __restrict__ W, int G) {
  const int g = blockIdx.x % G, b = blockIdx.x / G;
  double2* Wg = W + size_t(g) * M1 * M2;
  const int row = 4 * b + (threadIdx.x >> 6), l = threadIdx.x & 63;
  double2 v[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) v[r] = Wg[addr<LAYOUT>(row, l + 64 * r)];
#pragma unroll
  for (int r = 0; r < 16; ++r) Wg[addr<LAYOUT>(row, l + 64 * r)] = make_double2(v[r].y, v[r].x);
}

int main() {
  const int G = 240;
  const size_t n = size_t(G) * M1 * M2;
  double2* W; double* out;
  CHECK(hipMalloc(&W, n * sizeof(double2)));
  CHECK(hipMalloc(&out, 8));
  CHECK(hipMemset(W, 0, n * sizeof(double2)));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  const double gb = double(n) * 16 / 1e9;
  auto time = [&](auto launch, const char* name, double bytes_factor) {
    launch(); hipDeviceSynchronize();
    float best = 1e9;
    for (int rep = 0; rep < 5; ++rep) {
      hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      if (ms < best) best = ms;
    }
    printf("%-44s %8.1f us  %6.2f TB/s\n", name, best * 1e3, gb * bytes_factor / best);
    return 0;
  };
  const unsigned grid = G * (M2 / T);
#define RUN(L, MODE, name, f) time([&] { k_tile<L, MODE><<<grid, 192>>>(W, G, out); }, name, f)
  RUN(0, 0, "cols read   natural (256 B @ 16 KB)", 1);
  RUN(1, 0, "cols read   tiled 16 (48 KB contiguous)", 1);
  RUN(2, 0, "cols read   tiled 64 (256 B @ 1 KB)", 1);
  RUN(0, 1, "cols write  natural", 1);
  RUN(1, 1, "cols write  tiled 16", 1);
  RUN(2, 1, "cols write  tiled 64", 1);
  RUN(0, 2, "cols r+w    natural", 2);
  RUN(1, 2, "cols r+w    tiled 16", 2);
  RUN(2, 2, "cols r+w    tiled 64", 2);
  const unsigned grid4 = G * (M1 / 4);
  time([&] { k_rows4<0><<<grid4, 256>>>(W, G); }, "rows r+w    natural (16 KB contiguous / row)", 2);
  time([&] { k_rows4<1><<<grid4, 256>>>(W, G); }, "rows r+w    tiled 16 (256 B @ 48 KB)", 2);
  time([&] { k_rows4<2><<<grid4, 256>>>(W, G); }, "rows r+w    tiled 64 (1 KB @ 192 KB)", 2);
  return 0;
}
