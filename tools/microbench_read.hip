// microbench_read.hip - what a plain streaming read of freshly written correlation rows achieves (GPU box):
// the yardstick for k_peak_stream (peaks.hip), which reads 0.7 MB per row once.
//   hipcc -O3 --offload-arch=gfx950 tools/microbench_read.hip -o tools/microbench_read && ./tools/microbench_read
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ __launch_bounds__(256) void k_fill(double2* p, size_t n) {
  for (size_t i = size_t(blockIdx.x) * 256 + threadIdx.x; i < n; i += size_t(gridDim.x) * 256) p[i] = make_double2(double(i), 1.0);
}

// every workgroup streams one contiguous segment with U 16-byte loads in flight per lane (plus a register double buffer if DB)
template <int U, bool DB> __global__ __launch_bounds__(256) void k_read(const double2* __restrict__ p, size_t per_wg, double* out) {
  const double2* q = p + size_t(blockIdx.x) * per_wg;
  double acc = 0;
  double2 cur[U], nxt[U];
  if (DB) {
#pragma unroll
    for (int u = 0; u < U; ++u) cur[u] = q[u * 256 + threadIdx.x];
  }
  for (size_t base = 0; base < per_wg; base += U * 256) {
    if (DB) {
      const size_t ahead = base + U * 256 < per_wg ? base + U * 256 : base;
#pragma unroll
      for (int u = 0; u < U; ++u) nxt[u] = q[ahead + u * 256 + threadIdx.x];
    } else {
#pragma unroll
      for (int u = 0; u < U; ++u) cur[u] = q[base + u * 256 + threadIdx.x];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) acc += cur[u].x + cur[u].y;
    if (DB) {
#pragma unroll
      for (int u = 0; u < U; ++u) cur[u] = nxt[u];
    }
  }
  if (acc == 12345.678) out[0] = acc;
}

// the stream launch's geometry: workgroup (row, segment) reads `tiles` tiles of 1024 double2 of its row, double-buffered
__global__ __launch_bounds__(256) void k_read_rows(const double2* __restrict__ p, size_t row_stride, int S, int tiles, int row_tiles, double* out) {
  const int row = blockIdx.x / S, seg = blockIdx.x % S;
  const int t0 = seg * tiles, t1 = t0 + tiles < row_tiles ? t0 + tiles : row_tiles;
  const double2* q = p + size_t(row) * row_stride;
  double acc = 0;
  double2 cur[4], nxt[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) cur[u] = q[size_t(t0) * 1024 + u * 256 + threadIdx.x];
  for (int t = t0; t < t1; ++t) {
    const int ahead = t + 1 < t1 ? t + 1 : t;
#pragma unroll
    for (int u = 0; u < 4; ++u) nxt[u] = q[size_t(ahead) * 1024 + u * 256 + threadIdx.x];
#pragma unroll
    for (int u = 0; u < 4; ++u) acc += cur[u].x + cur[u].y;
#pragma unroll
    for (int u = 0; u < 4; ++u) cur[u] = nxt[u];
  }
  if (acc == 12345.678) out[0] = acc;
}
// the same work with the tiles of all rows interleaved: tile t of row r at (t * rows + r) * 1024
__global__ __launch_bounds__(256) void k_read_tilemajor(const double2* __restrict__ p, int rows, int S, int tiles, int row_tiles, double* out) {
  const int row = blockIdx.x / S, seg = blockIdx.x % S;
  const int t0 = seg * tiles, t1 = t0 + tiles < row_tiles ? t0 + tiles : row_tiles;
  double acc = 0;
  double2 cur[4], nxt[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) cur[u] = p[(size_t(t0) * rows + row) * 1024 + u * 256 + threadIdx.x];
  for (int t = t0; t < t1; ++t) {
    const int ahead = t + 1 < t1 ? t + 1 : t;
#pragma unroll
    for (int u = 0; u < 4; ++u) nxt[u] = p[(size_t(ahead) * rows + row) * 1024 + u * 256 + threadIdx.x];
#pragma unroll
    for (int u = 0; u < 4; ++u) acc += cur[u].x + cur[u].y;
#pragma unroll
    for (int u = 0; u < 4; ++u) cur[u] = nxt[u];
  }
  if (acc == 12345.678) out[0] = acc;
}

int main() {
  const size_t bytes = size_t(512) * 88200 * 8;             // 512 rows of the metric workload
  const size_t n2 = bytes / 16;
  double2* buf;
  double* out;
  CHECK(hipMalloc(&buf, bytes));
  CHECK(hipMalloc(&out, 8));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  auto run = [&](const char* name, auto launch) {
    float best = 1e9f;
    for (int rep = 0; rep < 5; ++rep) {
      k_fill<<<2048, 256>>>(buf, n2);                       // freshly written, like the column pass leaves it
      (void)hipEventRecord(e0);
      launch();
      (void)hipEventRecord(e1);
      (void)hipEventSynchronize(e1);
      float ms;
      (void)hipEventElapsedTime(&ms, e0, e1);
      best = ms < best ? ms : best;
    }
    printf("%-44s %7.1f us  %5.2f TB/s\n", name, best * 1e3, double(bytes) / (best * 1e-3) / 1e12);
    return 0;
  };
  for (int wgs : {1024, 2048, 4096, 8192}) {
    const size_t per = n2 / wgs / (8 * 256) * (8 * 256);
    char nm[96];
    snprintf(nm, sizeof nm, "%d workgroups, 4 loads, double buffer", wgs);
    run(nm, [&] { k_read<4, true><<<wgs, 256>>>(buf, per, out); });
    snprintf(nm, sizeof nm, "%d workgroups, 8 loads, double buffer", wgs);
    run(nm, [&] { k_read<8, true><<<wgs, 256>>>(buf, per, out); });
    snprintf(nm, sizeof nm, "%d workgroups, 8 loads", wgs);
    run(nm, [&] { k_read<8, false><<<wgs, 256>>>(buf, per, out); });
  }
  for (int rows : {256, 512}) {
    const size_t row_stride = 44100;                        // double2 per row (88200 doubles)
    const int row_tiles = 43;                               // whole 1024-double2 tiles per row (the tail is left out here)
    for (int tiles : {11, 4}) {
      const int S = (row_tiles + tiles - 1) / tiles;
      char nm[96];
      snprintf(nm, sizeof nm, "%d rows x %d segments of %d tiles, row-major", rows, S, tiles);
      run(nm, [&] { k_read_rows<<<rows * S, 256>>>(buf, row_stride, S, tiles, row_tiles, out); });
      snprintf(nm, sizeof nm, "%d rows x %d segments of %d tiles, tile-major", rows, S, tiles);
      run(nm, [&] { k_read_tilemajor<<<rows * S, 256>>>(buf, rows, S, tiles, row_tiles, out); });
    }
  }
  return 0;
}
