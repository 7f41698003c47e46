// GPU check of csrc/wave_reduce.h (DPP wave reductions): hipcc -O3 --offload-arch=gfx950 -I pyaudiolocalization_amd/csrc tools/wave_reduce_check.hip -o tools/bin/wave_reduce_check
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include "wave_reduce.h"
using namespace pal;
__global__ void k(double* out, int* iout, int seed) {
  const int lane = threadIdx.x;
  const double v = double((lane * 37 + seed) % 64) + 0.5 - 20.0;
  out[0 * 64 + lane] = wave_bcast63(wave_sum63(v));
  out[1 * 64 + lane] = wave_bcast63(wave_max63(v));
  out[2 * 64 + lane] = wave_bcast63(wave_min63(v));
  double a = ((lane + seed) % 7 == 3) ? double((lane * 11) % 13) : 0.0;
  int i = ((lane + seed) % 7 == 3) ? lane : -1;
  wave_arg63(a, i, [](double v1, int i1, double v2, int i2) { return v1 > v2 || (v1 == v2 && i1 < i2); });
  out[3 * 64 + lane] = wave_bcast63(a);
  iout[lane] = wave_bcast63(i);
  double b = ((lane + seed) % 5 == 1) ? double((lane * 3) % 4) : 0.0;
  int j = ((lane + seed) % 5 == 1) ? 1000 - lane : -1;
  wave_arg63(b, j, [](double v1, int i1, double v2, int i2) { return v1 > v2 || (v1 == v2 && i1 > i2); });
  out[4 * 64 + lane] = wave_bcast63(b);
  iout[64 + lane] = wave_bcast63(j);
}
int main() {
  double* d; int* di;
  hipMalloc(&d, 5 * 64 * 8); hipMalloc(&di, 128 * 4);
  int bad = 0;
  for (int seed = 0; seed < 9; ++seed) {
    k<<<1, 64>>>(d, di, seed);
    double h[5 * 64]; int hi[128];
    hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost); hipMemcpy(hi, di, sizeof hi, hipMemcpyDeviceToHost);
    double s = 0, mx = -1e300, mn = 1e300, av = 0, bv = 0; int ai = -1, bj = -1;
    for (int lane = 0; lane < 64; ++lane) {
      const double v = double((lane * 37 + seed) % 64) + 0.5 - 20.0;
      s += v; mx = std::fmax(mx, v); mn = std::fmin(mn, v);
      if ((lane + seed) % 7 == 3) { const double a = double((lane * 11) % 13); if (ai < 0 || a > av || (a == av && lane < ai)) { av = a; ai = lane; } }
      if ((lane + seed) % 5 == 1) { const double b = double((lane * 3) % 4); const int j = 1000 - lane; if (bj < 0 || b > bv || (b == bv && j > bj)) { bv = b; bj = j; } }
    }
    for (int lane = 0; lane < 64; ++lane)
      if (h[lane] != s || h[64 + lane] != mx || h[128 + lane] != mn || h[192 + lane] != av || hi[lane] != ai || h[256 + lane] != bv || hi[64 + lane] != bj) ++bad;
  }
  printf(bad ? "wave_reduce: %d MISMATCHES\n" : "wave_reduce: ALL OK\n", bad);
  return bad != 0;
}
