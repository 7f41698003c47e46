// Issue rate of v_mfma_f64_16x16x4_f64 against v_fma_f64 on every CU (diagnostic).
//   hipcc -O3 --offload-arch=gfx950 tools/mfma_f64_rate.hip -o tools/mfma_f64_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double double4_t __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k_mfma(double* out, int iters) {
  double4_t acc[12];
  for (int i = 0; i < 12; ++i) acc[i] = double4_t{0, 0, 0, 0};
  double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-6;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 12; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  }
  double s = 0;
  for (int i = 0; i < 12; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ __launch_bounds__(256) void k_fma(double* out, int iters) {
  double acc[48];
  for (int i = 0; i < 48; ++i) acc[i] = i;
  double a = threadIdx.x * 1e-3;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 48; ++i) acc[i] = __builtin_fma(acc[i], a, 1.0);
  }
  double s = 0;
  for (int i = 0; i < 48; ++i) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main() {
  double* d;
  hipMalloc(&d, sizeof(double) * 256 * 1024 * 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int waves_per_simd = 1; waves_per_simd <= 2; ++waves_per_simd) {
    const int blocks = 256 * waves_per_simd, iters = 2000;
    float ms;
    k_mfma<<<blocks, 256>>>(d, iters); hipDeviceSynchronize();
    hipEventRecord(e0); k_mfma<<<blocks, 256>>>(d, iters); hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
    const double mf = double(blocks) * 4 * iters * 12;           // wave-MFMAs
    printf("mfma f64 16x16x4: %d wave(s)/SIMD  %.3f ms  %.1f TFLOP/s  %.1f ns per MFMA per SIMD\n", waves_per_simd, ms,
           mf * 2048 / ms / 1e9, ms * 1e6 / (double(iters) * 12 * waves_per_simd));
    k_fma<<<blocks, 256>>>(d, iters); hipDeviceSynchronize();
    hipEventRecord(e0); k_fma<<<blocks, 256>>>(d, iters); hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
    const double fm = double(blocks) * 4 * iters * 48;
    printf("v_fma_f64:        %d wave(s)/SIMD  %.3f ms  %.1f TFLOP/s  %.2f ns per wave-FMA per SIMD\n", waves_per_simd, ms,
           fm * 128 / ms / 1e9, ms * 1e6 / (double(iters) * 48 * waves_per_simd));
  }
  return 0;
}
