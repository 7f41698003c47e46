// Do v_mfma_f64_16x16x4_f64 (matrix pipe) and v_fma_f64 (vector ALU) of DIFFERENT wavefronts on one SIMD overlap on gfx950?
//   hipcc -O3 --offload-arch=gfx950 tools/mfma_valu_overlap.hip -o tools/bin/mfma_valu_overlap
// A 512-lane workgroup puts two wavefronts on every SIMD.  mode 0: both run the FMA loop; 1: both the MFMA loop;
// 2: one of each.  If the pipes overlap, mode 2 takes about as long as ONE wavefront's loop alone (modes 3 / 4).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double double4_t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ double loop_mfma(int iters, double a, double b) {
  double4_t acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = double4_t{0, 0, 0, 0};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  }
  double s = 0;
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  return s;
}
__device__ __forceinline__ double loop_fma(int iters, double a) {
  double acc[32];
  for (int i = 0; i < 32; ++i) acc[i] = i;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 32; ++i) acc[i] = __builtin_fma(acc[i], a, 1.0);
  }
  double s = 0;
  for (int i = 0; i < 32; ++i) s += acc[i];
  return s;
}
// mode: 0 fma|fma  1 mfma|mfma  2 fma|mfma  3 fma|idle  4 mfma|idle
__global__ __launch_bounds__(512) void k_mix(double* out, int mode, int it_fma, int it_mfma) {
  const int second = __builtin_amdgcn_readfirstlane(int(threadIdx.x) >> 8);   // waves 4..7 = second wave of each SIMD
  const double a = 1.0 + threadIdx.x * 1e-9, b = threadIdx.x * 1e-3;
  double s = 0;
  const int what = second == 0 ? (mode == 1 || mode == 4 ? 1 : 0) : (mode == 0 ? 0 : (mode == 1 || mode == 2 ? 1 : 2));
  if (what == 0) s = loop_fma(it_fma, a);
  else if (what == 1) s = loop_mfma(it_mfma, a, b);
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main() {
  double* d;
  hipMalloc(&d, sizeof(double) * 512 * 1024);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const int blocks = 256, it_fma = 4000, it_mfma = 1000;     // 128 k wave-FMAs (4 cycles each) against 8 k MFMAs (64 cycles each)
  const char* names[5] = {"fma | fma", "mfma | mfma", "fma | mfma", "fma | idle", "mfma | idle"};
  for (int mode = 0; mode < 5; ++mode) {
    float ms;
    k_mix<<<blocks, 512>>>(d, mode, it_fma, it_mfma); hipDeviceSynchronize();
    hipEventRecord(e0); k_mix<<<blocks, 512>>>(d, mode, it_fma, it_mfma); hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
    const double nf = (mode == 0 ? 2 : (mode == 2 || mode == 3 ? 1 : 0)) * double(blocks) * 4 * it_fma * 32 * 128;
    const double nm = (mode == 1 ? 2 : (mode == 2 || mode == 4 ? 1 : 0)) * double(blocks) * 4 * it_mfma * 8 * 2048;
    printf("%-12s %8.3f ms   vector %6.1f TFLOP/s   matrix %6.1f TFLOP/s   sum %6.1f\n", names[mode], ms, nf / ms / 1e9, nm / ms / 1e9,
           (nf + nm) / ms / 1e9);
  }
  return 0;
}
