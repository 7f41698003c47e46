// reduce.h - workgroup reductions: 64-lane wavefront shuffles, then one LDS hop across the
// 4 wavefronts of a 256-lane workgroup.  `rd` / `ri` point at >= 4 LDS slots.
#pragma once
#include <hip/hip_runtime.h>

namespace pal {

__device__ inline double block_sum(double v, double* rd, int tid) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  __syncthreads();
  if ((tid & 63) == 0) rd[tid >> 6] = v;
  __syncthreads();
  return (rd[0] + rd[1]) + (rd[2] + rd[3]);
}

__device__ inline double block_max(double v, double* rd, int tid) {
  for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_down(v, o, 64));
  __syncthreads();
  if ((tid & 63) == 0) rd[tid >> 6] = v;
  __syncthreads();
  return fmax(fmax(rd[0], rd[1]), fmax(rd[2], rd[3]));
}

// arg-best over (value, index); index < 0 marks "no entry".
//   MODE 0: max value, lowest index on ties (np.argmax)   MODE 1: min value, lowest index on ties
//   MODE 2: max value, highest index on ties (peak priority)
template <int MODE> __device__ __forceinline__ bool arg_better(double a, int ia, double b, int ib) {
  if (MODE == 0) return a > b || (a == b && ia < ib);
  if (MODE == 1) return a < b || (a == b && ia < ib);
  return a > b || (a == b && ia > ib);
}

template <int MODE> __device__ inline void block_arg(double& v, int& i, double* rd, int* ri, int tid) {
  for (int o = 32; o > 0; o >>= 1) {
    const double ov = __shfl_down(v, o, 64);
    const int oi = __shfl_down(i, o, 64);
    if (oi >= 0 && (i < 0 || arg_better<MODE>(ov, oi, v, i))) { v = ov; i = oi; }
  }
  __syncthreads();
  if ((tid & 63) == 0) { rd[tid >> 6] = v; ri[tid >> 6] = i; }
  __syncthreads();
  v = rd[0];
  i = ri[0];
  for (int k = 1; k < 4; ++k)
    if (ri[k] >= 0 && (i < 0 || arg_better<MODE>(rd[k], ri[k], v, i))) { v = rd[k]; i = ri[k]; }
}

}  // namespace pal
