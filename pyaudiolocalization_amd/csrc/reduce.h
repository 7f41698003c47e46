// reduce.h - workgroup reductions: 64-lane wavefront shuffles, then one LDS hop across the NW
// wavefronts of the workgroup.  `rd` / `ri` point at >= NW LDS slots; results are valid in every lane.
#pragma once
#include <hip/hip_runtime.h>

namespace pal {

template <int NW = 4> __device__ inline double block_sum(double v, double* rd, int tid) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  __syncthreads();
  if ((tid & 63) == 0) rd[tid >> 6] = v;
  __syncthreads();
  double r = 0;
#pragma unroll
  for (int k = 0; k < NW; ++k) r += rd[k];
  return r;
}

template <int NW = 4> __device__ inline long long block_sum_ll(long long v, long long* rl, int tid) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  __syncthreads();
  if ((tid & 63) == 0) rl[tid >> 6] = v;
  __syncthreads();
  long long r = 0;
#pragma unroll
  for (int k = 0; k < NW; ++k) r += rl[k];
  return r;
}

template <int NW = 4> __device__ inline double block_max(double v, double* rd, int tid) {
  for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_down(v, o, 64));
  __syncthreads();
  if ((tid & 63) == 0) rd[tid >> 6] = v;
  __syncthreads();
  double r = rd[0];
#pragma unroll
  for (int k = 1; k < NW; ++k) r = fmax(r, rd[k]);
  return r;
}

// arg-best over (value, index); index < 0 marks "no entry".
//   MODE 0: max value, lowest index on ties (np.argmax)   MODE 1: min value, lowest index on ties
//   MODE 2: max value, highest index on ties (peak priority)
template <int MODE> __device__ __forceinline__ bool arg_better(double a, int ia, double b, int ib) {
  if (MODE == 0) return a > b || (a == b && ia < ib);
  if (MODE == 1) return a < b || (a == b && ia < ib);
  return a > b || (a == b && ia > ib);
}

template <int MODE, int NW = 4> __device__ inline void block_arg(double& v, int& i, double* rd, int* ri, int tid) {
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
#pragma unroll
  for (int k = 1; k < NW; ++k)
    if (ri[k] >= 0 && (i < 0 || arg_better<MODE>(rd[k], ri[k], v, i))) { v = rd[k]; i = ri[k]; }
}

}  // namespace pal
