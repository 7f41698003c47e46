// wave_reduce.h - reductions over the 64 lanes of a wavefront by DPP row shifts and row broadcasts (gfx9 / CDNA: row_shr,
// row_bcast:15, row_bcast:31).  Plain VALU moves - no LDS crossbar (ds_bpermute behind __shfl costs a hundred cycles per
// dependent step; the statistics of a column block run ten such reductions one after the other).  THE RESULT IS IN LANE 63;
// wave_bcast63 hands it to every lane through the scalar unit.
#pragma once
#include <hip/hip_runtime.h>

namespace pal {

template <int CTRL, int ROWS = 0xf> __device__ __forceinline__ int dpp_i(int old, int v) {   // lanes without a source (or outside ROWS) keep `old`
  return __builtin_amdgcn_update_dpp(old, v, CTRL, ROWS, 0xf, false);
}
template <int CTRL, int ROWS = 0xf> __device__ __forceinline__ double dpp_d(double old, double v) {
  return __hiloint2double(dpp_i<CTRL, ROWS>(__double2hiint(old), __double2hiint(v)), dpp_i<CTRL, ROWS>(__double2loint(old), __double2loint(v)));
}
constexpr int kShr1 = 0x111, kShr2 = 0x112, kShr4 = 0x114, kShr8 = 0x118, kBcast15 = 0x142, kBcast31 = 0x143;

// OP(a, b) associative and commutative with identity ID; all 64 lanes take part (masked lanes pass ID)
template <class OP> __device__ __forceinline__ double wave_reduce_d(double v, double id, OP op) {
  v = op(v, dpp_d<kShr1>(id, v));
  v = op(v, dpp_d<kShr2>(id, v));
  v = op(v, dpp_d<kShr4>(id, v));
  v = op(v, dpp_d<kShr8>(id, v));
  v = op(v, dpp_d<kBcast15, 0xa>(id, v));
  v = op(v, dpp_d<kBcast31, 0xc>(id, v));
  return v;
}
template <class OP> __device__ __forceinline__ int wave_reduce_i(int v, int id, OP op) {
  v = op(v, dpp_i<kShr1>(id, v));
  v = op(v, dpp_i<kShr2>(id, v));
  v = op(v, dpp_i<kShr4>(id, v));
  v = op(v, dpp_i<kShr8>(id, v));
  v = op(v, dpp_i<kBcast15, 0xa>(id, v));
  v = op(v, dpp_i<kBcast31, 0xc>(id, v));
  return v;
}
__device__ __forceinline__ double wave_sum63(double v) { return wave_reduce_d(v, 0.0, [](double a, double b) { return a + b; }); }
__device__ __forceinline__ double wave_max63(double v) { return wave_reduce_d(v, -__builtin_huge_val(), [](double a, double b) { return fmax(a, b); }); }
__device__ __forceinline__ double wave_min63(double v) { return wave_reduce_d(v, __builtin_huge_val(), [](double a, double b) { return fmin(a, b); }); }

// (value, index) pairs, index < 0 = no entry.  BETTER(v1, i1, v2, i2): entry 1 beats entry 2 (both present)
#ifdef PAL_ARG_SHFL
template <class BETTER> __device__ __forceinline__ void wave_arg63(double& v, int& i, BETTER better) {
  const int lane = threadIdx.x & 63;
  for (int o = 1; o < 64; o <<= 1) {
    const double ov = __shfl_up(v, o, 64);
    const int oi = __shfl_up(i, o, 64);
    const bool take = lane >= o && oi >= 0 && (i < 0 || better(ov, oi, v, i));
    v = take ? ov : v;
    i = take ? oi : i;
  }
}
template <class BETTER> __device__ __forceinline__ void wave_arg63_dpp(double& v, int& i, BETTER better) {
#else
template <class BETTER> __device__ __forceinline__ void wave_arg63(double& v, int& i, BETTER better) {
#endif
  auto step = [&](double ov, int oi) {
    const bool take = oi >= 0 && (i < 0 || better(ov, oi, v, i));
    v = take ? ov : v;
    i = take ? oi : i;
  };
  step(dpp_d<kShr1>(0.0, v), dpp_i<kShr1>(-1, i));
  step(dpp_d<kShr2>(0.0, v), dpp_i<kShr2>(-1, i));
  step(dpp_d<kShr4>(0.0, v), dpp_i<kShr4>(-1, i));
  step(dpp_d<kShr8>(0.0, v), dpp_i<kShr8>(-1, i));
  step(dpp_d<kBcast15, 0xa>(0.0, v), dpp_i<kBcast15, 0xa>(-1, i));
  step(dpp_d<kBcast31, 0xc>(0.0, v), dpp_i<kBcast31, 0xc>(-1, i));
}

__device__ __forceinline__ int wave_bcast63(int v) { return __builtin_amdgcn_readlane(v, 63); }
__device__ __forceinline__ double wave_bcast63(double v) {
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), 63), __builtin_amdgcn_readlane(__double2loint(v), 63));
}

}  // namespace pal
