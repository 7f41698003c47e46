// reg_fft.h - a 2^LM-point complex FFT held in the REGISTERS of M / PTS lanes (gfx950, fp64): PTS = 16 or 32 points per
// lane, Stockham autosort stages of radix 16 / 8 / 4 (fft_core.h stage plan) run on the lane's own registers, and LDS
// carries only the exchange between two stages, one PLANE at a time (real parts out, barrier, real parts in, barrier,
// then the imaginary parts): M doubles of LDS instead of M complex values.
//
// Users: the row tiles of the prime-factor route for N2 > 2048 (pfa_big.h: 8192 / 16384 points, 256 / 512 lanes) and the
// row pass of the four-step route (conv_kernels.h k_rows_wave: 1024 points on ONE wavefront, no workgroup barrier at all).
//
// Conventions: a stage of radix R runs PTS / R butterflies per lane (work items tid + LANES q); register q R + r holds
// input r of butterfly q, i.e. element tid + LANES (q + (PTS / R) r).  The last stage leaves lane `tid` with the elements
// tid + LANES s again - the same set it started with - so a pointwise product followed by the inverse transform needs
// no exchange, only a renaming of registers (BigTile::reg_of / slot_of).
#pragma once
#include "fft_core.h"

namespace pal {

template <int LM, int PTS> struct BigTile {
  static_assert(PTS == 16 || PTS == 32, "points per lane");
  static constexpr int kM = 1 << LM, kLanes = kM / PTS, kPts = PTS;
  // LDS position of element e of the plane: the low four bits are XOR-ed with the next four, which spreads the
  // stride-16 stores of the first stage over the banks (a ds_write_b64 conflicts inside groups of 16 lanes only) and
  // keeps 16 consecutive elements a permutation of 16 consecutive slots for the coalesced sides
  __device__ static __forceinline__ int pos(int e) { return e ^ ((e >> 4) & 15); }
  // register that holds element tid + LANES s before a stage of radix R / after the last stage of radix R
  __device__ static __forceinline__ constexpr int reg_of(int s, int R) { return (s % (PTS / R)) * R + s / (PTS / R); }
  __device__ static __forceinline__ constexpr int slot_of(int reg, int R) { return reg / R + (PTS / R) * (reg % R); }
  static constexpr int kR0 = stage_radix(LM, 0), kRL = stage_radix(LM, stage_tw_last(LM));
  // wavefronts per SIMD the register allocation must leave room for: two workgroups per CU where their LDS planes fit
  static constexpr int kWaves = LM == 13 ? 2 * (kLanes / 64) / 4 : 1;
};

// one stage on the PTS registers of a lane: PER = PTS / R butterflies (work items tid + LANES q), twiddles from the
// stage-major table in global memory (L2-resident: 16 B x 2^LM)
template <int LM, int PTS, bool INV, int LP>
__device__ __forceinline__ void big_stage(cd* v, const cd* __restrict__ tws, int tid) {
  constexpr int R = stage_radix(LM, LP), P = 1 << LP, LANES = BigTile<LM, PTS>::kLanes, PER = PTS / R;
#pragma unroll
  for (int q = 0; q < PER; ++q) {
    if constexpr (P > 1) {
      const int k = (tid + LANES * q) & (P - 1);
      const cd* t = tws + stage_tw_offset(LM, LP) + k;
#pragma unroll
      for (int r = 1; r < R; ++r) {
        const cd f = t[(r - 1) * P];
        v[q * R + r] = INV ? cmulc(v[q * R + r], f) : cmul(v[q * R + r], f);
      }
    }
    dftR<R, INV>(v + q * R);
  }
}

// exchange between the stage at LP (outputs at their autosort positions) and the next one (inputs i + r NB), one plane
// at a time through `plane` (M doubles)
template <int LM, int PTS, int LP>
__device__ __forceinline__ void big_exchange(double* plane, cd* v, int tid) {
  using B = BigTile<LM, PTS>;
  constexpr int M = 1 << LM, LANES = B::kLanes;
  constexpr int R = stage_radix(LM, LP), P = 1 << LP, PER = PTS / R;
  constexpr int LP2 = LP + stage_log2r(LM, LP), R2 = stage_radix(LM, LP2), PER2 = PTS / R2, NB2 = M / R2;
  int wpos[PER], rpos[PER2];
#pragma unroll
  for (int q = 0; q < PER; ++q) {
    const int i = tid + LANES * q, k = i & (P - 1);
    wpos[q] = (i - k) * R + k;
  }
#pragma unroll
  for (int q = 0; q < PER2; ++q) rpos[q] = tid + LANES * q;
  double xs[PTS];
#pragma unroll
  for (int q = 0; q < PER; ++q)
#pragma unroll
    for (int r = 0; r < R; ++r) plane[B::pos(wpos[q] + r * P)] = v[q * R + r].x;
  __syncthreads();
#pragma unroll
  for (int q = 0; q < PER2; ++q)
#pragma unroll
    for (int r = 0; r < R2; ++r) xs[q * R2 + r] = plane[B::pos(rpos[q] + r * NB2)];
  __syncthreads();
#pragma unroll
  for (int q = 0; q < PER; ++q)
#pragma unroll
    for (int r = 0; r < R; ++r) plane[B::pos(wpos[q] + r * P)] = v[q * R + r].y;
  __syncthreads();
#pragma unroll
  for (int q = 0; q < PER2; ++q)
#pragma unroll
    for (int r = 0; r < R2; ++r) v[q * R2 + r] = mk(xs[q * R2 + r], plane[B::pos(rpos[q] + r * NB2)]);
  __syncthreads();
}

template <int LM, int PTS, bool INV, int LP>
__device__ __forceinline__ void big_fft_from(double* plane, const cd* __restrict__ tws, cd* v, int tid) {
  big_stage<LM, PTS, INV, LP>(v, tws, tid);
  if constexpr (!stage_is_last(LM, LP)) {
    big_exchange<LM, PTS, LP>(plane, v, tid);
    big_fft_from<LM, PTS, INV, LP + stage_log2r(LM, LP)>(plane, tws, v, tid);
  }
}

// FFT_M of the PTS registers of every lane.  In: register BigTile::reg_of(s, kR0) = element tid + LANES s.  Out: register
// `reg` holds element tid + LANES BigTile::slot_of(reg, kRL) - the same SET of elements the lane started with.
template <int LM, int PTS, bool INV>
__device__ __forceinline__ void big_fft(double* plane, const cd* __restrict__ tws, cd* v, int tid) {
  static_assert(stage_radix(LM, 0) == 16, "radix-16 first stage");
  big_fft_from<LM, PTS, INV, 0>(plane, tws, v, tid);
}

}  // namespace pal
