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
#include "mixed_radix.h"

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
#ifndef PAL_BIG_TW_RECUR
#define PAL_BIG_TW_RECUR 1   // twiddles of the radix-4 / radix-8 stages: rows 1, 2 (, 4) loaded, the others one product deep
#endif

template <int LM, int PTS, bool INV, int LP>
__device__ __forceinline__ void big_stage(cd* v, const cd* __restrict__ tws, int tid) {
  constexpr int R = stage_radix(LM, LP), P = 1 << LP, LANES = BigTile<LM, PTS>::kLanes, PER = PTS / R;
#pragma unroll
  for (int q = 0; q < PER; ++q) {
    if constexpr (P > 1) {
      const int k = (tid + LANES * q) & (P - 1);
      const cd* t = tws + stage_tw_offset(LM, LP) + k;
      if constexpr (PAL_BIG_TW_RECUR && (R == 4 || R == 8)) {
        // the tables of the late stages (P >= 256: 28 ... 200 KB) do not stay in L1: fewer loads, a few more products
        cd f[R];
        f[1] = t[0];
        f[2] = t[P];
        f[3] = cmul(f[1], f[2]);
        if constexpr (R == 8) {
          f[4] = t[3 * P];
          f[5] = cmul(f[4], f[1]);
          f[6] = cmul(f[4], f[2]);
          f[7] = cmul(f[4], f[3]);
        }
#pragma unroll
        for (int r = 1; r < R; ++r) v[q * R + r] = INV ? cmulc(v[q * R + r], f[r]) : cmul(v[q * R + r], f[r]);
      } else {
#pragma unroll
        for (int r = 1; r < R; ++r) {
          const cd f = t[(r - 1) * P];
          v[q * R + r] = INV ? cmulc(v[q * R + r], f) : cmul(v[q * R + r], f);
        }
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

// ------------------------------------------------------------------ small DFTs of one lane
// N-point DFT of v[0 .. N) in natural order, N = 2^k <= 32 or 3 * 2^k <= 48, entirely in the registers of ONE lane (the
// column transforms of the four-step route when its rows are 8192 points long: conv_kernels.h k_colsreg_*).
// N = 3 P: 3-point butterflies over the inputs e, P + e, 2P + e, the twiddle exp(-/+ 2 pi i e s / N) on branch s, a P-point
// DFT per branch; output 3 k' + s.  N = 32: the same with two branches.  `roots` = exp(-2 pi i m / N), m < N, read through
// the scalar cache (wave-uniform, compile-time indices).
__device__ __forceinline__ cd uniform_root(const cd* roots, int m) {
  const auto* p = reinterpret_cast<const __attribute__((address_space(4))) double*>(reinterpret_cast<uintptr_t>(roots)) + 2 * m;
  return mk(p[0], p[1]);
}

template <int N, bool INV, int STRIDE = 1>   // roots[m * STRIDE] = exp(-2 pi i m / N): a table of N * STRIDE roots serves every divisor
__device__ __forceinline__ void reg_dft(cd* v, const cd* roots) {
  static_assert(N == 1 || N == 2 || N == 4 || N == 8 || N == 16 || N == 32 || N == 3 || N == 6 || N == 12 || N == 24 || N == 48 ||
                N == 18 || N == 20 || N == 22, "column length");
  if constexpr (N == 1) {
    return;
  } else if constexpr (N == 2 || N == 4 || N == 8 || N == 16) {
    dftR<N, INV>(v);
  } else if constexpr (N == 32 || N == 18 || N == 20 || N == 22) {
    // N = 2 P: X[2k] = DFT_P(x[e] + x[P + e]), X[2k + 1] = DFT_P((x[e] - x[P + e]) exp(-/+ 2 pi i e / N)); P = 16, or 9 / 10 / 11
    // on the real-symmetric butterflies of the Rader row pass (mixed_radix.h)
    constexpr int P = N / 2;
    cd y[2][P];
#pragma unroll
    for (int e = 0; e < P; ++e) {
      y[0][e] = v[e] + v[P + e];
      const cd d = v[e] - v[P + e];
      const cd w = uniform_root(roots, e * STRIDE);
      y[1][e] = e == 0 ? d : (INV ? cmulc(d, w) : cmul(d, w));
    }
    if constexpr (P == 16) {
      dftR<P, INV>(y[0]);
      dftR<P, INV>(y[1]);
    } else {
      dft_sym<P, INV>(y[0]);
      dft_sym<P, INV>(y[1]);
    }
#pragma unroll
    for (int k = 0; k < P; ++k) { v[2 * k] = y[0][k]; v[2 * k + 1] = y[1][k]; }
  } else {
    constexpr int P = N / 3;
    const double h = 0.86602540378443864676;            // sin(pi/3)
    cd y[3][P];
#pragma unroll
    for (int e = 0; e < P; ++e) {
      const cd a0 = v[e], a1 = v[P + e], a2 = v[2 * P + e];
      const cd sum = a1 + a2, dif = a1 - a2;
      const cd mid = mk(a0.x - 0.5 * sum.x, a0.y - 0.5 * sum.y);
      const cd rot = INV ? mk(-h * dif.y, h * dif.x) : mk(h * dif.y, -h * dif.x);   // (-/+ i sqrt(3)/2) (a1 - a2)
      y[0][e] = a0 + sum;
      cd y1 = mid + rot, y2 = mid - rot;
      if (e > 0) {
        const cd w1 = uniform_root(roots, e * STRIDE), w2 = uniform_root(roots, 2 * e * STRIDE);
        y1 = INV ? cmulc(y1, w1) : cmul(y1, w1);
        y2 = INV ? cmulc(y2, w2) : cmul(y2, w2);
      }
      y[1][e] = y1;
      y[2][e] = y2;
    }
    if constexpr (P > 1) {
      dftR<P, INV>(y[0]);
      dftR<P, INV>(y[1]);
      dftR<P, INV>(y[2]);
    }
#pragma unroll
    for (int k = 0; k < P; ++k) { v[3 * k] = y[0][k]; v[3 * k + 1] = y[1][k]; v[3 * k + 2] = y[2][k]; }
  }
}

}  // namespace pal
