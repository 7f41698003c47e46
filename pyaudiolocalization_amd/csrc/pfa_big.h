// pfa_big.h - row pass of the prime-factor route for row lengths 2048 < N2 <= 8192 (gfx950, fp64).
//
// k_pfa_rows (pfa_kernels.h) keeps two chirp-convolution tiles of 2^lm <= 4096 complex points in LDS, which bounds N2 at
// 2048.  The lengths of BASELINE's configurations split further out - 47 999 = 7 x 6857, 95 999 = 17 x 5647 - and so do
// most of the lengths the synchronisation padding produces (utils.py:448-456): any n with an odd divisor N1 <= 127 whose
// cofactor stays under 8192.  A 16 384-point tile is 256 KB as complex doubles, more than a CU's 160 KB of LDS, so here
//
//   - the tile lives in REGISTERS: one workgroup of M / 16 lanes (1024 at 16 384 points), 16 points per lane, one
//     workgroup per row of Y (the two rows k1 / N1 - k1 that share their whitened bins are two workgroups: the
//     whitening is done twice, 8 % of the arithmetic, for half the registers);
//   - LDS only carries the exchange between two Stockham stages, one PLANE at a time: real parts out, barrier, real
//     parts in, barrier, then the imaginary parts - M doubles (128 KB at 16 384 points, one workgroup per CU; 64 KB and
//     two per CU at 8192);
//   - the last forward stage leaves lane `tid` with the bins tid + LANES s, which are exactly the inputs of its first
//     inverse butterfly: the product with the chirp spectrum needs no exchange.
//
// Same transform as k_pfa_rows otherwise: x[e] = (R^p + i R^q)[k1, e] b[e] (or the conjugate combination of the
// reversed row for N1 - k1), circular convolution with the chirp kernel through FFT_M, chirp and column twiddle on the
// way out, Y[row][m2] to global memory.  The chirp-spectrum setup (k_pfa_hhat_big) runs the same forward transform.
#pragma once
#include "conv_kernels.h"
#include "pfa_kernels.h"

namespace pal {

template <int LM> struct BigTile {
  static constexpr int kM = 1 << LM, kLanes = kM / 16;
  // LDS position of element e of the plane: the low four bits are XOR-ed with the next four, which spreads the
  // stride-16 stores of the first stage over the banks (a ds_write_b64 conflicts inside groups of 16 lanes only) and
  // keeps 16 consecutive elements a permutation of 16 consecutive slots for the coalesced sides
  __device__ static __forceinline__ int pos(int e) { return e ^ ((e >> 4) & 15); }
};

// one stage on the 16 registers of a lane: PER = 16 / R butterflies (work items tid + LANES q), twiddles from the
// stage-major table in global memory (L2-resident: 16 B x 2^LM)
template <int LM, bool INV, int LP>
__device__ __forceinline__ void big_stage(cd* v, const cd* __restrict__ tws, int tid) {
  constexpr int R = stage_radix(LM, LP), P = 1 << LP, LANES = BigTile<LM>::kLanes, PER = 16 / R;
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
template <int LM, int LP>
__device__ __forceinline__ void big_exchange(double* plane, cd* v, int tid) {
  constexpr int M = 1 << LM, LANES = BigTile<LM>::kLanes;
  constexpr int R = stage_radix(LM, LP), P = 1 << LP, PER = 16 / R;
  constexpr int LP2 = LP + stage_log2r(LM, LP), R2 = stage_radix(LM, LP2), PER2 = 16 / R2, NB2 = M / R2;
  int wpos[PER], rpos[PER2];
#pragma unroll
  for (int q = 0; q < PER; ++q) {
    const int i = tid + LANES * q, k = i & (P - 1);
    wpos[q] = (i - k) * R + k;
  }
#pragma unroll
  for (int q = 0; q < PER2; ++q) rpos[q] = tid + LANES * q;
  double xs[16];
#pragma unroll
  for (int q = 0; q < PER; ++q)
#pragma unroll
    for (int r = 0; r < R; ++r) plane[BigTile<LM>::pos(wpos[q] + r * P)] = v[q * R + r].x;
  __syncthreads();
#pragma unroll
  for (int q = 0; q < PER2; ++q)
#pragma unroll
    for (int r = 0; r < R2; ++r) xs[q * R2 + r] = plane[BigTile<LM>::pos(rpos[q] + r * NB2)];
  __syncthreads();
#pragma unroll
  for (int q = 0; q < PER; ++q)
#pragma unroll
    for (int r = 0; r < R; ++r) plane[BigTile<LM>::pos(wpos[q] + r * P)] = v[q * R + r].y;
  __syncthreads();
#pragma unroll
  for (int q = 0; q < PER2; ++q)
#pragma unroll
    for (int r = 0; r < R2; ++r) v[q * R2 + r] = mk(xs[q * R2 + r], plane[BigTile<LM>::pos(rpos[q] + r * NB2)]);
  __syncthreads();
}

template <int LM, bool INV, int LP>
__device__ __forceinline__ void big_fft_from(double* plane, const cd* __restrict__ tws, cd* v, int tid) {
  big_stage<LM, INV, LP>(v, tws, tid);
  if constexpr (!stage_is_last(LM, LP)) {
    big_exchange<LM, LP>(plane, v, tid);
    big_fft_from<LM, INV, LP + stage_log2r(LM, LP)>(plane, tws, v, tid);
  }
}

// FFT_M of the 16 registers of every lane.  In: v[r] = element tid + LANES r (first-stage butterfly of a radix-16 first
// stage).  Out: the last stage's outputs; with RL = its radix and PER = 16 / RL, register q RL + r holds element
// tid + LANES (q + PER r) - the same SET of elements the lane started with.
template <int LM, bool INV>
__device__ __forceinline__ void big_fft(double* plane, const cd* __restrict__ tws, cd* v, int tid) {
  static_assert(stage_radix(LM, 0) == 16, "radix-16 first stage");
  big_fft_from<LM, INV, 0>(plane, tws, v, tid);
}
template <int LM> __device__ __forceinline__ constexpr int big_out_slot(int reg) {   // s of element tid + LANES s held by register `reg` after big_fft
  constexpr int RL = stage_radix(LM, stage_tw_last(LM)), PER = 16 / RL;
  return reg / RL + PER * (reg % RL);
}

// chirp spectrum: hhat[e] = scale * FFT_M(h)[e], h[d mod M] = conj(b[|d|]) for |d| < N2 (PfaChirpIn)
template <int LM>
__global__ __launch_bounds__(BigTile<LM>::kLanes) void k_pfa_hhat_big(const cd* __restrict__ b, int N2, cd* __restrict__ hhat, double scale,
                                                                      const cd* __restrict__ tws) {
  constexpr int M = 1 << LM, LANES = BigTile<LM>::kLanes;
  __shared__ double plane[M];
  const int tid = threadIdx.x;
  cd v[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int e = tid + LANES * r;
    v[r] = e < N2 ? cconj(b[e]) : (M - e < N2 ? cconj(b[M - e]) : mk(0, 0));
  }
  big_fft<LM, false>(plane, tws, v, tid);
#pragma unroll
  for (int reg = 0; reg < 16; ++reg) hhat[tid + LANES * big_out_slot<LM>(reg)] = cscale(v[reg], scale);
}

// grid = G * N1 workgroups (transform fastest), one per row of Y
template <int LM>
__global__ __launch_bounds__(BigTile<LM>::kLanes) void k_pfa_rows_big(PfaRowsArgs a) {
  constexpr int M = 1 << LM, LANES = BigTile<LM>::kLanes;
  __shared__ double plane[M];
  const int tid = threadIdx.x;
  const int g = blockIdx.x % a.G, row = blockIdx.x / a.G;
  const int N1 = a.N1, N2 = a.N2;
  const bool rev = 2 * row > N1 - 1;                          // rows above (N1-1)/2: the reversed conjugate combination of row N1 - row
  const int k1 = rev ? N1 - row : row;
  const auto* qp = reinterpret_cast<const __attribute__((address_space(4))) int*>(reinterpret_cast<uintptr_t>(a.quad)) + 4 * g;
  const int4 q = make_int4(qp[0], qp[1], qp[2], qp[3]);
  const size_t mic = size_t(a.NR) * N2, off = size_t(k1) * N2;
  const bool second = q.z >= 0;
  const cd* sa = a.SP + size_t(q.x) * mic + off;
  const cd* sb = a.SP + size_t(q.y) * mic + off;
  const cd* sc = second ? a.SP + size_t(q.z) * mic + off : sa;   // branch-free loads: a missing second pair re-reads the first one
  const cd* sd = second ? a.SP + size_t(q.w) * mic + off : sb;
  const double keep2 = second ? 1.0 : 0.0;
  // ---- inputs: the whitened pair bins times the chirp; elements beyond N2 <= M / 2 are zero padding (r >= 8 always)
  cd v[16];
#pragma unroll
  for (int r0 = 0; r0 < 8; r0 += 2) {                         // two elements at a time: eight spectrum loads in flight
    cd va[2], vb[2], vc[2], vd[2], ch[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int e = tid + LANES * (r0 + u), ee = e < N2 ? e : N2 - 1;
      va[u] = sa[ee]; vb[u] = sb[ee]; vc[u] = sc[ee]; vd[u] = sd[ee];
      ch[u] = a.b[ee];
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int e = tid + LANES * (r0 + u);
      const cd r1 = whiten(va[u], vb[u]);
      const cd r2 = cscale(whiten(vc[u], vd[u]), keep2);
      const cd x = rev ? mk(r1.x + r2.y, r2.x - r1.y) : mk(r1.x - r2.y, r1.y + r2.x);   // conj(R^p) + i conj(R^q), or R^p + i R^q
      v[r0 + u] = e < N2 ? cmul(x, ch[u]) : mk(0, 0);
    }
  }
#pragma unroll
  for (int r = 8; r < 16; ++r) v[r] = mk(0, 0);
  big_fft<LM, false>(plane, a.twfull, v, tid);
  // ---- product with the chirp spectrum: the lane holds bins tid + LANES s, the inputs of its first inverse butterfly
  {
    cd u[16];
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
      const int s = big_out_slot<LM>(reg);
      u[s] = cmul(v[reg], a.hhat[tid + LANES * s]);
    }
#pragma unroll
    for (int s = 0; s < 16; ++s) v[s] = u[s];
  }
  big_fft<LM, true>(plane, a.twfull, v, tid);
  // ---- outputs e = tid + LANES s < N2: chirp, column twiddle exp(2 pi i u1 row m2 / N1), Y[row][m2] with m2 = e, or
  //      -e mod N2 for the reversed rows
  if (rev && k1 == 0) return;                                  // (never: row 0 is not reversed)
  cd* const Yrow = a.Y + (size_t(g) * N1 + row) * N2;
  const auto* rt = reinterpret_cast<const __attribute__((address_space(4))) int*>(reinterpret_cast<uintptr_t>(a.rowtab)) + 2 * row;
  const unsigned uk = unsigned(rt[0]), n1 = unsigned(N1);
#pragma unroll
  for (int reg = 0; reg < 16; ++reg) {
    const int s = big_out_slot<LM>(reg);
    const int e = tid + LANES * s;
    if (s < 8 && e < N2) {                                     // (N2 <= M / 2: slots 8 .. 15 are never below N2)
      const int m2 = rev ? (e ? N2 - e : 0) : e;
      const unsigned x = uk * unsigned(m2);                   // < 127 * 8192 < 2^24: exact in float
      unsigned idx = x - unsigned(float(x) * a.inv) * n1;     // x mod N1, off by at most one N1 either way
      idx = min(idx, idx + n1);
      idx = min(idx, idx - n1);
      Yrow[m2] = cmulc(cmul(v[reg], a.b[e]), a.r1[idx]);      // r1 holds exp(-2 pi i q / N1)
    }
  }
}

}  // namespace pal
