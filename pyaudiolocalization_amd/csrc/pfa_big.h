// pfa_big.h - row pass of the prime-factor route for row lengths 2048 < N2 <= 8192 (gfx950, fp64).
//
// k_pfa_rows (pfa_kernels.h) keeps two chirp-convolution tiles of 2^lm <= 4096 complex points in LDS, which bounds N2 at
// 2048.  The lengths of BASELINE's configurations split further out - 47 999 = 7 x 6857, 95 999 = 17 x 5647 - and so do
// most of the lengths the synchronisation padding produces (utils.py:448-456): any n with an odd divisor N1 <= 127 whose
// cofactor stays under 8192.  A 16 384-point tile is 256 KB as complex doubles, more than a CU's 160 KB of LDS, so here
//
//   - the tile lives in REGISTERS: one workgroup of M / 32 lanes holding 32 points per lane - 256 lanes at 8192 points (two
//     workgroups per CU, their phases overlap), 512 lanes at 16 384 (one per CU); either way two wavefronts per SIMD and up
//     to 256 registers per lane.  (Round 2's first version held 16 points per lane: 1024 lanes at 16 384 points, pinned to
//     128 registers by its 16 wavefronts, 268 B of spills per lane - 441 us per launch of 240 x 7 rows against 322 now; the
//     8192-point tile measured 500 us per 240 x 23 rows with 16 points x 512 lanes, 435 with 32 x 256.)  One workgroup per row of Y (the two rows k1 / N1 - k1 that share their whitened bins
//     are two workgroups: the whitening is done twice, 8 % of the arithmetic, for half the registers);
//   - LDS only carries the exchange between two Stockham stages, one PLANE at a time: real parts out, barrier, real
//     parts in, barrier, then the imaginary parts - M doubles (128 KB at 16 384 points, 64 KB at 8192);
//   - a stage of radix R runs PTS / R butterflies per lane (work items tid + LANES q); register q R + r holds input
//     r of butterfly q, i.e. element tid + LANES (q + (PTS / R) r).  The last forward stage therefore leaves lane `tid`
//     with the bins tid + LANES s, the same set its first inverse butterflies read: the product with the chirp
//     spectrum needs no exchange, only a renaming of registers.
//
// Same transform as k_pfa_rows otherwise: x[e] = (R^p + i R^q)[k1, e] b[e] (or the conjugate combination of the
// reversed row for N1 - k1), circular convolution with the chirp kernel through FFT_M, chirp and column twiddle on the
// way out, Y[row][m2] to global memory.  The chirp-spectrum setup (k_pfa_hhat_big) runs the same forward transform.
#pragma once
#include "conv_kernels.h"
#include "pfa_kernels.h"
#include "reg_fft.h"


namespace pal {

// chirp spectrum: hhat[e] = scale * FFT_M(h)[e], h[d mod M] = conj(b[|d|]) for |d| < N2 (PfaChirpIn)
template <int LM, int PTS>
__global__ __launch_bounds__((BigTile<LM, PTS>::kLanes)) void k_pfa_hhat_big(const cd* __restrict__ b, int N2, cd* __restrict__ hhat, double scale,
                                                                           const cd* __restrict__ tws) {
  using B = BigTile<LM, PTS>;
  constexpr int M = 1 << LM, LANES = B::kLanes;
  __shared__ double plane[M];
  const int tid = threadIdx.x;
  cd v[PTS];
#pragma unroll
  for (int s = 0; s < PTS; ++s) {
    const int e = tid + LANES * s;
    v[B::reg_of(s, B::kR0)] = e < N2 ? cconj(b[e]) : (M - e < N2 ? cconj(b[M - e]) : mk(0, 0));
  }
  big_fft<LM, PTS, false>(plane, tws, v, tid);
#pragma unroll
  for (int reg = 0; reg < PTS; ++reg) hhat[tid + LANES * B::slot_of(reg, B::kRL)] = cscale(v[reg], scale);
}

// grid = G * N1 workgroups (row_work_item: transform fastest), one per row of Y
template <int LM, int PTS>
__global__ __launch_bounds__((BigTile<LM, PTS>::kLanes)) __attribute__((amdgpu_waves_per_eu(BigTile<LM, PTS>::kWaves))) void k_pfa_rows_big(PfaRowsArgs a) {
  using B = BigTile<LM, PTS>;
  constexpr int M = 1 << LM, LANES = B::kLanes, HALF = PTS / 2;
  __shared__ double plane[M];
  const int tid = threadIdx.x;
  int g, row;
  if (!row_work_item(blockIdx.x, a.G, a.N1, a.xcd, g, row)) return;
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
  // ---- inputs: the whitened pair bins times the chirp; elements beyond N2 <= M / 2 are zero padding (slots >= PTS / 2 always)
  cd v[PTS];
#pragma unroll
  for (int s0 = 0; s0 < HALF; s0 += 2) {                      // two elements at a time: eight spectrum loads in flight
    cd va[2], vb[2], vc[2], vd[2], ch[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int e = tid + LANES * (s0 + u), ee = e < N2 ? e : N2 - 1;
      va[u] = sa[ee]; vb[u] = sb[ee]; vc[u] = sc[ee]; vd[u] = sd[ee];
      ch[u] = a.b[ee];
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int e = tid + LANES * (s0 + u);
      const cd r1 = whiten(va[u], vb[u]);
      const cd r2 = cscale(whiten(vc[u], vd[u]), keep2);
      const cd x = rev ? mk(r1.x + r2.y, r2.x - r1.y) : mk(r1.x - r2.y, r1.y + r2.x);   // conj(R^p) + i conj(R^q), or R^p + i R^q
      v[B::reg_of(s0 + u, B::kR0)] = e < N2 ? cmul(x, ch[u]) : mk(0, 0);
    }
    if constexpr (PTS > 16) asm volatile("" ::: "memory");    // (keeps the next round's loads below: all 80 in flight would not fit the registers)
  }
#pragma unroll
  for (int s = HALF; s < PTS; ++s) v[B::reg_of(s, B::kR0)] = mk(0, 0);
  big_fft<LM, PTS, false>(plane, a.twfull, v, tid);
  // ---- product with the chirp spectrum: the lane holds bins tid + LANES s, the inputs of its first inverse butterflies
  {
    cd u[PTS];
#pragma unroll
    for (int reg = 0; reg < PTS; ++reg) {
      if constexpr (PTS > 16) { if (reg % 8 == 0) asm volatile("" ::: "memory"); }   // eight chirp-spectrum loads in flight, not all 32
      const int s = B::slot_of(reg, B::kRL);
      u[B::reg_of(s, B::kR0)] = cmul(v[reg], a.hhat[tid + LANES * s]);
    }
#pragma unroll
    for (int r = 0; r < PTS; ++r) v[r] = u[r];
  }
  {
    // The inverse stages use the same twiddles and the same LDS positions as the forward ones.  Seen through the same
    // values the compiler keeps all of them live from the forward transform on - about a hundred registers, spilled to
    // scratch; opaque copies of the table offset and of the lane index make it load / compute them again instead.
    size_t again = 0;
    int tid2 = tid;
    asm volatile("" : "+s"(again), "+v"(tid2));
    big_fft<LM, PTS, true>(plane, a.twfull + again, v, tid2);
  }
  // ---- outputs e = tid + LANES s < N2: chirp, column twiddle exp(2 pi i u1 row m2 / N1), Y[row][m2] with m2 = e, or
  //      -e mod N2 for the reversed rows
  if (rev && k1 == 0) return;                                  // (never: row 0 is not reversed)
  cd* const Yrow = a.Y + (size_t(g) * N1 + row) * N2;
  const auto* rt = reinterpret_cast<const __attribute__((address_space(4))) int*>(reinterpret_cast<uintptr_t>(a.rowtab)) + 2 * row;
  const unsigned uk = unsigned(rt[0]), n1 = unsigned(N1);
#pragma unroll
  for (int reg = 0; reg < PTS; ++reg) {
    if constexpr (PTS > 16) { if (reg % 8 == 0) asm volatile("" ::: "memory"); }
    const int s = B::slot_of(reg, B::kRL);
    const int e = tid + LANES * s;
    if (s < HALF && e < N2) {                                  // (N2 <= M / 2: the upper slots are never below N2)
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
