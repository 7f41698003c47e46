// pfa_rader89.h - the 89-point column DFT of the prime-factor route in O(N log N) form (gfx950, fp64).
//
// The dense real-symmetric column DFT of pfa_kernels.h spends 44 x 44 x 4 multiply-adds per column (both pairs of the
// packed transform): 45 % of the fp64 operations of the whole path at n = 88199 = 89 x 991.  89 is prime, so with a
// primitive root g (3) Rader's re-indexing t = g^a, j = g^-b turns
//     c[t] = sum_j Y[j] w^(j t),   w = exp(+2 pi i / 89)          (inverse transform, utils.py:118)
// into  c[g^a] = Y[0] + (u * h)[a],  c[0] = Y[0] + sum_b u[b],  u[b] = Y[g^-b],  h[d] = w^(g^d):  a CYCLIC convolution of
// length 88 = 8 x 11.  The factors are coprime, so b -> (b mod 8, b mod 11) makes it a two-dimensional cyclic
// convolution that the separable 8 x 11 DFT diagonalises - no twiddle factors between the axes (as in mixed_radix.h).
//
// One workgroup of four wavefronts transforms 62 columns, lane = column, and the 8 x 11 array of a column is spread
// over the wavefronts: wavefront w owns the rows b8 = w and w + 4 (22 values per lane).
//   stage A (registers)   11-point DFTs along b11 of both rows, then the radix-2 step of the 8-point DFT between them:
//                         e = A + B, o = (A - B) W8^w
//   exchange 1 (LDS)      wavefront w' collects e and o of all four wavefronts for ITS frequencies k11 (3 w' .. 3 w' + 2)
//   stage B (registers)   per k11: two 4-point DFTs complete the 8-point DFT, the product with the kernel's spectrum H
//                         (wave-uniform: scalar operands), two inverse 4-point DFTs; Y[0] enters at the origin, which
//                         also yields c[0]
//   exchange 2 (LDS)      back to the owner of the rows
//   stage C (registers)   radix-2 step with W8^-w, 11-point inverse DFTs: c[g^a] for a = CRT(b8, a11)
// 4640 vector instructions per column against 8100, and every wavefront loads only its 22 rows of Y (the dense form
// loads all 89 in every wavefront).  The exchanges move one plane (real parts, then imaginary parts) of 88 x 64 doubles
// at a time: 45 KB of LDS, which the histograms of the statistics reuse afterwards.
//
// The stage arithmetic is __host__ __device__: tests/host/test_rader89.cpp runs it with the exchanges emulated.
#pragma once
#include "fft_core.h"
#include "mixed_radix.h"

namespace pal {

constexpr int kR89 = 89;
constexpr int kR89Slots = 22;                 // values of a column per wavefront: two rows of eleven

struct Rader89Tab {                           // built by the host (make_rader89_tab)
  int rowsel[4][24];                          // [wave][slot]: input row j = g^-CRT(b8, b11), slot = 11 s + b11, b8 = wave + 4 s
  int tmap[4][24];                            // [wave][slot]: output index t = g^CRT(b8, a11)
  cd H[4][3][8];                              // [wave][q][k8]: kernel spectrum / 88 at k11 = 3 wave + q (zero where k11 > 10)
};

PAL_HD int r89_crt(int b8, int b11) {         // b < 88 with b = b8 (mod 8), b = b11 (mod 11):  33 = 1 (mod 8), 0 (mod 11);  56 = 0 (mod 8), 1 (mod 11)
  return (33 * b8 + 56 * b11) % 88;
}

inline void make_rader89_tab(Rader89Tab& t) {
  const int g = 3, p = kR89;                  // 3 is a primitive root of 89
  int gp[88], gm[88];
  long long x = 1;
  for (int a = 0; a < 88; ++a, x = x * g % p) gp[a] = int(x);
  for (int a = 0; a < 88; ++a) gm[a] = gp[(88 - a) % 88];      // g^-a
  for (int w = 0; w < 4; ++w)
    for (int s = 0; s < 24; ++s) { t.rowsel[w][s] = 0; t.tmap[w][s] = 0; }
  for (int w = 0; w < 4; ++w)
    for (int s = 0; s < 2; ++s)
      for (int b11 = 0; b11 < 11; ++b11) {
        const int b = r89_crt(w + 4 * s, b11);
        t.rowsel[w][11 * s + b11] = gm[b];
        t.tmap[w][11 * s + b11] = gp[b];
      }
  // H[k8][k11] = 1/88 sum_d h2[d8][d11] exp(-2 pi i (d8 k8 / 8 + d11 k11 / 11)),  h2[d8][d11] = w^(g^CRT(d8, d11))
  const long double two_pi = 6.283185307179586476925286766559005768L;
  for (int w = 0; w < 4; ++w)
    for (int q = 0; q < 3; ++q)
      for (int k8 = 0; k8 < 8; ++k8) {
        const int k11 = 3 * w + q;
        long double re = 0, im = 0;
        if (k11 < 11)
          for (int d8 = 0; d8 < 8; ++d8)
            for (int d11 = 0; d11 < 11; ++d11) {
              const long double ah = two_pi * (long double)gp[r89_crt(d8, d11)] / (long double)p;
              const long double ak = -two_pi * ((long double)((d8 * k8) % 8) / 8.0L + (long double)((d11 * k11) % 11) / 11.0L);
              const long double a = ah + ak;
              re += cosl(a);
              im += sinl(a);
            }
        t.H[w][q][k8] = mk(double(re / 88.0L), double(im / 88.0L));
      }
}

PAL_HD void r89_w8(int w, double& c, double& s) {             // W8^w = exp(-2 pi i w / 8) = (c, -s) in rotc's convention
  const double h = 0.70710678118654752440;
  c = w == 0 ? 1.0 : (w == 1 ? h : (w == 2 ? 0.0 : -h));
  s = w == 0 ? 0.0 : (w == 2 ? 1.0 : h);
}

// stage A: a[11], b[11] = the rows b8 = w and w + 4 of a column  ->  a = e, b = o (in place)
PAL_HD void r89_stage_a(cd* a, cd* b, int w) {
  dft_sym<11, false>(a);
  dft_sym<11, false>(b);
  double c, s;
  r89_w8(w, c, s);
#pragma unroll
  for (int k = 0; k < 11; ++k) {
    const cd sum = a[k] + b[k], dif = a[k] - b[k];
    a[k] = sum;
    b[k] = rotc<false>(dif, c, s);
  }
}

// stage B for one frequency k11: e[4], o[4] over the wavefronts  ->  z0[4], z1[4] (in place).  `origin` (k11 == 0): Y[0] is added at
// the origin of the spectrum, and c0 receives Y[0] + the sum of the column (the DFT's value at t = 0)
PAL_HD void r89_stage_b(cd* e, cd* o, const cd* H8, bool origin, cd y0, cd& c0) {
  dft4<false>(e);                                             // X[2 q]
  dft4<false>(o);                                             // X[2 q + 1]
  if (origin) c0 = y0 + e[0];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    e[q] = cmul(e[q], H8[2 * q]);
    o[q] = cmul(o[q], H8[2 * q + 1]);
  }
  if (origin) e[0] = e[0] + y0;
  dft4<true>(e);
  dft4<true>(o);
}

// stage C: z0[11], z1[11] of the rows b8 = w, w + 4  ->  the outputs of those rows (in place): z0 = row w, z1 = row w + 4
PAL_HD void r89_stage_c(cd* z0, cd* z1, int w) {
  double c, s;
  r89_w8(w, c, s);
#pragma unroll
  for (int k = 0; k < 11; ++k) {
    const cd t = rotc<true>(z1[k], c, s);
    const cd p = z0[k] + t, m = z0[k] - t;
    z0[k] = p;
    z1[k] = m;
  }
  dft_sym<11, true>(z0);
  dft_sym<11, true>(z1);
}

}  // namespace pal

#ifdef __HIPCC__
namespace pal {

// The column transform of one workgroup (four wavefronts, lane = column): loads, stages A - C and the two exchanges.
//   Yg     column of this lane: Yg[j * N2] = row j
//   xch    LDS, 88 x 64 doubles: one plane of an exchange at a time
//   out    [22]: c[t] for t = tab.tmap[w][i] (rows b8 = w: i < 11, b8 = w + 4: i >= 11); c0: c[0] (wavefront 0 only)
// Every wavefront of the workgroup must call it (eight barriers inside).
__device__ __forceinline__ void r89_columns(const cd* __restrict__ Yg, int N2, int w, int lane, const Rader89Tab* __restrict__ tab,
                                            double* __restrict__ xch, cd* out, cd& c0) {
  // wave-uniform tables through the scalar cache
  const auto* rs = reinterpret_cast<const __attribute__((address_space(4))) int*>(reinterpret_cast<uintptr_t>(&tab->rowsel[w][0]));
  const auto* Hs = reinterpret_cast<const __attribute__((address_space(4))) double*>(reinterpret_cast<uintptr_t>(&tab->H[w][0][0]));
  cd v[22];
#pragma unroll
  for (int i = 0; i < 22; ++i) v[i] = Yg[size_t(rs[i]) * N2];           // all 22 rows in flight at once
  const cd y0 = Yg[0];
  r89_stage_a(v, v + 11, w);                                          // v[k11] = e, v[11 + k11] = o
  // ---- exchange 1: wavefront w collects e, o of all four wavefronts at k11 = 3 w + q
  cd e4[3][4], o4[3][4];
  double* const mine = xch + lane;
#pragma unroll
  for (int plane = 0; plane < 2; ++plane) {
#pragma unroll
    for (int i = 0; i < 22; ++i) mine[((w * 2 + i / 11) * 11 + i % 11) * 64] = plane ? v[i].y : v[i].x;
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      const int k11 = 3 * w + q < 11 ? 3 * w + q : 10;                  // (wavefront 3 has two frequencies: the third slot repeats the last one, unused)
#pragma unroll
      for (int ws = 0; ws < 4; ++ws) {
        const double a = mine[((ws * 2 + 0) * 11 + k11) * 64], b = mine[((ws * 2 + 1) * 11 + k11) * 64];
        if (plane) { e4[q][ws].y = a; o4[q][ws].y = b; } else { e4[q][ws].x = a; o4[q][ws].x = b; }
      }
    }
    __syncthreads();
  }
  // ---- stage B
  c0 = mk(0, 0);
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    cd H8[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) H8[k] = mk(Hs[(q * 8 + k) * 2], Hs[(q * 8 + k) * 2 + 1]);
    cd c0q = mk(0, 0);
    r89_stage_b(e4[q], o4[q], H8, q == 0 && w == 0, y0, c0q);
    if (q == 0) c0 = c0q;
  }
  // ---- exchange 2: z0, z1 of row pair ws at k11 go back to wavefront ws
#pragma unroll
  for (int plane = 0; plane < 2; ++plane) {
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      const int k11 = 3 * w + q;
      if (k11 < 11) {                                                    // (uniform)
#pragma unroll
        for (int ws = 0; ws < 4; ++ws) {
          mine[((ws * 2 + 0) * 11 + k11) * 64] = plane ? e4[q][ws].y : e4[q][ws].x;
          mine[((ws * 2 + 1) * 11 + k11) * 64] = plane ? o4[q][ws].y : o4[q][ws].x;
        }
      }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 22; ++i) {
      const double a = mine[((w * 2 + i / 11) * 11 + i % 11) * 64];
      if (plane) out[i].y = a; else out[i].x = a;
    }
    __syncthreads();
  }
  r89_stage_c(out, out + 11, w);
}

}  // namespace pal
#endif
