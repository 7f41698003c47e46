// mixed_radix.h - radix 9 / 10 / 11 butterflies and the prime-factor stages of the Rader row pass (gfx950,
// fp64).  Rader's algorithm turns the N2-point DFT of a prime N2 into a cyclic convolution of length N2 - 1; for
// N2 = 991 that is 990 = 11 x 9 x 10 points instead of the 2048-point chirp convolution (pfa_rader.h).
//
// The butterflies use the real-symmetric form of a length-R DFT: with a_k = x_k + x_{R-k}, b_k = x_k - x_{R-k},
//   X_m, X_{R-m} = x_0 + sum_k cos(2 pi k m / R) a_k  -/+  i sum_k sin(2 pi k m / R) b_k        (forward)
// i.e. ((R-1)/2)^2 real-by-complex multiply-adds for the cosine part and as many for the sine part.  The tables
// below were generated with 40-digit arithmetic (tools: mpmath) and are exact to the last bit of a double.
//
// The index math is __host__ __device__ so that tests/host/test_fft_core.cpp runs the same stage code on the CPU.
#pragma once
#include "fft_core.h"

namespace pal {

template <int R> struct Roots;   // cos / sin of 2 pi q / R
template <> struct Roots<9> {
  static PAL_HD constexpr double c(int q) {
    constexpr double t[9] = {1.0, 7.660444431189780352e-1, 1.7364817766693034885e-1, -5.0e-1, -9.3969262078590838405e-1, -9.3969262078590838405e-1, -5.0e-1, 1.7364817766693034885e-1, 7.660444431189780352e-1};
    return t[q];
  }
  static PAL_HD constexpr double s(int q) {
    constexpr double t[9] = {0.0, 6.4278760968653932632e-1, 9.8480775301220805937e-1, 8.6602540378443864676e-1, 3.4202014332566873304e-1, -3.4202014332566873304e-1, -8.6602540378443864676e-1, -9.8480775301220805937e-1, -6.4278760968653932632e-1};
    return t[q];
  }
};
template <> struct Roots<10> {
  static PAL_HD constexpr double c(int q) {
    constexpr double t[10] = {1.0, 8.090169943749474241e-1, 3.090169943749474241e-1, -3.090169943749474241e-1, -8.090169943749474241e-1, -1.0, -8.090169943749474241e-1, -3.090169943749474241e-1, 3.090169943749474241e-1, 8.090169943749474241e-1};
    return t[q];
  }
  static PAL_HD constexpr double s(int q) {
    constexpr double t[10] = {0.0, 5.8778525229247312917e-1, 9.5105651629515357212e-1, 9.5105651629515357212e-1, 5.8778525229247312917e-1, 4.1340642196527976473e-43, -5.8778525229247312917e-1, -9.5105651629515357212e-1, -9.5105651629515357212e-1, -5.8778525229247312917e-1};
    return t[q];
  }
};
template <> struct Roots<11> {
  static PAL_HD constexpr double c(int q) {
    constexpr double t[11] = {1.0, 8.4125353283118116886e-1, 4.1541501300188642553e-1, -1.4231483827328514044e-1, -6.5486073394528506406e-1, -9.5949297361449738989e-1, -9.5949297361449738989e-1, -6.5486073394528506406e-1, -1.4231483827328514044e-1, 4.1541501300188642553e-1, 8.4125353283118116886e-1};
    return t[q];
  }
  static PAL_HD constexpr double s(int q) {
    constexpr double t[11] = {0.0, 5.4064081745559758211e-1, 9.0963199535451837141e-1, 9.8982144188093273238e-1, 7.5574957435425828377e-1, 2.8173255684142969771e-1, -2.8173255684142969771e-1, -7.5574957435425828377e-1, -9.8982144188093273238e-1, -9.0963199535451837141e-1, -5.4064081745559758211e-1};
    return t[q];
  }
};

// length-R DFT in registers, R = 9, 10 or 11 (any R with a Roots<R> table), natural order in and out
template <int R, bool INV> PAL_HD void dft_sym(cd* v) {
  constexpr int H = (R - 1) / 2;
  constexpr bool EVEN = (R % 2) == 0;
  cd a[H + 1], b[H + 1];
#pragma unroll
  for (int k = 1; k <= H; ++k) {
    a[k] = v[k] + v[R - k];
    b[k] = v[k] - v[R - k];
  }
  const cd x0 = v[0];
  const cd xm = EVEN ? v[R / 2] : mk(0, 0);           // the self-paired input of an even length
  cd out[R];
  cd sum = x0;
#pragma unroll
  for (int k = 1; k <= H; ++k) sum = sum + a[k];
  out[0] = EVEN ? sum + xm : sum;
#pragma unroll
  for (int m = 1; m <= H; ++m) {
    cd cr = EVEN ? ((m & 1) ? x0 - xm : x0 + xm) : x0;
    cd si = mk(0, 0);
#pragma unroll
    for (int k = 1; k <= H; ++k) {
      const int q = (k * m) % R;
      const double c = Roots<R>::c(q), s = Roots<R>::s(q);
      cr = mk(__builtin_fma(c, a[k].x, cr.x), __builtin_fma(c, a[k].y, cr.y));
      si = mk(__builtin_fma(s, b[k].x, si.x), __builtin_fma(s, b[k].y, si.y));
    }
    // forward: e^{-i t} = cos t - i sin t, so X_m = cr - i si and X_{R-m} = cr + i si; the inverse swaps them
    const cd lo = mk(cr.x + si.y, cr.y - si.x), hi = mk(cr.x - si.y, cr.y + si.x);
    out[m] = INV ? hi : lo;
    out[R - m] = INV ? lo : hi;
  }
  if (EVEN) {                                         // X_{R/2} = sum_k (-1)^k x_k
    cd alt = ((R / 2) & 1) ? x0 - xm : x0 + xm;
#pragma unroll
    for (int k = 1; k <= H; ++k) alt = (k & 1) ? alt - a[k] : alt + a[k];
    out[R / 2] = alt;
  }
#pragma unroll
  for (int r = 0; r < R; ++r) v[r] = out[r];
}

// length 10 = 2 x 5 by the prime-factor map (no twiddles between coprime factors): inputs n = 5 n1 + 2 n2, outputs
// k = 5 k1 + 6 k2 (mod 10): five radix-2 butterflies, then two length-5 DFTs in the symmetric form.  92 real
// operations against 124 for the generic form above; the seam of the Rader row pass runs two of these per lane.
template <bool INV> PAL_HD void dft5_sym(cd& u0, cd& u1, cd& u2, cd& u3, cd& u4) {
  constexpr double c1 = Roots<10>::c(2), c2 = Roots<10>::c(4), s1 = Roots<10>::s(2), s2 = Roots<10>::s(4);   // 2 pi / 5, 4 pi / 5
  const cd a1 = u1 + u4, a2 = u2 + u3, b1 = u1 - u4, b2 = u2 - u3;
  const cd x0 = u0;
  u0 = x0 + a1 + a2;
  const cd cr1 = mk(__builtin_fma(c2, a2.x, __builtin_fma(c1, a1.x, x0.x)), __builtin_fma(c2, a2.y, __builtin_fma(c1, a1.y, x0.y)));
  const cd cr2 = mk(__builtin_fma(c1, a2.x, __builtin_fma(c2, a1.x, x0.x)), __builtin_fma(c1, a2.y, __builtin_fma(c2, a1.y, x0.y)));
  const cd si1 = mk(__builtin_fma(s2, b2.x, s1 * b1.x), __builtin_fma(s2, b2.y, s1 * b1.y));
  const cd si2 = mk(__builtin_fma(-s1, b2.x, s2 * b1.x), __builtin_fma(-s1, b2.y, s2 * b1.y));
  // forward: X_m = cr - i si, X_{5-m} = cr + i si; the inverse swaps them
  const cd lo1 = mk(cr1.x + si1.y, cr1.y - si1.x), hi1 = mk(cr1.x - si1.y, cr1.y + si1.x);
  const cd lo2 = mk(cr2.x + si2.y, cr2.y - si2.x), hi2 = mk(cr2.x - si2.y, cr2.y + si2.x);
  u1 = INV ? hi1 : lo1;
  u4 = INV ? lo1 : hi1;
  u2 = INV ? hi2 : lo2;
  u3 = INV ? lo2 : hi2;
}
template <bool INV> PAL_HD void dft10_pfa(cd* v) {
  cd e0 = v[0] + v[5], e1 = v[2] + v[7], e2 = v[4] + v[9], e3 = v[6] + v[1], e4 = v[8] + v[3];   // k1 = 0
  cd o0 = v[0] - v[5], o1 = v[2] - v[7], o2 = v[4] - v[9], o3 = v[6] - v[1], o4 = v[8] - v[3];   // k1 = 1
  dft5_sym<INV>(e0, e1, e2, e3, e4);
  dft5_sym<INV>(o0, o1, o2, o3, o4);
  v[0] = e0; v[6] = e1; v[2] = e2; v[8] = e3; v[4] = e4;       // k = 6 k2 mod 10
  v[5] = o0; v[1] = o1; v[7] = o2; v[3] = o3; v[9] = o4;       // k = 5 + 6 k2 mod 10
}
template <> PAL_HD void dft_sym<10, false>(cd* v) { dft10_pfa<false>(v); }
template <> PAL_HD void dft_sym<10, true>(cd* v) { dft10_pfa<true>(v); }

// plain (unswizzled) tile of `pitch` elements per sub-transform: the odd strides of these stages spread over the banks
struct PlainTile {
  static constexpr bool kLds = true;
  cd* data;
  int pitch;
  PAL_HD cd operator()(int t, int e) const { return data[t * pitch + e]; }
  PAL_HD void operator()(int t, int e, cd v) const { data[t * pitch + e] = v; }
};

// The cyclic convolution of length L = R1 R2 R3 with pairwise coprime factors (990 = 11 x 9 x 10) needs no twiddles at
// all: s -> (s mod R2, s mod R3, s mod R1) is a ring isomorphism of Z_L onto Z_R2 x Z_R3 x Z_R1, so the convolution
// over Z_L is a three-dimensional cyclic convolution in those coordinates, and the separable 3-D DFT (plain length-R
// DFTs along each axis) diagonalises it.  Element s lives at
//     pos(s) = (s mod R2) + R2 (s mod R3) + R2 R3 (s mod R1)
// of its tile.  A stage is the set of butterflies along one axis: every butterfly reads and writes the SAME R elements
// (in place, no autosort), so a stage needs one barrier (behind its stores) instead of two, and the three axes can run
// in any order.
template <int R1, int R2, int R3> struct Axes {
  static constexpr int L = R1 * R2 * R3;
  PAL_HD static constexpr int pos(int s) { return s % R2 + R2 * (s % R3) + R2 * R3 * (s % R1); }
  // butterfly j of the stage along an axis: first element and element stride
  PAL_HD static constexpr int base1(int j) { return j; }                                  // axis R1: j < R2 R3, stride R2 R3
  PAL_HD static constexpr int base2(int j) { return R2 * j; }                             // axis R2: j < R3 R1, stride 1
  PAL_HD static constexpr int base3(int j) { return j % R2 + R2 * R3 * (j / R2); }        // axis R3: j < R2 R1, stride R2
  static constexpr int kStride1 = R2 * R3, kStride2 = 1, kStride3 = R2;
};

template <int R, class In> PAL_HD void axis_load(const In& in, int t, int base, int stride, cd* v) {
#pragma unroll
  for (int r = 0; r < R; ++r) v[r] = in(t, base + r * stride);
}
template <int R, class Out> PAL_HD void axis_store(const Out& out, int t, int base, int stride, const cd* v) {
#pragma unroll
  for (int r = 0; r < R; ++r) out(t, base + r * stride, v[r]);
}

}  // namespace pal
