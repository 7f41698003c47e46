// pfa_forward.h - forward spectra of the microphone frames through the same prime-factor cut as the PHAT inverse
// (gfx950, fp64): n = N1 N2, input index m = m2 + N2 t, output index k = CRT(k1, k2):
//
//     X[k1, k2] = sum_m2 e^{-2 pi i u2 k2 m2 / N2} . e^{-2 pi i u1 k1 m2 / N1} . sum_t e^{-2 pi i k1 t / N1} z[m2 + N2 t]
//
// i.e. the transpose of pfa.hip's inverse: dense N1-point DFTs down the columns first (k_pfa_fwd_cols: the frames are
// real and zero beyond L = (n + 1) / 2, so only t <= (N1 - 1) / 2 contributes), then the column twiddle and N2-point
// DFTs along the rows (k_pfa_fwd_rows_rader: Rader's cyclic convolution of N2 - 1 = 990 points, pfa_rader.h).  Two real
// frames ride one complex transform (z = x_a + i x_b); the row kernel owns rows k1 and N1 - k1 and separates them on the
// way out, X_a[k] = (Z[k] + conj Z[n - k]) / 2, X_b[k] = (Z[k] - conj Z[n - k]) / 2i, writing SP[mic][k1][.] in the
// generator order the inverse's row pass reads (position s holds bin k2 = g^-s, bin 0 last).
//
// Replaces, for plans with Rader rows, the four-step chirp convolution of the forward transform (bluestein.hip:
// one real frame per 196608-point transform, three passes over the workspace): numpy.fft.fft(sig, n) of utils.py:114-115.
#pragma once
#include "pfa_rader.h"

namespace pal {

struct PfaFwdColsArgs {
  const double* frames;  // [rows][stride]
  size_t stride;
  int len, rows;         // samples per frame (the rest of the n points is zero), frames in this group
  cd* Y;                 // [G][N1][N2]: column DFTs A[k1][m2] of the packed frames
  const double* T;       // cos / sin table of the inverse's column pass (symmetric in its two indices)
  int N1, N2, G, nch;
};

// grid = (G * ceil(N2 / 64), ceil(nch / 4)); one lane per column m2, the wavefronts take chunks of TC output rows k1
// (and their mirrors N1 - k1).  With a = x_a[m2 + N2 t], b = x_b[m2 + N2 t], c/s = cos/sin(2 pi k1 t / N1):
//     A[k1]      = sum a c + sum b s + i (sum b c - sum a s),      A[N1 - k1] = sum a c - sum b s + i (sum b c + sum a s)
template <int TC, int UNR>
__global__ __launch_bounds__(256) void k_pfa_fwd_cols(PfaFwdColsArgs a) {
  const int lane = threadIdx.x & 63;
  const int ch = int(blockIdx.y) * 4 + __builtin_amdgcn_readfirstlane(int(threadIdx.x) >> 6);
  if (ch >= a.nch) return;
  const int N1 = a.N1, N2 = a.N2, h = (N1 - 1) / 2;
  const int g = blockIdx.x % a.G, cb = blockIdx.x / a.G;
  const int m2 = cb * 64 + lane;
  const bool live = m2 < N2;
  const int m2c = live ? m2 : N2 - 1;
  const bool second = 2 * g + 1 < a.rows;
  const double* xa = a.frames + size_t(2 * g) * a.stride;
  const double* xb = second ? xa + a.stride : xa;
  const int last = a.len - 1;
  // branch-free loads: the index is clamped and the value masked where it is USED (a select on the loaded value right
  // behind the load would make every load wait for its data: the batches must stay in flight)
  auto at = [&](int t) { const int m = m2c + N2 * t; return m <= last ? m : last; };
  auto inside = [&](int t) { return m2c + N2 * t <= last; };
  double ac[TC], as[TC], bc[TC], bs[TC];
#pragma unroll
  for (int tt = 0; tt < TC; ++tt) ac[tt] = as[tt] = bc[tt] = bs[tt] = 0.0;
  double suma = 0.0, sumb = 0.0;
  const double ra0 = xa[at(0)], rb0 = xb[at(0)];
  const double* Tj = a.T + size_t(ch) * 2 * TC;
  const size_t tstep = size_t(a.nch) * 2 * TC;
  double va[UNR], vb[UNR];
#pragma unroll
  for (int u = 0; u < UNR; ++u) {
    const int t = 1 + u <= h ? 1 + u : h;
    va[u] = xa[at(t)];
    vb[u] = xb[at(t)];
  }
  const double a0 = inside(0) ? ra0 : 0.0, b0 = second && inside(0) ? rb0 : 0.0;
  for (int t = 1; t <= h; t += UNR) {
    double na[UNR], nb[UNR];
#pragma unroll
    for (int u = 0; u < UNR; ++u) {                           // next batch; past the end the last step is re-read: no branch
      const int tn = t + UNR + u <= h ? t + UNR + u : h;
      na[u] = xa[at(tn)];
      nb[u] = xb[at(tn)];
    }
#pragma unroll
    for (int u = 0; u < UNR; ++u, Tj += tstep) {              // steps beyond h meet zero rows of the table
      const bool in = t + u <= h && inside(t + u);
      const double av = in ? va[u] : 0.0, bv = in && second ? vb[u] : 0.0;
      suma += av;
      sumb += bv;
#pragma unroll
      for (int tt = 0; tt < TC; ++tt) {
        const double c = Tj[tt], sn = Tj[TC + tt];
        ac[tt] = __builtin_fma(c, av, ac[tt]);
        as[tt] = __builtin_fma(sn, av, as[tt]);
        bc[tt] = __builtin_fma(c, bv, bc[tt]);
        bs[tt] = __builtin_fma(sn, bv, bs[tt]);
      }
    }
#pragma unroll
    for (int u = 0; u < UNR; ++u) { va[u] = na[u]; vb[u] = nb[u]; }
  }
  if (!live) return;
  cd* Yg = a.Y + size_t(g) * N1 * N2 + m2;
  if (ch == 0) Yg[0] = mk(a0 + suma, b0 + sumb);
#pragma unroll
  for (int tt = 0; tt < TC; ++tt) {
    const int k1 = ch * TC + tt + 1;
    if (k1 <= h) {
      Yg[size_t(N2) * k1] = mk(a0 + ac[tt] + bs[tt], b0 + bc[tt] - as[tt]);
      Yg[size_t(N2) * (N1 - k1)] = mk(a0 + ac[tt] - bs[tt], b0 + bc[tt] + as[tt]);
    }
  }
}

struct PfaFwdRowsArgs {
  const cd* Y;           // [G][N1][N2] column DFTs
  cd* SP;                // [rows][NR][N2] spectra of this group's frames, rows in generator order
  const cd* bhat;        // 3-D spectrum (prime-factor positions) of w^(g^s) with w = exp(-2 pi i u2 / N2), scaled by 1 / L
  const cd* r1;          // exp(-2 pi i q / N1)
  const int* qidx;       // [N2]: position pos(-log_g e) of input column e (entry 0: L)
  const int2* rowtab;    // per row of Y: (u1 row mod N1, -)
  int N1, N2, NR, G, rows;
  float inv;             // 1 / N1
};

// grid = G * NR workgroups of 256 lanes: rows k1 (tile 0) and N1 - k1 (tile 1) of one packed transform.
template <int R1, int R2, int R3>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3))) void k_pfa_fwd_rows_rader(PfaFwdRowsArgs a) {
  constexpr int L = R1 * R2 * R3;
  __shared__ cd data[2 * L];
  __shared__ cd total[2];              // sum of the tile-0 / tile-1 inputs behind column 0 (from the convolution's spectrum)
  __shared__ cd dc[2];                 // input column 0 of both tiles
  const int tid = threadIdx.x;
  const int g = blockIdx.x % a.G, k1 = blockIdx.x / a.G;
  const int N1 = a.N1, N2 = a.N2;
  const unsigned n1 = unsigned(N1);
  const PlainTile tile{data, L};
  const int kr = k1 ? N1 - k1 : 0;
  const cd* y0 = a.Y + (size_t(g) * N1 + k1) * N2;
  const cd* y1 = a.Y + (size_t(g) * N1 + kr) * N2;
  const auto* rt0 = reinterpret_cast<const __attribute__((address_space(4))) int*>(reinterpret_cast<uintptr_t>(a.rowtab)) + 2 * k1;
  const auto* rt1 = reinterpret_cast<const __attribute__((address_space(4))) int*>(reinterpret_cast<uintptr_t>(a.rowtab)) + 2 * kr;
  const unsigned uk0 = unsigned(rt0[0]), uk1 = unsigned(rt1[0]);   // u1 k1 mod N1 of the two rows
  const auto mod_n1 = [&](unsigned x) {                       // x < 2^24: the float quotient is off by at most one
    unsigned r = x - __umul24(unsigned(float(x) * a.inv), n1);
    r = min(r, r + n1);
    return min(r, r - n1);
  };
  // ---- prologue: four input columns per lane (coalesced), column twiddle, scatter to the generator positions
  cd v0[4], v1[4];
  int qi[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int e = tid + 256 * u;
    const int ee = e < N2 ? e : N2 - 1;
    v0[u] = y0[ee];
    v1[u] = y1[ee];
    qi[u] = a.qidx[ee];
  }
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int e = tid + 256 * u;
    const cd x = cmul(v0[u], a.r1[mod_n1(__umul24(uk0, unsigned(e)))]);      // e^{-2 pi i u1 k1 m2 / N1}
    const cd z = cmul(v1[u], a.r1[mod_n1(__umul24(uk1, unsigned(e)))]);
    if (e < N2) {
      if (e == 0) { dc[0] = x; dc[1] = z; }
      else { data[qi[u]] = x; data[L + qi[u]] = z; }
    }
  }
  __syncthreads();
  // ---- the cyclic convolution: first stage along axis R1 (in place), then the shared stages
  using AX = Axes<R1, R2, R3>;
  {
    const int t1 = tid / (L / R1), i1 = tid % (L / R1);
    cd v[R1];
    if (tid < 2 * (L / R1)) {
      axis_load<R1>(tile, t1, AX::base1(i1), AX::kStride1, v);
      dft_sym<R1, false>(v);
      axis_store<R1>(tile, t1, AX::base1(i1), AX::kStride1, v);
    }
    __syncthreads();
  }
  rader_convolve<R1, R2, R3>(tile, a.bhat, total, tid);
  // ---- epilogue: output position p = pos(q) holds bin k2 = g^-q: Z[k1, k2] = x[0] + C0[pos(-q)] and
  //      Z[N1 - k1, N2 - k2] = x'[0] + C1[pos(-q + L/2)] (-1 = g^(L/2)).  pos() is a ring isomorphism, so both are
  //      coordinate-wise: negate the residues, add L/2 mod (R2, R3, R1).  Separate the two frames, store coalesced.
  const cd x0 = dc[0], z0 = dc[1];
  const cd sum0 = total[0] + x0, sum1 = total[1] + z0;
  const bool second = 2 * g + 1 < a.rows;
  const size_t mic = size_t(a.NR) * N2;
  cd* Sa = a.SP + size_t(2 * g) * mic + size_t(k1) * N2;
  cd* Sb = Sa + mic;
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int p = tid + 256 * u;
    if (p < N2) {
      cd Z0, Z1;
      if (p < L) {
        const int i2 = p % R2, i3 = (p / R2) % R3, i1 = p / (R2 * R3);
        const int n2 = i2 ? R2 - i2 : 0, n3 = i3 ? R3 - i3 : 0, n1 = i1 ? R1 - i1 : 0;        // -q
        constexpr int h2 = (L / 2) % R2, h3 = (L / 2) % R3, h1 = (L / 2) % R1;
        const int m2 = n2 + h2 < R2 ? n2 + h2 : n2 + h2 - R2, m3 = n3 + h3 < R3 ? n3 + h3 : n3 + h3 - R3,
                  m1 = n1 + h1 < R1 ? n1 + h1 : n1 + h1 - R1;                                     // -q + L/2
        Z0 = x0 + data[n2 + R2 * n3 + R2 * R3 * n1];
        Z1 = z0 + data[L + m2 + R2 * m3 + R2 * R3 * m1];
      } else {                                                // bin 0: the plain sums
        Z0 = sum0;
        Z1 = sum1;
      }
      Sa[p] = mk(0.5 * (Z0.x + Z1.x), 0.5 * (Z0.y - Z1.y));   // (Z + conj Z') / 2
      if (second) Sb[p] = mk(0.5 * (Z0.y + Z1.y), 0.5 * (Z1.x - Z0.x));   // (Z - conj Z') / 2i
    }
  }
}

}  // namespace pal
