// pfa_rader.h - row pass of the prime-factor route by Rader's algorithm (gfx950, fp64).
//
// k_pfa_rows (pfa_kernels.h) computes the N2-point DFTs as chirp convolutions of 2^lm >= 2 N2 - 1 points.  When N2
// is prime, Rader's re-indexing with a primitive root g of N2,
//     X[g^r] = x[0] + sum_q x[g^-q] w^(g^(r-q)),      X[0] = sum_k x[k],
// makes the same DFT a CYCLIC convolution of length L = N2 - 1, and for N2 = 991 that length is 990 = 11 x 9 x 10 with
// pairwise coprime factors: in the residues (s mod 9, s mod 10, s mod 11) the convolution is three-dimensional, and the
// transforms around the pointwise product are plain 9-, 10- and 11-point DFTs along one axis each - in place, no
// twiddle factors, no chirp multiplications (mixed_radix.h).  35 KB of LDS per workgroup instead of 68.  The rest is
// the row pass of pfa.hip unchanged: one workgroup per row pair (k1, N1 - k1) of one packed transform, whitened pair
// spectra built on the fly, tile 1 = the reversed row, column twiddle on the way out, Y[row][m2] to global memory.
//
//   forward    the rows of SP are stored in generator order at their prime-factor positions (x[g^-s] at pos(s), bin 0
//              last: the forward transform writes them that way), so the stage along the 11-axis reads its inputs
//              coalesced from global memory: the two halves of a wavefront own butterfly i of tile 0 and of tile 1,
//              each half whitens six of the eleven (shared) inputs and the halves trade with v_permlane32_swap;
//              9-axis (LDS), then the seam along the 10-axis: DFT, x the kernel's 3-D spectrum / (L n), inverse DFT
//   inverse    9-axis, 11-axis; C[s] lands at pos(s)
//   epilogue   four bins per lane: X[e] = x[0] + C[log_g e] through a position table, column twiddle, coalesced stores
#pragma once
#include "conv_kernels.h"
#include "mixed_radix.h"
#include "pfa_kernels.h"   // swap_pair

namespace pal {

struct PfaRaderArgs {
  const cd* SP;          // permuted spectra [mic][NR][N2], rows in generator order: x[g^-s] at pos(s), x[0] at L
  const int4* quad;      // mic rows (a, b) of pair p and (c, d) of pair q; c < 0: no second pair
  cd* Y;                 // [G][N1][N2]
  const cd* bhat;        // 3-D spectrum of w^(g^s) in prime-factor positions, scaled by 1 / (L n)
  const cd* r1;          // exp(-2 pi i q / N1)
  const int* ridx;       // [N2]: position of log_g e (entry 0 unused)
  const int2* rowtab;    // per row of Y: (u1 row mod N1, -)
  int N1, N2, NR, G;
  float inv;             // 1 / N1
  double scale;          // 1 / n: the x[0] and sum terms bypass the scaled convolution
  unsigned long long* stamps;   // diagnostics only (tools/microbench_pfa): 100 MHz clock reads of lane 0 per phase
  int xcd;               // 1: XCD-aware order of the workgroups (row_work_item, pfa_kernels.h)
};

// Stages 2-5 of the cyclic convolution on two L-point tiles in LDS (the first stage, along axis R1, has filled them):
// in prime-factor coordinates (mixed_radix.h) the convolution is three-dimensional and its stages are plain DFTs along
// one axis each, in place and without twiddles: forward along R2, the seam along R3 (forward DFT, x the kernel's 3-D
// spectrum `bhat`, inverse DFT on the same registers), inverse along R2 and R1.  One barrier per stage; ends with one.
// `total[tile]` (LDS) receives the sum of the tile's inputs, which the spectrum holds at its origin.
template <int R1, int R2, int R3>
__device__ __forceinline__ void rader_convolve(const PlainTile& tile, const cd* __restrict__ bhat, cd* total, int tid) {
  using AX = Axes<R1, R2, R3>;
  constexpr int L = AX::L, HALF = 128;
  static_assert(L / R1 <= HALF && L / R2 <= HALF && L / R3 <= HALF, "one butterfly per lane and stage");
  const int t = tid >> 7, i = tid & (HALF - 1);               // stage work item: tile, butterfly
  {
    cd v[R2];
    if (i < L / R2) {
      axis_load<R2>(tile, t, AX::base2(i), AX::kStride2, v);
      dft_sym<R2, false>(v);
      axis_store<R2>(tile, t, AX::base2(i), AX::kStride2, v);
    }
    __syncthreads();
  }
  {
    cd v[R3];
    if (i < L / R3) {
      const int base = AX::base3(i);
      axis_load<R3>(tile, t, base, AX::kStride3, v);
      dft_sym<R3, false>(v);
      if (i == 0) total[t] = v[0];                            // the 3-D spectrum at the origin = the sum of the tile's inputs
#pragma unroll
      for (int r = 0; r < R3; ++r) v[r] = cmul(v[r], bhat[base + r * AX::kStride3]);
      dft_sym<R3, true>(v);
      axis_store<R3>(tile, t, base, AX::kStride3, v);
    }
    __syncthreads();
  }
  {
    cd v[R2];
    if (i < L / R2) {
      axis_load<R2>(tile, t, AX::base2(i), AX::kStride2, v);
      dft_sym<R2, true>(v);
      axis_store<R2>(tile, t, AX::base2(i), AX::kStride2, v);
    }
    __syncthreads();
  }
  {
    // (the radix-R1 stages have 2 L / R1 = 180 butterflies: packed onto the first three wavefronts, the fourth only waits)
    const int t1 = tid / (L / R1), i1 = tid % (L / R1);
    cd v[R1];
    if (tid < 2 * (L / R1)) {
      axis_load<R1>(tile, t1, AX::base1(i1), AX::kStride1, v);
      dft_sym<R1, true>(v);
      axis_store<R1>(tile, t1, AX::base1(i1), AX::kStride1, v);
    }
    __syncthreads();
  }
}

template <int R1, int R2, int R3>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3))) void k_pfa_rows_rader(PfaRaderArgs a) {
  constexpr int L = R1 * R2 * R3, HALF = 128;
  static_assert(L / R1 <= HALF && L / R2 <= HALF && L / R3 <= HALF, "one butterfly per lane and stage");
  __shared__ cd data[2 * L];
  __shared__ cd total[2];              // sum of the tile-0 / tile-1 inputs (from the convolution's spectrum)
  __shared__ cd dc[2];                 // x[0] of both tiles
  const int tid = threadIdx.x;
  int g, k1;
  if (!row_work_item(blockIdx.x, a.G, a.NR, a.xcd, g, k1)) return;
  const int N1 = a.N1, N2 = a.N2;
  const PlainTile tile{data, L};
  unsigned long long* const stamps = a.stamps;
  int stamp_at = 0;
  auto stamp = [&]() {
    if (stamps && tid == 0) stamps[size_t(blockIdx.x) * 8 + stamp_at] = __builtin_amdgcn_s_memrealtime();
    ++stamp_at;
  };
  stamp();

  // ---- the first forward stage straight from global memory
  const auto* qp = reinterpret_cast<const __attribute__((address_space(4))) int*>(reinterpret_cast<uintptr_t>(a.quad)) + 4 * g;
  const int4 q = make_int4(qp[0], qp[1], qp[2], qp[3]);
  const size_t mic = size_t(a.NR) * N2, off = size_t(k1) * N2;
  const bool second = q.z >= 0;
  const cd* sa = a.SP + size_t(q.x) * mic + off;
  const cd* sb = a.SP + size_t(q.y) * mic + off;
  const cd* sc = second ? a.SP + size_t(q.z) * mic + off : sa;
  const cd* sd = second ? a.SP + size_t(q.w) * mic + off : sb;
  const double keep2 = second ? 1.0 : 0.0;
  int ri[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int e = tid + 256 * u;
    ri[u] = a.ridx[e < N2 ? e : N2 - 1];
  }
  // tile 0: R^p + i R^q at (k1, e);  tile 1: conj(R^p) + i conj(R^q) = the reversed row N1 - k1, held at the SAME
  // positions (it is transformed as the reversed sequence and its outputs are stored reversed)
  constexpr int NB1 = L / R1, HR = (R1 + 1) / 2;              // 90 butterflies per tile, 6 inputs per half-wavefront
  static_assert(NB1 <= 96 && 2 * HR >= R1, "three wavefronts of 32 butterfly pairs");
  if (tid < 192) {                                            // (wave-uniform)
    const int upper = (tid >> 5) & 1, bf = 32 * (tid >> 6) + (tid & 31);
    const int bfc = bf < NB1 ? bf : NB1 - 1;
    cd va[HR], vb[HR], vc[HR], vd[HR];
#pragma unroll
    for (int u = 0; u < HR; ++u) {
      const int j = upper * HR + u;
      const int at = bfc + NB1 * (j < R1 ? j : R1 - 1);
      va[u] = sa[at]; vb[u] = sb[at]; vc[u] = sc[at]; vd[u] = sd[at];
    }
    cd v[2 * HR];
#pragma unroll
    for (int u = 0; u < HR; ++u) {
      const cd r1 = whiten(va[u], vb[u]);
      const cd r2 = cscale(whiten(vc[u], vd[u]), keep2);
      const cd x = mk(r1.x - r2.y, r1.y + r2.x), z = mk(r1.x + r2.y, r2.x - r1.y);
      // lower half: v[u] = own x, v[HR + u] = the upper half's x;  upper half: v[u] = the lower half's z, v[HR + u] = own z
      swap_pair(x.x, z.x, v[u].x, v[HR + u].x);
      swap_pair(x.y, z.y, v[u].y, v[HR + u].y);
    }
    dft_sym<R1, false>(v);
    if (bf < NB1) axis_store<R1>(tile, upper, Axes<R1, R2, R3>::base1(bf), Axes<R1, R2, R3>::kStride1, v);   // in place
  } else if (tid == 192) {                                    // bin 0 (the row's last position) bypasses the convolution
    const cd r1 = whiten(sa[L], sb[L]);
    const cd r2 = cscale(whiten(sc[L], sd[L]), keep2);
    const cd x = mk(r1.x - r2.y, r1.y + r2.x), z = mk(r1.x + r2.y, r2.x - r1.y);
    dc[0] = x; dc[1] = z;
  }
  __syncthreads();
  stamp();

  rader_convolve<R1, R2, R3>(tile, a.bhat, total, tid);
  stamp();
  // ---- epilogue: X[e] = x[0] + C[log_g e] (X[0] = sum of the inputs), column twiddle, store
  const cd x0 = cscale(dc[0], a.scale), z0 = cscale(dc[1], a.scale);
  const cd sum0 = cscale(total[0] + dc[0], a.scale), sum1 = cscale(total[1] + dc[1], a.scale);
  const int kr = k1 ? N1 - k1 : 0;
  const auto* rt0 = reinterpret_cast<const __attribute__((address_space(4))) int*>(reinterpret_cast<uintptr_t>(a.rowtab)) + 2 * k1;
  const auto* rt1 = reinterpret_cast<const __attribute__((address_space(4))) int*>(reinterpret_cast<uintptr_t>(a.rowtab)) + 2 * kr;
  const unsigned uk0 = unsigned(rt0[0]), uk1 = unsigned(rt1[0]), n1 = unsigned(N1);
  cd* const Y0 = a.Y + (size_t(g) * N1 + k1) * N2;
  cd* const Y1 = a.Y + (size_t(g) * N1 + kr) * N2;
  // twiddle indices u1 row m2 mod N1 of this lane's bins e = tid + 256 u: one reduction, then wave-uniform steps
  const auto mod_n1 = [&](unsigned x) {                       // x < 2^24: the float quotient is off by at most one
    unsigned r = x - __umul24(unsigned(float(x) * a.inv), n1);
    r = min(r, r + n1);
    return min(r, r - n1);
  };
  unsigned idx0 = mod_n1(__umul24(uk0, unsigned(tid)));                        // tile 0: m2 = e
  unsigned idx1 = mod_n1(__umul24(uk1, unsigned(tid ? N2 - tid : 0)));         // tile 1: m2 = -e mod N2
  const unsigned st0 = mod_n1(uk0 * 256u), st1 = mod_n1(uk1 * 256u);          // scalar
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int e = tid + 256 * u;
    if (e < N2) {
      const cd X = e ? x0 + data[ri[u]] : sum0;
      const cd Z = e ? z0 + data[L + ri[u]] : sum1;
      Y0[e] = cmulc(X, a.r1[idx0]);                           // r1 holds exp(-2 pi i q / N1)
      if (k1) Y1[e ? N2 - e : 0] = cmulc(Z, a.r1[idx1]);      // row 0 pairs with itself: tile 1 would be a duplicate
    }
    idx0 += st0;
    idx0 = min(idx0, idx0 - n1);
    idx1 = (u == 0 && tid == 0) ? mod_n1(__umul24(uk1, unsigned(N2 - 256))) : idx1 + n1 - st1;   // (e = 0 maps to m2 = 0, not N2)
    idx1 = min(idx1, idx1 - n1);
  }
  stamp();
}

}  // namespace pal
