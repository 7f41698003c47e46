// pfa_kernels.h - device side of the prime-factor route (see pfa.hip for the algorithm); a header so that
// tools/microbench_pfa.hip can time the same kernels with ablated functors.
#pragma once
#include "conv_kernels.h"
#include "pfa_sample.h"   // kPfaTC, kPfaUnr

namespace pal {


// ------------------------------------------------------------------ functors of the chirp-spectrum setup kernel
template <int LM> struct PfaChirpIn {     // chirp kernel of the convolution: h[d mod M] = conj(b[|d|]), |d| < N2
  static constexpr bool kLds = false;
  const cd* b;
  int N2;
  __device__ cd operator()(int, int e) const {
    constexpr int M = 1 << LM;
    if (e < N2) return cconj(b[e]);
    if (M - e < N2) return cconj(b[M - e]);
    return mk(0, 0);
  }
};

template <int LM> struct PfaScaledOut {
  static constexpr bool kLds = false;
  cd* out;
  double scale;
  __device__ void operator()(int t, int e, cd v) const { if (t == 0) out[e] = cscale(v, scale); }
};

template <int LM> struct PfaLds {         // LDS sizes of the row pass
  static constexpr bool kCompact = LM >= 11;
  static constexpr int kM = 1 << LM, kLanes = 2 * kM / 16;
  // up to 2048-point tiles the last stage's twiddles live in registers (read once from the full table in global
  // memory, used by the seam and by the last inverse stage): the LDS table then ends before that stage
  static constexpr bool kRegTw = LM <= 11;
  static constexpr int kTwAll = kCompact ? stage_twc_size(LM) : stage_tw_size(LM);     // every stage (setup kernel)
  static constexpr int kTw = kRegTw ? stage_tw_offset(LM, stage_tw_last(LM)) : kTwAll;   // row pass
  static constexpr int kTwPad = (kTw + kLanes - 1) / kLanes * kLanes;   // whole rounds of the workgroup: an unguarded copy
};

template <int LM>
__global__ __launch_bounds__(PfaLds<LM>::kLanes) void k_pfa_hhat(const cd* __restrict__ b, int N2, cd* __restrict__ hhat,
                                                                 double scale, const cd* __restrict__ tws) {
  using L = PfaLds<LM>;
  __shared__ cd data[2 * L::kM];
  __shared__ cd tw[L::kTwAll];
  const int tid = threadIdx.x;
  for (int i = tid; i < L::kTwAll; i += L::kLanes) tw[i] = tws[i];
  wg_fft<LM, false, false, 2, L::kCompact>(data, tw, tid, PfaChirpIn<LM>{b, N2}, PfaScaledOut<LM>{hhat, scale});
}

struct PfaRowsArgs {
  const cd* SP;        // permuted spectra [mic][NR][N2]
  const int4* quad;    // mic rows (a, b) of pair p and (c, d) of pair q; c < 0: no second pair
  cd* Y;               // [G][N1][N2]
  const cd *b, *hhat, *r1, *tws;
  const cd* twfull;    // full (non-compact) stage table of the tile length: source of the register twiddles
  const int2* rowtab;  // per row of Y: (u1 row mod N1, u1 row P mod N1) with P = points per last-stage butterfly group
  int N1, N2, NR, G, u1;
  float inv;
  unsigned long long* stamps;   // diagnostics only (tools/microbench_pfa): 100 MHz clock reads of lane 0 per phase
  int xcd;             // 1: XCD-aware order of the workgroups (row_work_item)
};

// v_permlane32_swap(x, y): x' = {x[0:31], y[0:31]}, y' = {x[32:63], y[32:63]} (lane ranges of the two results).
// swap_pair(a, b, lo, hi):  lo = lanes 0-31 keep a, lanes 32-63 receive b of their partner lane (lane - 32);
//                           hi = lanes 0-31 receive a of their partner lane (lane + 32), lanes 32-63 keep b
__device__ __forceinline__ void swap_pair(double a, double b, double& lo, double& hi) {
  const auto l = __builtin_amdgcn_permlane32_swap(unsigned(__double2loint(a)), unsigned(__double2loint(b)), false, false);
  const auto h = __builtin_amdgcn_permlane32_swap(unsigned(__double2hiint(a)), unsigned(__double2hiint(b)), false, false);
  lo = __hiloint2double(int(h[0]), int(l[0]));
  hi = __hiloint2double(int(h[1]), int(l[1]));
}

// grid = G * NR workgroups, transform fastest so that neighbours share the tables.
//
// Stage plan of one workgroup (two tiles of M points, 16 points per lane):
//   1. first forward stage, hand-mapped: the two halves of a wavefront own the same butterfly index i of
//      tile 0 and tile 1.  Both tiles are built from the SAME whitened bins (tile 1 is the conjugate
//      combination), so each half whitens four of the eight non-zero inputs and the halves trade the other
//      tile's values with v_permlane32_swap - no LDS, no barrier, half the loads and reciprocal square roots.
//   2. the LDS-to-LDS middle stages of the forward transform.
//   3. fused seam: the last forward stage, the product with the chirp spectrum and the first inverse stage
//      run on the same sixteen registers (the last stage's outputs k + P r are exactly the inputs
//      i + (M/16) r' of a first-stage butterfly), which saves one LDS round trip and two barriers.
//   4. the remaining inverse stages; the last one hands Y to global memory.
template <int LM, bool TWG = false>   // TWG: twiddles straight from global memory (L1-resident table) instead of an LDS copy
__global__ __launch_bounds__(PfaLds<LM>::kLanes) void k_pfa_rows(PfaRowsArgs a) {
  using L = PfaLds<LM>;
  constexpr int M = L::kM, NB = M / 16;
  constexpr bool CT = L::kCompact;
  static_assert(stage_log2r(LM, 0) == 4, "the hand-mapped stages assume a radix-16 first stage");
  __shared__ cd data[2 * M];
  __shared__ cd tw_lds[TWG ? 1 : L::kTwPad];
  const cd* const tw = TWG ? a.tws : tw_lds;
  const int tid = threadIdx.x;
  int g, k1;
  if (!row_work_item(blockIdx.x, a.G, a.NR, a.xcd, g, k1)) return;
  unsigned long long* const stamps = a.stamps;
  int stamp_at = 0;
  auto stamp = [&]() {
    if (stamps && tid == 0) stamps[size_t(blockIdx.x) * 8 + stamp_at] = __builtin_amdgcn_s_memrealtime();
    ++stamp_at;
  };
  stamp();
  // the twiddle rows travel global -> registers -> LDS.  Their loads are issued first (the table is allocated to
  // 2^LM entries, so whole rounds need no guard: a guarded copy made the compiler wait for ALL outstanding loads
  // four times) and land while the pair table and the spectrum rows are being fetched.
  constexpr int kTwPer = TWG ? 0 : L::kTwPad / L::kLanes;
  cd twr[kTwPer + 1];
#pragma unroll
  for (int q = 0; q < kTwPer; ++q) twr[q] = a.tws[tid + q * L::kLanes];
  const LdsTile<LM, false, 2> tile{data};

  // ---- 1. first forward stage
  {
    // (constant address space: a uniform entry of a table nobody writes -> scalar load, not a vector load + wait)
    const auto* qp = reinterpret_cast<const __attribute__((address_space(4))) int*>(reinterpret_cast<uintptr_t>(a.quad)) + 4 * g;
    const int4 q = make_int4(qp[0], qp[1], qp[2], qp[3]);
    const size_t mic = size_t(a.NR) * a.N2, off = size_t(k1) * a.N2;
    const bool second = q.z >= 0;
    const cd* sa = a.SP + size_t(q.x) * mic + off;
    const cd* sb = a.SP + size_t(q.y) * mic + off;
    const cd* sc = second ? a.SP + size_t(q.z) * mic + off : sa;   // branch-free loads: a missing second pair re-reads
    const cd* sd = second ? a.SP + size_t(q.w) * mic + off : sb;   // the first one (cache hits) and is zeroed below
    const double keep2 = second ? 1.0 : 0.0;
    const bool upper = (tid >> 5) & 1;                        // upper half of the wavefront = tile 1
    const int i = (tid & 31) | ((tid >> 6) << 5);             // butterfly index, < NB
    cd va[4], vb[4], vc[4], vd[4], ch[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {                             // all twenty loads in flight before the first use
      const int e = i + NB * (u + (upper ? 4 : 0));           // < M/2; N2 <= M/2, the rest is zero padding
      const int ee = e < a.N2 ? e : a.N2 - 1;
      va[u] = sa[ee]; vb[u] = sb[ee]; vc[u] = sc[ee]; vd[u] = sd[ee];
      ch[u] = a.b[ee];
    }
#pragma unroll
    for (int q = 0; q < kTwPer; ++q) tw_lds[tid + q * L::kLanes] = twr[q];
    // Lower lanes (tile 0) whiten inputs r = 0..3 of butterfly i, upper lanes (tile 1) r = 4..7.  Every lane forms both
    // tiles' values of its bin,  x = R^p + i R^q (tile 0)  and  z = conj(R^p) + i conj(R^q) (tile 1),  and ONE half-wave
    // swap per register sorts them: r = u of both tiles is {x of the lower lanes, z of the lower lanes moved up}, r = 4 + u
    // is {x of the upper lanes moved down, z of the upper lanes}.
    cd v[16];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int e = i + NB * (u + (upper ? 4 : 0));
      const cd r1 = whiten(va[u], vb[u]);
      const cd r2 = cscale(whiten(vc[u], vd[u]), keep2);
      const cd c0 = e < a.N2 ? ch[u] : mk(0, 0);             // zero padding beyond N2
      const cd x = cmul(mk(r1.x - r2.y, r1.y + r2.x), c0), z = cmul(mk(r1.x + r2.y, r2.x - r1.y), c0);
      swap_pair(x.x, z.x, v[u].x, v[4 + u].x);
      swap_pair(x.y, z.y, v[u].y, v[4 + u].y);
      v[8 + u] = v[12 + u] = mk(0, 0);
    }
    dft16<false>(v);
    stage_store<LM, false, 0, 2>(tile, (upper ? NB : 0) + i, v);
    __syncthreads();                                          // also publishes `tw`
  }
  stamp();

  // the sixteen chirp-spectrum values of this lane's seam butterfly: requested now, used after the middle stages
  // (the registers are free: LDS, not the register file, limits this kernel to two wavefronts per SIMD)
  cd hh[16];
  constexpr int kFt = L::kRegTw ? 16 / stage_radix(LM, stage_tw_last(LM)) : 1, kFr = L::kRegTw ? stage_radix(LM, stage_tw_last(LM)) : 1;
  cd ftw[kFt][kFr];                                           // last-stage twiddles of this lane's butterflies k = i + NB q
  {
    const int i = tid % NB;
#pragma unroll
    for (int r = 0; r < 16; ++r) hh[r] = a.hhat[i + NB * r];
    if constexpr (L::kRegTw) {
      constexpr int lpl = stage_tw_last(LM), pl = 1 << lpl;
      const cd* src = a.twfull + stage_tw_offset(LM, lpl);
#pragma unroll
      for (int q = 0; q < kFt; ++q)
#pragma unroll
        for (int r = 1; r < kFr; ++r) ftw[q][r] = src[(r - 1) * pl + i + NB * q];
    }
    asm volatile("" ::: "memory");                            // keep the loads here (the scheduler would sink them to their use)
  }

  // ---- 2. forward middle stages
  wg_fft_middle<LM, false, false, 4, 2, CT>(data, tw, tid);
  stamp();

  // ---- 3. last forward stage x chirp spectrum x first inverse stage
  constexpr int LPL = stage_tw_last(LM), RL = stage_radix(LM, LPL), P = 1 << LPL, SPLIT = 16 / RL, HR = RL / 2;
  static_assert(RL == 8 || RL == 16, "last-stage radix");
  // This lane's butterflies of the last stage, in the seam AND in the last inverse stage: tile ts, k = is + NB q.
  const int ts = tid / NB, is = tid % NB;
  {
    cd u[16];
#pragma unroll
    for (int q = 0; q < SPLIT; ++q) {
      cd v[RL];
      const int k = is + NB * q;                              // < P: outputs k + P r
      if constexpr (L::kRegTw) stage_load_with<LM, false, false, LPL, 2>(tile, ts * P + k, v, ftw[q]);
      else stage_load<LM, false, false, LPL, 2, CT>(tile, tw, ts * P + k, v);
#pragma unroll
      for (int r = 0; r < RL; ++r) u[q + SPLIT * r] = cmul(v[r], hh[q + SPLIT * r]);   // element k + P r = is + NB (q + SPLIT r)
    }
    __syncthreads();                                          // every lane has read its inputs
    dft16<true>(u);
    stage_store<LM, false, 0, 2>(tile, tid, u);
    __syncthreads();
  }
  stamp();

  // ---- 4. remaining inverse stages; the last one is hand-mapped: only outputs e < M/2 can be below N2, each is
  //         scaled by the chirp b[e] and by the column twiddle exp(2 pi i u1 row m2 / N1) and stored at Y[row][m2]
  //         (tile 0: row k1, m2 = e; tile 1: row N1 - k1 of the reversed transform, m2 = -e mod N2).  The twiddle
  //         index u1 row m2 mod N1 advances by a wave-uniform step from one output of a butterfly to the next.
  {
    cd* const Yg = a.Y + size_t(g) * a.N1 * a.N2;
    const int N1 = a.N1, N2 = a.N2;
    // tile: wave-uniform where NB lanes are whole wavefronts; at 512 points a wavefront holds both tiles (per-lane values,
    // vector loads of the row table)
    const int t = NB >= 64 ? __builtin_amdgcn_readfirstlane(ts) : ts;
    const int row = t ? (k1 ? N1 - k1 : 0) : k1;
    // chirp and column-twiddle factors of this lane's outputs: requested before the inverse middle stages, which touch
    // only LDS, so that they have landed when the last stage needs them
    cd fb[SPLIT][HR], fr[SPLIT][HR];
    int m2s[SPLIT][HR];
    // scalar load (constant address space): u1 row mod N1 and the index step between outputs
    const auto* rt = reinterpret_cast<const __attribute__((address_space(4))) int*>(reinterpret_cast<uintptr_t>(a.rowtab)) + 2 * row;
    const unsigned uk = unsigned(rt[0]), step = unsigned(rt[1]), n1 = unsigned(N1);
#pragma unroll
    for (int q = 0; q < SPLIT; ++q) {
      const int k = is + NB * q;
      const int mb = t ? N2 - k : k;                                    // m2 of output r is mb -/+ P r (tile 1: except e = 0)
      const unsigned x = __umul24(uk, unsigned(mb));                    // < 2^24: exact in float
      unsigned idx = x - __umul24(unsigned(float(x) * a.inv), n1);      // x mod N1, off by at most one N1 either way
      idx = min(idx, idx + n1);                                         // (unsigned wrap-around picks the in-range value)
      idx = min(idx, idx - n1);
#pragma unroll
      for (int r = 0; r < HR; ++r) {
        const int e = k + P * r;
        const int ee = e < N2 ? e : N2 - 1;
        int m2 = t ? mb - P * r : e;
        unsigned ti = idx;
        if (t && e == 0) { m2 = 0; ti = 0; }
        m2s[q][r] = m2;
        fb[q][r] = a.b[ee];
        fr[q][r] = a.r1[ti];                                            // r1 holds exp(-2 pi i q / N1)
        if (t) { idx -= step; idx = min(idx, idx + n1); }
        else { idx += step; idx = min(idx, idx - n1); }
      }
    }
    asm volatile("" ::: "memory");                                      // keep these loads above the LDS-only stages
    wg_fft_middle<LM, false, true, 4, 2, CT>(data, tw, tid);
    if (!(t == 1 && k1 == 0)) {                                         // row 0 pairs with itself: tile 1 is a duplicate
      cd* const Yrow = Yg + size_t(row) * N2;
#pragma unroll
      for (int q = 0; q < SPLIT; ++q) {
        const int k = is + NB * q;
        cd v[RL];
        if constexpr (L::kRegTw) stage_load_with<LM, false, true, LPL, 2>(tile, ts * P + k, v, ftw[q]);
        else stage_load<LM, false, true, LPL, 2, CT>(tile, tw, ts * P + k, v);
#pragma unroll
        for (int r = 0; r < HR; ++r) {
          const int e = k + P * r;
          const cd z = cmulc(cmul(v[r], fb[q][r]), fr[q][r]);
          if (e < N2) Yrow[m2s[q][r]] = z;
        }
      }
    }
  }
  stamp();
}

// ------------------------------------------------------------------ column pass
// One lane per column m2 (64 consecutive columns per wavefront: coalesced 1 KB loads of Y, 512 B stores of a
// correlation row).  The four wavefronts of a workgroup take four chunks of kPfaTC output indices t; every
// wavefront produces both pairs of the packed transform (pair p = real parts, pair q = imaginary parts), so a
// cos / sin value fetched through the scalar cache feeds two FMAs.  With E_j = Y_j + Y_{N1-j}, O_j = Y_j - Y_{N1-j}:
//   p:  c[t] = Re Y_0 + sum_j cos(j t) Re E_j - sum_j sin(j t) Im O_j,   c[N1-t] = the same with + sin
//   q:  c[t] = Im Y_0 + sum_j cos(j t) Im E_j + sum_j sin(j t) Re O_j,   c[N1-t] = the same with - sin
// T[(j-1)][chunk][cos | sin][tt] is wave-uniform: scalar loads, SGPR operands of v_fmac_f64.  The pass is bound by
// memory latency, not arithmetic: kPfaUnr steps share one batch of loads, issued one batch ahead of its FMAs.
// the accumulation of one lane's column: cx/sy (pair p) and cy/sx (pair q) for the TC indices of chunk `ch`, the
// plain sums for t = 0.  Steps beyond h meet zero rows of the table; the next batch of loads is issued ahead.
template <int TC, int UNR>
__device__ __forceinline__ void pfa_cols_accumulate(const cd* __restrict__ Yg, int N1, int N2, int nch, int ch, const double* __restrict__ T,
                                                    cd& y0, double* cx, double* sy, double* cy, double* sx, double& sumx, double& sumy) {
  const int h = (N1 - 1) / 2;
#pragma unroll
  for (int tt = 0; tt < TC; ++tt) cx[tt] = sy[tt] = cy[tt] = sx[tt] = 0.0;
  sumx = sumy = 0.0;
  // (constant address space: wave-uniform entries of a table nobody writes -> scalar loads.  A kernel argument marked __restrict__
  //  gets them anyway; a pointer that arrives inside a struct - pfa_cols_fin.h's FinSrc - did not, and the dense finishing pass
  //  ran on vector loads with a wait behind each: 640 us against 270)
  const auto* Tj = reinterpret_cast<const __attribute__((address_space(4))) double*>(reinterpret_cast<uintptr_t>(T)) + size_t(ch) * 2 * TC;
  const size_t tstep = size_t(nch) * 2 * TC;
  y0 = Yg[0];
  cd yj[UNR], ym[UNR];
#pragma unroll
  for (int u = 0; u < UNR; ++u) {
    const int j = 1 + u <= h ? 1 + u : h;                     // (h = 0: row 0 twice, multiplied by zero table rows)
    yj[u] = Yg[size_t(j) * N2];
    ym[u] = Yg[size_t(h > 0 ? N1 - j : 0) * N2];
  }
  for (int j = 1; j <= h; j += UNR) {
    cd nj[UNR], nm[UNR];
#pragma unroll
    for (int u = 0; u < UNR; ++u) {                           // next batch; past the end the last rows are re-read: no branch
      const int jn = j + UNR + u <= h ? j + UNR + u : h;
      nj[u] = Yg[size_t(jn) * N2];
      nm[u] = Yg[size_t(N1 - jn) * N2];
    }
#pragma unroll
    for (int u = 0; u < UNR; ++u, Tj += tstep) {              // steps beyond h meet zero rows of the table
      const bool in = j + u <= h;
      const double ap = yj[u].x + ym[u].x, bp = yj[u].y - ym[u].y, aq = yj[u].y + ym[u].y, bq = yj[u].x - ym[u].x;
      sumx += in ? ap : 0.0;
      sumy += in ? aq : 0.0;
#pragma unroll
      for (int tt = 0; tt < TC; ++tt) {
        const double c = Tj[tt], sn = Tj[TC + tt];
        cx[tt] = __builtin_fma(c, ap, cx[tt]);
        sy[tt] = __builtin_fma(sn, bp, sy[tt]);
        cy[tt] = __builtin_fma(c, aq, cy[tt]);
        sx[tt] = __builtin_fma(sn, bq, sx[tt]);
      }
    }
#pragma unroll
    for (int u = 0; u < UNR; ++u) { yj[u] = nj[u]; ym[u] = nm[u]; }
  }
}

// one 64-column block of one transform, one chunk of output indices per wavefront
template <int TC, int UNR>
__device__ __forceinline__ void pfa_cols_block(const cd* __restrict__ Y, double* __restrict__ corr, size_t stride, int N1, int N2, int nch,
                                               const double* __restrict__ T, const int* __restrict__ zero_rows, int g, int cb, int ch) {
  const int lane = threadIdx.x & 63;
  const int m2 = cb * 64 + lane;
  const bool live = m2 < N2;
  const cd* Yg = Y + size_t(g) * N1 * N2 + (live ? m2 : N2 - 1);
  const int h = (N1 - 1) / 2;
  double cx[TC], sy[TC], cy[TC], sx[TC];
  double sumx, sumy;
  cd y0;
  pfa_cols_accumulate<TC, UNR>(Yg, N1, N2, nch, ch, T, y0, cx, sy, cy, sx, sumx, sumy);
  if (!live) return;
  double* outp = corr + size_t(2 * g) * stride + m2;
  double* outq = outp + stride;
  // a pair with a silent microphone has R = 0 in every bin: its row is exactly zero in the reference, while the packed
  // transform leaves the other pair's rounding noise (1e-17) in it
  const double kp = zero_rows && zero_rows[2 * g] ? 0.0 : 1.0, kq = zero_rows && zero_rows[2 * g + 1] ? 0.0 : 1.0;
  if (kp == 0.0) { y0.x = sumx = 0.0; }
  if (kq == 0.0) { y0.y = sumy = 0.0; }
#pragma unroll
  for (int tt = 0; tt < TC; ++tt) {
    cx[tt] *= kp; sy[tt] *= kp; cy[tt] *= kq; sx[tt] *= kq;
  }
  if (ch == 0) { outp[0] = y0.x + sumx; outq[0] = y0.y + sumy; }
#pragma unroll
  for (int tt = 0; tt < TC; ++tt) {
    const int t = ch * TC + tt + 1;
    if (t <= h) {
      outp[size_t(N2) * t] = y0.x + cx[tt] - sy[tt];
      outp[size_t(N2) * (N1 - t)] = y0.x + cx[tt] + sy[tt];
      outq[size_t(N2) * t] = y0.y + cy[tt] + sx[tt];
      outq[size_t(N2) * (N1 - t)] = y0.y + cy[tt] - sx[tt];
    }
  }
}

template <int TC, int UNR>
__global__ __launch_bounds__(256) void k_pfa_cols(const cd* __restrict__ Y, double* __restrict__ corr, size_t stride,
                                                  int N1, int N2, int G, int nch, const double* __restrict__ T,
                                                  const int* __restrict__ zero_rows) {
  const int ch = int(blockIdx.y) * 4 + __builtin_amdgcn_readfirstlane(int(threadIdx.x) >> 6);
  if (ch >= nch) return;
  pfa_cols_block<TC, UNR>(Y, corr, stride, N1, N2, nch, T, zero_rows, int(blockIdx.x % G), int(blockIdx.x / G), ch);
}


}  // namespace pal
