// pfa_fin_lean.h - the statistics of the finishing column pass (pfa_cols_fin.h) for the Rader-89 column transform without
// histograms: every WAVEFRONT on its own, no block barrier, no LDS, a third of the vector instructions (gfx950, fp64).
//
// Measured on the metric run (profiles/r03_c_fin_ablation.txt): the column transform is 1185 vector instructions per
// wavefront and 77 us per launch group, the statistics of pfa_cols_fin.h added 2510 and 150 us - their ten DPP (value, index)
// reductions per pass, the exec-mask branches of the per-sample index tracking and the block-level merges through LDS.
// The pass is bound by vector-instruction issue at three wavefronts per SIMD, so the remedy is fewer instructions:
//
//   * pass A keeps five running values per row (max, min, sum, sum of squares, sum of magnitudes: one instruction each per
//     sample) and no index; the wavefront's maximum V is ONE max reduction, and the first index of it is found by the scalar
//     unit: ballot(x == V) per slot, lowest slot index t first, lowest lane inside a slot - exact for ties, zero rows included;
//   * the sums and the minimum stay in the lanes until the very end and are reduced once, together with the SNR window sums;
//   * candidates (samples at or above 0.8 V, peaks inside the lag window's margins, plateaus) are rare: ballot first, one
//     readlane when a single lane holds one, a reduction only when several do;
//   * the slots whose output index meets the lag window or the SNR window are a bit mask (one ballot over the lanes that
//     hold the slot table): a slot outside costs a scalar bit test;
//   * every wavefront polls the siblings' maxima itself and publishes its own FinPartial and `done` word
//     (FinArgs.pw = 4 entries per block): the block's wavefronts never wait for each other after the transform.
//
// Semantics are those of pfa_cols_fin.h's general path with per-wavefront instead of per-block search floors (a floor only
// prunes the candidate search; a wavefront whose bounded search finds no strict peak searches all its samples).
#pragma once

namespace pal {

// (max_raw / min_raw, pfa_cols_stats.h: no canonicalising self-maximum in front - the operands are results of arithmetic)
__device__ __forceinline__ double wave_max63r(double v) { return wave_reduce_d(v, -__builtin_huge_val(), [](double a, double b) { return max_raw(a, b); }); }
__device__ __forceinline__ double wave_min63r(double v) { return wave_reduce_d(v, __builtin_huge_val(), [](double a, double b) { return min_raw(a, b); }); }
__device__ __forceinline__ double readlane_d(double v, int l) {
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}

// (value, index) of a few lanes (index < 0: none) -> the best one, uniform.  BETTER as in wave_arg63
template <class BETTER> __device__ __forceinline__ void uniform_arg_sparse(double& v, int& i, BETTER better) {
  const unsigned long long cm = __ballot(i >= 0);
  if (cm == 0) { i = -1; v = 0; return; }
  if ((cm & (cm - 1)) == 0) {
    const int l = int(__builtin_ctzll(cm));
    v = readlane_d(v, l);
    i = __builtin_amdgcn_readlane(i, l);
    return;
  }
  wave_arg63(v, i, better);
  v = wave_bcast63(v);
  i = wave_bcast63(i);
}
// many lanes hold a candidate: one max reduction, then the index by ballot (HIGHEST index among equal values - peak priority)
__device__ __forceinline__ void uniform_arg_dense_hi(double& v, int& i) {
  const double h = wave_bcast63(wave_max63r(i >= 0 ? v : -__builtin_huge_val()));
  const unsigned long long tm = __ballot(i >= 0 && v == h);
  if (tm == 0) { i = -1; v = 0; return; }
  if ((tm & (tm - 1)) == 0) {
    i = __builtin_amdgcn_readlane(i, int(__builtin_ctzll(tm)));
  } else {
    int j = (i >= 0 && v == h) ? i : -1;
    j = wave_reduce_i(j, -1, [](int a, int b) { return a > b ? a : b; });
    i = wave_bcast63(j);
  }
  v = h;
}
// LOWEST index among equal values (np.argmax)
__device__ __forceinline__ void uniform_arg_dense_lo(double& v, int& i) {
  const double h = wave_bcast63(wave_max63r(i >= 0 ? v : -__builtin_huge_val()));
  const unsigned long long tm = __ballot(i >= 0 && v == h);
  if (tm == 0) { i = -1; v = 0; return; }
  if ((tm & (tm - 1)) == 0) {
    i = __builtin_amdgcn_readlane(i, int(__builtin_ctzll(tm)));
  } else {
    int j = (i >= 0 && v == h) ? i : INT_MAX;
    j = wave_reduce_i(j, INT_MAX, [](int a, int b) { return a < b ? a : b; });
    i = wave_bcast63(j);
  }
  v = h;
}
__device__ __forceinline__ double uniform_max_sparse(double v) {   // -inf = none
  if (__ballot(v > -__builtin_huge_val()) == 0) return -__builtin_huge_val();
  return wave_bcast63(wave_max63r(v));
}

// Two quantities in one butterfly: every EVEN lane ends with OP over the wavefront of a, every ODD lane with OP of b.  The lanes
// trade the other quantity with their neighbour first (quad_perm), then fold 2, 4, 8 lanes apart inside the rows of sixteen
// (quad_perm, row_ror) and across them (v_permlane16_swap / v_permlane32_swap: both halves receive the other's value) -
// 26 vector instructions for the pair against 36 for two prefix chains, and the result sits in every lane of its parity.
constexpr int kQuadXor1 = 0xB1, kQuadXor2 = 0x4E, kRor4 = 0x124, kRor8 = 0x128;
template <class OP> __device__ __forceinline__ double wave_pair_reduce(double a, double b, OP op) {
  const bool odd = threadIdx.x & 1;
  double c = odd ? b : a;
  const double d = odd ? a : b;
  c = op(c, dpp_d<kQuadXor1>(0.0, d));
  c = op(c, dpp_d<kQuadXor2>(0.0, c));
  c = op(c, dpp_d<kRor4>(0.0, c));
  c = op(c, dpp_d<kRor8>(0.0, c));
  {
    const unsigned lo = unsigned(__double2loint(c)), hi = unsigned(__double2hiint(c));
    const auto l = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
    const auto h = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    c = op(__hiloint2double(int(h[0]), int(l[0])), __hiloint2double(int(h[1]), int(l[1])));
  }
  {
    const unsigned lo = unsigned(__double2loint(c)), hi = unsigned(__double2hiint(c));
    const auto l = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
    const auto h = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
    c = op(__hiloint2double(int(h[0]), int(l[0])), __hiloint2double(int(h[1]), int(l[1])));
  }
  return c;
}
__device__ __forceinline__ double wave_pair_sum(double a, double b) { return wave_pair_reduce(a, b, [](double x, double y) { return x + y; }); }
__device__ __forceinline__ double wave_pair_min(double a, double b) { return wave_pair_reduce(a, b, [](double x, double y) { return min_raw(x, y); }); }

// One wavefront of a Rader-89 column block: ro[i] = c[t] of slot i + 1 (t = lane i of slot_t), c0 = c[0] (slot 0, wavefront 0
// only); real parts = row 2 g, imaginary parts = row 2 g + 1.  Returns true in the transform's last block (it finishes the rows).
// General form (dense column DFTs, strips): `emask` bit s = slot s exists (slot 0 = c0; ALL: every slot 1 .. 22 exists and slot 0 in
// wavefront 0 only - the Rader-89 case); `edge_block`: the wavefront's columns touch the grid's first or last column; the
// wavefront's entries sit at index (cb, wave) of fa.pw per block.
// PART (the four-step workspace, sample m = N2 t + m2 < n): the row ends inside slot `pslot` - its samples are valid where `pvalid`
// (the others hold zeros, which the sums may see but maxima, minima and the index search must not), later slots do not exist.
template <bool ALL, bool PART, int NS, class STAMP>
__device__ __forceinline__ bool fin_lean_r89(const cd (&ro)[NS], const cd c0, const int slot_t, const unsigned emask, const int wave, const int lane,
                                             const int g, const int cb, const int nblk, const int N1, const int N2, const int rows, const int c_lo,
                                             const int m2, const bool own, const bool inner, const bool edge_block, const int pslot, const bool pvalid,
                                             const PeakArgs& pa, const FinArgs& fa, STAMP&& stamp) {
  static_assert(NS + 1 <= 32, "slot masks are 32 bits");
  const int n = pa.n, S = pa.splits, PW = fa.pw;
  const unsigned long long pvm = PART ? __ballot(pvalid) : ~0ull;
  const unsigned long long ownm = __ballot(own), innerm = __ballot(inner);
  const int nrow = 2 * g + 1 < rows ? 2 : 1;
  const bool windowed = fa.windowed != 0;
  auto T = [&](int slot) { return slot == 0 ? 0 : __builtin_amdgcn_readlane(slot_t, slot - 1); };
  auto for_slots = [&](int r, auto&& fn) {                     // fn(sample, slot): slot is a constant once unrolled
    if (ALL ? wave == 0 : (emask & 1u) != 0) fn(r ? c0.y : c0.x, 0);
#pragma unroll
    for (int i = 0; i < NS; ++i)
      if (ALL || (emask >> (i + 1) & 1u)) fn(r ? ro[i].y : ro[i].x, i + 1);   // (uniform)
  };
  auto slot_mask = [&](int t_lo, int t_hi) -> unsigned {       // slots whose output index lies in [t_lo, t_hi]
    const unsigned long long b = __ballot(lane < NS && slot_t >= t_lo && slot_t <= t_hi);
    const unsigned m = (unsigned(b) << 1) | (t_lo <= 0 && t_hi >= 0 ? 1u : 0u);
    return ALL ? (wave == 0 ? m : m & ~1u) : m & emask;
  };
  // the lag window and its margins of distance - 1 samples (the row's end points are never peaks)
  int lo1 = 1, hi1 = 0;
  unsigned wmask = 0;
  if (windowed) {
    const int elo = fa.win_lo - (pa.dist - 1), ehi = fa.win_hi + (pa.dist - 1);
    lo1 = elo > 1 ? elo : 1;
    hi1 = ehi < n - 2 ? ehi : n - 2;
    if (lo1 <= hi1) wmask = slot_mask(lo1 / N2, hi1 / N2);
  }

  double vn[2], s1[2], s2[2], a1[2];                           // per lane until the end
  FinPartial pt[2];                                            // uniform fields; the sums are filled in at the end
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    vn[r] = INFINITY; s1[r] = s2[r] = a1[r] = 0;
    pt[r].hb = 0; pt[r].plat = -INFINITY; pt[r].hw = 0; pt[r].platw = -INFINITY; pt[r].hm = 0;
    pt[r].mb = pt[r].mw = pt[r].mm = -1; pt[r].pad = 0;
    if (r >= nrow) continue;                                   // (uniform)
    // ---- pass A
    double vm = -INFINITY, vmin = INFINITY, t1 = 0, t2 = 0, ta = 0;
    for_slots(r, [&](double x, int slot) {
      if (PART && slot == pslot) {                             // (uniform) the row's last, partial slot
        vm = pvalid ? max_raw(vm, x) : vm;
        vmin = pvalid ? min_raw(vmin, x) : vmin;
      } else {
        vm = max_raw(vm, x);
        vmin = min_raw(vmin, x);
      }
      t1 += x;
      t2 = __builtin_fma(x, x, t2);
      ta += fabs(x);
    });
    if (!own) { vm = -INFINITY; vmin = INFINITY; t1 = t2 = ta = 0; }
    vn[r] = vmin; s1[r] = t1; s2[r] = t2; a1[r] = ta;
    const double V = wave_bcast63(wave_max63r(vm));
    // first index of V - and, while the one sample that holds it is at hand, whether it is a strict peak: it then is the
    // wavefront's highest peak, and pass B has nothing to search
    int im = INT_MAX, nhit = 0;
    bool vpeak = false;
    for_slots(r, [&](double x, int slot) {
      const unsigned long long mk = __ballot(x == V) & ownm & (PART && slot == pslot ? pvm : ~0ull);
      if (mk) {                                                // (uniform)
        const int l = int(__builtin_ctzll(mk));                // (own lanes are 1 .. 62: both neighbour lanes exist)
        const int cand = N2 * T(slot) + (c_lo - 1) + l;
        im = cand < im ? cand : im;
        nhit += __builtin_popcountll(mk);
        vpeak = (innerm >> l & 1ull) && (!PART || (cand >= 1 && cand <= n - 2)) && readlane_d(x, l - 1) < V && readlane_d(x, l + 1) < V;
      }
    });
    if (im == INT_MAX) im = -1;
    // the wavefront's maximum and the first index of it go out NOW, in one 16-byte store that nobody waits for
    if (lane == 0)
      st_agent16(fa.emax + ((size_t(2 * g + r) * S + cb) * PW + wave) * 2, im >= 0 ? V : -INFINITY, double(fa.epoch) * kEpochUnit + double(im + 1));

    // ---- pass B: the highest strict peak.  The maximum itself, if it is one (and single: equal samples are told apart by the
    //      search); else the samples at or above 0.8 V, else all of them - exact each time
#if !defined(PAL_ABL_LEAN) || PAL_ABL_LEAN >= 2
    if (nhit == 1 && vpeak) {
      pt[r].hb = V; pt[r].mb = im; pt[r].plat = -INFINITY;
    } else {
#pragma nounroll
      for (int attempt = 0; attempt < 2; ++attempt) {
        const double pfloor = attempt == 0 && V > 0 ? 0.8 * V : -INFINITY;
        const double myfloor = inner ? pfloor : INFINITY;      // (the grid's edge columns are the finishing block's)
        double hb = -INFINITY, plat = -INFINITY;
        int mb = -1;
        for_slots(r, [&](double x, int slot) {
          if (__ballot(x >= myfloor)) {
            const int m = m2 + N2 * T(slot);
            const double left = from_lower_lane(x), right = from_upper_lane(x);
            const bool here = inner && (!PART || (m >= 1 && m <= n - 2));   // (the row's end points are never peaks; four-step / row chunks: the row ends inside a slot)
            const bool cand = here && x >= pfloor && (x > hb || (x == hb && m > mb));
            const bool pk = cand && left < x && right < x;
            hb = pk ? x : hb;
            mb = pk ? m : mb;
            plat = here && x >= pfloor && left == x ? fmax(plat, x) : plat;
          }
        });
        uniform_arg_sparse(hb, mb, [](double v1, int i1, double v2, int i2) { return higher(v1, i1, v2, i2); });
        pt[r].hb = hb; pt[r].mb = mb;
        pt[r].plat = uniform_max_sparse(plat);
        if (mb >= 0 || pfloor == -INFINITY) break;            // (uniform)
      }
    }
#endif
    // ---- the lag window and its margins: their highest strict peaks, whatever their height (a few slots meet them)
#if defined(PAL_ABL_LEAN) && PAL_ABL_LEAN < 3
    if (false) {
#else
    if (wmask) {
#endif
      double hq = -INFINITY, hg = -INFINITY, platw = -INFINITY;
      int mq = -1, mg = -1;
      for_slots(r, [&](double x, int slot) {
        if (!(wmask >> slot & 1u)) return;                     // (uniform)
        const int m = m2 + N2 * T(slot);
        const double left = from_lower_lane(x), right = from_upper_lane(x);
        const bool in = inner && m >= lo1 && m <= hi1;
        platw = in && (left == x || right == x) ? fmax(platw, x) : platw;
        const bool pk = in && left < x && right < x;
        const bool inw = m >= fa.win_lo && m <= fa.win_hi;
        if (pk && inw && (x > hq || (x == hq && m > mq))) { hq = x; mq = m; }
        if (pk && !inw && (x > hg || (x == hg && m > mg))) { hg = x; mg = m; }
      });
      uniform_arg_dense_hi(hq, mq);
      uniform_arg_sparse(hg, mg, [](double v1, int i1, double v2, int i2) { return higher(v1, i1, v2, i2); });
      pt[r].hw = hq; pt[r].mw = mq; pt[r].hm = hg; pt[r].mm = mg;
      pt[r].platw = uniform_max_sparse(platw);
    }
  }
  stamp();                                                     // 2: passes A and B
#if defined(PAL_ABL_LEAN) && PAL_ABL_LEAN < 4
  if (vn[0] + s1[0] + s2[0] + a1[0] + vn[1] + s1[1] + s2[1] + a1[1] + pt[0].hb + pt[1].hb + pt[0].hw + pt[1].hw == 1.2345e300) fa.status[3] = 1;
  return false;
#endif

  // ---- the grid's edge columns go to the finishing block as they are (blocks 0 and nblk - 1 only)
  if (edge_block) {                                            // (uniform)
    const bool mine = own && (m2 <= 1 || m2 >= N2 - 2);
    if (__ballot(mine)) {
      const int e = m2 <= 1 ? m2 : 3 - (N2 - 1 - m2);
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        if (r >= nrow) continue;
        double* dst = fa.edge + (size_t(2 * g + r) * 4 + (mine ? e : 0)) * N1;
        for_slots(r, [&](double x, int slot) {
          if (mine) st_agent(dst + T(slot), x);
        });
      }
    }
  }

  // ---- stored-row form (FinArgs.corr): the samples go to HBM as well, and nothing below waits for a sibling - the finisher reads
  //      the SNR window around the row's maximum from the stored row
  const bool stored = fa.corr != nullptr;
  if (stored && fa.store_rows) {
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      if (r >= nrow) continue;
      double* const out = fa.corr + size_t(2 * g + r) * fa.stride + m2;
      for_slots(r, [&](double x, int slot) {
        const int t = T(slot);
        if (own && (!PART || N2 * t + m2 < n)) out[size_t(N2) * t] = x;
      });
    }
  }

  // ---- the lanes' sums and minima, once: rows 2 g and 2 g + 1 share a butterfly (even lanes end with row 2 g's total, odd
  //      lanes with the other row's).  They stand HERE so that the siblings' maxima have more time to arrive
  const double rmin = wave_pair_min(vn[0], vn[1]);
  const double r1 = wave_pair_sum(s1[0], s1[1]), r2 = wave_pair_sum(s2[0], s2[1]), ra = wave_pair_sum(a1[0], a1[1]);

  // ---- the row's argmax from the siblings' maxima (published long ago), then the SNR window sums of this wavefront's samples
  double w1[2] = {0, 0}, w2[2] = {0, 0};
  bool gave_up = false;
  bool have_w[2] = {false, false};
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    if (r >= nrow || stored) continue;                         // (uniform)
    const int row = 2 * g + r;
    double bv = -INFINITY;
    int bi = -1;
    const double* em = fa.emax + size_t(row) * S * PW * 2;
    const double want = double(fa.epoch);
    bool late = false;
    for (int q = lane; q < S * PW; q += 64) {
      double v = 0, code = 0;
      int spins = 0;
      for (;;) {                                               // (the entry of this launch: its second word carries the launch number)
        ld_agent16(em + 2 * q, v, code);
        if (floor(code / kEpochUnit) == want) break;
        if (++spins > kSpinLimit) { late = true; break; }
        __builtin_amdgcn_s_sleep(8);
      }
      const int i = int(code - want * kEpochUnit) - 1;
      if (!late && i >= 0 && i < n && (bi < 0 || arg_better<0>(v, i, bv, bi))) { bv = v; bi = i; }
    }
    if (__ballot(late)) { gave_up = true; continue; }          // (uniform) the window sums are not valid
    uniform_arg_dense_lo(bv, bi);
    int imax = bi;
    if (imax < 0 || imax >= n) imax = 0;
    const int A = imax - pa.snr_w > 0 ? imax - pa.snr_w : 0, B = imax + pa.snr_w < n ? imax + pa.snr_w : n;   // [A, B)
    const int tA = A / N2, tB = (B - 1) / N2;
    const unsigned smask = slot_mask(tA, tB);
    if (smask == 0) continue;                                  // (uniform)
    have_w[r] = true;
    const bool inA = m2 >= A - tA * N2, inB = m2 < B - tB * N2;
    double u1 = 0, u2 = 0;
    for_slots(r, [&](double x, int slot) {
      if (!(smask >> slot & 1u)) return;                       // (uniform)
      const int t = T(slot);
      const bool in = own && (t > tA || inA) && (t < tB || inB);
      const double xm = in ? x : 0.0;
      u1 += xm;
      u2 = __builtin_fma(xm, xm, u2);
    });
    w1[r] = u1; w2[r] = u2;
  }
  if (gave_up && lane == 0) atomicAdd(fa.status + 13, 1);
  stamp();                                                     // 3: the row's argmax, window sums
#if defined(PAL_ABL_LEAN) && PAL_ABL_LEAN < 5
  if (rmin + r1 + r2 + ra + pt[0].hb + pt[1].hb + pt[0].hw + pt[1].hw + w1[0] + w2[0] + w1[1] + w2[1] == 1.2345e300) fa.status[3] = 1;
  return false;
#endif

  // ---- the window sums' butterfly; lane r publishes row r's results, then the wavefront's `done` word follows
  {
    double q1 = 0, q2 = 0;
    if (have_w[0] || have_w[1]) { q1 = wave_pair_sum(w1[0], w1[1]); q2 = wave_pair_sum(w2[0], w2[1]); }   // (uniform)
    if (lane < nrow) {
      FinPartial o;
      const bool second = lane == 1;
      o.hb = second ? pt[1].hb : pt[0].hb; o.plat = second ? pt[1].plat : pt[0].plat;
      o.hw = second ? pt[1].hw : pt[0].hw; o.platw = second ? pt[1].platw : pt[0].platw; o.hm = second ? pt[1].hm : pt[0].hm;
      o.mb = second ? pt[1].mb : pt[0].mb; o.mw = second ? pt[1].mw : pt[0].mw; o.mm = second ? pt[1].mm : pt[0].mm;
      o.vmin = rmin; o.s1 = r1; o.s2 = r2; o.a1 = ra; o.w1 = q1; o.w2 = q2;
      o.pad = gave_up ? 1 : 0;
      st_words(fa.parts + (size_t(2 * g + lane) * S + cb) * PW + wave, o);
    }
  }
#if !defined(PAL_ABL_LEAN) || PAL_ABL_LEAN != 6
  stores_done();                                               // this wavefront's stores (results, edge columns) have landed
#endif
  if (lane == 0) st_agent(fa.done + (size_t(g) * nblk + cb) * PW + wave, fa.epoch);
  stamp();                                                     // 4: published
  return cb == nblk - 1;
}


// ---- the finishing WAVEFRONT: one row from the wavefronts' published results (64 lanes, uniform control flow; the counterpart of
//      pfa_cols_fin.h fin_row for FinArgs.pw = 4 and no histograms - same rules, same order of the tests) ----
__device__ __forceinline__ void fin_row_wave(const PeakArgs& pa, const FinArgs& fa, int row, int N1, int N2, int lane) {
  const int S = pa.splits, n = pa.n, P = S * fa.pw;
  const bool windowed = fa.windowed != 0;
  const bool want_median = pa.method == 0;
  double vmax = 0, vmin = INFINITY, hb = 0, plat = -INFINITY, s1 = 0, s2 = 0, a1 = 0, w1 = 0, w2 = 0;
  int imax = -1, mb = -1;
  double hw = 0, hm = 0, platw = -INFINITY;
  int mw = -1, mm = -1;
  bool abandoned = false;
  {
    const double* em = fa.emax + size_t(row) * P * 2;
    for (int q = lane; q < P; q += 64) {
      const double v = ld_agent(em + 2 * q);
      const int i = int(ld_agent(em + 2 * q + 1) - double(fa.epoch) * kEpochUnit) - 1;      // (complete: every wavefront of the transform is done)
      if (i >= 0 && i < n && (imax < 0 || arg_better<0>(v, i, vmax, imax))) { vmax = v; imax = i; }
    }
  }
  for (int q = lane; q < P; q += 64) {
    const FinPartial pt = ld_words(fa.parts + size_t(row) * P + q);
    vmin = fmin(vmin, pt.vmin);
    if (pt.mb >= 0 && (mb < 0 || higher(pt.hb, pt.mb, hb, mb))) { hb = pt.hb; mb = pt.mb; }
    plat = fmax(plat, pt.plat);
    s1 += pt.s1; s2 += pt.s2; a1 += pt.a1;
    w1 += pt.w1; w2 += pt.w2;
    abandoned = abandoned || pt.pad != 0;
    if (windowed) {
      if (pt.mw >= 0 && (mw < 0 || higher(pt.hw, pt.mw, hw, mw))) { hw = pt.hw; mw = pt.mw; }
      if (pt.mm >= 0 && (mm < 0 || higher(pt.hm, pt.mm, hm, mm))) { hm = pt.hm; mm = pt.mm; }
      platw = fmax(platw, pt.platw);
    }
  }
  // the grid's first and last column: neighbours in another output index (m - 1 = (N2 - 1, t - 1), m + 1 = (0, t + 1))
  const double* E = fa.edge + size_t(row) * 4 * N1;
  for (int k = lane; k < 2 * N1; k += 64) {
    const bool first = k < N1;
    const int t = first ? k : k - N1;
    const int m = first ? N2 * t : N2 * t + N2 - 1;
    if (m < 1 || m > n - 2) continue;                          // the row's end points are never peaks
    const double x = ld_agent(first ? E + t : E + 3 * N1 + t);
    const double xl = ld_agent(first ? E + 3 * N1 + t - 1 : E + 2 * N1 + t);
    const double xr = ld_agent(first ? E + N1 + t : E + t + 1);
    const bool tie = xl == x || xr == x;
    const bool pk = xl < x && xr < x;
    const bool inw = windowed && m >= fa.win_lo && m <= fa.win_hi;
    const bool inm = windowed && !inw && m >= fa.win_lo - (pa.dist - 1) && m <= fa.win_hi + (pa.dist - 1);
    if (tie) plat = fmax(plat, x);
    if (tie && (inw || inm)) platw = fmax(platw, x);
    if (pk && (mb < 0 || higher(x, m, hb, mb))) { hb = x; mb = m; }
    if (pk && inw && (mw < 0 || higher(x, m, hw, mw))) { hw = x; mw = m; }
    if (pk && inm && (mm < 0 || higher(x, m, hm, mm))) { hm = x; mm = m; }
  }
  // across the lanes: everything ends uniform
  uniform_arg_dense_lo(vmax, imax);
  uniform_arg_dense_hi(hb, mb);
  if (windowed) {
    uniform_arg_dense_hi(hw, mw);
    uniform_arg_sparse(hm, mm, [](double v1, int i1, double v2, int i2) { return higher(v1, i1, v2, i2); });
    platw = uniform_max_sparse(platw);
  }
  plat = uniform_max_sparse(plat);
  vmin = wave_bcast63(wave_min63r(vmin));
  s1 = wave_bcast63(wave_sum63(s1)); s2 = wave_bcast63(wave_sum63(s2)); a1 = wave_bcast63(wave_sum63(a1));
  w1 = wave_bcast63(wave_sum63(w1)); w2 = wave_bcast63(wave_sum63(w2));
  abandoned = __ballot(abandoned) != 0;

  bool flag = false;                                           // the row needs its samples: stored-row path at the end of the call
  int why = 0;                                                 // (diagnostics: which rule flagged it)
  if (imax < 0 || imax >= n) { imax = 0; flag = true; why |= 1; }
  if (abandoned) { flag = true; why |= 1; }
  if (fa.corr) {                                               // stored-row form: the SNR window from the row itself (written by this XCD's blocks)
    const int lo = imax - pa.snr_w > 0 ? imax - pa.snr_w : 0, hi = imax + pa.snr_w < n ? imax + pa.snr_w : n;
    const double* rowp = fa.corr + size_t(row) * fa.stride;
    double u1 = 0, u2 = 0;
    for (int k = lo + lane; k < hi; k += 64) {
      const double x = ld_agent(rowp + k);
      u1 += x;
      u2 = __builtin_fma(x, x, u2);
    }
    w1 = wave_bcast63(wave_sum63(u1));
    w2 = wave_bcast63(wave_sum63(u2));
  }
  // a tie that may outrank the best strict peak (plateaus are resolved from the stored row)
  if (plat > -INFINITY && (mb < 0 || plat >= hb)) { flag = true; why |= 2; }
  if (windowed && platw > -INFINITY && (mw < 0 || platw >= hw)) { flag = true; why |= 4; }

  // ---- SNR (utils.py:238-250): totals minus the window around the maximum
  const int wlo_s = imax - pa.snr_w > 0 ? imax - pa.snr_w : 0;
  const int whi_s = imax + pa.snr_w < n ? imax + pa.snr_w : n;
  const double nn = double(n - (whi_s - wlo_s));
  const double o1 = s1 - w1, o2 = s2 - w2;
  if (!(o2 >= 0.25 * s2)) { flag = true; why |= 8; }           // the window holds most of the energy: two-pass sum of the noise region
  double var = (o2 - o1 * o1 / nn) / nn;
  if (var < 0) var = 0;
  const double noise = sqrt(var);
  const double snr = noise == 0.0 ? INFINITY : vmax / noise;

  // ---- primary threshold (utils.py:144-149): exact ('adaptive'), or the bound on the median (see pfa_cols_fin.h fin_row)
  double tlo = 0, thi = 0;
  if (!want_median) {
    double va = (s2 - a1 * a1 / double(n)) / double(n);
    if (va < 0) va = 0;
    tlo = thi = pa.mult * (a1 / double(n) + sqrt(va));         // utils.py:147
  } else {
    thi = pa.mult * sqrt(2.0 * s2 / double(n)) * (1.0 + 1e-12);
    tlo = -INFINITY;
  }

  // ---- the fallback chain of utils.py:152-179 for ONE peak
  const double mean_abs = a1 / double(n);
  int branch = 0, sel = imax;
  double sel_h = vmax;
  bool argmax_fallback = false;
  if (!flag) {
    bool alt = false;
    if (mb >= 0 && hb >= thi) {
    } else if (mb >= 0 && hb >= tlo) {
      flag = true;                                             // inside the median's interval
      why |= 32;
    } else {
      branch |= PAL_BR_ALT_THRESHOLD;
      alt = true;
      if (!(mb >= 0 && hb >= mean_abs)) { branch |= PAL_BR_ARGMAX_NO_PEAKS; argmax_fallback = true; }
    }
    if (!flag && !argmax_fallback) {
      if (!windowed) {
        sel = mb; sel_h = hb;                                  // the highest peak of the row is kept by the distance rule
      } else {
        const double t_lo = alt ? mean_abs : tlo, t_hi = alt ? mean_abs : thi;
        bool found = false;
        if (mw >= 0 && hw >= t_hi) found = true;
        else if (mw >= 0 && hw >= t_lo) { flag = true; why |= 64; }
        if (!flag && !found) {                                 // no peak of the first search inside the window: mean(|corr|), then argmax
          branch |= PAL_BR_WINDOW_RETRY;
          if (mw >= 0 && hw >= mean_abs) found = true;
          else { branch |= PAL_BR_ARGMAX_WINDOW; argmax_fallback = true; }
        }
        if (found) {
          // (pfa_cols_fin.h fin_row: only a HIGHER peak in the margins can suppress the window's best peak)
          const bool near = mw - fa.win_lo < pa.dist - 1 || fa.win_hi - mw < pa.dist - 1;
          if (near && mm >= 0 && higher(hm, mm, hw, mw)) { flag = true; why |= 128; }
          else { sel = mw; sel_h = hw; }
        }
      }
    }
    if (argmax_fallback) { sel = imax; sel_h = vmax; }
  }
  if (lane == 0) {
    fa.need[row] = flag ? 1 : 0;
    if (flag) {
      atomicAdd(fa.status + 4, 1);
      for (int b = 0; b < 8; ++b)
        if (why >> b & 1) atomicAdd(fa.status + 5 + b, 1);
    } else {
      pal_pair_record r;
      r.k_sel = sel; r.branch = branch; r.k_argmax = imax; r.n_sel = 1;
      r.cmax = vmax; r.cmin = vmin; r.snr = snr; r.sel_height = sel_h;
      fa.table[row] = r;
    }
  }
}

}  // namespace pal
