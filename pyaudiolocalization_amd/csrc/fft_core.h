// fft_core.h - fp64 complex building blocks of the exact-length DFT engine (gfx950).
//
// The reference computes every transform with numpy.fft at the exact, non-smooth length
// n = n1+n2-1 (utils.py:113-118) or 2N (signal_processing.py:68-72).  On the GPU an exact
// length-n DFT is a Bluestein chirp convolution over a length M = M1 x M2 (2^k or 3 * 2^k) that is cut
// four-step style so that both sub-transforms are resident in LDS:
//
//   a workgroup owns NSUB independent length-N sub-FFTs (4096 points / 256 lanes, or 3072 / 192 with
//   a radix-3 column stage), Stockham autosort, radix 16 / 8 (+ a radix 4 / 2 tail), 16 points per
//   lane in registers, in place.  The FIRST stage reads its inputs through a functor (global memory,
//   or a loader that builds the input on the fly) and the LAST stage hands its outputs to a functor
//   (global memory, a storer, or LDS with a pointwise multiply), so a transform costs one LDS round
//   trip per stage boundary and none for staging - LDS stores (about 80 B/clk/CU) are the scarce
//   resource of these kernels, not the butterflies.
//
// Index math lives in __host__ __device__ functions so tests/host/test_fft_core.cpp can
// execute the identical stage code lane by lane on the CPU.
#pragma once
#include <hip/hip_runtime.h>

#define PAL_HD __host__ __device__ __forceinline__

namespace pal {

struct cd { double x, y; };

PAL_HD cd mk(double x, double y) { cd r; r.x = x; r.y = y; return r; }
PAL_HD cd operator+(cd a, cd b) { return mk(a.x + b.x, a.y + b.y); }
PAL_HD cd operator-(cd a, cd b) { return mk(a.x - b.x, a.y - b.y); }
// complex products with explicit fused multiply-adds (two instructions per component on gfx950,
// independent of the translation unit's -ffp-contract setting)
PAL_HD cd cmul(cd a, cd b) {
  return mk(__builtin_fma(a.x, b.x, -(a.y * b.y)), __builtin_fma(a.x, b.y, a.y * b.x));
}
PAL_HD cd cmulc(cd a, cd b) {   // a * conj(b)
  return mk(__builtin_fma(a.x, b.x, a.y * b.y), __builtin_fma(a.y, b.x, -(a.x * b.y)));
}
PAL_HD cd cconj(cd a) { return mk(a.x, -a.y); }
PAL_HD cd cscale(cd a, double s) { return mk(a.x * s, a.y * s); }

// multiply by -i (forward transform) or +i (inverse transform)
template <bool INV> PAL_HD cd rot90(cd a) { return INV ? mk(-a.y, a.x) : mk(a.y, -a.x); }
// multiply by exp(-/+ 2 pi i * (cos, sin pair given for the forward direction))
template <bool INV> PAL_HD cd rotc(cd a, double c, double s) {   // forward factor = (c, -s)
  return INV ? mk(__builtin_fma(a.x, c, -(a.y * s)), __builtin_fma(a.x, s, a.y * c))
             : mk(__builtin_fma(a.x, c, a.y * s), __builtin_fma(a.y, c, -(a.x * s)));
}

constexpr int kPoints = 4096;   // complex points per workgroup (power-of-two passes)
constexpr int kLanes = 256;     // lanes per workgroup (4 wavefronts)

// ---------------------------------------------------------------- small DFTs, in registers
template <bool INV> PAL_HD void dft2(cd* v) {
  cd a = v[0], b = v[1];
  v[0] = a + b;
  v[1] = a - b;
}

template <bool INV> PAL_HD void dft4(cd* v) {
  cd t0 = v[0] + v[2], t1 = v[0] - v[2], t2 = v[1] + v[3], t3 = rot90<INV>(v[1] - v[3]);
  v[0] = t0 + t2;
  v[1] = t1 + t3;
  v[2] = t0 - t2;
  v[3] = t1 - t3;
}

template <bool INV> PAL_HD void dft8(cd* v) {
  const double h = 0.70710678118654752440;
  cd e[4] = {v[0], v[2], v[4], v[6]};
  cd o[4] = {v[1], v[3], v[5], v[7]};
  dft4<INV>(e);
  dft4<INV>(o);
  // o[k] *= exp(-/+ 2 pi i k / 8)
  cd o1 = rotc<INV>(o[1], h, h);
  cd o2 = rot90<INV>(o[2]);
  cd o3 = rotc<INV>(o[3], -h, h);
  v[0] = e[0] + o[0];
  v[4] = e[0] - o[0];
  v[1] = e[1] + o1;
  v[5] = e[1] - o1;
  v[2] = e[2] + o2;
  v[6] = e[2] - o2;
  v[3] = e[3] + o3;
  v[7] = e[3] - o3;
}

// 16 = 4 x 4: input n = 4a + b, output k = c + 4d;  X[c + 4d] = sum_b W4^(bd) W16^(bc) sum_a W4^(ac) x[4a + b]
template <bool INV> PAL_HD void dft16(cd* v) {
  const double c1 = 0.92387953251128675613, s1 = 0.38268343236508977173;   // cos, sin of pi/8
  const double h = 0.70710678118654752440;
  cd u[4][4];
#pragma unroll
  for (int b = 0; b < 4; ++b) {
    cd t[4] = {v[b], v[4 + b], v[8 + b], v[12 + b]};
    dft4<INV>(t);
#pragma unroll
    for (int c = 0; c < 4; ++c) u[b][c] = t[c];
  }
  u[1][1] = rotc<INV>(u[1][1], c1, s1);     // W16^1
  u[1][2] = rotc<INV>(u[1][2], h, h);       // W16^2
  u[1][3] = rotc<INV>(u[1][3], s1, c1);     // W16^3
  u[2][1] = rotc<INV>(u[2][1], h, h);       // W16^2
  u[2][2] = rot90<INV>(u[2][2]);            // W16^4
  u[2][3] = rotc<INV>(u[2][3], -h, h);      // W16^6
  u[3][1] = rotc<INV>(u[3][1], s1, c1);     // W16^3
  u[3][2] = rotc<INV>(u[3][2], -h, h);      // W16^6
  u[3][3] = rotc<INV>(u[3][3], -c1, -s1);   // W16^9
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    cd t[4] = {u[0][c], u[1][c], u[2][c], u[3][c]};
    dft4<INV>(t);
#pragma unroll
    for (int d = 0; d < 4; ++d) v[c + 4 * d] = t[d];
  }
}

template <int R, bool INV> PAL_HD void dftR(cd* v) {
  if (R == 16) dft16<INV>(v);
  else if (R == 8) dft8<INV>(v);
  else if (R == 4) dft4<INV>(v);
  else dft2<INV>(v);
}

// ---------------------------------------------------------------- stage plan
// log2 of the radix of the stage that starts with lp bits done in a length-2^ln transform: 16 when exactly
// four or at least seven bits remain, 4 when five remain (so that the LAST stage is the radix-8 one: the fused seam of the
// prime-factor row pass needs a last stage of radix 8 or 16), else 8, then a 4 / 2 tail:
// 4:[16] 5:[4,8] 6:[8,8] 7:[16,8] 8:[16,16] 9:[16,4,8] 10:[16,8,8] 11:[16,16,8] 12:[16,16,16] 13:[16,16,4,8] 14:[16,16,8,8]
PAL_HD constexpr int stage_log2r(int ln, int lp) {
  return (ln - lp) == 4 || (ln - lp) >= 7 ? 4 : ((ln - lp) == 5 ? 2 : ((ln - lp) >= 3 ? 3 : (ln - lp)));
}
PAL_HD constexpr int stage_radix(int ln, int lp) { return 1 << stage_log2r(ln, lp); }
PAL_HD constexpr bool stage_is_last(int ln, int lp) { return lp + stage_log2r(ln, lp) >= ln; }

// Stage-major twiddle table: for every stage with P > 1, entries [(r-1)*P + k] = exp(-2 pi i k r / (P R)),
// k < P, 1 <= r < R, so that the lanes of a wavefront (consecutive k) read consecutive entries.
PAL_HD constexpr int stage_tw_offset(int ln, int lp) {
  int off = 0;
  for (int q = stage_log2r(ln, 0); q < lp; q += stage_log2r(ln, q)) off += (stage_radix(ln, q) - 1) * (1 << q);
  return off;
}
PAL_HD constexpr int stage_tw_size(int ln) { return stage_tw_offset(ln, ln); }   // < 2^ln
// Compact variant: the LAST stage (the one with the largest P) keeps only the rows r = 1, 2, 4 (, 8) at the same
// offset - entries [b*P + k] = exp(-2 pi i k 2^b / (P R)) - and the other factors are products of those rows
// (at most three multiplies deep).  A 2048-point transform then needs 1008 table entries instead of 2032, which
// is what lets two 64 KB workgroups share a CU's LDS.
PAL_HD constexpr int stage_tw_last(int ln) {
  int lp = 0;
  while (!stage_is_last(ln, lp)) lp += stage_log2r(ln, lp);
  return lp;
}
PAL_HD constexpr int stage_twc_size(int ln) {
  return stage_tw_offset(ln, stage_tw_last(ln)) + stage_log2r(ln, stage_tw_last(ln)) * (1 << stage_tw_last(ln));
}

// LDS address of element e of sub-transform t.
//   COLS: t fastest (lanes of a wavefront walk the columns of a tile, no conflicts for >= 8 columns).
//   ROWS: e fastest with an XOR swizzle of the low three bits by bits 3..5 and 6..8, which spreads the
//         stride-16 / stride-8 stores of the first stage over all banks.
template <int LOG2N, bool COLS, int NSUB = (kPoints >> LOG2N)> PAL_HD int lds_addr(int t, int e) {
  constexpr int N = 1 << LOG2N;
  return COLS ? e * NSUB + t : t * N + (e ^ (((e >> 3) ^ (e >> 6)) & 7));
}

// element accessors of the in-LDS tile (the default source / sink of a stage)
//
// A stage touches elements e = e0 + c with a per-lane part e0 and a compile-time part c whose bits do not
// overlap (e0 = i < N/R, c = r N/R on the load side; e0 = j0, c = r P on the store side).  The XOR swizzle of the
// rows layout then splits as well, swz(e0 | c) = swz(e0) ^ swz(c), so the address is `base(t, e0)` once per work
// item plus, per element, a constant offset and at most one XOR with a constant below 8 - instead of a fresh
// swizzle per element (a fifth of the row kernels' vector instructions were address arithmetic).
template <int LOG2N, bool COLS, int NSUB = (kPoints >> LOG2N)> struct LdsTile {
  static constexpr bool kLds = true;
  static constexpr bool kDirect = true;     // offers base() / at(): stage_load / stage_store use the split addressing
  cd* data;
  PAL_HD cd operator()(int t, int e) const { return data[lds_addr<LOG2N, COLS, NSUB>(t, e)]; }
  PAL_HD void operator()(int t, int e, cd v) const { data[lds_addr<LOG2N, COLS, NSUB>(t, e)] = v; }
  PAL_HD int base(int t, int e0) const { return lds_addr<LOG2N, COLS, NSUB>(t, e0); }
  static PAL_HD constexpr int split(int b, int c) {   // address of element e0 | c given b = base(t, e0)
    // rows: (e0 | c) ^ swz(e0) ^ swz(c); bits >= 3 of c meet zeros of b (add = xor: an immediate offset), bits 0-2 are xor-ed
    return COLS ? b + c * NSUB : (b ^ ((c ^ (c >> 3) ^ (c >> 6)) & 7)) + (c & ~7);
  }
  PAL_HD cd at(int b, int c) const { return data[split(b, c)]; }
  PAL_HD void at(int b, int c, cd v) const { data[split(b, c)] = v; }
};

template <class T, class = void> struct has_direct { static constexpr bool value = false; };
template <class T> struct has_direct<T, decltype(void(T::kDirect))> { static constexpr bool value = true; };

// work item w in [0, POINTS/R) -> (butterfly i, sub-transform t)
template <int LOG2N, bool COLS, int R, int NSUB = (kPoints >> LOG2N)> PAL_HD void item_of(int w, int& i, int& t) {
  constexpr int N = 1 << LOG2N, NB = N / R;
  if (COLS) { t = w % NSUB; i = w / NSUB; } else { i = w % NB; t = w / NB; }
}

// read the R inputs of work item w through `in(t, e)`, apply the stage twiddles, run the radix-R DFT
template <int LOG2N, bool COLS, bool INV, int LOG2P, int NSUB, bool COMPACT = false, class In>
PAL_HD void stage_load(const In& in, const cd* tw, int w, cd* v) {
  constexpr int R = stage_radix(LOG2N, LOG2P), N = 1 << LOG2N, P = 1 << LOG2P, NB = N / R;
  int i, t;
  item_of<LOG2N, COLS, R, NSUB>(w, i, t);
  if constexpr (has_direct<In>::value) {
    const int b = in.base(t, i);
#pragma unroll
    for (int r = 0; r < R; ++r) v[r] = in.at(b, r * NB);
  } else {
#pragma unroll
    for (int r = 0; r < R; ++r) v[r] = in(t, i + r * NB);
  }
  if (P > 1) {
    const int k = i & (P - 1);
    const cd* tws = tw + stage_tw_offset(LOG2N, LOG2P);
    if (COMPACT && stage_is_last(LOG2N, LOG2P)) {
      cd f[R];
#pragma unroll
      for (int b = 0; (1 << b) < R; ++b) f[1 << b] = tws[b * P + k];
#pragma unroll
      for (int r = 3; r < R; ++r) {
        const int hi = r >= 8 ? 8 : (r >= 4 ? 4 : 2);
        if (r != hi) f[r] = cmul(f[hi], f[r - hi]);
      }
#pragma unroll
      for (int r = 1; r < R; ++r) v[r] = INV ? cmulc(v[r], f[r]) : cmul(v[r], f[r]);
    } else {
#pragma unroll
      for (int r = 1; r < R; ++r) {
        cd f = tws[(r - 1) * P + k];
        v[r] = INV ? cmulc(v[r], f) : cmul(v[r], f);
      }
    }
  }
  dftR<R, INV>(v);
}

// the same with the stage's twiddle factors handed over in registers: f[r] = exp(-2 pi i k r / (P R)), 1 <= r < R
template <int LOG2N, bool COLS, bool INV, int LOG2P, int NSUB, class In>
PAL_HD void stage_load_with(const In& in, int w, cd* v, const cd* f) {
  constexpr int R = stage_radix(LOG2N, LOG2P), N = 1 << LOG2N, NB = N / R;
  int i, t;
  item_of<LOG2N, COLS, R, NSUB>(w, i, t);
  if constexpr (has_direct<In>::value) {
    const int b = in.base(t, i);
#pragma unroll
    for (int r = 0; r < R; ++r) v[r] = in.at(b, r * NB);
  } else {
#pragma unroll
    for (int r = 0; r < R; ++r) v[r] = in(t, i + r * NB);
  }
#pragma unroll
  for (int r = 1; r < R; ++r) v[r] = INV ? cmulc(v[r], f[r]) : cmul(v[r], f[r]);
  dftR<R, INV>(v);
}

// hand the R outputs of work item w to `out(t, e, value)` at their autosort positions
template <int LOG2N, bool COLS, int LOG2P, int NSUB, class Out>
PAL_HD void stage_store(const Out& out, int w, const cd* v) {
  constexpr int R = stage_radix(LOG2N, LOG2P), P = 1 << LOG2P;
  int i, t;
  item_of<LOG2N, COLS, R, NSUB>(w, i, t);
  const int k = i & (P - 1);
  const int j0 = (i - k) * R + k;
  if constexpr (has_direct<Out>::value) {
    const int b = out.base(t, j0);
#pragma unroll
    for (int r = 0; r < R; ++r) out.at(b, r * P, v[r]);
  } else {
#pragma unroll
    for (int r = 0; r < R; ++r) out(t, j0 + r * P, v[r]);
  }
}

// ---------------------------------------------------------------- radix-3 outer stage (column mode)
// A column transform of length 3N keeps its three length-N sub-sequences as sub-transforms q*T + c of the
// LDS tile (T columns, NSUB = 3T).  Element e of sub-transform (q, c) is row  q*N + e  on the time side and
// row  3e + q  on the frequency side, so one 3-point butterfly per (e, c) links both sides:
//   forward (decimation in frequency, before the N-point FFTs):  y_r[e] = w^(r e) sum_s w3^(r s) x[e + s N]
//   inverse (after the inverse N-point FFTs):                    x[e + s N] = sum_r w3^(-r s) conj(w^(r e)) y_r[e]
// with w = exp(-2 pi i / 3N) read from `roots` (exp(-2 pi i q / 3N), q < 3N) and w3 = exp(-2 pi i / 3).
// `in(q, c, e)` supplies the three inputs and `out(q, c, e, value)` takes the three outputs.
template <int T, bool INV, class In, class Out>
PAL_HD void radix3_item(const In& in, const Out& out, const cd* roots, int w) {
  const int c = w % T, e = w / T;
  cd a0 = in(0, c, e), a1 = in(1, c, e), a2 = in(2, c, e);
  if (INV) {
    a1 = cmulc(a1, roots[e]);
    a2 = cmulc(a2, roots[2 * e]);
  }
  const double h = 0.86602540378443864676;            // sin(pi/3)
  const cd sum = a1 + a2, dif = a1 - a2;
  const cd mid = mk(a0.x - 0.5 * sum.x, a0.y - 0.5 * sum.y);
  const cd rot = INV ? mk(-h * dif.y, h * dif.x) : mk(h * dif.y, -h * dif.x);   // (+/- i sqrt(3)/2) (a1 - a2)
  cd y0 = a0 + sum, y1 = mid + rot, y2 = mid - rot;
  if (!INV) {
    y1 = cmul(y1, roots[e]);
    y2 = cmul(y2, roots[2 * e]);
  }
  out(0, c, e, y0);
  out(1, c, e, y1);
  out(2, c, e, y2);
}

// the (q, c, e) view of an LDS tile with NSUB = 3T sub-transforms
template <int LOG2N, int T> struct LdsTile3 {
  cd* data;
  PAL_HD cd operator()(int q, int c, int e) const { return data[lds_addr<LOG2N, true, 3 * T>(q * T + c, e)]; }
  PAL_HD void operator()(int q, int c, int e, cd v) const { data[lds_addr<LOG2N, true, 3 * T>(q * T + c, e)] = v; }
};

#if defined(__HIPCC__)
// ---------------------------------------------------------------- workgroup transform (device)
// One stage: every lane loads its work items through `in`, all LDS reads finish before any LDS write (in
// place), outputs go to `out`, and LDS outputs are visible to all lanes on return.
template <int LOG2N, bool COLS, bool INV, int LOG2P, int NSUB, bool COMPACT, class In, class Out>
__device__ __forceinline__ void wg_stage(const cd* tw, int tid, const In& in, const Out& out) {
  constexpr int R = stage_radix(LOG2N, LOG2P);
  constexpr int POINTS = NSUB << LOG2N, LANES = POINTS / 16;   // 16 points per lane: 256 lanes (192 with a radix-3 stage)
  constexpr int PER = POINTS / R / LANES;                      // work items per lane: 1, 2, 4 or 8
  cd v[PER][R];
#pragma unroll
  for (int q = 0; q < PER; ++q) stage_load<LOG2N, COLS, INV, LOG2P, NSUB, COMPACT>(in, tw, tid + LANES * q, v[q]);
  if constexpr (In::kLds && Out::kLds) __syncthreads();
#pragma unroll
  for (int q = 0; q < PER; ++q) stage_store<LOG2N, COLS, LOG2P, NSUB>(out, tid + LANES * q, v[q]);
  if constexpr (Out::kLds) __syncthreads();
}

template <int LOG2N, bool COLS, bool INV, int LOG2P, int NSUB, bool COMPACT, class FirstIn, class LastOut>
__device__ __forceinline__ void wg_fft_from(cd* data, const cd* tw, int tid, const FirstIn& first, const LastOut& last) {
  if constexpr (LOG2P < LOG2N) {
    constexpr bool kFirst = LOG2P == 0, kLast = stage_is_last(LOG2N, LOG2P);
    const LdsTile<LOG2N, COLS, NSUB> tile{data};
    if constexpr (kFirst && kLast) wg_stage<LOG2N, COLS, INV, LOG2P, NSUB, COMPACT>(tw, tid, first, last);
    else if constexpr (kFirst) wg_stage<LOG2N, COLS, INV, LOG2P, NSUB, COMPACT>(tw, tid, first, tile);
    else if constexpr (kLast) wg_stage<LOG2N, COLS, INV, LOG2P, NSUB, COMPACT>(tw, tid, tile, last);
    else wg_stage<LOG2N, COLS, INV, LOG2P, NSUB, COMPACT>(tw, tid, tile, tile);
    wg_fft_from<LOG2N, COLS, INV, LOG2P + stage_log2r(LOG2N, LOG2P), NSUB, COMPACT>(data, tw, tid, first, last);
  }
}

// the LDS-to-LDS stages from LOG2P up to, not including, the last one
template <int LOG2N, bool COLS, bool INV, int LOG2P, int NSUB, bool COMPACT>
__device__ __forceinline__ void wg_fft_middle(cd* data, const cd* tw, int tid) {
  if constexpr (!stage_is_last(LOG2N, LOG2P)) {
    const LdsTile<LOG2N, COLS, NSUB> tile{data};
    wg_stage<LOG2N, COLS, INV, LOG2P, NSUB, COMPACT>(tw, tid, tile, tile);
    wg_fft_middle<LOG2N, COLS, INV, LOG2P + stage_log2r(LOG2N, LOG2P), NSUB, COMPACT>(data, tw, tid);
  }
}

// NSUB transforms of length 2^LOG2N through the LDS tile `data`.  `first(t, e)` feeds the first stage and
// `last(t, e, value)` receives the last stage's outputs; either may be an LdsTile (kLds = true) or touch
// global memory (kLds = false).  Contract: if `first` reads LDS the caller has synchronised after filling
// it; if `last` writes LDS the result is visible to all lanes on return; if `last` does not, the tile may
// still be read by other lanes on return - synchronise before overwriting it.  `tw` (stage-major table in
// LDS; the compact layout when COMPACT) must be visible before the call whenever the transform has more than
// one stage.
template <int LOG2N, bool COLS, bool INV, int NSUB, bool COMPACT = false, class FirstIn, class LastOut>
__device__ __forceinline__ void wg_fft(cd* data, const cd* tw, int tid, const FirstIn& first, const LastOut& last) {
  wg_fft_from<LOG2N, COLS, INV, 0, NSUB, COMPACT>(data, tw, tid, first, last);
}
#endif

}  // namespace pal
