// fft_core.h - fp64 complex building blocks of the exact-length DFT engine (gfx950).
//
// The reference computes every transform with numpy.fft at the exact, non-smooth length
// n = n1+n2-1 (utils.py:113-118) or 2N (signal_processing.py:68-72).  On the GPU an exact
// length-n DFT is a Bluestein chirp convolution over a power-of-two length M that is cut
// four-step style into M = M1 x M2 with both sub-transforms resident in LDS:
//
//   a workgroup of 256 lanes (4 wavefronts of 64) owns 4096 complex doubles (64 KiB of LDS)
//   = T = 4096/N independent length-N sub-FFTs, Stockham autosort, radix 8 (+ one radix 4/2
//   tail), 16 points per lane in registers, two barriers per stage, in place.
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

constexpr int kPoints = 4096;   // complex points per workgroup
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
  cd o1 = INV ? mk((o[1].x - o[1].y) * h, (o[1].x + o[1].y) * h) : mk((o[1].x + o[1].y) * h, (o[1].y - o[1].x) * h);
  cd o2 = rot90<INV>(o[2]);
  cd o3 = INV ? mk((-o[3].x - o[3].y) * h, (o[3].x - o[3].y) * h) : mk((o[3].y - o[3].x) * h, (-o[3].x - o[3].y) * h);
  v[0] = e[0] + o[0];
  v[4] = e[0] - o[0];
  v[1] = e[1] + o1;
  v[5] = e[1] - o1;
  v[2] = e[2] + o2;
  v[6] = e[2] - o2;
  v[3] = e[3] + o3;
  v[7] = e[3] - o3;
}

template <int R, bool INV> PAL_HD void dftR(cd* v) {
  if (R == 8) dft8<INV>(v);
  else if (R == 4) dft4<INV>(v);
  else dft2<INV>(v);
}

// ---------------------------------------------------------------- stage plan
// radix of the stage that starts at log2(P) = lp in a length-2^ln transform: 8 while >= 3 bits remain.
PAL_HD constexpr int stage_radix(int ln, int lp) { return (ln - lp) >= 3 ? 8 : ((ln - lp) == 2 ? 4 : 2); }
PAL_HD constexpr int stage_log2r(int ln, int lp) { return (ln - lp) >= 3 ? 3 : (ln - lp); }

// Stage-major twiddle table: for every stage with P > 1, entries [(r-1)*P + k] = exp(-2 pi i k r / (P R)),
// k < P, 1 <= r < R, so that the lanes of a wavefront (consecutive k) read consecutive entries.
PAL_HD constexpr int stage_tw_offset(int ln, int lp) {
  int off = 0;
  for (int q = stage_log2r(ln, 0); q < lp; q += stage_log2r(ln, q)) off += (stage_radix(ln, q) - 1) * (1 << q);
  return off;
}
PAL_HD constexpr int stage_tw_size(int ln) { return stage_tw_offset(ln, ln); }   // < 2^ln

// LDS address of element e of sub-transform t.
//   COLS: t fastest (lanes of a wavefront walk the T columns of a tile, the copy to and from global
//         memory keeps the same order, no conflicts for T >= 8).
//   ROWS: e fastest with an XOR swizzle of the low three bits by the next three, which spreads the
//         stride-8 writes of the first radix-8 stage over all banks.
template <int LOG2N, bool COLS, int NSUB = (kPoints >> LOG2N)> PAL_HD int lds_addr(int t, int e) {
  constexpr int N = 1 << LOG2N;
  return COLS ? e * NSUB + t : t * N + (e ^ ((e >> 3) & 7));
}

// work item w in [0, 4096/R) -> (butterfly i, sub-transform t)
template <int LOG2N, bool COLS, int R, int NSUB = (kPoints >> LOG2N)> PAL_HD void item_of(int w, int& i, int& t) {
  constexpr int N = 1 << LOG2N, NB = N / R;
  if (COLS) { t = w % NSUB; i = w / NSUB; } else { i = w % NB; t = w / NB; }
}

// read the R inputs of work item w, apply the stage twiddles, run the radix-R DFT
template <int LOG2N, bool COLS, bool INV, int LOG2P, int NSUB = (kPoints >> LOG2N)>
PAL_HD void stage_load(const cd* data, const cd* tw, int w, cd* v) {
  constexpr int R = stage_radix(LOG2N, LOG2P), N = 1 << LOG2N, P = 1 << LOG2P, NB = N / R;
  int i, t;
  item_of<LOG2N, COLS, R, NSUB>(w, i, t);
#pragma unroll
  for (int r = 0; r < R; ++r) v[r] = data[lds_addr<LOG2N, COLS, NSUB>(t, i + r * NB)];
  if (P > 1) {
    const int k = i & (P - 1);
    const cd* tws = tw + stage_tw_offset(LOG2N, LOG2P);
#pragma unroll
    for (int r = 1; r < R; ++r) {
      cd f = tws[(r - 1) * P + k];
      v[r] = INV ? cmulc(v[r], f) : cmul(v[r], f);
    }
  }
  dftR<R, INV>(v);
}

// write the R outputs of work item w to their autosort positions
template <int LOG2N, bool COLS, int LOG2P, int NSUB = (kPoints >> LOG2N)>
PAL_HD void stage_store(cd* data, int w, const cd* v) {
  constexpr int R = stage_radix(LOG2N, LOG2P), P = 1 << LOG2P;
  int i, t;
  item_of<LOG2N, COLS, R, NSUB>(w, i, t);
  const int k = i & (P - 1);
  const int j0 = (i - k) * R + k;
#pragma unroll
  for (int r = 0; r < R; ++r) data[lds_addr<LOG2N, COLS, NSUB>(t, j0 + r * P)] = v[r];
}

// ---------------------------------------------------------------- radix-3 outer stage (column mode)
// A column transform of length 3N keeps its three length-N sub-sequences as sub-transforms q*T + c of the
// LDS tile (T columns, NSUB = 3T).  Element e of sub-transform (q, c) is row  q*N + e  on the time side and
// row  3e + q  on the frequency side, so one in-place 3-point butterfly per (e, c) links both sides:
//   forward (decimation in frequency, before the N-point FFTs):  y_r[e] = w^(r e) sum_s w3^(r s) x[e + s N]
//   inverse (after the inverse N-point FFTs):                    x[e + s N] = sum_r w3^(-r s) conj(w^(r e)) y_r[e]
// with w = exp(-2 pi i / 3N) read from `roots` (exp(-2 pi i q / 3N), q < 3N) and w3 = exp(-2 pi i / 3).
template <int LOG2N, int T, bool INV> PAL_HD void radix3_item(cd* data, const cd* roots, int w) {
  constexpr int NSUB = 3 * T;
  const int c = w % T, e = w / T;
  cd a0 = data[lds_addr<LOG2N, true, NSUB>(c, e)];
  cd a1 = data[lds_addr<LOG2N, true, NSUB>(T + c, e)];
  cd a2 = data[lds_addr<LOG2N, true, NSUB>(2 * T + c, e)];
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
  data[lds_addr<LOG2N, true, NSUB>(c, e)] = y0;
  data[lds_addr<LOG2N, true, NSUB>(T + c, e)] = y1;
  data[lds_addr<LOG2N, true, NSUB>(2 * T + c, e)] = y2;
}

#if defined(__HIPCC__)
// ---------------------------------------------------------------- workgroup transform (device)
template <int LOG2N, bool COLS, bool INV, int LOG2P, int NSUB>
__device__ __forceinline__ void wg_fft_from(cd* data, const cd* tw, int tid) {
  if constexpr (LOG2P < LOG2N) {
    constexpr int R = stage_radix(LOG2N, LOG2P);
    constexpr int POINTS = NSUB << LOG2N, LANES = POINTS / 16;   // 16 points per lane: 256 lanes (192 with a radix-3 stage)
    constexpr int PER = POINTS / R / LANES;                      // work items per lane: 2, 4 or 8
    cd v[PER][R];
#pragma unroll
    for (int q = 0; q < PER; ++q) stage_load<LOG2N, COLS, INV, LOG2P, NSUB>(data, tw, tid + LANES * q, v[q]);
    __syncthreads();
#pragma unroll
    for (int q = 0; q < PER; ++q) stage_store<LOG2N, COLS, LOG2P, NSUB>(data, tid + LANES * q, v[q]);
    __syncthreads();
    wg_fft_from<LOG2N, COLS, INV, LOG2P + stage_log2r(LOG2N, LOG2P), NSUB>(data, tw, tid);
  }
}

// NSUB transforms of length 2^LOG2N (default 4096 points), in place in LDS; the caller has already
// synchronised after filling `data` and `tw`, and the result is visible to all lanes on return.
template <int LOG2N, bool COLS, bool INV, int NSUB = (kPoints >> LOG2N)>
__device__ __forceinline__ void wg_fft(cd* data, const cd* tw, int tid) {
  wg_fft_from<LOG2N, COLS, INV, 0, NSUB>(data, tw, tid);
}
#endif

}  // namespace pal
