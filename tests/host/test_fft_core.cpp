// Host execution of the workgroup FFT stage code of csrc/fft_core.h, lane by lane, against a
// naive O(N^2) DFT.  Built and run by tests/test_host_fft_core.py (no GPU needed):
//   hipcc -O2 -I pyaudiolocalization_amd/csrc tests/host/test_fft_core.cpp -o /tmp/test_fft_core
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "fft_core.h"
#include "mixed_radix.h"

using namespace pal;

static void make_stage_tw(int ln, std::vector<cd>& tw) {
  tw.assign(1 << ln, mk(0, 0));
  for (int lp = stage_log2r(ln, 0); lp < ln; lp += stage_log2r(ln, lp)) {
    const int R = stage_radix(ln, lp), P = 1 << lp, off = stage_tw_offset(ln, lp);
    for (int r = 1; r < R; ++r)
      for (int k = 0; k < P; ++k) {
        const double a = -2.0 * M_PI * double(k * r) / double(P * R);
        tw[off + (r - 1) * P + k] = mk(std::cos(a), std::sin(a));
      }
  }
}

// the compact table of the prime-factor row pass: the last stage keeps the rows r = 1, 2, 4 (, 8) only
static void make_stage_tw_compact(int ln, std::vector<cd>& tw) {
  make_stage_tw(ln, tw);
  const int lp = stage_tw_last(ln), R = stage_radix(ln, lp), P = 1 << lp, off = stage_tw_offset(ln, lp);
  for (int i = off; i < (1 << ln); ++i) tw[i] = mk(1e300, 1e300);       // poison what the compact layout must not read
  for (int b = 0; (1 << b) < R; ++b)
    for (int k = 0; k < P; ++k) {
      const double a = -2.0 * M_PI * double(k * (1 << b)) / double(P * R);
      tw[off + b * P + k] = mk(std::cos(a), std::sin(a));
    }
}

// every stage through the LDS-tile accessor, reads of a stage before its writes (what the barriers enforce);
// STOP: first stage NOT to run (LOG2N = all of them)
template <int LOG2N, bool COLS, bool INV, int LOG2P, int NSUB = (kPoints >> LOG2N), bool COMPACT = false, int STOP = LOG2N>
static void emu_from(std::vector<cd>& data, const std::vector<cd>& tw) {
  if constexpr (LOG2P < STOP) {
    constexpr int R = stage_radix(LOG2N, LOG2P);
    constexpr int ITEMS = (NSUB << LOG2N) / R;
    const LdsTile<LOG2N, COLS, NSUB> tile{data.data()};
    std::vector<cd> regs(size_t(ITEMS) * R);
    for (int w = 0; w < ITEMS; ++w) stage_load<LOG2N, COLS, INV, LOG2P, NSUB, COMPACT>(tile, tw.data(), w, &regs[size_t(w) * R]);
    for (int w = 0; w < ITEMS; ++w) stage_store<LOG2N, COLS, LOG2P, NSUB>(tile, w, &regs[size_t(w) * R]);
    emu_from<LOG2N, COLS, INV, LOG2P + stage_log2r(LOG2N, LOG2P), NSUB, COMPACT, STOP>(data, tw);
  }
}

static double naive_err(const std::vector<cd>& plain, size_t off, int len, int o, bool inv, cd got) {
  long double sx = 0, sy = 0;
  for (int i = 0; i < len; ++i) {
    const long double a = (inv ? 2.0L : -2.0L) * M_PIl * (long double)((long long)o * i % len) / len;
    const long double cs = cosl(a), sn = sinl(a);
    sx += plain[off + i].x * cs - plain[off + i].y * sn;
    sy += plain[off + i].x * sn + plain[off + i].y * cs;
  }
  return std::fmax(std::fabs(double(got.x - sx)), std::fabs(double(got.y - sy)));
}

// column transform of length 3N (radix-3 outer stage + N-point sub-transforms), T = 1024/N columns per tile
template <int LOG2N, bool INV> static double check3() {
  constexpr int N = 1 << LOG2N, T = 1024 / N, NSUB = 3 * T, M1 = 3 * N;
  std::vector<cd> tw;
  make_stage_tw(LOG2N, tw);
  std::vector<cd> roots(M1);
  for (int q = 0; q < M1; ++q) roots[q] = mk(std::cos(-2.0 * M_PI * q / M1), std::sin(-2.0 * M_PI * q / M1));
  std::vector<cd> data(size_t(NSUB) << LOG2N), plain(size_t(T) * M1);
  const LdsTile3<LOG2N, T> tile3{data.data()};
  for (int c = 0; c < T; ++c)
    for (int i = 0; i < M1; ++i) {
      const cd v = mk(drand48() - 0.5, drand48() - 0.5);
      plain[size_t(c) * M1 + i] = v;
      // forward input: time side, row i = q N + e;  inverse input: frequency side, row i = 3 e + q
      tile3(INV ? i % 3 : i / N, c, INV ? i / 3 : i % N, v);
    }
  if (!INV)
    for (int w = 0; w < N * T; ++w) radix3_item<T, false>(tile3, tile3, roots.data(), w);
  emu_from<LOG2N, true, INV, 0, NSUB>(data, tw);
  if (INV)
    for (int w = 0; w < N * T; ++w) radix3_item<T, true>(tile3, tile3, roots.data(), w);
  double worst = 0;
  for (int c = 0; c < T; c += (T > 4 ? T / 4 : 1))
    for (int o = 0; o < M1; ++o) {
      // forward output: frequency side, row o = 3 e + q;  inverse output: time side, row o = q N + e
      const cd got = tile3(INV ? o / N : o % 3, c, INV ? o % N : o / 3);
      worst = std::fmax(worst, naive_err(plain, size_t(c) * M1, M1, o, INV, got));
    }
  return worst / std::sqrt(double(M1));
}

template <int LOG2N> static int run3() {
  const double f = check3<LOG2N, false>(), b = check3<LOG2N, true>();
  const int bad = !(f < 1e-14) + !(b < 1e-14);
  std::printf("3N=%5d cols fwd %.2e inv %.2e %s\n", 3 << LOG2N, f, b, bad ? "FAIL" : "ok");
  return bad;
}

template <int LOG2N, bool COLS, bool INV> static double check() {
  constexpr int N = 1 << LOG2N, T = kPoints / N;
  std::vector<cd> tw;
  make_stage_tw(LOG2N, tw);
  std::vector<cd> data(kPoints);
  std::vector<cd> plain(kPoints);   // plain[t*N + e]
  for (int t = 0; t < T; ++t)
    for (int e = 0; e < N; ++e) {
      cd v = mk(drand48() - 0.5, drand48() - 0.5);
      plain[t * N + e] = v;
      data[lds_addr<LOG2N, COLS>(t, e)] = v;
    }
  emu_from<LOG2N, COLS, INV, 0>(data, tw);
  double worst = 0;
  const int tstep = T > 8 ? T / 8 : 1;
  for (int t = 0; t < T; t += tstep)
    for (int k = 0; k < N; ++k) worst = std::fmax(worst, naive_err(plain, size_t(t) * N, N, k, INV, data[lds_addr<LOG2N, COLS>(t, k)]));
  return worst / std::sqrt(double(N));
}

template <int LOG2N> static int run() {
  const double e[4] = {check<LOG2N, false, false>(), check<LOG2N, false, true>(), check<LOG2N, true, false>(),
                       check<LOG2N, true, true>()};
  int bad = 0;
  for (int i = 0; i < 4; ++i) bad += !(e[i] < 1e-14);
  std::printf("N=%5d rows fwd %.2e inv %.2e | cols fwd %.2e inv %.2e | tw entries %d first radix %d %s\n", 1 << LOG2N, e[0], e[1],
              e[2], e[3], stage_tw_size(LOG2N), stage_radix(LOG2N, 0), bad ? "FAIL" : "ok");
  return bad;
}

// ---- the in-LDS circular convolution of the prime-factor row pass (pfa_kernels.h, k_pfa_rows steps 2-4), two
// tiles of M points: forward stages, then the fused seam (last forward stage, pointwise product, first inverse
// stage on the same sixteen registers, lane mapping as in the kernel), then the remaining inverse stages
template <int LM> static int run_conv() {
  constexpr int M = 1 << LM, NB = M / 16, LANES = 2 * M / 16;
  constexpr bool CT = LM >= 11;
  constexpr int LPL = stage_tw_last(LM), RL = stage_radix(LM, LPL), P = 1 << LPL, SPLIT = 16 / RL;
  std::vector<cd> tw;
  if (CT) make_stage_tw_compact(LM, tw); else make_stage_tw(LM, tw);
  std::vector<cd> data(2 * M), x(2 * M), hh(M);
  const LdsTile<LM, false, 2> tile{data.data()};
  for (int t = 0; t < 2; ++t)
    for (int e = 0; e < M; ++e) { x[t * M + e] = mk(drand48() - 0.5, drand48() - 0.5); tile(t, e, x[t * M + e]); }
  for (int e = 0; e < M; ++e) hh[e] = mk(drand48() - 0.5, drand48() - 0.5);
  emu_from<LM, false, false, 0, 2, CT, LPL>(data, tw);                       // forward, all stages but the last
  std::vector<cd> regs(size_t(LANES) * 16);
  for (int tid = 0; tid < LANES; ++tid) {                                    // seam: every read ...
    const int t = tid / NB, i = tid % NB;
    cd* u = &regs[size_t(tid) * 16];
    for (int q = 0; q < SPLIT; ++q) {
      cd v[RL];
      const int k = i + NB * q;
      stage_load<LM, false, false, LPL, 2, CT>(tile, tw.data(), t * P + k, v);
      for (int r = 0; r < RL; ++r) u[q + SPLIT * r] = cmul(v[r], hh[k + P * r]);
    }
    dft16<true>(u);
  }
  for (int tid = 0; tid < LANES; ++tid) stage_store<LM, false, 0, 2>(tile, tid, &regs[size_t(tid) * 16]);   // ... before any write
  emu_from<LM, false, true, 4, 2, CT>(data, tw);                             // inverse, the stages after the first
  // reference: y = IDFT(DFT(x) . hh), unnormalised like the kernel
  double worst = 0;
  for (int t = 0; t < 2; ++t) {
    std::vector<cd> X(M);
    for (int k = 0; k < M; ++k) {
      long double sx = 0, sy = 0;
      for (int e = 0; e < M; ++e) {
        const long double a = -2.0L * M_PIl * (long double)((long long)k * e % M) / M;
        sx += x[t * M + e].x * cosl(a) - x[t * M + e].y * sinl(a);
        sy += x[t * M + e].x * sinl(a) + x[t * M + e].y * cosl(a);
      }
      X[k] = cmul(mk(double(sx), double(sy)), hh[k]);
    }
    for (int o = 0; o < M; o += 37) worst = std::fmax(worst, naive_err(X, 0, M, o, true, tile(t, o)));
  }
  worst /= double(M);
  const int bad = !(worst < 1e-13);
  std::printf("M=%5d in-LDS convolution (compact twiddles %d, seam %d x radix-%d) err %.2e %s\n", M, int(CT), SPLIT, RL, worst,
              bad ? "FAIL" : "ok");
  return bad;
}

// ---- prime-factor stages of the Rader row pass (mixed_radix.h): the 990-point cyclic convolution as a 3-D convolution
// in the residues mod 9, 10, 11 - forward DFTs along the three axes, product with the 3-D spectrum of the kernel,
// inverse DFTs - against the direct O(L^2) sum
template <int R, bool INV, class AX> static void emu_axis(std::vector<cd>& data, int which) {
  const PlainTile tile{data.data(), AX::L};
  const int nb = AX::L / R;
  for (int j = 0; j < nb; ++j) {
    const int base = which == 1 ? AX::base1(j) : (which == 2 ? AX::base2(j) : AX::base3(j));
    const int stride = which == 1 ? AX::kStride1 : (which == 2 ? AX::kStride2 : AX::kStride3);
    cd v[R];
    axis_load<R>(tile, 0, base, stride, v);
    dft_sym<R, INV>(v);
    axis_store<R>(tile, 0, base, stride, v);
  }
}

static int run_axes() {
  using AX = Axes<11, 9, 10>;
  constexpr int L = AX::L;
  std::vector<cd> a(L), b(L), A(L), B(L);
  for (int s = 0; s < L; ++s) { a[s] = mk(drand48() - 0.5, drand48() - 0.5); b[s] = mk(drand48() - 0.5, drand48() - 0.5); }
  int seen = 0;
  std::vector<int> hit(L, 0);
  for (int s = 0; s < L; ++s) { A[AX::pos(s)] = a[s]; B[AX::pos(s)] = b[s]; seen += !hit[AX::pos(s)]++; }
  emu_axis<11, false, AX>(A, 1); emu_axis<9, false, AX>(A, 2); emu_axis<10, false, AX>(A, 3);
  emu_axis<11, false, AX>(B, 1); emu_axis<9, false, AX>(B, 2); emu_axis<10, false, AX>(B, 3);
  for (int p = 0; p < L; ++p) A[p] = cmul(A[p], B[p]);
  emu_axis<10, true, AX>(A, 3); emu_axis<9, true, AX>(A, 2); emu_axis<11, true, AX>(A, 1);
  double worst = 0;
  for (int s = 0; s < L; s += 3) {
    long double cx = 0, cy = 0;
    for (int q = 0; q < L; ++q) {
      const cd x = a[q], y = b[(s - q + L) % L];
      cx += (long double)x.x * y.x - (long double)x.y * y.y;
      cy += (long double)x.x * y.y + (long double)x.y * y.x;
    }
    const cd got = A[AX::pos(s)];
    worst = std::fmax(worst, std::hypot(got.x / L - double(cx), got.y / L - double(cy)));
  }
  const int bad = !(seen == L && worst < 1e-13);
  std::printf("L=  990 cyclic convolution by prime-factor stages 11 x 9 x 10 (positions distinct: %d) err %.2e %s\n", seen, worst,
              bad ? "FAIL" : "ok");
  return bad;
}

int main() {
  srand48(12345);
  int bad = 0;
  bad += run_axes();
  bad += run_conv<10>();
  bad += run_conv<11>();
  bad += run_conv<12>();
  bad += run<4>();
  bad += run<5>();
  bad += run<6>();
  bad += run<7>();
  bad += run<8>();
  bad += run<9>();
  bad += run<10>();
  bad += run<11>();
  bad += run3<4>();
  bad += run3<5>();
  bad += run3<6>();
  bad += run3<7>();
  bad += run3<8>();
  std::printf(bad ? "FAILED\n" : "ALL OK\n");
  return bad ? 1 : 0;
}
