// Host execution of the workgroup FFT stage code of csrc/fft_core.h, lane by lane, against a
// naive O(N^2) DFT.  Built and run by tests/test_host_fft_core.py (no GPU needed):
//   hipcc -O2 -I pyaudiolocalization_amd/csrc tests/host/test_fft_core.cpp -o /tmp/test_fft_core
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "fft_core.h"

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

// every stage through the LDS-tile accessor, reads of a stage before its writes (what the barriers enforce)
template <int LOG2N, bool COLS, bool INV, int LOG2P, int NSUB = (kPoints >> LOG2N)>
static void emu_from(std::vector<cd>& data, const std::vector<cd>& tw) {
  if constexpr (LOG2P < LOG2N) {
    constexpr int R = stage_radix(LOG2N, LOG2P);
    constexpr int ITEMS = (NSUB << LOG2N) / R;
    const LdsTile<LOG2N, COLS, NSUB> tile{data.data()};
    std::vector<cd> regs(size_t(ITEMS) * R);
    for (int w = 0; w < ITEMS; ++w) stage_load<LOG2N, COLS, INV, LOG2P, NSUB>(tile, tw.data(), w, &regs[size_t(w) * R]);
    for (int w = 0; w < ITEMS; ++w) stage_store<LOG2N, COLS, LOG2P, NSUB>(tile, w, &regs[size_t(w) * R]);
    emu_from<LOG2N, COLS, INV, LOG2P + stage_log2r(LOG2N, LOG2P), NSUB>(data, tw);
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

int main() {
  srand48(12345);
  int bad = 0;
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
