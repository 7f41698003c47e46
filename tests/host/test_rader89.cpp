// Host execution of the stage arithmetic of csrc/pfa_rader89.h (the 89-point column DFT by Rader's algorithm over
// 8 x 11), with the two LDS exchanges between the four wavefronts emulated by plain arrays, against a naive DFT.
// Built and run by tests/test_host_fft_core.py (no GPU needed).
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "pfa_rader89.h"

using namespace pal;

int main() {
  Rader89Tab tab;
  make_rader89_tab(tab);
  // every input row is read exactly once, every output index written exactly once
  std::vector<int> seen_in(89, 0), seen_out(89, 0);
  for (int w = 0; w < 4; ++w)
    for (int i = 0; i < 22; ++i) { ++seen_in[tab.rowsel[w][i]]; ++seen_out[tab.tmap[w][i]]; }
  for (int j = 1; j < 89; ++j)
    if (seen_in[j] != 1 || seen_out[j] != 1) { printf("index map broken at %d\n", j); return 1; }
  double worst = 0;
  for (int trial = 0; trial < 20; ++trial) {
    std::vector<cd> Y(89);
    srand(17 + trial);
    for (auto& y : Y) y = mk(rand() / double(RAND_MAX) - 0.5, rand() / double(RAND_MAX) - 0.5);
    if (trial == 1) for (auto& y : Y) y = mk(1.0, 0.0);
    if (trial == 2) { for (auto& y : Y) y = mk(0, 0); Y[5] = mk(0.0, 2.0); }
    // stage A per wavefront
    cd v[4][22];
    for (int w = 0; w < 4; ++w) {
      for (int i = 0; i < 22; ++i) v[w][i] = Y[tab.rowsel[w][i]];
      r89_stage_a(v[w], v[w] + 11, w);
    }
    // exchange 1 + stage B + exchange 2
    cd z[4][22];
    cd c0 = mk(0, 0);
    for (int w = 0; w < 4; ++w)
      for (int q = 0; q < 3; ++q) {
        const int k11 = 3 * w + q;
        if (k11 > 10) continue;
        cd e4[4], o4[4];
        for (int ws = 0; ws < 4; ++ws) { e4[ws] = v[ws][k11]; o4[ws] = v[ws][11 + k11]; }
        cd c0q = mk(0, 0);
        r89_stage_b(e4, o4, tab.H[w][q], w == 0 && q == 0, Y[0], c0q);
        if (w == 0 && q == 0) c0 = c0q;
        for (int ws = 0; ws < 4; ++ws) { z[ws][k11] = e4[ws]; z[ws][11 + k11] = o4[ws]; }
      }
    std::vector<cd> got(89);
    got[0] = c0;
    for (int w = 0; w < 4; ++w) {
      r89_stage_c(z[w], z[w] + 11, w);
      for (int i = 0; i < 22; ++i) got[tab.tmap[w][i]] = z[w][i];
    }
    for (int t = 0; t < 89; ++t) {
      long double re = 0, im = 0;
      for (int j = 0; j < 89; ++j) {
        const long double a = 6.283185307179586476925286766559005768L * (long double)((j * t) % 89) / 89.0L;
        re += (long double)Y[j].x * cosl(a) - (long double)Y[j].y * sinl(a);
        im += (long double)Y[j].x * sinl(a) + (long double)Y[j].y * cosl(a);
      }
      const double err = std::hypot(got[t].x - double(re), got[t].y - double(im));
      if (err > worst) worst = err;
    }
  }
  printf("rader89: worst error %.3e\n", worst);
  if (worst > 2e-14) { printf("FAILED\n"); return 1; }
  printf("ALL OK\n");
  return 0;
}
