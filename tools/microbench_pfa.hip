// Ablation timings of the prime-factor kernels on the metric geometry (n = 88199 = 89 x 991, G transforms).
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -fno-signed-zeros -I pyaudiolocalization_amd/csrc -I include \
//         tools/microbench_pfa.hip -o tools/microbench_pfa
// Random data (the timing does not depend on values); not part of the product or the tests.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "pfa_kernels.h"
#include "pfa_rader.h"

using namespace pal;

namespace pal {   // the table generators live in bluestein.hip; the microbench only needs valid memory
__global__ void k_make_chirp(cd*, int, int) {}
__global__ void k_make_roots(cd*, int, double) {}
__global__ void k_make_stage_tw(cd*, int, bool) {}
}

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// ---- column pass variants: MODE 0 product shape, 1 no table loads, 2 no Y loads; UNR = j-steps per load batch
template <int TC, int MODE, int UNR>
__global__ __launch_bounds__(256) void cols_var(const cd* __restrict__ Y, double* __restrict__ corr, size_t stride,
                                                int N1, int N2, int G, int nch, const double* __restrict__ T) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(int(threadIdx.x) >> 6);
  const int role = wave & 1, ch = int(blockIdx.y) * 2 + (wave >> 1);
  if (ch >= nch) return;
  const int g = blockIdx.x % G, cb = blockIdx.x / G;
  const int m2 = cb * 64 + lane;
  const bool live = m2 < N2;
  const cd* Yg = Y + size_t(g) * N1 * N2 + (live ? m2 : N2 - 1);
  const int h = (N1 - 1) / 2;
  double accC[TC], accS[TC];
#pragma unroll
  for (int tt = 0; tt < TC; ++tt) accC[tt] = accS[tt] = 0.0;
  double sumE = 0.0;
  const double* Tj = T + size_t(ch) * 2 * TC;
  const size_t tstep = size_t(nch) * 2 * TC;
  for (int j = 1; j + UNR - 1 <= h; j += UNR) {
    cd yj[UNR], ym[UNR];
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      if (MODE == 2) { yj[u] = mk(double(j + u), 1.0); ym[u] = mk(0.5, double(lane)); }
      else { yj[u] = Yg[size_t(j + u) * N2]; ym[u] = Yg[size_t(N1 - j - u) * N2]; }
    }
#pragma unroll
    for (int u = 0; u < UNR; ++u, Tj += tstep) {
      const double a = role ? yj[u].y + ym[u].y : yj[u].x + ym[u].x;
      const double b = role ? yj[u].x - ym[u].x : yj[u].y - ym[u].y;
      sumE += a;
#pragma unroll
      for (int tt = 0; tt < TC; ++tt) {
        const double c = MODE == 1 ? 0.25 * (tt + 1) : Tj[tt], s = MODE == 1 ? 0.125 * (tt + 2) : Tj[TC + tt];
        accC[tt] = __builtin_fma(c, a, accC[tt]);
        accS[tt] = __builtin_fma(s, b, accS[tt]);
      }
    }
  }
  const cd y0 = Yg[0];
  const double base = role ? y0.y : y0.x;
  if (!live) return;
  double* out = corr + size_t(2 * g + role) * stride + m2;
  if (ch == 0) out[0] = base + sumE;
#pragma unroll
  for (int tt = 0; tt < TC; ++tt) {
    const int t = ch * TC + tt + 1;
    if (t <= h) {
      const double s = role ? accS[tt] : -accS[tt];
      out[size_t(N2) * t] = base + accC[tt] + s;
      out[size_t(N2) * (N1 - t)] = base + accC[tt] - s;
    }
  }
}

// ---- column pass, register version with a software pipeline: loads run D iterations ahead of the FMAs
template <int TC, int D>
__global__ __launch_bounds__(256) void cols_pipe(const cd* __restrict__ Y, double* __restrict__ corr, size_t stride,
                                                 int N1, int N2, int G, int nch, const double* __restrict__ T) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(int(threadIdx.x) >> 6);
  const int role = wave & 1, ch = int(blockIdx.y) * 2 + (wave >> 1);
  if (ch >= nch) return;
  const int g = blockIdx.x % G, cb = blockIdx.x / G;
  const int m2 = cb * 64 + lane;
  const bool live = m2 < N2;
  const cd* Yg = Y + size_t(g) * N1 * N2 + (live ? m2 : N2 - 1);
  const int h = (N1 - 1) / 2;
  double accC[TC], accS[TC];
#pragma unroll
  for (int tt = 0; tt < TC; ++tt) accC[tt] = accS[tt] = 0.0;
  double sumE = 0.0;
  const double* Tj = T + size_t(ch) * 2 * TC;
  const size_t tstep = size_t(nch) * 2 * TC;
  constexpr int RING = D + 1;
  cd yj[RING], ym[RING];
#pragma unroll
  for (int u = 0; u < D; ++u) {
    const int j = 1 + u <= h ? 1 + u : h;
    yj[u] = Yg[size_t(j) * N2]; ym[u] = Yg[size_t(N1 - j) * N2];
  }
  for (int j0 = 1; j0 <= h; j0 += RING) {
#pragma unroll
    for (int u = 0; u < RING; ++u) {
      const int j = j0 + u;
      const int jn = j + D <= h ? j + D : h;
      yj[(u + D) % RING] = Yg[size_t(jn) * N2];
      ym[(u + D) % RING] = Yg[size_t(N1 - jn) * N2];
      if (j <= h) {
        const double a = role ? yj[u].y + ym[u].y : yj[u].x + ym[u].x;
        const double b = role ? yj[u].x - ym[u].x : yj[u].y - ym[u].y;
        sumE += a;
#pragma unroll
        for (int tt = 0; tt < TC; ++tt) {
          accC[tt] = __builtin_fma(Tj[tt], a, accC[tt]);
          accS[tt] = __builtin_fma(Tj[TC + tt], b, accS[tt]);
        }
        Tj += tstep;
      }
    }
  }
  const cd y0 = Yg[0];
  const double base = role ? y0.y : y0.x;
  if (!live) return;
  double* out = corr + size_t(2 * g + role) * stride + m2;
  if (ch == 0) out[0] = base + sumE;
#pragma unroll
  for (int tt = 0; tt < TC; ++tt) {
    const int t = ch * TC + tt + 1;
    if (t <= h) {
      const double s = role ? accS[tt] : -accS[tt];
      out[size_t(N2) * t] = base + accC[tt] + s;
      out[size_t(N2) * (N1 - t)] = base + accC[tt] - s;
    }
  }
}

// ---- column pass, both pairs in one wave: TC output indices per wave, half the table traffic per FMA
template <int TC, int UNR>
__global__ __launch_bounds__(256) void cols_merged(const cd* __restrict__ Y, double* __restrict__ corr, size_t stride,
                                                   int N1, int N2, int G, int nch, const double* __restrict__ T) {
  const int lane = threadIdx.x & 63;
  const int ch = int(blockIdx.y) * 4 + __builtin_amdgcn_readfirstlane(int(threadIdx.x) >> 6);
  if (ch >= nch) return;
  const int g = blockIdx.x % G, cb = blockIdx.x / G;
  const int m2 = cb * 64 + lane;
  const bool live = m2 < N2;
  const cd* Yg = Y + size_t(g) * N1 * N2 + (live ? m2 : N2 - 1);
  const int h = (N1 - 1) / 2;
  double cx[TC], sy[TC], cy[TC], sx[TC];
#pragma unroll
  for (int tt = 0; tt < TC; ++tt) cx[tt] = sy[tt] = cy[tt] = sx[tt] = 0.0;
  double sumx = 0.0, sumy = 0.0;
  const double* Tj = T + size_t(ch) * 2 * TC;
  const size_t tstep = size_t(nch) * 2 * TC;
  cd yj[UNR], ym[UNR];
#pragma unroll
  for (int u = 0; u < UNR; ++u) { yj[u] = Yg[size_t(1 + u) * N2]; ym[u] = Yg[size_t(N1 - 1 - u) * N2]; }
  for (int j = 1; j <= h; j += UNR) {
    cd nj[UNR], nm[UNR];
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const int jn = j + UNR + u <= h ? j + UNR + u : h;
      nj[u] = Yg[size_t(jn) * N2]; nm[u] = Yg[size_t(N1 - jn) * N2];
    }
#pragma unroll
    for (int u = 0; u < UNR; ++u, Tj += tstep) {
      const double ap = yj[u].x + ym[u].x, bp = yj[u].y - ym[u].y, aq = yj[u].y + ym[u].y, bq = yj[u].x - ym[u].x;
      sumx += ap; sumy += aq;
#pragma unroll
      for (int tt = 0; tt < TC; ++tt) {
        const double c = Tj[tt], s = Tj[TC + tt];
        cx[tt] = __builtin_fma(c, ap, cx[tt]);
        sy[tt] = __builtin_fma(s, bp, sy[tt]);
        cy[tt] = __builtin_fma(c, aq, cy[tt]);
        sx[tt] = __builtin_fma(s, bq, sx[tt]);
      }
    }
#pragma unroll
    for (int u = 0; u < UNR; ++u) { yj[u] = nj[u]; ym[u] = nm[u]; }
  }
  const cd y0 = Yg[0];
  if (!live) return;
  double* outp = corr + size_t(2 * g) * stride + m2;
  double* outq = outp + stride;
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

// ---- column pass traffic only: every load of a wave in flight at once, the same stores, no arithmetic
template <int MODE>   // 0: product layout Y[k1][N2];  1: blocked layout Y[cb][k1][64] (contiguous per workgroup)
__global__ __launch_bounds__(256) void cols_copy(const cd* __restrict__ Y, double* __restrict__ corr, size_t stride,
                                                 int N1, int N2, int G) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(int(threadIdx.x) >> 6);
  const int role = wave & 1, ch = wave >> 1;
  const int g = blockIdx.x % G, cb = blockIdx.x / G;
  const int m2 = cb * 64 + lane;
  const bool live = m2 < N2;
  const cd* Yg = MODE == 0 ? Y + size_t(g) * N1 * N2 + (live ? m2 : N2 - 1) : Y + size_t(g) * N1 * N2 + size_t(cb) * N1 * 64 + lane;
  const size_t rs = MODE == 0 ? size_t(N2) : 64;
  double acc[2] = {0, 0};
  // each wave reads every row once (like the product: four waves share the rows through L1)
  for (int j0 = 0; j0 < 88; j0 += 22) {
    cd v[22];
#pragma unroll
    for (int u = 0; u < 22; ++u) v[u] = Yg[size_t(j0 + u) * rs];
#pragma unroll
    for (int u = 0; u < 22; ++u) { acc[0] += v[u].x; acc[1] += v[u].y; }
  }
  if (!live) return;
  double* out = corr + size_t(2 * g + role) * stride + m2;
  for (int tt = 0; tt < 22; ++tt) {
    const int t = ch * 22 + tt + 1;
    out[size_t(N2) * t] = acc[0];
    out[size_t(N2) * (N1 - t)] = acc[1];
  }
}

// ---- row pass variants
template <int LM> struct ConstIn {
  static constexpr bool kLds = false;
  __device__ cd operator()(int t, int e) const { return e >= (1 << (LM - 1)) ? mk(0, 0) : mk(double(e), double(t)); }
};
template <int LM> struct ConstHhat {
  static constexpr bool kLds = true;
  cd* data;
  __device__ void operator()(int t, int e, cd v) const { data[lds_addr<LM, false, 2>(t, e)] = cmul(v, mk(0.5, 0.25)); }
};
template <int LM> struct Sink {
  static constexpr bool kLds = false;
  cd* base;
  __device__ void operator()(int t, int e, cd v) const { if (e < (1 << (LM - 1)) && v.x == 123.456) base[0] = v; }
};

template <int LM, int MODE>   // 1 = compute only, 2 = memory only
__global__ __launch_bounds__(PfaLds<LM>::kLanes) void rows_var(PfaRowsArgs a) {
  using L = PfaLds<LM>;
  __shared__ cd data[2 * L::kM];
  __shared__ cd tw[L::kTw];
  const int tid = threadIdx.x;
  const int g = blockIdx.x % a.G, k1 = blockIdx.x / a.G;
  for (int i = tid; i < L::kTw; i += L::kLanes) tw[i] = a.tws[i];
  if (MODE == 1) {
    wg_fft<LM, false, false, 2, L::kCompact>(data, tw, tid, ConstIn<LM>{}, ConstHhat<LM>{data});
    wg_fft<LM, false, true, 2, L::kCompact>(data, tw, tid, LdsTile<LM, false, 2>{data}, Sink<LM>{a.Y});
  } else {
    const int4 q = a.quad[g];
    const size_t mic = size_t(a.NR) * a.N2, off = size_t(k1) * a.N2;
    const cd* sa = a.SP + size_t(q.x) * mic + off;
    const cd* sb = a.SP + size_t(q.y) * mic + off;
    const cd* sc = a.SP + size_t(q.z) * mic + off;
    const cd* sd = a.SP + size_t(q.w) * mic + off;
    const bool share = q.z == q.x;
    const int t = tid / (L::kLanes / 2), i = tid % (L::kLanes / 2);
    cd* Yg = a.Y + size_t(g) * a.N1 * a.N2;
    const int row = t == 0 ? k1 : (k1 ? a.N1 - k1 : 0);
    for (int r = 0; r < 8; ++r) {
      const int e = i + r * (L::kLanes / 2);
      const int ee = e < a.N2 ? e : a.N2 - 1;
      const cd va = sa[ee], vb = sb[ee], vd = sd[ee];
      const cd vc = share ? va : sc[ee];
      cd z = mk(va.x + vb.x + vc.x + vd.x, va.y + vb.y + vc.y + vd.y);
      z = cmul(z, a.b[ee]);
      z = cmul(z, a.hhat[e]);
      z = cmul(z, a.hhat[e + (1 << (LM - 1))]);
      if (e < a.N2) Yg[size_t(row) * a.N2 + (t ? (ee ? a.N2 - ee : 0) : ee)] = z;
    }
  }
}

template <class F> static float time_it(const char* name, int reps, F launch) {
  hipEvent_t a, b;
  CHECK(hipEventCreate(&a));
  CHECK(hipEventCreate(&b));
  launch();
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(a));
  for (int i = 0; i < reps; ++i) launch();
  CHECK(hipEventRecord(b));
  CHECK(hipEventSynchronize(b));
  float ms = 0;
  CHECK(hipEventElapsedTime(&ms, a, b));
  CHECK(hipGetLastError());
  printf("  %-44s %8.1f us\n", name, ms * 1000.0f / reps);
  return ms / reps;
}

int main(int argc, char** argv) {
  const int G = argc > 1 ? atoi(argv[1]) : 128;
  const int N1 = 89, N2 = 991, NR = 45, mics = 64, n = N1 * N2;
  constexpr int LM = 11;
  const size_t stride = size_t(n) + 1;
  cd *SP, *Y, *b, *hhat, *r1, *tws;
  double *corr, *T;
  int4* quad;
  CHECK(hipMalloc(&SP, sizeof(cd) * mics * NR * N2));
  CHECK(hipMalloc(&Y, sizeof(cd) * size_t(G) * (n + 128)));
  CHECK(hipMalloc(&b, sizeof(cd) * N2));
  CHECK(hipMalloc(&hhat, sizeof(cd) << LM));
  CHECK(hipMalloc(&r1, sizeof(cd) * N1));
  CHECK(hipMalloc(&tws, sizeof(cd) << LM));
  CHECK(hipMalloc(&corr, sizeof(double) * 2 * G * (stride + 128)));
  const int h = 44, nch = 2;
  CHECK(hipMalloc(&T, sizeof(double) * h * 4 * 2 * 22));
  CHECK(hipMalloc(&quad, sizeof(int4) * G));
  std::vector<double> rnd(size_t(mics) * NR * N2 * 2);
  srand(1);
  for (auto& v : rnd) v = rand() / double(RAND_MAX) - 0.5;
  CHECK(hipMemcpy(SP, rnd.data(), rnd.size() * 8, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(b, rnd.data(), sizeof(cd) * N2, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(hhat, rnd.data(), sizeof(cd) << LM, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(r1, rnd.data(), sizeof(cd) * N1, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(tws, rnd.data(), sizeof(cd) << LM, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(T, rnd.data(), sizeof(double) * h * 4 * 2 * 22, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(Y, rnd.data(), sizeof(cd) * size_t(G < 32 ? G : 32) * n, hipMemcpyHostToDevice));
  std::vector<int4> q(G);
  for (int g = 0; g < G; ++g) {   // consecutive i<j pairs: (i, 2g+1), (i, 2g+2) share the first mic most of the time
    const int i = g % 20;
    q[g] = make_int4(i, 21 + (2 * g) % 40, i, 21 + (2 * g + 1) % 40);
  }
  CHECK(hipMemcpy(quad, q.data(), sizeof(int4) * G, hipMemcpyHostToDevice));
  int2* rowtab;
  CHECK(hipMalloc(&rowtab, sizeof(int2) * N1));
  {
    std::vector<int2> rt(N1);
    for (int r = 0; r < N1; ++r) rt[r] = make_int2((68 * r) % N1, (68 * r % N1) * 256 % N1);
    CHECK(hipMemcpy(rowtab, rt.data(), sizeof(int2) * N1, hipMemcpyHostToDevice));
  }
  PfaRowsArgs a{SP, quad, Y, b, hhat, r1, tws, tws, rowtab, N1, N2, NR, G, 68, 1.0f / float(N1), nullptr};
  const unsigned grid = unsigned(G) * NR, nblk = (N2 + 63) / 64;
  printf("G = %d transforms (%d pairs)\n", G, 2 * G);
  time_it("rows: product", 20, [&] { k_pfa_rows<LM><<<dim3(grid), dim3(PfaLds<LM>::kLanes)>>>(a); });
  time_it("rows: product, twiddles from global", 20, [&] { k_pfa_rows<LM, true><<<dim3(grid), dim3(PfaLds<LM>::kLanes)>>>(a); });
  time_it("rows: compute only", 20, [&] { rows_var<LM, 1><<<dim3(grid), dim3(PfaLds<LM>::kLanes)>>>(a); });
  time_it("rows: memory only", 20, [&] { rows_var<LM, 2><<<dim3(grid), dim3(PfaLds<LM>::kLanes)>>>(a); });
  {   // phase clocks of every workgroup of one product launch
    unsigned long long* st;
    CHECK(hipMalloc(&st, sizeof(unsigned long long) * 8 * grid));
    PfaRowsArgs as = a;
    as.stamps = st;
    k_pfa_rows<LM><<<dim3(grid), dim3(PfaLds<LM>::kLanes)>>>(as);
    CHECK(hipDeviceSynchronize());
    std::vector<unsigned long long> hs(size_t(8) * grid);
    CHECK(hipMemcpy(hs.data(), st, hs.size() * 8, hipMemcpyDeviceToHost));
    const char* names[5] = {"loads + whiten + stage 1", "forward middle", "seam", "inverse rest + store", "total"};
    for (int ph = 0; ph < 5; ++ph) {
      std::vector<double> d(grid);
      for (unsigned w = 0; w < grid; ++w)
        d[w] = ph < 4 ? double(hs[size_t(w) * 8 + ph + 1] - hs[size_t(w) * 8 + ph]) / 100.0 : double(hs[size_t(w) * 8 + 4] - hs[size_t(w) * 8]) / 100.0;
      std::sort(d.begin(), d.end());
      printf("  rows phase %-28s median %6.2f us  p90 %6.2f us\n", names[ph], d[grid / 2], d[grid * 9 / 10]);
    }
    unsigned long long t0 = ~0ull, t1 = 0;
    for (unsigned w = 0; w < grid; ++w) { t0 = std::min(t0, hs[size_t(w) * 8]); t1 = std::max(t1, hs[size_t(w) * 8 + 4]); }
    printf("  rows launch span %.1f us\n", double(t1 - t0) / 100.0);
  }
  PfaRaderArgs ra_keep{};
  {   // Rader row pass: random but valid tables (timing only)
    constexpr int LR = 990;
    int *qidx, *ridx;
    cd *bhat, *t2f, *t2i, *t3f, *t3i;
    CHECK(hipMalloc(&qidx, sizeof(int) * N2));
    CHECK(hipMalloc(&ridx, sizeof(int) * N2));
    std::vector<int> perm(N2);
    for (int e = 0; e < N2; ++e) perm[e] = e ? (e * 7) % LR : 0;
    CHECK(hipMemcpy(qidx, perm.data(), sizeof(int) * N2, hipMemcpyHostToDevice));
    for (int e = 0; e < N2; ++e) perm[e] = e ? (e * 13) % LR : 0;
    CHECK(hipMemcpy(ridx, perm.data(), sizeof(int) * N2, hipMemcpyHostToDevice));
    CHECK(hipMalloc(&bhat, sizeof(cd) * 1024)); CHECK(hipMalloc(&t2f, sizeof(cd) * 1024)); CHECK(hipMalloc(&t2i, sizeof(cd) * 1024));
    CHECK(hipMalloc(&t3f, sizeof(cd) * 1024)); CHECK(hipMalloc(&t3i, sizeof(cd) * 1024));
    for (cd* p : {bhat, t2f, t2i, t3f, t3i}) CHECK(hipMemcpy(p, rnd.data(), sizeof(cd) * 1024, hipMemcpyHostToDevice));
    unsigned long long* st;
    CHECK(hipMalloc(&st, sizeof(unsigned long long) * 8 * grid));
    PfaRaderArgs ra{SP, quad, Y, bhat, r1, ridx, rowtab, N1, N2, NR, G, 1.0f / float(N1), 1.0 / double(n), nullptr};
    time_it("rows (Rader 11 x 9 x 10): product", 20, [&] { k_pfa_rows_rader<11, 9, 10><<<dim3(grid), dim3(256)>>>(ra); });
    ra_keep = ra;
    ra.stamps = st;
    k_pfa_rows_rader<11, 9, 10><<<dim3(grid), dim3(256)>>>(ra);
    CHECK(hipDeviceSynchronize());
    std::vector<unsigned long long> hs(size_t(8) * grid);
    CHECK(hipMemcpy(hs.data(), st, hs.size() * 8, hipMemcpyDeviceToHost));
    const char* names[4] = {"loads + whiten + radix 11", "radix 9, seam, inverse 9 / 11", "epilogue", "total"};
    for (int ph = 0; ph < 4; ++ph) {
      std::vector<double> d(grid);
      for (unsigned w = 0; w < grid; ++w)
        d[w] = ph < 3 ? double(hs[size_t(w) * 8 + ph + 1] - hs[size_t(w) * 8 + ph]) / 100.0 : double(hs[size_t(w) * 8 + 3] - hs[size_t(w) * 8]) / 100.0;
      std::sort(d.begin(), d.end());
      printf("  rader phase %-26s median %6.2f us  p90 %6.2f us\n", names[ph], d[grid / 2], d[grid * 9 / 10]);
    }
  }
  const dim3 cg(unsigned(G) * nblk, 1);
  time_it("cols: product", 20, [&] { k_pfa_cols<kPfaTC, kPfaUnr><<<cg, dim3(256)>>>(Y, corr, stride, N1, N2, G, 4, T, nullptr); });
  time_it("cols: product with 3 steps per batch", 20, [&] { k_pfa_cols<kPfaTC, 3><<<cg, dim3(256)>>>(Y, corr, stride, N1, N2, G, 4, T, nullptr); });
  time_it("cols: product with 2 steps per batch", 20, [&] { k_pfa_cols<kPfaTC, 2><<<cg, dim3(256)>>>(Y, corr, stride, N1, N2, G, 4, T, nullptr); });
  time_it("cols: merged pairs, 11 t per wave, 1 step ahead", 20, [&] { cols_merged<11, 1><<<cg, dim3(256)>>>(Y, corr, stride, N1, N2, G, 4, T); });
  time_it("cols: merged pairs, 11 t per wave, 2 steps/iter", 20, [&] { cols_merged<11, 2><<<cg, dim3(256)>>>(Y, corr, stride, N1, N2, G, 4, T); });
  time_it("cols: merged pairs, 11 t per wave, 4 steps/iter", 20, [&] { cols_merged<11, 4><<<cg, dim3(256)>>>(Y, corr, stride, N1, N2, G, 4, T); });
  time_it("cols: merged pairs, 11 t per wave, 8 steps/iter", 20, [&] { cols_merged<11, 8><<<cg, dim3(256)>>>(Y, corr, stride, N1, N2, G, 4, T); });
  time_it("cols: merged pairs, 22 t per wave, 2 steps/iter", 20, [&] { cols_merged<22, 2><<<dim3(unsigned(G) * nblk, 1), dim3(128)>>>(Y, corr, stride, N1, N2, G, 2, T); });
  time_it("cols: merged pairs, 22 t per wave, 4 steps/iter", 20, [&] { cols_merged<22, 4><<<dim3(unsigned(G) * nblk, 1), dim3(128)>>>(Y, corr, stride, N1, N2, G, 2, T); });
  time_it("cols: merged pairs, 6 t per wave, 4 steps/iter", 20, [&] { cols_merged<6, 4><<<dim3(unsigned(G) * nblk, 2), dim3(256)>>>(Y, corr, stride, N1, N2, G, 8, T); });
  time_it("cols: traffic only, product layout", 20, [&] { cols_copy<0><<<cg, dim3(256)>>>(Y, corr, stride, N1, N2, G); });
  time_it("cols: traffic only, blocked Y layout", 20, [&] { cols_copy<1><<<cg, dim3(256)>>>(Y, corr, stride, N1, N2, G); });
  time_it("cols: register pipeline, 1 ahead", 20, [&] { cols_pipe<22, 1><<<cg, dim3(256)>>>(Y, corr, stride, N1, N2, G, nch, T); });
  time_it("cols: register pipeline, 2 ahead", 20, [&] { cols_pipe<22, 2><<<cg, dim3(256)>>>(Y, corr, stride, N1, N2, G, nch, T); });
  time_it("cols: register pipeline, 3 ahead", 20, [&] { cols_pipe<22, 3><<<cg, dim3(256)>>>(Y, corr, stride, N1, N2, G, nch, T); });
  time_it("cols: register pipeline, 5 ahead", 20, [&] { cols_pipe<22, 5><<<cg, dim3(256)>>>(Y, corr, stride, N1, N2, G, nch, T); });
  time_it("cols: variant unroll 1", 20, [&] { cols_var<22, 0, 1><<<cg, dim3(256)>>>(Y, corr, stride, N1, N2, G, nch, T); });
  time_it("cols: variant unroll 2", 20, [&] { cols_var<22, 0, 2><<<cg, dim3(256)>>>(Y, corr, stride, N1, N2, G, nch, T); });
  time_it("cols: variant unroll 4", 20, [&] { cols_var<22, 0, 4><<<cg, dim3(256)>>>(Y, corr, stride, N1, N2, G, nch, T); });
  time_it("cols: no table loads, unroll 1", 20, [&] { cols_var<22, 1, 1><<<cg, dim3(256)>>>(Y, corr, stride, N1, N2, G, nch, T); });
  time_it("cols: no table loads, unroll 4", 20, [&] { cols_var<22, 1, 4><<<cg, dim3(256)>>>(Y, corr, stride, N1, N2, G, nch, T); });
  time_it("cols: no Y loads, N2 = 992", 20, [&] { cols_var<22, 2, 1><<<cg, dim3(256)>>>(Y, corr, stride, N1, 992, G, nch, T); });
  time_it("cols: no Y loads, unroll 1", 20, [&] { cols_var<22, 2, 1><<<cg, dim3(256)>>>(Y, corr, stride, N1, N2, G, nch, T); });
  {   // do the row pass (VALU / LDS) and the column pass (memory) overlap when launched on two streams?
    hipStream_t sa, sb;
    CHECK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking));
    CHECK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
    cd* Y2;
    CHECK(hipMalloc(&Y2, sizeof(cd) * size_t(G) * (n + 128)));
    PfaRowsArgs a2 = a;
    a2.Y = Y2;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    double* corr2;
    CHECK(hipMalloc(&corr2, sizeof(double) * 2 * G * (stride + 128)));
    for (int mode = 3; mode < 7; ++mode) {
      CHECK(hipDeviceSynchronize());
      auto t0 = std::chrono::steady_clock::now();
      const int reps = 20;
      for (int i = 0; i < reps; ++i) {
        if (mode == 3 || mode == 4) {
          k_pfa_cols<kPfaTC, kPfaUnr><<<cg, dim3(256), 0, sa>>>(Y, corr, stride, N1, N2, G, 4, T, nullptr);
          k_pfa_cols<kPfaTC, kPfaUnr><<<cg, dim3(256), 0, mode == 4 ? sb : sa>>>(Y2, corr2, stride, N1, N2, G, 4, T, nullptr);
        } else {
          k_pfa_rows<LM><<<dim3(grid), dim3(PfaLds<LM>::kLanes), 0, sa>>>(a);
          k_pfa_rows<LM><<<dim3(grid), dim3(PfaLds<LM>::kLanes), 0, mode == 6 ? sb : sa>>>(a2);
        }
      }
      CHECK(hipDeviceSynchronize());
      const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / reps;
      const char* nm[4] = {"cols, cols on one stream", "cols || cols on two streams", "rows, rows on one stream", "rows || rows on two streams"};
      printf("  %-44s %8.1f us per iteration (wall)\n", nm[mode - 3], us);
    }
    for (int pad : {0, 18, 45})     // Rader rows with dynamic LDS padding (4 / 3 / 2 workgroups per CU) beside the column pass
      for (int mode = 0; mode < 3; ++mode) {
        if (mode == 1 && pad) continue;
        PfaRaderArgs r2 = ra_keep;
        r2.Y = Y2;
        CHECK(hipDeviceSynchronize());
        auto t0 = std::chrono::steady_clock::now();
        const int reps = 20;
        for (int i = 0; i < reps; ++i) {
          if (mode != 1) k_pfa_rows_rader<11, 9, 10><<<dim3(grid), dim3(256), size_t(pad) * 1024, sa>>>(r2);
          if (mode != 0) k_pfa_cols<kPfaTC, kPfaUnr><<<cg, dim3(256), 0, mode == 2 ? sb : sa>>>(Y, corr, stride, N1, N2, G, 4, T, nullptr);
        }
        CHECK(hipDeviceSynchronize());
        const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / reps;
        printf("  rader pad %2d KB: %-30s %8.1f us per iteration (wall)\n", pad, mode == 0 ? "rows alone" : mode == 1 ? "cols alone" : "rows || cols on two streams", us);
      }
    for (int mode = 0; mode < 3; ++mode) {
      CHECK(hipDeviceSynchronize());
      auto t0 = std::chrono::steady_clock::now();
      const int reps = 20;
      for (int i = 0; i < reps; ++i) {
        if (mode != 1) k_pfa_rows<LM><<<dim3(grid), dim3(PfaLds<LM>::kLanes), 0, sa>>>(a2);
        if (mode != 0) k_pfa_cols<kPfaTC, kPfaUnr><<<cg, dim3(256), 0, mode == 2 ? sb : sa>>>(Y, corr, stride, N1, N2, G, 4, T, nullptr);
      }
      CHECK(hipDeviceSynchronize());
      const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / reps;
      printf("  %-44s %8.1f us per iteration (wall)\n", mode == 0 ? "rows alone" : mode == 1 ? "cols alone" : "rows || cols on two streams", us);
    }
  }
  return 0;
}
