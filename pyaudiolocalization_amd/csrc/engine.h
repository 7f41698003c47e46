// engine.h - internal C++ state behind the pal_handle of include/pal_hip.h.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <map>
#include <string>
#include <tuple>
#include <vector>

#include "../../include/pal_hip.h"
#include "fft_core.h"

namespace pal {

// One circular convolution of length M = M1 x M2 (four-step, both factors in LDS).
// M2 = 2^l2; M1 = 2^l1, or 3 * 2^l1 (r3) when the 3 * 2^k length is the smaller fit.
struct Conv {
  int l1 = 0, l2 = 0;
  bool r3 = false;
  size_t m = 0;                  // points per transform
  cd* chat = nullptr;            // FFT_M of the chirp kernel in [k1][k2] order, pre-scaled
  cd* twA = nullptr;             // exp(-2 pi i q / M1), q < M1
  cd* twB = nullptr;             // exp(-2 pi i r / M),  r < M2
  size_t M() const { return m; }
  int M1() const { return (r3 ? 3 : 1) << l1; }
  int M2() const { return 1 << l2; }
};

// Exact-length-n DFT plan (Bluestein / chirp-z): forward real -> half spectrum, inverse pairs.
struct Plan {
  int n = 0;        // DFT length
  int H = 0;        // n/2 + 1 bins of the half spectrum
  int lin = 0;      // longest real input of a forward transform
  int nout = 0;     // outputs kept by an inverse transform (n for PHAT, N of 2N for fractional delay)
  cd* w = nullptr;  // chirp exp(i pi j^2 / n), j < n
  Conv fwd;         // forward:  lin inputs -> H outputs
  Conv inv;         // inverse:  n inputs  -> n outputs (two real sequences per complex transform)
};

struct ProfileSlot {
  double ms = 0;
  int64_t launches = 0;
};

struct Engine {
  int device = 0;
  hipStream_t stream = nullptr;    // FFT passes, copies, everything a caller can order against
  hipStream_t stream2 = nullptr;   // peak selection of launch group g while the passes of g+1 run on `stream`
  hipEvent_t ev_corr[2] = {}, ev_peaks[2] = {};   // hand-offs of the two correlation buffers between the streams
  bool overlap = true;             // alternate launch groups between the two streams (PAL_OVERLAP=0 turns it off)
  bool allow_r3 = true;            // PAL_RADIX3=0 forces power-of-two convolution lengths
  std::string err;
  int chunk = 128;                              // transforms per launch group (256 PHAT rows per peak-kernel launch: one per CU)
  std::map<std::tuple<int, int, int>, Plan> plans;   // (n, lin, nout) -> plan
  cd* stage_tw[12] = {};                        // stage-major twiddles per log2 N
  // growable device scratch
  void* ws[12] = {};
  size_t ws_bytes[12] = {};
  // profiling
  bool profiling = false;
  std::vector<hipEvent_t> ev_pool;
  size_t ev_used = 0;
  struct Pending { int slot; hipEvent_t a, b; };
  std::vector<Pending> pending;
  std::vector<std::string> slot_names;
  std::vector<ProfileSlot> slots;
  // rccl
  void* comm = nullptr;
  // device copy of the (trial, i<j) pair table of the last all-pairs shape
  void* quads = nullptr;
  int quad_B = 0, quad_M = 0;

  int fail(int code, const char* fmt, ...);
  int check(hipError_t e, const char* what);
  int scratch(int idx, size_t bytes, void** out);
  const cd* stage_table(int ln);
  int get_plan(int n, int lin, int nout, Plan** out);
  int alloc_conv(Conv& c, size_t needed);   // geometry, tables and chirp-spectrum storage for >= `needed` points
  void free_conv(Conv& c);
  int build_conv(Conv& c, const cd* w, int n, int neg_count, int pos_count, bool conj_kernel, double extra_scale);
  // profiling helpers
  int prof_slot(const char* name);
  void prof_begin(int slot, hipEvent_t* a, hipStream_t on);
  void prof_end(int slot, hipEvent_t a, hipStream_t on);
  void prof_flush();

  // pipelines (all pointers are device pointers)
  int forward_spectra(Plan& pl, const double* frames, size_t frame_stride, int rows, int len, cd* spectra);
  int pair_correlations(Plan& pl, const cd* spectra, const int4* quads, int64_t npairs, int n2,
                        const pal_phat_params& prm, pal_pair_record* table, int32_t* ksel_multi, double* corr_out);
  int peaks(const double* corr, size_t stride, int rows, int n, int n2, const pal_phat_params& prm,
            pal_pair_record* table, int32_t* ksel_multi, hipStream_t on);
};

struct ProfScope {
  Engine* e;
  int slot;
  hipEvent_t a = nullptr;
  hipStream_t on;
  ProfScope(Engine* eng, const char* name, hipStream_t s = nullptr) : e(eng), slot(-1), on(s ? s : eng->stream) {
    if (e->profiling) {
      slot = e->prof_slot(name);
      e->prof_begin(slot, &a, on);
    }
  }
  ~ProfScope() {
    if (slot >= 0) e->prof_end(slot, a, on);
  }
};

inline int ceil_log2(size_t v) {
  int l = 0;
  while ((size_t(1) << l) < v) ++l;
  return l;
}

#define PAL_TRY(expr)                \
  do {                               \
    int _rc = (expr);                \
    if (_rc != PAL_OK) return _rc;   \
  } while (0)

#define PAL_HIP(expr) PAL_TRY(check((expr), #expr))

}  // namespace pal
