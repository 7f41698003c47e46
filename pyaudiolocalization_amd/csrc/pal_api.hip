// pal_api.hip - extern "C" surface of libpal_hip.so (include/pal_hip.h) and the engine's
// house-keeping: errors, scratch, HIP-event profiling, the RCCL gather.
#include <dlfcn.h>

#include <cmath>
#include <cstring>

#include "engine.h"

namespace pal {

static std::string g_create_error;

int Engine::fail(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  err = buf;
  return code;
}

int Engine::check(hipError_t e, const char* what) {
  if (e == hipSuccess) return PAL_OK;
  const int code = e == hipErrorOutOfMemory ? PAL_ERR_NOMEM : PAL_ERR_HIP;
  return fail(code, "%s: %s", what, hipGetErrorString(e));
}

int Engine::scratch(int idx, size_t bytes, void** out) {
  if (ws_bytes[idx] < bytes) {
    if (ws[idx]) {                       // either stream may still be using the old block
      PAL_HIP(hipStreamSynchronize(stream));
      PAL_HIP(hipStreamSynchronize(stream2));
      PAL_HIP(hipStreamSynchronize(stream3));
      PAL_HIP(hipFree(ws[idx]));
      ws[idx] = nullptr;
      ws_bytes[idx] = 0;
    }
    PAL_HIP(hipMalloc(&ws[idx], bytes));
    PAL_HIP(hipMemsetAsync(ws[idx], 0, bytes, stream));
    PAL_HIP(hipStreamSynchronize(stream));   // the block may first be used on the other stream: zeroes must have landed
    ws_bytes[idx] = bytes;
  }
  *out = ws[idx];
  return PAL_OK;
}

// ---- profiling: one event pair per launch, resolved after the stream drains ----
int Engine::prof_slot(const char* name) {
  for (size_t i = 0; i < slot_names.size(); ++i)
    if (slot_names[i] == name) return int(i);
  slot_names.emplace_back(name);
  slots.emplace_back();
  return int(slot_names.size() - 1);
}

static hipEvent_t take_event(Engine* e) {
  if (e->ev_used == e->ev_pool.size()) {
    hipEvent_t ev;
    if (hipEventCreate(&ev) != hipSuccess) return nullptr;
    e->ev_pool.push_back(ev);
  }
  return e->ev_pool[e->ev_used++];
}

void Engine::prof_begin(int, hipEvent_t* a, hipStream_t on) {
  if (pending.size() >= 16384) {
    hipStreamSynchronize(stream);
    hipStreamSynchronize(stream2);
    hipStreamSynchronize(stream3);
    prof_flush();
  }
  *a = take_event(this);
  if (*a) hipEventRecord(*a, on);
}

void Engine::prof_end(int slot, hipEvent_t a, hipStream_t on) {
  hipEvent_t b = take_event(this);
  if (!a || !b) return;
  hipEventRecord(b, on);
  pending.push_back({slot, a, b});
}

void Engine::prof_flush() {
  for (const Pending& p : pending) {
    float ms = 0;
    if (hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) {
      slots[p.slot].ms += ms;
      slots[p.slot].launches += 1;
    }
  }
  pending.clear();
  ev_used = 0;
}

// ---- status word written by k_peaks on an internal overflow ----
static int check_status(Engine* e) {
  if (!e->ws[7]) return PAL_OK;
  int st = 0;
  int rc = e->check(hipMemcpy(&st, e->ws[7], sizeof st, hipMemcpyDeviceToHost), "status read");
  if (rc != PAL_OK) return rc;
  if (getenv("PAL_DEBUG_FALLBACK")) {   // diagnostics: rows whose median needed the radix select since the last report
    int slow = 0, exact = 0;
    if (hipMemcpy(&slow, static_cast<int*>(e->ws[7]) + 1, sizeof slow, hipMemcpyDeviceToHost) == hipSuccess && slow) {
      fprintf(stderr, "[pal] %d row(s) took the radix-select fallback\n", slow);
      hipMemset(static_cast<int*>(e->ws[7]) + 1, 0, sizeof slow);
    }
    if (hipMemcpy(&exact, static_cast<int*>(e->ws[7]) + 3, sizeof exact, hipMemcpyDeviceToHost) == hipSuccess && exact) {
      fprintf(stderr, "[pal] %d row(s) needed the exact median (a threshold comparison inside the histogram interval)\n", exact);
      hipMemset(static_cast<int*>(e->ws[7]) + 3, 0, sizeof exact);
    }
    int flagged = 0;
    if (hipMemcpy(&flagged, static_cast<int*>(e->ws[7]) + 4, sizeof flagged, hipMemcpyDeviceToHost) == hipSuccess && flagged) {
      int why[8] = {};
      (void)hipMemcpy(why, static_cast<int*>(e->ws[7]) + 5, sizeof why, hipMemcpyDeviceToHost);
      int gaveup = 0;
      (void)hipMemcpy(&gaveup, static_cast<int*>(e->ws[7]) + 13, sizeof gaveup, hipMemcpyDeviceToHost);
      fprintf(stderr, "[pal] %d row(s) of the finishing column pass went through the stored-row path (no maximum / abandoned %d, tie %d, tie in window %d, "
                      "SNR window energy %d, histogram windows %d, threshold interval %d, window interval %d, window edge %d; waits given up %d)\n",
              flagged, why[0], why[1], why[2], why[3], why[4], why[5], why[6], why[7], gaveup);
      hipMemset(static_cast<int*>(e->ws[7]) + 4, 0, sizeof flagged + sizeof why);
      hipMemset(static_cast<int*>(e->ws[7]) + 13, 0, sizeof gaveup);
    }
  }
  int in = 0;                                // word 2: input problems found by device-side checks
  rc = e->check(hipMemcpy(&in, static_cast<int*>(e->ws[7]) + 2, sizeof in, hipMemcpyDeviceToHost), "status read");
  if (rc != PAL_OK) return rc;
  if (in) {
    hipMemset(static_cast<int*>(e->ws[7]) + 2, 0, sizeof in);
    if (in & 2) return e->fail(PAL_ERR_INVALID, "pair list references a row outside the frame batch");
    return e->fail(PAL_ERR_INVALID, "non-finite sample (NaN or infinity) in a frame: the pair packed with that microphone's "
                                    "pairs would be affected too, the table of this call is not valid");
  }
  if (st) {
    hipMemset(e->ws[7], 0, sizeof st);
    return e->fail(PAL_ERR_INTERNAL, "peak selection: suppression chain exceeded the on-chip memo/stack (rows fell back to argmax)");
  }
  return PAL_OK;
}

static int validate(Engine* e, const pal_phat_params* p) {
  if (!p) return e->fail(PAL_ERR_INVALID, "params is NULL");
  if (!(p->fs > 0)) return e->fail(PAL_ERR_INVALID, "fs must be positive");
  if (p->peak_distance < 1) return e->fail(PAL_ERR_INVALID, "`distance` must be greater or equal to 1");
  if (p->num_peaks < 1 || p->num_peaks > PAL_MAX_PEAKS)
    return e->fail(PAL_ERR_INVALID, "num_peaks %d outside 1..%d", p->num_peaks, PAL_MAX_PEAKS);
  if (p->threshold_method != 0 && p->threshold_method != 1) return e->fail(PAL_ERR_INVALID, "threshold_method must be 0 or 1");
  return PAL_OK;
}

// row-major i<j pairs of every trial, two pairs per complex transform
static void build_quads(int B, int M, std::vector<int4>& q) {
  const int64_t P = int64_t(M) * (M - 1) / 2, np = P * B;
  q.assign(size_t((np + 1) / 2), make_int4(0, 0, -1, -1));
  int64_t k = 0;
  for (int b = 0; b < B; ++b)
    for (int i = 0; i < M; ++i)
      for (int j = i + 1; j < M; ++j, ++k) {
        int4& t = q[size_t(k / 2)];
        if (k & 1) { t.z = b * M + i; t.w = b * M + j; }
        else { t.x = b * M + i; t.y = b * M + j; }
      }
}

// The same pairs in blocks of `blk` x `blk` microphones (block rows I <= J; inside a block i ascending, j ascending): a launch group
// of 480 pairs then touches some 64 spectra instead of 480 (C4: 256 microphones - every pair of one row i brings its own spectrum j,
// 3 MB each).  perm[k] = row-major index of the pair processed k-th; the records are scattered back behind the call.
static void build_quads_blocked(int B, int M, int blk, std::vector<int4>& q, std::vector<int>& perm) {
  const int64_t P = int64_t(M) * (M - 1) / 2, np = P * B;
  q.assign(size_t((np + 1) / 2), make_int4(0, 0, -1, -1));
  perm.resize(size_t(np));
  int64_t k = 0;
  for (int b = 0; b < B; ++b)
    for (int I = 0; I < M; I += blk)
      for (int J = I; J < M; J += blk)
        for (int i = I; i < I + blk && i < M; ++i)
          for (int j = (J > i ? J : i + 1); j < J + blk && j < M; ++j, ++k) {
            int4& t = q[size_t(k / 2)];
            if (k & 1) { t.z = b * M + i; t.w = b * M + j; }
            else { t.x = b * M + i; t.y = b * M + j; }
            perm[size_t(k)] = int(int64_t(b) * P + int64_t(i) * M - int64_t(i) * (i + 1) / 2 + (j - i - 1));
          }
}

__global__ __launch_bounds__(256) void k_scatter_records(const pal_pair_record* __restrict__ src, const int* __restrict__ perm, int64_t n,
                                                         pal_pair_record* __restrict__ dst) {
  const int64_t k = int64_t(blockIdx.x) * 256 + threadIdx.x;
  if (k < n) dst[perm[k]] = src[k];
}

static int all_pairs_dev(Engine* e, const double* d_frames, int B, int M, int L, const pal_phat_params* prm,
                         pal_pair_record* d_table, double* d_corr) {
  PAL_TRY(validate(e, prm));
  if (B < 1 || M < 2 || L < 1) return e->fail(PAL_ERR_INVALID, "need B >= 1, M >= 2, L >= 1 (got %d, %d, %d)", B, M, L);
  if (int64_t(B) * M > INT32_MAX / 2) return e->fail(PAL_ERR_UNSUPPORTED, "too many rows");
  if (L > (1 << 20)) return e->fail(PAL_ERR_UNSUPPORTED, "frame length %d exceeds 2^20", L);
  Plan* pl = nullptr;
  PAL_TRY(e->get_plan(2 * L - 1, L, 2 * L - 1, &pl));
  const int rows = B * M;
  void* sp = nullptr;
  PAL_TRY(e->scratch(2, size_t(rows) * pl->spec_stride() * sizeof(cd) + size_t(rows) * sizeof(int), &sp));
  cd* spectra = static_cast<cd*>(sp);
  int* nonzero = reinterpret_cast<int*>(spectra + size_t(rows) * pl->spec_stride());   // per frame row: any non-zero sample
  PAL_TRY(e->forward_spectra(*pl, d_frames, size_t(L), rows, L, spectra, nonzero));
  // the pair table depends on (B, M) only: keep it on the device between calls of the same shape
  const size_t nquads = size_t((int64_t(B) * M * (M - 1) / 2 + 1) / 2);
  if (e->pair_block > 1 && M >= 96 && !d_corr && int64_t(B) * M * (M - 1) / 2 < INT32_MAX) {
    const int64_t np = int64_t(B) * M * (M - 1) / 2;
    if (e->blk_B != B || e->blk_M != M || !e->quads_blk) {
      std::vector<int4> quads;
      std::vector<int> perm;
      build_quads_blocked(B, M, e->pair_block, quads, perm);
      PAL_TRY(e->check(hipStreamSynchronize(e->stream), "stream sync"));
      if (e->quads_blk) { (void)hipFree(e->quads_blk); e->quads_blk = nullptr; }
      if (e->perm_blk) { (void)hipFree(e->perm_blk); e->perm_blk = nullptr; }
      PAL_TRY(e->check(hipMalloc(&e->quads_blk, nquads * sizeof(int4)), "quads alloc"));
      PAL_TRY(e->check(hipMalloc(&e->perm_blk, size_t(np) * sizeof(int)), "pair order alloc"));
      PAL_TRY(e->check(hipMemcpyAsync(e->quads_blk, quads.data(), nquads * sizeof(int4), hipMemcpyHostToDevice, e->stream), "quads"));
      PAL_TRY(e->check(hipMemcpyAsync(e->perm_blk, perm.data(), size_t(np) * sizeof(int), hipMemcpyHostToDevice, e->stream), "pair order"));
      PAL_TRY(e->check(hipStreamSynchronize(e->stream), "quads sync"));
      e->blk_B = B;
      e->blk_M = M;
    }
    void* tp = nullptr;
    PAL_TRY(e->scratch(23, size_t(np) * sizeof(pal_pair_record), &tp));
    PAL_TRY(e->pair_correlations(*pl, spectra, rows, e->quads_blk, np, L, *prm, static_cast<pal_pair_record*>(tp), nullptr, nullptr, nonzero));
    k_scatter_records<<<dim3(unsigned((np + 255) / 256)), dim3(256), 0, e->stream>>>(static_cast<const pal_pair_record*>(tp), e->perm_blk, np, d_table);
    return e->check(hipGetLastError(), "k_scatter_records");
  }
  if (e->quad_B != B || e->quad_M != M || !e->quads) {
    std::vector<int4> quads;
    build_quads(B, M, quads);
    if (e->quads) { PAL_TRY(e->check(hipStreamSynchronize(e->stream), "stream sync")); (void)hipFree(e->quads); e->quads = nullptr; }
    PAL_TRY(e->check(hipMalloc(&e->quads, nquads * sizeof(int4)), "quads alloc"));
    PAL_TRY(e->check(hipMemcpyAsync(e->quads, quads.data(), nquads * sizeof(int4), hipMemcpyHostToDevice, e->stream), "quads"));
    PAL_TRY(e->check(hipStreamSynchronize(e->stream), "quads sync"));
    e->quad_B = B;
    e->quad_M = M;
  }
  void* qp = e->quads;
  const int64_t np = int64_t(B) * M * (M - 1) / 2;
  return e->pair_correlations(*pl, spectra, rows, static_cast<const int4*>(qp), np, L, *prm, d_table, nullptr, d_corr, nonzero);
}

// explicit pair list over R equal-length rows (host buffers): upload, then the device-resident form
static int pairs_host(Engine* e, const double* rows_in, int R, int L, const int32_t* pairs, int64_t P,
                      const pal_phat_params* prm, pal_pair_record* table) {
  PAL_TRY(validate(e, prm));
  if (!rows_in || !pairs || !table) return e->fail(PAL_ERR_INVALID, "NULL buffer");
  if (R < 1 || L < 1 || P < 1) return e->fail(PAL_ERR_INVALID, "need R >= 1, L >= 1, P >= 1");
  for (int64_t k = 0; k < 2 * P; ++k)
    if (pairs[k] < 0 || pairs[k] >= R) return e->fail(PAL_ERR_INVALID, "pair %lld references row outside 0..%d", (long long)(k / 2), R - 1);
  void *df = nullptr, *dt = nullptr, *dp = nullptr;
  PAL_TRY(e->scratch(4, size_t(R) * L * sizeof(double), &df));
  PAL_TRY(e->scratch(5, size_t(P) * sizeof(pal_pair_record), &dt));
  PAL_TRY(e->scratch(15, size_t(2 * P) * sizeof(int32_t), &dp));
  PAL_TRY(e->check(hipMemcpyAsync(df, rows_in, size_t(R) * L * sizeof(double), hipMemcpyHostToDevice, e->stream), "rows upload"));
  PAL_TRY(e->check(hipMemcpyAsync(dp, pairs, size_t(2 * P) * sizeof(int32_t), hipMemcpyHostToDevice, e->stream), "pairs upload"));
  PAL_TRY(e->check(hipStreamSynchronize(e->stream), "upload sync"));
  PAL_TRY(e->pairs_dev(static_cast<const double*>(df), R, L, static_cast<const int32_t*>(dp), P, *prm, static_cast<pal_pair_record*>(dt)));
  return e->check(hipMemcpyAsync(table, dt, size_t(P) * sizeof(pal_pair_record), hipMemcpyDeviceToHost, e->stream), "table download");
}

}  // namespace pal

using namespace pal;

extern "C" {

int pal_abi_version(void) { return PAL_ABI_VERSION; }

int pal_create(int device, pal_handle* out) {
  if (!out) return PAL_ERR_INVALID;
  *out = nullptr;
  int count = 0;
  hipError_t rc = hipGetDeviceCount(&count);
  if (rc != hipSuccess || device < 0 || device >= count) {
    g_create_error = rc != hipSuccess ? std::string("hipGetDeviceCount: ") + hipGetErrorString(rc)
                                      : "device ordinal out of range (" + std::to_string(count) + " HIP devices visible)";
    return PAL_ERR_HIP;
  }
  Engine* e = new Engine();
  e->device = device;
  if ((rc = hipSetDevice(device)) != hipSuccess || (rc = hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking)) != hipSuccess ||
      (rc = hipStreamCreateWithFlags(&e->stream2, hipStreamNonBlocking)) != hipSuccess ||
      (rc = hipStreamCreateWithFlags(&e->stream3, hipStreamNonBlocking)) != hipSuccess ||
      (rc = hipEventCreateWithFlags(&e->ev_join3, hipEventDisableTiming)) != hipSuccess ||
      (rc = hipEventCreateWithFlags(&e->ev_fin, hipEventDisableTiming)) != hipSuccess ||
      (rc = hipEventCreateWithFlags(&e->ev_corr[0], hipEventDisableTiming)) != hipSuccess ||
      (rc = hipEventCreateWithFlags(&e->ev_corr[1], hipEventDisableTiming)) != hipSuccess ||
      (rc = hipEventCreateWithFlags(&e->ev_peaks[0], hipEventDisableTiming)) != hipSuccess ||
      (rc = hipEventCreateWithFlags(&e->ev_peaks[1], hipEventDisableTiming)) != hipSuccess) {
    g_create_error = std::string("device init: ") + hipGetErrorString(rc);
    delete e;
    return PAL_ERR_HIP;
  }
  {
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && cus > 0) e->cu_count = cus;
  }
  const char* env = getenv("PAL_CHUNK");
  if (env && atoi(env) > 0) { e->chunk = atoi(env); e->chunk_auto = false; }
  env = getenv("PAL_OVERLAP");
  if (env) e->overlap = atoi(env);
  env = getenv("PAL_RADIX3");
  if (env) e->allow_r3 = atoi(env) != 0;
  env = getenv("PAL_PFA");
  if (env) e->allow_pfa = atoi(env) != 0;
  env = getenv("PAL_RADER");
  if (env) e->allow_rader = atoi(env) != 0;
  env = getenv("PAL_PFA_BIG");
  if (env) e->allow_big = atoi(env) != 0;
  env = getenv("PAL_FOUR_REG");
  if (env) e->four_reg = atoi(env);
  env = getenv("PAL_XCD_ROWS");
  if (env) e->xcd_rows = atoi(env) != 0;
  env = getenv("PAL_PFA_FWD");
  if (env) e->pfa_forward = atoi(env) != 0;
  env = getenv("PAL_FUSED");
  if (env) e->fuse_peaks = atoi(env) != 0;
  env = getenv("PAL_R89");
  if (env) e->allow_r89 = atoi(env) != 0;
  env = getenv("PAL_FIN_SERIAL");
  if (env) e->fin_serial = atoi(env) != 0;
  env = getenv("PAL_FIN");
  if (env) e->fin_cols = atoi(env) != 0;
  env = getenv("PAL_FIN_DENSE");
  if (env) e->fin_dense = atoi(env) != 0 ? 1 : 0;
  e->fin_strips = getenv("PAL_FIN_STRIPS") != nullptr;
  env = getenv("PAL_FIN_FOUR");
  e->fin_four = env && atoi(env) != 0;
  env = getenv("PAL_FIN_WIDE");
  e->fin_wide = env && atoi(env) != 0;
  e->fin_hist = getenv("PAL_FIN_HIST") != nullptr;
  env = getenv("PAL_ROWS_LEAN");
  if (env) e->rows_lean = atoi(env) != 0;
  env = getenv("PAL_PAIR_BLOCK");
  if (env) e->pair_block = atoi(env);
  env = getenv("PAL_ROWS_LEAN_MIN");
  if (env && atoll(env) >= 1) e->rows_lean_min = atoll(env);
  env = getenv("PAL_LEAN_STORE");
  if (env) e->lean_store = atoi(env) != 0;
  env = getenv("PAL_DEBUG_MEMO");
  if (env) e->debug_memo = atoi(env);
  env = getenv("PAL_PFA_SUB");
  if (env) e->pfa_sub = atoi(env);
  env = getenv("PAL_MAX_PLANS");
  if (env && atoi(env) >= 2) e->max_plans = atoi(env);
  *out = reinterpret_cast<pal_handle>(e);
  return PAL_OK;
}

void pal_destroy(pal_handle h) {
  if (!h) return;
  Engine* e = reinterpret_cast<Engine*>(h);
  hipSetDevice(e->device);
  pal_comm_destroy(h);
  (void)e->clear_plans();                     // drains the three streams first
  for (cd* p : e->stage_tw) if (p) hipFree(p);
  for (cd* p : e->stage_twc) if (p) hipFree(p);
  for (void* p : e->ws) if (p) hipFree(p);
  if (e->quads) hipFree(e->quads);
  if (e->quads_blk) hipFree(e->quads_blk);
  if (e->perm_blk) hipFree(e->perm_blk);
  for (hipEvent_t ev : e->ev_pool) hipEventDestroy(ev);
  for (int k = 0; k < 2; ++k) { hipEventDestroy(e->ev_corr[k]); hipEventDestroy(e->ev_peaks[k]); }
  hipStreamDestroy(e->stream2);
  hipStreamDestroy(e->stream3);
  hipEventDestroy(e->ev_join3);
  hipEventDestroy(e->ev_fin);
  hipStreamDestroy(e->stream);
  delete e;
}

const char* pal_last_error(pal_handle h) {
  if (!h) return g_create_error.c_str();
  return reinterpret_cast<Engine*>(h)->err.c_str();
}

#define ENGINE(h)                                   \
  if (!(h)) return PAL_ERR_INVALID;                 \
  Engine* e = reinterpret_cast<Engine*>(h);         \
  if (hipSetDevice(e->device) != hipSuccess) return e->fail(PAL_ERR_HIP, "hipSetDevice(%d) failed", e->device)

int pal_synchronize(pal_handle h) {
  ENGINE(h);
  PAL_TRY(e->check(hipStreamSynchronize(e->stream), "stream sync"));
  if (e->profiling) e->prof_flush();
  return check_status(e);
}

int pal_clear_plans(pal_handle h) {
  ENGINE(h);
  return e->clear_plans();
}

int pal_set_max_plans(pal_handle h, int max_plans) {
  ENGINE(h);
  if (max_plans < 2 || max_plans > 4096) return e->fail(PAL_ERR_INVALID, "max_plans %d outside 2..4096", max_plans);
  e->max_plans = max_plans;
  return PAL_OK;
}

int pal_plan_stats(pal_handle h, int64_t* built, int64_t* evicted) {
  ENGINE(h);
  if (built) *built = e->plans_built;
  if (evicted) *evicted = e->plans_evicted;
  return PAL_OK;
}

int pal_set_chunk(pal_handle h, int chunk) {
  ENGINE(h);
  if (chunk < 0 || chunk > 4096) return e->fail(PAL_ERR_INVALID, "chunk %d outside 0..4096", chunk);
  if (chunk > 0) { e->chunk = chunk; e->chunk_auto = false; }
  return PAL_OK;
}

int pal_device_alloc(pal_handle h, size_t bytes, void** dptr) {
  ENGINE(h);
  if (!dptr) return e->fail(PAL_ERR_INVALID, "dptr is NULL");
  return e->check(hipMalloc(dptr, bytes ? bytes : 1), "hipMalloc");
}

int pal_device_free(pal_handle h, void* dptr) {
  ENGINE(h);
  PAL_TRY(e->check(hipStreamSynchronize(e->stream), "stream sync"));
  PAL_TRY(e->check(hipStreamSynchronize(e->stream2), "stream sync"));
  PAL_TRY(e->check(hipStreamSynchronize(e->stream3), "stream sync"));
  return e->check(hipFree(dptr), "hipFree");
}

int pal_upload(pal_handle h, void* dptr, const void* host, size_t bytes) {
  ENGINE(h);
  PAL_TRY(e->check(hipMemcpyAsync(dptr, host, bytes, hipMemcpyHostToDevice, e->stream), "upload"));
  return e->check(hipStreamSynchronize(e->stream), "upload sync");
}

int pal_download(pal_handle h, void* host, const void* dptr, size_t bytes) {
  ENGINE(h);
  PAL_TRY(e->check(hipMemcpyAsync(host, dptr, bytes, hipMemcpyDeviceToHost, e->stream), "download"));
  return e->check(hipStreamSynchronize(e->stream), "download sync");
}

int pal_gcc_phat_all_pairs_dev(pal_handle h, const double* d_frames, int B, int M, int L, const pal_phat_params* prm,
                               pal_pair_record* d_table) {
  ENGINE(h);
  if (!d_frames || !d_table) return e->fail(PAL_ERR_INVALID, "NULL buffer");
  return all_pairs_dev(e, d_frames, B, M, L, prm, d_table, nullptr);
}

int pal_gcc_phat_all_pairs(pal_handle h, const double* frames, int B, int M, int L, const pal_phat_params* prm,
                           pal_pair_record* table, double* corr) {
  ENGINE(h);
  if (!frames || !table) return e->fail(PAL_ERR_INVALID, "NULL buffer");
  if (B < 1 || M < 2 || L < 1) return e->fail(PAL_ERR_INVALID, "need B >= 1, M >= 2, L >= 1 (got %d, %d, %d)", B, M, L);
  const size_t fbytes = size_t(B) * M * L * sizeof(double);
  const int64_t np = int64_t(B) * M * (M - 1) / 2;
  void *df = nullptr, *dt = nullptr, *dc = nullptr;
  PAL_TRY(e->scratch(4, fbytes, &df));
  PAL_TRY(e->scratch(5, size_t(np) * sizeof(pal_pair_record), &dt));
  if (corr) PAL_TRY(e->scratch(6, size_t(np) * size_t(2 * L - 1) * sizeof(double), &dc));
  PAL_TRY(e->check(hipMemcpyAsync(df, frames, fbytes, hipMemcpyHostToDevice, e->stream), "frames upload"));
  PAL_TRY(all_pairs_dev(e, static_cast<const double*>(df), B, M, L, prm, static_cast<pal_pair_record*>(dt),
                        static_cast<double*>(dc)));
  PAL_TRY(e->check(hipMemcpyAsync(table, dt, size_t(np) * sizeof(pal_pair_record), hipMemcpyDeviceToHost, e->stream), "table download"));
  if (corr) PAL_TRY(e->check(hipMemcpyAsync(corr, dc, size_t(np) * size_t(2 * L - 1) * sizeof(double), hipMemcpyDeviceToHost, e->stream), "corr download"));
  return pal_synchronize(h);
}

int pal_gcc_phat_pairs(pal_handle h, const double* rows, int R, int L, const int32_t* pairs, int64_t P,
                       const pal_phat_params* prm, pal_pair_record* table) {
  ENGINE(h);
  PAL_TRY(pairs_host(e, rows, R, L, pairs, P, prm, table));
  return pal_synchronize(h);
}

int pal_gcc_phat_pairs_dev(pal_handle h, const double* d_rows, int R, int L, const int32_t* d_pairs, int64_t P,
                           const pal_phat_params* prm, pal_pair_record* d_table) {
  ENGINE(h);
  PAL_TRY(validate(e, prm));
  if (!d_rows || !d_pairs || !d_table) return e->fail(PAL_ERR_INVALID, "NULL buffer");
  return e->pairs_dev(d_rows, R, L, d_pairs, P, *prm, d_table);
}

static int single_pair(Engine* e, const double* sig1, int n1, const double* sig2, int n2, const pal_phat_params* prm,
                       int32_t* k_out, pal_pair_record* rec, double* corr) {
  if (!sig1 || !sig2) return e->fail(PAL_ERR_INVALID, "NULL signal");
  if (n1 < 1 || n2 < 1) return e->fail(PAL_ERR_INVALID, "empty signal");
  if (n1 > (1 << 20) || n2 > (1 << 20)) return e->fail(PAL_ERR_UNSUPPORTED, "signal longer than 2^20 samples");
  const int n = n1 + n2 - 1, lin = n1 > n2 ? n1 : n2;
  Plan* pl = nullptr;
  PAL_TRY(e->get_plan(n, lin, n, &pl));
  void *df = nullptr, *sp = nullptr, *dc = nullptr, *dt = nullptr, *qp = nullptr, *dk = nullptr;
  PAL_TRY(e->scratch(4, size_t(2) * lin * sizeof(double), &df));
  PAL_TRY(e->scratch(2, size_t(2) * pl->spec_stride() * sizeof(cd), &sp));
  PAL_TRY(e->scratch(6, size_t(n) * sizeof(double), &dc));
  PAL_TRY(e->scratch(5, sizeof(pal_pair_record) + PAL_MAX_PEAKS * sizeof(int32_t), &dt));
  PAL_TRY(e->scratch(3, sizeof(int4), &qp));
  dk = static_cast<char*>(dt) + sizeof(pal_pair_record);
  double* d = static_cast<double*>(df);
  PAL_TRY(e->check(hipMemcpyAsync(d, sig1, size_t(n1) * sizeof(double), hipMemcpyHostToDevice, e->stream), "sig1 upload"));
  PAL_TRY(e->check(hipMemcpyAsync(d + lin, sig2, size_t(n2) * sizeof(double), hipMemcpyHostToDevice, e->stream), "sig2 upload"));
  const int4 quad = make_int4(0, 1, -1, -1);
  PAL_TRY(e->check(hipMemcpyAsync(qp, &quad, sizeof quad, hipMemcpyHostToDevice, e->stream), "quad upload"));
  PAL_TRY(e->check(hipStreamSynchronize(e->stream), "upload sync"));
  cd* S = static_cast<cd*>(sp);
  PAL_TRY(e->forward_spectra(*pl, d, size_t(lin), 1, n1, S));
  PAL_TRY(e->forward_spectra(*pl, d + lin, size_t(lin), 1, n2, S + pl->spec_stride()));
  pal_phat_params dummy{};
  PAL_TRY(e->pair_correlations(*pl, S, 2, static_cast<const int4*>(qp), 1, n2, prm ? *prm : dummy,
                               prm ? static_cast<pal_pair_record*>(dt) : nullptr,
                               prm ? static_cast<int32_t*>(dk) : nullptr, static_cast<double*>(dc)));
  if (corr) PAL_TRY(e->check(hipMemcpyAsync(corr, dc, size_t(n) * sizeof(double), hipMemcpyDeviceToHost, e->stream), "corr download"));
  if (prm && rec) PAL_TRY(e->check(hipMemcpyAsync(rec, dt, sizeof(pal_pair_record), hipMemcpyDeviceToHost, e->stream), "record download"));
  if (prm && k_out) PAL_TRY(e->check(hipMemcpyAsync(k_out, dk, size_t(prm->num_peaks) * sizeof(int32_t), hipMemcpyDeviceToHost, e->stream), "k download"));
  return PAL_OK;
}

int pal_phat_correlation(pal_handle h, const double* sig1, int n1, const double* sig2, int n2, double* corr) {
  ENGINE(h);
  if (!corr) return e->fail(PAL_ERR_INVALID, "corr is NULL");
  PAL_TRY(single_pair(e, sig1, n1, sig2, n2, nullptr, nullptr, nullptr, corr));
  return pal_synchronize(h);
}

int pal_get_time_delays_phat(pal_handle h, const double* sig1, int n1, const double* sig2, int n2,
                             const pal_phat_params* prm, int32_t* k_out, pal_pair_record* rec, double* corr) {
  ENGINE(h);
  PAL_TRY(validate(e, prm));
  PAL_TRY(single_pair(e, sig1, n1, sig2, n2, prm, k_out, rec, corr));
  return pal_synchronize(h);
}

int pal_corr_metrics(pal_handle h, const double* corr, int n, pal_pair_record* rec) {
  ENGINE(h);
  if (!corr || !rec || n < 1) return e->fail(PAL_ERR_INVALID, "bad correlation row");
  void *dc = nullptr, *dt = nullptr;
  PAL_TRY(e->scratch(6, size_t(n) * sizeof(double), &dc));
  PAL_TRY(e->scratch(5, sizeof(pal_pair_record), &dt));
  PAL_TRY(e->check(hipMemcpyAsync(dc, corr, size_t(n) * sizeof(double), hipMemcpyHostToDevice, e->stream), "corr upload"));
  pal_phat_params p{};
  p.fs = 1; p.threshold_method = -1; p.peak_distance = 1; p.num_peaks = 1; p.max_expected_delay = NAN;
  PAL_TRY(e->peaks(static_cast<const double*>(dc), size_t(n), 1, n, 1, p, static_cast<pal_pair_record*>(dt), nullptr, e->stream));
  PAL_TRY(e->check(hipMemcpyAsync(rec, dt, sizeof(pal_pair_record), hipMemcpyDeviceToHost, e->stream), "record download"));
  return pal_synchronize(h);
}

int pal_plan_info(pal_handle h, int L, int32_t* n, int32_t* conv_len, int32_t* m1, int32_t* m2) {
  ENGINE(h);
  if (L < 1 || L > (1 << 20)) return e->fail(PAL_ERR_INVALID, "bad frame length");
  Plan* pl = nullptr;
  PAL_TRY(e->get_plan(2 * L - 1, L, 2 * L - 1, &pl));
  if (n) *n = pl->n;
  if (conv_len) *conv_len = int32_t(pl->inv.M());
  if (m1) *m1 = pl->inv.M1();
  if (m2) *m2 = pl->inv.M2();
  return PAL_OK;
}

int pal_pair_group_size(pal_handle h, int L, int32_t* transforms) {
  ENGINE(h);
  if (L < 1) return e->fail(PAL_ERR_INVALID, "frame length %d", L);
  if (transforms) *transforms = e->pair_group(2 * L - 1);
  return PAL_OK;
}

int pal_plan_factors(pal_handle h, int L, int32_t* n1, int32_t* n2, int32_t* tile_len) {
  ENGINE(h);
  if (L < 1 || L > (1 << 20)) return e->fail(PAL_ERR_INVALID, "bad frame length");
  Plan* pl = nullptr;
  PAL_TRY(e->get_plan(2 * L - 1, L, 2 * L - 1, &pl));
  if (n1) *n1 = pl->pfa.n1;
  if (n2) *n2 = pl->pfa.n2;
  if (tile_len) *tile_len = pl->pfa.on() ? (pl->pfa.rader ? pl->pfa.n2 - 1 : 1 << pl->pfa.lm) : 0;
  return PAL_OK;
}

// ---- profiling -----------------------------------------------------------------------------
int pal_profile_begin(pal_handle h) {
  ENGINE(h);
  PAL_TRY(e->check(hipStreamSynchronize(e->stream), "stream sync"));
  e->pending.clear();
  e->ev_used = 0;
  for (auto& s : e->slots) s = ProfileSlot();
  e->profiling = true;
  return PAL_OK;
}

int pal_profile_sampling(pal_handle h, int every) {
  ENGINE(h);
  if (every < 1) return e->fail(PAL_ERR_INVALID, "sampling period must be >= 1");
  e->prof_every = every;
  e->prof_tick = 0;
  return PAL_OK;
}

int pal_profile_end(pal_handle h) {
  ENGINE(h);
  PAL_TRY(e->check(hipStreamSynchronize(e->stream), "stream sync"));
  e->prof_flush();
  e->profiling = false;
  return PAL_OK;
}

int pal_profile_get(pal_handle h, const char* name, double* total_ms, int64_t* launches) {
  ENGINE(h);
  if (!name) return e->fail(PAL_ERR_INVALID, "name is NULL");
  for (size_t i = 0; i < e->slot_names.size(); ++i)
    if (e->slot_names[i] == name) {
      if (total_ms) *total_ms = e->slots[i].ms;
      if (launches) *launches = e->slots[i].launches;
      return PAL_OK;
    }
  if (total_ms) *total_ms = 0;
  if (launches) *launches = 0;
  return PAL_OK;
}

int pal_profile_entry(pal_handle h, int index, char* name, int cap, double* total_ms, int64_t* launches) {
  ENGINE(h);
  if (index < 0 || size_t(index) >= e->slot_names.size()) return PAL_ERR_INVALID;
  if (name && cap > 0) snprintf(name, size_t(cap), "%s", e->slot_names[size_t(index)].c_str());
  if (total_ms) *total_ms = e->slots[size_t(index)].ms;
  if (launches) *launches = e->slots[size_t(index)].launches;
  return PAL_OK;
}

// ---- RCCL gather (librccl is opened on first use so that the library loads on hosts without it) ----
namespace {
struct NcclId { char bytes[128]; };               // ncclUniqueId is an opaque 128-byte struct passed by value
struct Rccl {
  void* lib = nullptr;
  int (*GetUniqueId)(NcclId*) = nullptr;
  int (*CommInitRank)(void**, int, NcclId, int) = nullptr;
  int (*AllGather)(const void*, void*, size_t, int, void*, hipStream_t) = nullptr;
  int (*CommDestroy)(void*) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
};
Rccl g_rccl;
bool rccl_open() {
  if (g_rccl.lib) return true;
  void* lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
  if (!lib) lib = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
  if (!lib) return false;
  g_rccl.GetUniqueId = reinterpret_cast<decltype(g_rccl.GetUniqueId)>(dlsym(lib, "ncclGetUniqueId"));
  g_rccl.CommInitRank = reinterpret_cast<decltype(g_rccl.CommInitRank)>(dlsym(lib, "ncclCommInitRank"));
  g_rccl.AllGather = reinterpret_cast<decltype(g_rccl.AllGather)>(dlsym(lib, "ncclAllGather"));
  g_rccl.CommDestroy = reinterpret_cast<decltype(g_rccl.CommDestroy)>(dlsym(lib, "ncclCommDestroy"));
  g_rccl.GetErrorString = reinterpret_cast<decltype(g_rccl.GetErrorString)>(dlsym(lib, "ncclGetErrorString"));
  if (!g_rccl.GetUniqueId || !g_rccl.CommInitRank || !g_rccl.AllGather || !g_rccl.CommDestroy) return false;
  g_rccl.lib = lib;
  return true;
}
}  // namespace

int pal_comm_unique_id(void* id128) {
  if (!id128) return PAL_ERR_INVALID;
  if (!rccl_open()) return PAL_ERR_COMM;
  return g_rccl.GetUniqueId(static_cast<NcclId*>(id128)) == 0 ? PAL_OK : PAL_ERR_COMM;
}

int pal_comm_init(pal_handle h, int nranks, int rank, const void* id128) {
  ENGINE(h);
  if (!id128 || nranks < 1 || rank < 0 || rank >= nranks) return e->fail(PAL_ERR_INVALID, "bad communicator geometry");
  if (!rccl_open()) return e->fail(PAL_ERR_COMM, "librccl not loadable: %s", dlerror());
  if (e->comm) return e->fail(PAL_ERR_INVALID, "communicator already initialised");
  NcclId id;
  memcpy(&id, id128, sizeof id);
  const int rc = g_rccl.CommInitRank(&e->comm, nranks, id, rank);
  if (rc != 0) return e->fail(PAL_ERR_COMM, "ncclCommInitRank: %s", g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "error");
  return PAL_OK;
}

int pal_comm_all_gather(pal_handle h, const void* d_send, void* d_recv, size_t bytes_per_rank) {
  ENGINE(h);
  if (!e->comm) return e->fail(PAL_ERR_COMM, "communicator not initialised");
  const int rc = g_rccl.AllGather(d_send, d_recv, bytes_per_rank, /* ncclInt8 */ 0, e->comm, e->stream);
  if (rc != 0) return e->fail(PAL_ERR_COMM, "ncclAllGather: %s", g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "error");
  return PAL_OK;
}

int pal_comm_destroy(pal_handle h) {
  ENGINE(h);
  if (e->comm && g_rccl.CommDestroy) {
    hipStreamSynchronize(e->stream);
    g_rccl.CommDestroy(e->comm);
  }
  e->comm = nullptr;
  return PAL_OK;
}

}  // extern "C"
