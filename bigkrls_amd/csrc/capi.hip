// extern "C" surface of libbigkrls_hip.so: context, device buffers, the Level-1
// host-pointer drop-ins (one per .Call routine of the reference's
// src/RcppExports.cpp:147-160) and the Level-2 device-resident operators.
#include "common.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <mutex>

namespace bk {

static thread_local std::string g_last_error;
void set_error(const std::string& msg) { g_last_error = msg; }

int ws_get(bigkrls_ctx* ctx, int slot, int64_t nbytes, void** out) {
  if (slot < 0 || slot >= bigkrls_ctx::kSlots) {
    set_error("workspace slot out of range");
    return BIGKRLS_EINVAL;
  }
  if (nbytes < 256) nbytes = 256;
  if (ctx->ws_bytes[slot] < nbytes) {
    if (ctx->ws[slot]) {
      BK_HIP(hipStreamSynchronize(ctx->stream));
      BK_HIP(hipFree(ctx->ws[slot]));
      ctx->ws[slot] = nullptr;
      ctx->ws_bytes[slot] = 0;
    }
    // grow with some slack so that slowly growing requests do not reallocate every call
    int64_t want = nbytes + nbytes / 8;
    hipError_t e = hipMalloc(&ctx->ws[slot], (size_t)want);
    if (e != hipSuccess) {
      want = nbytes;
      e = hipMalloc(&ctx->ws[slot], (size_t)want);
    }
    if (e != hipSuccess) {
      ctx->ws[slot] = nullptr;
      set_error("workspace allocation of " + std::to_string(nbytes) + " bytes failed: " +
                hipGetErrorString(e));
      return BIGKRLS_ENOMEM;
    }
    ctx->ws_bytes[slot] = want;
    ctx->ws_generation++;
    // BIGKRLS_POISON=1 (diagnostics): a fresh slab starts as all-ones bytes (NaN as a double, -1 as an index), so that
    // anything read before it is written shows up at once instead of depending on what the memory held before
    if (ws_poison()) {   // (on the context's stream: hipMemset runs on the NULL stream, asynchronously to the host)
      BK_HIP(hipMemsetAsync(ctx->ws[slot], 0xFF, (size_t)want, ctx->stream));
      BK_HIP(hipStreamSynchronize(ctx->stream));
    }
  }
  *out = ctx->ws[slot];
  return BIGKRLS_OK;
}

bool ws_poison() {
  static const bool on = getenv("BIGKRLS_POISON") != nullptr;
  return on;
}

// BIGKRLS_POISON=1: every slab the context holds back to all-ones bytes (called at the start of a fit: what the
// previous call left behind must not be read by this one)
int ws_poison_all(bigkrls_ctx* ctx) {
  if (!ws_poison()) return BIGKRLS_OK;
  BK_HIP(hipStreamSynchronize(ctx->stream));
  if (ctx->side_stream && !ctx->side_is_main) BK_HIP(hipStreamSynchronize(ctx->side_stream));
  if (ctx->bg_stream && !ctx->side_is_main) BK_HIP(hipStreamSynchronize(ctx->bg_stream));
  for (int i = 0; i < bigkrls_ctx::kSlots; ++i)
    if (ctx->ws[i] && i != SLOT_COMM_SMALL) BK_HIP(hipMemsetAsync(ctx->ws[i], 0xFF, (size_t)ctx->ws_bytes[i], ctx->stream));
  BK_HIP(hipStreamSynchronize(ctx->stream));
  return BIGKRLS_OK;
}

int pinned_get(bigkrls_ctx* ctx, int64_t ndoubles, double** out) {
  if (ctx->h_pinned_doubles < ndoubles) {
    if (ctx->h_pinned) BK_HIP(hipHostFree(ctx->h_pinned));
    ctx->h_pinned = nullptr;
    int64_t want = std::max<int64_t>(ndoubles, 4096);
    BK_HIP(hipHostMalloc((void**)&ctx->h_pinned, (size_t)want * sizeof(double), hipHostMallocDefault));
    ctx->h_pinned_doubles = want;
  }
  *out = ctx->h_pinned;
  return BIGKRLS_OK;
}

int PinnedFetch::add(void* host_dst, const void* dev_src, size_t bytes) {
  if (!base_) BK_TRY(pinned_get(ctx_, cap_, &base_));
  const size_t nd = (bytes + 7) / 8;
  BK_REQUIRE((int64_t)(used_ + nd) <= cap_, "PinnedFetch: capacity exceeded");
  BK_HIP(hipMemcpyAsync(base_ + used_, dev_src, bytes, hipMemcpyDeviceToHost, ctx_->stream));
  items_.push_back({host_dst, used_, bytes});
  used_ += nd;
  return BIGKRLS_OK;
}

int PinnedFetch::finish() {
  BK_HIP(hipStreamSynchronize(ctx_->stream));
  for (const Item& it : items_) std::memcpy(it.dst, base_ + it.off, it.bytes);
  items_.clear();
  used_ = 0;
  return BIGKRLS_OK;
}

int PinnedStage::reserve(size_t bytes) {
  if (ctx_->h_stage_bytes < (int64_t)bytes) {
    BK_HIP(hipStreamSynchronize(ctx_->stream));          // (an earlier call's uploads may still be queued)
    if (ctx_->h_stage) BK_HIP(hipHostFree(ctx_->h_stage));
    ctx_->h_stage = nullptr;
    ctx_->h_stage_bytes = 0;
    const size_t want = bytes + bytes / 4;
    BK_HIP(hipHostMalloc((void**)&ctx_->h_stage, want, hipHostMallocDefault));
    ctx_->h_stage_bytes = (int64_t)want;
  }
  used_ = fixed_ = 0;
  return BIGKRLS_OK;
}

void* PinnedStage::alloc(size_t bytes) {
  const size_t a = (used_ + 63) & ~(size_t)63;
  if (!ctx_->h_stage || a + bytes > (size_t)ctx_->h_stage_bytes) return nullptr;
  used_ = a + bytes;
  return ctx_->h_stage + a;
}

void* PinnedStage::fixed(size_t bytes) {
  void* p = alloc(bytes);
  if (p) fixed_ = used_;
  return p;
}

int PinnedStage::send(void* dev_dst, const void* slice, size_t bytes) {
  if (bytes == 0) return BIGKRLS_OK;
  BK_HIP(hipMemcpyAsync(dev_dst, slice, bytes, hipMemcpyHostToDevice, ctx_->stream));
  return BIGKRLS_OK;
}

int PinnedStage::put(void* dev_dst, const void* host_src, size_t bytes) {
  if (bytes == 0) return BIGKRLS_OK;
  void* p = alloc(bytes);
  if (!p) {
    // the arena is full: everything queued so far has to execute before its slices can be reused
    BK_HIP(hipStreamSynchronize(ctx_->stream));
    reset();
    p = alloc(bytes);
    BK_REQUIRE(p, "PinnedStage: an upload larger than the arena");
  }
  std::memcpy(p, host_src, bytes);
  return send(dev_dst, p, bytes);
}

int ensure_dyn_smem(bigkrls_ctx* ctx, const void* kernel, size_t bytes) {
  for (const void* k : ctx->dyn_smem_done)
    if (k == kernel) return BIGKRLS_OK;
  BK_HIP(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
  ctx->dyn_smem_done.push_back(kernel);
  return BIGKRLS_OK;
}

int resident_capacity(bigkrls_ctx* ctx, const void* kernel, int* cap, int threads) {
  for (auto& kv : ctx->resident_cap)
    if (kv.first == kernel) { *cap = kv.second; return BIGKRLS_OK; }
  int per_cu = 0, ncu = 0;
  BK_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, threads, 0));
  BK_HIP(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, ctx->device));
  *cap = std::max(1, per_cu * ncu);
  ctx->resident_cap.emplace_back(kernel, *cap);
  return BIGKRLS_OK;
}

int side_stream_get(bigkrls_ctx* ctx) {
  if (!ctx->side_stream && !ctx->side_is_main) {   // (side_is_main: the main stream may be the NULL stream, i.e. a null handle)
    // BIGKRLS_NO_SIDE=1 (diagnostics): no second stream -- the "look-ahead" work is queued on the main stream, every
    // fork / join becomes a no-op. Slower; separates cross-stream ordering from everything else when hunting a
    // nondeterminism (tools/oversub_single.py --arms).
    if (getenv("BIGKRLS_NO_SIDE")) {
      ctx->side_stream = ctx->stream;
      ctx->bg_stream = ctx->stream;
      ctx->side_is_main = true;
      BK_HIP(hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming));
      BK_HIP(hipEventCreateWithFlags(&ctx->ev_join, hipEventDisableTiming));
      BK_HIP(hipEventCreateWithFlags(&ctx->ev_join2, hipEventDisableTiming));
      BK_HIP(hipEventCreateWithFlags(&ctx->ev_pq, hipEventDisableTiming));
      return BIGKRLS_OK;
    }
    // highest priority: its short latency-bound launches must not queue behind the thousands of
    // workgroups of the throughput kernel they overlap with
    int prio_lo = 0, prio_hi = 0;
    BK_HIP(hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi));
    BK_HIP(hipStreamCreateWithPriority(&ctx->side_stream, hipStreamNonBlocking, prio_hi));
    BK_HIP(hipStreamCreateWithPriority(&ctx->bg_stream, hipStreamNonBlocking, prio_lo));
    BK_HIP(hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming));
    BK_HIP(hipEventCreateWithFlags(&ctx->ev_join, hipEventDisableTiming));
    BK_HIP(hipEventCreateWithFlags(&ctx->ev_join2, hipEventDisableTiming));
    BK_HIP(hipEventCreateWithFlags(&ctx->ev_pq, hipEventDisableTiming));
  }
  return BIGKRLS_OK;
}

static bigkrls_ctx::ProfEntry* prof_entry(bigkrls_ctx* ctx, const char* name) {
  for (auto& e : ctx->prof)
    if (e.name == name) return &e;
  ctx->prof.emplace_back();
  ctx->prof.back().name = name;
  return &ctx->prof.back();
}

int prof_begin(bigkrls_ctx* ctx, const char* name, double work, hipStream_t stream) {
  if (!ctx->profile) return BIGKRLS_OK;
  bigkrls_ctx::ProfSample s{};
  auto take = [&](hipEvent_t* e) -> int {
    if (!ctx->prof_pool.empty()) { *e = ctx->prof_pool.back(); ctx->prof_pool.pop_back(); return BIGKRLS_OK; }
    BK_HIP(hipEventCreate(e));
    return BIGKRLS_OK;
  };
  BK_TRY(take(&s.e0));
  BK_TRY(take(&s.e1));
  s.work = work;
  BK_HIP(hipEventRecord(s.e0, stream ? stream : ctx->stream));
  prof_entry(ctx, name)->pending.push_back(s);
  return BIGKRLS_OK;
}

int prof_end(bigkrls_ctx* ctx, const char* name, hipStream_t stream) {
  if (!ctx->profile) return BIGKRLS_OK;
  auto* e = prof_entry(ctx, name);
  if (e->pending.empty()) return BIGKRLS_OK;
  BK_HIP(hipEventRecord(e->pending.back().e1, stream ? stream : ctx->stream));
  return BIGKRLS_OK;
}

static int prof_flush(bigkrls_ctx* ctx) {
  BK_HIP(hipStreamSynchronize(ctx->stream));
  if (ctx->side_stream) BK_HIP(hipStreamSynchronize(ctx->side_stream));
  if (ctx->bg_stream) BK_HIP(hipStreamSynchronize(ctx->bg_stream));
  for (auto& e : ctx->prof) {
    for (auto& s : e.pending) {
      float ms = 0.f;
      BK_HIP(hipEventElapsedTime(&ms, s.e0, s.e1));
      e.ms += ms;
      e.work += s.work;
      e.launches += 1;
      ctx->prof_pool.push_back(s.e0);
      ctx->prof_pool.push_back(s.e1);
    }
    e.pending.clear();
  }
  return BIGKRLS_OK;
}

static int check_ctx(bigkrls_ctx* ctx) {
  if (!ctx) {
    set_error("null context");
    return BIGKRLS_EINVAL;
  }
  BK_HIP(hipSetDevice(ctx->device));
  return BIGKRLS_OK;
}

// default context for the Level-1 entry points (device 0, created on first use)
static std::mutex g_default_mutex;
static bigkrls_ctx* g_default_ctx = nullptr;
static int default_ctx(bigkrls_ctx** out) {
  std::lock_guard<std::mutex> lock(g_default_mutex);
  if (!g_default_ctx) {
    int s = bigkrls_ctx_create(0, &g_default_ctx);
    if (s != BIGKRLS_OK) return s;
  }
  *out = g_default_ctx;
  return BIGKRLS_OK;
}

// RAII device staging buffer for Level 1
struct DevBuf {
  double* p = nullptr;
  ~DevBuf() {
    if (p) (void)hipFree(p);
  }
  int alloc(int64_t n) {
    if (n <= 0) n = 1;
    hipError_t e = hipMalloc((void**)&p, (size_t)n * sizeof(double));
    if (e != hipSuccess) {
      p = nullptr;
      set_error(std::string("device allocation failed: ") + hipGetErrorString(e));
      return BIGKRLS_ENOMEM;
    }
    return BIGKRLS_OK;
  }
  int upload(bigkrls_ctx* ctx, const double* h, int64_t n) {
    BK_TRY(alloc(n));
    if (n > 0) BK_HIP(hipMemcpyAsync(p, h, (size_t)n * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    return BIGKRLS_OK;
  }
  int download(bigkrls_ctx* ctx, double* h, int64_t n) {
    if (n > 0) BK_HIP(hipMemcpyAsync(h, p, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    BK_HIP(hipStreamSynchronize(ctx->stream));
    return BIGKRLS_OK;
  }
};

}  // namespace bk

using namespace bk;

extern "C" {

int bigkrls_version(void) { return 100; }

const char* bigkrls_last_error(void) { return bk::g_last_error.c_str(); }

int bigkrls_device_count(int* count) {
  if (!count) return BIGKRLS_EINVAL;
  int c = 0;
  hipError_t e = hipGetDeviceCount(&c);
  if (e != hipSuccess) {
    *count = 0;
    set_error(std::string("hipGetDeviceCount failed: ") + hipGetErrorString(e));
    return BIGKRLS_ENODEVICE;
  }
  *count = c;
  return BIGKRLS_OK;
}

static int ctx_create_common(int device, void* stream, bool own, bigkrls_ctx** out) {
  if (!out) return BIGKRLS_EINVAL;
  int c = 0;
  if (hipGetDeviceCount(&c) != hipSuccess || c <= 0) {
    set_error("no HIP device available: libbigkrls_hip has no CPU fallback");
    return BIGKRLS_ENODEVICE;
  }
  if (device < 0 || device >= c) {
    set_error("device index out of range");
    return BIGKRLS_ENODEVICE;
  }
  BK_HIP(hipSetDevice(device));
  bigkrls_ctx* ctx = new bigkrls_ctx();
  ctx->device = device;
  if (own) {
    hipError_t e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
      delete ctx;
      set_error(std::string("hipStreamCreate failed: ") + hipGetErrorString(e));
      return BIGKRLS_EHIP;
    }
    ctx->owns_stream = true;
  } else {
    ctx->stream = (hipStream_t)stream;
    ctx->owns_stream = false;
  }
  *out = ctx;
  return BIGKRLS_OK;
}

int bigkrls_ctx_create(int device, bigkrls_ctx** ctx) { return ctx_create_common(device, nullptr, true, ctx); }

int bigkrls_ctx_create_on_stream(int device, void* hip_stream, bigkrls_ctx** ctx) {
  return ctx_create_common(device, hip_stream, false, ctx);
}

int bigkrls_ctx_release_workspace(bigkrls_ctx* ctx) {
  BK_TRY(check_ctx(ctx));
  BK_HIP(hipStreamSynchronize(ctx->stream));
  for (int i = 0; i < bigkrls_ctx::kSlots; ++i) {
    if (ctx->ws[i]) BK_HIP(hipFree(ctx->ws[i]));
    ctx->ws[i] = nullptr;
    ctx->ws_bytes[i] = 0;
  }
  ctx->ws_generation++;
  if (ctx->s1_graph_exec) {        // (points into the workspace that has just gone)
    (void)hipGraphExecDestroy((hipGraphExec_t)ctx->s1_graph_exec);
    ctx->s1_graph_exec = nullptr;
  }
  return BIGKRLS_OK;
}

int bigkrls_ctx_destroy(bigkrls_ctx* ctx) {
  if (!ctx) return BIGKRLS_OK;
  (void)hipSetDevice(ctx->device);
  (void)bigkrls_ctx_release_workspace(ctx);      // (also drops the captured stage-1 graph)
  if (ctx->dist_s1 && ctx->dist_s1_free) ctx->dist_s1_free(ctx->dist_s1);
  if (ctx->h_pinned) (void)hipHostFree(ctx->h_pinned);
  if (ctx->h_stage) (void)hipHostFree(ctx->h_stage);
  if (ctx->h_plan) (void)hipHostFree(ctx->h_plan);
  for (hipEvent_t e : ctx->prof_pool) (void)hipEventDestroy(e);
  if (ctx->side_stream && !ctx->side_is_main) (void)hipStreamDestroy(ctx->side_stream);
  if (ctx->bg_stream && !ctx->side_is_main) (void)hipStreamDestroy(ctx->bg_stream);
  if (ctx->ev_fork) (void)hipEventDestroy(ctx->ev_fork);
  if (ctx->ev_join) (void)hipEventDestroy(ctx->ev_join);
  if (ctx->ev_join2) (void)hipEventDestroy(ctx->ev_join2);
  if (ctx->ev_pq) (void)hipEventDestroy(ctx->ev_pq);
  if (ctx->ev_graph) (void)hipEventDestroy(ctx->ev_graph);
  if (ctx->graph_stream) (void)hipStreamDestroy(ctx->graph_stream);
  if (ctx->owns_stream && ctx->stream) (void)hipStreamDestroy(ctx->stream);
  delete ctx;
  return BIGKRLS_OK;
}

int bigkrls_ctx_sync(bigkrls_ctx* ctx) {
  BK_TRY(check_ctx(ctx));
  BK_HIP(hipStreamSynchronize(ctx->stream));
  return BIGKRLS_OK;
}

void* bigkrls_ctx_stream(bigkrls_ctx* ctx) { return ctx ? (void*)ctx->stream : nullptr; }

int bigkrls_ctx_set_profile(bigkrls_ctx* ctx, int enable) {
  BK_TRY(check_ctx(ctx));
  BK_TRY(prof_flush(ctx));
  ctx->profile = enable != 0;
  if (enable) ctx->prof.clear();
  return BIGKRLS_OK;
}

int bigkrls_ctx_get_profile(bigkrls_ctx* ctx, const char* name, double* total_ms, double* total_work,
                            int64_t* launches) {
  BK_TRY(check_ctx(ctx));
  BK_REQUIRE(name && total_ms && total_work && launches, "get_profile: null argument");
  BK_TRY(prof_flush(ctx));
  *total_ms = 0; *total_work = 0; *launches = 0;
  for (auto& e : ctx->prof)
    if (e.name == name) { *total_ms = e.ms; *total_work = e.work; *launches = e.launches; }
  return BIGKRLS_OK;
}

int bigkrls_ctx_get_counters(bigkrls_ctx* ctx, int64_t out[3]) {
  BK_TRY(check_ctx(ctx));
  BK_REQUIRE(out, "get_counters: null output");
  out[0] = ctx->n_redone;
  out[1] = ctx->n_replayed;
  out[2] = ctx->n_replica_diff;
  return BIGKRLS_OK;
}

int64_t bigkrls_ctx_workspace_bytes(bigkrls_ctx* ctx) {
  if (!ctx) return 0;
  int64_t s = 0;
  for (int i = 0; i < bigkrls_ctx::kSlots; ++i) s += ctx->ws_bytes[i];
  return s;
}

int bigkrls_dev_alloc(bigkrls_ctx* ctx, int64_t nbytes, void** dptr) {
  BK_TRY(check_ctx(ctx));
  BK_REQUIRE(dptr && nbytes >= 0, "dev_alloc: bad arguments");
  if (nbytes == 0) nbytes = 8;
  hipError_t e = hipMalloc(dptr, (size_t)nbytes);
  if (e != hipSuccess) {
    *dptr = nullptr;
    set_error("hipMalloc of " + std::to_string(nbytes) + " bytes failed: " + hipGetErrorString(e));
    return BIGKRLS_ENOMEM;
  }
  return BIGKRLS_OK;
}

int bigkrls_dev_free(bigkrls_ctx* ctx, void* dptr) {
  BK_TRY(check_ctx(ctx));
  if (dptr) {
    BK_HIP(hipStreamSynchronize(ctx->stream));
    BK_HIP(hipFree(dptr));
  }
  return BIGKRLS_OK;
}

int bigkrls_h2d(bigkrls_ctx* ctx, void* dst, const void* src, int64_t nbytes) {
  BK_TRY(check_ctx(ctx));
  if (nbytes <= 0) return BIGKRLS_OK;
  BK_REQUIRE(dst && src, "h2d: null pointer");
  BK_HIP(hipMemcpyAsync(dst, src, (size_t)nbytes, hipMemcpyHostToDevice, ctx->stream));
  BK_HIP(hipStreamSynchronize(ctx->stream));
  return BIGKRLS_OK;
}

int bigkrls_d2h(bigkrls_ctx* ctx, void* dst, const void* src, int64_t nbytes) {
  BK_TRY(check_ctx(ctx));
  if (nbytes <= 0) return BIGKRLS_OK;
  BK_REQUIRE(dst && src, "d2h: null pointer");
  BK_HIP(hipMemcpyAsync(dst, src, (size_t)nbytes, hipMemcpyDeviceToHost, ctx->stream));
  BK_HIP(hipStreamSynchronize(ctx->stream));
  return BIGKRLS_OK;
}

int bigkrls_d2d(bigkrls_ctx* ctx, void* dst, const void* src, int64_t nbytes) {
  BK_TRY(check_ctx(ctx));
  if (nbytes <= 0) return BIGKRLS_OK;
  BK_REQUIRE(dst && src, "d2d: null pointer");
  BK_HIP(hipMemcpyAsync(dst, src, (size_t)nbytes, hipMemcpyDeviceToDevice, ctx->stream));
  return BIGKRLS_OK;
}

int bigkrls_event_create(void** ev) {
  BK_REQUIRE(ev, "event_create: null");
  hipEvent_t e;
  BK_HIP(hipEventCreate(&e));
  *ev = (void*)e;
  return BIGKRLS_OK;
}
int bigkrls_event_destroy(void* ev) {
  if (ev) BK_HIP(hipEventDestroy((hipEvent_t)ev));
  return BIGKRLS_OK;
}
int bigkrls_event_record(bigkrls_ctx* ctx, void* ev) {
  BK_TRY(check_ctx(ctx));
  BK_HIP(hipEventRecord((hipEvent_t)ev, ctx->stream));
  return BIGKRLS_OK;
}
int bigkrls_event_elapsed_ms(void* a, void* b, double* ms) {
  BK_REQUIRE(a && b && ms, "event_elapsed: null");
  BK_HIP(hipEventSynchronize((hipEvent_t)b));
  float f = 0.f;
  BK_HIP(hipEventElapsedTime(&f, (hipEvent_t)a, (hipEvent_t)b));
  *ms = (double)f;
  return BIGKRLS_OK;
}

// ---------------------------------------------------------------------------
// Level 2
// ---------------------------------------------------------------------------
int bigkrls_dev_kernel_block(bigkrls_ctx* ctx, const double* A, int64_t u, int64_t lda,
                             const double* B, int64_t v, int64_t ldb, int64_t p, double sigma,
                             double* out, int64_t ldo, int64_t diag_shift) {
  BK_TRY(check_ctx(ctx));
  return kernel_block(ctx, A, u, lda, B, v, ldb, p, sigma, out, ldo, diag_shift);
}

int bigkrls_dev_gemm(bigkrls_ctx* ctx, int ta, int tb, int64_t m, int64_t n, int64_t k,
                     double alpha, const double* A, int64_t lda, const double* B, int64_t ldb,
                     double beta, double* C, int64_t ldc) {
  BK_TRY(check_ctx(ctx));
  return gemm(ctx, ta, tb, m, n, k, alpha, A, lda, B, ldb, beta, C, ldc);
}

int bigkrls_dev_multdiag(bigkrls_ctx* ctx, const double* A, int64_t n, int64_t k, int64_t lda,
                         const double* diag, double* out, int64_t ldo) {
  BK_TRY(check_ctx(ctx));
  return multdiag(ctx, A, n, k, lda, diag, out, ldo);
}

int bigkrls_dev_eigen(bigkrls_ctx* ctx, const double* A, int64_t n, int64_t lda, int64_t n_vals,
                      double* vals, int64_t n_vecs_max, double keep_thresh, double* vecs,
                      int64_t ldv, int64_t* h_n_vecs) {
  BK_TRY(check_ctx(ctx));
  return eigen(ctx, A, n, lda, n_vals, vals, n_vecs_max, keep_thresh, vecs, ldv, h_n_vecs);
}

int bigkrls_dev_eigen_part(bigkrls_ctx* ctx, const double* A, int64_t n, int64_t lda, int64_t n_vals,
                           double* vals, int64_t n_vecs_max, double keep_thresh, double* vecs,
                           int64_t ldv, int64_t* h_n_vecs, int32_t part_index, int32_t part_count) {
  BK_TRY(check_ctx(ctx));
  return eigen(ctx, A, n, lda, n_vals, vals, n_vecs_max, keep_thresh, vecs, ldv, h_n_vecs, part_index,
               part_count);
}

int bigkrls_dev_fill_random(bigkrls_ctx* ctx, double* p, int64_t count, uint32_t seed) {
  BK_TRY(check_ctx(ctx));
  return fill_random(ctx, p, count, seed);
}

int bigkrls_dev_lanczos_projected(bigkrls_ctx* ctx, const double* d_A_blocks, const double* d_beta_blocks,
                                  int64_t steps, int64_t b, double* d_T) {
  BK_TRY(check_ctx(ctx));
  return lanczos_projected(ctx, d_A_blocks, d_beta_blocks, (int)steps, (int)b, d_T);
}

int bigkrls_dev_cholqr2(bigkrls_ctx* ctx, double* W, double* tmp, int64_t n, int64_t b, double* h_R,
                        int32_t* h_breakdown, double* d_R) {
  BK_TRY(check_ctx(ctx));
  int brk = 0;
  const int rc = cholqr2_block(ctx, W, tmp, n, (int)b, h_R, &brk, d_R);
  if (h_breakdown) *h_breakdown = brk;
  return rc;
}

int bigkrls_dev_copy_matrix(bigkrls_ctx* ctx, const double* src, int64_t m, int64_t n, int64_t lds,
                            double* dst, int64_t ldd) {
  BK_TRY(check_ctx(ctx));
  return copy_matrix(ctx, src, m, n, lds, dst, ldd);
}

int bigkrls_dev_qty(bigkrls_ctx* ctx, const double* Q, int64_t n, int64_t k, int64_t ldq,
                    const double* y, double* a) {
  BK_TRY(check_ctx(ctx));
  return qty(ctx, Q, n, k, ldq, y, a);
}

int bigkrls_dev_solveforc(bigkrls_ctx* ctx, const double* Q, int64_t n_rows, int64_t k,
                          int64_t ldq, const double* d, const double* a, double lambda, double* c,
                          double* h_Le) {
  BK_TRY(check_ctx(ctx));
  return solveforc(ctx, Q, n_rows, k, ldq, d, a, lambda, c, h_Le);
}

int bigkrls_dev_lambda_search(bigkrls_ctx* ctx, const double* Q, int64_t n, int64_t k, int64_t ldq,
                              const double* d, const double* a, const double* h_vals_all,
                              int64_t n_vals, double h_L, double h_U, double h_tol,
                              double* h_lambda, int64_t* h_nprobes, double* h_trace,
                              int64_t max_trace) {
  BK_TRY(check_ctx(ctx));
  return lambda_search(ctx, Q, n, k, ldq, d, a, h_vals_all, n_vals, h_L, h_U, h_tol, h_lambda,
                       h_nprobes, h_trace, max_trace);
}

int bigkrls_lambda_bounds(const double* h_vals_all, int64_t n_vals, int64_t n, double* h_L,
                          double* h_U) {
  return lambda_bounds(h_vals_all, n_vals, n, h_L, h_U);
}

int bigkrls_dev_deriv_rows(bigkrls_ctx* ctx, const double* Krows, int64_t n, int64_t n_rows,
                           int64_t ldk, int64_t row0, const double* X_full, int64_t p, int64_t ldx,
                           const int32_t* h_is_binary, const double* c, double sigma, double* D,
                           int64_t ldd, double* S, int64_t lds) {
  BK_TRY(check_ctx(ctx));
  return deriv_rows(ctx, Krows, n, n_rows, ldk, row0, X_full, p, ldx, h_is_binary, c, sigma, D,
                    ldd, S, lds);
}

int bigkrls_dev_deriv_var(bigkrls_ctx* ctx, const double* Q, int64_t n, int64_t k, int64_t ldq,
                          const double* wv, const double* S, int64_t p, int64_t lds,
                          const double* h_scale, double* h_var) {
  BK_TRY(check_ctx(ctx));
  return deriv_var(ctx, Q, n, k, ldq, wv, S, p, lds, h_scale, h_var);
}

int bigkrls_dev_gemv(bigkrls_ctx* ctx, int trans, int64_t m, int64_t n, double alpha,
                     const double* A, int64_t lda, const double* x, double beta, double* y) {
  BK_TRY(check_ctx(ctx));
  return gemv(ctx, trans, m, n, alpha, A, lda, x, beta, y);
}

int bigkrls_dev_dot(bigkrls_ctx* ctx, int64_t n, const double* x, const double* y, double* h_out) {
  BK_TRY(check_ctx(ctx));
  return dot_host(ctx, n, x, y, h_out);
}

int bigkrls_dev_diag(bigkrls_ctx* ctx, const double* A, int64_t n, int64_t lda, double* out) {
  BK_TRY(check_ctx(ctx));
  return diag_extract(ctx, A, n, lda, out);
}

int bigkrls_dev_scale(bigkrls_ctx* ctx, int64_t n, double alpha, double* x) {
  BK_TRY(check_ctx(ctx));
  return scale(ctx, n, alpha, x);
}

// ---------------------------------------------------------------------------
// Level 1: host-pointer drop-ins
// ---------------------------------------------------------------------------
int bigkrls_gauss_kernel(const double* X, int64_t n, int64_t p, double sigma, double* out) {
  BK_REQUIRE(X && out && n > 0 && p > 0, "gauss_kernel: bad arguments");
  bigkrls_ctx* ctx;
  BK_TRY(default_ctx(&ctx));
  BK_TRY(check_ctx(ctx));
  DevBuf dX, dK;
  BK_TRY(dX.upload(ctx, X, n * p));
  BK_TRY(dK.alloc(n * n));
  BK_TRY(kernel_block(ctx, dX.p, n, n, dX.p, n, n, p, sigma, dK.p, n, 0));
  return dK.download(ctx, out, n * n);
}

int bigkrls_temp_kernel(const double* A, int64_t u, const double* B, int64_t v, int64_t p,
                        double sigma, double* out) {
  BK_REQUIRE(A && B && out && u > 0 && v > 0 && p > 0, "temp_kernel: bad arguments");
  bigkrls_ctx* ctx;
  BK_TRY(default_ctx(&ctx));
  BK_TRY(check_ctx(ctx));
  DevBuf dA, dB, dO;
  BK_TRY(dA.upload(ctx, A, u * p));
  BK_TRY(dB.upload(ctx, B, v * p));
  BK_TRY(dO.alloc(u * v));
  BK_TRY(kernel_block(ctx, dA.p, u, u, dB.p, v, v, p, sigma, dO.p, u, -1));
  return dO.download(ctx, out, u * v);
}

int bigkrls_eigen(const double* A, int64_t n, int64_t neig, double* vals, double* vecs) {
  BK_REQUIRE(A && vals && vecs && n > 0 && neig > 0 && neig <= n, "eigen: bad arguments");
  bigkrls_ctx* ctx;
  BK_TRY(default_ctx(&ctx));
  BK_TRY(check_ctx(ctx));
  DevBuf dA, dvals, dvecs;
  BK_TRY(dA.upload(ctx, A, n * n));
  BK_TRY(dvals.alloc(neig));
  BK_TRY(dvecs.alloc(n * neig));
  int64_t nv = 0;
  BK_TRY(eigen(ctx, dA.p, n, n, neig, dvals.p, neig, -1.0, dvecs.p, n, &nv));
  BK_TRY(dvals.download(ctx, vals, neig));
  return dvecs.download(ctx, vecs, n * neig);
}

int bigkrls_solveforc(const double* Q, int64_t n, int64_t k, const double* vals, int64_t nvals,
                      const double* y, double lambda, double* Le, double* coeffs) {
  BK_REQUIRE(Q && vals && y && Le && coeffs && n > 0 && k > 0 && nvals >= k,
             "solveforc: bad arguments");
  bigkrls_ctx* ctx;
  BK_TRY(default_ctx(&ctx));
  BK_TRY(check_ctx(ctx));
  DevBuf dQ, dd, dy, da, dc;
  BK_TRY(dQ.upload(ctx, Q, n * k));
  BK_TRY(dd.upload(ctx, vals, k));  // only the first k eigenvalues are used (quirk Q1)
  BK_TRY(dy.upload(ctx, y, n));
  BK_TRY(da.alloc(k));
  BK_TRY(dc.alloc(n));
  BK_TRY(qty(ctx, dQ.p, n, k, n, dy.p, da.p));
  BK_TRY(solveforc(ctx, dQ.p, n, k, n, dd.p, da.p, lambda, dc.p, Le));
  return dc.download(ctx, coeffs, n);
}

int bigkrls_multdiag(const double* A, int64_t n, int64_t k, const double* diag, double* out) {
  BK_REQUIRE(A && diag && out && n > 0 && k > 0, "multdiag: bad arguments");
  bigkrls_ctx* ctx;
  BK_TRY(default_ctx(&ctx));
  BK_TRY(check_ctx(ctx));
  DevBuf dA, dd, dO;
  BK_TRY(dA.upload(ctx, A, n * k));
  BK_TRY(dd.upload(ctx, diag, k));
  BK_TRY(dO.alloc(n * k));
  BK_TRY(multdiag(ctx, dA.p, n, k, n, dd.p, dO.p, n));
  return dO.download(ctx, out, n * k);
}

static int l1_gemm(int ta, int tb, int64_t m, int64_t n, int64_t k, const double* A, int64_t a_rows,
                   int64_t a_cols, const double* B, int64_t b_rows, int64_t b_cols, double* out) {
  bigkrls_ctx* ctx;
  BK_TRY(default_ctx(&ctx));
  BK_TRY(check_ctx(ctx));
  DevBuf dA, dB, dO;
  BK_TRY(dA.upload(ctx, A, a_rows * a_cols));
  const double* pB = nullptr;
  if (B == A && a_rows == b_rows && a_cols == b_cols) {
    pB = dA.p;
  } else {
    BK_TRY(dB.upload(ctx, B, b_rows * b_cols));
    pB = dB.p;
  }
  BK_TRY(dO.alloc(m * n));
  BK_TRY(gemm(ctx, ta, tb, m, n, k, 1.0, dA.p, a_rows, pB, b_rows, 0.0, dO.p, m));
  return dO.download(ctx, out, m * n);
}

int bigkrls_crossprod(const double* A, int64_t n, int64_t ak, const double* B, int64_t bk,
                      double* out) {
  BK_REQUIRE(A && B && out && n > 0 && ak > 0 && bk > 0, "crossprod: bad arguments");
  return l1_gemm(1, 0, ak, bk, n, A, n, ak, B, n, bk, out);
}

int bigkrls_xtx(const double* A, int64_t n, int64_t k, double* out) {
  BK_REQUIRE(A && out && n > 0 && k > 0, "xtx: bad arguments");
  return l1_gemm(1, 0, k, k, n, A, n, k, A, n, k, out);
}

int bigkrls_tcrossprod(const double* A, int64_t an, int64_t k, const double* B, int64_t bn,
                       double* out) {
  BK_REQUIRE(A && B && out && an > 0 && bn > 0 && k > 0, "tcrossprod: bad arguments");
  return l1_gemm(0, 1, an, bn, k, A, an, k, B, bn, k, out);
}

int bigkrls_xxt(const double* A, int64_t n, int64_t k, double* out) {
  BK_REQUIRE(A && out && n > 0 && k > 0, "xxt: bad arguments");
  return l1_gemm(0, 1, n, n, k, A, n, k, A, n, k, out);
}

// BigDerivMat takes V explicitly (src/bigderiv_v3.cpp:114). The drop-in honours
// that contract: s'Vs is evaluated as s'(V s) with V s one N x N x P GEMM.
__global__ void l1_coldots_kernel(int n, int p, const double* __restrict__ S,
                                  const double* __restrict__ VS, double* __restrict__ out) {
  __shared__ double sh[4];
  const double* s = S + (int64_t)blockIdx.x * n;
  const double* v = VS + (int64_t)blockIdx.x * n;
  double acc = 0.0;
  for (int i = threadIdx.x; i < n; i += 256) acc += s[i] * v[i];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) out[blockIdx.x] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

int bigkrls_neffective(const double* X, int64_t n, int64_t p, double* neff) {
  BK_REQUIRE(X && neff && n > 0 && p > 0, "neffective: bad arguments");
  bigkrls_ctx* ctx;
  BK_TRY(default_ctx(&ctx));
  BK_TRY(check_ctx(ctx));
  DevBuf dX;
  BK_TRY(dX.upload(ctx, X, n * p));
  return neffective(ctx, dX.p, n, n, p, neff);
}

int bigkrls_dev_neffective(bigkrls_ctx* ctx, const double* X, int64_t n, int64_t ldx, int64_t p,
                           double* h_neff) {
  BK_TRY(check_ctx(ctx));
  return neffective(ctx, X, n, ldx, p, h_neff);
}

int bigkrls_derivmat(const double* X, int64_t n, int64_t p, const double* K, const double* V,
                     double* D, double* var, const double* coeffs, double sigma) {
  BK_REQUIRE(X && K && V && D && var && coeffs && n > 0 && p > 0, "derivmat: bad arguments");
  bigkrls_ctx* ctx;
  BK_TRY(default_ctx(&ctx));
  BK_TRY(check_ctx(ctx));
  // binary flags exactly as src/bigderiv_v3.cpp:28-31 (unique values == 2)
  std::vector<int32_t> isbin(p);
  std::vector<double> scale(p);
  for (int64_t j = 0; j < p; ++j) {
    const double* x = X + j * n;
    double a = x[0], b = 0.0;
    int nu = 1;
    double lo = x[0], hi = x[0];
    for (int64_t i = 1; i < n && nu <= 2; ++i) {
      if (x[i] == a) continue;
      if (nu == 1) { b = x[i]; nu = 2; }
      else if (x[i] != b) nu = 3;
    }
    for (int64_t i = 1; i < n; ++i) { lo = std::min(lo, x[i]); hi = std::max(hi, x[i]); }
    isbin[j] = (nu == 2) ? 1 : 0;
    if (isbin[j]) {
      const double sd = 1.0 / (hi - lo);
      scale[j] = 2.0 * sd * sd / ((double)n * (double)n);          // :85
    } else {
      scale[j] = 4.0 / (sigma * sigma * (double)n * (double)n);    // :105
    }
  }
  DevBuf dX, dK, dV, dc, dD, dS, dVS, dout;
  BK_TRY(dX.upload(ctx, X, n * p));
  BK_TRY(dK.upload(ctx, K, n * n));
  BK_TRY(dc.upload(ctx, coeffs, n));
  BK_TRY(dD.alloc(n * p));
  BK_TRY(dS.alloc(n * p));
  BK_TRY(deriv_rows(ctx, dK.p, n, n, n, 0, dX.p, p, n, isbin.data(), dc.p, sigma, dD.p, n, dS.p, n));
  BK_TRY(dD.download(ctx, D, n * p));
  // reuse K's device buffer for V
  BK_HIP(hipMemcpyAsync(dK.p, V, (size_t)n * n * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  BK_TRY(dVS.alloc(n * p));
  BK_TRY(dout.alloc(p));
  BK_TRY(gemm(ctx, 0, 0, n, p, n, 1.0, dK.p, n, dS.p, n, 0.0, dVS.p, n));
  hipLaunchKernelGGL(l1_coldots_kernel, dim3((unsigned)p), dim3(256), 0, ctx->stream, (int)n,
                     (int)p, (const double*)dS.p, (const double*)dVS.p, dout.p);
  BK_CHECK_LAUNCH();
  std::vector<double> raw(p);
  BK_TRY(dout.download(ctx, raw.data(), p));
  for (int64_t j = 0; j < p; ++j) var[j] = scale[j] * raw[j];
  return BIGKRLS_OK;
}

}  // extern "C"
