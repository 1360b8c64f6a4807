// Multi-GPU side of the library: one rank object per process (bigkrls_comm), its collectives (RCCL over xGMI, or a
// caller-supplied callback table), the row-block gather helper and the dense eigensolver with stage 1 partitioned by
// column blocks over the ranks. The reference's parallel path is PSOCK workers over one mmap'd K, one derivative
// column each (R/bigKRLS.R:337-363); here K itself is partitioned by row block and never
// gathered. bigkrls_fit_dist (csrc/fit.hip) is the whole fit over these pieces.
#include "common.h"

#include <dlfcn.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <cstring>
#include <mutex>

namespace bk {
namespace {

// librccl is opened at run time: a single-GPU user needs no RCCL, and a process that already holds one (PyTorch
// bundles its own) keeps using exactly that one.
struct RcclApi {
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  bool ok = false;
  std::string why;
};

RcclApi& rccl() {
  static RcclApi api;
  static std::once_flag once;
  std::call_once(once, [] {
    void* h = nullptr;
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    if (const char* user = getenv("BIGKRLS_RCCL_LIB")) {   // an RCCL build of the caller's choice: that one or none
      h = dlopen(user, RTLD_NOW | RTLD_LOCAL);
    } else {
      for (const char* nm : names) {             // one that is already mapped first
        h = dlopen(nm, RTLD_NOW | RTLD_NOLOAD);
        if (h) break;
      }
      if (!h)
        for (const char* nm : names) {
          h = dlopen(nm, RTLD_NOW | RTLD_LOCAL);
          if (h) break;
        }
    }
    if (!h) {
      const char* e = dlerror();                   // (a second call would return NULL: the first one clears the state)
      api.why = std::string("librccl not found: ") + (e ? e : "");
      return;
    }
    auto sym = [&](const char* nm) { return dlsym(h, nm); };
    api.GetUniqueId = (decltype(api.GetUniqueId))sym("ncclGetUniqueId");
    api.CommInitRank = (decltype(api.CommInitRank))sym("ncclCommInitRank");
    api.CommDestroy = (decltype(api.CommDestroy))sym("ncclCommDestroy");
    api.AllReduce = (decltype(api.AllReduce))sym("ncclAllReduce");
    api.AllGather = (decltype(api.AllGather))sym("ncclAllGather");
    api.Broadcast = (decltype(api.Broadcast))sym("ncclBroadcast");
    api.GetErrorString = (decltype(api.GetErrorString))sym("ncclGetErrorString");
    api.ok = api.GetUniqueId && api.CommInitRank && api.CommDestroy && api.AllReduce && api.AllGather &&
             api.Broadcast && api.GetErrorString;
    if (!api.ok) api.why = "librccl lacks a required symbol";
  });
  return api;
}

int nccl_fail(const char* what, ncclResult_t r) {
  set_error(std::string(what) + " failed: " + rccl().GetErrorString(r));
  return BIGKRLS_EHIP;
}

#define BK_NCCL(expr)                                  \
  do {                                                 \
    ncclResult_t _r = (expr);                          \
    if (_r != ncclSuccess) return nccl_fail(#expr, _r); \
  } while (0)

int cb_fail(const char* what, int rc) {
  set_error(std::string("collective callback ") + what + " returned " + std::to_string(rc));
  return BIGKRLS_EHIP;
}

}  // namespace

// BIGKRLS_TRACE_DIR (diagnostics, csrc/trace.hip): hash of what a collective is given and of what it delivers.
// Best effort: a trace record that cannot be written (an allocation or launch that fails under the very memory pressure
// the tool is used under) must not keep this rank out of a collective its peers are already waiting in -- the status of
// the record taken BEFORE a collective is only reported after the collective has been issued.
static int comm_trace(bigkrls_comm* comm, const char* tag, const double* p, int64_t count, int64_t extra) {
  if (!trace_on() || !comm->ctx) return BIGKRLS_OK;
  return trace_point(comm->ctx, comm->ctx->stream, tag, p, count, extra);
}

int comm_all_reduce(bigkrls_comm* comm, double* dbuf, int64_t count, int op) {
  if (!comm || count <= 0) return BIGKRLS_OK;
  const int rc_trace = comm_trace(comm, "L:ar_in", dbuf, count, op);
  if (comm->use_cb) {
    if (comm->ctx) BK_HIP(hipStreamSynchronize(comm->ctx->stream));
    const int rc = comm->cb.all_reduce(comm->cb.user, dbuf, count, op);
    if (rc) return cb_fail("all_reduce", rc);
    const int rc_out = comm_trace(comm, "C:ar_out", dbuf, count, op);
    return rc_trace != BIGKRLS_OK ? rc_trace : rc_out;
  }
  BK_NCCL(rccl().AllReduce(dbuf, dbuf, (size_t)count, ncclFloat64, op == COMM_MIN ? ncclMin : ncclSum,
                           (ncclComm_t)comm->nccl, comm->ctx->stream));
  const int rc_out = comm_trace(comm, "C:ar_out", dbuf, count, op);
  return rc_trace != BIGKRLS_OK ? rc_trace : rc_out;
}

int comm_all_gather(bigkrls_comm* comm, const double* dsend, double* drecv, int64_t count) {
  if (!comm || count <= 0) return BIGKRLS_OK;
  const int rc_trace = comm_trace(comm, "L:ag_in", dsend, count, 0);
  if (comm->use_cb) {
    if (comm->ctx) BK_HIP(hipStreamSynchronize(comm->ctx->stream));
    const int rc = comm->cb.all_gather(comm->cb.user, dsend, drecv, count);
    if (rc) return cb_fail("all_gather", rc);
    const int rc_out = comm_trace(comm, "C:ag_out", drecv, count * comm->nranks, 0);
    return rc_trace != BIGKRLS_OK ? rc_trace : rc_out;
  }
  BK_NCCL(rccl().AllGather(dsend, drecv, (size_t)count, ncclFloat64, (ncclComm_t)comm->nccl, comm->ctx->stream));
  const int rc_out = comm_trace(comm, "C:ag_out", drecv, count * comm->nranks, 0);
  return rc_trace != BIGKRLS_OK ? rc_trace : rc_out;
}

int comm_broadcast(bigkrls_comm* comm, double* dbuf, int64_t count, int root) {
  if (!comm || count <= 0) return BIGKRLS_OK;
  const int rc_trace = comm_trace(comm, comm->rank == root ? "L:bc_in_root" : "L:bc_in", dbuf, count, root);
  if (comm->use_cb) {
    if (comm->ctx) BK_HIP(hipStreamSynchronize(comm->ctx->stream));
    const int rc = comm->cb.broadcast(comm->cb.user, dbuf, count, root);
    if (rc) return cb_fail("broadcast", rc);
    const int rc_out = comm_trace(comm, "C:bc_out", dbuf, count, root);
    return rc_trace != BIGKRLS_OK ? rc_trace : rc_out;
  }
  BK_NCCL(rccl().Broadcast(dbuf, dbuf, (size_t)count, ncclFloat64, root, (ncclComm_t)comm->nccl, comm->ctx->stream));
  const int rc_out = comm_trace(comm, "C:bc_out", dbuf, count, root);
  return rc_trace != BIGKRLS_OK ? rc_trace : rc_out;
}

// the device word(s) and the pinned scratch the status agreement goes through: allocated when the communicator is
// created, so that a rank short of memory later can still take part in comm_agree
static int comm_prealloc(bigkrls_ctx* ctx) {
  void* p = nullptr;
  double* hp = nullptr;
  BK_TRY(ws_get(ctx, SLOT_COMM_SMALL, 64 * sizeof(double), &p));
  return pinned_get(ctx, 64, &hp);
}

int comm_all_reduce_host(bigkrls_comm* comm, double* h_vals, int64_t count, int op) {
  if (!comm || count <= 0) return BIGKRLS_OK;
  bigkrls_ctx* ctx = comm->ctx;
  BK_REQUIRE(ctx, "this communicator was created without a device context: it can only be passed to bigkrls_comm_check");
  void* p = nullptr;
  BK_TRY(ws_get(ctx, SLOT_COMM_SMALL, std::max<int64_t>(count, 64) * sizeof(double), &p));
  double* hp = nullptr;
  BK_TRY(pinned_get(ctx, count, &hp));
  std::memcpy(hp, h_vals, (size_t)count * sizeof(double));
  BK_HIP(hipMemcpyAsync(p, hp, (size_t)count * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  BK_TRY(comm_all_reduce(comm, (double*)p, count, op));
  BK_HIP(hipMemcpyAsync(hp, p, (size_t)count * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  BK_HIP(hipStreamSynchronize(ctx->stream));
  std::memcpy(h_vals, hp, (size_t)count * sizeof(double));
  return BIGKRLS_OK;
}

int comm_agree(bigkrls_comm* comm, int local_status) {
  if (!comm) return local_status;
  // the all-reduce itself must not be skipped by a failed rank: its peers are waiting in it
  std::string local_msg = local_status != BIGKRLS_OK ? std::string(bigkrls_last_error()) : std::string();
  double v = -(double)local_status;      // MIN of the negated codes = the largest code
  const int rc = comm_all_reduce_host(comm, &v, 1, COMM_MIN);
  if (rc != BIGKRLS_OK) {                // the collective itself failed: nothing was agreed
    if (local_status != BIGKRLS_OK) set_error(local_msg);
    return local_status != BIGKRLS_OK ? local_status : rc;
  }
  // EVERY rank returns the same (largest) code, also one that failed locally with a smaller one: what the caller
  // does next -- give up, or replay the decomposition after BK_EWATCHDOG -- must be the same decision everywhere
  const int worst = (int)(-v + 0.5);
  if (worst == BIGKRLS_OK) return BIGKRLS_OK;
  if (local_status == worst) set_error(local_msg);
  else if (local_status != BIGKRLS_OK)
    set_error(local_msg + " (and another rank of the multi-GPU fit failed with status " + std::to_string(worst) + ")");
  else set_error("another rank of the multi-GPU fit failed with status " + std::to_string(worst));
  return worst;
}

void dist_partition(int64_t n, int world, int64_t align, int rank, int64_t* nb, int64_t* r0, int64_t* r1) {
  int64_t b = (n + world - 1) / world;
  b = (b + align - 1) / align * align;
  if (nb) *nb = b;
  if (r0) *r0 = std::min<int64_t>((int64_t)rank * b, n);
  if (r1) *r1 = std::min<int64_t>((int64_t)(rank + 1) * b, n);
}

int comm_gather_rows(bigkrls_comm* comm, const double* local, int64_t nloc, int64_t ldl, int64_t cols, int64_t nb,
                     int64_t n, double* out, int64_t ldo) {
  bigkrls_ctx* ctx = comm->ctx;
  const int world = comm->nranks;
  void* p = nullptr;
  BK_TRY(ws_get(ctx, SLOT_COMM_STAGE, (int64_t)(world + 1) * nb * cols * sizeof(double), &p));
  double* send = (double*)p;
  double* recv = send + nb * cols;
  if (nloc < nb) BK_HIP(hipMemsetAsync(send, 0, (size_t)(nb * cols) * sizeof(double), ctx->stream));
  if (nloc > 0) BK_TRY(copy_matrix(ctx, local, nloc, cols, ldl, send, nb));
  BK_TRY(comm_all_gather(comm, send, recv, nb * cols));
  for (int r = 0; r < world; ++r) {
    const int64_t lo = (int64_t)r * nb, rows = std::min<int64_t>(nb, n - lo);
    if (rows <= 0) break;
    BK_TRY(copy_matrix(ctx, recv + (int64_t)r * nb * cols, rows, cols, nb, out + lo, ldo));
  }
  return BIGKRLS_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// Dense symmetric eigendecomposition with stage 1 (dense -> band, 4/3 N^3 flops) partitioned by column blocks
// ("Eigen, dense tridiagonalisation"). Per 64-column panel: one broadcast of the panel
// strip from its owner, the replicated panel QR, this rank's contribution to Y = A22 V (its own columns times its own
// rows of V: the plain product along the block's contiguous dimension), one all-reduce of Y (N x 64), the replicated
// thin products, and the update of the own columns --
// with look-ahead: the columns of the NEXT panel are updated first by their owner, its strip is broadcast and its
// factorisation started on the look-ahead stream beside the update of the remaining columns. The reduced matrix
// ends up replicated; stage 2 and the divide & conquer are replicated (latency-bound, no flops to share), the
// back-transform is split by eigenvector column and assembled with an all-gather of the column blocks -- the RCCL
// exchange north_star names. A rank that fails locally keeps taking part in the collectives of the loop (its peers
// are inside them) and the failure is agreed on before the next phase.
// ---------------------------------------------------------------------------------------------------------------
int eigen_dense_dist(bigkrls_comm* comm, double* A, int64_t n, int64_t nb, int64_t neig, double eigtrunc,
                     double* dvals, double* dQ, int64_t* h_lastkeeper) {
  bigkrls_ctx* ctx = comm->ctx;
  hipStream_t st = ctx->stream;
  const int world = comm->nranks, rank = comm->rank;
  constexpr int64_t b = 64;
  BK_REQUIRE(nb % b == 0 && n > 4 * b, "eigen_dense_dist: the column blocks must be multiples of 64 and n > 256");
  const int64_t c0 = std::min<int64_t>((int64_t)rank * nb, n), c1 = std::min<int64_t>((int64_t)(rank + 1) * nb, n);
  const int64_t ncl = c1 - c0;
  int live = comm_agree(comm, dist_s1_open(ctx, n));
  BK_TRY(live);
  {
    // panels per trailing update: derived from rank-local state (environment, workspace), and it decides the sequence
    // of collectives below -- every rank runs the schedule of the most restrictive one
    double mode = (double)dist_s1_local_agg_mode(ctx, n);
    BK_TRY(comm_all_reduce_host(comm, &mode, 1, COMM_MIN));
    BK_TRY(dist_s1_set_agg_mode(ctx, n, (int)mode));
  }
  void* pm = nullptr;
  // strip (n x 64), Y (n x 64)
  BK_TRY(comm_agree(comm, ws_get(ctx, SLOT_DIST_MISC, 2 * n * b * sizeof(double), &pm)));
  double* sbuf = (double*)pm;
  double* Y = sbuf + n * b;
  auto has_panel = [&](int64_t k) { return k + b < n && n - k - b > 1; };
  int status = BIGKRLS_OK;       // first local failure; collectives keep running
  auto local = [&](int rc) {
    if (rc != BIGKRLS_OK && status == BIGKRLS_OK) status = rc;
  };
  // rows k..n of the global columns k..k+w (inside one owner's block) on every rank, as an (n - k) x w matrix
  auto bcast_strip = [&](int64_t k, int64_t w) -> int {
    const int owner = (int)(k / nb);
    if (owner == rank && status == BIGKRLS_OK) local(copy_matrix(ctx, A + (k - c0) * n + k, n - k, w, n, sbuf, n - k));
    return comm_broadcast(comm, sbuf, w * (n - k), owner);
  };
  int64_t k = 0;
  if (has_panel(0)) {
    BK_TRY(bcast_strip(0, b));
    if (status == BIGKRLS_OK) local(dist_s1_panel(ctx, n, 0, sbuf));
  }
  // ---- four / two panels per trailing update while the trailing matrix is large (the single-GPU loop's aggregation
  //      without its pieces): the update of a group is applied after its last panel; the later panels' products run on
  //      the stale column blocks and are corrected with the group's pending blocks (replicated thin products)
  for (int G = dist_s1_group_size(ctx, n, k); G > 0; G = dist_s1_group_size(ctx, n, k)) {
    const int64_t k0 = k;
    for (int j = 0; j < G; ++j, k += b) {
      const int64_t m = n - k - b;
      const int64_t la0 = std::min<int64_t>(std::max<int64_t>(k + b - c0, 0), ncl);
      const int64_t nact = ncl - la0;
      double* Aact = A + la0 * n + (k + b);
      const int64_t row0 = nact > 0 ? (c0 + la0) - (k + b) : 0;
      if (status == BIGKRLS_OK) local(dist_s1_av(ctx, n, k, Aact, n, nact, row0, Y, m));
      BK_TRY(comm_all_reduce(comm, Y, m * b, COMM_SUM));
      if (status == BIGKRLS_OK) local(dist_s1_thin_group(ctx, n, k, Y, k0));
      if (status == BIGKRLS_OK) local(dist_s1_trace(ctx, n, k));
      const int64_t nxt = k + b;
      int64_t first = 0;
      if (has_panel(nxt)) {
        if (nxt / nb == rank) {                          // the next panel's columns: every pending block, before the strip leaves
          first = std::min<int64_t>(b, nact);
          if (status == BIGKRLS_OK) local(dist_s1_update_cols_group(ctx, n, k, Aact, n, first, row0, k0, j + 1));
        }
        BK_TRY(bcast_strip(nxt, b));
        if (status == BIGKRLS_OK) local(dist_s1_panel_begin(ctx, n, nxt, sbuf));
      }
      if (j == G - 1 && status == BIGKRLS_OK)            // the group's update of the own columns, beside the next factorisation
        local(dist_s1_update_cols_group(ctx, n, k, Aact + first * n, n, nact - first, row0 + first, k0, G));
    }
  }
  while (has_panel(k)) {
    const int64_t m = n - k - b;
    const int64_t la0 = std::min<int64_t>(std::max<int64_t>(k + b - c0, 0), ncl);   // first own column inside the trailing matrix
    const int64_t nact = ncl - la0;
    double* Aact = A + la0 * n + (k + b);
    const int64_t row0 = nact > 0 ? (c0 + la0) - (k + b) : 0;
    // Y = A22 V as the sum over the ranks of A22[:, own columns] V[own rows]   (waits for panel k's factorisation)
    if (status == BIGKRLS_OK) local(dist_s1_av(ctx, n, k, Aact, n, nact, row0, Y, m));
    BK_TRY(comm_all_reduce(comm, Y, m * b, COMM_SUM));
    if (status == BIGKRLS_OK) local(dist_s1_thin(ctx, n, k, Y));
    if (status == BIGKRLS_OK) local(dist_s1_trace(ctx, n, k));
    const int64_t nxt = k + b;
    int64_t first = 0;                                   // own columns already updated before the look-ahead
    if (has_panel(nxt)) {
      if (nxt / nb == rank) {                            // the next panel's columns are the first active ones of their owner
        first = std::min<int64_t>(b, nact);
        if (status == BIGKRLS_OK) local(dist_s1_update_cols(ctx, n, k, Aact, n, first, row0));
      }
      BK_TRY(bcast_strip(nxt, b));
      if (status == BIGKRLS_OK) local(dist_s1_panel_begin(ctx, n, nxt, sbuf));
    }
    if (status == BIGKRLS_OK) local(dist_s1_update_cols(ctx, n, k, Aact + first * n, n, nact - first, row0 + first));
    k += b;
  }
  while (k < n) {                                        // what is left of the trailing matrix: not panels
    const int64_t owner_end = std::min<int64_t>((k / nb + 1) * nb, n);
    const int64_t w = std::min<int64_t>(b, owner_end - k);
    BK_TRY(bcast_strip(k, w));
    if (status == BIGKRLS_OK) local(dist_s1_put(ctx, n, k, sbuf, w));
    k += w;
  }
  BK_TRY(comm_agree(comm, status));
  // stage 2, divide & conquer and this rank's slice of the back-transform; a fired watchdog of a persistent kernel
  // (this rank's or another's) comes back as BK_EWATCHDOG on EVERY rank: the caller replays the decomposition
  int64_t nv = 0;
  void* pq = nullptr;
  status = eigen(ctx, nullptr, n, n, neig, dvals, neig, eigtrunc, dQ, n, &nv, rank, world, EIG_RESUME);
  BK_TRY(comm_agree(comm, status));
  {
    // The decomposition is replicated with deterministic kernels, so every rank keeps the same number of pairs; the
    // exchange below is sized by it. Checked rather than assumed: a rank that disagrees (memory corruption, a GPU
    // fault) must come back as an error on every rank, not as a collective with mismatched sizes.
    double mm[2] = {(double)nv, -(double)nv};
    BK_TRY(comm_all_reduce_host(comm, mm, 2, COMM_MIN));
    if (mm[0] != -mm[1]) {
      set_error("eigen (distributed): the ranks disagree on the number of kept eigenpairs (" + std::to_string((long long)mm[0]) +
                " ... " + std::to_string((long long)-mm[1]) + "); the replicated decomposition was not reproduced bit for bit");
      return BK_EWATCHDOG;      // every rank sees the same two numbers: all of them replay (csrc/fit.hip), none gives up alone
    }
  }
  if (h_lastkeeper) *h_lastkeeper = nv;
  if (world == 1 || nv == 0) return BIGKRLS_OK;
  // all-gather of the back-transformed column blocks (rank r holds columns nv r / world .. nv (r + 1) / world)
  int64_t pmax = 0;
  for (int r = 0; r < world; ++r) pmax = std::max<int64_t>(pmax, nv * (r + 1) / world - nv * r / world);
  BK_TRY(comm_agree(comm, ws_get(ctx, SLOT_COMM_STAGE, (int64_t)(world + 1) * pmax * n * sizeof(double), &pq)));
  double* send = (double*)pq;
  double* recv = send + pmax * n;
  const int64_t my0 = nv * rank / world, mine = nv * (rank + 1) / world - my0;
  BK_HIP(hipMemsetAsync(send, 0, (size_t)(pmax * n) * sizeof(double), st));
  if (mine > 0) BK_HIP(hipMemcpyAsync(send, dQ + my0 * n, (size_t)(mine * n) * sizeof(double), hipMemcpyDeviceToDevice, st));
  BK_TRY(comm_all_gather(comm, send, recv, pmax * n));
  for (int r = 0; r < world; ++r) {
    const int64_t q0 = nv * r / world, cnt = nv * (r + 1) / world - q0;
    if (cnt > 0)
      BK_HIP(hipMemcpyAsync(dQ + q0 * n, recv + (int64_t)r * pmax * n, (size_t)(cnt * n) * sizeof(double),
                            hipMemcpyDeviceToDevice, st));
  }
  return BIGKRLS_OK;
}

}  // namespace bk

using namespace bk;

extern "C" {

int bigkrls_comm_unique_id(void* id_out) {
  BK_REQUIRE(id_out, "comm_unique_id: null output");
  static_assert(sizeof(ncclUniqueId) == BIGKRLS_UNIQUE_ID_BYTES, "unique id size");
  if (!rccl().ok) {
    set_error("RCCL is not available: " + rccl().why);
    return BIGKRLS_ENODEVICE;
  }
  ncclUniqueId id;
  BK_NCCL(rccl().GetUniqueId(&id));
  std::memcpy(id_out, &id, sizeof id);
  return BIGKRLS_OK;
}

int bigkrls_comm_create(bigkrls_ctx* ctx, int32_t nranks, int32_t rank, const void* unique_id, bigkrls_comm** out) {
  BK_REQUIRE(ctx && unique_id && out, "comm_create: null argument");
  BK_REQUIRE(nranks >= 1 && rank >= 0 && rank < nranks, "comm_create: bad rank / nranks");
  if (!rccl().ok) {
    set_error("RCCL is not available: " + rccl().why);
    return BIGKRLS_ENODEVICE;
  }
  BK_HIP(hipSetDevice(ctx->device));
  ncclUniqueId id;
  std::memcpy(&id, unique_id, sizeof id);
  BK_TRY(comm_prealloc(ctx));
  ncclComm_t c = nullptr;
  BK_NCCL(rccl().CommInitRank(&c, nranks, id, rank));
  bigkrls_comm* comm = new bigkrls_comm();
  comm->ctx = ctx;
  comm->nranks = nranks;
  comm->rank = rank;
  comm->nccl = (void*)c;
  (void)trace_host("L:comm_rccl", nullptr, 0, (int64_t)rank * 1000 + nranks);
  *out = comm;
  return BIGKRLS_OK;
}

int bigkrls_comm_create_callbacks(bigkrls_ctx* ctx, int32_t nranks, int32_t rank, const bigkrls_collectives* table,
                                  bigkrls_comm** out) {
  BK_REQUIRE(table && out, "comm_create_callbacks: null argument");
  BK_REQUIRE(table->struct_bytes == (int64_t)sizeof(bigkrls_collectives), "comm_create_callbacks: table struct size mismatch");
  BK_REQUIRE(table->all_reduce && table->all_gather && table->broadcast, "comm_create_callbacks: a callback is missing");
  BK_REQUIRE(nranks >= 1 && rank >= 0 && rank < nranks, "comm_create_callbacks: bad rank / nranks");
  if (ctx) BK_TRY(comm_prealloc(ctx));
  bigkrls_comm* comm = new bigkrls_comm();
  comm->ctx = ctx;
  comm->nranks = nranks;
  comm->rank = rank;
  comm->use_cb = true;
  comm->cb = *table;
  (void)trace_host("L:comm_callbacks", nullptr, 0, (int64_t)rank * 1000 + nranks);
  *out = comm;
  return BIGKRLS_OK;
}

int bigkrls_comm_destroy(bigkrls_comm* comm) {
  if (!comm) return BIGKRLS_OK;
  if (comm->nccl) {
    if (comm->ctx) (void)hipStreamSynchronize(comm->ctx->stream);
    (void)rccl().CommDestroy((ncclComm_t)comm->nccl);
  }
  delete comm;
  return BIGKRLS_OK;
}

int bigkrls_comm_forget_context(bigkrls_comm* comm) {
  BK_REQUIRE(comm, "comm_forget_context: null communicator");
  comm->ctx = nullptr;
  return BIGKRLS_OK;
}

int bigkrls_comm_rank(bigkrls_comm* comm, int32_t* rank, int32_t* nranks) {
  BK_REQUIRE(comm, "comm_rank: null communicator");
  if (rank) *rank = comm->rank;
  if (nranks) *nranks = comm->nranks;
  return BIGKRLS_OK;
}

int bigkrls_comm_check(bigkrls_comm* comm, double* buf, int64_t count) {
  BK_REQUIRE(comm && buf && count > 0, "comm_check: bad arguments");
  BK_TRY(comm_all_reduce(comm, buf, count, COMM_SUM));
  BK_TRY(comm_all_reduce(comm, buf + count, count, COMM_MIN));
  BK_TRY(comm_all_gather(comm, buf + 2 * count, buf + 4 * count, count));
  BK_TRY(comm_broadcast(comm, buf + 3 * count, count, comm->nranks - 1));
  if (comm->ctx && !comm->use_cb) BK_HIP(hipStreamSynchronize(comm->ctx->stream));
  return BIGKRLS_OK;
}

}  // extern "C"
