// Shared declarations for libbigkrls_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include <string>
#include <utility>
#include <vector>

#include "../../include/bigkrls.h"

namespace bk {

void set_error(const std::string& msg);

#define BK_HIP(expr)                                                                   \
  do {                                                                                 \
    hipError_t _e = (expr);                                                            \
    if (_e != hipSuccess) {                                                            \
      bk::set_error(std::string(#expr) + " failed: " + hipGetErrorString(_e) + " at " + \
                    __FILE__ + ":" + std::to_string(__LINE__));                        \
      return (_e == hipErrorOutOfMemory) ? BIGKRLS_ENOMEM : BIGKRLS_EHIP;              \
    }                                                                                  \
  } while (0)

#define BK_TRY(expr)             \
  do {                           \
    int _s = (expr);             \
    if (_s != BIGKRLS_OK) return _s; \
  } while (0)

#define BK_CHECK_LAUNCH() BK_HIP(hipGetLastError())

#define BK_REQUIRE(cond, msg)                       \
  do {                                              \
    if (!(cond)) {                                  \
      bk::set_error(std::string("invalid argument: ") + msg); \
      return BIGKRLS_EINVAL;                        \
    }                                               \
  } while (0)

}  // namespace bk

// The context: one device, one stream, a pool of reusable workspace slabs.
struct bigkrls_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  bool owns_stream = false;
  // side stream + events for look-ahead inside the eigensolver (panel QR of the next block
  // column runs concurrently with the rest of the trailing update); created on first use
  // per-context (== per-device) launch set-up that must not be cached process-wide: kernels whose
  // dynamic-LDS limit was raised, and co-resident workgroup capacities of the persistent kernels
  std::vector<const void*> dyn_smem_done;
  std::vector<std::pair<const void*, int>> resident_cap;
  // set while a decomposition is retried after the watchdog of a persistent kernel fired: the
  // panel QR runs one launch per column and the bulge chasing one launch per wavefront
  bool no_resident = false;
  // state of a row-block distributed stage 1 between bigkrls_dev_s1_open and bigkrls_dev_eigen_resume
  void* dist_s1 = nullptr;
  void (*dist_s1_free)(void*) = nullptr;
  hipStream_t side_stream = nullptr;
  // lowest-priority stream for work nobody waits for soon (what is precomputed for the back-transforms while the divide
  // & conquer runs: its kernels must not take compute units from the critical path's); created with the side stream
  hipStream_t bg_stream = nullptr;
  bool side_is_main = false;   // BIGKRLS_NO_SIDE (diagnostics): side_stream is the main stream itself
  hipEvent_t ev_fork = nullptr, ev_join = nullptr, ev_join2 = nullptr, ev_pq = nullptr;
  // workspace slots: slot i is grown on demand and reused across calls
  static constexpr int kSlots = 48;
  void* ws[kSlots] = {nullptr};
  int64_t ws_bytes[kSlots] = {0};
  // pinned host scratch for small scalar read-backs
  double* h_pinned = nullptr;
  int64_t h_pinned_doubles = 0;
  // pinned arena the small host -> device uploads of the divide & conquer go through (PinnedStage)
  char* h_stage = nullptr;
  int64_t h_stage_bytes = 0;
  // host copy of the standardised X of the last fit (kept between fits: 40 MB at N = 100 000, P = 50, whose allocation
  // and first touch cost milliseconds per call)
  std::vector<double> h_xs;
  // ... and the small pinned buffer the group table of the stage-1 back-transform's precompute is uploaded from
  char* h_plan = nullptr;
  int64_t h_plan_bytes = 0;
  // optional HIP-event sampling of named kernels (bench.py roofline numbers)
  bool profile = false;
  struct ProfSample { hipEvent_t e0, e1; double work; };
  struct ProfEntry { std::string name; std::vector<ProfSample> pending; double ms = 0, work = 0; int64_t launches = 0; };
  std::vector<ProfEntry> prof;
  std::vector<hipEvent_t> prof_pool;   // recycled timing events (creating two per sample costs more than the sample)
  // recovery paths taken since the context was created (bigkrls_ctx_get_counters): decompositions redone because the
  // fit's check against K failed / replayed with per-step launches after a watchdog (or a disagreement between ranks) /
  // replicated eigenvalues of this rank that were not rank 0's bit for bit (multi-GPU fits)
  int64_t n_redone = 0, n_replayed = 0, n_replica_diff = 0;
  // set by the eigensolver beside an error code when what failed can only be a fault of the run on a finite symmetric
  // input (a block recurrence that does not hold against K, non-finite entries after the tridiagonalisation); read and
  // cleared by the fit, which validated its input and redoes such a decomposition once (csrc/fit.hip)
  bool corrupt_run = false;
  // set by the fit around a decomposition it is going to verify against K itself (ALL kept pairs, csrc/fit.hip): the
  // block Lanczos then leaves out its own sample check of the last block of Ritz pairs against K (one more K-times-
  // block product: 13 ms at N = 50 000, 50 ms at N = 100 000); not set for the redo after a failed check
  bool caller_verifies = false;
  // device-side predicate of the next gemm() launches (kernel and split-K reduction return at once while *gemm_run_if
  // == 0): the T-factor chain of a stage-1 panel only runs when pq_chol left the panel to the Householder kernel
  const int* gemm_run_if = nullptr;
  // the stage-1 panel loop of the last size, captured as a hipGraph (stage1_run, csrc/eigen.hip); valid
  // while the workspace it points into has not been reallocated (ws_generation)
  void* s1_graph_exec = nullptr;
  int s1_graph_n = 0, s1_graph_warm_n = 0, s1_graph_seen = 0;
  hipStream_t graph_stream = nullptr;   // (contexts on the default stream: where the graph is captured and replayed)
  hipEvent_t ev_graph = nullptr;
  int64_t s1_graph_gen = -1, ws_generation = 0;
  const void* s1_graph_W = nullptr;
};

// One rank of a multi-GPU job (one process per GPU): the context it computes on and the collectives that connect
// it to its peers -- RCCL (ncclCommInitRank from a caller-distributed unique id; librccl is opened at run time) or a
// caller-supplied table of callbacks (tests: host-staged collectives that let several ranks share one GPU).
struct bigkrls_comm {
  bigkrls_ctx* ctx = nullptr;
  int nranks = 1, rank = 0;
  void* nccl = nullptr;          // ncclComm_t
  bool use_cb = false;
  bigkrls_collectives cb{};
};

namespace bk {

enum CommOp { COMM_SUM = 0, COMM_MIN = 1 };
// collectives on device buffers of doubles, ordered on comm->ctx->stream (in place for the reduction)
int comm_all_reduce(bigkrls_comm* comm, double* dbuf, int64_t count, int op);
int comm_all_gather(bigkrls_comm* comm, const double* dsend, double* drecv, int64_t count_per_rank);
int comm_broadcast(bigkrls_comm* comm, double* dbuf, int64_t count, int root);
// the same on a few host scalars (staged through the context's pinned buffer; synchronises the stream)
int comm_all_reduce_host(bigkrls_comm* comm, double* h_vals, int64_t count, int op);
// every rank returns the worst of the ranks' local status codes (one MIN all-reduce): no rank enters the next
// collective while a peer has already failed
int comm_agree(bigkrls_comm* comm, int local_status);
// equal row blocks of nb = ceil(n / world) rows rounded up to a multiple of `align`; the last ranks may be short or empty
void dist_partition(int64_t n, int world, int64_t align, int rank, int64_t* nb, int64_t* r0, int64_t* r1);
// out (n x cols, ldo) = the row blocks `local` (nloc x cols, ldl) of all ranks stacked in rank order
int comm_gather_rows(bigkrls_comm* comm, const double* local, int64_t nloc, int64_t ldl, int64_t cols, int64_t nb,
                     int64_t n, double* out, int64_t ldo);
// dense eigensolver with stage 1 partitioned by column blocks (A = K[:, c0:c1], n x ncl, ld n, overwritten);
// vals (device, neig), Q (device, n x neig, ld n) end up replicated
int eigen_dense_dist(bigkrls_comm* comm, double* A, int64_t n, int64_t nb, int64_t neig, double eigtrunc,
                     double* dvals, double* dQ, int64_t* h_lastkeeper);

// workspace slot ids (each caller family uses its own so nested calls never alias)
enum Slot {
  SLOT_GEMM_SPLITK = 0,
  SLOT_NORMS_A = 1,
  SLOT_NORMS_B = 2,
  SLOT_SOLVE_PART = 3,
  SLOT_SCALAR = 4,
  SLOT_DERIV_B = 5,
  SLOT_DERIV_KB = 6,
  SLOT_DERIV_T = 7,
  SLOT_EIG_A = 8,
  SLOT_EIG_MISC = 9,
  SLOT_EIG_Q0 = 10,
  SLOT_EIG_Q1 = 11,
  SLOT_EIG_U = 12,
  SLOT_EIG_VEC = 13,
  SLOT_L1_A = 14,
  SLOT_L1_B = 15,
  SLOT_EIG_PANEL = 16,
  SLOT_EIG_INT = 17,
  SLOT_EIG_BT = 18,
  SLOT_EIG_Z = 19,
  SLOT_EIG_DESC = 20,
  SLOT_EIG_VV = 21,
  SLOT_KRY_B = 22,
  SLOT_KRY_W = 23,
  SLOT_KRY_C = 24,
  SLOT_KRY_T = 25,
  SLOT_KRY_Y = 26,
  SLOT_SIDE_SPLITK = 27,   // split-K partials of GEMMs issued on the look-ahead stream
  SLOT_EIG_T2 = 28,        // compact-WY T factors of the stage-2 back-transform tasks
  SLOT_EIG_VBIG = 29,      // merged reflector blocks of the stage-1 back-transform
  SLOT_EIG_TBIG = 30,      // ... and their T factors
  SLOT_FIT_SMALL = 31,     // bigkrls_fit / bigkrls_predict: X, y, eigenvalues, c, yhat, D, S, ...
  SLOT_FIT_Q = 32,         // ... eigenvectors (n x Neig)
  SLOT_FIT_M = 33,         // ... Q diag(w) / K_new V
  SLOT_FIT_K = 34,         // ... the kernel when the caller does not want it back
  SLOT_EIG_AGG = 35,       // stage 1: reflector blocks of the panel groups whose trailing update is pending
  SLOT_COMM_STAGE = 36,    // multi-GPU: send / receive staging of the row-block all-gathers
  SLOT_COMM_SMALL = 37,    // ... status words and scalars that are all-reduced
  SLOT_DIST_A = 38,        // ... the working copy of this rank's column block of K in the partitioned stage 1
  SLOT_DIST_MISC = 39,     // ... panel strips and the Y = A22 V exchange buffers
  SLOT_DIST_K = 40,        // ... this rank's column block of K when the caller does not want it back
  SLOT_DIST_V = 41,        // ... column blocks of the variance matrices (same)
  SLOT_EIG_FLAGS = 42,     // completion flags of the persistent stage-2 back-transform's tasks
  SLOT_FIT_VERIFY = 43,    // the fit's check of a decomposition against K: Q r, Q (lambda o r), K Q r
};

int ws_get(bigkrls_ctx* ctx, int slot, int64_t nbytes, void** out);
bool ws_poison();                      // BIGKRLS_POISON (diagnostics): fresh workspace starts as all-ones bytes ...
int ws_poison_all(bigkrls_ctx* ctx);   // ... and every slab is reset to them at the start of a fit
// ---- trace.hip: diagnostic hashes of buffers (BIGKRLS_TRACE_DIR; off otherwise) -----------------
bool trace_on();
bool trace_fine();   // ... and BIGKRLS_TRACE_FINE: per-panel / per-level hashes inside the eigensolver (serialises it)
// hash of `count` doubles (or 8-byte words) at dev_ptr, computed on `st` (synchronised), appended to the process's trace
int trace_point(bigkrls_ctx* ctx, hipStream_t st, const char* tag, const void* dev_ptr, int64_t count, int64_t extra = 0);
int trace_host(const char* tag, const void* host_ptr, int64_t count, int64_t extra = 0);
// Bracket one launch with HIP events when ctx->profile is on: call prof_begin before the
// launch and prof_end after it; `work` is the launch's algorithmic bytes (or flops).
int prof_begin(bigkrls_ctx* ctx, const char* name, double work, hipStream_t stream = nullptr);
int prof_end(bigkrls_ctx* ctx, const char* name, hipStream_t stream = nullptr);
int pinned_get(bigkrls_ctx* ctx, int64_t ndoubles, double** out);
// Device -> host read-backs of a few small arrays through the context's pinned buffer: add() enqueues a copy into
// the next slice, finish() synchronises the stream and moves the slices to their destinations. (An asynchronous
// copy into pageable memory makes the runtime pin and unpin the user pages; inside loops that showed up as
// sporadic stalls of 1-2 s.) `capacity_doubles` must cover everything added before finish().
class PinnedFetch {
 public:
  PinnedFetch(bigkrls_ctx* ctx, int64_t capacity_doubles) : ctx_(ctx), cap_(capacity_doubles) {}
  int add(void* host_dst, const void* dev_src, size_t bytes);
  int finish();

 private:
  struct Item { void* dst; size_t off, bytes; };
  bigkrls_ctx* ctx_;
  int64_t cap_;
  double* base_ = nullptr;
  size_t used_ = 0;   // doubles
  std::vector<Item> items_;
};

// Host -> device uploads of small arrays from a pinned arena owned by the context (the reverse of PinnedFetch). An
// asynchronous copy from pageable memory is staged by the runtime call by call (and is the one runtime path this
// library has seen misbehave under load); here the source is pinned memory the caller fills in
// place (alloc) or that put() copies into. A slice stays untouched until reset(), which the caller invokes only after
// the stream has been synchronised. fixed(): slices that survive reset() (allocated before the first put / alloc).
class PinnedStage {
 public:
  explicit PinnedStage(bigkrls_ctx* ctx) : ctx_(ctx) {}
  int reserve(size_t bytes);                  // synchronises the stream if the arena has to grow; empties the arena
  void* fixed(size_t bytes);                  // 64-byte aligned, kept across reset(); nullptr when the arena is full
  void* alloc(size_t bytes);                  // 64-byte aligned slice of this cycle; nullptr when the arena is full
  void reset() { used_ = fixed_; }
  int send(void* dev_dst, const void* slice, size_t bytes);           // slice -> device, asynchronous on ctx->stream
  int put(void* dev_dst, const void* host_src, size_t bytes);         // copy into a fresh slice, then send

 private:
  bigkrls_ctx* ctx_;
  size_t used_ = 0, fixed_ = 0;
};

// ---- gemm.hip -----------------------------------------------------------------
int gemm(bigkrls_ctx* ctx, int ta, int tb, int64_t m, int64_t n, int64_t k, double alpha,
         const double* A, int64_t lda, const double* B, int64_t ldb, double beta, double* C,
         int64_t ldc);
// C (m x n, n <= 48) = A (m x k) B (k x n), both not transposed: the 128 x 48 tile of the marginal-effects pass
int gemm_nn_skinny48(bigkrls_ctx* ctx, int64_t m, int64_t n, int64_t k, const double* A, int64_t lda, const double* B,
                     int64_t ldb, double* C, int64_t ldc);
int kernel_block(bigkrls_ctx* ctx, const double* A, int64_t u, int64_t lda, const double* B,
                 int64_t v, int64_t ldb, int64_t p, double sigma, double* out, int64_t ldo,
                 int64_t diag_shift);

int syrk_lower(bigkrls_ctx* ctx, int64_t m, int64_t k, double alpha, const double* A, int64_t lda,
               const double* B, int64_t ldb, double* C, int64_t ldc);

// tile columns [tn_begin, tn_end) of the lower tile triangle only (tn_end < 0: all of them)
int syrk_mirror(bigkrls_ctx* ctx, int64_t m, int64_t k, double alpha, const double* A, int64_t lda,
                const double* B, int64_t ldb, double* C, int64_t ldc, int tn_begin = 0,
                int tn_end = -1, bool narrow_tiles = false, bool skip_first_column = false);
// the same update restricted to the 64-wide columns [c64_begin, c64_end) (128 x 64 tiles)
int syrk_mirror_cols(bigkrls_ctx* ctx, int64_t m, int64_t k, double alpha, const double* A, int64_t lda,
                     const double* B, int64_t ldb, double* C, int64_t ldc, int c64_begin, int c64_end);
// C = alpha A B' where the product is symmetric (A = Q diag(w), B = Q): half the MFMA work of the GEMM, exactly
// symmetric result
int syrk_mirror_set(bigkrls_ctx* ctx, int64_t m, int64_t k, double alpha, const double* A, int64_t lda,
                    const double* B, int64_t ldb, double* C, int64_t ldc);
int side_stream_get(bigkrls_ctx* ctx);
// raise a kernel's dynamic shared-memory limit once per context (device)
int ensure_dyn_smem(bigkrls_ctx* ctx, const void* kernel, size_t bytes);
// co-resident workgroups of `kernel` (`threads` per workgroup, static LDS only) on this context's device
int resident_capacity(bigkrls_ctx* ctx, const void* kernel, int* cap, int threads = 256);

// effective sample size from the mean absolute pairwise row correlation (src/Neffective.cpp)
int neffective(bigkrls_ctx* ctx, const double* X, int64_t n, int64_t ldx, int64_t p, double* h_out);

// batched GEMM for the divide & conquer merges: per-problem descriptors on device
struct GemmDesc {
  const double* A;
  const double* B;
  double* C;
  int64_t lda, ldb, ldc;
  int32_t m, n, k;
  int32_t pad;
  const int* kidx;  // optional: column gather for A (absolute column offsets from A)
};
int gemm_batched_nn(bigkrls_ctx* ctx, const GemmDesc* d_descs, int n_batch, int max_m, int max_n);

// ---- vecops.hip ---------------------------------------------------------------
int gemv(bigkrls_ctx* ctx, int trans, int64_t m, int64_t n, double alpha, const double* A,
         int64_t lda, const double* x, double beta, double* y);
int dot_host(bigkrls_ctx* ctx, int64_t n, const double* x, const double* y, double* h_out);
int multdiag(bigkrls_ctx* ctx, const double* A, int64_t n, int64_t k, int64_t lda,
             const double* diag, double* out, int64_t ldo);
int diag_extract(bigkrls_ctx* ctx, const double* A, int64_t n, int64_t lda, double* out);
int scale(bigkrls_ctx* ctx, int64_t n, double alpha, double* x);
int row_sqnorms(bigkrls_ctx* ctx, const double* A, int64_t n, int64_t p, int64_t lda, double* out);
int copy_matrix(bigkrls_ctx* ctx, const double* A, int64_t m, int64_t n, int64_t lda, double* B,
                int64_t ldb);

// ---- solveforc.hip ------------------------------------------------------------
int qty(bigkrls_ctx* ctx, const double* Q, int64_t n, int64_t k, int64_t ldq, const double* y,
        double* a);
int solveforc(bigkrls_ctx* ctx, const double* Q, int64_t n_rows, int64_t k, int64_t ldq,
              const double* d, const double* a, double lambda, double* c, double* h_Le);
int lambda_bounds(const double* vals, int64_t n_vals, int64_t n, double* L, double* U);
// Q: the n_rows x k row block this rank holds of the n_total x k eigenvector matrix (comm == nullptr: all of it)
int lambda_search(bigkrls_ctx* ctx, const double* Q, int64_t n_rows, int64_t k, int64_t ldq,
                  const double* d, const double* a, const double* h_vals_all, int64_t n_vals,
                  double L, double U, double tol, double* h_lambda, int64_t* h_nprobes,
                  double* h_trace, int64_t max_trace, bigkrls_comm* comm = nullptr, int64_t n_total = 0);

// ---- deriv.hip ----------------------------------------------------------------
int deriv_rows(bigkrls_ctx* ctx, const double* Krows, int64_t n, int64_t n_rows, int64_t ldk,
               int64_t row0, const double* X, int64_t p, int64_t ldx, const int32_t* h_is_binary,
               const double* c, double sigma, double* D, int64_t ldd, double* S, int64_t lds, double* kc_out = nullptr,
               const double* extra = nullptr, int64_t n_extra = 0, double* extra_out = nullptr);
int deriv_var(bigkrls_ctx* ctx, const double* Q, int64_t n, int64_t k, int64_t ldq,
              const double* wv, const double* S, int64_t p, int64_t lds, const double* h_scale,
              double* h_var);

// ---- eigen.hip ----------------------------------------------------------------
// mode: EIG_FULL decomposes A; EIG_SETUP_ONLY only lays out the workspace of the two-stage path
// (used by the distributed stage 1); EIG_RESUME continues after an externally driven stage 1.
enum EigMode { EIG_FULL = 0, EIG_SETUP_ONLY = 1, EIG_RESUME = 2 };
// Internal status (never crosses the C ABI): the watchdog of a persistent kernel fired in a decomposition whose
// stage 1 was driven from outside (EIG_RESUME). The distributed fit agrees on it over the ranks and replays the
// decomposition with the launch-per-step kernels on every rank (csrc/fit.hip); larger than every public code so
// that the agreement (a MAX) prefers it to a rank-local OK.
constexpr int BK_EWATCHDOG = 90;
int eigen(bigkrls_ctx* ctx, const double* A, int64_t n, int64_t lda, int64_t n_vals, double* vals,
          int64_t n_vecs_max, double keep_thresh, double* vecs, int64_t ldv, int64_t* h_n_vecs,
          int part_index = 0, int part_count = 1, int mode = EIG_FULL);
// block Lanczos for the n_vals largest pairs with the products K B_j sharded over the ranks of `comm`: Kcols is this
// rank's column block K[:, r0:r1] (n x (r1 - r0), ld n), nb the partition's block size; everything else replicated
int eigen_krylov_dist(bigkrls_comm* comm, const double* Kcols, int64_t n, int64_t r0, int64_t r1, int64_t nb,
                      int64_t n_vals, double* vals, int64_t n_vecs_max, double keep_thresh, double* vecs, int64_t ldv,
                      int64_t* h_n_vecs);
// Row-block distributed stage 1 (dense -> band), one call per panel step between the caller's
// collectives; see include/bigkrls.h (bigkrls_dev_s1_*).
int dist_s1_open(bigkrls_ctx* ctx, int64_t n);
int dist_s1_panel(bigkrls_ctx* ctx, int64_t n, int64_t k, const double* strip);
int dist_s1_av(bigkrls_ctx* ctx, int64_t n, int64_t k, const double* Acols, int64_t lda, int64_t ncols,
               int64_t row0, double* Ypart, int64_t ldy);
int dist_s1_update(bigkrls_ctx* ctx, int64_t n, int64_t k, double* Y, double* Acols, int64_t lda, int64_t ncols,
                   int64_t row0);
// p[0 .. count) = uniform values in [-0.5, 0.5) from a counter-based generator (element index and seed only): the
// start block of the block Lanczos, identical on every device
int fill_random(bigkrls_ctx* ctx, double* p, int64_t count, uint32_t seed);
// Cholesky-QR twice of the n x b block W (b <= 128) on the device (Gram product, register-tile Cholesky + inverse,
// W R^-1), in place with `tmp` (n x b) as scratch; h_R (b x b, host, column-major) receives R = R2 R1,
// *h_breakdown is set when a pivot was not positive. Synchronises the stream. (csrc/eigen.hip)
int cholqr2_block(bigkrls_ctx* ctx, double* W, double* tmp, int64_t n, int b, double* h_R, int* h_breakdown,
                  double* d_R = nullptr);
// T (m x m, m = steps b, column-major, device) = the block-tridiagonal projected matrix of a block Lanczos run from
// its diagonal blocks A_j (symmetrised) and sub-diagonal factors beta_{j+1} (upper triangular), b x b each
int lanczos_projected(bigkrls_ctx* ctx, const double* d_A_blocks, const double* d_beta_blocks, int steps, int b,
                      double* d_T);
int dist_s1_panel_begin(bigkrls_ctx* ctx, int64_t n, int64_t k, const double* strip);
int dist_s1_thin(bigkrls_ctx* ctx, int64_t n, int64_t k, double* Y);
int dist_s1_update_cols(bigkrls_ctx* ctx, int64_t n, int64_t k, double* Acols, int64_t lda, int64_t ncols,
                        int64_t row0);
int dist_s1_put(bigkrls_ctx* ctx, int64_t n, int64_t k, const double* strip, int64_t ncols);
int dist_s1_trace(bigkrls_ctx* ctx, int64_t n, int64_t k);   // diagnostics: hashes of panel k's replicated factors
// several panels per trailing update (the group that starts at k0): see csrc/eigen.hip
int dist_s1_group_size(bigkrls_ctx* ctx, int64_t n, int64_t k0);
int dist_s1_local_agg_mode(bigkrls_ctx* ctx, int64_t n);            // 4 / 2 / 0: what this rank's settings allow
int dist_s1_set_agg_mode(bigkrls_ctx* ctx, int64_t n, int mode);    // ... and what the ranks agreed on (their minimum)
int dist_s1_thin_group(bigkrls_ctx* ctx, int64_t n, int64_t k, double* Y, int64_t k0);
int dist_s1_update_cols_group(bigkrls_ctx* ctx, int64_t n, int64_t k, double* Acols, int64_t lda, int64_t ncols,
                              int64_t row0, int64_t k0, int nblk);

}  // namespace bk
