// solveforc + golden-section lambda search.
//
// Reference: src/solveforc.cpp:13-65 forms, row by row, the lower triangle of
// G^-1 = Q diag(1/(d+lambda)) Q' (N^2 K/2 flops per probe, two N x K transposes).
// Here one probe reads Q exactly once (8 N K bytes):
//     c_i = sum_k Q_ik (a_k / (d_k+lambda)),   g_i = sum_k Q_ik^2 / (d_k+lambda)
// with a = Q'y hoisted out of the search, then Le = sum_i (c_i/g_i)^2 by a
// wave-reduced pass. The search control flow is R/bigKRLS_Rcpp_functions.R:5-82
// statement for statement (bounds loops, 0.381966, branch on S1 < S2).
#include "common.h"

#include <cmath>
#include <cstring>
#include <limits>

namespace bk {

__device__ __forceinline__ double wave_sum_sf(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}

int qty(bigkrls_ctx* ctx, const double* Q, int64_t n, int64_t k, int64_t ldq, const double* y,
        double* a) {
  return gemv(ctx, 1, n, k, 1.0, Q, ldq, y, 0.0, a);
}

// w[k] = 1/(d_k+lambda), wa[k] = a_k/(d_k+lambda)
__global__ void sf_weights_kernel(int k, const double* __restrict__ d, const double* __restrict__ a,
                                  double lambda, double* __restrict__ w, double* __restrict__ wa) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < k) {
    const double wi = 1.0 / (d[i] + lambda);
    w[i] = wi;
    wa[i] = a[i] * wi;
  }
}

constexpr int SF_KC = 256;  // eigenpairs staged per LDS chunk

// grid (row blocks of 256, splits over k). Coalesced: consecutive lanes read
// consecutive rows of one eigenvector (column of Q).
__global__ __launch_bounds__(256) void sf_partial_kernel(int n_rows, int k, int k_per_split,
                                                         const double* __restrict__ Q, int64_t ldq,
                                                         const double* __restrict__ w,
                                                         const double* __restrict__ wa,
                                                         double* __restrict__ pc,
                                                         double* __restrict__ pg) {
  __shared__ double sw[SF_KC];
  __shared__ double swa[SF_KC];
  const int row = blockIdx.x * 256 + threadIdx.x;
  const int kb = blockIdx.y * k_per_split;
  const int ke = min(k, kb + k_per_split);
  double c0 = 0.0, c1 = 0.0, g0 = 0.0, g1 = 0.0;
  for (int base = kb; base < ke; base += SF_KC) {
    const int cnt = min(SF_KC, ke - base);
    __syncthreads();
    if (threadIdx.x < cnt) {
      sw[threadIdx.x] = w[base + threadIdx.x];
      swa[threadIdx.x] = wa[base + threadIdx.x];
    }
    __syncthreads();
    if (row < n_rows) {
      const double* q = Q + row + (int64_t)base * ldq;
      int j = 0;
      for (; j + 1 < cnt; j += 2) {
        const double q0 = q[(int64_t)j * ldq];
        const double q1 = q[(int64_t)(j + 1) * ldq];
        c0 += q0 * swa[j];
        g0 += q0 * q0 * sw[j];
        c1 += q1 * swa[j + 1];
        g1 += q1 * q1 * sw[j + 1];
      }
      if (j < cnt) {
        const double q0 = q[(int64_t)j * ldq];
        c0 += q0 * swa[j];
        g0 += q0 * q0 * sw[j];
      }
    }
  }
  if (row < n_rows) {
    pc[(int64_t)blockIdx.y * n_rows + row] = c0 + c1;
    pg[(int64_t)blockIdx.y * n_rows + row] = g0 + g1;
  }
}

__global__ __launch_bounds__(256) void sf_finish_kernel(int n_rows, int splits,
                                                        const double* __restrict__ pc,
                                                        const double* __restrict__ pg,
                                                        double* __restrict__ c,
                                                        double* __restrict__ le_part) {
  __shared__ double sh[4];
  const int row = blockIdx.x * 256 + threadIdx.x;
  double t = 0.0;
  if (row < n_rows) {
    double cs = 0.0, gs = 0.0;
    for (int z = 0; z < splits; ++z) {
      cs += pc[(int64_t)z * n_rows + row];
      gs += pg[(int64_t)z * n_rows + row];
    }
    if (c != nullptr) c[row] = cs;
    const double r = cs / gs;
    t = r * r;
  }
  t = wave_sum_sf(t);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = t;
  __syncthreads();
  if (threadIdx.x == 0) le_part[blockIdx.x] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

__global__ __launch_bounds__(256) void sf_sum_kernel(int n, const double* __restrict__ part,
                                                     double* __restrict__ out) {
  __shared__ double sh[4];
  double s = 0.0;
  for (int i = threadIdx.x; i < n; i += 256) s += part[i];
  s = wave_sum_sf(s);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) out[0] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

int solveforc(bigkrls_ctx* ctx, const double* Q, int64_t n_rows, int64_t k, int64_t ldq,
              const double* d, const double* a, double lambda, double* c, double* h_Le) {
  BK_REQUIRE(n_rows > 0 && k > 0 && n_rows < (1ll << 31) && k < (1ll << 31),
             "solveforc: bad dimensions");
  BK_REQUIRE(Q && d && a && h_Le, "solveforc: null pointer");
  const int rb = (int)((n_rows + 255) / 256);
  int splits = (1024 + rb - 1) / rb;
  const int max_splits = (int)((k + SF_KC - 1) / SF_KC);
  if (splits > max_splits) splits = max_splits;
  if (splits < 1) splits = 1;
  int kps = (int)(((k + splits - 1) / splits + SF_KC - 1) / SF_KC * SF_KC);
  splits = (int)((k + kps - 1) / kps);
  // workspace: w[k], wa[k], pc[splits*n], pg[splits*n], le_part[rb], le[1]
  const int64_t nd = 2 * k + 2 * (int64_t)splits * n_rows + rb + 8;
  void* p = nullptr;
  BK_TRY(ws_get(ctx, SLOT_SOLVE_PART, nd * sizeof(double), &p));
  double* w = (double*)p;
  double* wa = w + k;
  double* pc = wa + k;
  double* pg = pc + (int64_t)splits * n_rows;
  double* lep = pg + (int64_t)splits * n_rows;
  double* le = lep + rb;
  hipLaunchKernelGGL(sf_weights_kernel, dim3((unsigned)((k + 255) / 256)), dim3(256), 0,
                     ctx->stream, (int)k, d, a, lambda, w, wa);
  BK_CHECK_LAUNCH();
  hipLaunchKernelGGL(sf_partial_kernel, dim3(rb, splits), dim3(256), 0, ctx->stream, (int)n_rows,
                     (int)k, kps, Q, ldq, (const double*)w, (const double*)wa, pc, pg);
  BK_CHECK_LAUNCH();
  hipLaunchKernelGGL(sf_finish_kernel, dim3(rb), dim3(256), 0, ctx->stream, (int)n_rows, splits,
                     (const double*)pc, (const double*)pg, c, lep);
  BK_CHECK_LAUNCH();
  hipLaunchKernelGGL(sf_sum_kernel, dim3(1), dim3(256), 0, ctx->stream, rb, (const double*)lep, le);
  BK_CHECK_LAUNCH();
  double* hp = nullptr;
  BK_TRY(pinned_get(ctx, 1, &hp));
  BK_HIP(hipMemcpyAsync(hp, le, sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  BK_HIP(hipStreamSynchronize(ctx->stream));
  *h_Le = hp[0];
  return BIGKRLS_OK;
}

// ---- bounds (R/bigKRLS_Rcpp_functions.R:16-36) --------------------------------
static double ratio_sum(const double* d, int64_t n, double t) {
  // R's sum() accumulates in extended precision; long double matches it on x86-64
  long double s = 0.0L;
  for (int64_t i = 0; i < n; ++i) s += (long double)(d[i] / (d[i] + t));
  return (double)s;
}
// The same sum in plain double with four accumulators (vectorisable: ~5x faster than the dependent extended-precision
// chain): used only to FIND the neighbourhood of the step at which a loop stops; the step itself is then settled with
// the exact sum (ratio_sum) on either side of it.
static double ratio_sum_fast(const double* d, int64_t n, double t) {
  double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
  int64_t i = 0;
  for (; i + 4 <= n; i += 4) {
    s0 += d[i] / (d[i] + t);
    s1 += d[i + 1] / (d[i + 1] + t);
    s2 += d[i + 2] / (d[i + 2] + t);
    s3 += d[i + 3] / (d[i + 3] + t);
  }
  for (; i < n; ++i) s0 += d[i] / (d[i] + t);
  return (s0 + s1) + (s2 + s3);
}
// smallest s in [1, smax] with stop(s) true, given stop(0) false and stop monotone (false ... false true ... true):
// located with `guess` (a cheap approximation of stop), settled with `stop` itself
template <class Guess, class Stop>
static int64_t first_stop(int64_t smax, Guess guess, Stop stop) {
  int64_t lo = 0, hi = 1;                      // guess false at lo
  while (hi < smax && !guess(hi)) { lo = hi; hi *= 2; }
  if (hi > smax) hi = smax;
  while (hi - lo > 1) {
    const int64_t mid = lo + (hi - lo) / 2;
    if (guess(mid)) hi = mid; else lo = mid;
  }
  int64_t s = hi;                              // the exact predicate decides: walk to its first true step
  while (s > 1 && stop(s - 1)) --s;
  while (s < smax && !stop(s)) ++s;
  return s;
}

int lambda_bounds(const double* vals, int64_t n_vals, int64_t n, double* L, double* U) {
  BK_REQUIRE(vals && n_vals > 0 && L && U, "lambda_bounds: bad arguments");
  for (int64_t i = 0; i < n_vals; ++i)
    if (std::isnan(vals[i])) {
      set_error("Missing eigenvalues prevent bigKRLS from obtaining the regularization parameter lambda.");
      return BIGKRLS_EINVAL;
    }
  // U <- n; while (sum(d/(d+U)) < 1) U <- U - 1        (:17-21)
  // The sum is monotone decreasing in U, so the first U (counting down from n) with sum >= 1 is a search over the
  // integer step count; the loop's own arithmetic (U - 1 repeatedly, exact for integers) is reproduced.
  {
    const double u = (double)n;
    if (!(ratio_sum(vals, n_vals, u) < 1.0)) {
      *U = u;
    } else {
      // step s: U = n - s; at s = n (U = 0) the sum is the count of positive terms >= 1
      const int64_t s = first_stop(
          n, [&](int64_t k) { return !(ratio_sum_fast(vals, n_vals, (double)(n - k)) < 1.0); },
          [&](int64_t k) { return !(ratio_sum(vals, n_vals, (double)(n - k)) < 1.0); });
      *U = (double)(n - s);
    }
  }
  // L <- eps; q <- which.min(abs(d - max(d)/1000)); while (sum(d/(d+L)) > q) L <- L + 0.05  (:28-33)
  {
    double mx = vals[0];
    for (int64_t i = 1; i < n_vals; ++i) mx = vals[i] > mx ? vals[i] : mx;
    int64_t qi = 0;
    double best = std::fabs(vals[0] - mx / 1000.0);
    for (int64_t i = 1; i < n_vals; ++i) {
      const double v = std::fabs(vals[i] - mx / 1000.0);
      if (v < best) { best = v; qi = i; }
    }
    const double q = (double)(qi + 1);
    // the L sequence is generated by repeated addition exactly as the loop does
    std::vector<double> seq;
    seq.push_back(std::numeric_limits<double>::epsilon());
    auto L_at = [&](int64_t s) {
      while ((int64_t)seq.size() <= s) seq.push_back(seq.back() + 0.05);
      return seq[s];
    };
    if (!(ratio_sum(vals, n_vals, L_at(0)) > q)) {
      *L = L_at(0);
    } else {
      const int64_t cap = (int64_t)1 << 28;   // L > 1.3e7: the sum is far below any q >= 1 long before this
      const int64_t s = first_stop(
          cap, [&](int64_t k) { return !(ratio_sum_fast(vals, n_vals, L_at(k)) > q); },
          [&](int64_t k) { return !(ratio_sum(vals, n_vals, L_at(k)) > q); });
      *L = L_at(s);
    }
  }
  return BIGKRLS_OK;
}

// ---- golden-section search as a device-resident loop ------------------------------------------------------
// The search state lives in HBM; probe launch i first CONSUMES the loss of probe i-1 (every workgroup sums the
// row blocks' partial losses in the same fixed order and takes the same branch of R's loop, workgroup 0 records
// state and trace) and then evaluates the loss at the lambda that decision asks for. One launch per probe, no
// read-back inside the search: the host enqueues a chain of launches, the ones behind the terminating decision
// return at once, and state + trace come back in one copy. The arithmetic of the control step is R's, statement
// for statement and without fused multiply-adds (R/bigKRLS_Rcpp_functions.R:38-77).
struct SfState {
  double L, U, X1, X2, S1, S2, tol, cur, lambda, last_le;
  int pending;   // the slot (1 or 2) the probe at `cur` fills
  int done;
  int np;        // losses consumed so far (= probes completed)
  int fresh;     // no probe is in flight: the next launch starts one without consuming (start / resumed chain)
};

__device__ __forceinline__ SfState sf_step(SfState s, double le) {
  const double G = 0.381966;
  s.last_le = le;
  if (s.pending == 1) s.S1 = le; else s.S2 = le;
  s.np += 1;
  if (s.np == 1) {               // S1 <- loo(X1) done; S2 <- loo(X2) next (:40-41)
    s.cur = s.X2;
    s.pending = 2;
    return s;
  }
  if (fabs(__dsub_rn(s.S1, s.S2)) > s.tol) {                       // while (abs(S1 - S2) > tol)  (:55)
    if (s.S1 < s.S2) {                                            // :57-62
      s.U = s.X2;
      s.X2 = s.X1;
      s.X1 = __dadd_rn(s.L, __dmul_rn(G, __dsub_rn(s.U, s.L)));
      s.S2 = s.S1;
      s.cur = s.X1;
      s.pending = 1;
    } else {                                                      // :66-71
      s.L = s.X1;
      s.X1 = s.X2;
      s.X2 = __dsub_rn(s.U, __dmul_rn(G, __dsub_rn(s.U, s.L)));
      s.S1 = s.S2;
      s.cur = s.X2;
      s.pending = 2;
    }
  } else {
    s.done = 1;
    s.lambda = (s.S1 < s.S2) ? s.X1 : s.X2;                       // :77
  }
  return s;
}

// sum of `n` partial losses in a fixed order, identical in every workgroup (256 threads)
__device__ __forceinline__ double sf_block_total(const double* __restrict__ part, int n, double* sh) {
  double s = 0.0;
  for (int i = threadIdx.x; i < n; i += 256) s += part[i];
  s = wave_sum_sf(s);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
  __syncthreads();
  return (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

// grid (row blocks of ROWS rows, splits over k); the 256 threads of a workgroup are ROWS rows x 256 / ROWS groups
// over the eigenpairs (a small N leaves most CUs idle with one row per thread: N = 20 000 is 79 workgroups of 256
// rows, 625 of 32). consume_only: the launch that closes a chain (one workgroup).
template <int ROWS>
__global__ __launch_bounds__(256) void sf_probe_kernel(int n_rows, int k, int k_per_split,
                                                       const double* __restrict__ Q, int64_t ldq,
                                                       const double* __restrict__ d, const double* __restrict__ a,
                                                       const SfState* __restrict__ st_in, SfState* __restrict__ st_out,
                                                       const double* __restrict__ le_in, double* __restrict__ le_out,
                                                       int n_le, double* __restrict__ pc, double* __restrict__ pg,
                                                       double* __restrict__ trace, int max_trace, int consume_only) {
  constexpr int KG = 256 / ROWS;
  __shared__ double sw[SF_KC];
  __shared__ double swa[SF_KC];
  __shared__ double sh[4];
  __shared__ double red[2][KG][ROWS];
  SfState s = *st_in;
  if (!s.done) {
    if (!s.fresh) {
      const double le = sf_block_total(le_in, n_le, sh);
      if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0 && trace != nullptr && s.np < max_trace) {
        trace[2 * s.np] = s.cur;
        trace[2 * s.np + 1] = le;
      }
      s = sf_step(s, le);
    }
    s.fresh = consume_only;
  }
  if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) *st_out = s;
  if (s.done || consume_only) return;
  const double lambda = s.cur;
  const int r = threadIdx.x % ROWS, kg = threadIdx.x / ROWS;
  const int row = blockIdx.x * ROWS + r;
  const int kb = blockIdx.y * k_per_split;
  const int ke = min(k, kb + k_per_split);
  double c0 = 0.0, c1 = 0.0, g0 = 0.0, g1 = 0.0;
  for (int base = kb; base < ke; base += SF_KC) {
    const int cnt = min(SF_KC, ke - base);
    __syncthreads();
    if (threadIdx.x < cnt) {
      const double wi = 1.0 / (d[base + threadIdx.x] + lambda);
      sw[threadIdx.x] = wi;
      swa[threadIdx.x] = a[base + threadIdx.x] * wi;
    }
    __syncthreads();
    if (row < n_rows) {
      const double* q = Q + row + (int64_t)base * ldq;
      int j = kg;
      // four eigenvector entries in flight per thread and pass
      for (; j + 3 * KG < cnt; j += 4 * KG) {
        const double q0 = q[(int64_t)j * ldq], q1 = q[(int64_t)(j + KG) * ldq];
        const double q2 = q[(int64_t)(j + 2 * KG) * ldq], q3 = q[(int64_t)(j + 3 * KG) * ldq];
        c0 += q0 * swa[j];
        g0 += q0 * q0 * sw[j];
        c1 += q1 * swa[j + KG];
        g1 += q1 * q1 * sw[j + KG];
        c0 += q2 * swa[j + 2 * KG];
        g0 += q2 * q2 * sw[j + 2 * KG];
        c1 += q3 * swa[j + 3 * KG];
        g1 += q3 * q3 * sw[j + 3 * KG];
      }
      for (; j < cnt; j += KG) {
        const double q0 = q[(int64_t)j * ldq];
        c0 += q0 * swa[j];
        g0 += q0 * q0 * sw[j];
      }
    }
  }
  double cs = c0 + c1, gs = g0 + g1;
  if (KG > 1) {   // the groups' partial sums of a row, added in group order
    red[0][kg][r] = cs;
    red[1][kg][r] = gs;
    __syncthreads();
    if (kg == 0) {
      cs = red[0][0][r];
      gs = red[1][0][r];
#pragma unroll
      for (int q = 1; q < KG; ++q) {
        cs += red[0][q][r];
        gs += red[1][q][r];
      }
    }
  }
  if (gridDim.y > 1) {
    if (row < n_rows && kg == 0) {
      pc[(int64_t)blockIdx.y * n_rows + row] = cs;
      pg[(int64_t)blockIdx.y * n_rows + row] = gs;
    }
    return;
  }
  double t = 0.0;
  if (row < n_rows && kg == 0) {
    const double rr = cs / gs;
    t = rr * rr;
  }
  t = wave_sum_sf(t);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = t;
  __syncthreads();
  if (threadIdx.x == 0) le_out[blockIdx.x] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

// splits > 1: the row blocks' losses from the k-split partial sums (skipped once the search is done)
__global__ __launch_bounds__(256) void sf_probe_finish_kernel(int n_rows, int splits, const double* __restrict__ pc,
                                                              const double* __restrict__ pg,
                                                              const SfState* __restrict__ st,
                                                              double* __restrict__ le_part) {
  __shared__ double sh[4];
  if (st->done) return;
  const int row = blockIdx.x * 256 + threadIdx.x;
  double t = 0.0;
  if (row < n_rows) {
    double cs = 0.0, gs = 0.0;
    for (int z = 0; z < splits; ++z) {
      cs += pc[(int64_t)z * n_rows + row];
      gs += pg[(int64_t)z * n_rows + row];
    }
    const double r = cs / gs;
    t = r * r;
  }
  t = wave_sum_sf(t);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = t;
  __syncthreads();
  if (threadIdx.x == 0) le_part[blockIdx.x] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

int lambda_search(bigkrls_ctx* ctx, const double* Q, int64_t n, int64_t k, int64_t ldq,
                  const double* d, const double* a, const double* h_vals_all, int64_t n_vals,
                  double L, double U, double tol, double* h_lambda, int64_t* h_nprobes,
                  double* h_trace, int64_t max_trace, bigkrls_comm* comm, int64_t n_total) {
  BK_REQUIRE(h_lambda, "lambda_search: null output");
  if (n_total <= 0) n_total = n;
  BK_REQUIRE(n >= 0 && n_total > 0 && k > 0 && n_total < (1ll << 31) && k < (1ll << 31), "lambda_search: bad dimensions");
  BK_REQUIRE((Q || n == 0) && d && a, "lambda_search: null pointer");
  BK_REQUIRE(comm || n == n_total, "lambda_search: a row block needs a communicator");
  if (tol <= 0.0) tol = 1e-3 * (double)n_total;  // R/bigKRLS_Rcpp_functions.R:11-12
  if (L < 0.0 || U < 0.0) {
    double l0, u0;
    BK_TRY(lambda_bounds(h_vals_all, n_vals, n_total, &l0, &u0));
    if (L < 0.0) L = l0;
    if (U < 0.0) U = u0;
  }
  hipStream_t st = ctx->stream;
  const int rows = (n <= 32 * 16384) ? 32 : 256;          // rows per workgroup of the probe kernel
  const int rb = (int)std::max<int64_t>((n + rows - 1) / rows, 1);
  const int rbf = (int)std::max<int64_t>((n + 255) / 256, 1);   // row blocks of the finish kernel (k split over workgroups)
  int splits = (1024 + rb - 1) / rb;
  const int max_splits = (int)((k + SF_KC - 1) / SF_KC);
  if (splits > max_splits) splits = max_splits;
  if (splits < 1) splits = 1;
  const int kps = (int)(((k + splits - 1) / splits + SF_KC - 1) / SF_KC * SF_KC);
  splits = (int)((k + kps - 1) / kps);
  const int nle = splits > 1 ? rbf : rb;   // partial losses a probe leaves behind
  constexpr int CHAIN = 48;        // probe launches per chain (a search takes 15-40)
  constexpr int DEV_TRACE = 512;   // probes recorded on the device
  // workspace: state[2], le[2][rb], trace[2 DEV_TRACE], pc / pg [splits n]
  const int64_t st_d = (2 * (int64_t)sizeof(SfState) + 7) / 8;
  const int64_t nd = st_d + 2 * rb + 2 * DEV_TRACE + (splits > 1 ? 2 * (int64_t)splits * n : 0) + 16;
  void* p = nullptr;
  {
    const int rc = ws_get(ctx, SLOT_SOLVE_PART, nd * sizeof(double), &p);
    BK_TRY(comm ? comm_agree(comm, rc) : rc);     // (before the first all-reduce of the chain)
  }
  SfState* dstate = (SfState*)p;
  double* le = (double*)p + st_d;
  double* dtrace = le + 2 * rb;
  double* pc = dtrace + 2 * DEV_TRACE;
  double* pg = pc + (splits > 1 ? (int64_t)splits * n : 0);
  double* lsum = pg + (splits > 1 ? (int64_t)splits * n : 0);   // [2]: the all-reduced loss of a probe (row-block search)
  // initial state (:38-39): X1 <- L + .381966 (U - L); X2 <- U - .381966 (U - L)
  double* hp = nullptr;
  const int64_t hp_d = st_d + 2 * DEV_TRACE;
  BK_TRY(pinned_get(ctx, hp_d, &hp));
  SfState s0{};
  {
    const volatile double G = 0.381966;   // (volatile: no contraction of the two statements into FMAs)
    const volatile double span = U - L;
    const volatile double gs = G * span;
    s0.L = L; s0.U = U;
    s0.X1 = L + gs;
    s0.X2 = U - gs;
  }
  s0.tol = tol;
  s0.cur = s0.X1;
  s0.pending = 1;
  s0.fresh = 1;
  std::memcpy(hp, &s0, sizeof(SfState));
  BK_HIP(hipMemcpyAsync(&dstate[0], hp, sizeof(SfState), hipMemcpyHostToDevice, st));
  SfState fin{};
  int launched = 0;
  for (int chain = 0; chain < 256; ++chain) {
    for (int i = 0; i <= CHAIN; ++i, ++launched) {
      const int cur = launched & 1;
      const bool close = (i == CHAIN);
      const bool samp = ctx->profile && chain == 0 && i == 1;
      if (samp) BK_TRY(prof_begin(ctx, "solveforc_probe", 8.0 * (double)n * (double)k));
      // (a row-block search consumes the loss summed over the ranks: one double)
      const double* le_prev = comm ? lsum + (cur ^ 1) : le + (cur ^ 1) * rb;
      auto kern = rows == 32 ? sf_probe_kernel<32> : sf_probe_kernel<256>;
      hipLaunchKernelGGL(kern, close ? dim3(1, 1) : dim3(rb, splits), dim3(256), 0, st, (int)n, (int)k, kps, Q, ldq,
                         d, a, (const SfState*)&dstate[cur], &dstate[cur ^ 1], le_prev, le + cur * rb, comm ? 1 : nle,
                         pc, pg, dtrace, DEV_TRACE, close ? 1 : 0);
      if (samp) BK_TRY(prof_end(ctx, "solveforc_probe"));
      BK_CHECK_LAUNCH();
      if (!close && splits > 1) {
        hipLaunchKernelGGL(sf_probe_finish_kernel, dim3(rbf), dim3(256), 0, st, (int)n, splits, (const double*)pc,
                           (const double*)pg, (const SfState*)&dstate[cur ^ 1], le + cur * rb);
        BK_CHECK_LAUNCH();
      }
      if (comm && !close) {
        // the probe's loss over all rows: this rank's sum, then one all-reduce of a scalar,
        // stream-ordered like the launches around it
        hipLaunchKernelGGL(sf_sum_kernel, dim3(1), dim3(256), 0, st, nle, (const double*)(le + cur * rb), lsum + cur);
        BK_CHECK_LAUNCH();
        BK_TRY(comm_all_reduce(comm, lsum + cur, 1, COMM_SUM));
      }
    }
    // the closing launch consumed the chain's last loss without starting a probe: a search that is not done
    // resumes from exactly that state with its next launch
    BK_HIP(hipMemcpyAsync(hp, &dstate[launched & 1], sizeof(SfState), hipMemcpyDeviceToHost, st));
    if (h_trace && max_trace > 0)   // (8 KB: cheaper than a second synchronisation once the search is known to be done)
      BK_HIP(hipMemcpyAsync(hp + st_d, dtrace, (size_t)(2 * DEV_TRACE) * sizeof(double), hipMemcpyDeviceToHost, st));
    BK_HIP(hipStreamSynchronize(st));
    std::memcpy(&fin, hp, sizeof(SfState));
    if (fin.done) break;
    if (fin.np > 10000) break;   // (a NaN loss ends R's loop at once: abs(NaN) > tol is FALSE)
  }
  if (!fin.done) {
    set_error("lambda_search: golden section did not terminate (NaN loss?)");
    return BIGKRLS_ENOCONV;
  }
  *h_lambda = fin.lambda;
  if (h_nprobes) *h_nprobes = fin.np;
  if (h_trace && max_trace > 0) {
    const int64_t cnt = std::min<int64_t>(std::min<int64_t>(fin.np, max_trace), DEV_TRACE);
    std::memcpy(h_trace, hp + st_d, (size_t)(2 * cnt) * sizeof(double));
  }
  return BIGKRLS_OK;
}

}  // namespace bk
