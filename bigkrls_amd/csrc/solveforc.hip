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

int lambda_bounds(const double* vals, int64_t n_vals, int64_t n, double* L, double* U) {
  BK_REQUIRE(vals && n_vals > 0 && L && U, "lambda_bounds: bad arguments");
  for (int64_t i = 0; i < n_vals; ++i)
    if (std::isnan(vals[i])) {
      set_error("Missing eigenvalues prevent bigKRLS from obtaining the regularization parameter lambda.");
      return BIGKRLS_EINVAL;
    }
  // U <- n; while (sum(d/(d+U)) < 1) U <- U - 1        (:17-21)
  // The sum is monotone decreasing in U, so the first U (counting down from n) with
  // sum >= 1 is found by bisection over the integer step count; the loop's own
  // arithmetic (U - 1 repeatedly, exact for integers) is reproduced.
  {
    double u = (double)n;
    if (!(ratio_sum(vals, n_vals, u) < 1.0)) {
      *U = u;
    } else {
      // find smallest s >= 1 with sum(d/(d+(n-s))) >= 1; guard at U -> 0
      int64_t lo = 0, hi = 1;  // predicate false at lo (sum < 1), search hi where true
      while (hi < n && ratio_sum(vals, n_vals, (double)(n - hi)) < 1.0) {
        lo = hi;
        hi *= 2;
      }
      if (hi > n) hi = n;  // U = 0: sum = n_vals positive terms = count >= 1
      while (hi - lo > 1) {
        const int64_t mid = lo + (hi - lo) / 2;
        if (ratio_sum(vals, n_vals, (double)(n - mid)) < 1.0) lo = mid; else hi = mid;
      }
      *U = (double)(n - hi);
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
      int64_t lo = 0, hi = 1;
      const int64_t cap = (int64_t)1 << 40;
      while (hi < cap && ratio_sum(vals, n_vals, L_at(hi)) > q) {
        lo = hi;
        hi *= 2;
        if (hi > (1 << 28)) break;  // L > 1.3e7: sum is far below any q >= 1 long before this
      }
      while (hi - lo > 1) {
        const int64_t mid = lo + (hi - lo) / 2;
        if (ratio_sum(vals, n_vals, L_at(mid)) > q) lo = mid; else hi = mid;
      }
      *L = L_at(hi);
    }
  }
  return BIGKRLS_OK;
}

int lambda_search(bigkrls_ctx* ctx, const double* Q, int64_t n, int64_t k, int64_t ldq,
                  const double* d, const double* a, const double* h_vals_all, int64_t n_vals,
                  double L, double U, double tol, double* h_lambda, int64_t* h_nprobes,
                  double* h_trace, int64_t max_trace) {
  BK_REQUIRE(h_lambda, "lambda_search: null output");
  if (tol <= 0.0) tol = 1e-3 * (double)n;  // R/bigKRLS_Rcpp_functions.R:11-12
  if (L < 0.0 || U < 0.0) {
    double l0, u0;
    BK_TRY(lambda_bounds(h_vals_all, n_vals, n, &l0, &u0));
    if (L < 0.0) L = l0;
    if (U < 0.0) U = u0;
  }
  int64_t np = 0;
  int status = BIGKRLS_OK;
  auto loo = [&](double lam) -> double {
    double le = 0.0;
    int s = solveforc(ctx, Q, n, k, ldq, d, a, lam, nullptr, &le);
    if (s != BIGKRLS_OK) status = s;
    if (h_trace && np < max_trace) {
      h_trace[2 * np] = lam;
      h_trace[2 * np + 1] = le;
    }
    ++np;
    return le;
  };
  const double G = 0.381966;
  double X1 = L + G * (U - L);
  double X2 = U - G * (U - L);
  double S1 = loo(X1);
  double S2 = loo(X2);
  while (status == BIGKRLS_OK && std::fabs(S1 - S2) > tol) {
    if (S1 < S2) {
      U = X2;
      X2 = X1;
      X1 = L + G * (U - L);
      S2 = S1;
      S1 = loo(X1);
    } else {
      L = X1;
      X1 = X2;
      X2 = U - G * (U - L);
      S1 = S2;
      S2 = loo(X2);
    }
    if (np > 10000) {
      set_error("lambda_search: golden section did not terminate (NaN loss?)");
      return BIGKRLS_ENOCONV;
    }
  }
  if (status != BIGKRLS_OK) return status;
  *h_lambda = (S1 < S2) ? X1 : X2;
  if (h_nprobes) *h_nprobes = np;
  return BIGKRLS_OK;
}

}  // namespace bk
