// Marginal effects (pointwise derivatives) and the variance of their averages.
//
// Reference: src/bigderiv_v3.cpp:13-111. Per column it builds N x N temporaries
// (L = differences % K, or adj_T/adj_C) and multiplies N x N x N twice
// (L' V L at :105, (exp(adj*phi) % K) * V' at :82-84): 4 N^3 flops per column.
//
// Here the same numbers come from ONE pass over K for all columns:
//   KB = K [1, c, {x_j, x_j o c | b_j, b_j o c}_j]          (N x N x (2+2P) fp64 MFMA)
//   continuous: D_rj = (-2/sigma)(x_rj (Kc)_r - (K(x_j o c))_r),  s_rj = x_rj (K1)_r - (K x_j)_r
//   binary    : group sums S1,O1,Sc,Oc from K b_j, K(b_j o c); see finalize kernel
//   var_j     = scale_j * s_j' V s_j = scale_j * sum_k wv_k (q_k' s_j)^2        (V never formed)
// K is symmetric, so the row block [row0,row0+n_rows) of K is handed in as the
// contiguous column block K[:, row0:row0+n_rows] (n x n_rows, ld ldk).
#include "common.h"

#include <cmath>

namespace bk {

__device__ __forceinline__ double wave_min(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmin(v, __shfl_down(v, off, 64));
  return v;
}
__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_down(v, off, 64));
  return v;
}

// minmax[2j] = min(X[:,j]), minmax[2j+1] = max(X[:,j]); one block per column
__global__ __launch_bounds__(256) void col_minmax_kernel(int n, const double* __restrict__ X,
                                                         int64_t ldx, double* __restrict__ minmax) {
  __shared__ double smin[4], smax[4];
  const double* x = X + (int64_t)blockIdx.x * ldx;
  double lo = INFINITY, hi = -INFINITY;
  for (int i = threadIdx.x; i < n; i += 256) {
    const double v = x[i];
    lo = fmin(lo, v);
    hi = fmax(hi, v);
  }
  lo = wave_min(lo);
  hi = wave_max(hi);
  if ((threadIdx.x & 63) == 0) {
    smin[threadIdx.x >> 6] = lo;
    smax[threadIdx.x >> 6] = hi;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    minmax[2 * blockIdx.x] = fmin(fmin(smin[0], smin[1]), fmin(smin[2], smin[3]));
    minmax[2 * blockIdx.x + 1] = fmax(fmax(smax[0], smax[1]), fmax(smax[2], smax[3]));
  }
}

// B (n x (2+2p)): col 0 = 1, col 1 = c, col 2+2j = x_j or b_j, col 3+2j = that o c
__global__ void build_b_kernel(int n, int p, const double* __restrict__ X, int64_t ldx,
                               const double* __restrict__ c, const int* __restrict__ is_binary,
                               const double* __restrict__ minmax, double* __restrict__ B) {
  const int64_t total = (int64_t)n * (p + 1);
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (int64_t)gridDim.x * blockDim.x) {
    const int i = (int)(e % n);
    const int j = (int)(e / n) - 1;
    const double ci = c[i];
    if (j < 0) {
      B[i] = 1.0;
      B[(int64_t)n + i] = ci;
    } else {
      double v = X[i + (int64_t)j * ldx];
      if (is_binary[j]) v = (v == minmax[2 * j + 1]) ? 1.0 : 0.0;
      B[(int64_t)(2 + 2 * j) * n + i] = v;
      B[(int64_t)(3 + 2 * j) * n + i] = v * ci;
    }
  }
}

__global__ void deriv_finalize_kernel(int n_rows, int p, int row0, const double* __restrict__ X,
                                      int64_t ldx, const int* __restrict__ is_binary,
                                      const double* __restrict__ minmax,
                                      const double* __restrict__ KB, double sigma,
                                      double* __restrict__ D, int64_t ldd, double* __restrict__ S,
                                      int64_t lds) {
  const int64_t total = (int64_t)n_rows * p;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (int64_t)gridDim.x * blockDim.x) {
    const int r = (int)(e % n_rows);
    const int j = (int)(e / n_rows);
    const double K1 = KB[r];
    const double Kc = KB[(int64_t)n_rows + r];
    const double Kv = KB[(int64_t)(2 + 2 * j) * n_rows + r];
    const double Kvc = KB[(int64_t)(3 + 2 * j) * n_rows + r];
    const double x = X[(row0 + r) + (int64_t)j * ldx];
    double dv, sv;
    if (!is_binary[j]) {
      dv = (-2.0 / sigma) * (x * Kc - Kvc);   // src/bigderiv_v3.cpp:103
      sv = x * K1 - Kv;                       // row sums of L (:102,:105)
    } else {
      const double z0 = minmax[2 * j], z1 = minmax[2 * j + 1];
      const double sd = 1.0 / (z1 - z0);                    // :36
      const double phi = -1.0 / (sd * sd * sigma);          // :37
      const double E = exp(phi), Einv = exp(-phi);
      const bool hi = (x == z1);
      const double S1 = hi ? Kv : K1 - Kv;
      const double O1 = hi ? K1 - Kv : Kv;
      const double Sc = hi ? Kvc : Kc - Kvc;
      const double Oc = hi ? Kc - Kvc : Kvc;
      dv = sd * (hi ? 1.0 : -1.0) * ((1.0 - E) * Sc + (1.0 - Einv) * Oc);   // :69-71
      const double kt = hi ? S1 + Einv * O1 : E * S1 + O1;                  // :66
      const double kc = hi ? E * S1 + O1 : S1 + Einv * O1;                  // :67
      sv = kt - kc;                                                         // :82-84 collapsed
    }
    D[r + (int64_t)j * ldd] = dv;
    S[r + (int64_t)j * lds] = sv;
  }
}

int deriv_rows(bigkrls_ctx* ctx, const double* Krows, int64_t n, int64_t n_rows, int64_t ldk,
               int64_t row0, const double* X, int64_t p, int64_t ldx, const int32_t* h_is_binary,
               const double* c, double sigma, double* D, int64_t ldd, double* S, int64_t lds, double* kc_out,
               const double* extra, int64_t n_extra, double* extra_out) {
  // `extra` (n x n_extra, ld n): more operand columns for the same pass over K; K extra -> extra_out (n_rows x n_extra).
  // The fit sends the two +-1 combinations of its kept eigenvectors along: the check of the decomposition against K
  // costs no pass over K of its own (csrc/fit.hip).
  BK_REQUIRE(n > 0 && n_rows > 0 && p > 0 && n < (1ll << 31) && p < (1 << 20),
             "deriv_rows: bad dimensions");
  BK_REQUIRE(row0 >= 0 && row0 + n_rows <= n, "deriv_rows: row block out of range");
  BK_REQUIRE(Krows && X && h_is_binary && c && D && S, "deriv_rows: null pointer");
  BK_REQUIRE(n_extra >= 0 && (n_extra == 0 || (extra && extra_out)), "deriv_rows: bad extra operand");
  const int64_t nb0 = 2 + 2 * p, nb = nb0 + n_extra;
  void *pb = nullptr, *pkb = nullptr, *pt = nullptr;
  BK_TRY(ws_get(ctx, SLOT_DERIV_B, n * nb * sizeof(double), &pb));
  BK_TRY(ws_get(ctx, SLOT_DERIV_KB, n_rows * nb * sizeof(double), &pkb));
  BK_TRY(ws_get(ctx, SLOT_DERIV_T, (2 * p + 8) * sizeof(double) + p * sizeof(int32_t), &pt));
  double* minmax = (double*)pt;
  int* d_isbin = (int*)(minmax + 2 * p + 8);
  {
    // (through the context's pinned upload arena: no asynchronous copy out of the caller's pageable memory)
    PinnedStage up(ctx);
    BK_TRY(up.reserve((size_t)p * sizeof(int32_t) + 256));
    BK_TRY(up.put(d_isbin, h_is_binary, (size_t)p * sizeof(int32_t)));
  }
  hipLaunchKernelGGL(col_minmax_kernel, dim3((unsigned)p), dim3(256), 0, ctx->stream, (int)n, X,
                     ldx, minmax);
  BK_CHECK_LAUNCH();
  int blocks = (int)std::min<int64_t>((n * (p + 1) + 255) / 256, 4096);
  hipLaunchKernelGGL(build_b_kernel, dim3(blocks), dim3(256), 0, ctx->stream, (int)n, (int)p, X,
                     ldx, c, (const int*)d_isbin, (const double*)minmax, (double*)pb);
  BK_CHECK_LAUNCH();
  if (n_extra > 0)
    BK_HIP(hipMemcpyAsync((double*)pb + nb0 * n, extra, (size_t)(n * n_extra) * sizeof(double), hipMemcpyDeviceToDevice,
                          ctx->stream));
  // KB (n_rows x nb) = Krows' (n_rows x n) * B (n x nb). The whole (exactly symmetric) K: K B, the product that
  // streams K along its contiguous dimension (the transposed-operand GEMM runs at about a third of its rate)
  // (33 .. 48 operand columns -- P = 16 .. 23 -- on the 128 x 48 tile: the 128 x 64 tile of gemm() would run the MFMA
  //  units on up to a third of padding; BIGKRLS_DERIV48=0: the generic kernel, for cross-checks)
  static const bool use48 = [] { const char* e = getenv("BIGKRLS_DERIV48"); return !(e && e[0] == '0'); }();
  if (n_rows == n && row0 == 0 && nb > 32 && nb <= 48 && n >= 4096 && use48)
    BK_TRY(gemm_nn_skinny48(ctx, n, nb, n, Krows, ldk, (const double*)pb, n, (double*)pkb, n));
  else if (n_rows == n && row0 == 0)
    BK_TRY(gemm(ctx, 0, 0, n, nb, n, 1.0, Krows, ldk, (const double*)pb, n, 0.0, (double*)pkb, n));
  else
    BK_TRY(gemm(ctx, 1, 0, n_rows, nb, n, 1.0, Krows, ldk, (const double*)pb, n, 0.0, (double*)pkb,
                n_rows));
  blocks = (int)std::min<int64_t>((n_rows * p + 255) / 256, 4096);
  hipLaunchKernelGGL(deriv_finalize_kernel, dim3(blocks), dim3(256), 0, ctx->stream, (int)n_rows,
                     (int)p, (int)row0, X, ldx, (const int*)d_isbin, (const double*)minmax,
                     (const double*)pkb, sigma, D, ldd, S, lds);
  BK_CHECK_LAUNCH();
  // K c is column 1 of the product: the caller's fitted values (R/bigKRLS.R:291) without a pass over K of their own
  if (kc_out != nullptr)
    BK_HIP(hipMemcpyAsync(kc_out, (const double*)pkb + n_rows, (size_t)n_rows * sizeof(double), hipMemcpyDeviceToDevice,
                          ctx->stream));
  if (n_extra > 0)
    BK_HIP(hipMemcpyAsync(extra_out, (const double*)pkb + nb0 * n_rows, (size_t)(n_rows * n_extra) * sizeof(double),
                          hipMemcpyDeviceToDevice, ctx->stream));
  // (the upload arena and the product buffers are free again when this returns)
  BK_HIP(hipStreamSynchronize(ctx->stream));
  return BIGKRLS_OK;
}

// out[j] = sum_k wv_k T[k,j]^2 ; one block per column
__global__ __launch_bounds__(256) void wcolsumsq_kernel(int k, const double* __restrict__ T,
                                                        int64_t ldt, const double* __restrict__ wv,
                                                        double* __restrict__ out) {
  __shared__ double sh[4];
  const double* t = T + (int64_t)blockIdx.x * ldt;
  double s = 0.0;
  for (int i = threadIdx.x; i < k; i += 256) s += wv[i] * t[i] * t[i];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) out[blockIdx.x] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

int deriv_var(bigkrls_ctx* ctx, const double* Q, int64_t n, int64_t k, int64_t ldq,
              const double* wv, const double* S, int64_t p, int64_t lds, const double* h_scale,
              double* h_var) {
  BK_REQUIRE(n > 0 && k > 0 && p > 0, "deriv_var: bad dimensions");
  BK_REQUIRE(Q && wv && S && h_scale && h_var, "deriv_var: null pointer");
  void* pt = nullptr;
  BK_TRY(ws_get(ctx, SLOT_DERIV_KB, (k * p + p) * sizeof(double), &pt));
  double* T = (double*)pt;
  double* out = T + k * p;
  BK_TRY(gemm(ctx, 1, 0, k, p, n, 1.0, Q, ldq, S, lds, 0.0, T, k));   // T = Q'S
  hipLaunchKernelGGL(wcolsumsq_kernel, dim3((unsigned)p), dim3(256), 0, ctx->stream, (int)k,
                     (const double*)T, k, wv, out);
  BK_CHECK_LAUNCH();
  double* hp = nullptr;
  BK_TRY(pinned_get(ctx, p, &hp));
  BK_HIP(hipMemcpyAsync(hp, out, p * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  BK_HIP(hipStreamSynchronize(ctx->stream));
  for (int64_t j = 0; j < p; ++j) h_var[j] = h_scale[j] * hp[j];
  return BIGKRLS_OK;
}

}  // namespace bk
