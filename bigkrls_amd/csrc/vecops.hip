// Bandwidth-bound vector / matrix-vector kernels (wave64 reductions, coalesced
// column-major streaming). Replaces src/multdiag.cpp:13-24 and the bigalgebra
// dgemv/daxpy calls of R/bigKRLS.R:291,294,601.
#include "common.h"

namespace bk {

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}

// block-wide sum (blockDim.x multiple of 64, <= 1024); result valid in thread 0
__device__ __forceinline__ double block_sum(double v, double* sh) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (lane == 0) sh[w] = v;
  __syncthreads();
  double r = 0.0;
  if (w == 0) {
    const int nw = (blockDim.x + 63) >> 6;
    r = (lane < nw) ? sh[lane] : 0.0;
    r = wave_sum(r);
  }
  __syncthreads();
  return r;
}

// ---- y = alpha * A' x + beta * y : one wave per column -----------------------
__global__ __launch_bounds__(256) void gemv_t_kernel(int m, int n, double alpha,
                                                     const double* __restrict__ A, int64_t lda,
                                                     const double* __restrict__ x, double beta,
                                                     double* __restrict__ y) {
  const int lane = threadIdx.x & 63;
  const int col = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (col >= n) return;
  const double* a = A + (int64_t)col * lda;
  double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
  int i = lane;
  for (; i + 192 < m; i += 256) {
    s0 += a[i] * x[i];
    s1 += a[i + 64] * x[i + 64];
    s2 += a[i + 128] * x[i + 128];
    s3 += a[i + 192] * x[i + 192];
  }
  for (; i < m; i += 64) s0 += a[i] * x[i];
  double s = wave_sum((s0 + s1) + (s2 + s3));
  if (lane == 0) y[col] = (beta == 0.0) ? alpha * s : alpha * s + beta * y[col];
}

// ---- y = alpha * A x + beta * y : thread per row, column chunks --------------
__global__ __launch_bounds__(256) void gemv_n_partial_kernel(int m, int n, int cols_per_block,
                                                             const double* __restrict__ A,
                                                             int64_t lda,
                                                             const double* __restrict__ x,
                                                             double* __restrict__ partial) {
  const int row = blockIdx.x * 256 + threadIdx.x;
  const int c0 = blockIdx.y * cols_per_block;
  const int c1 = min(n, c0 + cols_per_block);
  if (row >= m) return;
  const double* a = A + row;
  double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
  int c = c0;
  for (; c + 3 < c1; c += 4) {
    s0 += a[(int64_t)c * lda] * x[c];
    s1 += a[(int64_t)(c + 1) * lda] * x[c + 1];
    s2 += a[(int64_t)(c + 2) * lda] * x[c + 2];
    s3 += a[(int64_t)(c + 3) * lda] * x[c + 3];
  }
  for (; c < c1; ++c) s0 += a[(int64_t)c * lda] * x[c];
  partial[(int64_t)blockIdx.y * m + row] = (s0 + s1) + (s2 + s3);
}

__global__ void gemv_n_reduce_kernel(int m, int splits, double alpha,
                                     const double* __restrict__ partial, double beta,
                                     double* __restrict__ y) {
  const int row = blockIdx.x * blockDim.x + threadIdx.x;
  if (row >= m) return;
  double s = 0.0;
  for (int z = 0; z < splits; ++z) s += partial[(int64_t)z * m + row];
  y[row] = (beta == 0.0) ? alpha * s : alpha * s + beta * y[row];
}

int gemv(bigkrls_ctx* ctx, int trans, int64_t m, int64_t n, double alpha, const double* A,
         int64_t lda, const double* x, double beta, double* y) {
  BK_REQUIRE(m >= 0 && n >= 0 && m < (1ll << 31) && n < (1ll << 31), "gemv: bad dimensions");
  if (trans) {
    if (n == 0) return BIGKRLS_OK;
    hipLaunchKernelGGL(gemv_t_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, ctx->stream,
                       (int)m, (int)n, alpha, A, lda, x, beta, y);
    BK_CHECK_LAUNCH();
    return BIGKRLS_OK;
  }
  if (m == 0) return BIGKRLS_OK;
  const int rb = (int)((m + 255) / 256);
  int splits = (2048 + rb - 1) / rb;
  if (splits > (n + 63) / 64) splits = (int)((n + 63) / 64);
  if (splits < 1) splits = 1;
  const int cpb = (int)((n + splits - 1) / splits);
  splits = (int)((n + cpb - 1) / std::max(cpb, 1));
  if (splits < 1) splits = 1;
  void* p = nullptr;
  BK_TRY(ws_get(ctx, SLOT_GEMM_SPLITK, (int64_t)splits * m * sizeof(double), &p));
  hipLaunchKernelGGL(gemv_n_partial_kernel, dim3(rb, splits), dim3(256), 0, ctx->stream, (int)m,
                     (int)n, std::max(cpb, 1), A, lda, x, (double*)p);
  BK_CHECK_LAUNCH();
  hipLaunchKernelGGL(gemv_n_reduce_kernel, dim3(rb), dim3(256), 0, ctx->stream, (int)m, splits,
                     alpha, (const double*)p, beta, y);
  BK_CHECK_LAUNCH();
  return BIGKRLS_OK;
}

// ---- dot product to host ------------------------------------------------------
__global__ __launch_bounds__(256) void dot_partial_kernel(int64_t n, const double* __restrict__ x,
                                                          const double* __restrict__ y,
                                                          double* __restrict__ partial) {
  __shared__ double sh[4];
  double s = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
    s += x[i] * y[i];
  s = block_sum(s, sh);
  if (threadIdx.x == 0) partial[blockIdx.x] = s;
}

__global__ __launch_bounds__(256) void sum_final_kernel(int n, const double* __restrict__ partial,
                                                        double* __restrict__ out) {
  __shared__ double sh[4];
  double s = 0.0;
  for (int i = threadIdx.x; i < n; i += 256) s += partial[i];
  s = block_sum(s, sh);
  if (threadIdx.x == 0) out[0] = s;
}

int dot_host(bigkrls_ctx* ctx, int64_t n, const double* x, const double* y, double* h_out) {
  BK_REQUIRE(n >= 0 && h_out, "dot: bad arguments");
  void* p = nullptr;
  BK_TRY(ws_get(ctx, SLOT_SCALAR, 2048 * sizeof(double), &p));
  double* part = (double*)p;
  int blocks = (int)std::min<int64_t>(std::max<int64_t>((n + 255) / 256, 1), 1024);
  hipLaunchKernelGGL(dot_partial_kernel, dim3(blocks), dim3(256), 0, ctx->stream, n, x, y, part);
  BK_CHECK_LAUNCH();
  hipLaunchKernelGGL(sum_final_kernel, dim3(1), dim3(256), 0, ctx->stream, blocks,
                     (const double*)part, part + 1024);
  BK_CHECK_LAUNCH();
  double* hp = nullptr;
  BK_TRY(pinned_get(ctx, 1, &hp));
  BK_HIP(hipMemcpyAsync(hp, part + 1024, sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  BK_HIP(hipStreamSynchronize(ctx->stream));
  *h_out = hp[0];
  return BIGKRLS_OK;
}

// ---- out[:,i] = A[:,i] * diag[i]  (src/multdiag.cpp:17-18) --------------------
__global__ void multdiag_kernel(int n, int k, const double* __restrict__ A, int64_t lda,
                                const double* __restrict__ diag, double* __restrict__ out,
                                int64_t ldo) {
  const int64_t total = (int64_t)n * k;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = e % n, c = e / n;
    out[r + c * ldo] = A[r + c * lda] * diag[c];
  }
}

int multdiag(bigkrls_ctx* ctx, const double* A, int64_t n, int64_t k, int64_t lda,
             const double* diag, double* out, int64_t ldo) {
  BK_REQUIRE(n >= 0 && k >= 0 && n < (1ll << 31) && k < (1ll << 31), "multdiag: bad dimensions");
  if (n == 0 || k == 0) return BIGKRLS_OK;
  int blocks = (int)std::min<int64_t>((n * k + 255) / 256, 4096);
  hipLaunchKernelGGL(multdiag_kernel, dim3(blocks), dim3(256), 0, ctx->stream, (int)n, (int)k, A,
                     lda, diag, out, ldo);
  BK_CHECK_LAUNCH();
  return BIGKRLS_OK;
}

__global__ void diag_kernel(int n, const double* __restrict__ A, int64_t lda,
                            double* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = A[(int64_t)i * lda + i];
}

int diag_extract(bigkrls_ctx* ctx, const double* A, int64_t n, int64_t lda, double* out) {
  if (n <= 0) return BIGKRLS_OK;
  hipLaunchKernelGGL(diag_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream,
                     (int)n, A, lda, out);
  BK_CHECK_LAUNCH();
  return BIGKRLS_OK;
}

__global__ void scale_kernel(int64_t n, double alpha, double* __restrict__ x) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x)
    x[i] *= alpha;
}

int scale(bigkrls_ctx* ctx, int64_t n, double alpha, double* x) {
  if (n <= 0) return BIGKRLS_OK;
  int blocks = (int)std::min<int64_t>((n + 255) / 256, 4096);
  hipLaunchKernelGGL(scale_kernel, dim3(blocks), dim3(256), 0, ctx->stream, n, alpha, x);
  BK_CHECK_LAUNCH();
  return BIGKRLS_OK;
}

// ---- squared row norms of a column-major n x p matrix -------------------------
__global__ void row_sqnorms_kernel(int n, int p, const double* __restrict__ A, int64_t lda,
                                   double* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double s = 0.0;
  for (int c = 0; c < p; ++c) {
    const double v = A[i + (int64_t)c * lda];
    s += v * v;
  }
  out[i] = s;
}

int row_sqnorms(bigkrls_ctx* ctx, const double* A, int64_t n, int64_t p, int64_t lda,
                double* out) {
  if (n <= 0) return BIGKRLS_OK;
  hipLaunchKernelGGL(row_sqnorms_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                     ctx->stream, (int)n, (int)p, A, lda, out);
  BK_CHECK_LAUNCH();
  return BIGKRLS_OK;
}

__global__ void copy_matrix_kernel(int m, int n, const double* __restrict__ A, int64_t lda,
                                   double* __restrict__ B, int64_t ldb) {
  const int64_t total = (int64_t)m * n;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = e % m, c = e / m;
    B[r + c * ldb] = A[r + c * lda];
  }
}

int copy_matrix(bigkrls_ctx* ctx, const double* A, int64_t m, int64_t n, int64_t lda, double* B,
                int64_t ldb) {
  if (m <= 0 || n <= 0) return BIGKRLS_OK;
  if (lda == m && ldb == m) {
    BK_HIP(hipMemcpyAsync(B, A, (size_t)m * n * sizeof(double), hipMemcpyDeviceToDevice,
                          ctx->stream));
    return BIGKRLS_OK;
  }
  int blocks = (int)std::min<int64_t>((m * n + 255) / 256, 8192);
  hipLaunchKernelGGL(copy_matrix_kernel, dim3(blocks), dim3(256), 0, ctx->stream, (int)m, (int)n,
                     A, lda, B, ldb);
  BK_CHECK_LAUNCH();
  return BIGKRLS_OK;
}

}  // namespace bk
