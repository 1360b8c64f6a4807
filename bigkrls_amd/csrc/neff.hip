// Effective sample size from the mean absolute pairwise correlation of the rows of X
// (replaces the scalar triple loop of src/Neffective.cpp:13-65):
//     z_i = (x_i - mean(x_i)) / |x_i - mean(x_i)|      (rows de-meaned and normalised, :29-44)
//     r   = sum_{i > j} | z_i . z_j |                   (:52-55)
//     Neffective = N (1 - 2 r / N^2) + 1                (:61-64)
// The N^2 P/2 dot products are the lower wave tiles of a Gram matrix Z Z' on fp64 MFMA; nothing
// of size N x N is ever written: every wave folds |g| over the strictly-lower elements of its
// 32 x 32 tile into one partial, and the partials are summed in a fixed order (deterministic).
#include "common.h"

namespace bk {

typedef double d4 __attribute__((ext_vector_type(4)));

namespace {

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// one thread per row: Z[i, :] = (X[i, :] - mean) / sqrt(sum of squares); Z is n x p, ld n
__global__ void neff_rowstd_kernel(const double* __restrict__ X, int64_t ldx, int n, int p,
                                   double* __restrict__ Z) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double s = 0.0;
  for (int j = 0; j < p; ++j) s += X[i + (int64_t)j * ldx];
  const double mean = s / p;
  double ss = 0.0;
  for (int j = 0; j < p; ++j) {
    const double d = X[i + (int64_t)j * ldx] - mean;
    ss += d * d;
  }
  const double nrm = sqrt(ss);      // a constant row gives 0/0 = NaN, as in the reference
  for (int j = 0; j < p; ++j) Z[i + (int64_t)j * n] = (X[i + (int64_t)j * ldx] - mean) / nrm;
}

// wave per 32 x 32 lower tile of G = Z Z'; partial[w] = sum over the tile of |G[m][n]|, m > n
__global__ __launch_bounds__(256) void neff_abs_lower_kernel(const double* __restrict__ Z, int n, int p,
                                                             int tiles, int64_t ntiles,
                                                             double* __restrict__ partial) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t w = (int64_t)blockIdx.x * 4 + wave;
  if (w >= ntiles) return;
  // lower-triangular tiles column by column: column c starts at c*tiles - c(c-1)/2
  int tn = (int)((2.0 * tiles + 1.0 - sqrt((2.0 * tiles + 1.0) * (2.0 * tiles + 1.0) - 8.0 * (double)w)) * 0.5);
  while (tn > 0 && (int64_t)tn * tiles - (int64_t)tn * (tn - 1) / 2 > w) --tn;
  while ((int64_t)(tn + 1) * tiles - (int64_t)(tn + 1) * tn / 2 <= w) ++tn;
  const int tm = tn + (int)(w - ((int64_t)tn * tiles - (int64_t)tn * (tn - 1) / 2));
  const int m0 = tm * 32, n0 = tn * 32;
  const int lm = lane & 15, lk = lane >> 4;
  d4 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = (d4){0.0, 0.0, 0.0, 0.0};
  const double* a0p = Z + min(m0 + lm, n - 1);
  const double* a1p = Z + min(m0 + 16 + lm, n - 1);
  const double* b0p = Z + min(n0 + lm, n - 1);
  const double* b1p = Z + min(n0 + 16 + lm, n - 1);
  constexpr int KS = 4;
  for (int kc0 = 0; kc0 < p; kc0 += 4 * KS) {
    double fa0[KS], fa1[KS], fb0[KS], fb1[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const int64_t kc = min(kc0 + 4 * s + lk, p - 1);
      fa0[s] = a0p[kc * n];
      fa1[s] = a1p[kc * n];
      fb0[s] = b0p[kc * n];
      fb1[s] = b1p[kc * n];
    }
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const bool kv = (kc0 + 4 * s + lk) < p;
      const double a0 = kv ? fa0[s] : 0.0, a1 = kv ? fa1[s] : 0.0;
      const double b0 = kv ? fb0[s] : 0.0, b1 = kv ? fb1[s] : 0.0;
      acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(b0, a0, acc[0][0], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(b0, a1, acc[1][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(b1, a0, acc[0][1], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(b1, a1, acc[1][1], 0, 0, 0);
    }
  }
  // acc[i][j][r] = G[m0 + 16 i + lm][n0 + 16 j + lk + 4 r]
  double s = 0.0;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = m0 + 16 * i + lm, nn = n0 + 16 * j + lk + 4 * r;
        if (m < n && m > nn) s += fabs(acc[i][j][r]);
      }
  s = wave_sum(s);
  if (lane == 0) partial[w] = s;
}

// deterministic sum of the partials: one block, fixed strides, fixed tree
__global__ __launch_bounds__(1024) void neff_reduce_kernel(const double* __restrict__ partial, int64_t np,
                                                           double* __restrict__ out) {
  __shared__ double sh[16];
  double s = 0.0;
  for (int64_t e = threadIdx.x; e < np; e += 1024) s += partial[e];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
    for (int q = 0; q < 16; ++q) t += sh[q];
    out[0] = t;
  }
}

}  // namespace

int neffective(bigkrls_ctx* ctx, const double* X, int64_t n, int64_t ldx, int64_t p, double* h_out) {
  BK_REQUIRE(X && h_out && n > 0 && p > 0 && ldx >= n, "neffective: bad arguments");
  BK_REQUIRE(n < (1ll << 31) && p < (1ll << 31), "neffective: dimension too large");
  const int tiles = (int)((n + 31) / 32);
  const int64_t ntiles = (int64_t)tiles * (tiles + 1) / 2;
  void *pz = nullptr, *pp = nullptr;
  BK_TRY(ws_get(ctx, SLOT_DERIV_KB, n * p * sizeof(double), &pz));
  BK_TRY(ws_get(ctx, SLOT_DERIV_T, (ntiles + 1) * sizeof(double), &pp));
  double* Z = (double*)pz;
  double* partial = (double*)pp;
  hipLaunchKernelGGL(neff_rowstd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, X, ldx,
                     (int)n, (int)p, Z);
  BK_CHECK_LAUNCH();
  BK_REQUIRE((ntiles + 3) / 4 < (1ll << 31), "neffective: too many tiles");
  hipLaunchKernelGGL(neff_abs_lower_kernel, dim3((unsigned)((ntiles + 3) / 4)), dim3(256), 0, ctx->stream,
                     (const double*)Z, (int)n, (int)p, tiles, ntiles, partial);
  BK_CHECK_LAUNCH();
  hipLaunchKernelGGL(neff_reduce_kernel, dim3(1), dim3(1024), 0, ctx->stream, (const double*)partial, ntiles,
                     partial + ntiles);
  BK_CHECK_LAUNCH();
  double r = 0.0;
  PinnedFetch pf(ctx, 1);
  BK_TRY(pf.add(&r, partial + ntiles, sizeof(double)));
  BK_TRY(pf.finish());
  const double N = (double)n;
  const double mean_abs_cor = 2.0 * r / (N * N);   // src/Neffective.cpp:61
  *h_out = N * (1.0 - mean_abs_cor) + 1.0;         // :64
  return BIGKRLS_OK;
}

}  // namespace bk
