// Symmetric eigendecomposition on one MI355X, fp64, written from scratch.
//
// Replaces src/eigen.cpp:13-30 (arma::eig_sym -> LAPACK dsyevd; arma::eigs_sym ->
// ARPACK) and the truncation logic of bEigen (R/bigKRLS_Rcpp_functions.R:173-199).
//
//   phase 1  Householder tridiagonalisation, blocked (panel of 64 reflectors):
//            per column a coalesced wave-reduced symv over the trailing matrix
//            (HBM-bound half), per panel one rank-128 fp64 MFMA update
//            A22 -= [V W][W V]' (MFMA-bound half).
//   phase 2  Cuppen / Gu-Eisenstat divide & conquer on the tridiagonal matrix,
//            torn down to 1 x 1 leaves. Host does the O(n) deflation scan per
//            merge; the device does rank-one rotations, the secular equation
//            (one wave per root, bracketed geometric bisection on the shifted
//            variable, so every pole-root difference has full relative accuracy),
//            the Loewner re-derivation of z, and the eigenvector update as
//            batched fp64 MFMA GEMMs with gathered columns. Only the eigenvector
//            columns that will be kept (lastkeeper) are formed at the last merge.
//   phase 3  back-transform Z <- (I - V T V') Z per reflector panel (compact WY),
//            three fp64 MFMA GEMMs per panel.
//
// The eigenvalues are returned descending, the eigenvectors in matching columns
// (src/eigen.cpp:28-29). Signs are arbitrary (the reference flips them and never
// returns them to the user: R/bigKRLS_Rcpp_functions.R:186, quirk Q9).
#include "common.h"

#include <algorithm>
#include <cmath>
#include <numeric>
#include <cstdlib>
#include <string>
#include <chrono>
#include <cstdio>
#include <cstring>

namespace bk {
namespace {

constexpr int TRD_NB = 64;
constexpr double DEPS = 2.220446049250313e-16;

__device__ __forceinline__ double wsum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;  // every lane holds the identical total
}
__device__ __forceinline__ double wprod(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v *= __shfl_xor(v, off, 64);
  return v;
}
__device__ __forceinline__ double bsum256(double v, double* sh) {
  v = wsum(v);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  const double r = (sh[0] + sh[1]) + (sh[2] + sh[3]);
  __syncthreads();
  return r;
}

// =============================================================================
// phase 1: tridiagonalisation
// P1 = [V | Wm], P2 = [Wm | V], each n x 2*pw (ld n); column k of V at P1[:,k],
// column k of Wm at P1[:,pw+k].
// =============================================================================

// K1: finalise w_{i-1}, bring column c up to date, d[c], partial ||x[1:]||^2
__global__ __launch_bounds__(256) void trd_k1(double* __restrict__ W, int n, int c, int i, int pw,
                                              double* __restrict__ P1, double* __restrict__ P2,
                                              const double* __restrict__ part2, int np2,
                                              const double* __restrict__ tau,
                                              double* __restrict__ d, double* __restrict__ part1) {
  __shared__ double sV[TRD_NB], sW[TRD_NB], sh[4];
  __shared__ double s_alpha2;
  const int tid = threadIdx.x;
  const int r = c + blockIdx.x * 256 + tid;
  const int64_t N = n;
  double a = 0.0;
  if (i > 0) {
    if (tid == 0) {
      double s = 0.0;
      for (int q = 0; q < np2; ++q) s += part2[q];
      s_alpha2 = -0.5 * tau[c - 1] * s;
    }
    __syncthreads();
    const double alpha2 = s_alpha2;
    double v_last = 0.0, w_last = 0.0;
    if (r < n) {
      v_last = P1[r + (int64_t)(i - 1) * N];
      w_last = P1[r + (int64_t)(pw + i - 1) * N] + alpha2 * v_last;
      if (r > c) {  // row c of this column is never read again: leave it (no race with readers)
        P1[r + (int64_t)(pw + i - 1) * N] = w_last;
        P2[r + (int64_t)(i - 1) * N] = w_last;
      }
    }
    if (tid < i) {
      sV[tid] = P1[c + (int64_t)tid * N];
      sW[tid] = (tid == i - 1)
                    ? P1[c + (int64_t)(pw + i - 1) * N] + alpha2 * P1[c + (int64_t)(i - 1) * N]
                    : P1[c + (int64_t)(pw + tid) * N];
    }
    __syncthreads();
    if (r < n) {
      a = W[r + (int64_t)c * N];
      for (int k = 0; k < i - 1; ++k)
        a -= P1[r + (int64_t)k * N] * sW[k] + P1[r + (int64_t)(pw + k) * N] * sV[k];
      a -= v_last * sW[i - 1] + w_last * sV[i - 1];
      W[r + (int64_t)c * N] = a;
    }
  } else if (r < n) {
    a = W[r + (int64_t)c * N];
  }
  if (r == c) d[c] = a;
  double ss = (r < n && r >= c + 2) ? a * a : 0.0;
  ss = bsum256(ss, sh);
  if (tid == 0) part1[blockIdx.x] = ss;
}

// finalise the last w column of a panel (rows >= row_begin)
__global__ __launch_bounds__(256) void trd_fin(int n, int c, int i, int pw,
                                               double* __restrict__ P1, double* __restrict__ P2,
                                               const double* __restrict__ part2, int np2,
                                               const double* __restrict__ tau) {
  __shared__ double s_alpha2;
  if (threadIdx.x == 0) {
    double s = 0.0;
    for (int q = 0; q < np2; ++q) s += part2[q];
    s_alpha2 = -0.5 * tau[c] * s;
  }
  __syncthreads();
  const int64_t N = n;
  const int r = c + 1 + blockIdx.x * 256 + threadIdx.x;
  if (r < n) {
    const double wf = P1[r + (int64_t)(pw + i) * N] + s_alpha2 * P1[r + (int64_t)i * N];
    P1[r + (int64_t)(pw + i) * N] = wf;
    P2[r + (int64_t)i * N] = wf;
  }
}

// -----------------------------------------------------------------------------
// Symmetric symv, tiled: y = A22 v reading ONLY the lower triangle of A22 (the
// HBM-bound half of the tridiagonalisation: 8 L(L+1)/2 bytes per column instead of
// 8 L^2). The trailing matrix is cut into strips of 32 columns and row segments of
// RS rows; block (strip s, segment g) streams its tile once with 16-byte loads
// (lane = 2 consecutive rows, a wave covers 128 rows per instruction, 16 column
// loads in flight per lane) and produces
//   row partials  Prow[s][R]  = sum_{col in strip s, col <= R} A[R,col] v[col]
//   col partials  Pcol[g][col] = sum_{R in segment g, R > col} A[R,col] v[R]
// (32 per-lane accumulators, folded across the wave by a halving butterfly).
// A second small kernel sums the partials in a fixed order (deterministic, no
// atomics) 32 strips at a time; trd_k3 finishes the sum.
// -----------------------------------------------------------------------------
constexpr int SV_CW = 32;
#ifndef BK_SV_B
#define BK_SV_B 16
#endif
#ifndef BK_SV_OCC
#define BK_SV_OCC 2
#endif
#ifndef BK_SV_NCH
#define BK_SV_NCH 1
#endif
constexpr int SV_B = BK_SV_B;    // column loads in flight per lane and batch

// |(alpha, x)| from alpha and ss = |x|^2 (Householder: beta = -sign(alpha) |(alpha, x)|). ss is a plain sum of squares
// already, so the scaled hypot only pays where alpha^2 + ss leaves the range in which the plain form is accurate; on
// the latency chains of pq_resident / bc_resident (one reflector per column step / per hop) the square root of the
// sum is ~40 dependent instructions shorter.
__device__ __forceinline__ double hh_norm(double alpha, double ss) {
  const double s2 = fma(alpha, alpha, ss);
  if (s2 > 1e-280 && s2 < 1e280) return sqrt(s2);
  return hypot(alpha, sqrt(ss));
}

template <int VEC> struct RowVec;
template <> struct RowVec<1> { double v[1]; };
template <> struct alignas(16) RowVec<2> { double v[2]; };

__device__ __forceinline__ void house_scalars(const double* __restrict__ part1, int np1, double alpha,
                                              double& beta, double& t, double& sc) {
  double ss = 0.0;
  for (int q = 0; q < np1; ++q) ss += part1[q];
  if (ss == 0.0) {
    beta = alpha; t = 0.0; sc = 0.0;
  } else {
    beta = -copysign(hh_norm(alpha, ss), alpha);
    t = (beta - alpha) / beta;
    sc = 1.0 / (alpha - beta);
  }
}

template <int VEC>
__global__ __launch_bounds__(256, BK_SV_OCC) void trd_symv_tiles(
    const double* __restrict__ W, int n, int c, int i, int pw, const double* __restrict__ P1,
    const double* __restrict__ part1, int np1, int RS, int nstrips, double* __restrict__ Prow,
    double* __restrict__ Pcol, double* __restrict__ tvec, double* __restrict__ e,
    double* __restrict__ tau) {
  __shared__ double s_scale;
  __shared__ double s_vc[SV_CW];
  __shared__ double s_col[4][SV_CW];
  __shared__ double s_diag[SV_CW];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t N = n;
  const int t0 = c + 1;
  const double* x = W + (int64_t)c * N;  // x[R], R >= t0
  if (tid == 0) {
    double beta, t, sc;
    house_scalars(part1, np1, x[t0], beta, t, sc);
    s_scale = sc;
    if (blockIdx.x == 0 && blockIdx.y == 0) { e[c] = beta; tau[c] = t; }
  }
  __syncthreads();
  const double scale = s_scale;
  auto vrow = [&](int R) -> double {
    return (R < t0 || R >= n) ? 0.0 : ((R == t0) ? 1.0 : x[R] * scale);
  };
  if ((int)blockIdx.x >= nstrips) {
    // panel dots t1 = Wm'v (q < i), t2 = V'v (i <= q < 2i), row-segmented like the tiles:
    // block (column group, segment g) -> partial sums Ppan[g][q]; trd_k3t adds the segments.
    const int q0 = (((int)blockIdx.x - nstrips) * 4 + wave) * 4;
    const int total = 2 * i;
    const int r0 = (t0 - (t0 % VEC)) + (int)blockIdx.y * RS;
    if (r0 >= n || q0 >= total) return;
    const int r1 = min(n, r0 + RS);
    const double* ptr[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int q = min(q0 + u, total - 1);
      ptr[u] = (q < i) ? P1 + (int64_t)(pw + q) * N : P1 + (int64_t)(q - i) * N;
    }
    double acc[4] = {0.0, 0.0, 0.0, 0.0};
    for (int R = r0 + lane * VEC; R < r1; R += 64 * VEC) {
      const int Rl = min(R, n - VEC);
      double v[VEC];
#pragma unroll
      for (int w2 = 0; w2 < VEC; ++w2) v[w2] = vrow(R + w2);
      RowVec<VEC> a[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) a[u] = *reinterpret_cast<const RowVec<VEC>*>(ptr[u] + Rl);
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int w2 = 0; w2 < VEC; ++w2) acc[u] += a[u].v[w2] * v[w2];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const double sm = wsum(acc[u]);
      if (lane == 0 && q0 + u < total) tvec[(int64_t)blockIdx.y * (2 * TRD_NB) + q0 + u] = sm;
    }
    return;
  }
  const int s = blockIdx.x, g = blockIdx.y;
  const int j0 = t0 + SV_CW * s;
  const int ncols = min(SV_CW, n - j0);
  const int j0e = j0 - (j0 % VEC);
  const int seg0 = j0e + g * RS;
  if (seg0 >= n) return;
  const int seg1 = min(n, seg0 + RS);
  if (tid < SV_CW) s_vc[tid] = (tid < ncols) ? vrow(j0 + tid) : 0.0;
  __syncthreads();
  // the 32 v[col] values are wave-uniform: keep them in SGPRs for the whole block
  double vcs[SV_CW];
#pragma unroll
  for (int k = 0; k < SV_CW; ++k) {
    const double t = s_vc[k];
    const int lo = __builtin_amdgcn_readfirstlane(__double2loint(t));
    const int hi = __builtin_amdgcn_readfirstlane(__double2hiint(t));
    vcs[k] = __hiloint2double(hi, lo);
  }
  constexpr int CH = 64 * VEC;
  double col[SV_CW];
#pragma unroll
  for (int k = 0; k < SV_CW; ++k) col[k] = 0.0;
  if (tid < SV_CW) s_diag[tid] = 0.0;
  const double* Abase = W + (int64_t)j0 * N;
  constexpr int NCH = BK_SV_NCH;  // consecutive 128-row chunks a wave reads per column visit
  for (int chunk = seg0 + wave * NCH * CH; chunk < seg1; chunk += 4 * NCH * CH) {
    int R[NCH], Rl[NCH];
    double v[NCH][VEC], racc[NCH][VEC];
#pragma unroll
    for (int h = 0; h < NCH; ++h) {
      R[h] = chunk + h * CH + lane * VEC;
      Rl[h] = min(R[h], n - VEC);
#pragma unroll
      for (int u = 0; u < VEC; ++u) { v[h][u] = vrow(R[h] + u); racc[h][u] = 0.0; }
    }
    if (chunk < j0 + SV_CW) {
      // the rows that hold the diagonal block (wave 0, segment 0 only): triangular masks,
      // column sums reduced per column straight away (rare path, kept small)
#pragma unroll
      for (int h = 0; h < NCH; ++h) {
#pragma unroll 1
        for (int cc = 0; cc < ncols; ++cc) {
          RowVec<VEC> a;
#pragma unroll
          for (int u = 0; u < VEC; ++u) a.v[u] = 0.0;
          if (R[h] < n) a = *reinterpret_cast<const RowVec<VEC>*>(Abase + (int64_t)cc * N + R[h]);
          const double vc = s_vc[cc];
          const int ca = j0 + cc;
          double t = 0.0;
#pragma unroll
          for (int u = 0; u < VEC; ++u) {
            const int Ru = R[h] + u;
            racc[h][u] += ((Ru >= ca) ? a.v[u] : 0.0) * vc;
            t += ((Ru > ca) ? a.v[u] : 0.0) * v[h][u];
          }
          t = wsum(t);
          if (lane == 0) s_diag[cc] += t;
        }
      }
    } else {
#pragma unroll
      for (int b = 0; b < SV_CW; b += SV_B) {
        // unconditional loads: out-of-range rows/columns are clamped to valid addresses and
        // neutralised through v[u] == 0 / s_vc[cc] == 0 (their outputs are never stored)
        RowVec<VEC> a[SV_B][NCH];
#pragma unroll
        for (int k = 0; k < SV_B; ++k) {
          const int cc = min(b + k, ncols - 1);
#pragma unroll
          for (int h = 0; h < NCH; ++h)
            a[k][h] = *reinterpret_cast<const RowVec<VEC>*>((Abase + (int64_t)cc * N) + (unsigned)Rl[h]);
        }
#pragma unroll
        for (int k = 0; k < SV_B; ++k) {
          const int cc = b + k;
          const double vc = vcs[cc];
#pragma unroll
          for (int h = 0; h < NCH; ++h)
#pragma unroll
            for (int u = 0; u < VEC; ++u) {
              racc[h][u] += a[k][h].v[u] * vc;
              col[cc] += a[k][h].v[u] * v[h][u];
            }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
#pragma unroll
    for (int h = 0; h < NCH; ++h)
#pragma unroll
      for (int u = 0; u < VEC; ++u)
        if (R[h] + u < n) Prow[(int64_t)s * N + R[h] + u] = racc[h][u];
  }
  // fold the 32 per-lane column accumulators across the wave: after the five halving
  // steps lane l holds the total of column l >> 1 (lanes 2k and 2k+1 both).
#pragma unroll
  for (int half = 16, off = 32; half >= 1; half >>= 1, off >>= 1) {
    const bool hi = (lane & off) != 0;
#pragma unroll
    for (int k = 0; k < half; ++k) {
      const double send = hi ? col[k] : col[k + half];
      const double keep = hi ? col[k + half] : col[k];
      col[k] = keep + __shfl_xor(send, off, 64);
    }
  }
  const double tot = col[0] + __shfl_xor(col[0], 1, 64);
  if ((lane & 1) == 0) s_col[wave][lane >> 1] = tot;
  __syncthreads();
  if (tid < ncols)
    Pcol[(int64_t)g * N + j0 + tid] =
        ((s_col[0][tid] + s_col[1][tid]) + (s_col[2][tid] + s_col[3][tid])) + s_diag[tid];
}

// first level of the partial sum (32 strips per group) + v stored into the panel
__global__ __launch_bounds__(256) void trd_symv_reduce(
    const double* __restrict__ W, int n, int c, int i, int pw, int RS, int vec,
    const double* __restrict__ part1, int np1, const double* __restrict__ Prow,
    const double* __restrict__ Pcol, double* __restrict__ Prow2, double* __restrict__ P1,
    double* __restrict__ P2) {
  __shared__ double s_scale;
  const int64_t N = n;
  const int t0 = c + 1;
  const double* x = W + (int64_t)c * N;
  if (threadIdx.x == 0) {
    double beta, t, sc;
    house_scalars(part1, np1, x[t0], beta, t, sc);
    s_scale = sc;
  }
  __syncthreads();
  const int R = t0 + blockIdx.x * 256 + threadIdx.x;
  if (R >= n) return;
  const int q = blockIdx.y;
  const int strip = (R - t0) / SV_CW;
  const int s_lo = 32 * q, s_hi = min(32 * q + 31, strip);
  double sum = 0.0;
  for (int s = s_lo; s <= s_hi; ++s) sum += Prow[(int64_t)s * N + R];
  if (q == 0) {
    const int j0 = t0 + SV_CW * strip;
    const int j0e = j0 - (j0 % vec);
    const int nseg = (n - j0e + RS - 1) / RS;
    for (int g = 0; g < nseg; ++g) sum += Pcol[(int64_t)g * N + R];
    const double v = (R == t0) ? 1.0 : x[R] * s_scale;
    P1[R + (int64_t)i * N] = v;
    P2[R + (int64_t)(pw + i) * N] = v;
  }
  Prow2[(int64_t)q * N + R] = sum;
}

// K3 for the tiled symv: y_R = sum_q Prow2[q][R]
__global__ __launch_bounds__(256) void trd_k3t(double* __restrict__ W, int n, int c, int i, int pw,
                                               double* __restrict__ P1, double* __restrict__ P2,
                                               const double* __restrict__ Prow2,
                                               const double* __restrict__ tvec, int npseg,
                                               const double* __restrict__ tau,
                                               double* __restrict__ part2) {
  __shared__ double st1[TRD_NB], st2[TRD_NB], sh[4];
  const int tid = threadIdx.x;
  const int64_t N = n;
  const int L = n - c - 1;
  if (tid < i) {
    double a1 = 0.0, a2 = 0.0;
    for (int g = 0; g < npseg; ++g) {
      a1 += tvec[g * (2 * TRD_NB) + tid];
      a2 += tvec[g * (2 * TRD_NB) + i + tid];
    }
    st1[tid] = a1;
    st2[tid] = a2;
  }
  __syncthreads();
  const int r = blockIdx.x * 256 + tid;
  double pv = 0.0;
  if (r < L) {
    const int64_t row = c + 1 + r;
    const double v = P1[row + (int64_t)i * N];
    const int nq = r / (SV_CW * 32) + 1;
    double s = 0.0;
    for (int q = 0; q < nq; ++q) s += Prow2[(int64_t)q * N + row];
    for (int k = 0; k < i; ++k)
      s -= P1[row + (int64_t)k * N] * st1[k] + P1[row + (int64_t)(pw + k) * N] * st2[k];
    const double wt = tau[c] * s;
    P1[row + (int64_t)(pw + i) * N] = wt;
    P2[row + (int64_t)i * N] = wt;
    W[row + (int64_t)c * N] = v;
    pv = wt * v;
  }
  pv = bsum256(pv, sh);
  if (tid == 0) part2[blockIdx.x] = pv;
}

struct SymvWs {
  double* Prow;
  double* Pcol;
  double* Prow2;
  double* Ppan;   // [segment][2*TRD_NB] partial panel dots
};

int tridiagonalize(bigkrls_ctx* ctx, double* W, int n, double* d, double* e, double* tau,
                   double* P1, double* P2, double* scratch /* y[n] + tvec[2nb] + part1 + part2 */,
                   const SymvWs& sw) {
  const int maxb = (n + 255) / 256 + 1;
  double* y = scratch;
  double* tvec = y + n;
  double* part1 = tvec + 2 * TRD_NB;
  double* part2 = part1 + maxb;
  hipStream_t st = ctx->stream;
  const int64_t N = n;
  const int vec = (n % 2 == 0) ? 2 : 1;
  const int CH = 64 * vec;
  for (int j0 = 0; j0 < n - 1; j0 += TRD_NB) {
    const int pw = std::min(TRD_NB, n - 1 - j0);
    int np2 = 0;
    for (int i = 0; i < pw; ++i) {
      const int c = j0 + i;
      const int nb1 = (n - c + 255) / 256;
      hipLaunchKernelGGL(trd_k1, dim3(nb1), dim3(256), 0, st, W, n, c, i, pw, P1, P2,
                         (const double*)part2, np2, (const double*)tau, d, part1);
      const int L = n - c - 1;
      const int nb3 = (L + 255) / 256;
      const bool sample = ctx->profile && (c % 8 == 0);
      {
        const int nstrips = (L + SV_CW - 1) / SV_CW;
        const int rsq = 4 * CH * BK_SV_NCH;  // segments are whole numbers of block iterations
        int RS = ((L / 6 + rsq - 1) / rsq) * rsq;
        RS = std::max(rsq, std::min(RS, 8192));
        const int nsegmax = (L + 1 + RS - 1) / RS;
        const int t0e = (c + 1) - ((c + 1) % vec);
        const int npan = (2 * i + 15) / 16;
        const int nq = (nstrips + 31) / 32;
        if (sample) BK_TRY(prof_begin(ctx, "symv", 4.0 * (double)L * (double)(L + 1)));
        if (vec == 2)
          hipLaunchKernelGGL(trd_symv_tiles<2>, dim3(nstrips + npan, nsegmax), dim3(256), 0, st,
                             (const double*)W, n, c, i, pw, (const double*)P1, (const double*)part1,
                             nb1, RS, nstrips, sw.Prow, sw.Pcol, sw.Ppan, e, tau);
        else
          hipLaunchKernelGGL(trd_symv_tiles<1>, dim3(nstrips + npan, nsegmax), dim3(256), 0, st,
                             (const double*)W, n, c, i, pw, (const double*)P1, (const double*)part1,
                             nb1, RS, nstrips, sw.Prow, sw.Pcol, sw.Ppan, e, tau);
        hipLaunchKernelGGL(trd_symv_reduce, dim3(nb3, nq), dim3(256), 0, st, (const double*)W, n, c, i,
                           pw, RS, vec, (const double*)part1, nb1, (const double*)sw.Prow,
                           (const double*)sw.Pcol, sw.Prow2, P1, P2);
        if (sample) BK_TRY(prof_end(ctx, "symv"));
        const int npseg = (n - (t0e) + RS - 1) / RS;
        hipLaunchKernelGGL(trd_k3t, dim3(nb3), dim3(256), 0, st, W, n, c, i, pw, P1, P2,
                           (const double*)sw.Prow2, (const double*)sw.Ppan, npseg, (const double*)tau,
                           part2);
      }
      np2 = nb3;
    }
    BK_CHECK_LAUNCH();
    const int cl = j0 + pw - 1;
    const int j1 = j0 + pw;
    const int nbf = (n - cl - 1 + 255) / 256;
    hipLaunchKernelGGL(trd_fin, dim3(nbf), dim3(256), 0, st, n, cl, pw - 1, pw, P1, P2,
                       (const double*)part2, np2, (const double*)tau);
    BK_CHECK_LAUNCH();
    const int64_t mt = n - j1;
    BK_TRY(prof_begin(ctx, "trailing_update",
                      (double)mt * (double)(mt + 1) * 2.0 * pw));
    BK_TRY(syrk_lower(ctx, mt, 2 * pw, -1.0, P1 + j1, N, P2 + j1, N, W + j1 + (int64_t)j1 * N, N));
    BK_TRY(prof_end(ctx, "trailing_update"));
  }
  BK_HIP(hipMemcpyAsync(d + (n - 1), W + (int64_t)(n - 1) * N + (n - 1), sizeof(double),
                        hipMemcpyDeviceToDevice, st));
  return BIGKRLS_OK;
}

// =============================================================================
// phase 2: divide & conquer
// =============================================================================
struct MergeDesc {
  int s, n1, m, K;
  int k1, k2, k3, Kneed;
  int rot_off, nrot, ndef, rev;   // rev: columns stored in descending order of the roots (the root of the tree)
  double rho;
  // where the merge's secular vectors go: column c at Ubase + uoff + c * uld -- the block (s, s) of the N x N buffer U
  // (uoff = s + s N, uld = N), or, on a level whose operators stay factored, a contiguous K x K block of the stash
  long long uoff;
  int uld, pad_;
};

// Leaves of the divide & conquer tree (17..32 rows when n > 64): implicit QL with eigenvectors
// (EISPACK tql2 / "tqli"), one wave per leaf. Every lane runs the same scalar recurrence on d, e in
// LDS (identical values, so the redundant stores are harmless) and lane r applies the rotations to
// row r of the eigenvector block. Replaces the five lowest merge levels (of ~1 ms of launches, copies
// and synchronisations each) by one launch.
constexpr int DC_LEAF = 32;
__global__ __launch_bounds__(64) void dc_leaf_ql(const int2* __restrict__ leaves, const double* __restrict__ dd,
                                                 const double* __restrict__ ee, double* __restrict__ Q0,
                                                 double* __restrict__ Q1, int64_t ld,
                                                 double* __restrict__ lam_out, int* __restrict__ fail) {
  __shared__ double sd[DC_LEAF], se[DC_LEAF + 1], sZ[DC_LEAF][DC_LEAF + 1];
  const int2 lf = leaves[blockIdx.x];
  const int s = lf.x, m = lf.y, lane = threadIdx.x;
  if (lane < m) {
    sd[lane] = dd[s + lane];
    se[lane] = (lane < m - 1) ? ee[s + lane] : 0.0;
    for (int c = 0; c < m; ++c) sZ[lane][c] = (c == lane) ? 1.0 : 0.0;
  }
  __syncthreads();
  for (int l = 0; l < m; ++l) {
    int iter = 0, mm;
    do {
      for (mm = l; mm < m - 1; ++mm) {
        const double dsum = fabs(sd[mm]) + fabs(sd[mm + 1]);
        if (fabs(se[mm]) <= DEPS * dsum) break;
      }
      if (mm != l) {
        if (++iter > 80) {          // no convergence: reported, never silently accepted
          if (lane == 0) *fail = 1;
          break;
        }
        double g = (sd[l + 1] - sd[l]) / (2.0 * se[l]);
        double r = hypot(g, 1.0);
        g = sd[mm] - sd[l] + se[l] / (g + copysign(r, g));
        double sn = 1.0, cs = 1.0, p = 0.0;
        int i;
        for (i = mm - 1; i >= l; --i) {
          const double f = sn * se[i], b = cs * se[i];
          r = hypot(f, g);
          se[i + 1] = r;
          if (r == 0.0) {
            sd[i + 1] -= p;
            se[mm] = 0.0;
            break;
          }
          sn = f / r;
          cs = g / r;
          g = sd[i + 1] - p;
          r = (sd[i] - g) * sn + 2.0 * cs * b;
          p = sn * r;
          sd[i + 1] = g + p;
          g = cs * r - b;
          if (lane < m) {
            const double zf = sZ[lane][i + 1], zi = sZ[lane][i];
            sZ[lane][i + 1] = sn * zi + cs * zf;
            sZ[lane][i] = cs * zi - sn * zf;
          }
        }
        if (r == 0.0 && i >= l) continue;
        sd[l] -= p;
        se[l] = g;
        se[mm] = 0.0;
      }
    } while (mm != l);
  }
  __syncthreads();
  if (lane < m) {
    lam_out[s + lane] = sd[lane];
    for (int c = 0; c < m; ++c) {
      const double v = sZ[lane][c];
      Q0[(int64_t)(s + lane) + (int64_t)(s + c) * ld] = v;
      Q1[(int64_t)(s + lane) + (int64_t)(s + c) * ld] = v;
    }
  }
}

// zero the m x m diagonal blocks (s, m) = blocks[b] of both copies of Q: one workgroup per block column (16-byte stores
// where the column start allows; the 2-D memset of the runtime took as long as zeroing the whole matrices)
__global__ __launch_bounds__(256) void dc_zero_blocks(const int2* __restrict__ blocks, double* __restrict__ Q0,
                                                      double* __restrict__ Q1, int64_t ld) {
  const int2 sm = blocks[blockIdx.y];
  const int c = blockIdx.x;
  if (c >= sm.y) return;
  const int64_t o = sm.x + (int64_t)(sm.x + c) * ld;
  double* a = Q0 + o;
  double* b = Q1 + o;
  for (int r = threadIdx.x; r < sm.y; r += 256) { a[r] = 0.0; b[r] = 0.0; }
}

__global__ void dc_init_identity(double* __restrict__ Q, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) Q[(int64_t)i * n + i] = 1.0;
}

// z[s+t] = last row of the left child (t < n1) / first row of the right child
__global__ void dc_gather_z(const MergeDesc* __restrict__ descs, const double* __restrict__ Q,
                            int64_t ld, double* __restrict__ z) {
  const MergeDesc d = descs[blockIdx.y];
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= d.m) return;
  const int row = (t < d.n1) ? d.s + d.n1 - 1 : d.s + d.n1;
  z[d.s + t] = Q[row + (int64_t)(d.s + t) * ld];
}

// apply the deflation rotations of one merge, in order, to every row
__global__ void dc_rotate(const MergeDesc* __restrict__ descs, const int* __restrict__ merge_ids,
                          const int* __restrict__ ra, const int* __restrict__ rb,
                          const double* __restrict__ rc, const double* __restrict__ rs,
                          double* __restrict__ Q, int64_t ld) {
  const MergeDesc d = descs[merge_ids[blockIdx.y]];
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= d.m) return;
  double* q = Q + (d.s + r) + (int64_t)d.s * ld;
  for (int t = 0; t < d.nrot; ++t) {
    const int a = ra[d.rot_off + t], b = rb[d.rot_off + t];
    const double c = rc[d.rot_off + t], s = rs[d.rot_off + t];
    const double xa = q[(int64_t)a * ld], xb = q[(int64_t)b * ld];
    q[(int64_t)a * ld] = c * xa + s * xb;
    q[(int64_t)b * ld] = c * xb - s * xa;
  }
}

// 1 / x from the hardware seed and two Newton steps (5 instructions against ~20 for the IEEE division sequence; the
// result is within an ulp of the quotient, no special cases: the secular sums only see nonzero, finite differences).
// The secular solve is bound by the instructions it issues per term: K^2 terms per Newton pass, 14 496^2 at the root
// of an N = 20 000 decomposition.
__device__ __forceinline__ double rcp_fast(double x) {
  double r = __builtin_amdgcn_rcp(x);
  r = fma(fma(-x, r, 1.0), r, r);
  r = fma(fma(-x, r, 1.0), r, r);
  return r;
}

// One wave per root of 1 + rho * sum_i w_i^2/(dlam_i - lambda) = 0.
// Writes lambda_j and the pair (dorg_j, tau_j) with lambda_j = dorg_j + tau_j, dorg_j the pole the root is measured
// from: delta_ij = dlam_i - lambda_j is evaluated where it is needed as (dlam_i - dorg_j) - tau_j, the form that
// keeps its relative accuracy (LAPACK dlaed4's delta), instead of being stored as a K x K matrix -- at the root of an
// N = 20 000 decomposition that matrix was 1.7 GB written row-scattered here and read column-strided by dc_zhat.
// R roots per wave (round 6): every evaluation of the secular function streams the K poles and weights -- 232 KB per
// root and evaluation at the root of an N = 20 000 tree, ~100 000 evaluations: the kernel was bound by L2 bandwidth, not
// by its reciprocals. A wave now carries DC_SEC_R neighbouring roots through their iterations together: dl_i and w_i^2
// are loaded once per pole for all of them. Per root the arithmetic, the order of the sums and the iteration are exactly
// those of one root per wave (bitwise the same roots); a root that has converged simply stops updating its state while
// its neighbours finish.
constexpr int DC_SEC_R = 4;
__global__ __launch_bounds__(256) void dc_secular(const MergeDesc* __restrict__ descs,
                                                  const double* __restrict__ dlam,
                                                  const double* __restrict__ w,
                                                  double* __restrict__ lam, double* __restrict__ dorg_out,
                                                  double* __restrict__ tau_out) {
  constexpr int R = DC_SEC_R;
  const MergeDesc d = descs[blockIdx.y];
  const int lane = threadIdx.x & 63;
  const int j0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * R;
  const int K = d.K;
  if (j0 >= K) return;
  const int base = d.s;
  const double* dl = dlam + base;
  const double* ww = w + base;
  const double rho = d.rho;
  int org[R];
  double sgn[R], lo[R], hi[R], dorg[R], pa[R], pb[R], x[R];
  bool live[R], two_poles[R];
  // ---- brackets: the value at the midpoint of the root's interval (all but the last root), one pass over the poles -----
  {
    double dj[R], half[R], rl[R], rr[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int j = j0 + r;
      live[r] = j < K;
      const int jc = live[r] ? j : K - 1;                 // (a slot past the end repeats the last root and is never stored)
      dj[r] = dl[jc];
      half[r] = (jc < K - 1) ? 0.5 * (dl[jc + 1] - dj[r]) : 0.0;
      rl[r] = 0.0; rr[r] = 0.0;
    }
    double s2 = 0.0;
    for (int i = lane; i < K; i += 64) {
      const double wi = ww[i], w2 = wi * wi, di = dl[i];
      s2 += w2;
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const int jc = min(j0 + r, K - 1);
        if (jc < K - 1) {                                  // (uniform per r)
          const double t = w2 * rcp_fast((di - dj[r]) - half[r]);
          if (i > jc) rr[r] += t;
          else rl[r] -= t;
        }
      }
    }
    s2 = wsum(s2);
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int jc = min(j0 + r, K - 1);
      if (jc < K - 1) {
        const double srr = wsum(rr[r]), srl = wsum(rl[r]);
        const double g = 1.0 + rho * (srr - srl);
        hi[r] = half[r];
        if (g >= 0.0) {
          org[r] = jc; sgn[r] = 1.0;
          const double Rr = 1.0 + rho * srr;
          lo[r] = rho * ww[jc] * ww[jc] / Rr;
        } else {
          org[r] = jc + 1; sgn[r] = -1.0;
          const double Rr = -1.0 + rho * srl;
          lo[r] = (Rr > 0.0) ? rho * ww[jc + 1] * ww[jc + 1] / Rr : 0.0;
        }
      } else {
        org[r] = jc; sgn[r] = 1.0;
        hi[r] = rho * s2 * (1.0 + 8.0 * DEPS);
        lo[r] = rho * ww[jc] * ww[jc];
      }
      lo[r] = fmin(lo[r], hi[r]) * (1.0 - 8.0 * DEPS);
      if (!(lo[r] > 0.0)) lo[r] = hi[r] * 1e-300;
      dorg[r] = dl[org[r]];
      // the two poles next to the root, measured from the origin (one of them is 0); the last root has no upper pole
      two_poles[r] = jc < K - 1;
      pa[r] = two_poles[r] ? dl[jc] - dorg[r] : 0.0;
      pb[r] = two_poles[r] ? dl[jc + 1] - dorg[r] : 0.0;
      x[r] = (hi[r] > 4.0 * lo[r]) ? sqrt(lo[r]) * sqrt(hi[r]) : 0.5 * (lo[r] + hi[r]);
    }
  }
  // Safeguarded Newton inside the bracket [lo, hi] (in |tau|): every evaluation yields f, f' and
  // the sum of absolute terms (the rounding-error scale of f); the bracket is updated from the
  // sign of f, the Newton step is taken when it stays strictly inside the bracket and otherwise
  // the (geometric) midpoint. Stops when |f| is at the rounding level of its own evaluation
  // (LAPACK dlaed4's criterion) or the bracket has collapsed; the bisection alone needed ~60-100
  // evaluations per root, this needs ~6-12.
  bool run[R];
#pragma unroll
  for (int r = 0; r < R; ++r) run[r] = live[r];
  for (int it = 0; it < 200; ++it) {
    bool any = false;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      if (run[r] && (!(hi[r] > lo[r]) || !(x[r] > lo[r]) || !(x[r] < hi[r]))) run[r] = false;
      any = any || run[r];
    }
    if (!any) break;                                       // uniform (every lane holds the same scalars)
    double tau[R], psi[R], phi[R], dpsi[R], dphi[R];
#pragma unroll
    for (int r = 0; r < R; ++r) { tau[r] = sgn[r] * x[r]; psi[r] = phi[r] = dpsi[r] = dphi[r] = 0.0; }
    // psi: the poles up to j (terms < 0), phi: the poles above (terms > 0); derivatives likewise
    for (int i = lane; i < K; i += 64) {
      const double wi = ww[i], w2 = wi * wi, di = dl[i];
#pragma unroll
      for (int r = 0; r < R; ++r) {
        if (run[r]) {                                      // (uniform per r)
          const double inv = rcp_fast((di - dorg[r]) - tau[r]);
          const double t = w2 * inv;
          if (i > j0 + r) { phi[r] += t; dphi[r] += t * inv; }
          else            { psi[r] += t; dpsi[r] += t * inv; }
        }
      }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
      if (!run[r]) continue;
      const double spsi = wsum(psi[r]), sphi = wsum(phi[r]);
      const double sdpsi = rho * wsum(dpsi[r]), sdphi = rho * wsum(dphi[r]);
      const double g = 1.0 + rho * (spsi + sphi);
      const double gp = sdpsi + sdphi;                     // df/dtau > 0
      const double ga = 1.0 + rho * (sphi - spsi);         // sum of the absolute terms
      const bool pos = (g >= 0.0);
      if (sgn[r] > 0.0) { if (pos) hi[r] = x[r]; else lo[r] = x[r]; }
      else              { if (pos) lo[r] = x[r]; else hi[r] = x[r]; }
      if (fabs(g) <= 8.0 * DEPS * ga) { lo[r] = x[r]; hi[r] = x[r]; run[r] = false; continue; }
      // Step: psi and phi are each replaced by s + a / (pole - tau) through the nearest pole below / above with the
      // value and slope they have here (the "middle way" of Li 1994, the scheme of LAPACK's dlaed4): the resulting
      // quadratic in the increment eta is solved in the form that does not cancel. A plain Newton step in tau where
      // that is not available (last root) or leaves the bracket; the (geometric) midpoint when Newton leaves it too.
      double xn = sgn[r] * (tau[r] - g / gp);
      if (two_poles[r]) {
        const double DA = pa[r] - tau[r], DB = pb[r] - tau[r];   // < 0 < (inside the interval)
        const double c = g - DA * sdpsi - DB * sdphi;
        const double a = (DA + DB) * g - DA * DB * gp;
        const double b = DA * DB * g;
        double eta;
        if (c == 0.0) {
          eta = b / a;
        } else {
          const double disc = sqrt(fabs(a * a - 4.0 * b * c));
          eta = (a <= 0.0) ? (a - disc) / (2.0 * c) : 2.0 * b / (a + disc);
        }
        const double xr = sgn[r] * (tau[r] + eta);
        if (g * eta < 0.0 && xr > lo[r] && xr < hi[r]) xn = xr;  // (f increases: the step must go against the sign of f)
      }
      const double mid = (hi[r] > 4.0 * lo[r]) ? sqrt(lo[r]) * sqrt(hi[r]) : 0.5 * (lo[r] + hi[r]);
      x[r] = (xn > lo[r] && xn < hi[r]) ? xn : mid;
    }
  }
  if (lane == 0) {
#pragma unroll
    for (int r = 0; r < R; ++r) {
      if (live[r]) {
        const double tau = sgn[r] * 0.5 * (lo[r] + hi[r]);
        lam[base + j0 + r] = dorg[r] + tau;
        dorg_out[base + j0 + r] = dorg[r];
        tau_out[base + j0 + r] = tau;
      }
    }
  }
}

// Loewner: zhat_i = sign(w_i) sqrt| delta_ii * prod_{j != i} delta_ij / (dlam_i - dlam_j) |, delta_ij from the roots'
// (dorg_j, tau_j) pairs: everything this reads is K-long and cache-resident. One wave per pole (row rowpos(i)).
__global__ __launch_bounds__(256) void dc_zhat(const MergeDesc* __restrict__ descs,
                                               const double* __restrict__ dlam,
                                               const double* __restrict__ w,
                                               const int* __restrict__ pole_of_row,
                                               const double* __restrict__ dorg, const double* __restrict__ tau,
                                               double* __restrict__ zhat) {
  const MergeDesc d = descs[blockIdx.y];
  const int lane = threadIdx.x & 63;
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int K = d.K;
  if (r >= K) return;
  const int base = d.s;
  const int i = pole_of_row[base + r];
  const double di = dlam[base + i];
  double p = 1.0;
  for (int j = lane; j < K; j += 64) {
    const double delta = (di - dorg[base + j]) - tau[base + j];
    p *= (j == i) ? delta : delta * rcp_fast(di - dlam[base + j]);
  }
  p = wprod(p);
  if (lane == 0) zhat[base + r] = copysign(sqrt(fabs(p)), w[base + i]);
}

// U[:,cp] = normalised zhat / delta[:,cp]   (one block per kept column; row r <-> pole pole_of_row(r), column cp <->
// root cp, or root K-1-cp where the columns are stored in descending order: the root of the tree, MergeDesc::rev)
__global__ __launch_bounds__(256) void dc_vectors(const MergeDesc* __restrict__ descs,
                                                  const double* __restrict__ zhat,
                                                  const double* __restrict__ dlam,
                                                  const int* __restrict__ pole_of_row,
                                                  const double* __restrict__ dorg, const double* __restrict__ tau,
                                                  double* __restrict__ U, const double* __restrict__ gf,
                                                  const double* __restrict__ gl, double* __restrict__ of,
                                                  double* __restrict__ ol) {
  // Column cp of the merge's secular-vector matrix, normalised: the entries are formed twice (norm, then value) rather
  // than stored, re-read and rescaled -- a division is cheaper than two more passes over K x K doubles (0.96 GB per pass
  // at the level below the root of an N = 20 000 tree). With gf != nullptr (a level whose operators stay factored) the
  // column's products with the gathered first / last rows of the children, of[j] = gf' u_j and ol[j] = gl' u_j, are
  // taken from the same pass (they were a kernel of their own that read the block once more).
  __shared__ double sh[4];
  const MergeDesc d = descs[blockIdx.y];
  const int cp = blockIdx.x;
  if (cp >= d.Kneed) return;
  const int base = d.s;
  const int j = d.rev ? d.K - 1 - cp : cp;
  const double dj = dorg[base + j], tj = tau[base + j];
  double* u = U + d.uoff + (int64_t)cp * d.uld;
  const double* zh = zhat + base;
  double ss = 0.0;
  for (int r = threadIdx.x; r < d.K; r += 256) {
    const double v = zh[r] / ((dlam[base + pole_of_row[base + r]] - dj) - tj);
    ss += v * v;
  }
  ss = bsum256(ss, sh);
  const double inv = 1.0 / sqrt(ss);
  double a = 0.0, b = 0.0;
  for (int r = threadIdx.x; r < d.K; r += 256) {
    const double v = zh[r] / ((dlam[base + pole_of_row[base + r]] - dj) - tj);
    const double un = v * inv;
    u[r] = un;
    if (gf != nullptr) {       // (uniform)
      a += gf[base + r] * un;
      b += gl[base + r] * un;
    }
  }
  if (gf != nullptr) {
    a = bsum256(a, sh);
    b = bsum256(b, sh);
    if (threadIdx.x == 0) { of[base + cp] = a; ol[base + cp] = b; }
  }
}

__global__ void dc_copy_deflated(const MergeDesc* __restrict__ descs,
                                 const int* __restrict__ defsrc, const int* __restrict__ defdst,
                                 const double* __restrict__ Qc, double* __restrict__ Qn,
                                 int64_t ld) {
  const MergeDesc d = descs[blockIdx.y];
  const int idx = blockIdx.x;
  if (idx >= d.ndef) return;
  const int src = defsrc[d.s + idx], dst = defdst[d.s + idx];
  const double* a = Qc + d.s + (int64_t)(d.s + src) * ld;
  double* b = Qn + d.s + (int64_t)(d.s + dst) * ld;
  for (int r = threadIdx.x; r < d.m; r += blockDim.x) b[r] = a[r];
}

// ---- top levels kept factored (few eigenvectors wanted) --------------------------------------
// yf[s+t] = first row of the left child's eigenvector matrix (t < n1, else 0), yl[s+t] = last row
// of the right child's (t >= n1, else 0): the first and last row of the parent's block matrix
__global__ void dc_gather_rows(const MergeDesc* __restrict__ descs, const double* __restrict__ Q,
                               int64_t ld, double* __restrict__ yf, double* __restrict__ yl) {
  const MergeDesc d = descs[blockIdx.y];
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= d.m) return;
  const int64_t col = (int64_t)(d.s + t) * ld;
  yf[d.s + t] = (t < d.n1) ? Q[d.s + col] : 0.0;
  yl[d.s + t] = (t < d.n1) ? 0.0 : Q[d.s + d.m - 1 + col];
}

// X[:, t] = column src_cols[t] of the root's merge operator (X zeroed beforehand): a non-deflated
// column c < K carries U[:, c] in the rows srccol, a deflated one is a unit vector
__global__ void dc_lazy_root_x(int nv, const int* __restrict__ cols, int K,
                               const int* __restrict__ srccol, const int* __restrict__ defsrc,
                               const double* __restrict__ U, int64_t ldu, double* __restrict__ X,
                               int64_t ldx) {
  const int t = blockIdx.x;
  if (t >= nv) return;
  const int c = cols[t];
  double* x = X + (int64_t)t * ldx;
  if (c < K) {
    const double* u = U + (int64_t)c * ldu;
    for (int i = threadIdx.x; i < K; i += blockDim.x) x[srccol[i]] = u[i];
  } else if (threadIdx.x == 0) {
    x[defsrc[c - K]] = 1.0;
  }
}

// Xn = (merge operator) X for every merge of a level: rows srccol take T = U X[0:K], rows defsrc
// take the deflated rows X[K:m]
__global__ void dc_lazy_scatter(const MergeDesc* __restrict__ descs, const int* __restrict__ srccol,
                                const int* __restrict__ defsrc, const double* __restrict__ Tm,
                                const double* __restrict__ X, double* __restrict__ Xn, int64_t ld) {
  const MergeDesc d = descs[blockIdx.y];
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= d.m) return;
  const int64_t col = (int64_t)blockIdx.z * ld;
  if (r < d.K) Xn[d.s + srccol[d.s + r] + col] = Tm[d.s + r + col];
  else Xn[d.s + defsrc[d.s + r - d.K] + col] = X[d.s + r + col];
}

// the recorded column rotations Q <- Q G_1 ... G_r of a merge, moved to the other factor:
// X <- G_1 (G_2 (... (G_r X))) on the rows of X, one thread per column of X
__global__ void dc_lazy_rotate(const MergeDesc* __restrict__ descs, const int* __restrict__ merge_ids,
                               const int* __restrict__ ra, const int* __restrict__ rb,
                               const double* __restrict__ rc, const double* __restrict__ rs,
                               double* __restrict__ X, int64_t ld, int nv) {
  const MergeDesc d = descs[merge_ids[blockIdx.y]];
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= nv) return;
  double* x = X + d.s + (int64_t)t * ld;
  for (int i = d.nrot - 1; i >= 0; --i) {
    const int a = ra[d.rot_off + i], b = rb[d.rot_off + i];
    const double c = rc[d.rot_off + i], s = rs[d.rot_off + i];
    const double xa = x[a], xb = x[b];
    x[a] = c * xa - s * xb;
    x[b] = s * xa + c * xb;
  }
}

#ifdef BK_FAULT_INJECT
// test build only: keeps the stream busy for `ticks` of the 100 MHz clock, so that what is queued behind it executes late
__global__ void dc_fault_spin(long long ticks) {
  const unsigned long long t0 = wall_clock64();
  while ((long long)(wall_clock64() - t0) < ticks) {}
}
// test build only: one eigenvalue moved to the next representable double (a last-bit deviation of ONE rank's replica)
__global__ void fault_nudge_ulp(double* v) { *v = nextafter(*v, 2.0 * *v + 1.0); }
// test build only: two eigenvector columns exchanged (orthonormal columns, right eigenvalues, wrong pairing)
__global__ void fault_swap_cols(double* a, double* b, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) { const double t = a[i]; a[i] = b[i]; b[i] = t; }
}
#endif
__global__ void dc_iota(int* __restrict__ p, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = i;
}

__global__ void gather_cols(int n, int nv, const int* __restrict__ src, const double* __restrict__ Q,
                            int64_t ldq, double* __restrict__ Z, int64_t ldz) {
  const int64_t total = (int64_t)n * nv;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (int64_t)gridDim.x * blockDim.x) {
    const int r = (int)(e % n), c = (int)(e / n);
    Z[r + (int64_t)c * ldz] = Q[r + (int64_t)src[c] * ldq];
  }
}

struct Node {
  int s, m, left, right, depth;
  std::vector<double> dv;  // eigenvalue of storage column s+t
  int ksorted = 0;         // dv[0 .. ksorted) is ascending (the secular roots of the merge that produced the node)
};

// host side of one merge (LAPACK dlaed2's scan, re-derived): fills the packed
// per-level arrays at offset s.
// The n-long arrays are slices of the context's pinned upload arena (PinnedStage::fixed), laid out like their device
// copies -- [dlam | w] and [pole_of_row | srccol | cpos | defsrc | defdst] -- so that a level uploads them with two
// copies straight from where host_merge wrote them.
struct LevelArrays {
  double *dlam = nullptr, *w = nullptr;
  int *pole_of_row = nullptr, *srccol = nullptr, *cpos = nullptr, *defsrc = nullptr, *defdst = nullptr;
  std::vector<int> rowpos;
  std::vector<double> rc, rs;
  std::vector<int> ra, rb;
};

// one level whose merge operators stay factored (see divide_conquer)
struct LazyLevel {
  std::vector<MergeDesc> descs;
  std::vector<int> srccol, defsrc, ra, rb, rot_merges;
  std::vector<double> rc, rs;
  std::vector<int64_t> stash_off;   // offset of every merge's K x K secular-vector block in the stash
  int max_m = 0, maxK = 0;
};

void host_merge(const Node& L, const Node& R, double ecut, const double* zraw, MergeDesc& md,
                LevelArrays& A, std::vector<double>& defvals) {
  const int n1 = L.m, n2 = R.m, m = n1 + n2, s = L.s;
  const double theta = (ecut >= 0.0) ? 1.0 : -1.0;
  const double rho = 2.0 * std::fabs(ecut);
  // scratch reused across the merges of a level (thousands of small merges at the bottom of the tree)
  static thread_local std::vector<double> z, D;
  static thread_local std::vector<std::pair<double, int>> keyed, kbuf;
  static thread_local std::vector<int> order, typ, nd, df;
  z.resize(m); D.resize(m); keyed.resize(m); kbuf.resize(m); order.resize(m); typ.resize(m);
  nd.clear(); df.clear();
  const double isq2 = 1.0 / std::sqrt(2.0);
  double dmax = 0.0, zmax = 0.0;
  for (int t = 0; t < m; ++t) {
    z[t] = zraw[s + t] * ((t < n1) ? 1.0 : theta) * isq2;
    D[t] = (t < n1) ? L.dv[t] : R.dv[t - n1];
    keyed[t] = {D[t], t};
    typ[t] = (t < n1) ? 1 : 3;
    dmax = std::max(dmax, std::fabs(D[t]));
    zmax = std::max(zmax, std::fabs(z[t]));
  }
  // ascending in D, ties in storage order: what a stable sort of the indices gives, on (value, index) pairs. Each
  // child's list is its ascending secular roots followed by its deflated values: only the two tails are sorted, the
  // four runs are merged (the root of an N = 20 000 tree: 0.25 instead of 0.7 ms)
  {
    int kl = std::min(L.ksorted, n1), kr = std::min(R.ksorted, m - n1);
    if (!std::is_sorted(keyed.begin(), keyed.begin() + kl)) kl = 0;          // (never seen; costs O(K))
    if (!std::is_sorted(keyed.begin() + n1, keyed.begin() + n1 + kr)) kr = 0;
    std::sort(keyed.begin() + kl, keyed.begin() + n1);
    std::sort(keyed.begin() + n1 + kr, keyed.end());
    // (std::merge into the scratch list and back: std::inplace_merge allocates its buffer at every call -- three calls
    //  per merge, thousands of merges per level)
    std::merge(keyed.begin(), keyed.begin() + kl, keyed.begin() + kl, keyed.begin() + n1, kbuf.begin());
    std::merge(keyed.begin() + n1, keyed.begin() + n1 + kr, keyed.begin() + n1 + kr, keyed.end(), kbuf.begin() + n1);
    std::merge(kbuf.begin(), kbuf.begin() + n1, kbuf.begin() + n1, kbuf.end(), keyed.begin());
  }
  for (int t = 0; t < m; ++t) order[t] = keyed[t].second;
  const double tol = 8.0 * DEPS * std::max(dmax, zmax);
  md.s = s; md.n1 = n1; md.m = m; md.rho = rho;
  md.rot_off = (int)A.ra.size(); md.nrot = 0;
  if (rho * zmax <= tol) {
    for (int t : order) df.push_back(t);
  } else {
    int pj = -1;
    for (int t : order) {
      if (rho * std::fabs(z[t]) <= tol) { df.push_back(t); typ[t] = 4; continue; }
      if (pj < 0) { pj = t; continue; }
      double sv = z[pj], cv = z[t];
      const double tt = D[t] - D[pj];
      // dlaed2's test |tt c s| <= tol with c = z_t / tau, s = -z_pj / tau, tau = hypot(z_t, z_pj), in the form that
      // needs neither the root nor the divisions (z is normalised: no overflow): they are only formed for the rare pair
      // that deflates -- the scan visits every non-deflated entry of every level, and hypot + two divisions per entry
      // were most of its time
      const double ss = cv * cv + sv * sv;
      const double tau_safe = (ss > 1e-280) ? 0.0 : std::hypot(cv, sv);     // (squares that underflow: the safe form)
      const bool defl = (ss > 1e-280) ? (std::fabs(tt * cv * sv) <= tol * ss)
                                      : (std::fabs(tt * (cv / tau_safe) * (sv / tau_safe)) <= tol);
      if (defl) {
        const double tau = (ss > 1e-280) ? std::sqrt(ss) : tau_safe;
        cv /= tau; sv = -sv / tau;
        z[t] = tau; z[pj] = 0.0;
        if (typ[t] != typ[pj]) typ[t] = 2;
        typ[pj] = 4;
        A.ra.push_back(pj); A.rb.push_back(t); A.rc.push_back(cv); A.rs.push_back(sv);
        md.nrot++;
        const double tmp = D[pj] * cv * cv + D[t] * sv * sv;
        D[t] = D[pj] * sv * sv + D[t] * cv * cv;
        D[pj] = tmp;
        df.push_back(pj);
        pj = t;
      } else {
        nd.push_back(pj);
        pj = t;
      }
    }
    if (pj >= 0) nd.push_back(pj);
  }
  const int K = (int)nd.size();
  md.K = K; md.Kneed = K; md.ndef = m - K;
  // poles ascending = nd order; rows of U grouped by column type 1,2,3
  int k1 = 0, k2 = 0, k3 = 0;
  for (int t : nd) { if (typ[t] == 1) ++k1; else if (typ[t] == 2) ++k2; else ++k3; }
  md.k1 = k1; md.k2 = k2; md.k3 = k3;
  int p1 = 0, p2 = k1, p3 = k1 + k2;
  for (int i = 0; i < K; ++i) {
    const int t = nd[i];
    A.dlam[s + i] = D[t];
    A.w[s + i] = z[t];
    int row = (typ[t] == 1) ? p1++ : (typ[t] == 2) ? p2++ : p3++;
    A.rowpos[s + i] = row;
    A.pole_of_row[s + row] = i;
    A.srccol[s + row] = t;
    A.cpos[s + i] = i;
  }
  defvals.resize(m - K);
  for (int q = 0; q < m - K; ++q) {
    A.defsrc[s + q] = df[q];
    A.defdst[s + q] = K + q;
    defvals[q] = D[df[q]];
  }
}

// pinned upload arena of one decomposition: the level arrays of the divide & conquer + one level's other uploads
static size_t dc_stage_bytes(int n) {
  return (size_t)n * (2 * sizeof(double) + 5 * sizeof(int)) + (size_t)n * 96 + ((size_t)2 << 20);
}

int divide_conquer(bigkrls_ctx* ctx, int n, const std::vector<double>& hd,
                   const std::vector<double>& he, double* Q0, double* Q1, double* U,
                   int64_t n_vals, int64_t n_vecs_max, double keep_thresh,
                   std::vector<double>& vals_desc, std::vector<int>& src_cols, double** Qfinal,
                   bool allow_lazy = true) {
  hipStream_t st = ctx->stream;
  const int64_t N = n;
  // ---- tree -----------------------------------------------------------------
  // (BIGKRLS_DCLEAF=1: tear down to 1 x 1 leaves, no QL leaf solver)
  const char* leaf_env = getenv("BIGKRLS_DCLEAF");
  const int leaf_max = (n > 2 * DC_LEAF && !(leaf_env && std::string(leaf_env) == "1")) ? DC_LEAF : 1;
  std::vector<Node> nodes;
  nodes.reserve(2 * n);
  std::vector<double> dadj = hd;
  {
    struct Item { int s, m, depth, parent, side; };
    std::vector<Item> stack;
    stack.push_back({0, n, 0, -1, 0});
    while (!stack.empty()) {
      Item it = stack.back();
      stack.pop_back();
      Node nd;
      nd.s = it.s; nd.m = it.m; nd.left = nd.right = -1; nd.depth = it.depth;
      const int id = (int)nodes.size();
      nodes.push_back(nd);
      if (it.parent >= 0) {
        if (it.side == 0) nodes[it.parent].left = id; else nodes[it.parent].right = id;
      }
      if (it.m > leaf_max) {
        const int n1 = it.m / 2;
        const int cut = it.s + n1 - 1;  // e[cut] couples rows cut, cut+1
        const double rho = std::fabs(he[cut]);
        dadj[cut] -= rho;
        dadj[cut + 1] -= rho;
        stack.push_back({it.s, n1, it.depth + 1, id, 0});
        stack.push_back({it.s + n1, it.m - n1, it.depth + 1, id, 1});
      }
    }
  }
  int maxdepth = 0;
  for (auto& nd : nodes) {
    maxdepth = std::max(maxdepth, nd.depth);
    if (nd.m == 1) nd.dv.assign(1, dadj[nd.s]);
  }
  std::vector<std::vector<int>> by_depth(maxdepth + 1);
  for (int id = 0; id < (int)nodes.size(); ++id)
    if (nodes[id].left >= 0) by_depth[nodes[id].depth].push_back(id);

  // ---- device state ---------------------------------------------------------
  // every host -> device upload below goes through the context's pinned arena: the level arrays live in it (uploaded
  // from where they are written), everything else is copied into a slice of the current level's cycle
  PinnedStage stage(ctx);
  BK_TRY(stage.reserve(dc_stage_bytes(n)));
  LevelArrays A;
  A.dlam = (double*)stage.fixed((size_t)2 * n * sizeof(double));
  int* a_int = (int*)stage.fixed((size_t)5 * n * sizeof(int));
  BK_REQUIRE(A.dlam && a_int, "divide & conquer: pinned arena too small");
  A.w = A.dlam + n;
  A.pole_of_row = a_int; A.srccol = a_int + n; A.cpos = a_int + 2 * n; A.defsrc = a_int + 3 * n; A.defdst = a_int + 4 * n;
  std::memset(A.dlam, 0, (size_t)2 * n * sizeof(double));
  std::memset(a_int, 0, (size_t)5 * n * sizeof(int));
  A.rowpos.resize(n);
  // Few eigenvectors wanted (truncation or Neig << N): the merge operators of the top LAZY_TOP
  // levels below the root stay factored. Q of depth Dl+1 is the last one formed; above it only the
  // first and last row of every node's eigenvector matrix (what the parent's z needs) are
  // propagated, the K x K secular-vector blocks are stashed, and at the end the operators are
  // applied right-to-left to the N x nv kept columns of the root:
  //   Q[:, kept] = Q_{Dl+1} M_Dl ... M_1 M_0[:, kept]      (2 K^2 nv flops per merge instead of 2 m K^2).
  // BIGKRLS_DC=explicit forms every level, =factored forces this path at any size; a root that
  // keeps more than N/8 columns redoes the divide & conquer explicitly.
  constexpr int LAZY_TOP = 3;
  const char* dc_env = getenv("BIGKRLS_DC");
  const std::string dc_mode = dc_env ? dc_env : "";
  const bool lazy = allow_lazy && maxdepth - 1 > LAZY_TOP && dc_mode != "explicit" &&
                    ((n >= 4096 && (keep_thresh > 0.0 || n_vecs_max * 8 <= N)) ||
                     (dc_mode == "factored" && n >= 128));   // (forced: the tests run small hard spectra)
  const int Dl = lazy ? LAZY_TOP : -1;
  // Both copies of Q start as the identity. What is READ of them are the diagonal blocks of the nodes whose eigenvector
  // matrices are formed explicitly: a merge reads its two children's blocks and the zero blocks between them (rotations
  // and deflated columns run over the parent's full height) and writes the parent's whole block in the other copy.
  // With factored top levels the largest such blocks are the depth-Dl nodes' (8 blocks of 2 500^2 at N = 20 000: 0.8 GB
  // of zeros instead of 6.4 GB, -1.1 ms); without, the whole matrix.
  std::vector<int> zb;       // (s, m) of the depth-Dl nodes
  if (lazy)
    for (const Node& nd : nodes)
      if (nd.depth == Dl && nd.m > 0) { zb.push_back(nd.s); zb.push_back(nd.m); }
  if (lazy && !zb.empty()) {
    // (the list goes up through the integer workspace of the levels, which is first used after the leaves)
    void* pzi = nullptr;
    BK_TRY(ws_get(ctx, SLOT_EIG_INT, (int64_t)11 * n * sizeof(int), &pzi));
    int* zlist = (int*)stage.fixed(zb.size() * sizeof(int));     // (a slice that no later upload of this call reuses)
    BK_REQUIRE(zlist, "divide & conquer: pinned arena too small");
    std::memcpy(zlist, zb.data(), zb.size() * sizeof(int));
    BK_TRY(stage.send(pzi, zlist, zb.size() * sizeof(int)));
    int zmax = 0;
    for (size_t i = 1; i < zb.size(); i += 2) zmax = std::max(zmax, zb[i]);
    hipLaunchKernelGGL(dc_zero_blocks, dim3(zmax, (unsigned)(zb.size() / 2)), dim3(256), 0, st, (const int2*)pzi, Q0, Q1, N);
    BK_CHECK_LAUNCH();
  } else {
    BK_HIP(hipMemsetAsync(Q0, 0, (size_t)N * N * sizeof(double), st));
    BK_HIP(hipMemsetAsync(Q1, 0, (size_t)N * N * sizeof(double), st));
  }
  hipLaunchKernelGGL(dc_init_identity, dim3((n + 255) / 256), dim3(256), 0, st, Q0, n);
  hipLaunchKernelGGL(dc_init_identity, dim3((n + 255) / 256), dim3(256), 0, st, Q1, n);
  BK_CHECK_LAUNCH();
  // double arrays: z, dlam, w, lam, zhat, rc, rs  (7n) ; int arrays: 8n ; descs
  void* pd = nullptr;
  BK_TRY(ws_get(ctx, SLOT_EIG_MISC, (int64_t)14 * n * sizeof(double), &pd));
  double* d_z = (double*)pd;
  double* d_dlam = d_z + n;
  double* d_w = d_dlam + n;
  double* d_lam = d_w + n;
  double* d_zhat = d_lam + n;
  double* d_rc = d_zhat + n;
  double* d_rs = d_rc + n;
  double* d_gf = d_rs + n;      // factored levels: gathered first / last rows in, merged rows out
  double* d_gl = d_gf + n;
  double* d_of = d_gl + n;
  double* d_ol = d_of + n;
  double* d_dorg = d_ol + n;    // per root: the pole it is measured from, and its offset from that pole
  double* d_tau = d_dorg + n;
  void* pi = nullptr;
  BK_TRY(ws_get(ctx, SLOT_EIG_INT, (int64_t)11 * n * sizeof(int), &pi));
  int* d_rowpos = (int*)pi;
  int* d_pole = d_rowpos + n;
  int* d_srccol = d_pole + n;
  int* d_cpos = d_srccol + n;
  int* d_defsrc = d_cpos + n;
  int* d_defdst = d_defsrc + n;
  int* d_ra = d_defdst + n;
  int* d_rb = d_ra + n;
  int* d_rotids = d_rb + n;
  int* d_iota = d_rotids + n;
  const int max_merges = n / 2 + 1;
  void* pdesc = nullptr;
  BK_TRY(ws_get(ctx, SLOT_EIG_DESC,
                (int64_t)max_merges * (sizeof(MergeDesc) + 2 * sizeof(GemmDesc)), &pdesc));
  MergeDesc* d_descs = (MergeDesc*)pdesc;
  GemmDesc* d_gdescs = (GemmDesc*)(d_descs + max_merges);

  double* Qc = Q0;
  double* Qn = Q1;
  std::vector<double> hz(n), hlam(n);

  // ---- leaves larger than 1 x 1: QL on the device, eigenvector blocks into both copies of Q --------
  if (leaf_max > 1) {
    const auto t_leaf = std::chrono::steady_clock::now();
    std::vector<int> lf;   // (s, m) pairs
    for (const Node& nd : nodes)
      if (nd.left < 0 && nd.m > 1) { lf.push_back(nd.s); lf.push_back(nd.m); }
    const int nleaf = (int)lf.size() / 2;
    if (nleaf > 0) {
      int* d_fail = d_iota;                        // (d_iota is filled later, only on the factored path)
      BK_HIP(hipMemsetAsync(d_fail, 0, sizeof(int), st));
      BK_TRY(stage.put(d_dlam, dadj.data(), n * sizeof(double)));
      BK_TRY(stage.put(d_w, he.data(), n * sizeof(double)));
      BK_TRY(stage.put(d_rowpos, lf.data(), lf.size() * sizeof(int)));
      hipLaunchKernelGGL(dc_leaf_ql, dim3(nleaf), dim3(64), 0, st, (const int2*)d_rowpos, (const double*)d_dlam,
                         (const double*)d_w, Q0, Q1, N, d_lam, d_fail);
      BK_CHECK_LAUNCH();
      int h_fail = 0;
      PinnedFetch pf(ctx, (int64_t)n + 1);
      BK_TRY(pf.add(hlam.data(), d_lam, n * sizeof(double)));
      BK_TRY(pf.add(&h_fail, d_fail, sizeof(int)));
      BK_TRY(pf.finish());
      if (h_fail != 0) {
        set_error("eigen: the QL iteration of a divide & conquer leaf did not converge");
        return BIGKRLS_ENOCONV;
      }
      for (Node& nd : nodes)
        if (nd.left < 0 && nd.m > 1) nd.dv.assign(hlam.begin() + nd.s, hlam.begin() + nd.s + nd.m);
      if (trace_fine()) {
        BK_TRY(trace_host("R:dc_leaf_lam", hlam.data(), n, nleaf));
        BK_TRY(trace_point(ctx, st, "R:dc_leaf_Q", Q0, N * N, nleaf));
      }
    }
    if (getenv("BIGKRLS_VERBOSE"))
      fprintf(stderr, "[bigkrls]   d&c leaves: %6d of up to %d rows (QL) %8.2f ms\n", nleaf, leaf_max,
              std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_leaf).count());
  }

  int64_t nv_final = 0;
  const bool verbose = getenv("BIGKRLS_VERBOSE") != nullptr;
  std::vector<LazyLevel> lazy_levels(lazy ? Dl + 1 : 0);   // indexed by depth
  std::vector<double> bf, bl, yf, yl;                      // boundary rows of the current frontier / of a level
  double* stash = nullptr;
  int64_t stash_used = 0;
  if (lazy) { bf.assign(n, 0.0); bl.assign(n, 0.0); yf.assign(n, 0.0); yl.assign(n, 0.0); }
  for (int depth = maxdepth - 1; depth >= 0; --depth) {
    const std::vector<int>& ids = by_depth[depth];
    const int nm = (int)ids.size();
    if (nm == 0) continue;
    const auto t_level = std::chrono::steady_clock::now();
    auto t_mark = t_level;
    double ms_split[5] = {0, 0, 0, 0, 0};   // (verbose) wait for z | host scans | uploads + secular | wait for the roots | rest
    auto lap = [&](int slot) {
      const auto now = std::chrono::steady_clock::now();
      ms_split[slot] += std::chrono::duration<double, std::milli>(now - t_mark).count();
      t_mark = now;
    };
    std::vector<MergeDesc> descs(nm);
    int max_m = 0;
    for (int q = 0; q < nm; ++q) {
      const Node& P = nodes[ids[q]];
      descs[q] = MergeDesc{};
      descs[q].s = P.s; descs[q].n1 = nodes[P.left].m; descs[q].m = P.m;
      max_m = std::max(max_m, P.m);
    }
    const bool lazy_level = lazy && depth <= Dl;   // this level's merge operators stay factored
    const bool lazy_kids = lazy && depth < Dl;     // ... and so do the children's eigenvector matrices
    stage.reset();      // (the previous level ended with a synchronisation: its uploads have executed)
    BK_TRY(stage.put(d_descs, descs.data(), nm * sizeof(MergeDesc)));
    if (!lazy_kids) {
      for (int b0 = 0; b0 < nm; b0 += 65535) {
        const int nb = std::min(65535, nm - b0);
        hipLaunchKernelGGL(dc_gather_z, dim3((max_m + 63) / 64, nb), dim3(64), 0, st,
                           (const MergeDesc*)(d_descs + b0), (const double*)Qc, N, d_z);
        if (lazy_level)
          hipLaunchKernelGGL(dc_gather_rows, dim3((max_m + 63) / 64, nb), dim3(64), 0, st,
                             (const MergeDesc*)(d_descs + b0), (const double*)Qc, N, d_gf, d_gl);
      }
      BK_CHECK_LAUNCH();
      PinnedFetch pf(ctx, 3 * (int64_t)n);
      BK_TRY(pf.add(hz.data(), d_z, n * sizeof(double)));
      if (lazy_level) {
        BK_TRY(pf.add(yf.data(), d_gf, n * sizeof(double)));
        BK_TRY(pf.add(yl.data(), d_gl, n * sizeof(double)));
      }
      BK_TRY(pf.finish());
    } else {
      for (int q = 0; q < nm; ++q) {
        const MergeDesc& md = descs[q];
        for (int t = 0; t < md.m; ++t) {
          const bool left = t < md.n1;
          hz[md.s + t] = left ? bl[md.s + t] : bf[md.s + t];
          yf[md.s + t] = left ? bf[md.s + t] : 0.0;
          yl[md.s + t] = left ? 0.0 : bl[md.s + t];
        }
      }
      BK_HIP(hipStreamSynchronize(st));
    }

    lap(0);
    if (trace_fine()) BK_TRY(trace_host("R:dc_z", hz.data(), n, depth));
    // host deflation scans
    A.ra.clear(); A.rb.clear(); A.rc.clear(); A.rs.clear();
    std::vector<std::vector<double>> defvals(nm);
    std::vector<int> rot_merges;
    int maxK = 0, max_ndef = 0;
    for (int q = 0; q < nm; ++q) {
      const Node& P = nodes[ids[q]];
      const Node& Ln = nodes[P.left];
      const Node& Rn = nodes[P.right];
      const int cut = P.s + Ln.m - 1;
      host_merge(Ln, Rn, he[cut], hz.data(), descs[q], A, defvals[q]);
      if (descs[q].nrot > 0) rot_merges.push_back(q);
      maxK = std::max(maxK, descs[q].K);
      max_ndef = std::max(max_ndef, descs[q].ndef);
    }
    const bool is_root = (depth == 0);
    // where this level's secular vectors are written: straight into the stash on a factored level below the root (their
    // K x K blocks used to be formed in U and copied: 1.9 GB of copies at N = 20 000), else block (s, s) of U
    const bool u_to_stash = lazy_level && !is_root;
    if (u_to_stash && stash == nullptr) stash = Qn;
    std::vector<int64_t> level_stash_off(u_to_stash ? nm : 0);
    for (int q = 0; q < nm; ++q) {
      MergeDesc& md = descs[q];
      if (u_to_stash) {
        level_stash_off[q] = stash_used;
        md.uoff = stash_used;
        md.uld = std::max(md.K, 1);
        stash_used += (int64_t)md.K * md.K;
      } else {
        md.uoff = md.s + (int64_t)md.s * N;
        md.uld = n;
      }
    }
    double* Ubase = u_to_stash ? stash : U;
    if (is_root) {
      // store roots in descending order so that the kept ones are a prefix
      const int K = descs[0].K, s = descs[0].s;
      for (int j = 0; j < K; ++j) A.cpos[s + j] = K - 1 - j;
      descs[0].rev = 1;
    }
    if (lazy_level && !is_root) {
      // the deflation rotations act on the columns of the children's matrices, hence on yf, yl
      for (int q = 0; q < nm; ++q) {
        const MergeDesc& md = descs[q];
        double* f = yf.data() + md.s;
        double* l = yl.data() + md.s;
        for (int t = 0; t < md.nrot; ++t) {
          const int a = A.ra[md.rot_off + t], b = A.rb[md.rot_off + t];
          const double c = A.rc[md.rot_off + t], sn = A.rs[md.rot_off + t];
          const double fa = f[a], fb = f[b], la = l[a], lb = l[b];
          f[a] = c * fa + sn * fb; f[b] = c * fb - sn * fa;
          l[a] = c * la + sn * lb; l[b] = c * lb - sn * la;
        }
      }
    }
    lap(1);
    BK_TRY(stage.put(d_descs, descs.data(), nm * sizeof(MergeDesc)));
    // [dlam | w] -> d_dlam, d_w and [pole_of_row | srccol | cpos | defsrc | defdst] -> d_pole ... d_defdst: the device
    // arrays are laid out the same way (d_cpos receives the host's cpos: no kernel of the level loop reads it)
    BK_TRY(stage.send(d_dlam, A.dlam, (size_t)2 * n * sizeof(double)));
    BK_TRY(stage.send(d_pole, A.pole_of_row, (size_t)5 * n * sizeof(int)));
    const int nrot_total = (int)A.ra.size();
    if (nrot_total > 0) {
      BK_TRY(stage.put(d_ra, A.ra.data(), nrot_total * sizeof(int)));
      BK_TRY(stage.put(d_rb, A.rb.data(), nrot_total * sizeof(int)));
      BK_TRY(stage.put(d_rc, A.rc.data(), nrot_total * sizeof(double)));
      BK_TRY(stage.put(d_rs, A.rs.data(), nrot_total * sizeof(double)));
      BK_TRY(stage.put(d_rotids, rot_merges.data(), rot_merges.size() * sizeof(int)));
      const int nrm = (int)rot_merges.size();
      for (int b0 = 0; b0 < nrm && !lazy_kids; b0 += 65535) {
        const int nb = std::min(65535, nrm - b0);
        hipLaunchKernelGGL(dc_rotate, dim3((max_m + 63) / 64, nb), dim3(64), 0, st,
                           (const MergeDesc*)d_descs, (const int*)(d_rotids + b0), (const int*)d_ra,
                           (const int*)d_rb, (const double*)d_rc, (const double*)d_rs, Qc, N);
      }
      BK_CHECK_LAUNCH();
    }
    if (maxK > 0) {
      for (int b0 = 0; b0 < nm; b0 += 65535) {
        const int nb = std::min(65535, nm - b0);
        hipLaunchKernelGGL(dc_secular, dim3((maxK + 4 * DC_SEC_R - 1) / (4 * DC_SEC_R), nb), dim3(256), 0, st,
                           (const MergeDesc*)(d_descs + b0), (const double*)d_dlam,
                           (const double*)d_w, d_lam, d_dorg, d_tau);
        hipLaunchKernelGGL(dc_zhat, dim3((maxK + 3) / 4, nb), dim3(256), 0, st,
                           (const MergeDesc*)(d_descs + b0), (const double*)d_dlam,
                           (const double*)d_w, (const int*)d_pole, (const double*)d_dorg,
                           (const double*)d_tau, d_zhat);
      }
      BK_CHECK_LAUNCH();
      lap(2);
      PinnedFetch pf(ctx, (int64_t)n);
      BK_TRY(pf.add(hlam.data(), d_lam, n * sizeof(double)));
      BK_TRY(pf.finish());
    }
    BK_HIP(hipStreamSynchronize(st));
    lap(3);
    if (trace_fine()) {
      BK_TRY(trace_host("R:dc_dlam", A.dlam, n, depth));
      BK_TRY(trace_host("R:dc_w", A.w, n, depth));
      BK_TRY(trace_host("R:dc_roots", hlam.data(), n, maxK));
    }
    // new eigenvalue lists in storage order
    for (int q = 0; q < nm; ++q) {
      Node& P = nodes[ids[q]];
      const int K = descs[q].K, s = P.s;
      P.dv.assign(P.m, 0.0);
      for (int j = 0; j < K; ++j) {
        const double lv = hlam[s + j];
        if (!std::isfinite(lv)) {
          set_error("eigen: secular equation produced a non-finite root");
          return BIGKRLS_ENOCONV;
        }
        P.dv[A.cpos[s + j]] = lv;
      }
      for (int t = 0; t < P.m - K; ++t) P.dv[K + t] = defvals[q][t];
      P.ksorted = (depth == 0) ? 0 : K;      // (the root stores its roots in descending order)
      nodes[P.left].dv.clear(); nodes[P.left].dv.shrink_to_fit();
      nodes[P.right].dv.clear(); nodes[P.right].dv.shrink_to_fit();
    }
    if (is_root) {
      const Node& P = nodes[ids[0]];
      // descending order of all n values, ties in storage order (= a stable sort of the indices): the K roots are stored
      // in descending order already, so only the deflated tail is sorted and the two runs are merged
      std::vector<int> ord(n);
      std::iota(ord.begin(), ord.end(), 0);
      {
        auto desc = [&](int a, int b) { return P.dv[a] > P.dv[b]; };
        const int Kr = descs[0].K;
        if (std::is_sorted(ord.begin(), ord.begin() + Kr, desc)) {
          std::stable_sort(ord.begin() + Kr, ord.end(), desc);
          std::inplace_merge(ord.begin(), ord.begin() + Kr, ord.end(), desc);
        } else {
          std::stable_sort(ord.begin(), ord.end(), desc);
        }
      }
      vals_desc.resize(n);
      for (int t = 0; t < n; ++t) vals_desc[t] = P.dv[ord[t]];
      int64_t nv = n_vecs_max;
      if (keep_thresh >= 0.0) {
        int64_t keep = 0;
        const double thr = keep_thresh * vals_desc[0];
        for (int64_t t = 0; t < n_vals; ++t)
          if (vals_desc[t] >= thr) keep = t + 1;   // max(which(values >= eigtrunc*values[1]))
        nv = std::min<int64_t>(nv, std::max<int64_t>(keep, 1));
      }
      nv_final = nv;
      src_cols.resize(nv);
      int kneed = 0;
      for (int64_t t = 0; t < nv; ++t) {
        src_cols[t] = ord[t];
        if (ord[t] < descs[0].K) ++kneed;
      }
      descs[0].Kneed = kneed;
      if (lazy && nv * 8 > N) {
        BK_HIP(hipStreamSynchronize(st));
        if (verbose) fprintf(stderr, "[bigkrls]   d&c: %lld columns kept, redoing the top levels explicitly\n", (long long)nv);
        return divide_conquer(ctx, n, hd, he, Q0, Q1, U, n_vals, n_vecs_max, keep_thresh, vals_desc,
                              src_cols, Qfinal, false);
      }
      BK_TRY(stage.put(d_descs, descs.data(), nm * sizeof(MergeDesc)));
    }
    // eigenvector update
    int maxKneed = 0;
    for (int q = 0; q < nm; ++q) maxKneed = std::max(maxKneed, descs[q].Kneed);
    // Descriptors of this level's batched products: a source of an asynchronous host -> device copy, declared at the
    // level's scope so that it lives until the synchronisation at the end of the level. (It used to be a local of the
    // branch that fills it, destroyed before that synchronisation. On this runtime that was harmless -- hipMemcpyAsync
    // takes its copy of a pageable source before it returns, at every size, also in bursts and under load:
    // tools/pageable_h2d_probe.hip, BIGKRLS_FAULT=dc_gd_clobber -- but nothing in the API promises it.)
    std::vector<GemmDesc> gd;
    if (u_to_stash) {
      // the children's first / last rows gathered into the order of the merge's secular-vector rows: dc_vectors pushes
      // them through the merge in the pass that writes the vectors
      std::vector<double> gfh(n, 0.0), glh(n, 0.0);
      for (int q = 0; q < nm; ++q) {
        const MergeDesc& md = descs[q];
        for (int i = 0; i < md.K; ++i) {
          gfh[md.s + i] = yf[md.s + A.srccol[md.s + i]];
          glh[md.s + i] = yl[md.s + A.srccol[md.s + i]];
        }
      }
      BK_TRY(stage.put(d_gf, gfh.data(), n * sizeof(double)));
      BK_TRY(stage.put(d_gl, glh.data(), n * sizeof(double)));
    }
    if (maxKneed > 0) {
      for (int b0 = 0; b0 < nm; b0 += 65535) {
        const int nb = std::min(65535, nm - b0);
        hipLaunchKernelGGL(dc_vectors, dim3(maxKneed, nb), dim3(256), 0, st,
                           (const MergeDesc*)(d_descs + b0), (const double*)d_zhat, (const double*)d_dlam,
                           (const int*)d_pole, (const double*)d_dorg, (const double*)d_tau, Ubase,
                           u_to_stash ? (const double*)d_gf : (const double*)nullptr, (const double*)d_gl, d_of, d_ol);
      }
      BK_CHECK_LAUNCH();
    }
    if (lazy_level) {
      LazyLevel& L = lazy_levels[depth];
      L.descs = descs;
      L.srccol.assign(A.srccol, A.srccol + n);
      L.defsrc.assign(A.defsrc, A.defsrc + n);
      L.max_m = max_m; L.maxK = maxK;
      if (lazy_kids) {   // (at depth Dl the rotations went into the explicit Q of depth Dl+1)
        L.ra = A.ra; L.rb = A.rb; L.rc = A.rc; L.rs = A.rs; L.rot_merges = rot_merges;
      } else {
        for (auto& md : L.descs) md.nrot = 0;
      }
      if (!is_root) {
        // the secular-vector blocks were written straight into the stash (the idle ping-pong copy of Q: the U buffer is
        // reused by the next level); push the boundary rows through the merge
        L.stash_off = level_stash_off;
        std::vector<double> ofh(n), olh(n);
        PinnedFetch pf(ctx, 2 * (int64_t)n);
        BK_TRY(pf.add(ofh.data(), d_of, n * sizeof(double)));
        BK_TRY(pf.add(olh.data(), d_ol, n * sizeof(double)));
        BK_TRY(pf.finish());
        for (int q = 0; q < nm; ++q) {
          const MergeDesc& md = descs[q];
          for (int j = 0; j < md.K; ++j) { bf[md.s + j] = ofh[md.s + j]; bl[md.s + j] = olh[md.s + j]; }
          for (int t = 0; t < md.m - md.K; ++t) {
            bf[md.s + md.K + t] = yf[md.s + A.defsrc[md.s + t]];
            bl[md.s + md.K + t] = yl[md.s + A.defsrc[md.s + t]];
          }
        }
      }
    } else if (maxKneed > 0) {
      gd.reserve(2 * nm);
      int gm = 0, gn = 0;
      for (int q = 0; q < nm; ++q) {
        const MergeDesc& md = descs[q];
        if (md.Kneed <= 0) continue;
        const int k12 = md.k1 + md.k2, k23 = md.k2 + md.k3;
        const int n2 = md.m - md.n1;
        const int64_t o = md.s + (int64_t)md.s * N;
        // rows of the left child that no kept column touches must still be written
        GemmDesc g1{};
        g1.A = Qc + o; g1.B = U + o; g1.C = Qn + o;
        g1.lda = N; g1.ldb = N; g1.ldc = N;
        g1.m = md.n1; g1.n = md.Kneed; g1.k = k12; g1.kidx = d_srccol + md.s;
        gd.push_back(g1);
        GemmDesc g2{};
        g2.A = Qc + o + md.n1; g2.B = U + o + md.k1; g2.C = Qn + o + md.n1;
        g2.lda = N; g2.ldb = N; g2.ldc = N;
        g2.m = n2; g2.n = md.Kneed; g2.k = k23; g2.kidx = d_srccol + md.s + md.k1;
        gd.push_back(g2);
        gm = std::max(gm, std::max(md.n1, n2));
        gn = std::max(gn, md.Kneed);
      }
#ifdef BK_FAULT_INJECT
      // BIGKRLS_FAULT=dc_lag (test build): the stream runs 2 ms behind the host, as it does with many processes on one
      // GPU -- the copy below executes long after hipMemcpyAsync has returned. =dc_gd_clobber additionally overwrites
      // the descriptors right after the launch (what freeing them early amounted to): tools/dc_async_source_probe.py
      // uses it to see WHEN the runtime reads a pageable source (result: before hipMemcpyAsync returns).
      const char* dc_fault = getenv("BIGKRLS_FAULT");
      const bool dc_lag = dc_fault && (std::string(dc_fault) == "dc_lag" || std::string(dc_fault) == "dc_gd_clobber");
      if (dc_lag) hipLaunchKernelGGL(dc_fault_spin, dim3(1), dim3(64), 0, st, 200000LL);
#endif
      BK_TRY(stage.put(d_gdescs, gd.data(), gd.size() * sizeof(GemmDesc)));
      BK_TRY(gemm_batched_nn(ctx, d_gdescs, (int)gd.size(), gm, gn));
#ifdef BK_FAULT_INJECT
      if (dc_fault && std::string(dc_fault) == "dc_gd_clobber") std::memset((void*)gd.data(), 0, gd.size() * sizeof(GemmDesc));
      if (dc_lag) {   // heap churn: blocks of the descriptors' size, zero-filled and freed (lands on anything freed too early)
        for (int rep = 0; rep < 4; ++rep) {
          std::vector<char> junk(gd.size() * sizeof(GemmDesc) + 16, 0);
          asm volatile("" ::"r"(junk.data()) : "memory");
        }
      }
#endif
    }
    if (max_ndef > 0 && !lazy_level) {
      for (int b0 = 0; b0 < nm; b0 += 65535) {
        const int nb = std::min(65535, nm - b0);
        hipLaunchKernelGGL(dc_copy_deflated, dim3(max_ndef, nb), dim3(64), 0, st,
                           (const MergeDesc*)(d_descs + b0), (const int*)d_defsrc,
                           (const int*)d_defdst, (const double*)Qc, Qn, N);
      }
      BK_CHECK_LAUNCH();
    }
    // host vectors (gd, descs, A.*) are sources of asynchronous copies: alive and unmodified up to here
    BK_HIP(hipStreamSynchronize(st));
    if (trace_fine() && !lazy_level) {
      BK_TRY(trace_point(ctx, st, "R:dc_Qn", Qn, N * N, depth));
    }
    if (!lazy_level) std::swap(Qc, Qn);
    lap(4);
    if (verbose)
      fprintf(stderr, "[bigkrls]   d&c depth %2d%s: %6d merges, largest %6d (non-deflated %6d, %d rotations) %8.2f ms"
                      "  [wait z %.2f | host scans %.2f | upload+launch %.2f | wait roots %.2f | vectors %.2f]\n",
              depth, lazy_level ? " (factored)" : "", nm, max_m, maxK, (int)A.ra.size(),
              std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_level).count(),
              ms_split[0], ms_split[1], ms_split[2], ms_split[3], ms_split[4]);
  }
  if (n == 1) {
    vals_desc.assign(1, dadj[0]);
    src_cols.assign(1, 0);
    nv_final = 1;
  }
  if (lazy) {
    // ---- apply the factored levels to the kept columns, root first ---------------------------
    const auto t_apply = std::chrono::steady_clock::now();
    const int nv = (int)nv_final;
    const LazyLevel& R = lazy_levels[0];
    const int K0 = R.descs[0].K, kc = R.descs[0].Kneed;
    // the root's secular vectors occupy the first kc columns of U; the rest of U holds X, its
    // ping-pong partner and the product buffer (nv <= N/8)
    double* Xa = U + N * (int64_t)kc;
    double* Xb = Xa + N * (int64_t)nv;
    double* Tm = Xb + N * (int64_t)nv;
    stage.reset();      // (the root level ended with a synchronisation)
    BK_HIP(hipMemsetAsync(Xa, 0, (size_t)N * nv * sizeof(double), st));
    BK_TRY(stage.put(d_cpos, src_cols.data(), nv * sizeof(int)));
    BK_TRY(stage.put(d_srccol, R.srccol.data(), n * sizeof(int)));
    BK_TRY(stage.put(d_defsrc, R.defsrc.data(), n * sizeof(int)));
    hipLaunchKernelGGL(dc_iota, dim3((n + 255) / 256), dim3(256), 0, st, d_iota, n);
    hipLaunchKernelGGL(dc_lazy_root_x, dim3(nv), dim3(256), 0, st, nv, (const int*)d_cpos, K0,
                       (const int*)d_srccol, (const int*)d_defsrc, (const double*)U, N, Xa, N);
    BK_CHECK_LAUNCH();
    auto apply_rotations = [&](const LazyLevel& L, double* X) -> int {
      if (L.ra.empty() || L.rot_merges.empty()) return BIGKRLS_OK;
      const int nrt = (int)L.ra.size(), nrm = (int)L.rot_merges.size();
      BK_TRY(stage.put(d_ra, L.ra.data(), nrt * sizeof(int)));
      BK_TRY(stage.put(d_rb, L.rb.data(), nrt * sizeof(int)));
      BK_TRY(stage.put(d_rc, L.rc.data(), nrt * sizeof(double)));
      BK_TRY(stage.put(d_rs, L.rs.data(), nrt * sizeof(double)));
      BK_TRY(stage.put(d_rotids, L.rot_merges.data(), nrm * sizeof(int)));
      hipLaunchKernelGGL(dc_lazy_rotate, dim3((nv + 63) / 64, nrm), dim3(64), 0, st,
                         (const MergeDesc*)d_descs, (const int*)d_rotids, (const int*)d_ra,
                         (const int*)d_rb, (const double*)d_rc, (const double*)d_rs, X, N, nv);
      BK_CHECK_LAUNCH();
      return BIGKRLS_OK;
    };
    BK_TRY(stage.put(d_descs, R.descs.data(), sizeof(MergeDesc)));
    BK_TRY(apply_rotations(R, Xa));
    double* cur = Xa;
    double* oth = Xb;
    std::vector<GemmDesc> gd;
    for (int d = 1; d <= Dl; ++d) {
      const LazyLevel& L = lazy_levels[d];
      const int nm = (int)L.descs.size();
      BK_TRY(stage.put(d_descs, L.descs.data(), nm * sizeof(MergeDesc)));
      BK_TRY(stage.put(d_srccol, L.srccol.data(), n * sizeof(int)));
      BK_TRY(stage.put(d_defsrc, L.defsrc.data(), n * sizeof(int)));
      gd.clear();
      for (int q = 0; q < nm; ++q) {
        const MergeDesc& md = L.descs[q];
        if (md.K <= 0) continue;
        GemmDesc g{};
        g.A = stash + L.stash_off[q]; g.B = cur + md.s; g.C = Tm + md.s;
        g.lda = md.K; g.ldb = N; g.ldc = N;
        g.m = md.K; g.n = nv; g.k = md.K; g.kidx = d_iota;
        gd.push_back(g);
      }
      if (!gd.empty()) {
        BK_TRY(stage.put(d_gdescs, gd.data(), gd.size() * sizeof(GemmDesc)));
        BK_TRY(gemm_batched_nn(ctx, d_gdescs, (int)gd.size(), L.maxK, nv));
      }
      hipLaunchKernelGGL(dc_lazy_scatter, dim3((L.max_m + 255) / 256, nm, nv), dim3(256), 0, st,
                         (const MergeDesc*)d_descs, (const int*)d_srccol, (const int*)d_defsrc,
                         (const double*)Tm, (const double*)cur, oth, N);
      BK_CHECK_LAUNCH();
      BK_TRY(apply_rotations(L, oth));    // (every upload has its own slice of the pinned arena: no synchronisation here)
      std::swap(cur, oth);
    }
    // the explicit eigenvector matrices of depth Dl+1, one block per depth-Dl merge: that level's
    // deflation rotations were applied to them in place and couple the two children's columns
    {
      const LazyLevel& L = lazy_levels[Dl];
      gd.clear();
      int gm = 0;
      for (const MergeDesc& md : L.descs) {
        GemmDesc g{};
        g.A = Qc + md.s + (int64_t)md.s * N; g.B = cur + md.s; g.C = oth + md.s;
        g.lda = N; g.ldb = N; g.ldc = N;
        g.m = md.m; g.n = nv; g.k = md.m; g.kidx = d_iota;
        gd.push_back(g);
        gm = std::max(gm, md.m);
      }
      BK_TRY(stage.put(d_gdescs, gd.data(), gd.size() * sizeof(GemmDesc)));
      BK_TRY(gemm_batched_nn(ctx, d_gdescs, (int)gd.size(), gm, nv));
      BK_HIP(hipStreamSynchronize(st));
    }
    for (int t = 0; t < nv; ++t) src_cols[t] = t;
    *Qfinal = oth;
    if (verbose)
      fprintf(stderr, "[bigkrls]   d&c: factored levels applied to %d columns %8.2f ms\n", nv,
              std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_apply).count());
    return BIGKRLS_OK;
  }
  (void)nv_final;
  *Qfinal = Qc;
  return BIGKRLS_OK;
}

// =============================================================================
// phase 3: back-transform
// =============================================================================
// Vp (ne x pw): reflectors of panel [j0, j0+pw) restricted to rows j0+1..n-1
__global__ void bt_extract_panel(const double* __restrict__ W, int n, int j0, int pw,
                                 double* __restrict__ Vp, int ne) {
  const int64_t total = (int64_t)ne * pw;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (int64_t)gridDim.x * blockDim.x) {
    const int r = (int)(e % ne), k = (int)(e / ne);
    const int row = j0 + 1 + r, c = j0 + k;
    double v = 0.0;
    if (row == c + 1) v = 1.0;
    else if (row > c + 1) v = W[row + (int64_t)c * n];
    Vp[e] = v;
  }
}

// T (pw x pw upper triangular) from G = Vp'Vp and tau (LAPACK dlarft, forward/columnwise)
__global__ __launch_bounds__(256) void bt_build_t(const double* __restrict__ G, int pw,
                                                  const double* __restrict__ tau,
                                                  double* __restrict__ T, const int* __restrict__ run_if) {
  if (run_if != nullptr && *run_if == 0) return;    // (a stage-1 panel whose T factor pq_chol has already written)
  // Blocked recurrence: the four 16 x 16 diagonal blocks of T are built at the same time (one wave
  // each, 16 dependent column steps: T[0:i, i] = -tau_i T[0:i,0:i] G[0:i, i] inside the block),
  // then block column j = 1, 2, 3 follows from T(0:16j, j) = -T(0:16j, 0:16j) G(0:16j, j) T_jj
  // (two small products each): 16 + 6 barriers instead of 64 steps with 16-deep dependent sums.
  constexpr int NB = TRD_NB, LT = TRD_NB + 1, BB = 16;
  __shared__ double sT[NB * LT];   // sT[r][k]
  __shared__ double sG[NB * LT];   // sG[k][i] = v_k' v_i
  __shared__ double sX[(NB - BB) * (BB + 1)];
  __shared__ double stau[NB];
  const int t = threadIdx.x;
  for (int e = t; e < NB * LT; e += 256) { sT[e] = 0.0; sG[e] = 0.0; }
  if (t < NB) stau[t] = (t < pw) ? tau[t] : 0.0;
  __syncthreads();
  for (int e = t; e < pw * pw; e += 256) sG[(e % pw) * LT + (e / pw)] = G[e];   // one batch of global loads
  __syncthreads();
  {
    const int w = t >> 6, lane = t & 63, rl = lane >> 2, part = lane & 3;
    const int c0 = BB * w, r = c0 + rl;
    for (int ii = 0; ii < BB; ++ii) {
      const int i = c0 + ii;
      const double ti = stau[i];
      double acc = 0.0;
      if (rl < ii)
        for (int k = r + part; k < i; k += 4) acc += sT[r * LT + k] * sG[k * LT + i];
      acc += __shfl_xor(acc, 1, 64);
      acc += __shfl_xor(acc, 2, 64);
      if (part == 0) {
        if (rl < ii) sT[r * LT + i] = -ti * acc;
        else if (rl == ii) sT[i * LT + i] = ti;
      }
      __syncthreads();
    }
  }
  for (int j = 1; j < NB / BB; ++j) {
    const int R = BB * j, cj = BB * j;
    for (int e = t; e < R * BB; e += 256) {       // X = G(0:R, j) T_jj
      const int r = e >> 4, c = e & 15;
      double acc = 0.0;
#pragma unroll
      for (int k = 0; k < BB; ++k) acc += sG[r * LT + cj + k] * sT[(cj + k) * LT + cj + c];
      sX[r * (BB + 1) + c] = acc;
    }
    __syncthreads();
    for (int e = t; e < R * BB; e += 256) {       // T(0:R, j) = -T(0:R, 0:R) X
      const int r = e >> 4, c = e & 15;
      double acc = 0.0;
      for (int k = r; k < R; ++k) acc += sT[r * LT + k] * sX[k * (BB + 1) + c];
      sT[r * LT + cj + c] = -acc;
    }
    __syncthreads();
  }
  for (int e = t; e < pw * pw; e += 256) T[e] = sT[(e % pw) * LT + (e / pw)];
}

int back_transform(bigkrls_ctx* ctx, const double* W, int n, const double* tau, double* Z,
                   int64_t ldz, int nv) {
  if (n < 3 || nv <= 0) return BIGKRLS_OK;
  void* p = nullptr;
  const int64_t need = (int64_t)n * TRD_NB + 3 * (int64_t)TRD_NB * TRD_NB + 2 * (int64_t)TRD_NB * nv;
  BK_TRY(ws_get(ctx, SLOT_EIG_BT, need * sizeof(double), &p));
  double* Vp = (double*)p;
  double* G = Vp + (int64_t)n * TRD_NB;
  double* T = G + TRD_NB * TRD_NB;
  double* W1 = T + 2 * TRD_NB * TRD_NB;
  double* W2 = W1 + (int64_t)TRD_NB * nv;
  hipStream_t st = ctx->stream;
  const int ncol = n - 2;  // reflectors c = 0..n-3 are non-trivial (c = n-2 has tau = 0)
  const int npanel = (ncol + TRD_NB - 1) / TRD_NB;
  for (int pb = npanel - 1; pb >= 0; --pb) {
    const int j0 = pb * TRD_NB;
    const int pw = std::min(TRD_NB, ncol - j0);
    const int ne = n - j0 - 1;
    int blocks = (int)std::min<int64_t>(((int64_t)ne * pw + 255) / 256, 4096);
    hipLaunchKernelGGL(bt_extract_panel, dim3(blocks), dim3(256), 0, st, W, n, j0, pw, Vp, ne);
    BK_CHECK_LAUNCH();
    BK_TRY(gemm(ctx, 1, 0, pw, pw, ne, 1.0, Vp, ne, Vp, ne, 0.0, G, pw));
    hipLaunchKernelGGL(bt_build_t, dim3(1), dim3(256), 0, st, (const double*)G, pw, tau + j0, T, (const int*)nullptr);
    BK_CHECK_LAUNCH();
    double* Zs = Z + (j0 + 1);
    BK_TRY(gemm(ctx, 1, 0, pw, nv, ne, 1.0, Vp, ne, Zs, ldz, 0.0, W1, pw));
    BK_TRY(gemm(ctx, 0, 0, pw, nv, pw, 1.0, T, pw, W1, pw, 0.0, W2, pw));
    BK_TRY(gemm(ctx, 0, 0, ne, nv, pw, -1.0, Vp, ne, W2, pw, 1.0, Zs, ldz));
  }
  return BIGKRLS_OK;
}

#include "eigen_2stage.inc"

}  // namespace

// ---------------------------------------------------------------------------
// Top-k eigenpairs for Neig << N (the reference's `eigs_sym` branch, src/eigen.cpp:18-22):
// block Lanczos with full re-orthogonalisation. All O(N^2) work is GEMM (K times a 128-column
// block per step); the projected block-tridiagonal matrix is solved by the dense solver above.
//   B_0 = orth(random N x b);  per step j:  W = K B_j;  C = B_all' W;  W -= B_all C  (twice: CGS2)
//   A_j = sym(C_j) (diagonal block of T);  W = B_{j+1} beta_{j+1}  (Cholesky QR, twice)
// Ritz pairs (theta_i, y_i) of T; residual of pair i = |beta_m y_i[last block]| (checked every
// few steps, stop when all k are below tol * theta_1). Finally Q = B Y is refined by one
// Rayleigh-Ritz step against K itself (k x k), which also returns the true residual level.
// ---------------------------------------------------------------------------
namespace {

__global__ void kry_fill_random(double* __restrict__ p, int64_t total, unsigned seed) {
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (int64_t)gridDim.x * blockDim.x) {
    unsigned long long x = (unsigned long long)e * 0x9E3779B97F4A7C15ull + seed;
    x ^= x >> 31; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 29; x *= 0x94D049BB133111EBull; x ^= x >> 32;
    p[e] = (double)(x >> 11) * (1.0 / 9007199254740992.0) - 0.5;
  }
}

// out[c] = || KQ[:, c] - theta[c] Q[:, c] ||^2, one workgroup per column (the true residual of a Ritz pair)
__global__ __launch_bounds__(256) void kry_resid_sq_kernel(const double* __restrict__ KQ, const double* __restrict__ Q,
                                                            const double* __restrict__ theta, int64_t n, int64_t row0,
                                                            int64_t nrows, double* __restrict__ out) {
  __shared__ double part[4];
  const int c = blockIdx.x, tid = threadIdx.x;
  const double th = theta[c];
  const double* kq = KQ + (int64_t)c * n + row0;      // (multi-GPU: the sum over this rank's rows)
  const double* q = Q + (int64_t)c * n + row0;
  double acc = 0.0;
  for (int64_t i = tid; i < nrows; i += 256) {
    const double r = kq[i] - th * q[i];
    acc += r * r;
  }
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
  if ((tid & 63) == 0) part[tid >> 6] = acc;
  __syncthreads();
  if (tid == 0) out[c] = (part[0] + part[1]) + (part[2] + part[3]);
}

// One workgroup: upper Cholesky factor R (G = R'R) and its inverse of a b x b SPD matrix (b <= 128),
// in LDS and in place. G (column-major, ld b) is symmetrised first; R goes to Rout (zero below the
// diagonal), inv(R) overwrites G. *flag is set when a pivot is not positive (breakdown).
constexpr int KRY_B = 128;
__global__ __launch_bounds__(256) void kry_chol_inv_kernel(double* __restrict__ G, int b, double* __restrict__ Rout,
                                                            int* __restrict__ flag, double* __restrict__ stats) {
  // Register tiles: thread (ti, tj) of a 16 x 16 grid owns the entries (ti + 16 a, tj + 16 c), a, c = 0..7 (cyclic, so
  // that the shrinking active part stays spread over all threads). Both phases are 128 rank-1 steps with ONE barrier
  // each: the owners of row k publish it through a double-buffered LDS vector, everybody updates its 8 x 8 tile.
  //   Cholesky (right-looking):  R[k][:] = A[k][:] / sqrt(A[k][k]);  A[i][j] -= R[k][i] R[k][j]
  //   inverse (Gauss-Jordan on [R | I], k descending):  X[k][:] /= R[k][k];  X[i][:] -= R[i][k] X[k][:]  (i < k)
  // Rows and columns >= b are padded with the identity. (The previous version -- one LDS element at a time with an
  // integer division per element, then one thread per column of the inverse -- took 0.6 ms per call, twice per
  // Lanczos step on the critical path.)
  __shared__ double sR[KRY_B][KRY_B + 1];
  __shared__ double rowk[2][KRY_B];
  const int tid = threadIdx.x, ti = tid & 15, tj = tid >> 4;
  double t[8][8];
#pragma unroll
  for (int a = 0; a < 8; ++a)
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const int i = ti + 16 * a, j = tj + 16 * c;
      t[a][c] = (i < b && j < b) ? 0.5 * (G[i + (int64_t)j * b] + G[j + (int64_t)i * b]) : (i == j ? 1.0 : 0.0);
    }
  for (int e = tid; e < KRY_B * (KRY_B + 1); e += 256) (&sR[0][0])[e] = 0.0;
  double dmin = 1e300, dmax = 0.0;
  if (tid == 0) {                 // trace(G) = |W|_F^2: bounds every residual |W y| when the factorisation breaks down
    double tr = 0.0;
    for (int i = 0; i < b; ++i) tr += G[i + (int64_t)i * b];
    stats[2] = tr;
  }
  __syncthreads();
#pragma unroll
  for (int ka = 0; ka < 8; ++ka) {
#pragma unroll 1
    for (int kt = 0; kt < 16; ++kt) {
      const int k = 16 * ka + kt;
      double* rk = rowk[k & 1];
      if (ti == kt) {
#pragma unroll
        for (int c = 0; c < 8; ++c) rk[tj + 16 * c] = t[ka][c];
      }
      __syncthreads();
      const double d = rk[k];
      if (k < b) {                        // (the pivots: squared norms of the columns orthogonalised so far; uniform)
        dmin = d < dmin ? d : dmin;
        dmax = d > dmax ? d : dmax;
      }
      if (!(d > 0.0) || !isfinite(d)) {   // uniform
        if (tid == 0) {
          *flag = 1;
          stats[0] = dmin;
          stats[1] = dmax;
        }
        return;
      }
      const double rinv = 1.0 / sqrt(d);
      double ri[8], rj[8];
#pragma unroll
      for (int a = 0; a < 8; ++a) ri[a] = rk[ti + 16 * a] * rinv;
#pragma unroll
      for (int c = 0; c < 8; ++c) rj[c] = rk[tj + 16 * c] * rinv;
      if (tid < KRY_B && tid >= k) sR[k][tid] = rk[tid] * rinv;
      // (entries in rows / columns <= k are dead: the tiles a < ka or c < ka, 60 % of the updates over the sweep, are
      //  left alone -- ka is a compile-time constant of the unrolled outer loop)
#pragma unroll
      for (int a = ka; a < 8; ++a)
#pragma unroll
        for (int c = ka; c < 8; ++c) t[a][c] = fma(-ri[a], rj[c], t[a][c]);
    }
  }
  if (tid == 0) {
    stats[0] = dmin;
    stats[1] = dmax;
  }
  __syncthreads();
  // ---- X = inv(R) ------------------------------------------------------------------------------------
#pragma unroll
  for (int a = 0; a < 8; ++a)
#pragma unroll
    for (int c = 0; c < 8; ++c) t[a][c] = (ti + 16 * a == tj + 16 * c) ? 1.0 : 0.0;
#pragma unroll
  for (int ka = 7; ka >= 0; --ka) {
#pragma unroll 1
    for (int kt = 15; kt >= 0; --kt) {
      const int k = 16 * ka + kt;
      double* rk = rowk[k & 1];
      if (ti == kt) {
        const double dinv = 1.0 / sR[k][k];
#pragma unroll
        for (int c = 0; c < 8; ++c) {
          t[ka][c] *= dinv;
          rk[tj + 16 * c] = t[ka][c];
        }
      }
      __syncthreads();
      double f[8], xk[8];
#pragma unroll
      for (int a = 0; a < 8; ++a) f[a] = (ti + 16 * a < k) ? sR[ti + 16 * a][k] : 0.0;
#pragma unroll
      for (int c = 0; c < 8; ++c) xk[c] = rk[tj + 16 * c];
      // (f = 0 in the rows >= k, x_k = 0 in the columns < k: the tiles a > ka or c < ka would add -0 * x)
#pragma unroll
      for (int a = 0; a <= ka; ++a)
#pragma unroll
        for (int c = ka; c < 8; ++c) t[a][c] = fma(-f[a], xk[c], t[a][c]);
    }
  }
#pragma unroll
  for (int a = 0; a < 8; ++a)
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const int i = ti + 16 * a, j = tj + 16 * c;
      if (i < b && j < b) {
        G[i + (int64_t)j * b] = (i <= j) ? t[a][c] : 0.0;
        Rout[i + (int64_t)j * b] = (i <= j) ? sR[i][j] : 0.0;
      }
    }
}

// The projected block-tridiagonal matrix T (m x m, m = steps * b) from the blocks kept on the device:
// diagonal blocks sym(A_j), sub-diagonal blocks beta_{j+1} (upper triangular) and their transposes.
__global__ void kry_assemble_t(const double* __restrict__ Aall, const double* __restrict__ Ball, int steps, int b,
                               double* __restrict__ T) {
  const int64_t m = (int64_t)steps * b, total = m * m;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = e % m, c = e / m;
    const int jr = (int)(r / b), jc = (int)(c / b), lr = (int)(r % b), lc = (int)(c % b);
    double v = 0.0;
    if (jr == jc) {
      const double* Aj = Aall + (int64_t)jr * b * b;
      v = 0.5 * (Aj[lr + (int64_t)lc * b] + Aj[lc + (int64_t)lr * b]);
    } else if (jr == jc + 1) {          // T[j+1, j] = beta_{j+1}
      if (lr <= lc) v = Ball[(int64_t)jc * b * b + lr + (int64_t)lc * b];
    } else if (jc == jr + 1) {          // T[j, j+1] = beta_{j+1}'
      if (lc <= lr) v = Ball[(int64_t)jr * b * b + lc + (int64_t)lr * b];
    }
    T[e] = v;
  }
}

// The projected matrix of the COMPRESSED basis [B Y (the k Ritz vectors of the check at s1 steps) | B_s1 ... B_{s2-1}]:
//   [ diag(theta)   C'                         ]     C = beta_{s1} Y[last block rows, :]  (b x k)
//   [ C             A_s1        beta_{s1+1}'   ]
//   [               beta_{s1+1} A_{s1+1}   ... ]     dimension k + (s2 - s1) b
__global__ void kry_assemble_compressed(const double* __restrict__ theta, const double* __restrict__ Cm, int k,
                                        const double* __restrict__ Aall, const double* __restrict__ Ball, int s1, int s2,
                                        int b, double* __restrict__ T) {
  const int64_t m = (int64_t)k + (int64_t)(s2 - s1) * b, total = m * m;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = e % m, c = e / m;
    double v = 0.0;
    if (r < k && c < k) {
      v = (r == c) ? theta[r] : 0.0;
    } else if (r >= k && c < k) {
      if (r - k < b) v = Cm[(r - k) + c * b];
    } else if (r < k) {
      if (c - k < b) v = Cm[(c - k) + r * b];
    } else {
      const int jr = s1 + (int)((r - k) / b), jc = s1 + (int)((c - k) / b), lr = (int)((r - k) % b), lc = (int)((c - k) % b);
      if (jr == jc) {
        const double* Aj = Aall + (int64_t)jr * b * b;
        v = 0.5 * (Aj[lr + (int64_t)lc * b] + Aj[lc + (int64_t)lr * b]);
      } else if (jr == jc + 1) {
        if (lr <= lc) v = Ball[(int64_t)jc * b * b + lr + (int64_t)lc * b];
      } else if (jc == jr + 1) {
        if (lc <= lr) v = Ball[(int64_t)jr * b * b + lc + (int64_t)lr * b];
      }
    }
    T[e] = v;
  }
}

// W (n x b, ld n) <- orthonormal basis of its columns by Cholesky QR, twice; Rout (host, b x b upper)
// gets the triangular factor with W_in = W_out Rout. tmp is an n x b scratch, dG holds 4 b^2 doubles
// and an int. Everything runs on the device (Gram matrix, factorisation and inverse in one workgroup,
// W R^-1 as a GEMM); one synchronisation at the end brings Rout and the breakdown flag to the host.
// Multi-GPU (comm != nullptr): the rows [ro, ro + nr) of W are this rank's; the Gram matrix is the all-reduced sum of
// the ranks' W_loc' W_loc (128 x 128 doubles per pass), its factorisation is replicated (identical input, deterministic
// kernel: identical factors on every rank) and W_loc R^-1 is local -- CholeskyQR2 in its communication-avoiding form.
int kry_cholqr(bigkrls_ctx* ctx, double** W, double** tmp, int64_t n, int b, double* dG,
               std::vector<double>& Rout, bool* breakdown, double* dRkeep = nullptr, bigkrls_comm* comm = nullptr,
               int64_t ro = 0, int64_t nr = -1, double* pivots = nullptr) {
  hipStream_t st = ctx->stream;
  BK_REQUIRE(b <= KRY_B, "kry_cholqr: block too wide");
  if (nr < 0) nr = n;
  double* dR1 = dG + (int64_t)b * b;
  double* dR2 = dR1 + (int64_t)b * b;
  double* dRacc = dR2 + (int64_t)b * b;
  int* dflag = (int*)(dRacc + (int64_t)b * b);
  *breakdown = false;
  BK_HIP(hipMemsetAsync(dflag, 0, 7 * sizeof(double), st));
  for (int pass = 0; pass < 2; ++pass) {
    if (nr > 0) BK_TRY(gemm(ctx, 1, 0, b, b, nr, 1.0, *W + ro, n, *W + ro, n, 0.0, dG, b));
    else BK_HIP(hipMemsetAsync(dG, 0, (size_t)b * b * sizeof(double), st));
    if (comm) BK_TRY(comm_all_reduce(comm, dG, (int64_t)b * b, COMM_SUM));
    // (behind the flag: [flag | smallest pivot, largest pivot, trace of the Gram matrix: pass 0 | the same: pass 1])
    hipLaunchKernelGGL(kry_chol_inv_kernel, dim3(1), dim3(256), 0, st, dG, b, pass == 0 ? dR1 : dR2, dflag,
                       (double*)dflag + 1 + 3 * pass);
    BK_CHECK_LAUNCH();
    if (nr > 0) BK_TRY(gemm(ctx, 0, 0, nr, b, b, 1.0, *W + ro, n, dG, b, 0.0, *tmp + ro, n));
    std::swap(*W, *tmp);
  }
  BK_TRY(gemm(ctx, 0, 0, b, b, b, 1.0, dR2, b, dR1, b, 0.0, dRacc, b));   // Rout = R2 R1
  if (dRkeep) BK_HIP(hipMemcpyAsync(dRkeep, dRacc, (size_t)b * b * sizeof(double), hipMemcpyDeviceToDevice, st));
  // R2 R1 and the flag (they sit next to each other on the device) come down in one copy through the
  // context's pinned buffer
  double* hp = nullptr;
  BK_TRY(pinned_get(ctx, (int64_t)b * b + 8, &hp));
  BK_HIP(hipMemcpyAsync(hp, dRacc, ((size_t)b * b + 7) * sizeof(double), hipMemcpyDeviceToHost, st));
  BK_HIP(hipStreamSynchronize(st));
  Rout.assign(hp, hp + (size_t)b * b);
  int h_flag = 0;
  std::memcpy(&h_flag, hp + (size_t)b * b, sizeof(int));
  if (pivots) std::memcpy(pivots, hp + (size_t)b * b + 1, 6 * sizeof(double));
  if (h_flag != 0) *breakdown = true;   // (W then holds garbage; the caller stops with the blocks it has)
  return BIGKRLS_OK;
}

}  // namespace

int fill_random(bigkrls_ctx* ctx, double* p, int64_t count, uint32_t seed) {
  BK_REQUIRE(p && count >= 0, "fill_random: bad arguments");
  if (count == 0) return BIGKRLS_OK;
  hipLaunchKernelGGL(kry_fill_random, dim3(2048), dim3(256), 0, ctx->stream, p, count, (unsigned)seed);
  BK_CHECK_LAUNCH();
  return BIGKRLS_OK;
}

int lanczos_projected(bigkrls_ctx* ctx, const double* d_A_blocks, const double* d_beta_blocks, int steps, int b,
                      double* d_T) {
  BK_REQUIRE(d_A_blocks && d_T && steps >= 1 && b >= 1 && (steps == 1 || d_beta_blocks), "lanczos_projected: bad arguments");
  const int64_t m = (int64_t)steps * b;
  hipLaunchKernelGGL(kry_assemble_t, dim3((unsigned)std::min<int64_t>((m * m + 255) / 256, 8192)), dim3(256), 0,
                     ctx->stream, d_A_blocks, d_beta_blocks, steps, b, d_T);
  BK_CHECK_LAUNCH();
  return BIGKRLS_OK;
}

int cholqr2_block(bigkrls_ctx* ctx, double* W, double* tmp, int64_t n, int b, double* h_R, int* h_breakdown,
                  double* d_R) {
  BK_REQUIRE(W && tmp && h_R && h_breakdown && n > 0 && b > 0 && b <= KRY_B, "cholqr2_block: bad arguments");
  void* pg = nullptr;
  BK_TRY(ws_get(ctx, SLOT_KRY_C, (5 * (int64_t)b * b + 8) * sizeof(double), &pg));
  std::vector<double> R;
  bool breakdown = false;
  double *w = W, *t = tmp;
  BK_TRY(kry_cholqr(ctx, &w, &t, n, b, (double*)pg, R, &breakdown, d_R));   // two swaps: the result is back in W
  std::memcpy(h_R, R.data(), (size_t)b * b * sizeof(double));
  *h_breakdown = breakdown ? 1 : 0;
  return BIGKRLS_OK;
}


int eigen(bigkrls_ctx* ctx, const double* A, int64_t n64, int64_t lda, int64_t n_vals, double* vals,
          int64_t n_vecs_max, double keep_thresh, double* vecs, int64_t ldv, int64_t* h_n_vecs,
          int part_index, int part_count, int mode);

static thread_local std::string kry_diag;   // what the last block Lanczos did at its convergence checks (for the error text)

// W (n x cols, ld n) = K B (n x cols, ld n): the only way the block Lanczos touches K. Single GPU: the whole matrix
// A. Multi-GPU: this rank's column block Kcols = K[:, r0:r1] gives the rows r0:r1 of K B (K symmetric: Kcols' B),
// one all-gather of the row blocks assembles the product on every rank ("Eigen, partial").
struct KTimes {
  const double* A = nullptr;
  int64_t lda = 0;
  bigkrls_comm* comm = nullptr;
  const double* Kcols = nullptr;
  int64_t r0 = 0, r1 = 0, nb = 0;
};

// Multi-GPU: every n-row array of the iteration (the basis B, W, the Ritz vectors) is ROW-SHARDED in place -- rank r
// computes and reads only its rows [r0, r1) of it, the other rows are not valid. The one product that needs whole
// columns is this one: the block's rows are all-gathered (in place: n x cols doubles per step, the exchange
// north_star names), then W[r0:r1, :] = K[r0:r1, :] B from this rank's column block of K. Nothing of W is gathered:
// the re-orthogonalisation, the Cholesky QR and the Ritz vectors work on the row blocks and all-reduce their small
// Gram / coefficient matrices.
static int k_times(bigkrls_ctx* ctx, const KTimes& op, int64_t n, double* B, int64_t cols, double* W) {
  if (!op.comm) return gemm(ctx, 0, 0, n, cols, n, 1.0, op.A, op.lda, B, n, 0.0, W, n);
  const int64_t nloc = op.r1 - op.r0;
  BK_TRY(comm_gather_rows(op.comm, B + op.r0, nloc, n, cols, op.nb, n, B, n));
  if (nloc > 0) BK_TRY(gemm(ctx, 1, 0, nloc, cols, n, 1.0, op.Kcols, n, B, n, 0.0, W + op.r0, n));
  return BIGKRLS_OK;
}

// control decisions of the iteration (breakdown, convergence, verification) on values agreed by all ranks: a
// last-bit difference between replicas must never let one rank leave the loop while the others enter the next
// all-gather. Element-wise minimum over the ranks.
static int kry_agree_min(const KTimes& op, double* vals, int count) {
  if (!op.comm) return BIGKRLS_OK;
  return comm_all_reduce_host(op.comm, vals, count, COMM_MIN);
}

// a local status that every rank must share before the next collective (allocations: the one local failure that can
// realistically differ between ranks); single GPU: the status itself
static int kry_agree_status(const KTimes& op, int rc) { return op.comm ? comm_agree(op.comm, rc) : rc; }

static int eigen_krylov(bigkrls_ctx* ctx, const KTimes& kop, int64_t n, int64_t k, double* vals,
                        int64_t n_vecs_max, double keep_thresh, double* vecs, int64_t ldv,
                        int64_t* h_n_vecs, int part_index, int part_count) {
  hipStream_t st = ctx->stream;
  constexpr int b = 128;
  const double tol = 1e-10;
  kry_diag.clear();
  const int64_t maxdim = std::min<int64_t>(n / 2 / b * b, std::max<int64_t>(16 * k, 4096) / b * b);
  const int maxsteps = (int)(maxdim / b);
  void *pB = nullptr, *pW = nullptr, *pC = nullptr, *pC2 = nullptr;
  {
    int rc = ws_get(ctx, SLOT_KRY_B, n * maxdim * sizeof(double), &pB);
    if (rc == BIGKRLS_OK) rc = ws_get(ctx, SLOT_KRY_W, 2 * n * (int64_t)std::max<int64_t>(b, k) * sizeof(double), &pW);
    if (rc == BIGKRLS_OK)
      rc = ws_get(ctx, SLOT_KRY_C, (maxdim * b + 5 * b * b + 8 + k * k + 2 * k + 2 * (int64_t)maxsteps * b * b + k + b * k) *
                                       sizeof(double), &pC);
    if (rc == BIGKRLS_OK && kop.comm)   // the staging of the per-step all-gather, before the first one is entered
      rc = ws_get(ctx, SLOT_COMM_STAGE, (int64_t)(kop.comm->nranks + 1) * kop.nb * std::max<int64_t>(b, k) * sizeof(double), &pC2);
    BK_TRY(kry_agree_status(kop, rc));
  }
  double* B = (double*)pB;
  double* W = (double*)pW;
  double* W2 = W + n * std::max<int64_t>(b, k);
  // this rank's rows of every n-row array (see k_times); one GPU: all of them
  bigkrls_comm* const comm = kop.comm;
  const int64_t ro = comm ? kop.r0 : 0, nr = comm ? kop.r1 - kop.r0 : n;
  // C (rows x cols, ld rows) = sum over the ranks of  Xloc' Yloc  (X: n x rows, Y: n x cols, ld n)
  auto gram = [&](const double* X, int64_t rows, const double* Y, int64_t cols, double* Cout) -> int {
    if (nr > 0) BK_TRY(gemm(ctx, 1, 0, rows, cols, nr, 1.0, X + ro, n, Y + ro, n, 0.0, Cout, rows));
    else BK_HIP(hipMemsetAsync(Cout, 0, (size_t)(rows * cols) * sizeof(double), st));
    if (comm) BK_TRY(comm_all_reduce(comm, Cout, rows * cols, COMM_SUM));
    return BIGKRLS_OK;
  };
  double* C = (double*)pC;
  double* dG = C + maxdim * b;            // Cholesky-QR scratch: Gram / inverse, R1, R2, R2 R1, flag
  double* dA = dG + 4 * b * b + 8;
  // diagonal blocks A_j and sub-diagonal factors beta_{j+1} of the projected matrix stay on the device
  double* dAall = dA + b * b + k * k + 2 * k;
  double* dBall = dAall + (int64_t)maxsteps * b * b;
  double* dPrev = dBall + (int64_t)maxsteps * b * b;   // [theta (k) | C (b x k)] of the last full check (for the estimate)
  std::vector<double> Rtmp;                        // beta of the last step (host copy, for the Ritz residuals)
  bool breakdown = false;
  // ---- B_0 ----------------------------------------------------------------------------------
  BK_TRY(fill_random(ctx, W, n * b, 20240229u));
  BK_TRY(kry_cholqr(ctx, &W, &W2, n, b, dG, Rtmp, &breakdown, nullptr, comm, ro, nr));
  {
    double okv = breakdown ? 0.0 : 1.0;
    BK_TRY(kry_agree_min(kop, &okv, 1));
    breakdown = !(okv > 0.5);
  }
  BK_REQUIRE(!breakdown, "eigen (Krylov): start block is rank deficient");
  BK_HIP(hipMemcpyAsync(B, W, n * b * sizeof(double), hipMemcpyDeviceToDevice, st));
  int steps = 0;
  int64_t dim = b;
  std::vector<double> theta;       // Ritz values of the last full check (descending)
  double wnorm_sq = 0.0;           // |W|_F^2 of the block the last step produced (before its QR)
  double wmax_seen = 0.0;          // the largest direction any new block had (square root of the largest Gram pivot)
  bool early_check_done = false;   // the check ahead of the schedule on a (numerically) invariant Krylov space, see below
  double dropped = 0.0;            // |W|_F of the blocks replaced by random ones: what every residual estimate misses
  int full_steps = 0;              // steps of the last full check (0: none yet)
  bool next_is_estimate = false;   // the next check decomposes the compressed projected problem (see below)
  void* pY = nullptr;
  bool converged = false;
  // each check is a dense eigensolve of T (latency-bound, ~12 us per row of T); the first one comes at a
  // subspace of 5k columns, or of 2.5k where a step costs more than such a check (sizes only: the schedule, and
  // with it the result, must not depend on timing)
  const double step_s_est = 2.0 * (double)n * (double)n * b / 50e12;
  const bool check_is_cheap = 12e-6 * 4.0 * (double)k < step_s_est;
  // (C4: the pairs converge at step 27, C5 at step 23; a first check at 4k / 2k columns -- step 16 in both -- sits on the
  //  plateau of the worst residual and only costs a dense eigensolve of T: 5k / 2.5k columns, step 20, leaves three
  //  checks at C4 (20, 25, 27) and two at C5 (20, 23) instead of four and three)
  int next_check = (int)std::max<int64_t>(2, ((check_is_cheap ? 5 : 10) * k / 2 + b - 1) / b);
  if (const char* fc = getenv("BIGKRLS_KRY_FIRST_CHECK")) next_check = std::max(2, atoi(fc));   // (development)
  while (true) {
    // ---- one block Lanczos step: W = K B_j, orthogonalised against every block so far (CGS2) ---
    double* Bj = B + (int64_t)steps * b * n;
    // (bench.py: HIP-event sampling of the step's dominant product, 2 n^2 b flops, every launch)
    if (ctx->profile) BK_TRY(prof_begin(ctx, "lanczos_kb", 2.0 * (double)n * (double)n * b));
    BK_TRY(k_times(ctx, kop, n, Bj, b, W));
    if (ctx->profile) BK_TRY(prof_end(ctx, "lanczos_kb"));
    // Two Gram-Schmidt passes where the projection cancels (the blocks of the three-term recurrence, B_{j-1} and B_j:
    // |K B_j| shrinks to |beta_{j+1}|), one where it does not: with every earlier block kept orthogonal to rounding,
    // K B_j has components of only eps |K| along B_0 ... B_{j-2}, which a single pass removes to rounding. So the
    // first pass runs against the last two blocks only (256 of the dim columns), the second against all of them
    // -- 4 n (dim + 256) b flops per step instead of 8 n dim b. BIGKRLS_KRY_CGS=2 (development): both passes against
    // every block, as until round 6.
    static const bool cgs_full = [] { const char* e = getenv("BIGKRLS_KRY_CGS"); return e && e[0] == '2'; }();
    const int64_t loc0 = cgs_full ? 0 : (int64_t)std::max(0, steps - 1) * b;     // first column of the first pass
    if (ctx->profile) BK_TRY(prof_begin(ctx, "lanczos_cgs2", 4.0 * (double)n * (double)(2 * dim - loc0) * b));
    for (int pass = 0; pass < 2; ++pass) {
      const int64_t c0 = pass == 0 ? loc0 : 0, dimp = dim - c0;
      BK_TRY(gram(B + c0 * n, dimp, W, b, C));             // (multi-GPU: local rows + one all-reduce of dimp x 128)
      if (pass == 0)   // A_j = B_j' K B_j: the last b rows of the first coefficient block
        BK_TRY(copy_matrix(ctx, C + ((int64_t)steps * b - c0), b, b, dimp, dAall + (int64_t)steps * b * b, b));
      if (nr > 0) BK_TRY(gemm(ctx, 0, 0, nr, b, dimp, -1.0, B + c0 * n + ro, n, C, dimp, 1.0, W + ro, n));
    }
    if (ctx->profile) BK_TRY(prof_end(ctx, "lanczos_cgs2"));
    // piv: smallest / largest pivot and trace of the Gram matrix W'W, first and second pass of the Cholesky QR (the
    // same on every rank: the Gram matrices are all-reduced)
    double piv[6] = {0, 0, 0, 0, 0, 0};
    BK_TRY(kry_cholqr(ctx, &W, &W2, n, b, dG, Rtmp, &breakdown, dBall + (int64_t)steps * b * b, comm, ro, nr, piv));   // synchronises the stream
    if (getenv("BIGKRLS_KRY_PIVOTS"))    // (development: how well conditioned the new block was)
      fprintf(stderr, "[bigkrls] block Lanczos step %d: Gram pivots %.3e ... %.3e (ratio %.1e), second pass %.3e ... %.3e%s\n", steps,
              piv[0], piv[1], piv[1] > 0 ? piv[0] / piv[1] : 0.0, piv[3], piv[4], breakdown ? "  BREAKDOWN" : "");
    wnorm_sq = piv[2];
    // An ill-conditioned new block (cond(W) ~ sqrt(largest / smallest pivot) above 1e4): W was orthogonal to the
    // earlier blocks to eps |W|, but its smallest directions were scaled up by the QR, and with them what they had
    // left along the earlier blocks -- eps cond(W). One more Gram-Schmidt pass of the (now orthonormal) block
    // against all blocks and one more Cholesky QR; beta takes the third factor. What the pass removes from the block
    // recurrence is of the size of the rounding of W itself. (Numerically low-rank kernels: P = 2, 3.)
    if (!breakdown && piv[1] > 0.0 && piv[0] < 1e-8 * piv[1]) {
      BK_TRY(gram(B, dim, W, b, C));
      if (nr > 0) BK_TRY(gemm(ctx, 0, 0, nr, b, dim, -1.0, B + ro, n, C, dim, 1.0, W + ro, n));
      std::vector<double> R3, prev(Rtmp);
      bool bd3 = false;
      BK_TRY(kry_cholqr(ctx, &W, &W2, n, b, dG, R3, &bd3, nullptr, comm, ro, nr));
      if (bd3) {
        breakdown = true;
      } else {
        for (int c = 0; c < b; ++c)          // beta = R3 beta (both upper triangular)
          for (int r = 0; r <= c; ++r) {
            double acc = 0.0;
            for (int t = r; t <= c; ++t) acc += R3[r + (size_t)t * b] * prev[t + (size_t)c * b];
            Rtmp[r + (size_t)c * b] = acc;
          }
        double* hp = nullptr;
        BK_TRY(pinned_get(ctx, (int64_t)b * b, &hp));
        std::memcpy(hp, Rtmp.data(), (size_t)b * b * sizeof(double));
        BK_HIP(hipMemcpyAsync(dBall + (int64_t)steps * b * b, hp, (size_t)b * b * sizeof(double), hipMemcpyHostToDevice, st));
        BK_HIP(hipStreamSynchronize(st));
        if (getenv("BIGKRLS_VERBOSE")) fprintf(stderr, "[bigkrls] block Lanczos step %d: ill-conditioned block (pivot ratio %.1e) re-orthogonalised\n", steps, piv[0] / piv[1]);
      }
    }
    // The Krylov space is invariant to working precision when the new block is nothing but rounding: its largest
    // direction a millionth of the largest one any block had. Going on would normalise noise into unit vectors for
    // step after step until a pivot turns negative (P = 2: ten such steps before the first scheduled check); the
    // check comes now instead, as soon as the subspace has k columns, and decides with the true residuals.
    // ... and while it has fewer than k columns (Neig above the numerical rank of K), the block that is nothing but
    // rounding is REPLACED by a fresh random block orthogonalised against all blocks so far (what ARPACK does on an
    // invariant subspace): normalising the rounding instead works for a few steps, then the noise -- amplified by K
    // from step to step -- turns ill-conditioned and a pivot negative, short of k columns. The replaced W is what the
    // block recurrence loses: only a block below 1e-10 of the largest direction seen -- below the tolerance times
    // lambda_1, since no direction of a W exceeds lambda_1 -- is replaced, and its norm enters every later residual.
    // beta of this step is zero (the new block is not coupled to the old ones).
    {
      const double wfro = std::sqrt(std::max(wnorm_sq, 0.0));
      if (wmax_seen > 0.0 && wfro <= 1e-10 * wmax_seen && (int64_t)(steps + 1) * b < k && steps + 1 < maxsteps) {
        BK_TRY(fill_random(ctx, W, n * b, 20240229u + 7919u * (uint32_t)(steps + 1)));
        for (int pass = 0; pass < 2; ++pass) {
          BK_TRY(gram(B, dim, W, b, C));
          if (nr > 0) BK_TRY(gemm(ctx, 0, 0, nr, b, dim, -1.0, B + ro, n, C, dim, 1.0, W + ro, n));
        }
        std::vector<double> Rr;
        bool bdr = false;
        BK_TRY(kry_cholqr(ctx, &W, &W2, n, b, dG, Rr, &bdr, nullptr, comm, ro, nr));
        {
          double okv = bdr ? 0.0 : 1.0;
          BK_TRY(kry_agree_min(kop, &okv, 1));
          bdr = !(okv > 0.5);
        }
        if (!bdr) {
          breakdown = false;
          dropped += wfro;                 // (a bound for what all the replaced blocks together took from the recurrence)
          std::fill(Rtmp.begin(), Rtmp.end(), 0.0);
          BK_HIP(hipMemsetAsync(dBall + (int64_t)steps * b * b, 0, (size_t)b * b * sizeof(double), st));
          piv[1] = 0.0;       // (nothing of this step counts as a direction of a Krylov block)
          if (getenv("BIGKRLS_VERBOSE"))
            fprintf(stderr, "[bigkrls] block Lanczos step %d: invariant subspace of %lld columns (|W| = %.1e): continued with a random block\n",
                    steps, (long long)((steps + 1) * b), wfro);
        }
      }
    }
    {
      const double wmax = std::sqrt(std::max(piv[1], 0.0));
      if (!breakdown && !early_check_done && wmax <= 1e-6 * wmax_seen && (int64_t)(steps + 1) * b >= k) {
        next_check = std::min(next_check, steps + 1);      // (once: the checks after it follow their own forecasts)
        early_check_done = true;
      }
      wmax_seen = std::max(wmax_seen, wmax);
    }
    ++steps;
    {
      double okv = breakdown ? 0.0 : 1.0;
      BK_TRY(kry_agree_min(kop, &okv, 1));
      breakdown = !(okv > 0.5);
    }
    const bool last = breakdown || steps >= maxsteps;
    // ---- convergence check on the projected problem ---------------------------------------------
    if (last || steps >= next_check) {
      const int64_t m = (int64_t)steps * b;
      const std::vector<double>& beta = Rtmp;       // T[steps, steps - 1]: couples the last block to the next one
      // residuals |beta y_i[last block]| of the first k Ritz pairs of a projected problem whose eigenvectors end in the
      // b rows ylast (b x k, host)
      struct Resid { double worst = 0.0, worst_kept = 0.0; int64_t n_conv = 0; };
      auto residuals = [&](const double* th, const double* ylast) {
        Resid r;
        if (breakdown) {
          // the factor beta of the last block does not exist (its Gram matrix is not positive definite): every
          // residual |W y| is bounded by |W|_F -- zero to working precision when the Krylov space is invariant, which
          // is the breakdown that means convergence; anything else is reported as not converged (dense path)
          r.worst = r.worst_kept = std::sqrt(std::max(wnorm_sq, 0.0));
          return r;
        }
        for (int64_t i = 0; i < k; ++i) {
          double r2 = 0.0;
          for (int rr = 0; rr < b; ++rr) {
            double sacc = 0.0;
            for (int c = rr; c < b; ++c) sacc += beta[rr + (size_t)c * b] * ylast[c + (size_t)i * b];
            r2 += sacc * sacc;
          }
          r.worst = std::max(r.worst, std::sqrt(r2));
          if (std::sqrt(r2) <= tol * std::fabs(th[0])) ++r.n_conv;
          if (keep_thresh >= 0.0 && th[i] >= keep_thresh * th[0]) r.worst_kept = std::max(r.worst_kept, std::sqrt(r2));
        }
        r.worst = std::max(r.worst, dropped);      // (blocks replaced by random ones, see the step loop)
        return r;
      };
      // the m x m projected matrix (estimate == false), or the compressed one of the estimate (below), is decomposed by the
      // dense path; theta (mm values) and the last b rows of Y come down through the context's pinned buffer
      std::vector<double> Ylast((size_t)b * k), th;
      auto solve_projected = [&](double* dT, int64_t mm, double* dvalsT) -> int {
        int64_t nvY = 0;
        // (replicated, but after a fault one rank's copy may fail where its peers' do not: agreed, so nobody leaves alone)
        BK_TRY(kry_agree_status(kop, eigen(ctx, dT, mm, mm, mm, dvalsT, k, -1.0, (double*)pY, mm, &nvY, 0, 1, EIG_FULL)));
        double* hp = nullptr;
        BK_TRY(pinned_get(ctx, mm + (int64_t)b * k, &hp));
        BK_HIP(hipMemcpyAsync(hp, dvalsT, mm * sizeof(double), hipMemcpyDeviceToHost, st));
        BK_HIP(hipMemcpy2DAsync(hp + mm, b * sizeof(double), (double*)pY + (mm - b), mm * sizeof(double),
                                b * sizeof(double), k, hipMemcpyDeviceToHost, st));
        BK_HIP(hipStreamSynchronize(st));
        th.assign(hp, hp + mm);
        std::memcpy(Ylast.data(), hp + mm, (size_t)b * k * sizeof(double));
        return BIGKRLS_OK;
      };
      auto agree_worst = [&](double& worst, double& theta1) -> int {
        double ag[2] = {-worst, theta1};     // the largest residual and the smallest theta_1 over the ranks
        BK_TRY(kry_agree_min(kop, ag, 2));
        worst = -ag[0];
        theta1 = ag[1];
        return BIGKRLS_OK;
      };
      // Next check. The history of the worst residual is a plateau (O(1) while the subspace does not reach the
      // k-th eigenvalue yet) followed by a collapse at a steady x25 - x45 per step (measured: N = 100 000,
      // k = 1024: 2.6e-2, 6.0e-4, 1.4e-5, 3.1e-7; N = 50 000, k = 512: 6.0e-4, 3.5e-5, 1.4e-6, 6.1e-8), so a rate
      // fitted to two plateau samples overshoots by many steps. The distance to the tolerance at an assumed
      // collapse rate is used instead: an optimistic rate where a check (a dense eigensolve of T, ~12 us per
      // row) is cheaper than a step (2 n^2 b flops), so undershooting costs little, a cautious one otherwise.
      auto steps_to_go = [&](double worst, double theta1) {
        const double gain = check_is_cheap ? 40.0 : 15.0;
        int inc = (worst > 0.0) ? (int)std::ceil(std::log(worst / (tol * theta1)) / std::log(gain)) : 1;
        return std::max(1, std::min(inc, std::max(2, steps / 2)));
      };
      // ---- an ESTIMATE instead of the full check (round 6): where the last full check (at full_steps) put the end of
      // the iteration four or more steps away -- a forecast made from the plateau, which the check at its end only
      // ever corrected (C4: 20 -> 25 -> 27) -- the check at that point decomposes the COMPRESSED projected problem:
      // the k Ritz vectors of the last full check plus the blocks since, dimension k + 128 d instead of 128 steps
      // (C4, step 25: 1 152 instead of 3 200). The Lanczos relation holds exactly for that basis, so its residuals are
      // true residuals and track the full check's within a factor of two; the directions dropped at the compression
      // never come back, so it cannot replace the full check (its converged set can miss wanted eigenvalues,
      // tools/experiments/DEAD_ENDS.md): it only says where the next FULL check goes, and a full check follows at
      // once should it report convergence.
      if (m < k) {      // (a breakdown before the subspace had k columns: nothing to decompose -- the dense path)
        dim = m;
        break;
      }
      bool full = true;
      const int64_t mc = k + (int64_t)(steps - full_steps) * b;
      if (next_is_estimate && !last && full_steps > 0 && 2 * mc <= m && !getenv("BIGKRLS_KRY_NOEST")) {
        void* pT = nullptr;
        {
          int rc = ws_get(ctx, SLOT_KRY_T, (mc * mc + mc) * sizeof(double), &pT);
          if (rc == BIGKRLS_OK) rc = ws_get(ctx, SLOT_KRY_Y, mc * k * sizeof(double), &pY);
          BK_TRY(kry_agree_status(kop, rc));
        }
        double* dT = (double*)pT;
        hipLaunchKernelGGL(kry_assemble_compressed, dim3((unsigned)std::min<int64_t>((mc * mc + 255) / 256, 8192)), dim3(256), 0,
                           st, (const double*)dPrev, (const double*)(dPrev + k), (int)k, (const double*)dAall,
                           (const double*)dBall, full_steps, steps, b, dT);
        BK_CHECK_LAUNCH();
        BK_TRY(solve_projected(dT, mc, dT + mc * mc));
        Resid r = residuals(th.data(), Ylast.data());
        double worst = r.worst, theta1 = std::fabs(th[0]);
        BK_TRY(agree_worst(worst, theta1));
        if (getenv("BIGKRLS_VERBOSE"))
          fprintf(stderr, "[bigkrls] block Lanczos estimate: steps=%d (compressed from %d: %lld instead of %lld) worst=%.3e converged=%lld of %lld theta0=%.4e\n",
                  steps, full_steps, (long long)mc, (long long)m, worst, (long long)r.n_conv, (long long)k, th[0]);
        {
          char buf[160];
          snprintf(buf, sizeof buf, " [estimate steps=%d worst=%.3e theta0=%.6e]", steps, worst, th[0]);
          kry_diag += buf;
        }
        next_is_estimate = false;            // (what follows an estimate is a full check)
        if (!(worst <= tol * theta1)) {
          full = false;
          next_check = steps + steps_to_go(worst, theta1);
        }
      }
      if (full) {
        void *pT = nullptr;
        {
          int rc = ws_get(ctx, SLOT_KRY_T, (m * m + m) * sizeof(double), &pT);
          if (rc == BIGKRLS_OK) rc = ws_get(ctx, SLOT_KRY_Y, m * k * sizeof(double), &pY);
          BK_TRY(kry_agree_status(kop, rc));
        }
        double* dT = (double*)pT;
        double* dvalsT = dT + m * m;
        hipLaunchKernelGGL(kry_assemble_t, dim3((unsigned)std::min<int64_t>((m * m + 255) / 256, 8192)), dim3(256), 0, st,
                           (const double*)dAall, (const double*)dBall, steps, b, dT);
        BK_CHECK_LAUNCH();
        BK_TRY(solve_projected(dT, m, dvalsT));
        theta = th;
        Resid r = residuals(theta.data(), Ylast.data());
        double worst = r.worst, theta1 = std::fabs(theta[0]);
        BK_TRY(agree_worst(worst, theta1));
        if (getenv("BIGKRLS_VERBOSE"))
          fprintf(stderr, "[bigkrls] block Lanczos check: steps=%d worst=%.3e worst(kept)=%.3e converged=%lld of %lld theta0=%.4e\n",
                  steps, worst, r.worst_kept, (long long)r.n_conv, (long long)k, theta[0]);
        {
          char buf[160];
          snprintf(buf, sizeof buf, " [check steps=%d worst=%.3e theta0=%.6e breakdown=%d]", steps, worst, theta[0], (int)breakdown);
          kry_diag += buf;
        }
        if (worst <= tol * theta1 || last) {
          converged = worst <= tol * theta1;
          dim = m;
          break;
        }
        const int inc = steps_to_go(worst, theta1);
        next_check = steps + inc;
        // what the estimate at next_check needs of this check: the k Ritz values and C = beta Y[last block rows, :]
        // (b x k), the coupling of the Ritz vectors to the next block -- kept on the device behind the blocks of T
        next_is_estimate = inc >= 4;
        full_steps = steps;
        if (next_is_estimate) {
          double* hp = nullptr;
          BK_TRY(pinned_get(ctx, k + (int64_t)b * k, &hp));
          for (int64_t i = 0; i < k; ++i) hp[i] = theta[i];
          for (int64_t i = 0; i < k; ++i)
            for (int rr = 0; rr < b; ++rr) {
              double sacc = 0.0;
              for (int c = rr; c < b; ++c) sacc += beta[rr + (size_t)c * b] * Ylast[c + (size_t)i * b];
              hp[k + rr + i * b] = sacc;
            }
          BK_HIP(hipMemcpyAsync(dPrev, hp, (size_t)(k + (int64_t)b * k) * sizeof(double), hipMemcpyHostToDevice, st));
          BK_HIP(hipStreamSynchronize(st));     // (the pinned buffer is the eigensolver's as well)
        }
      }
      if (getenv("BIGKRLS_KRY_CHECK_EVERY")) { next_check = steps + 1; next_is_estimate = false; }   // (development: the convergence history)
    }
    BK_HIP(hipMemcpyAsync(B + (int64_t)steps * b * n, W, n * b * sizeof(double), hipMemcpyDeviceToDevice, st));
    dim = (int64_t)(steps + 1) * b;
  }
  if (getenv("BIGKRLS_VERBOSE")) fprintf(stderr, "[bigkrls] block Lanczos: n=%lld k=%lld steps=%d dim=%lld converged=%d\n", (long long)n, (long long)k, steps, (long long)dim, (int)converged);
#ifdef BK_FAULT_INJECT
  {
    const char* fault = getenv("BIGKRLS_FAULT");   // BIGKRLS_FAULT=noconv (test build): pretend the iteration stalled
    if (fault && std::string(fault) == "noconv") converged = false;
  }
#endif
  if (!converged && dim < k) {
    set_error("eigen (Krylov): breakdown with a subspace smaller than the number of requested pairs (a kernel of lower "
              "numerical rank than Neig: the dense path decomposes it -- taken automatically on one GPU; "
              "bigkrls_fit_dist: eigen_mode \"dense\")");
    return BIGKRLS_ENOCONV;
  }
  if (!converged) {
    set_error("eigen (Krylov): not converged within the subspace limit; use the dense path (BIGKRLS_EIGK=dense);" + kry_diag);
    return BIGKRLS_ENOCONV;
  }
  // ---- Ritz vectors Q = B Y -------------------------------------------------------------------------
  // With full re-orthogonalisation the Ritz pairs of T are Ritz pairs of K and |beta y_last| is their residual.
  // That is verified against K itself on the block of pairs that converge last (the smallest min(k, 128) Ritz
  // values: one more K-times-block product); only if the true residuals are not at the estimated level are
  // all k pairs refined by a Rayleigh-Ritz step against K (k more columns of K Q and a k x k eigenproblem),
  // which is what every call did before (BIGKRLS_KRY_REFINE=1 still forces it).
  double* Q = W;                        // n x k (W, W2 are n x max(b,k))
  double* KQ = W2;
  if (nr > 0) BK_TRY(gemm(ctx, 0, 0, nr, k, dim, 1.0, B + ro, n, (double*)pY, dim, 0.0, Q + ro, n));
  double* dH = dA + b * b;              // k x k + 2k
  double* dvalsH = dH + k * k;
  std::vector<double> hv(theta.begin(), theta.begin() + k);     // Ritz values of T, descending
  bool refine = false;
  {
    const char* rf = getenv("BIGKRLS_KRY_REFINE");
    refine = rf && std::string(rf) == "1";
  }
  const double* dvals_final = nullptr;
  if (!refine && ctx->caller_verifies && !getenv("BIGKRLS_KRY_SAMPLE")) {     // (the variable: A/B timing)
    // the fit checks ALL kept pairs against K itself right after this call (two +-1 combinations, tighter than this
    // sample's error threshold and as tight as its refinement threshold times sqrt(k)); if that fails, the decomposition
    // is redone with the flag cleared, i.e. with the sample check and, where it asks for it, the refinement below
    void* pT = nullptr;
    BK_TRY(ws_get(ctx, SLOT_KRY_T, (dim * dim + dim) * sizeof(double), &pT));
    dvals_final = (const double*)pT + dim * dim;
  } else if (!refine) {
    const int64_t bs = std::min<int64_t>(k, b), c0 = k - bs;
    // theta of T sits at the head of SLOT_KRY_T's value vector (device): dvalsT of the last check
    void* pT = nullptr;
    BK_TRY(ws_get(ctx, SLOT_KRY_T, (dim * dim + dim) * sizeof(double), &pT));
    const double* dtheta = (const double*)pT + dim * dim;
    BK_TRY(k_times(ctx, kop, n, Q + c0 * n, bs, KQ));
    hipLaunchKernelGGL(kry_resid_sq_kernel, dim3((unsigned)bs), dim3(256), 0, st, (const double*)KQ,
                       (const double*)(Q + c0 * n), dtheta + c0, n, ro, nr, dvalsH);
    BK_CHECK_LAUNCH();
    if (comm) BK_TRY(comm_all_reduce(comm, dvalsH, bs, COMM_SUM));       // the squared norms over all rows
    double* hp = nullptr;
    BK_TRY(pinned_get(ctx, bs, &hp));
    BK_HIP(hipMemcpyAsync(hp, dvalsH, bs * sizeof(double), hipMemcpyDeviceToHost, st));
    BK_HIP(hipStreamSynchronize(st));
    double rmax = 0.0;
    for (int64_t i = 0; i < bs; ++i) rmax = std::max(rmax, std::sqrt(std::max(hp[i], 0.0)));
    if (getenv("BIGKRLS_VERBOSE"))
      fprintf(stderr, "[bigkrls] block Lanczos: true residual of the last %lld pairs %.3e (tolerance %.3e)\n", (long long)bs,
              rmax, tol * std::fabs(theta[0]));
    {
      double ag[2] = {-rmax, std::fabs(theta[0])};
      BK_TRY(kry_agree_min(kop, ag, 2));
      if (!(-ag[0] <= 10.0 * tol * ag[1])) refine = true;    // also NaN
      // ... and a residual that is not even small against theta_1 means that the recurrence K B_j = ... did not hold
      // (a product or an exchange delivered wrong data): the Ritz pairs of T are then no pairs of K at all. Every rank
      // sees the same agreed figures, so every rank returns the error.
      if (!(-ag[0] <= 1e-3 * ag[1])) {
        char buf[200];
        snprintf(buf, sizeof buf, "eigen (Krylov): the Ritz pairs fail the check against K itself (residual %.3e, theta_1 %.3e): "
                                  "the block recurrence was corrupted", -ag[0], ag[1]);
        set_error(buf);
        ctx->corrupt_run = true;
        return BIGKRLS_ENOCONV;
      }
    }
    dvals_final = dtheta;
  }
  void* pZ = nullptr;
  if (refine) {
    BK_TRY(k_times(ctx, kop, n, Q, k, KQ));
    BK_TRY(gram(Q, k, KQ, k, dH));
    BK_TRY(ws_get(ctx, SLOT_KRY_Y, std::max<int64_t>(dim * k, k * k) * sizeof(double), &pZ));
    int64_t nvZ = 0;
    BK_TRY(eigen(ctx, dH, k, k, k, dvalsH, k, -1.0, (double*)pZ, k, &nvZ, 0, 1, EIG_FULL));
    {
      PinnedFetch pf(ctx, k);
      BK_TRY(pf.add(hv.data(), dvalsH, k * sizeof(double)));
      BK_TRY(pf.finish());
    }
    dvals_final = dvalsH;
  }
  BK_HIP(hipMemcpyAsync(vals, dvals_final, k * sizeof(double), hipMemcpyDeviceToDevice, st));
  int64_t nv = 0;
  if (keep_thresh >= 0.0) {
    for (int64_t i = 0; i < k; ++i)
      if (hv[i] >= keep_thresh * hv[0]) nv = i + 1;     // max(which(values >= eigtrunc * values[1]))
  } else {
    nv = k;
  }
  nv = std::min<int64_t>(nv, n_vecs_max);
  if (h_n_vecs) *h_n_vecs = nv;
  if (nv > 0) {
    if (nr > 0) {
      if (refine) BK_TRY(gemm(ctx, 0, 0, nr, nv, k, 1.0, Q + ro, n, (double*)pZ, k, 0.0, vecs + ro, ldv));
      else BK_TRY(copy_matrix(ctx, Q + ro, nr, nv, n, vecs + ro, ldv));
    }
    // every rank returns all rows of the kept eigenvectors (the later passes of the fit read row blocks of Q but
    // build column blocks of Q diag(w) Q'): one all-gather of the row blocks, n x nv doubles
    if (comm) BK_TRY(comm_gather_rows(comm, vecs + ro, nr, ldv, nv, kop.nb, n, vecs, ldv));
    const int64_t pc0 = nv * part_index / part_count, pc1 = nv * (part_index + 1) / part_count;
    if (pc0 > 0) BK_HIP(hipMemsetAsync(vecs, 0, (size_t)pc0 * ldv * sizeof(double), st));
    if (pc1 < nv) BK_HIP(hipMemsetAsync(vecs + pc1 * ldv, 0, (size_t)(nv - pc1) * ldv * sizeof(double), st));
  }
#ifdef BK_FAULT_INJECT
  {
    const char* fault = getenv("BIGKRLS_FAULT");   // BIGKRLS_FAULT=vals_ulp: see the dense path
    if (fault && std::string(fault) == "vals_ulp" && nv > 1) {
      fault_nudge_ulp<<<1, 1, 0, st>>>(vals + nv / 2);
      BK_HIP(hipGetLastError());
    }
    // BIGKRLS_FAULT=kry_swap: two kept Ritz vectors exchanged in the first call (see eig_swap in the dense path)
    static int kswap_calls = 0;
    const bool ks = fault && std::string(fault) == "kry_swap";
    if (!ks) kswap_calls = 0;
    if (ks && kswap_calls++ == 0 && nv > 3 && !comm) {
      fault_swap_cols<<<(unsigned)((n + 255) / 256), 256, 0, st>>>(vecs + (nv / 2) * ldv, vecs + (nv / 2 + 1) * ldv, (int)n);
      BK_HIP(hipGetLastError());
    }
  }
#endif
  BK_HIP(hipStreamSynchronize(st));
  return BIGKRLS_OK;
}

// The watchdog of a persistent kernel (pq_resident / bc_resident) fired: its workgroups were not
// co-resident, e.g. because another process or stream held part of the GPU. A is untouched, so the
// decomposition is redone once, in the same call, with the launch-per-step kernels (no spinning).
static int eigen_retry_without_resident(bigkrls_ctx* ctx, const double* A, int64_t n64, int64_t lda,
                                        int64_t n_vals, double* vals, int64_t n_vecs_max, double keep_thresh,
                                        double* vecs, int64_t ldv, int64_t* h_n_vecs, int part_index,
                                        int part_count) {
  if (ctx->side_stream) (void)hipStreamSynchronize(ctx->side_stream);   // look-ahead work still queued
  if (ctx->bg_stream) (void)hipStreamSynchronize(ctx->bg_stream);
  (void)hipStreamSynchronize(ctx->stream);
  if (ctx->no_resident) {
    set_error("eigen: watchdog of a persistent kernel fired although none should have been launched");
    return BIGKRLS_EHIP;
  }
  if (getenv("BIGKRLS_VERBOSE"))
    fprintf(stderr, "[bigkrls] eigen: persistent-kernel watchdog fired; retrying with per-step launches\n");
  ctx->n_replayed++;
  ctx->no_resident = true;
  const int rc = eigen(ctx, A, n64, lda, n_vals, vals, n_vecs_max, keep_thresh, vecs, ldv, h_n_vecs, part_index,
                       part_count, EIG_FULL);
  ctx->no_resident = false;
  return rc;
}

int eigen_krylov_dist(bigkrls_comm* comm, const double* Kcols, int64_t n, int64_t r0, int64_t r1, int64_t nb,
                      int64_t n_vals, double* vals, int64_t n_vecs_max, double keep_thresh, double* vecs, int64_t ldv,
                      int64_t* h_n_vecs) {
  BK_REQUIRE(comm && comm->ctx && (Kcols || r1 <= r0) && vals && n_vals > 0 && n_vals <= n, "eigen_krylov_dist: bad arguments");
  KTimes op;
  op.comm = comm;
  op.Kcols = Kcols;
  op.r0 = r0;
  op.r1 = r1;
  op.nb = nb;
  // every rank keeps all the kept eigenvector columns (part 0 of 1): the later passes work on row blocks of Q
  return eigen_krylov(comm->ctx, op, n, n_vals, vals, n_vecs_max, keep_thresh, vecs, ldv, h_n_vecs, 0, 1);
}

// Stage-1 state of a row-block distributed reduction, kept in the context between
// bigkrls_dev_s1_open and bigkrls_dev_eigen_resume.
struct DistS1 {
  S1Ops ops;
  int n = 0;
  bool panel_pending = false;   // a panel factorisation is running on the look-ahead stream (ev_join marks its end)
  int agg_mode = 0;             // panels per trailing update the ranks agreed on: 4 (groups of four, then pairs), 2 (pairs), 0 (none)
};

// The stage-1 panel loop as a captured hipGraph (round 6; OPT-IN, off by default). The loop has no host synchronisation
// and no host decision that depends on device data -- every launch dimension follows from n -- and its two streams meet
// only through ev_fork / ev_join / ev_join2 / ev_pq, the fork-join shape stream capture accepts: 824 nodes at N = 5 000,
// 3 823 at N = 20 000 (capture 1 / 5 ms, instantiation 12 / 16 ms, the launch call 2.4 / 11 ms of host time). Measured,
// same box, fresh processes: on contexts with a stream of their own the replayed graph was 3-6 ms faster at N = 20 000
// and equal at N = 5 000 (profiles/r06/r06b_stage1_graph_ab_*.log, before the gate below existed); on the default
// stream -- what bench.py and every Python-driven fit use -- and with the final kernels it is EQUAL OR SLOWER at
// N = 10 000 / 14 000 / 20 000 (0.1183 vs 0.1181, 0.2018 vs 0.2012-0.2019, 0.4044-0.4050 vs 0.4039-0.4045 s:
// r06c_stage1_graph_ab_*.log): the loop is not bound by launch overhead. Kept as an experiment switch, results bitwise
// those of the plain loop (tests/test_gpu_level1.py).
//   BIGKRLS_S1_GRAPH=-1: from the third decomposition of a size with n >= S1_GRAPH_MIN_N (the first two run the plain
//   loop: workspace growth, hipFuncSetAttribute and the side stream's creation must not happen inside a capture), the
//   executable graph cached for as long as the workspace it points into has not moved (ctx->ws_generation); =2: cached
//   from the second call on, any n; =1: capture + instantiate + launch at every call from the second on, timed under
//   BIGKRLS_VERBOSE. Never while profiling / tracing / after a watchdog. A context on the process's default stream
//   (which cannot be captured) captures on, and replays through, a stream of the context's own, ordered against the
//   default stream by two events.
constexpr int S1_GRAPH_MIN_N = 8192;
static int stage1_run(bigkrls_ctx* ctx, double* W, int n, double* taus1, const Stage1Ws& s1) {
  const char* genv = getenv("BIGKRLS_S1_GRAPH");           // (per call: the tests switch it)
  const int gmode = genv ? atoi(genv) : 0;
  if (gmode == 0 || ctx->profile || trace_on() || ctx->no_resident || ctx->side_is_main || (gmode < 0 && n < S1_GRAPH_MIN_N))
    return stage1_to_band(ctx, W, n, taus1, s1);
  if (ctx->s1_graph_warm_n != n) {
    ctx->s1_graph_warm_n = n;
    ctx->s1_graph_seen = 0;
  }
  if (ctx->s1_graph_seen < (gmode < 0 ? 2 : 1)) {
    ctx->s1_graph_seen++;
    return stage1_to_band(ctx, W, n, taus1, s1);
  }
  hipStream_t user = ctx->stream, st = user;
  if (user == nullptr) {             // the default stream cannot be captured
    if (!ctx->graph_stream) {
      BK_HIP(hipStreamCreateWithFlags(&ctx->graph_stream, hipStreamNonBlocking));
      BK_HIP(hipEventCreateWithFlags(&ctx->ev_graph, hipEventDisableTiming));
    }
    st = ctx->graph_stream;
  }
  const bool verbose = getenv("BIGKRLS_VERBOSE") != nullptr;
  auto now = [] { return std::chrono::steady_clock::now(); };
  auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) {
    return std::chrono::duration<double, std::milli>(b - a).count();
  };
  const bool hit = gmode != 1 && ctx->s1_graph_exec && ctx->s1_graph_n == n && ctx->s1_graph_gen == ctx->ws_generation &&
                   ctx->s1_graph_W == (const void*)W;
  const auto t0 = now();
  if (!hit) {
    if (ctx->s1_graph_exec) {
      BK_HIP(hipStreamSynchronize(user));
      if (st != user) BK_HIP(hipStreamSynchronize(st));
      (void)hipGraphExecDestroy((hipGraphExec_t)ctx->s1_graph_exec);
      ctx->s1_graph_exec = nullptr;
    }
    hipGraph_t graph = nullptr;
    BK_HIP(hipStreamBeginCapture(st, hipStreamCaptureModeRelaxed));
    ctx->stream = st;
    const int rc = stage1_to_band(ctx, W, n, taus1, s1);
    ctx->stream = user;
    const hipError_t ec = hipStreamEndCapture(st, &graph);      // (always: the stream must leave capture mode)
    if (rc != BIGKRLS_OK) {
      if (graph) (void)hipGraphDestroy(graph);
      return rc;
    }
    if (ec != hipSuccess || !graph) {
      set_error(std::string("stage 1: stream capture failed: ") + hipGetErrorString(ec));
      return BIGKRLS_EHIP;
    }
    const auto t1 = now();
    hipGraphExec_t exec = nullptr;
    const hipError_t ei = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
    size_t nnodes = 0;
    (void)hipGraphGetNodes(graph, nullptr, &nnodes);
    (void)hipGraphDestroy(graph);
    if (ei != hipSuccess) {
      set_error(std::string("stage 1: hipGraphInstantiate failed: ") + hipGetErrorString(ei));
      return BIGKRLS_EHIP;
    }
    ctx->s1_graph_exec = (void*)exec;
    ctx->s1_graph_n = n;
    ctx->s1_graph_gen = ctx->ws_generation;
    ctx->s1_graph_W = (const void*)W;
    if (verbose)
      fprintf(stderr, "[bigkrls]   stage 1 as a graph: %zu nodes, capture %.2f ms, instantiate %.2f ms\n", nnodes, ms(t0, t1),
              ms(t1, now()));
  }
  const auto t2 = now();
  if (st != user) {                  // everything queued on the default stream so far comes first ...
    BK_HIP(hipEventRecord(ctx->ev_graph, user));
    BK_HIP(hipStreamWaitEvent(st, ctx->ev_graph, 0));
  }
  BK_HIP(hipGraphLaunch((hipGraphExec_t)ctx->s1_graph_exec, st));
  if (st != user) {                  // ... and what follows on it waits for the graph
    BK_HIP(hipEventRecord(ctx->ev_graph, st));
    BK_HIP(hipStreamWaitEvent(user, ctx->ev_graph, 0));
  }
  if (verbose) {
    const auto t3 = now();
    BK_HIP(hipStreamSynchronize(user));
    fprintf(stderr, "[bigkrls]   stage 1 as a graph: launch call %.2f ms, launch to completion %.2f ms%s\n", ms(t2, t3),
            ms(t2, now()), hit ? " (cached executable graph)" : "");
  }
  return BIGKRLS_OK;
}

int eigen(bigkrls_ctx* ctx, const double* A, int64_t n64, int64_t lda, int64_t n_vals, double* vals,
          int64_t n_vecs_max, double keep_thresh, double* vecs, int64_t ldv, int64_t* h_n_vecs,
          int part_index, int part_count, int mode) {
  BK_REQUIRE(part_count >= 1 && part_index >= 0 && part_index < part_count, "eigen: bad column partition");
  BK_REQUIRE(n64 > 0 && n64 < (1ll << 30), "eigen: bad matrix");
  if (mode != EIG_SETUP_ONLY) {
    BK_REQUIRE((A || mode == EIG_RESUME) && vals, "eigen: bad matrix");
    BK_REQUIRE(n_vals > 0 && n_vals <= n64, "eigen: n_vals out of range");
    BK_REQUIRE(n_vecs_max >= 0 && n_vecs_max <= n64, "eigen: n_vecs_max out of range");
    BK_REQUIRE(n_vecs_max == 0 || (vecs && ldv >= n64), "eigen: bad eigenvector buffer");
  }
  if (mode != EIG_FULL) BK_REQUIRE(n64 > 4 * S2_B, "eigen: the distributed dense path needs n > 256");
  if (mode == EIG_FULL) {
    // Neig << N: block Lanczos (the reference switches to eigs_sym for Neig < N, src/eigen.cpp:18-22);
    // BIGKRLS_EIGK=dense keeps the dense path, =krylov forces the iterative one when Neig <= N/4
    const char* ek = getenv("BIGKRLS_EIGK");
    const std::string mode = ek ? ek : "";
    // (measured: N = 12 000, Neig = 512 dense 0.33 s vs 0.47 s; N = 50 000, Neig = 512 dense 6.9 s vs 0.88 s)
    const bool small_k = n_vals * 8 <= n64 && n64 >= 16384;
    if (mode != "dense" && n_vals < n64 && (small_k || (mode == "krylov" && n_vals * 4 <= n64 && n64 >= 1024))) {
      KTimes whole;
      whole.A = A;
      whole.lda = lda;
      const int rc = eigen_krylov(ctx, whole, n64, n_vals, vals, n_vecs_max, keep_thresh, vecs, ldv, h_n_vecs,
                                  part_index, part_count);
      // A spectrum the iteration does not resolve within its subspace limit (or a breakdown) is
      // handed to the dense path in the same call -- A is untouched. Only an explicit
      // BIGKRLS_EIGK=krylov reports the non-convergence.
      if (rc != BIGKRLS_ENOCONV || mode == "krylov") return rc;
      // ... provided its workspace fits: W, Q0, Q1, U = 4 n^2 doubles beside what the caller holds (320 GB at
      // n = 100 000). Otherwise the non-convergence is the answer, with what the iteration saw.
      {
        size_t free_b = 0, total_b = 0;
        BK_HIP(hipMemGetInfo(&free_b, &total_b));
#ifdef BK_FAULT_INJECT
        if (const char* fault = getenv("BIGKRLS_FAULT_NOFIT")) {   // (test build) pretend the device is this full
          if (atoi(fault) != 0) free_b = 0;
        }
#endif
        int64_t held = 0;
        for (int sl : {SLOT_EIG_A, SLOT_EIG_Q0, SLOT_EIG_Q1, SLOT_EIG_U}) held += std::min<int64_t>(ctx->ws_bytes[sl], n64 * n64 * 8);
        // the Krylov workspace is released first if that is what it takes
        int64_t kry = 0;
        for (int sl : {SLOT_KRY_B, SLOT_KRY_W, SLOT_KRY_T, SLOT_KRY_Y}) kry += ctx->ws_bytes[sl];
        const double need = 4.0 * 8.0 * (double)n64 * (double)n64 * 1.02 - (double)held;
        if (need > (double)free_b + (double)kry) {
          char buf[256];
          snprintf(buf, sizeof buf,
                   "eigen: the block Lanczos did not converge within its subspace limit and the dense fallback does "
                   "not fit (needs %.1f GB more, %.1f GB free);", need / 1e9, ((double)free_b + (double)kry) / 1e9);
          set_error(std::string(buf) + kry_diag);
          return BIGKRLS_ENOCONV;
        }
        if (need > (double)free_b) {
          for (int sl : {SLOT_KRY_B, SLOT_KRY_W, SLOT_KRY_T, SLOT_KRY_Y}) {
            if (ctx->ws[sl]) { BK_HIP(hipStreamSynchronize(ctx->stream)); BK_HIP(hipFree(ctx->ws[sl])); ctx->ws[sl] = nullptr; ctx->ws_bytes[sl] = 0; }
          }
        }
      }
      if (getenv("BIGKRLS_VERBOSE"))
        fprintf(stderr, "[bigkrls] block Lanczos did not converge: dense path;%s\n", kry_diag.c_str());
    }
  }
  const int n = (int)n64;
  const int64_t N = n;
  hipStream_t st = ctx->stream;
  // BIGKRLS_VERBOSE: wall-clock of each phase on stderr (adds a stream synchronisation per phase)
  const bool verbose = getenv("BIGKRLS_VERBOSE") != nullptr;
  auto t_phase = std::chrono::steady_clock::now();
  auto tick = [&](const char* name) {
    if (!verbose) return;
    (void)hipStreamSynchronize(st);
    const auto now = std::chrono::steady_clock::now();
    fprintf(stderr, "[bigkrls] eigen n=%d: %-28s %8.2f ms\n", n, name,
            std::chrono::duration<double, std::milli>(now - t_phase).count());
    t_phase = now;
  };
  void *pW = nullptr, *pQ0 = nullptr, *pQ1 = nullptr, *pU = nullptr, *pP = nullptr, *pV = nullptr;
  BK_TRY(ws_get(ctx, SLOT_EIG_A, N * N * sizeof(double), &pW));
  BK_TRY(ws_get(ctx, SLOT_EIG_PANEL, 4 * N * TRD_NB * sizeof(double), &pP));
  BK_TRY(ws_get(ctx, SLOT_EIG_VEC, (5 * N + 4 * TRD_NB + 2 * ((N + 255) / 256 + 2)) * sizeof(double), &pV));
  double* W = (double*)pW;
  double* P1 = (double*)pP;
  double* P2 = P1 + 2 * N * TRD_NB;
  double* d = (double*)pV;
  double* e = d + N;
  double* tau = e + N;
  double* scratch = tau + N;
  if (mode == EIG_FULL) BK_TRY(copy_matrix(ctx, A, N, N, lda, W, N));
  BK_HIP(hipMemsetAsync(d, 0, 3 * N * sizeof(double), st));
  // tiled-symv partial buffers live in the (later) U slot of the divide & conquer
  const int64_t sv_prow = (N / SV_CW + 2) * N, sv_pcol = 10 * N, sv_prow2 = (N / (SV_CW * 32) + 2) * N;
  const int64_t u_doubles = std::max<int64_t>(N * N, sv_prow + sv_pcol + sv_prow2 + 16 * 2 * TRD_NB);
  BK_TRY(ws_get(ctx, SLOT_EIG_U, u_doubles * sizeof(double), &pU));
  SymvWs sw{(double*)pU, (double*)pU + sv_prow, (double*)pU + sv_prow + sv_pcol,
            (double*)pU + sv_prow + sv_pcol + sv_prow2};
  // two-stage (band) reduction is the default above 4 panels; BIGKRLS_EIG=1stage forces the
  // one-stage (symv) reduction (kept for small n and as a cross-check: the tests run both)
  const char* eig_env = getenv("BIGKRLS_EIG");
  const bool two_stage = mode != EIG_FULL || (!(eig_env && std::string(eig_env) == "1stage") && n > 4 * S2_B);
  double *taus1 = nullptr, *AB = nullptr, *VV = nullptr, *TT = nullptr;
  int64_t* d_soff = nullptr;
  std::vector<int64_t> bt2_toff;      // (source of an asynchronous copy: lives until the function returns)
  double* bt2_T = nullptr;
  int64_t* bt2_dtoff = nullptr;
  int* bt2_err = nullptr;             // watchdog word of the persistent stage-2 back-transform (when it ran)
  Bt1Plan bt1;
  double *bt1_V = nullptr, *bt1_T = nullptr;
  Stage1Ws s1{};
  if (two_stage) {
    const int64_t nb64 = N * S2_B;
    void *p2 = nullptr, *pvv = nullptr;
    const int64_t maxb = (N + 255) / 256 + 2;
    const int64_t nmail1 = 2 * maxb * 2 * S2_B;  // pq_resident: 2 buffers x workgroups x 64 word pairs
    const int64_t nmail = nmail1 + 2 * 16 * 2 * S2_B;   // + 2 buffers x 16 group boxes
    const int64_t npqc = PQC_MAIL_DOUBLES;   // pq_chol mailbox
    const int64_t npart = 2 * maxb + 2 * maxb * S2_B + S2_B + S2_B * S2_B + 2 * N + nmail;
    const int64_t ntall = (N / S2_B + 2) * S2_B * S2_B;
    const int64_t nfpart = (N / (S1F_CHUNKS * S2_B) + 3) * S2_B * S2_B;   // partial V'Y blocks of the fused small products
    const int64_t need = 7 * nb64 + 8 * S2_B * S2_B + npart + ntall + nfpart + npqc + 2 * N /*taus1, scales1*/ +
                         (int64_t)S2_LD * N /*AB*/ + N + 8 /*soff as int64*/;
    BK_TRY(ws_get(ctx, SLOT_EIG_BT, need * sizeof(double), &p2));
    double* q = (double*)p2;
    s1.Vp = q; q += nb64;
    s1.Y = q; q += nb64;
    s1.Y2 = q; q += nb64;
    s1.PZ1 = q; q += 2 * nb64;
    s1.PZ2 = q; q += 2 * nb64;
    s1.small = q; q += 8 * S2_B * S2_B;
    s1.part = q; q += npart;
    s1.mail = s1.part + (npart - nmail);
    s1.mail2 = s1.mail + nmail1;
    s1.err = (int*)scratch;
    if (mode != EIG_RESUME) {   // (a resumed decomposition keeps the watchdog word of its stage 1)
      BK_HIP(hipMemsetAsync(s1.mail, 0, nmail * sizeof(double), st));
      BK_HIP(hipMemsetAsync(s1.err, 0, sizeof(int), st));
    }
    s1.Tall = q; q += ntall;
    s1.fpart = q; q += nfpart;
    s1.pqc = q; q += npqc;
    s1.fallback = (int*)scratch + 4;
    if (mode != EIG_RESUME) {
      BK_HIP(hipMemsetAsync(s1.pqc, 0, npqc * sizeof(double), st));
      BK_HIP(hipMemsetAsync(s1.fallback, 0, sizeof(int), st));
    }
    taus1 = q; q += 2 * N;
    AB = q; q += (int64_t)S2_LD * N;
    d_soff = (int64_t*)q;
    Stage2Plan plan = stage2_plan(n);
    BK_TRY(ws_get(ctx, SLOT_EIG_VV, (plan.nrefl * (S2_B + 1) + 16) * sizeof(double), &pvv));
    VV = (double*)pvv;
    TT = VV + plan.nrefl * S2_B;
    {
      // (through the pinned upload arena, sized here for the whole decomposition so that it is not reallocated later)
      PinnedStage up(ctx);
      BK_TRY(up.reserve(std::max(dc_stage_bytes(n), plan.soff.size() * sizeof(int64_t) + 4096)));
      BK_TRY(up.put(d_soff, plan.soff.data(), plan.soff.size() * sizeof(int64_t)));
    }
    if (mode != EIG_RESUME) BK_HIP(hipMemsetAsync(taus1, 0, 2 * N * sizeof(double), st));
    if ((mode == EIG_FULL || mode == EIG_SETUP_ONLY) && n >= S1_AGG_MIN_M + 4 * S2_B) {
      // reflector blocks of the two panel groups whose trailing update is pending (stage1_to_band)
      void* pagg = nullptr;
      const int64_t blk = N * 4 * S2_B;
      BK_TRY(ws_get(ctx, SLOT_EIG_AGG, (4 * blk + 3 * 4 * S2_B * S2_B) * (int64_t)sizeof(double), &pagg));
      double* qa = (double*)pagg;
      s1.aggPZ1[0] = qa; s1.aggPZ2[0] = qa + blk; s1.aggPZ1[1] = qa + 2 * blk; s1.aggPZ2[1] = qa + 3 * blk;
      s1.aggC = qa + 4 * blk;
      s1.aggLd = N;
    }
    if (mode == EIG_SETUP_ONLY) {
      // the distributed stage 1 drives the panel steps itself (bigkrls_dev_s1_*): hand it the layout
      if (ctx->dist_s1 && ctx->dist_s1_free) ctx->dist_s1_free(ctx->dist_s1);
      DistS1* ds = new DistS1();
      ds->n = n;
      ctx->dist_s1 = ds;
      ctx->dist_s1_free = [](void* p) { delete (DistS1*)p; };
      BK_TRY(ds->ops.init(ctx, W, n, taus1, s1));
      BK_HIP(hipStreamSynchronize(st));   // plan.soff (host) was the source of an async copy
      return BIGKRLS_OK;
    }
    tick("setup + copy");
    if (mode == EIG_FULL) BK_TRY(stage1_run(ctx, W, n, taus1, s1));
    if (mode == EIG_FULL && getenv("BIGKRLS_VERBOSE") && s1.pqc) {
      double nfb = 0.0;
      PinnedFetch pfb(ctx, 1);       // (never a device -> pageable copy: the runtime would pin / unpin the page)
      BK_TRY(pfb.add(&nfb, s1.pqc + PQC_OFF_SLICES + PQC_NG + 32, sizeof(double)));
      BK_TRY(pfb.finish());
      fprintf(stderr, "[bigkrls]   panels left to the Householder kernel by pq_chol: %d\n", (int)nfb);
    }
    {
      // watchdog word of the register-resident panel QR: checked before stage 2 consumes the band
      int h_err1 = 0;
      PinnedFetch pf1(ctx, 1);
      BK_TRY(pf1.add(&h_err1, s1.err, sizeof(int)));
      BK_TRY(pf1.finish());
#ifdef BK_FAULT_INJECT
      // BIGKRLS_FAULT=watchdog (test build): pretend the watchdog fired on the first attempt
      const char* fault = getenv("BIGKRLS_FAULT");
      if (fault && std::string(fault) == "watchdog" && !ctx->no_resident) h_err1 = 1;
#endif
      if (h_err1 != 0 && mode == EIG_RESUME) {
        set_error("eigen: watchdog of the register-resident panel QR fired during the distributed stage 1");
        return BK_EWATCHDOG;     // the caller replays the decomposition on every rank (csrc/fit.hip)
      }
      if (h_err1 != 0) return eigen_retry_without_resident(ctx, A, n64, lda, n_vals, vals, n_vecs_max, keep_thresh,
                                                           vecs, ldv, h_n_vecs, part_index, part_count);
    }
    tick("stage 1 (dense -> band)");
    int blocks = (int)std::min<int64_t>(((int64_t)S2_LD * N + 255) / 256, 8192);
    hipLaunchKernelGGL(s1_extract_band, dim3(blocks), dim3(256), 0, st, (const double*)W, n, AB);
    BK_CHECK_LAUNCH();
    if (trace_on()) BK_TRY(trace_point(ctx, st, "R:eig_band", AB, (int64_t)S2_LD * N, mode));
    // progress flags / error word of the persistent bulge-chasing kernel: the (unused here)
    // tau and scratch vectors of the one-stage path
    int* bc_err = (int*)scratch;
    // merged block reflectors of the stage-1 back-transform (BIGKRLS_BT1=panel: one panel per step):
    // built on the look-ahead stream from stage 1's output while the main stream goes on
    const char* bt1_env = getenv("BIGKRLS_BT1");
    const bool bt1_grouped = n_vecs_max > 0 && !(bt1_env && std::string(bt1_env) == "panel");
    const int64_t LT1 = bt1_grp_for(n) * S2_B;
    double* bt1_G = nullptr;
    if (bt1_grouped) {
      bt1 = bt1_plan(n);
      void *pvb = nullptr, *ptb = nullptr;
      BK_TRY(ws_get(ctx, SLOT_EIG_VBIG, (bt1.vtotal + 16) * (int64_t)sizeof(double), &pvb));
      // per group: the merged T (LT x LT), its Gram matrix (LT x LT), the recurrence's scratch (LT x 64); then the group table
      const int64_t ng1 = (int64_t)bt1.k0.size();
      BK_TRY(ws_get(ctx, SLOT_EIG_TBIG,
                    (ng1 * (2 * LT1 * LT1 + LT1 * S2_B) + 16) * (int64_t)sizeof(double) + (ng1 + 1) * (int64_t)sizeof(Bt1Group),
                    &ptb));
      bt1_V = (double*)pvb;
      bt1_T = (double*)ptb;
      bt1_G = bt1_T + ng1 * LT1 * LT1;
      BK_TRY(side_stream_get(ctx));
    }
    BK_TRY(stage2_to_tridiag(ctx, AB, n, d_soff, VV, TT, d, e, (int*)tau, bc_err));
    // (BIGKRLS_BG=1, an A/B switch: what is precomputed for the back-transforms goes to a lowest-priority stream instead
    //  of the high-priority look-ahead stream. Measured slower: C3 0.4029-0.4043 vs 0.4009-0.4012 s, same box
    //  (profiles/r06/r06g_bg_stream_ab_*.log) -- the merged blocks then finish late behind the stage-2 back-transform)
    auto pre_stream = [&]() -> hipStream_t {
      static const bool bg = [] { const char* e = getenv("BIGKRLS_BG"); return e && e[0] == '1'; }();
      return bg ? ctx->bg_stream : ctx->side_stream;
    };
    // T factors of the stage-2 back-transform tasks (BIGKRLS_BT2=seq: reflector-by-reflector kernel)
    const char* bt2_env = getenv("BIGKRLS_BT2");
    if (n_vecs_max > 0 && n >= 3 && !(bt2_env && std::string(bt2_env) == "seq")) {
      bt2_toff = bt2_task_offsets(n);
      const int64_t ntasks = bt2_toff.back();
      void* pt2 = nullptr;
      BK_TRY(ws_get(ctx, SLOT_EIG_T2,
                    (ntasks * BT2_G * BT2_G + (int64_t)bt2_toff.size() + 8) * (int64_t)sizeof(double), &pt2));
      bt2_T = (double*)pt2;
      bt2_dtoff = (int64_t*)(bt2_T + ntasks * BT2_G * BT2_G);
      // on the look-ahead stream, behind the bulge chasing (queued now, while it runs: issuing the ~900 launches of the
      // merged blocks below takes the host 4 ms): they are not needed before the divide & conquer has finished
      BK_TRY(side_stream_get(ctx));
      hipStream_t side = pre_stream();
      BK_HIP(hipEventRecord(ctx->ev_fork, st));          // the bulge chasing is done
      BK_HIP(hipStreamWaitEvent(side, ctx->ev_fork, 0));
      BK_HIP(hipMemcpyAsync(bt2_dtoff, bt2_toff.data(), bt2_toff.size() * sizeof(int64_t), hipMemcpyHostToDevice, side));
      const int ngroups = (int)bt2_toff.size() - 1, ntmax = (n - 2) / S2_B + 1;
      BK_TRY(ensure_dyn_smem(ctx, (const void*)bt2_build_t, BT2T_SMEM));
      hipLaunchKernelGGL(bt2_build_t, dim3((unsigned)(((int64_t)ntmax * ngroups + 3) / 4)), dim3(256), BT2T_SMEM, side, n,
                         (const int64_t*)d_soff, (const double*)VV, (const double*)TT, (const int64_t*)bt2_dtoff, bt2_T,
                         ntmax, ngroups);
      BK_CHECK_LAUNCH();
      BK_HIP(hipEventRecord(ctx->ev_join2, side));       // what the stage-2 back-transform waits for
    }
    if (bt1_grouped) {
      // The merged block reflectors of the stage-1 back-transform, on the look-ahead stream BEHIND the T factors above
      // (round 6): after the bulge chasing (its workgroups fill every CU's LDS; sharing the GPU only slows it down),
      // beside the divide & conquer -- and, since that takes 17 ms at N = 20 000 and these launches 20, beside the first
      // milliseconds of the stage-2 back-transform, which no longer waits for them: only the stage-1 back-transform does.
      BK_HIP(hipEventRecord(ctx->ev_fork, st));
      BK_HIP(hipStreamWaitEvent(pre_stream(), ctx->ev_fork, 0));
      {
        const int64_t ng1 = (int64_t)bt1.k0.size();
        double* bt1_tmp = bt1_G + ng1 * LT1 * LT1;
        Bt1Group* d_groups = (Bt1Group*)(bt1_tmp + ng1 * LT1 * S2_B + 8);
        BK_TRY(bt1_precompute(ctx, W, n, taus1, s1.Tall, bt1, bt1_V, bt1_T, bt1_G, bt1_tmp, d_groups, pre_stream()));
      }
      BK_HIP(hipEventRecord(ctx->ev_join, pre_stream()));
    }
    int h_err = 0;
    PinnedFetch pf2(ctx, 1);
    BK_TRY(pf2.add(&h_err, bc_err, sizeof(int)));
    BK_TRY(pf2.finish());              // (also: plan.soff (host) was the source of an async copy)
    tick("stage 2 (band -> tridiagonal)");
    if (h_err != 0 && mode == EIG_RESUME) {
      set_error("eigen: watchdog of the LDS-resident bulge chasing fired after the distributed stage 1");
      return BK_EWATCHDOG;
    }
    if (h_err != 0)
      return eigen_retry_without_resident(ctx, A, n64, lda, n_vals, vals, n_vecs_max, keep_thresh, vecs, ldv,
                                          h_n_vecs, part_index, part_count);
  } else if (n >= 2) {
    BK_TRY(tridiagonalize(ctx, W, n, d, e, tau, P1, P2, scratch, sw));
  } else {
    BK_HIP(hipMemcpyAsync(d, W, sizeof(double), hipMemcpyDeviceToDevice, st));
  }
  std::vector<double> hd(n), he(n);
  {
    PinnedFetch pf(ctx, 2 * N);
    BK_TRY(pf.add(hd.data(), d, N * sizeof(double)));
    BK_TRY(pf.add(he.data(), e, N * sizeof(double)));
    BK_TRY(pf.finish());
  }
  for (int i = 0; i < n; ++i)
    if (!std::isfinite(hd[i]) || (i < n - 1 && !std::isfinite(he[i]))) {
      set_error("eigen: non-finite entries after tridiagonalisation (NaN/Inf in the input?)");
      ctx->corrupt_run = true;
      return BIGKRLS_EINVAL;
    }
  if (trace_on()) {
    BK_TRY(trace_host("R:eig_d", hd.data(), n, mode));
    BK_TRY(trace_host("R:eig_e", he.data(), n - 1, mode));
  }
  BK_TRY(ws_get(ctx, SLOT_EIG_Q0, N * N * sizeof(double), &pQ0));
  BK_TRY(ws_get(ctx, SLOT_EIG_Q1, N * N * sizeof(double), &pQ1));
  std::vector<double> vals_desc;
  std::vector<int> src_cols;
  double* Qfin = nullptr;
  BK_TRY(divide_conquer(ctx, n, hd, he, (double*)pQ0, (double*)pQ1, (double*)pU, n_vals,
                        n_vecs_max, keep_thresh, vals_desc, src_cols, &Qfin));
  tick("divide & conquer");
  if (trace_on()) {
    BK_TRY(trace_host("R:eig_vals", vals_desc.data(), n_vals, (int64_t)src_cols.size()));
  }
  PinnedStage up(ctx);      // (the divide & conquer ended with a synchronisation: the arena is free)
  BK_TRY(up.reserve((size_t)n_vals * sizeof(double) + (size_t)n * sizeof(int) + 4096));
  BK_TRY(up.put(vals, vals_desc.data(), n_vals * sizeof(double)));
  const int nv = (int)src_cols.size();
  if (h_n_vecs) *h_n_vecs = nv;
  if (nv > 0 && n_vecs_max > 0) {
    void* pidx = nullptr;
    BK_TRY(ws_get(ctx, SLOT_EIG_INT, (int64_t)10 * n * sizeof(int), &pidx));
    int* d_src = (int*)pidx;
    BK_TRY(up.put(d_src, src_cols.data(), nv * sizeof(int)));
    int blocks = (int)std::min<int64_t>((N * nv + 255) / 256, 8192);
    hipLaunchKernelGGL(gather_cols, dim3(blocks), dim3(256), 0, st, n, nv, (const int*)d_src,
                       (const double*)Qfin, N, vecs, ldv);
    BK_CHECK_LAUNCH();
    // Multi-GPU: every rank runs the (replicated) reduction and divide & conquer, but
    // back-transforms only its own slice of the kept eigenvector columns; the other columns are
    // returned as zeros, so that an all-reduce (sum) over the ranks assembles Q exactly.
    const int pc0 = (int)((int64_t)nv * part_index / part_count);
    const int pc1 = (int)((int64_t)nv * (part_index + 1) / part_count);
    if (pc0 > 0) BK_HIP(hipMemsetAsync(vecs, 0, (size_t)pc0 * ldv * sizeof(double), st));
    if (pc1 < nv) BK_HIP(hipMemsetAsync(vecs + (int64_t)pc1 * ldv, 0, (size_t)(nv - pc1) * ldv * sizeof(double), st));
    double* pvecs = vecs + (int64_t)pc0 * ldv;
    const int pnv = pc1 - pc0;
    if (pnv > 0 && two_stage) {
      tick("gather kept columns");
      if (bt2_T != nullptr) BK_HIP(hipStreamWaitEvent(st, ctx->ev_join2, 0));
      // (the watchdog word of the bulge chasing, read back as zero above, now serves the persistent back-transform)
      bt2_err = (bt2_T != nullptr) ? (int*)scratch : nullptr;
      BK_TRY(back_transform_stage2(ctx, n, d_soff, VV, TT, pvecs, ldv, pnv, bt2_dtoff, bt2_T,
                                   bt2_toff.empty() ? 0 : bt2_toff.back(), bt2_err));
      tick("back-transform stage 2");
      void* pw12 = nullptr;
      const int64_t bt1_w = (int64_t)bt1_grp_for(n) * S2_B;
      BK_TRY(ws_get(ctx, SLOT_EIG_Z, 2 * bt1_w * pnv * sizeof(double), &pw12));
      if (bt1_V != nullptr) {
        BK_HIP(hipStreamWaitEvent(st, ctx->ev_join, 0));
        BK_TRY(back_transform_stage1_grouped(ctx, n, bt1, bt1_V, bt1_T, pvecs, ldv, pnv, (double*)pw12,
                                             (double*)pw12 + bt1_w * pnv));
      }
      else
        BK_TRY(back_transform_stage1(ctx, W, n, taus1, pvecs, ldv, pnv, s1.Vp, s1.Tall, (double*)pw12,
                                     (double*)pw12 + (int64_t)S2_B * pnv));
      tick("back-transform stage 1");
    } else if (pnv > 0) {
      BK_TRY(back_transform(ctx, W, n, tau, pvecs, ldv, pnv));
    }
  }
  // (also when no column was back-transformed: nothing of this call may still run on the look-ahead stream)
  if (bt2_T != nullptr) BK_HIP(hipStreamWaitEvent(st, ctx->ev_join2, 0));
  if (bt1_V != nullptr) BK_HIP(hipStreamWaitEvent(st, ctx->ev_join, 0));
#ifdef BK_FAULT_INJECT
  // BIGKRLS_FAULT=eig_garbage (test build): the FIRST decomposition after the variable is set comes back with its
  // middle kept eigenvector scaled by 1.001 -- a wrong result without any error, the kind the fit's verification
  // (csrc/fit.hip) exists for; =eig_garbage_always: every decomposition does
  {
    static int garbage_calls = 0;
    const char* fault = getenv("BIGKRLS_FAULT");
    const bool once = fault && std::string(fault) == "eig_garbage", always = fault && std::string(fault) == "eig_garbage_always";
    if (!once && !always) garbage_calls = 0;
    // (a column the fit's check samples: the middle kept one; in a multi-GPU fit the last one, owned by the last rank)
    const int gc = part_count == 1 ? nv / 2 : nv - 1;
    if ((always || (once && garbage_calls++ == 0)) && nv > 0 && n_vecs_max > 0 && part_index == part_count - 1)
      BK_TRY(scale(ctx, N, 1.001, vecs + (int64_t)gc * ldv));
    // BIGKRLS_FAULT=eig_swap / eig_swap_always: two kept eigenvectors exchanged -- every column still has norm 1, the
    // combinations Q r keep |Q r|^2 = k: only the comparison with K Q r (on one GPU deferred to the fit's pass over K
    // for the marginal effects, csrc/fit.hip) can see it
    {
      static int swap_calls = 0;
      const bool sonce = fault && std::string(fault) == "eig_swap", salways = fault && std::string(fault) == "eig_swap_always";
      if (!sonce && !salways) swap_calls = 0;
      if ((salways || (sonce && swap_calls++ == 0)) && nv > 3 && n_vecs_max > 0 && part_count == 1 && keep_thresh >= 0.0) {
        fault_swap_cols<<<(N + 255) / 256, 256, 0, st>>>(vecs + (int64_t)(nv / 2) * ldv, vecs + (int64_t)(nv / 2 + 1) * ldv, (int)N);
        BK_HIP(hipGetLastError());
      }
    }
    // BIGKRLS_FAULT=vals_ulp (set in ONE rank's process): this rank's copy of the replicated eigenvalues differs from its
    // peers' in the last bit of one kept value -- a valid decomposition the fit's check against K lets through; the
    // multi-GPU fit must still run its lambda search on identical values everywhere (csrc/fit.hip: rank 0's are broadcast)
    if (fault && std::string(fault) == "vals_ulp" && nv > 1 && keep_thresh >= 0.0) {   // (not the inner solves of the Lanczos)
      fault_nudge_ulp<<<1, 1, 0, st>>>(vals + nv / 2);
      BK_HIP(hipGetLastError());
    }
  }
#endif
  if (trace_on() && nv > 0 && n_vecs_max > 0) {
    const int tc0 = (int)((int64_t)nv * part_index / part_count), tc1 = (int)((int64_t)nv * (part_index + 1) / part_count);
    if (tc1 > tc0 && ldv == N) BK_TRY(trace_point(ctx, st, "L:eig_Qpart", vecs + (int64_t)tc0 * ldv, (int64_t)(tc1 - tc0) * ldv, tc0));
  }
  if (bt2_err != nullptr) {
    // watchdog word of the persistent stage-2 back-transform: Z is garbage if it fired -> the decomposition is redone
    // with per-wavefront launches (by the caller's replay in the distributed fit)
    int h_err3 = 0;
    PinnedFetch pf3(ctx, 1);
    BK_TRY(pf3.add(&h_err3, bt2_err, sizeof(int)));
    BK_TRY(pf3.finish());
#ifdef BK_FAULT_INJECT
    const char* fault = getenv("BIGKRLS_FAULT");
    if (fault && std::string(fault) == "watchdog_bt2" && !ctx->no_resident) h_err3 = 1;
#endif
    if (h_err3 != 0 && mode == EIG_RESUME) {
      set_error("eigen: watchdog of the persistent stage-2 back-transform fired after the distributed stage 1");
      return BK_EWATCHDOG;
    }
    if (h_err3 != 0)
      return eigen_retry_without_resident(ctx, A, n64, lda, n_vals, vals, n_vecs_max, keep_thresh, vecs, ldv,
                                          h_n_vecs, part_index, part_count);
    return BIGKRLS_OK;
  }
  BK_HIP(hipStreamSynchronize(st));
  return BIGKRLS_OK;
}

// ---------------------------------------------------------------------------
// Row-block distributed stage 1 ("Eigen, dense"): the trailing matrix is
// partitioned by column blocks over the ranks (K is symmetric: a column block is the row block
// transposed); the reduced matrix W (band + reflectors), the panel QR and the thin products are
// replicated -- deterministic kernels, bitwise identical on every rank. One panel step is
//   [owner] strip = A[k:, k:k+b]  --broadcast-->  dist_s1_panel (QR, T factor)
//   dist_s1_av: rows of Y = A22 V that belong to this rank's columns  --all-gather-->  Y
//   dist_s1_update: Z from (V, Y, T), then A22[:, own columns] -= V Z[own,:]' + Z V[own,:]'
// and after the last panel the remaining columns are broadcast into W (dist_s1_put) and every rank
// calls eigen(..., EIG_RESUME) for stage 2, the divide & conquer and its slice of the back-transform.
// ---------------------------------------------------------------------------
static int dist_state(bigkrls_ctx* ctx, int64_t n, DistS1** out) {
  DistS1* ds = (DistS1*)ctx->dist_s1;
  if (!ds || ds->n != (int)n) {
    set_error("distributed stage 1: call bigkrls_dev_s1_open(ctx, n) first");
    return BIGKRLS_EINVAL;
  }
  *out = ds;
  return BIGKRLS_OK;
}

int dist_s1_open(bigkrls_ctx* ctx, int64_t n) {
#ifdef BK_FAULT_INJECT
  // BIGKRLS_FAULT=s1_open (test build, set in ONE rank's process): a local failure that the other ranks must learn of
  if (const char* f = getenv("BIGKRLS_FAULT"))
    if (std::string(f) == "s1_open") {
      set_error("injected fault: the stage-1 workspace of this rank could not be allocated");
      return BIGKRLS_ENOMEM;
    }
#endif
  return eigen(ctx, nullptr, n, n, n, nullptr, 0, -1.0, nullptr, n, nullptr, 0, 1, EIG_SETUP_ONLY);
}

int dist_s1_panel(bigkrls_ctx* ctx, int64_t n, int64_t k, const double* strip) {
  DistS1* ds = nullptr;
  BK_TRY(dist_state(ctx, n, &ds));
  BK_REQUIRE(strip && k >= 0 && k % S2_B == 0 && ds->ops.has_panel((int)k), "s1_panel: not a panel column");
  double* W = ds->ops.W;
  // rows k..n of the panel's columns: the final diagonal block and the sub-diagonal panel
  BK_TRY(copy_matrix(ctx, strip, n - k, S2_B, n - k, W + k + k * n, n));
  BK_TRY(ds->ops.panel_qr((int)k, ctx->stream));
  BK_TRY(ds->ops.build_T((int)k, ctx->stream));
  return BIGKRLS_OK;
}

// the panel factorisation and its T factor on the look-ahead stream, after everything queued on the main stream so
// far; the next dist_s1_av / dist_s1_thin (which read V and T) wait for it
int dist_s1_panel_begin(bigkrls_ctx* ctx, int64_t n, int64_t k, const double* strip) {
  DistS1* ds = nullptr;
  BK_TRY(dist_state(ctx, n, &ds));
  BK_REQUIRE(strip && k >= 0 && k % S2_B == 0 && ds->ops.has_panel((int)k), "s1_panel_begin: not a panel column");
  BK_REQUIRE(!ds->panel_pending, "s1_panel_begin: the previous panel has not been consumed");
  BK_TRY(side_stream_get(ctx));
  hipStream_t side = ctx->side_stream, st = ctx->stream;
  BK_HIP(hipEventRecord(ctx->ev_fork, st));
  BK_HIP(hipStreamWaitEvent(side, ctx->ev_fork, 0));
  double* W = ds->ops.W;
  ctx->stream = side;
  int rc = copy_matrix(ctx, strip, n - k, S2_B, n - k, W + k + k * n, n);
  ctx->stream = st;
  BK_TRY(rc);
  BK_TRY(ds->ops.panel_qr((int)k, side));
  BK_TRY(ds->ops.build_T((int)k, side));
  BK_HIP(hipEventRecord(ctx->ev_join, side));
  ds->panel_pending = true;
  return BIGKRLS_OK;
}

static int dist_s1_wait_panel(bigkrls_ctx* ctx, DistS1* ds) {
  if (ds->panel_pending) {
    BK_HIP(hipStreamWaitEvent(ctx->stream, ctx->ev_join, 0));
    ds->panel_pending = false;
  }
  return BIGKRLS_OK;
}

// This rank's contribution to Y = A22 V (m x 64): A22[:, own columns] V[own rows, :] -- the plain N,N product along the
// contiguous dimension of the column block (the transposed form A22[:, own]' V, which gives the own ROWS of Y, runs
// at half the rate). The ranks' contributions are summed by an all-reduce of m x 64 doubles.
int dist_s1_av(bigkrls_ctx* ctx, int64_t n, int64_t k, const double* Acols, int64_t lda, int64_t ncols,
               int64_t row0, double* Ypart, int64_t ldy) {
  DistS1* ds = nullptr;
  BK_TRY(dist_state(ctx, n, &ds));
  BK_TRY(dist_s1_wait_panel(ctx, ds));
  const int64_t m = n - k - S2_B;
  BK_REQUIRE(m > 0 && ncols >= 0 && Ypart && ldy >= m && (ncols == 0 || (Acols && lda >= m && row0 >= 0 && row0 + ncols <= m)),
             "s1_av: bad arguments");
  if (ncols == 0) {
    BK_HIP(hipMemsetAsync(Ypart, 0, (size_t)(ldy * S2_B) * sizeof(double), ctx->stream));
    return BIGKRLS_OK;
  }
  return gemm(ctx, 0, 0, m, S2_B, ncols, 1.0, Acols, lda, ds->ops.ws.Vp + row0, m, 0.0, Ypart, ldy);
}

// the thin products of panel k from the gathered Y = A22 V: PZ1 = [V | Z], PZ2 = [Z | V]
int dist_s1_thin(bigkrls_ctx* ctx, int64_t n, int64_t k, double* Y) {
  DistS1* ds = nullptr;
  BK_TRY(dist_state(ctx, n, &ds));
  BK_TRY(dist_s1_wait_panel(ctx, ds));
  BK_REQUIRE(Y && n - k - S2_B > 0, "s1_thin: bad arguments");
  return ds->ops.small_products((int)k, Y, nullptr);
}

// A22[:, cols] -= [V | Z] [Z | V][cols, :]' for `ncols` own columns whose first one is row `row0` of the trailing matrix
int dist_s1_update_cols(bigkrls_ctx* ctx, int64_t n, int64_t k, double* Acols, int64_t lda, int64_t ncols,
                        int64_t row0) {
  DistS1* ds = nullptr;
  BK_TRY(dist_state(ctx, n, &ds));
  const int64_t m = n - k - S2_B;
  BK_REQUIRE(m > 0 && ncols >= 0 && row0 >= 0 && row0 + ncols <= m, "s1_update_cols: bad arguments");
  if (ncols == 0) return BIGKRLS_OK;
  BK_REQUIRE(Acols && lda >= m, "s1_update_cols: bad column block");
  if (ds->panel_pending) BK_TRY(ds->ops.gate());   // (the next panel's factorisation, on the look-ahead stream, becomes resident first)
  const double *PZ1 = ds->ops.ws.PZ1, *PZ2 = ds->ops.ws.PZ2;
  // The own columns' diagonal block (rows row0 .. row0 + ncols of the trailing matrix) is symmetric: lower tiles
  // computed and mirrored (half the MFMA work of the plain product; with one rank that is the whole update);
  // the rows above and below it are plain products.
  if (ncols >= 128) {
    if (row0 > 0) BK_TRY(gemm(ctx, 0, 1, row0, ncols, 2 * S2_B, -1.0, PZ1, m, PZ2 + row0, m, 1.0, Acols, lda));
    BK_TRY(syrk_mirror(ctx, ncols, 2 * S2_B, -1.0, PZ1 + row0, m, PZ2 + row0, m, Acols + row0, lda, 0, -1, true));
    const int64_t below = m - row0 - ncols;
    if (below > 0)
      BK_TRY(gemm(ctx, 0, 1, below, ncols, 2 * S2_B, -1.0, PZ1 + row0 + ncols, m, PZ2 + row0, m, 1.0,
                  Acols + row0 + ncols, lda));
    return BIGKRLS_OK;
  }
  return gemm(ctx, 0, 1, m, ncols, 2 * S2_B, -1.0, PZ1, m, PZ2 + row0, m, 1.0, Acols, lda);
}

// ---- several panels per trailing update in the partitioned stage 1 (no pieces: the update of a group of G = 4 or 2
//      panels is applied after its last panel; panel j's product with the stale column blocks is corrected with the j
//      pending blocks of the group, as in stage1_to_band) -------------------------------------------------------------
// size of the group that may start at panel k0: 4 (trailing matrix >= 12 800 rows), 2 (>= 10 752), or 0
// what THIS rank would do (its environment, its workspace): the ranks take the minimum (dist_s1_set_agg_mode) -- the
// group size decides the sequence of collectives, so it must be the same everywhere
int dist_s1_local_agg_mode(bigkrls_ctx* ctx, int64_t n) {
  DistS1* ds = nullptr;
  if (dist_state(ctx, n, &ds) != BIGKRLS_OK || ds->ops.ws.aggPZ1[0] == nullptr || !ds->ops.fused_small) return 0;
  const char* e = getenv("BIGKRLS_S1AGG");     // read per decomposition, like the single-GPU loop
  const int env = e ? atoi(e) : -1;
  return env == 0 ? 0 : (env == 2 ? 2 : 4);
}
int dist_s1_set_agg_mode(bigkrls_ctx* ctx, int64_t n, int mode) {
  DistS1* ds = nullptr;
  BK_TRY(dist_state(ctx, n, &ds));
  ds->agg_mode = mode;
  return BIGKRLS_OK;
}
int dist_s1_group_size(bigkrls_ctx* ctx, int64_t n, int64_t k0) {
  DistS1* ds = nullptr;
  if (dist_state(ctx, n, &ds) != BIGKRLS_OK || ds->agg_mode == 0) return 0;
  const int env = ds->agg_mode == 2 ? 2 : -1;
  const int b = S2_B;
  auto panels = [&](int g) {
    for (int j = 0; j < g; ++j)
      if (!ds->ops.has_panel((int)(k0 + j * b))) return false;
    return true;
  };
  if (env != 2 && panels(4) && n - k0 - 4 * b >= S1_QUAD_MIN_M) return 4;
  if (panels(2) && n - k0 - 2 * b >= S1_AGG_MIN_M) return 2;
  return 0;
}

// the thin products of panel k = k0 + 64 j of the group that starts at k0, from the summed Y = (stale A22) V, written
// into block j of the group's [V Z ...] / [Z V ...] buffers (n x 512 each: the layout of stage1_to_band's groups of four)
int dist_s1_thin_group(bigkrls_ctx* ctx, int64_t n, int64_t k, double* Y, int64_t k0) {
  DistS1* ds = nullptr;
  BK_TRY(dist_state(ctx, n, &ds));
  BK_TRY(dist_s1_wait_panel(ctx, ds));
  const int b = S2_B;
  const int64_t m = n - k - b, off = k - k0;
  const int j = (int)(off / b);
  BK_REQUIRE(Y && m > 0 && off >= 0 && off % b == 0 && j < 4 && ds->ops.ws.aggPZ1[0], "s1_thin_group: bad arguments");
  double *PZ1 = ds->ops.ws.aggPZ1[0], *PZ2 = ds->ops.ws.aggPZ1[1];
  const int64_t ldg = ds->ops.ws.aggLd;
  if (j > 0) {        // Y -= U (U'^T V) over the j pending blocks: their update has not reached the column blocks yet
    double* C = ds->ops.ws.aggC;
    BK_TRY(gemm(ctx, 1, 0, 2 * b * j, b, m, 1.0, PZ2 + off, ldg, ds->ops.ws.Vp, m, 0.0, C, 2 * b * j));
    BK_TRY(gemm(ctx, 0, 0, m, b, 2 * b * j, -1.0, PZ1 + off, ldg, C, 2 * b * j, 1.0, Y, m));
  }
  const int64_t roff = off + (int64_t)j * 2 * b * ldg;
  return ds->ops.small_products_to((int)k, Y, nullptr, PZ1 + roff, PZ2 + roff, ldg);
}

// A22[:, cols] -= U U'[cols, :]' over the first `nblk` blocks of the group that starts at k0 (fewer than the whole group:
// for the next panel's columns only); the trailing matrix is panel k's
int dist_s1_update_cols_group(bigkrls_ctx* ctx, int64_t n, int64_t k, double* Acols, int64_t lda, int64_t ncols,
                              int64_t row0, int64_t k0, int nblk) {
  DistS1* ds = nullptr;
  BK_TRY(dist_state(ctx, n, &ds));
  const int b = S2_B;
  const int64_t m = n - k - b, off = k - k0;
  BK_REQUIRE(m > 0 && ncols >= 0 && row0 >= 0 && row0 + ncols <= m && off >= 0 && off % b == 0 && nblk >= 1 && nblk <= 4 &&
                 off / b < nblk && ds->ops.ws.aggPZ1[0], "s1_update_cols_group: bad arguments");
  if (ncols == 0) return BIGKRLS_OK;
  BK_REQUIRE(Acols && lda >= m, "s1_update_cols_group: bad column block");
  if (ds->panel_pending) BK_TRY(ds->ops.gate());   // (see dist_s1_update_cols)
  const int64_t ldg = ds->ops.ws.aggLd, kk = 2 * b * nblk;
  const double *PZ1 = ds->ops.ws.aggPZ1[0] + off, *PZ2 = ds->ops.ws.aggPZ1[1] + off;
  if (ncols >= 128) {
    if (row0 > 0) BK_TRY(gemm(ctx, 0, 1, row0, ncols, kk, -1.0, PZ1, ldg, PZ2 + row0, ldg, 1.0, Acols, lda));
    BK_TRY(syrk_mirror(ctx, ncols, kk, -1.0, PZ1 + row0, ldg, PZ2 + row0, ldg, Acols + row0, lda, 0, -1, true));
    const int64_t below = m - row0 - ncols;
    if (below > 0)
      BK_TRY(gemm(ctx, 0, 1, below, ncols, kk, -1.0, PZ1 + row0 + ncols, ldg, PZ2 + row0, ldg, 1.0, Acols + row0 + ncols, lda));
    return BIGKRLS_OK;
  }
  return gemm(ctx, 0, 1, m, ncols, kk, -1.0, PZ1, ldg, PZ2 + row0, ldg, 1.0, Acols, lda);
}

int dist_s1_update(bigkrls_ctx* ctx, int64_t n, int64_t k, double* Y, double* Acols, int64_t lda, int64_t ncols,
                   int64_t row0) {
  BK_TRY(dist_s1_thin(ctx, n, k, Y));
  return dist_s1_update_cols(ctx, n, k, Acols, lda, ncols, row0);
}

// diagnostics (BIGKRLS_TRACE_DIR): the replicated factors of panel k -- V, T, tau -- once its factorisation has been consumed
int dist_s1_trace(bigkrls_ctx* ctx, int64_t n, int64_t k) {
  if (!trace_on()) return BIGKRLS_OK;
  DistS1* ds = nullptr;
  BK_TRY(dist_state(ctx, n, &ds));
  BK_TRY(dist_s1_wait_panel(ctx, ds));
  const int64_t m = n - k - S2_B;
  BK_TRY(trace_point(ctx, ctx->stream, "R:s1_V", ds->ops.ws.Vp, m * S2_B, k));
  BK_TRY(trace_point(ctx, ctx->stream, "R:s1_T", ds->ops.Tof((int)k), S2_B * S2_B, k));
  return trace_point(ctx, ctx->stream, "R:s1_tau", ds->ops.taus1 + k, S2_B, k);
}

int dist_s1_put(bigkrls_ctx* ctx, int64_t n, int64_t k, const double* strip, int64_t ncols) {
  DistS1* ds = nullptr;
  BK_TRY(dist_state(ctx, n, &ds));
  BK_REQUIRE(strip && k >= 0 && ncols > 0 && k + ncols <= n, "s1_put: bad arguments");
  BK_TRY(dist_s1_wait_panel(ctx, ds));
  return copy_matrix(ctx, strip, n - k, ncols, n - k, ds->ops.W + k + k * n, n);
}

}  // namespace bk

#ifdef BK_BC_PROF
// profiling builds only (tools/bc_prof.py): the accumulated shader clocks of one bc_resident location per phase
extern "C" int bigkrls_debug_bc_trace(unsigned long long* out, int count) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(bk::bc_trace), (size_t)count * sizeof(unsigned long long)) == hipSuccess ? 0 : 1;
}
extern "C" int bigkrls_debug_bc_prof(long long* out8, int reset) {
  if (out8 && hipMemcpyFromSymbol(out8, HIP_SYMBOL(bk::bc_prof_acc), 8 * sizeof(long long)) != hipSuccess) return 1;
  if (out8) {   // slot 5 of the LDS-window kernel is unused: the column's send -> receipt time (10-ns ticks) travels there
    unsigned long long st[4] = {0, 0, 0, 0};
    if (hipMemcpyFromSymbol(st, HIP_SYMBOL(bk::bc_prof_stamp), sizeof st) != hipSuccess) return 1;
    if (getenv("BIGKRLS_BC_PROF_FLIGHT")) out8[5] = (long long)st[3];
  }
  if (reset) {
    long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(bk::bc_prof_acc), z, sizeof z) != hipSuccess) return 1;
    if (hipMemcpyToSymbol(HIP_SYMBOL(bk::bc_prof_stamp), z, 4 * sizeof(long long)) != hipSuccess) return 1;
  }
  return 0;
}
#endif
