// fp64 MFMA GEMM core for gfx950 (v_mfma_f64_16x16x4_f64) with pluggable epilogues.
//
// One core serves:
//   * the Gaussian kernel build  K = exp(-(|a_i|^2 + |b_j|^2 - 2 a_i.b_j)/sigma)
//     (replaces the scalar double loops of src/gauss_kernel.cpp:13-30 and
//     src/temp_kernel.cpp:13-30)  -- "NT" product with a fused norm+exp epilogue;
//   * the cross-products of src/crossprod.cpp:13-85 (A'B, A'A, AB', AA');
//   * every GEMM inside the eigensolver, the variance matrices and the
//     marginal-effects pass.
//
// Tiling (CDNA4, wave64): block tile 128 x BN x 16 with 256 threads = 4 waves in a
// 2 x 2 grid; each wave owns 64 x BN/2 as 4 x (BN/32) MFMA tiles of 16 x 16.
// Operands are staged global -> registers -> LDS (double buffered, one barrier per
// k-tile); LDS tiles are k-major with a rotate+xor swizzle so that both the
// fragment reads (ds_read_b64, 16 consecutive doubles per k row) and the
// transposing stores are bank-conflict free.
//
// MFMA operand mapping (f64 16x16x4: lane l supplies A[i=l&15][k=l>>4] and
// B[k=l>>4][j=l&15]; result reg r of lane l is D[i=(l>>4)+4r][j=l&15]):
// we feed the N-side fragment as the MFMA "A" and the M-side fragment as the MFMA
// "B", so that the lane index (l&15) runs along the column-major-contiguous M
// dimension of C and every 16 lanes store one 128-byte segment.
#include "common.h"
#include <algorithm>

namespace bk {

typedef double d4 __attribute__((ext_vector_type(4)));

constexpr int BM = 128;
constexpr int BK = 16;
constexpr int NT = 256;
#ifndef GEMM_OCC
#define GEMM_OCC 2
#endif

// LDS layouts of one W x BK operand stage (no swizzle: every index is a per-thread base plus a
// compile-time constant, so all fragment reads and staging stores use immediate offsets):
//   contiguous-x operands ("KX"):  [k][x], row stride W + GEMM_KX_PAD doubles. A half-wave of a
//       fragment read covers two k rows x 16 x; with a pad of 16 they land half the banks apart
//       (conflict-free), with 8 half of the lanes pay a 2-way conflict -- not measurable, and the
//       128 x 64 tile then needs 52 KB, so three workgroups fit a CU.
//   k-contiguous operands ("XK"):  [x][k], row stride BK + 2 = 18 doubles. A half-wave reads
//       16 x (18 lm mod 32: the 16 even banks) x 2 k: 32 distinct banks.
constexpr int LDS_XK = BK + 2;
#ifndef GEMM_KX_PAD
#define GEMM_KX_PAD 8
#endif
// plain GEMM kernels with tiles up to this width keep two k-tiles in flight (see gemm_tile)
#ifndef GEMM_DEEP_BN
#define GEMM_DEEP_BN 64
#endif
constexpr int lds_stage(int w) {   // room for either layout
  return BK * (w + GEMM_KX_PAD) > w * LDS_XK ? BK * (w + GEMM_KX_PAD) : w * LDS_XK;
}
// one layout only: the 128 x 64 tile of the trailing update (both operands x-contiguous) needs 52 KB with this and
// 54 KB with lds_stage() -- three workgroups fit the 160 KB of a CU only with the former (see syrk_mirror_kernel)
constexpr int lds_stage_of(bool contig_x, int w) { return contig_x ? BK * (w + GEMM_KX_PAD) : w * LDS_XK; }
template <bool CONTIG_X, int W>
__device__ __forceinline__ int lds_at(int k, int x) {
  return CONTIG_X ? k * (W + GEMM_KX_PAD) + x : x * LDS_XK + k;
}

// Load a W x 16 operand tile into registers. Element (x, k) lives at
// src[x*sx + k*sk]; CONTIG_X says which of the two strides is 1.
template <bool CONTIG_X, int W, bool GATHER = false>
__device__ __forceinline__ void tile_load(const double* __restrict__ src, int64_t ld, int x0,
                                          int k0, int xmax, int kmax, double (&r)[W / 16],
                                          const int* __restrict__ kidx = nullptr) {
  // Branch-free and select-free: out-of-range elements read a clamped (valid) address.
  //  * A predicated load puts every load in its own basic block and the compiler then drains
  //    vmcnt to 0 before the first MFMA of the k-tile; a select right after the load does the
  //    same. Either way the whole global latency is exposed once per k-tile.
  //  * Rows/columns past xmax only ever feed accumulator entries that are never stored.
  //  * k past kmax is zeroed by tile_zero_ktail() just before the LDS store of the last tile.
  const int t = threadIdx.x;
  if (CONTIG_X) {
    const int x = t % W;
    const int kb = t / W;
    constexpr int KS = NT / W;
    const int gx = x0 + x;
    const double* base = src + (gx < xmax ? gx : xmax - 1);
#pragma unroll
    for (int q = 0; q < W / 16; ++q) {
      const int gk = k0 + kb + q * KS;
      const int kc = gk < kmax ? gk : kmax - 1;
      const int64_t col = GATHER ? (int64_t)kidx[kc] : (int64_t)kc;
      r[q] = base[col * ld];
    }
  } else {
    const int k = t & 15;
    const int xb = t >> 4;
    const int gk = k0 + k;
    const double* base = src + (gk < kmax ? gk : kmax - 1);
#pragma unroll
    for (int q = 0; q < W / 16; ++q) {
      const int gx = x0 + xb + 16 * q;
      const int xc = gx < xmax ? gx : xmax - 1;
      r[q] = base[(int64_t)xc * ld];
    }
  }
}

// zero the staged elements whose k index lies past kmax (last, partial k-tile only)
template <bool CONTIG_X, int W>
__device__ __forceinline__ void tile_zero_ktail(int k0, int kmax, double (&r)[W / 16]) {
  const int t = threadIdx.x;
  if (CONTIG_X) {
    const int kb = t / W;
    constexpr int KS = NT / W;
#pragma unroll
    for (int q = 0; q < W / 16; ++q)
      if (k0 + kb + q * KS >= kmax) r[q] = 0.0;
  } else {
    if (k0 + (t & 15) >= kmax) {
#pragma unroll
      for (int q = 0; q < W / 16; ++q) r[q] = 0.0;
    }
  }
}

// Fast loader for interior k-tiles: p points at this thread's first element of the tile and
// the remaining elements follow at a fixed stride, so the address math is one 64-bit add each.
template <int W>
__device__ __forceinline__ void tile_load_strided(const double* __restrict__ p, int64_t stride,
                                                  double (&r)[W / 16]) {
#pragma unroll
  for (int q = 0; q < W / 16; ++q) r[q] = p[q * stride];
}

template <bool CONTIG_X, int W>
__device__ __forceinline__ void tile_store(double* __restrict__ lds, const double (&r)[W / 16]) {
  const int t = threadIdx.x;
  if (CONTIG_X) {
    const int x = t % W;
    const int kb = t / W;
    constexpr int KS = NT / W;
#pragma unroll
    for (int q = 0; q < W / 16; ++q) lds[lds_at<true, W>(kb + q * KS, x)] = r[q];
  } else {
    const int k = t & 15;
    const int xb = t >> 4;
#pragma unroll
    for (int q = 0; q < W / 16; ++q) lds[lds_at<false, W>(k, xb + 16 * q)] = r[q];
  }
}

struct GemmOperands {
  const double* A;
  const double* B;
  int64_t lda, ldb;
  int M, N, K;
  const int* kidx;  // optional gather of A's columns (NN only): op(A)(:,k) = A(:, kidx[k])
  const int* run_if = nullptr;  // optional device-side predicate (plain GEMM kernels only): nothing is done while *run_if == 0
};

// Computes the accumulators of the (m0, n0) block tile over k in [kbeg, kend).
// TA/TB: operand is used transposed (op(A) = A' with A stored K x M, etc.).
template <bool TA, bool TB, int BN, bool GATHER = false, bool DEEP = false>
__device__ __forceinline__ void gemm_tile(const GemmOperands& g, int m0, int n0, int kbeg,
                                          int kend, double* __restrict__ smem,
                                          d4 (&acc)[4][BN / 32]) {
  constexpr int NJ = BN / 32;
  // A tile: op(A)(m,k). not transposed: A[m + k lda] -> contiguous along m.
  constexpr bool A_CONTIG = !TA;
  // B tile: op(B)(k,n). transposed: B stored N x K, B[n + k ldb] -> contiguous along n.
  constexpr bool B_CONTIG = TB;
  // LDS stage offsets are kept as integers added to `smem` at each use: selecting between
  // pointers (double* As[2]) makes the compiler lose the LDS address space and emit flat loads,
  // whose completion is then tied to the outstanding global loads (vmcnt).
  constexpr int A_STAGE = lds_stage_of(A_CONTIG, BM), B_BASE = 2 * A_STAGE, B_STAGE = lds_stage_of(B_CONTIG, BN);

  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int wm = (wave & 1) * 64;
  const int wn = (wave >> 1) * (BN / 2);
  const int lm = lane & 15;
  const int lk = lane >> 4;

#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc[i][j] = (d4){0.0, 0.0, 0.0, 0.0};

  double ra[BM / 16];
  double rb[BN / 16];
  const int ntiles = (kend - kbeg + BK - 1) / BK;
  if (ntiles <= 0) return;

  // Per-thread strided pointers for the fast loader (valid for full k-tiles; an operand whose
  // non-contiguous x range leaves the matrix, or a gathered A, stays on the clamped loader).
  const int tid = threadIdx.x;
  const bool a_fast = !GATHER && (A_CONTIG || m0 + BM <= g.M);
  const bool b_fast = B_CONTIG || n0 + BN <= g.N;
  const double* pa;
  const double* pb;
  int64_t sa, sb, ia, ib;  // element stride inside a tile, pointer advance per k-tile
  if (A_CONTIG) {
    const int gx = m0 + tid % BM;
    pa = g.A + (gx < g.M ? gx : g.M - 1) + (int64_t)(kbeg + tid / BM) * g.lda;
    sa = (NT / BM) * g.lda;
    ia = BK * g.lda;
  } else {
    pa = g.A + (kbeg + (tid & 15)) + (int64_t)(a_fast ? m0 + (tid >> 4) : 0) * g.lda;
    sa = 16 * g.lda;
    ia = BK;
  }
  if (B_CONTIG) {
    const int gx = n0 + tid % BN;
    pb = g.B + (gx < g.N ? gx : g.N - 1) + (int64_t)(kbeg + tid / BN) * g.ldb;
    sb = (NT / BN) * g.ldb;
    ib = BK * g.ldb;
  } else {
    pb = g.B + (kbeg + (tid & 15)) + (int64_t)(b_fast ? n0 + (tid >> 4) : 0) * g.ldb;
    sb = 16 * g.ldb;
    ib = BK;
  }
  auto load_tiles = [&](int k0) {
    const bool full = k0 + BK <= kend;
    if (full && a_fast) tile_load_strided<BM>(pa, sa, ra);
    else tile_load<A_CONTIG, BM, GATHER>(g.A, g.lda, m0, k0, g.M, kend, ra, g.kidx);
    if (full && b_fast) tile_load_strided<BN>(pb, sb, rb);
    else tile_load<B_CONTIG, BN>(g.B, g.ldb, n0, k0, g.N, kend, rb);
    pa += ia;
    pb += ib;
  };

  auto mfma_tile = [&](int cur) {
    const double* as = smem + cur * A_STAGE;
    const double* bs = smem + B_BASE + cur * B_STAGE;
#pragma unroll
    for (int kk = 0; kk < BK; kk += 4) {
      double af[4], bf[NJ];
#pragma unroll
      for (int i = 0; i < 4; ++i) af[i] = as[lds_at<A_CONTIG, BM>(kk + lk, wm + i * 16 + lm)];
#pragma unroll
      for (int j = 0; j < NJ; ++j) bf[j] = bs[lds_at<B_CONTIG, BN>(kk + lk, wn + j * 16 + lm)];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(bf[j], af[i], acc[i][j], 0, 0, 0);
    }
  };

  if (DEEP) {
    // Two k-tiles in flight: the loads of tile t+2 are issued before the MFMAs of tile t, tile t+1
    // waits in the other register set and goes to LDS after them. A thin tile (BN = 64) has only
    // ~1.7 us of MFMA work per k-tile and CU -- less than the HBM latency a single tile in flight
    // would have to hide.
    double ra2[BM / 16];
    double rb2[BN / 16];
    auto load_into = [&](int k0, double (&xa)[BM / 16], double (&xb)[BN / 16]) {
      const bool full = k0 + BK <= kend;
      if (full && a_fast) tile_load_strided<BM>(pa, sa, xa);
      else tile_load<A_CONTIG, BM, GATHER>(g.A, g.lda, m0, k0, g.M, kend, xa, g.kidx);
      if (full && b_fast) tile_load_strided<BN>(pb, sb, xb);
      else tile_load<B_CONTIG, BN>(g.B, g.ldb, n0, k0, g.N, kend, xb);
      pa += ia;
      pb += ib;
    };
    auto stage = [&](int k0, int st, double (&xa)[BM / 16], double (&xb)[BN / 16]) {
      if (k0 + BK > kend) {
        tile_zero_ktail<A_CONTIG, BM>(k0, kend, xa);
        tile_zero_ktail<B_CONTIG, BN>(k0, kend, xb);
      }
      tile_store<A_CONTIG, BM>(smem + st * A_STAGE, xa);
      tile_store<B_CONTIG, BN>(smem + B_BASE + st * B_STAGE, xb);
    };
    load_into(kbeg, ra, rb);
    stage(kbeg, 0, ra, rb);
    if (ntiles > 1) load_into(kbeg + BK, ra2, rb2);
    __syncthreads();
    for (int t = 0; t < ntiles; t += 2) {
      // even tile t: LDS stage 0; tile t+1 waits in (ra2, rb2); tile t+2 loads into (ra, rb)
      if (t + 2 < ntiles) load_into(kbeg + (t + 2) * BK, ra, rb);
      mfma_tile(0);
      if (t + 1 < ntiles) stage(kbeg + (t + 1) * BK, 1, ra2, rb2);
      __syncthreads();
      if (t + 1 >= ntiles) break;
      // odd tile t+1: LDS stage 1; tile t+2 waits in (ra, rb); tile t+3 loads into (ra2, rb2)
      if (t + 3 < ntiles) load_into(kbeg + (t + 3) * BK, ra2, rb2);
      mfma_tile(1);
      if (t + 2 < ntiles) stage(kbeg + (t + 2) * BK, 0, ra, rb);
      __syncthreads();
    }
    return;
  }

  load_tiles(kbeg);
  if (kbeg + BK > kend) {
    tile_zero_ktail<A_CONTIG, BM>(kbeg, kend, ra);
    tile_zero_ktail<B_CONTIG, BN>(kbeg, kend, rb);
  }
  tile_store<A_CONTIG, BM>(smem, ra);
  tile_store<B_CONTIG, BN>(smem + B_BASE, rb);
  __syncthreads();

  for (int t = 0; t < ntiles; ++t) {
    const int cur = t & 1;
    const int k0 = kbeg + (t + 1) * BK;
    if (t + 1 < ntiles) load_tiles(k0);
    mfma_tile(cur);
    if (t + 1 < ntiles) {
      if (k0 + BK > kend) {  // partial last tile: k rows past the end must contribute zeros
        tile_zero_ktail<A_CONTIG, BM>(k0, kend, ra);
        tile_zero_ktail<B_CONTIG, BN>(k0, kend, rb);
      }
      tile_store<A_CONTIG, BM>(smem + (cur ^ 1) * A_STAGE, ra);
      tile_store<B_CONTIG, BN>(smem + B_BASE + (cur ^ 1) * B_STAGE, rb);
    }
    __syncthreads();
  }
}

// Visit every accumulator element owned by this lane: f(m, n, value).
template <int BN, class F>
__device__ __forceinline__ void acc_foreach(const d4 (&acc)[4][BN / 32], int m0, int n0, F f) {
  constexpr int NJ = BN / 32;
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int wm = (wave & 1) * 64;
  const int wn = (wave >> 1) * (BN / 2);
  const int lm = lane & 15;
  const int lk = lane >> 4;
#pragma unroll
  for (int j = 0; j < NJ; ++j)
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int m = m0 + wm + i * 16 + lm;
        const int n = n0 + wn + j * 16 + lk + 4 * r;
        f(m, n, acc[i][j][r]);
      }
}

constexpr size_t smem_bytes(int bn) { return (size_t)(2 * lds_stage(BM) + 2 * lds_stage(bn)) * sizeof(double); }
constexpr size_t smem_bytes_nt(int bn) {   // op(A) = A, op(B) = B': both tiles x-contiguous
  return (size_t)(2 * lds_stage_of(true, BM) + 2 * lds_stage_of(true, bn)) * sizeof(double);
}

// XCD-aware tile order: consecutive block ids round-robin over the 8 XCDs, so give
// each XCD a contiguous run of tiles (neighbouring tiles share operand panels in
// that XCD's L2). Bijective for any grid size.
__device__ __forceinline__ int xcd_remap(int bid, int nblocks) {
  const int nx = 8;
  const int q = nblocks / nx, rem = nblocks % nx;
  const int x = bid % nx, i = bid / nx;
  // blocks of xcd x: q + (x < rem) of them
  const int start = x * q + (x < rem ? x : rem);
  return start + i;
}

// ---------------------------------------------------------------------------
// plain GEMM kernels
// ---------------------------------------------------------------------------
// workgroups per CU of gemm_kernel<TA, TB, BN>: the N,N kernel with 64-wide tiles (A22 V of stage 1, K B_j of the
// block Lanczos) fits three with the layout-exact LDS stages when it gives up the second k-tile in flight
#ifndef GEMM_NN64_OCC
#define GEMM_NN64_OCC 2
#endif
template <bool TA, bool TB, int BN>
constexpr int gemm_occ() { return (!TA && !TB && BN == 64) ? GEMM_NN64_OCC : GEMM_OCC; }
template <bool TA, bool TB, int BN>
constexpr size_t gemm_smem_bytes() {
  return gemm_occ<TA, TB, BN>() > GEMM_OCC
             ? (size_t)(2 * lds_stage_of(!TA, BM) + 2 * lds_stage_of(TB, BN)) * sizeof(double)
             : smem_bytes(BN);
}
template <bool TA, bool TB, int BN>
__global__ __launch_bounds__(NT, (gemm_occ<TA, TB, BN>())) void gemm_kernel(GemmOperands g, double alpha, double beta,
                                                  double* __restrict__ C, int64_t ldc,
                                                  int tiles_m, int tiles_n, int k_chunk,
                                                  double* __restrict__ partial) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  if (g.run_if != nullptr && *g.run_if == 0) return;      // (uniform: a scalar load)
  const int ntile = tiles_m * tiles_n;
  const int tid = xcd_remap(blockIdx.x, ntile);
  const int tm = tid % tiles_m, tn = tid / tiles_m;
  const int m0 = tm * BM, n0 = tn * BN;
  const int z = blockIdx.y;
  const int kbeg = z * k_chunk;
  const int kend = min(g.K, kbeg + k_chunk);
  d4 acc[4][BN / 32];
  gemm_tile<TA, TB, BN, false, (BN <= GEMM_DEEP_BN && gemm_occ<TA, TB, BN>() <= GEMM_OCC)>(g, m0, n0, kbeg, kend, smem, acc);
  if (partial != nullptr) {
    double* P = partial + (int64_t)z * g.M * g.N;
    const int M = g.M, N = g.N;
    acc_foreach<BN>(acc, m0, n0, [&](int m, int n, double v) {
      if (m < M && n < N) P[(int64_t)m + (int64_t)n * M] = v;
    });
  } else {
    const int M = g.M, N = g.N;
    if (beta == 0.0) {
      acc_foreach<BN>(acc, m0, n0, [&](int m, int n, double v) {
        if (m < M && n < N) C[(int64_t)m + (int64_t)n * ldc] = alpha * v;
      });
    } else {
      // read-modify-write: fetch every C value of the lane in one batch, then combine and store
      constexpr int NJ = BN / 32;
      const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
      const int wm = (wave & 1) * 64, wn = (wave >> 1) * (BN / 2);
      const int lm = lane & 15, lk = lane >> 4;
      // one 64 x 16 column strip at a time, software pipelined: the loads of strip j+1 are
      // issued before the stores of strip j (vmcnt is in-order over loads and stores, so loads
      // issued after a strip's stores would also wait for those stores to retire)
      auto load_strip = [&](int j, d4 (&cold)[4]) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int m = m0 + wm + i * 16 + lm, n = n0 + wn + j * 16 + lk + 4 * r;
            cold[i][r] = (m < M && n < N) ? C[(int64_t)m + (int64_t)n * ldc] : 0.0;
          }
      };
      d4 cold[2][4];
      load_strip(0, cold[0]);
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        if (j + 1 < NJ) load_strip(j + 1, cold[(j + 1) & 1]);
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int m = m0 + wm + i * 16 + lm, n = n0 + wn + j * 16 + lk + 4 * r;
            if (m < M && n < N)
              C[(int64_t)m + (int64_t)n * ldc] = alpha * acc[i][j][r] + beta * cold[j & 1][i][r];
          }
      }
    }
  }
}

__global__ void splitk_reduce_kernel(const double* __restrict__ partial, int splits, int M, int N,
                                     double alpha, double beta, double* __restrict__ C,
                                     int64_t ldc, const int* __restrict__ run_if) {
  if (run_if != nullptr && *run_if == 0) return;
  // slabs are summed in the fixed order z = 0, 1, ... (deterministic); the loads of four slabs are
  // issued together -- with one load in flight per thread the kernel is latency-bound
  const int64_t total = (int64_t)M * N;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (int64_t)gridDim.x * blockDim.x) {
    const double* p = partial + e;
    double s = 0.0;
    int z = 0;
    for (; z + 4 <= splits; z += 4) {
      const double a0 = p[(int64_t)(z + 0) * total], a1 = p[(int64_t)(z + 1) * total];
      const double a2 = p[(int64_t)(z + 2) * total], a3 = p[(int64_t)(z + 3) * total];
      s += a0; s += a1; s += a2; s += a3;
    }
    for (; z < splits; ++z) s += p[(int64_t)z * total];
    const int m = (int)(e % M), n = (int)(e / M);
    const int64_t o = (int64_t)m + (int64_t)n * ldc;
    C[o] = (beta == 0.0) ? alpha * s : alpha * s + beta * C[o];
  }
}

template <bool TA, bool TB, int BN>
static int launch_gemm(bigkrls_ctx* ctx, const GemmOperands& g, double alpha, double beta,
                       double* C, int64_t ldc) {
  const int tiles_m = (g.M + BM - 1) / BM;
  const int tiles_n = (g.N + BN - 1) / BN;
  const int ntile = tiles_m * tiles_n;
  // split-K when the tile grid does not fill whole rounds of the GPU and K is long
  // The split count is chosen against the 512 workgroups the GPU holds at once (two per CU):
  // ntile * splits workgroups take ceil(ntile * splits / 512) rounds of K / splits k-steps each, plus
  // a per-split cost (prologue, partial store, reduction). A count that spills a few workgroups into
  // one more round pays that whole round: m = 20 000, n = 64 (157 tiles): 7 splits 1.02 ms, 6 or 3
  // splits 0.95 ms; m = 10 000 (79 tiles): 13 splits 0.30 ms, 6 splits 0.26 ms.
  int splits = 1;
  // (also above one round: 782 tiles -- N = 100 000 times a 128-column block -- fill 1.53 rounds unsplit,
  //  i.e. run at 76 %; five splits fill 7.6 of 8)
  if (ntile < 4096 && g.K >= 1024) {
    const int maxs = std::min(64, std::max(1, g.K / 256));
    // rough time model in us: a k-step of a workgroup ~0.06 us per unit of K; per split the partial
    // slab is written and read again (16 bytes per output element at ~5 TB/s) plus a fixed ~0.2 us
    // (measured for the block-Lanczos product, 391 tiles x K = 50 000: 5 splits 10.18 ms, 9 or 13 splits 9.98 ms,
    //  tools/kb_shape_probe.hip -- for these long products the slab traffic costs half of what the model charged)
    const double per_split = ((ntile >= 256 && g.K >= 16384) ? 1.6e-6 : 3.2e-6) * (double)g.M * (double)g.N + 0.2;
    double best = 1e30;
    constexpr int resident = 256 * gemm_occ<TA, TB, BN>();
    for (int sp = 1; sp <= maxs; ++sp) {
      const int rounds = (ntile * sp + resident - 1) / resident;
      const double cost = 0.06 * rounds * ((double)g.K / sp) + per_split * sp;
      if (cost < best - 1e-9) { best = cost; splits = sp; }
    }
  }
  {
    // (development: BIGKRLS_GEMM_SPLITS=<n> overrides the count for products with K >= 16384, tools/kb_shape_probe.hip)
    static const int env_splits = [] { const char* e = getenv("BIGKRLS_GEMM_SPLITS"); return e ? atoi(e) : 0; }();
    if (env_splits > 0 && g.K >= 16384) splits = env_splits;
  }
  int k_chunk = ((g.K + splits - 1) / splits + BK - 1) / BK * BK;
  if (k_chunk < BK) k_chunk = BK;
  splits = (g.K + k_chunk - 1) / k_chunk;
  if (splits < 1) splits = 1;
  double* partial = nullptr;
  if (splits > 1) {
    void* p = nullptr;
    // GEMMs issued on the look-ahead stream run concurrently with main-stream GEMMs: own partial buffer
    const int slot = (ctx->side_stream && (ctx->stream == ctx->side_stream || ctx->stream == ctx->bg_stream)) ? SLOT_SIDE_SPLITK : SLOT_GEMM_SPLITK;
    BK_TRY(ws_get(ctx, slot, (int64_t)splits * g.M * g.N * sizeof(double), &p));
    partial = (double*)p;
  }
  auto kern = gemm_kernel<TA, TB, BN>;
  constexpr size_t smem = gemm_smem_bytes<TA, TB, BN>();
  BK_TRY(ensure_dyn_smem(ctx, (const void*)kern, smem));
  dim3 grid(ntile, splits);
  hipLaunchKernelGGL(kern, grid, dim3(NT), smem, ctx->stream, g, alpha, beta, C, ldc,
                     tiles_m, tiles_n, k_chunk, partial);
  BK_CHECK_LAUNCH();
  if (splits > 1) {
    const int64_t total = (int64_t)g.M * g.N;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, ctx->stream, partial,
                       splits, g.M, g.N, alpha, beta, C, ldc, g.run_if);
    BK_CHECK_LAUNCH();
  }
  return BIGKRLS_OK;
}

template <int BN>
static int dispatch_trans(bigkrls_ctx* ctx, int ta, int tb, const GemmOperands& g, double alpha,
                          double beta, double* C, int64_t ldc) {
  if (!ta && !tb) return launch_gemm<false, false, BN>(ctx, g, alpha, beta, C, ldc);
  if (!ta && tb) return launch_gemm<false, true, BN>(ctx, g, alpha, beta, C, ldc);
  if (ta && !tb) return launch_gemm<true, false, BN>(ctx, g, alpha, beta, C, ldc);
  return launch_gemm<true, true, BN>(ctx, g, alpha, beta, C, ldc);
}

__global__ void scale_matrix_kernel(double* C, int64_t ldc, int M, int N, double beta) {
  const int64_t total = (int64_t)M * N;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t o = (e % M) + (e / M) * ldc;
    C[o] = (beta == 0.0) ? 0.0 : beta * C[o];
  }
}

int gemm(bigkrls_ctx* ctx, int ta, int tb, int64_t m, int64_t n, int64_t k, double alpha,
         const double* A, int64_t lda, const double* B, int64_t ldb, double beta, double* C,
         int64_t ldc) {
  BK_REQUIRE(m >= 0 && n >= 0 && k >= 0, "gemm: negative dimension");
  BK_REQUIRE(m < (1ll << 31) && n < (1ll << 31) && k < (1ll << 31), "gemm: dimension too large");
  if (m == 0 || n == 0) return BIGKRLS_OK;
  if (k == 0 || alpha == 0.0) {
    if (beta == 1.0) return BIGKRLS_OK;
    int blocks = (int)std::min<int64_t>((m * n + 255) / 256, 2048);
    hipLaunchKernelGGL(scale_matrix_kernel, dim3(blocks), dim3(256), 0, ctx->stream, C, ldc,
                       (int)m, (int)n, beta);
    BK_CHECK_LAUNCH();
    return BIGKRLS_OK;
  }
  BK_REQUIRE(A && B && C, "gemm: null pointer");
  GemmOperands g{A, B, lda, ldb, (int)m, (int)n, (int)k, nullptr, ctx->gemm_run_if};
  if (n <= 32) return dispatch_trans<32>(ctx, ta, tb, g, alpha, beta, C, ldc);
  if (n <= 64) return dispatch_trans<64>(ctx, ta, tb, g, alpha, beta, C, ldc);
  return dispatch_trans<128>(ctx, ta, tb, g, alpha, beta, C, ldc);
}

// ---------------------------------------------------------------------------
// Skinny N,N product C (M x N, N <= 48) = A (M x K, m-contiguous) B (K x N): the marginal-effects pass
// K [1, c, x_j, x_j o c ...] (csrc/deriv.hip), whose 2 + 2P operand columns (42 at P = 20) would pay for 64 in the
// 128 x 64 tile of gemm_kernel. Tile 128 x 48 x 16, four waves of 32 rows x 48 columns (2 x 3 MFMA tiles each), two
// k-tiles in flight, split-K into slabs summed by splitk_reduce_kernel in slab order (deterministic).
// ---------------------------------------------------------------------------
constexpr int SK_BN = 48;
constexpr size_t sk48_smem_bytes() { return (size_t)(2 * lds_stage_of(true, BM) + 2 * lds_stage_of(false, SK_BN)) * sizeof(double); }
__global__ __launch_bounds__(NT, 2) void gemm_nn48_kernel(GemmOperands g, int tiles_m, int k_chunk,
                                                         double* __restrict__ partial) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  constexpr int A_STAGE = lds_stage_of(true, BM), B_BASE = 2 * A_STAGE, B_STAGE = lds_stage_of(false, SK_BN);
  constexpr int NJ = SK_BN / 16;
  const int tm = xcd_remap(blockIdx.x, tiles_m);
  const int m0 = tm * BM;
  const int z = blockIdx.y;
  const int kbeg = z * k_chunk, kend = min(g.K, kbeg + k_chunk);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave * 32, lm = lane & 15, lk = lane >> 4;
  d4 acc[2][NJ];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc[i][j] = (d4){0.0, 0.0, 0.0, 0.0};
  const int ntiles = (kend - kbeg + BK - 1) / BK;
  const int tid = threadIdx.x;
  const bool b_fast = g.N >= SK_BN;
  const double* pa;
  const double* pb;
  {
    const int gx = m0 + tid % BM;
    pa = g.A + (gx < g.M ? gx : g.M - 1) + (int64_t)(kbeg + tid / BM) * g.lda;
    pb = g.B + (kbeg + (tid & 15)) + (int64_t)(b_fast ? (tid >> 4) : 0) * g.ldb;
  }
  const int64_t sa = (NT / BM) * g.lda, ia = BK * g.lda, sb = 16 * g.ldb;
  double ra[BM / 16], rb[SK_BN / 16], ra2[BM / 16], rb2[SK_BN / 16];
  auto load_into = [&](int k0, double (&xa)[BM / 16], double (&xb)[SK_BN / 16]) {
    const bool full = k0 + BK <= kend;
    if (full) tile_load_strided<BM>(pa, sa, xa);
    else tile_load<true, BM>(g.A, g.lda, m0, k0, g.M, kend, xa);
    if (full && b_fast) tile_load_strided<SK_BN>(pb, sb, xb);
    else tile_load<false, SK_BN>(g.B, g.ldb, 0, k0, g.N, kend, xb);
    pa += ia;
    pb += BK;
  };
  auto stage = [&](int k0, int st, double (&xa)[BM / 16], double (&xb)[SK_BN / 16]) {
    if (k0 + BK > kend) {
      tile_zero_ktail<true, BM>(k0, kend, xa);
      tile_zero_ktail<false, SK_BN>(k0, kend, xb);
    }
    tile_store<true, BM>(smem + st * A_STAGE, xa);
    tile_store<false, SK_BN>(smem + B_BASE + st * B_STAGE, xb);
  };
  auto mfma_tile = [&](int cur) {
    const double* as = smem + cur * A_STAGE;
    const double* bs = smem + B_BASE + cur * B_STAGE;
#pragma unroll
    for (int kk = 0; kk < BK; kk += 4) {
      double af[2], bf[NJ];
#pragma unroll
      for (int i = 0; i < 2; ++i) af[i] = as[lds_at<true, BM>(kk + lk, wm + i * 16 + lm)];
#pragma unroll
      for (int j = 0; j < NJ; ++j) bf[j] = bs[lds_at<false, SK_BN>(kk + lk, j * 16 + lm)];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(bf[j], af[i], acc[i][j], 0, 0, 0);
    }
  };
  if (ntiles > 0) {
    load_into(kbeg, ra, rb);
    stage(kbeg, 0, ra, rb);
    if (ntiles > 1) load_into(kbeg + BK, ra2, rb2);
    __syncthreads();
    for (int t = 0; t < ntiles; t += 2) {
      if (t + 2 < ntiles) load_into(kbeg + (t + 2) * BK, ra, rb);
      mfma_tile(0);
      if (t + 1 < ntiles) stage(kbeg + (t + 1) * BK, 1, ra2, rb2);
      __syncthreads();
      if (t + 1 >= ntiles) break;
      if (t + 3 < ntiles) load_into(kbeg + (t + 3) * BK, ra2, rb2);
      mfma_tile(1);
      if (t + 2 < ntiles) stage(kbeg + (t + 2) * BK, 0, ra, rb);
      __syncthreads();
    }
  }
  double* P = partial + (int64_t)z * g.M * g.N;
  const int M = g.M, N = g.N;
#pragma unroll
  for (int j = 0; j < NJ; ++j)
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int m = m0 + wm + i * 16 + lm, n = j * 16 + lk + 4 * r;
        if (m < M && n < N) P[(int64_t)m + (int64_t)n * M] = acc[i][j][r];
      }
}

int gemm_nn_skinny48(bigkrls_ctx* ctx, int64_t m, int64_t n, int64_t k, const double* A, int64_t lda, const double* B,
                     int64_t ldb, double* C, int64_t ldc) {
  BK_REQUIRE(m > 0 && n > 0 && n <= SK_BN && k > 0 && m < (1ll << 31) && k < (1ll << 31) && A && B && C, "gemm_nn_skinny48: bad arguments");
  GemmOperands g{A, B, lda, ldb, (int)m, (int)n, (int)k, nullptr};
  const int tiles_m = (int)((m + BM - 1) / BM);
  // two workgroups per CU: split K so that the grid fills whole rounds of the 512 slots (or as much of one as K allows)
  constexpr int resident = 256 * 2;
  int splits = 1;
  {
    const int maxs = (int)std::min<int64_t>(64, std::max<int64_t>(1, k / 256));
    const double per_split = 3.2e-6 * (double)m * (double)n + 0.2;
    double best = 1e30;
    for (int sp = 1; sp <= maxs; ++sp) {
      const int rounds = (tiles_m * sp + resident - 1) / resident;
      const double cost = 0.045 * rounds * ((double)k / sp) + per_split * sp;
      if (cost < best - 1e-9) { best = cost; splits = sp; }
    }
  }
  int k_chunk = (int)(((k + splits - 1) / splits + BK - 1) / BK * BK);
  splits = (int)((k + k_chunk - 1) / k_chunk);
  void* p = nullptr;
  const int slot = (ctx->side_stream && (ctx->stream == ctx->side_stream || ctx->stream == ctx->bg_stream)) ? SLOT_SIDE_SPLITK : SLOT_GEMM_SPLITK;
  BK_TRY(ws_get(ctx, slot, (int64_t)splits * m * n * sizeof(double), &p));
  BK_TRY(ensure_dyn_smem(ctx, (const void*)gemm_nn48_kernel, sk48_smem_bytes()));
  hipLaunchKernelGGL(gemm_nn48_kernel, dim3(tiles_m, splits), dim3(NT), sk48_smem_bytes(), ctx->stream, g, tiles_m, k_chunk,
                     (double*)p);
  BK_CHECK_LAUNCH();
  int blocks = (int)std::min<int64_t>((m * n + 255) / 256, 2048);
  hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, ctx->stream, (const double*)p, splits, (int)m, (int)n,
                     1.0, 0.0, C, ldc, (const int*)nullptr);
  BK_CHECK_LAUNCH();
  return BIGKRLS_OK;
}

// ---------------------------------------------------------------------------
// symmetric rank-k update of the lower triangle: C(lower) += alpha * A B'
// (A, B are m x k; used by the tridiagonalisation's trailing update
//  A22 -= [V W][W V]', whose consumers only ever read the lower triangle).
// Only tiles with tile_row >= tile_col are launched.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(NT, GEMM_OCC) void syrk_lower_kernel(GemmOperands g, double alpha,
                                                        double* __restrict__ C, int64_t ldc,
                                                        int tiles) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  // blockIdx.x enumerates the lower-triangular tiles column by column
  const int t = blockIdx.x;
  // tn = largest integer with tn*tiles - tn(tn-1)/2 <= t
  int tn = (int)((2.0 * tiles + 1.0 - sqrt((2.0 * tiles + 1.0) * (2.0 * tiles + 1.0) - 8.0 * t)) * 0.5);
  while (tn > 0 && tn * tiles - tn * (tn - 1) / 2 > t) --tn;
  while ((tn + 1) * tiles - (tn + 1) * tn / 2 <= t) ++tn;
  const int tm = tn + (t - (tn * tiles - tn * (tn - 1) / 2));
  const int m0 = tm * BM, n0 = tn * 128;
  d4 acc[4][4];
  gemm_tile<false, true, 128>(g, m0, n0, 0, g.K, smem, acc);
  const int M = g.M;
  acc_foreach<128>(acc, m0, n0, [&](int m, int n, double v) {
    if (m < M && n < M) {
      const int64_t o = (int64_t)m + (int64_t)n * ldc;
      C[o] += alpha * v;
    }
  });
}

// Same lower-triangular tile set, but the updated values are also mirrored into the upper
// triangle (through a per-wave 16 x 16 LDS transpose so that the mirrored stores are 128-byte
// segments too): C stays a fully stored, exactly symmetric matrix at half the MFMA work of a
// full GEMM update. Used by the band reduction, whose next step is the plain product A22 V.
// ACCUM = false: C = alpha A B' (nothing of C is read): the N x N variance matrices Q diag(w) Q' of the fit.
// Tile order of the mirrored rank-k update. R == 0: column by column from tile t_off (consecutive workgroups =
// consecutive XCDs take consecutive row tiles of one tile column, so every XCD streams the whole A operand through
// its L2 once per tile column). R > 0: XCD-aware, L2-blocked: the tiles of the BN-wide columns [c0, c1) are ordered
// band by band (a band = R consecutive 128-row tiles), inside a band column by column, and every XCD takes a
// contiguous eighth of that sequence (xcd_remap): an XCD keeps the R row panels of A of its band in its L2 and
// walks the columns, so each panel of B is fetched once per band instead of each panel of A once per column.
#ifdef BK_SYRK_TRACE
// development build only (tools/syrk_trace.sh): per-workgroup time stamps of the trailing update (100 MHz clock)
__device__ unsigned long long* g_syrk_trace = nullptr;
extern "C" __attribute__((visibility("default"))) int bk_syrk_trace_set(unsigned long long* p) {
  return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_syrk_trace), &p, sizeof(p));
}
#define BK_TRACE_STAMP(slot)                                                                      \
  if (g_syrk_trace && threadIdx.x == 0) {                                                         \
    g_syrk_trace[(size_t)blockIdx.x * 4 + (slot)] = wall_clock64();                               \
    if ((slot) == 0)                                                                              \
      g_syrk_trace[(size_t)blockIdx.x * 4 + 3] =                                                  \
          ((unsigned long long)__builtin_amdgcn_s_getreg(63508) << 32) | __builtin_amdgcn_s_getreg(63492); \
  }
#else
#define BK_TRACE_STAMP(slot)
#endif

struct SyrkMap {
  int tiles;            // 128-row tiles of the matrix
  int c0, c1;           // BN-wide tile columns of this launch
  int R, nband, band0;  // rows per band, number of bands, first band (absolute index)
  int t_off;            // R == 0 only
  int band_start[160];  // band b (relative) starts at sequence index band_start[b]; [nband] = number of tiles
};

#ifndef SYRK64_OCC
#define SYRK64_OCC 3
#endif
#ifndef SYRK64_DEEP
#define SYRK64_DEEP 0
#endif
#ifndef SYRK64_LDS_EXACT
#define SYRK64_LDS_EXACT 1
#endif
template <int BN, bool ACCUM = true>
__global__ __launch_bounds__(NT, (BN == 64 ? SYRK64_OCC : GEMM_OCC)) void syrk_mirror_kernel(
    GemmOperands g, double alpha, double* __restrict__ C, int64_t ldc, SyrkMap map) {
  // Tiles are 128 x BN. BN = 64 halves the accumulators; its 154 VGPRs let THREE workgroups share a CU with the
  // layout-exact 52 KB of LDS (SYRK64_LDS_EXACT = 1, the default since round 6): 5-7% faster than two in isolation
  // (m = 20 000: k = 128 1 817 -> 1 693 us, k = 256 2 404 -> 2 286). The panel factorisation of the next panel runs
  // beside this kernel, its workgroups need 107 KB of LDS and 256 VGPRs, and with three of these per CU a finished one
  // leaves a hole they do not fit into: the update is therefore launched behind a gate that lets the factorisation
  // become resident first (s1_gate, csrc/eigen_2stage.inc). Without the gate three per CU cost 13-42 ms at C3
  // (profiles/r03/r03m_*, r06/r06b_gate_ab_C3.log: 0.4221 vs 0.4087 s); -DSYRK64_LDS_EXACT=0 builds two per CU.
  extern __shared__ __attribute__((aligned(16))) double smem[];
  constexpr int NJ = BN / 32;
  constexpr int CPT = 128 / BN;   // tile columns per 128-wide column pair
  const int tiles = map.tiles;
  int tc, tm, p;
  if (map.R == 0) {
    const int t = blockIdx.x + map.t_off;
    // lower-triangular tiles, column by column; columns come in groups of CPT that share their
    // first row tile p: group p has CPT * (tiles - p) tiles and starts at CPT * (p tiles - p(p-1)/2)
    p = (int)((2.0 * tiles + 1.0 - sqrt((2.0 * tiles + 1.0) * (2.0 * tiles + 1.0) - 8.0 * t / CPT)) * 0.5);
    auto gstart = [&](int q) { return CPT * (q * tiles - q * (q - 1) / 2); };
    while (p > 0 && gstart(p) > t) --p;
    while (gstart(p + 1) <= t) ++p;
    const int rem = t - gstart(p);
    tc = CPT * p + rem / (tiles - p);          // tile column (BN wide)
    tm = p + rem % (tiles - p);                // tile row (128 high)
  } else {
    const int sq = xcd_remap(blockIdx.x, gridDim.x);
    int lo = 0, hi = map.nband;                // band_start[lo] <= sq < band_start[hi]
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (map.band_start[mid] <= sq) lo = mid; else hi = mid;
    }
    int sp = sq - map.band_start[lo];
    const int r0 = (map.band0 + lo) * map.R, r1 = min(r0 + map.R, tiles);
    // columns whose diagonal tile lies at or above r0 hold all r1 - r0 rows of the band ...
    const int cfull = max(min(map.c1, CPT * (r0 + 1)), map.c0);
    const int nfull = (cfull - map.c0) * (r1 - r0);
    if (sp < nfull) {
      tc = map.c0 + sp / (r1 - r0);
      tm = r0 + sp % (r1 - r0);
    } else {   // ... the columns through the band's diagonal corner hold the rows from their diagonal tile down
      sp -= nfull;
      tc = cfull;
      while (sp >= r1 - tc / CPT) { sp -= r1 - tc / CPT; ++tc; }
      tm = tc / CPT + sp;
    }
    p = tc / CPT;
  }
  const int m0 = tm * BM, n0 = tc * BN;
  d4 acc[4][NJ];
  BK_TRACE_STAMP(0)
  gemm_tile<false, true, BN, false, (BN == 64 && SYRK64_DEEP)>(g, m0, n0, 0, g.K, smem, acc);  // ends with a block barrier
  BK_TRACE_STAMP(1)
  const int M = g.M;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = (wave & 1) * 64, wn = (wave >> 1) * (BN / 2);
  const int lm = lane & 15, lk = lane >> 4;
  double* buf = smem + wave * (16 * 17);
  const bool diag_tile = (tm == p);
  if (!diag_tile && m0 + BM <= M) {
    // Interior tile (all 128 rows inside, strictly below the diagonal): no predicates. A predicated
    // load or store sits in its own basic block and the compiler waits for memory before each of
    // them, which serialises the whole read-modify-write (16 dependent HBM round trips per strip).
    const double* cb = C + (int64_t)(m0 + wm + lm) + (int64_t)(n0 + wn + lk) * ldc;   // lane's element of sub-tile (0, 0)
    double* cw = C + (int64_t)(m0 + wm + lm) + (int64_t)(n0 + wn + lk) * ldc;
    double* mw = C + (int64_t)(n0 + wn + lm) + (int64_t)(m0 + wm + lk) * ldc;         // its mirror image
    auto load_int = [&](int j, d4 (&cold)[4]) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) cold[i][r] = ACCUM ? cb[i * 16 + (int64_t)(j * 16 + 4 * r) * ldc] : 0.0;
    };
    d4 cbuf[2][4];
    load_int(0, cbuf[0]);
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      if (j + 1 < NJ) load_int(j + 1, cbuf[(j + 1) & 1]);
      d4 (&cold)[4] = cbuf[j & 1];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        double vv[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          vv[r] = cold[i][r] + alpha * acc[i][j][r];
          cw[i * 16 + (int64_t)(j * 16 + 4 * r) * ldc] = vv[r];
          buf[(lk + 4 * r) * 17 + lm] = vv[r];   // buf[n_local][m_local]
        }
        double tv[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) tv[r] = buf[lm * 17 + (lk + 4 * r)];
#pragma unroll
        for (int r = 0; r < 4; ++r) mw[j * 16 + (int64_t)(i * 16 + 4 * r) * ldc] = tv[r];
      }
    }
    BK_TRACE_STAMP(2)
    return;
  }
  // C is fetched one 64 x 16 column strip at a time, one strip ahead of the stores (vmcnt is
  // in-order over loads and stores: loads issued after a strip's stores would wait for them).
  auto load_strip = [&](int j, d4 (&cold)[4]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int m = m0 + wm + i * 16 + lm;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = n0 + wn + j * 16 + lk + 4 * r;
        cold[i][r] = (ACCUM && m < M && n < M && m >= n) ? C[(int64_t)m + (int64_t)n * ldc] : 0.0;
      }
    }
  };
  d4 coldbuf[2][4];
  load_strip(0, coldbuf[0]);
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    if (j + 1 < NJ) load_strip(j + 1, coldbuf[(j + 1) & 1]);
    d4 (&cold)[4] = coldbuf[j & 1];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int mt0 = m0 + wm + i * 16, nt0 = n0 + wn + j * 16;  // 16 x 16 sub-tile origin
      if (diag_tile && mt0 + 15 < nt0) continue;                  // entirely above the diagonal
      const int m = mt0 + lm;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = nt0 + lk + 4 * r;
        double v = 0.0;
        if (m < M && n < M && m >= n) {
          const int64_t o = (int64_t)m + (int64_t)n * ldc;
          v = cold[i][r] + alpha * acc[i][j][r];
          C[o] = v;
        }
        buf[(lk + 4 * r) * 17 + lm] = v;   // buf[n_local][m_local]
      }
      // mirrored store: lane' -> (n' = nt0 + lm, m' = mt0 + lk + 4 r')
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int mm = mt0 + lk + 4 * r, nn = nt0 + lm;
        const double v = buf[lm * 17 + (lk + 4 * r)];
        if (mm < M && nn < M && mm > nn) C[(int64_t)nn + (int64_t)mm * ldc] = v;
      }
    }
  }
  BK_TRACE_STAMP(2)
}

// Host side of SyrkMap: the BN-wide columns [c0, c1) of the lower tile triangle (tile rows from the diagonal down).
// rows_per_band == 0: the column-by-column order (development switch BIGKRLS_SYRK_ORDER=cols).
template <int SBN>
static int64_t syrk_map_build(SyrkMap& mp, int tiles, int c0, int c1, int k) {
  constexpr int CPT = 128 / SBN;
  mp.tiles = tiles; mp.c0 = c0; mp.c1 = c1; mp.t_off = 0; mp.nband = 0; mp.band0 = 0;
  static const int order_cols = [] { const char* e = getenv("BIGKRLS_SYRK_ORDER"); return e && std::string(e) == "cols"; }();
  static const int r_env = [] { const char* e = getenv("BIGKRLS_SYRK_R"); return e ? atoi(e) : 0; }();
  auto col_first = [&](int64_t c) -> int64_t {   // column-major index of the first tile of BN-wide column c
    const int64_t q = c / CPT;
    if (q >= tiles) return CPT * ((int64_t)tiles * tiles - (int64_t)tiles * (tiles - 1) / 2);
    return CPT * (q * tiles - q * (q - 1) / 2) + (c % CPT) * (tiles - q);
  };
  const int64_t nt = col_first(c1) - col_first(c0);
  if (order_cols || nt <= 0) {
    mp.R = 0;
    mp.t_off = (int)col_first(c0);
    return nt;
  }
  // rows per band: the band's A panels (R x 128 x k doubles) take about 2 MB of the XCD's 4 MB L2
  int R = r_env > 0 ? r_env : (int)std::min<int64_t>(32, std::max<int64_t>(2, (2 << 20) / ((int64_t)128 * 8 * std::max(k, 1))));
  const int first_row = c0 / CPT;
  while ((tiles - first_row + R - 1) / R > 159) ++R;
  mp.R = R;
  mp.band0 = first_row / R;
  int64_t acc = 0;
  int nb = 0;
  for (int b = mp.band0; b * R < tiles; ++b, ++nb) {
    mp.band_start[nb] = (int)acc;
    const int r0 = b * R, r1 = std::min(r0 + R, tiles);
    for (int tm = r0; tm < r1; ++tm) acc += std::max(0, std::min(c1, CPT * (tm + 1)) - c0);
  }
  mp.band_start[nb] = (int)acc;
  mp.nband = nb;
  return acc;   // == nt
}

template <int SBN, bool ACCUM>
static int launch_syrk_mirror(bigkrls_ctx* ctx, const GemmOperands& g, double alpha, double* C, int64_t ldc,
                              int tiles, int c0, int c1) {
  if (c0 >= c1) return BIGKRLS_OK;
  SyrkMap mp;
  const int64_t nt = syrk_map_build<SBN>(mp, tiles, c0, c1, g.K);
  if (nt <= 0) return BIGKRLS_OK;
  constexpr size_t smem = (SBN == 64 && SYRK64_LDS_EXACT) ? smem_bytes_nt(SBN) : smem_bytes(SBN);
  BK_TRY(ensure_dyn_smem(ctx, (const void*)syrk_mirror_kernel<SBN, ACCUM>, smem));
  hipLaunchKernelGGL((syrk_mirror_kernel<SBN, ACCUM>), dim3((unsigned)nt), dim3(NT), smem, ctx->stream, g,
                     alpha, C, ldc, mp);
  BK_CHECK_LAUNCH();
  return BIGKRLS_OK;
}

int syrk_mirror(bigkrls_ctx* ctx, int64_t m, int64_t k, double alpha, const double* A, int64_t lda,
                const double* B, int64_t ldb, double* C, int64_t ldc, int tn_begin, int tn_end,
                bool narrow_tiles, bool skip_first_column) {
  if (m <= 0 || k <= 0) return BIGKRLS_OK;
  BK_REQUIRE(m < (1ll << 31) && k < (1ll << 31), "syrk_mirror: dimension too large");
  GemmOperands g{A, B, lda, ldb, (int)m, (int)m, (int)k, nullptr};
  const int tiles = (int)((m + BM - 1) / BM);
  if (tn_end < 0 || tn_end > tiles) tn_end = tiles;
  if (tn_begin >= tn_end) return BIGKRLS_OK;
  // 128 x 64 tiles (three workgroups per CU) share the GPU better with a concurrent
  // register-resident panel QR; alone, 128 x 128 tiles are ~6 % faster (N = 20 000: 1.65 vs 1.76 ms)
  BK_REQUIRE(!skip_first_column || narrow_tiles, "syrk_mirror: skip_first_column needs 64-wide tiles");
  // (skip_first_column: the first 64-wide tile column of group tn_begin was updated by the caller)
  if (narrow_tiles)
    return launch_syrk_mirror<64, true>(ctx, g, alpha, C, ldc, tiles, 2 * tn_begin + (skip_first_column ? 1 : 0), 2 * tn_end);
  return launch_syrk_mirror<128, true>(ctx, g, alpha, C, ldc, tiles, tn_begin, tn_end);
}

// C = alpha A B' for a product that is symmetric (A = Q diag(w), B = Q): lower tiles computed, stored twice
int syrk_mirror_set(bigkrls_ctx* ctx, int64_t m, int64_t k, double alpha, const double* A, int64_t lda,
                    const double* B, int64_t ldb, double* C, int64_t ldc) {
  if (m <= 0) return BIGKRLS_OK;
  BK_REQUIRE(k > 0 && m < (1ll << 31) && k < (1ll << 31), "syrk_mirror_set: bad dimensions");
  GemmOperands g{A, B, lda, ldb, (int)m, (int)m, (int)k, nullptr};
  const int tiles = (int)((m + BM - 1) / BM);
  return launch_syrk_mirror<128, false>(ctx, g, alpha, C, ldc, tiles, 0, tiles);
}

// 128 x 64 tiles, columns [c64_begin, c64_end) in units of 64: the lower triangle (with the
// diagonal) of exactly those columns is updated and mirrored.
int syrk_mirror_cols(bigkrls_ctx* ctx, int64_t m, int64_t k, double alpha, const double* A, int64_t lda,
                     const double* B, int64_t ldb, double* C, int64_t ldc, int c64_begin, int c64_end) {
  if (m <= 0 || k <= 0) return BIGKRLS_OK;
  BK_REQUIRE(m < (1ll << 31) && k < (1ll << 31), "syrk_mirror_cols: dimension too large");
  GemmOperands g{A, B, lda, ldb, (int)m, (int)m, (int)k, nullptr};
  const int tiles = (int)((m + BM - 1) / BM);
  // (an odd last column of the matrix: the second column of the last group does not exist, but its
  //  tiles are enumerated; they lie entirely outside the matrix and store nothing)
  if (c64_end < 0 || c64_end > 2 * tiles) c64_end = 2 * tiles;
  if (c64_begin < 0) c64_begin = 0;
  return launch_syrk_mirror<64, true>(ctx, g, alpha, C, ldc, tiles, c64_begin, c64_end);
}

int syrk_lower(bigkrls_ctx* ctx, int64_t m, int64_t k, double alpha, const double* A, int64_t lda,
               const double* B, int64_t ldb, double* C, int64_t ldc) {
  if (m <= 0 || k <= 0) return BIGKRLS_OK;
  BK_REQUIRE(m < (1ll << 31) && k < (1ll << 31), "syrk_lower: dimension too large");
  GemmOperands g{A, B, lda, ldb, (int)m, (int)m, (int)k, nullptr};
  const int tiles = (int)((m + BM - 1) / BM);
  const int64_t nt = (int64_t)tiles * (tiles + 1) / 2;
  BK_TRY(ensure_dyn_smem(ctx, (const void*)syrk_lower_kernel, smem_bytes(128)));
  hipLaunchKernelGGL(syrk_lower_kernel, dim3((unsigned)nt), dim3(NT), smem_bytes(128), ctx->stream, g,
                     alpha, C, ldc, tiles);
  BK_CHECK_LAUNCH();
  return BIGKRLS_OK;
}

// ---------------------------------------------------------------------------
// batched NN GEMM (divide & conquer merges): one descriptor per problem
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(NT, GEMM_OCC) void gemm_batched_nn_kernel(const GemmDesc* __restrict__ descs,
                                                             int tiles_m_max) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const GemmDesc d = descs[blockIdx.y];
  const int tm = blockIdx.x % tiles_m_max, tn = blockIdx.x / tiles_m_max;
  const int m0 = tm * BM, n0 = tn * 128;
  if (m0 >= d.m || n0 >= d.n) return;
  GemmOperands g{d.A, d.B, d.lda, d.ldb, d.m, d.n, d.k, d.kidx};
  d4 acc[4][4];
  gemm_tile<false, false, 128, true>(g, m0, n0, 0, d.k, smem, acc);
  double* C = d.C;
  const int64_t ldc = d.ldc;
  const int M = d.m, N = d.n;
  acc_foreach<128>(acc, m0, n0, [&](int m, int n, double v) {
    if (m < M && n < N) C[(int64_t)m + (int64_t)n * ldc] = v;
  });
}

int gemm_batched_nn(bigkrls_ctx* ctx, const GemmDesc* d_descs, int n_batch, int max_m, int max_n) {
  if (n_batch <= 0 || max_m <= 0 || max_n <= 0) return BIGKRLS_OK;
  const int tiles_m = (max_m + BM - 1) / BM, tiles_n = (max_n + 127) / 128;
  BK_TRY(ensure_dyn_smem(ctx, (const void*)gemm_batched_nn_kernel, smem_bytes(128)));
  // grid.y is limited to 65535
  for (int b0 = 0; b0 < n_batch; b0 += 65535) {
    const int nb = std::min(65535, n_batch - b0);
    hipLaunchKernelGGL(gemm_batched_nn_kernel, dim3(tiles_m * tiles_n, nb), dim3(NT),
                       smem_bytes(128), ctx->stream, d_descs + b0, tiles_m);
    BK_CHECK_LAUNCH();
  }
  return BIGKRLS_OK;
}

// ---------------------------------------------------------------------------
// Gaussian kernel block: out[i,j] = exp(-(na_i + nb_j - 2 a_i.b_j)/sigma)
// ---------------------------------------------------------------------------
__device__ __forceinline__ double exp_nonpos(double x);
template <int BN>
__global__ __launch_bounds__(NT) void kernel_block_kernel(GemmOperands g,
                                                          const double* __restrict__ na,
                                                          const double* __restrict__ nb,
                                                          double neg_inv_sigma,
                                                          double* __restrict__ out, int64_t ldo,
                                                          int tiles_m, int tiles_n,
                                                          int64_t diag_shift) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int ntile = tiles_m * tiles_n;
  const int tid = xcd_remap(blockIdx.x, ntile);
  const int tm = tid % tiles_m, tn = tid / tiles_m;
  const int m0 = tm * BM, n0 = tn * BN;
  d4 acc[4][BN / 32];
  gemm_tile<false, true, BN>(g, m0, n0, 0, g.K, smem, acc);
  const int M = g.M, N = g.N;
  acc_foreach<BN>(acc, m0, n0, [&](int m, int n, double v) {
    if (m < M && n < N) {
      double d2 = na[m] + nb[n] - 2.0 * v;
      d2 = d2 > 0.0 ? d2 : 0.0;
      double kv = exp_nonpos(d2 * neg_inv_sigma);
      if (diag_shift >= 0 && (int64_t)m == (int64_t)n + diag_shift) kv = 1.0;
      out[(int64_t)m + (int64_t)n * ldo] = kv;
    }
  });
}

// exp(x) for x <= 0 (the only domain the Gaussian kernel needs), ~20 fp64 VALU ops:
// n = rint(x log2 e), r = x - n ln2 (two-term Cody-Waite), degree-13 Taylor/Horner on
// |r| <= ln2/2 (truncation 4e-18), scale by 2^n with v_ldexp_f64 (gradual underflow to 0).
// The kernel build is VALU-bound on the epilogue (rocprofv3: SQ_ACTIVE_INST_VALU), and the
// library exp() costs ~45 instructions per element.
__device__ __forceinline__ double exp_nonpos(double x) {
  x = fmax(x, -746.0);
  const double n = rint(x * 1.4426950408889634074);
  double r = fma(-n, 6.93147180369123816490e-01, x);
  r = fma(-n, 1.90821492927058770002e-10, r);
  double p = 1.6059043836821613e-10;                   // 1/13!
  p = fma(p, r, 2.08767569878681e-09);                 // 1/12!
  p = fma(p, r, 2.505210838544172e-08);                // 1/11!
  p = fma(p, r, 2.755731922398589e-07);                // 1/10!
  p = fma(p, r, 2.7557319223985893e-06);               // 1/9!
  p = fma(p, r, 2.48015873015873e-05);                 // 1/8!
  p = fma(p, r, 1.984126984126984e-04);                // 1/7!
  p = fma(p, r, 1.388888888888889e-03);                // 1/6!
  p = fma(p, r, 8.333333333333333e-03);                // 1/5!
  p = fma(p, r, 4.1666666666666664e-02);               // 1/4!
  p = fma(p, r, 1.6666666666666666e-01);               // 1/3!
  p = fma(p, r, 0.5);
  p = fma(p, r, 1.0);
  p = fma(p, r, 1.0);
  return ldexp(p, (int)n);
}

// The same with a 32-entry table: x log2(e) = (32 q + j + f) / 32, e^x = 2^q 2^(j/32) e^r with |r| <= ln2/64, where a
// degree-6 polynomial has a truncation error of 5e-18 -- 7 fused multiply-adds less per element than exp_nonpos, in
// kernels whose time is the fp64 VALU work of this epilogue. `tab` (LDS, 32 doubles) = 2^(j/32), correctly rounded.
__device__ __constant__ double kExp2Tab32[32] = {
    1.0, 1.0218971486541166, 1.0442737824274138, 1.0671404006768237, 1.0905077326652577, 1.1143867425958924,
    1.1387886347566916, 1.1637248587775775, 1.189207115002721, 1.215247359980469, 1.241857812073484,
    1.2690509571917332, 1.2968395546510096, 1.3252366431597413, 1.3542555469368927, 1.383909881963832,
    1.4142135623730951, 1.4451808069770467, 1.4768261459394993, 1.5091644275934228, 1.5422108254079407,
    1.5759808451078865, 1.6104903319492543, 1.645755478153965, 1.681792830507429, 1.718619298122478,
    1.7562521603732995, 1.7947090750031072, 1.8340080864093424, 1.8741676341103, 1.9152065613971474,
    1.9571441241754002};
__device__ __forceinline__ double exp_nonpos_tab(double x, const double* __restrict__ tab) {
  x = fmax(x, -746.0);
  const double nf = rint(x * 46.166241308446828384);           // 32 / ln 2
  const int n = (int)nf;
  double r = fma(-nf, 6.93147180369123816490e-01 / 32.0, x);    // Cody-Waite: the high part has 21 trailing zero bits
  r = fma(-nf, 1.90821492927058770002e-10 / 32.0, r);
  double p = 1.0 / 720.0;
  p = fma(p, r, 1.0 / 120.0);
  p = fma(p, r, 1.0 / 24.0);
  p = fma(p, r, 1.0 / 6.0);
  p = fma(p, r, 0.5);
  p = fma(p, r, 1.0);
  p *= r;                                                       // e^r - 1
  const double t = tab[n & 31];
  return ldexp(fma(t, p, t), n >> 5);
}

// XCD-aware, L2-blocked order of the one-wave-per-tile kernel builds. Consecutive workgroups round-robin over the 8
// XCDs, so with the plain tile order every XCD walks the whole matrix and streams all of X through its 4 MB L2 again
// and again (X is 8 MB at N = 50 000, P = 20: 1.14 x the algorithmic bytes moved, profiles/r03/r03k_kernel_build_pmc.log).
// Here the tiles are ordered band by band (a band = R consecutive tile rows, whose R x 32 rows of X take ~1 MB),
// inside a band column by column, and every XCD takes one contiguous eighth of that sequence (xcd_remap): it keeps its
// band's row panels in its L2 and streams each column panel once -- the order of the trailing update (SyrkMap), in
// closed form because a launch has up to millions of 32 x 32 tiles.
// Lower triangle of `tiles` tile rows: band b = rows [bR, min((b+1)R, tiles)), columns 0 .. r1-1, column tn holding
// the rows max(r0, tn) .. r1-1. Full bands before b hold R^2 b(b-1)/2 + b R(R+1)/2 tiles.
__device__ __forceinline__ void band_major_lower(int64_t sq, int tiles, int R, int& tm, int& tn) {
  const double Rd = (double)R;
  // start(b) = R^2 b(b-1)/2 + b R(R+1)/2 = (R^2/2) b^2 + (R/2) b  ->  b = floor of the positive root, then corrected
  int b = (int)((-0.5 * Rd + sqrt(0.25 * Rd * Rd + 2.0 * Rd * Rd * (double)sq)) / (Rd * Rd));
  auto start = [&](int q) { return (int64_t)R * R * q * (q - 1) / 2 + (int64_t)q * R * (R + 1) / 2; };
  while (b > 0 && start(b) > sq) --b;
  while ((int64_t)(b + 1) * R < tiles && start(b + 1) <= sq) ++b;
  const int r0 = b * R, r1 = min(r0 + R, tiles), Rb = r1 - r0;
  int64_t sp = sq - start(b);
  const int64_t nfull = (int64_t)r0 * Rb;                 // the columns left of the band's diagonal corner
  if (sp < nfull) {
    tn = (int)(sp / Rb);
    tm = r0 + (int)(sp % Rb);
    return;
  }
  sp -= nfull;                                            // corner: column r0 + c holds Rb - c rows
  int c = (int)(((2.0 * Rb + 1.0) - sqrt((2.0 * Rb + 1.0) * (2.0 * Rb + 1.0) - 8.0 * (double)sp)) * 0.5);
  auto cstart = [&](int q) { return (int64_t)q * Rb - (int64_t)q * (q - 1) / 2; };
  while (c > 0 && cstart(c) > sp) --c;
  while (c + 1 < Rb && cstart(c + 1) <= sp) ++c;
  tn = r0 + c;
  tm = tn + (int)(sp - cstart(c));
}
// rectangle of tiles_m x tiles_n tiles: bands of R tile rows, inside a band column by column
__device__ __forceinline__ void band_major_rect(int64_t sq, int tiles_m, int tiles_n, int R, int& tm, int& tn) {
  const int64_t per_band = (int64_t)R * tiles_n;
  const int b = (int)(sq / per_band);
  const int r0 = b * R, Rb = min(R, tiles_m - r0);
  const int64_t sp = sq - (int64_t)b * per_band;
  tn = (int)(sp / Rb);
  tm = r0 + (int)(sp % Rb);
}

typedef double d2v __attribute__((ext_vector_type(2)));

// Small-P variant (P <= 128: the fit's own regime, P = 5..50): the operands are a few
// MB and live in L2, the output is 8 N^2 bytes, so the kernel is HBM-write bound and the
// only job is to keep many independent store streams in flight. No LDS, no barriers:
// every wave owns one 32 x 32 output tile (2 x 2 MFMA tiles), pulls its fragments
// straight from global/L2 in MFMA operand layout (16 consecutive rows = one 128-byte
// segment per k), runs ceil(P/4) MFMA steps and the exp epilogue, and streams 128-byte
// store segments. ~100 VGPRs -> 4-5 waves per SIMD cover the store latency.
template <int KS>
__global__ __launch_bounds__(NT) void kernel_block_wave_kernel(
    const double* __restrict__ A, int64_t lda, int U, const double* __restrict__ B, int64_t ldb, int V,
    int P, const double* __restrict__ na, const double* __restrict__ nb, double neg_inv_sigma,
    double* __restrict__ out, int64_t ldo, int64_t diag_shift, int tiles_m, int tiles_n, int band_rows,
    int nt_stores, int64_t ntiles) {
  __shared__ double etab[32];
  if (threadIdx.x < 32) etab[threadIdx.x] = kExp2Tab32[threadIdx.x];
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const int64_t w = (int64_t)xcd_remap(blockIdx.x, gridDim.x) * 4 + (threadIdx.x >> 6);
  if (w >= ntiles) return;
  int tm, tn;
  band_major_rect(w, tiles_m, tiles_n, band_rows, tm, tn);
  const int m0 = tm * 32, n0 = tn * 32;
  const int lm = lane & 15, lk = lane >> 4;
  d4 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = (d4){0.0, 0.0, 0.0, 0.0};
  const int ma0 = min(m0 + lm, U - 1), ma1 = min(m0 + 16 + lm, U - 1);
  const int nb0 = min(n0 + lm, V - 1), nb1 = min(n0 + 16 + lm, V - 1);
  const double* a0p = A + ma0;
  const double* a1p = A + ma1;
  const double* b0p = B + nb0;
  const double* b1p = B + nb1;
  // norms of this lane's output columns / rows: issued first so that their latency hides
  // behind the fragment loads
  double nbv[2][4];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int r = 0; r < 4; ++r) nbv[j][r] = nb[min(n0 + j * 16 + lk + 4 * r, V - 1)];
  double nam[2];
  nam[0] = na[ma0];
  nam[1] = na[ma1];
  // K in chunks of KS MFMA steps (4 k each): all 4*KS fragment loads of a chunk are in
  // flight before the first MFMA; k >= P reads a clamped address and is zeroed.
  for (int kc0 = 0; kc0 < P; kc0 += 4 * KS) {
    double fa0[KS], fa1[KS], fb0[KS], fb1[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const int k = kc0 + 4 * s + lk;
      const int64_t kc = min(k, P - 1);
      fa0[s] = a0p[kc * lda];
      fa1[s] = a1p[kc * lda];
      fb0[s] = b0p[kc * ldb];
      fb1[s] = b1p[kc * ldb];
    }
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const bool kv = (kc0 + 4 * s + lk) < P;
      const double a0 = kv ? fa0[s] : 0.0, a1 = kv ? fa1[s] : 0.0;
      const double b0 = kv ? fb0[s] : 0.0, b1 = kv ? fb1[s] : 0.0;
      acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(b0, a0, acc[0][0], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(b0, a1, acc[1][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(b1, a0, acc[0][1], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(b1, a1, acc[1][1], 0, 0, 0);
    }
  }

  // 16-byte stores: adjacent lanes (rows 2t, 2t+1) swap one value so that the even lane
  // owns rows (2t, 2t+1) of column n(r) and the odd lane the same rows of column n(r+1).
  const bool odd = (lane & 1) != 0;
  const bool vec_ok = ((ldo & 1) == 0) && ((reinterpret_cast<uintptr_t>(out) & 15) == 0);
  // wave-uniform: does this 32 x 32 tile touch the forced-unit diagonal m == n + diag_shift ?
  const bool has_diag = diag_shift >= 0 && (int64_t)m0 < (int64_t)n0 + 32 + diag_shift &&
                        (int64_t)m0 + 32 > (int64_t)n0 + diag_shift;
  const int dsh = (int)diag_shift;
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int rp = 0; rp < 4; rp += 2) {
      double kv[2][2];  // [i][r - rp]
#pragma unroll
      for (int rr = 0; rr < 2; ++rr) {
        const int n = n0 + j * 16 + lk + 4 * (rp + rr);
        const double nbn = nbv[j][rp + rr];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const int m = m0 + i * 16 + lm;
          double d2 = fma(-2.0, acc[i][j][rp + rr], nam[i] + nbn);
          d2 = fmax(d2, 0.0);
          double e = exp_nonpos_tab(d2 * neg_inv_sigma, etab);
          if (has_diag && (m - n) == dsh) e = 1.0;
          kv[i][rr] = e;
        }
      }
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int m = m0 + i * 16 + lm;
        if (vec_ok) {
          const double send = odd ? kv[i][0] : kv[i][1];
          const double recv = __shfl_xor(send, 1, 64);
          const double lo = odd ? recv : kv[i][0];
          const double hi = odd ? kv[i][1] : recv;
          const int mrow = m & ~1;                              // first of the row pair
          const int n = n0 + j * 16 + lk + 4 * (rp + (odd ? 1 : 0));
          if (n < V) {
            double* dst = out + (int64_t)mrow + (int64_t)n * ldo;
            if (mrow + 1 < U) {
              if (nt_stores) {                                  // (uniform; see kb_nontemporal)
                d2v pr;
                pr.x = lo;
                pr.y = hi;
                __builtin_nontemporal_store(pr, reinterpret_cast<d2v*>(dst));
              } else {
                *reinterpret_cast<double2*>(dst) = make_double2(lo, hi);
              }
            } else if (mrow < U) {
              dst[0] = lo;
            }
          }
        } else {
#pragma unroll
          for (int rr = 0; rr < 2; ++rr) {
            const int n = n0 + j * 16 + lk + 4 * (rp + rr);
            if (m < U && n < V) out[(int64_t)m + (int64_t)n * ldo] = kv[i][rr];
          }
        }
      }
    }
}

// kb_store_tile with non-temporal stores
template <bool CHECK>
__device__ __forceinline__ void kbt_store_tile(double* __restrict__ out, int64_t ldo, int U, int V,
                                               int m0, int n0, const double (&e)[2][2][4], bool vec_ok) {
  const int lane = threadIdx.x & 63;
  const int lm = lane & 15, lk = lane >> 4;
  const bool odd = (lane & 1) != 0;
  if (!CHECK) {
    double* base = out + (int64_t)(m0 + (lm & ~1)) + (int64_t)(n0 + lk + (odd ? 4 : 0)) * ldo;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int rp = 0; rp < 4; rp += 2)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const double send = odd ? e[i][j][rp] : e[i][j][rp + 1];
          const double recv = __shfl_xor(send, 1, 64);
          d2v pr;
          pr.x = odd ? recv : e[i][j][rp];
          pr.y = odd ? e[i][j][rp + 1] : recv;
          __builtin_nontemporal_store(pr, reinterpret_cast<d2v*>(base + i * 16 + (int64_t)(j * 16 + 4 * rp) * ldo));
        }
    return;
  }
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int rp = 0; rp < 4; rp += 2)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int m = m0 + i * 16 + lm;
        if (vec_ok) {
          const double send = odd ? e[i][j][rp] : e[i][j][rp + 1];
          const double recv = __shfl_xor(send, 1, 64);
          d2v pr;
          pr.x = odd ? recv : e[i][j][rp];
          pr.y = odd ? e[i][j][rp + 1] : recv;
          const int mrow = m & ~1;                              // first of the row pair
          const int n = n0 + j * 16 + lk + 4 * (rp + (odd ? 1 : 0));
          if (n < V) {
            double* dst = out + (int64_t)mrow + (int64_t)n * ldo;
            if (mrow + 1 < U) __builtin_nontemporal_store(pr, reinterpret_cast<d2v*>(dst));
            else if (mrow < U) __builtin_nontemporal_store(pr.x, dst);
          }
        } else {
#pragma unroll
          for (int rr = 0; rr < 2; ++rr) {
            const int n = n0 + j * 16 + lk + 4 * (rp + rr);
            if (m < U && n < V) __builtin_nontemporal_store(e[i][j][rp + rr], out + (int64_t)m + (int64_t)n * ldo);
          }
        }
      }
}

// Paired 16-byte stores of one wave's 32 x 32 tile held in MFMA accumulator layout
// (e[i][j][r] = element (m0 + 16 i + (lane & 15), n0 + 16 j + (lane >> 4) + 4 r)): adjacent lanes
// (rows 2t, 2t+1) swap one value so that the even lane owns rows (2t, 2t+1) of column n(r) and the
// odd lane the same rows of column n(r+1).
template <bool CHECK = true>
__device__ __forceinline__ void kb_store_tile(double* __restrict__ out, int64_t ldo, int U, int V,
                                              int m0, int n0, const double (&e)[2][2][4], bool vec_ok) {
  const int lane = threadIdx.x & 63;
  const int lm = lane & 15, lk = lane >> 4;
  const bool odd = (lane & 1) != 0;
  if (!CHECK) {
    // interior tile, 16-byte aligned output: no predicates (a predicated store sits in a basic block of its own;
    // the 32 x 32 kernels spend a fifth of their instructions on the exec-mask bookkeeping of the general path)
    double* base = out + (int64_t)(m0 + (lm & ~1)) + (int64_t)(n0 + lk + (odd ? 4 : 0)) * ldo;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int rp = 0; rp < 4; rp += 2)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const double send = odd ? e[i][j][rp] : e[i][j][rp + 1];
          const double recv = __shfl_xor(send, 1, 64);
          const double lo = odd ? recv : e[i][j][rp];
          const double hi = odd ? e[i][j][rp + 1] : recv;
          *reinterpret_cast<double2*>(base + i * 16 + (int64_t)(j * 16 + 4 * rp) * ldo) = make_double2(lo, hi);
        }
    return;
  }
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int rp = 0; rp < 4; rp += 2)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int m = m0 + i * 16 + lm;
        if (vec_ok) {
          const double send = odd ? e[i][j][rp] : e[i][j][rp + 1];
          const double recv = __shfl_xor(send, 1, 64);
          const double lo = odd ? recv : e[i][j][rp];
          const double hi = odd ? e[i][j][rp + 1] : recv;
          const int mrow = m & ~1;                              // first of the row pair
          const int n = n0 + j * 16 + lk + 4 * (rp + (odd ? 1 : 0));
          if (n < V) {
            double* dst = out + (int64_t)mrow + (int64_t)n * ldo;
            if (mrow + 1 < U) *reinterpret_cast<double2*>(dst) = make_double2(lo, hi);
            else if (mrow < U) dst[0] = lo;
          }
        } else {
#pragma unroll
          for (int rr = 0; rr < 2; ++rr) {
            const int n = n0 + j * 16 + lk + 4 * (rp + rr);
            if (m < U && n < V) out[(int64_t)m + (int64_t)n * ldo] = e[i][j][rp + rr];
          }
        }
      }
}

// Symmetric Gram variant (A == B, square, unit diagonal): only the wave tiles on or below the
// diagonal are computed (half the MFMA, exp and operand traffic); an off-diagonal tile is stored
// twice, the second time transposed through a wave-private LDS buffer so that the mirrored
// stores are the same 16-byte row pairs.
template <int KS>
__global__ __launch_bounds__(NT) void kernel_block_sym_kernel(
    const double* __restrict__ A, int64_t lda, int U, int P, const double* __restrict__ na,
    double neg_inv_sigma, double* __restrict__ out, int64_t ldo, int tiles, int band_rows, int nt_stores,
    int64_t ntiles) {
  __shared__ double tbuf[4][32 * 33];
  __shared__ double etab[32];
  if (threadIdx.x < 32) etab[threadIdx.x] = kExp2Tab32[threadIdx.x];
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t w = (int64_t)xcd_remap(blockIdx.x, gridDim.x) * 4 + wave;
  if (w >= ntiles) return;
  int tm, tn;
  band_major_lower(w, tiles, band_rows, tm, tn);
  const int m0 = tm * 32, n0 = tn * 32;
  const int lm = lane & 15, lk = lane >> 4;
  d4 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = (d4){0.0, 0.0, 0.0, 0.0};
  const int ma0 = min(m0 + lm, U - 1), ma1 = min(m0 + 16 + lm, U - 1);
  const int nb0 = min(n0 + lm, U - 1), nb1 = min(n0 + 16 + lm, U - 1);
  const double* a0p = A + ma0;
  const double* a1p = A + ma1;
  const double* b0p = A + nb0;
  const double* b1p = A + nb1;
  double nbv[2][4];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int r = 0; r < 4; ++r) nbv[j][r] = na[min(n0 + j * 16 + lk + 4 * r, U - 1)];
  double nam[2];
  nam[0] = na[ma0];
  nam[1] = na[ma1];
  for (int kc0 = 0; kc0 < P; kc0 += 4 * KS) {
    double fa0[KS], fa1[KS], fb0[KS], fb1[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const int k = kc0 + 4 * s + lk;
      const int64_t kc = min(k, P - 1);
      fa0[s] = a0p[kc * lda];
      fa1[s] = a1p[kc * lda];
      fb0[s] = b0p[kc * lda];
      fb1[s] = b1p[kc * lda];
    }
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const bool kv = (kc0 + 4 * s + lk) < P;
      const double a0 = kv ? fa0[s] : 0.0, a1 = kv ? fa1[s] : 0.0;
      const double b0 = kv ? fb0[s] : 0.0, b1 = kv ? fb1[s] : 0.0;
      acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(b0, a0, acc[0][0], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(b0, a1, acc[1][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(b1, a0, acc[0][1], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(b1, a1, acc[1][1], 0, 0, 0);
    }
  }
  const bool vec_ok = ((ldo & 1) == 0) && ((reinterpret_cast<uintptr_t>(out) & 15) == 0);
  const bool diag_tile = (tm == tn);
  double e[2][2][4];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int n = n0 + j * 16 + lk + 4 * r;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int m = m0 + i * 16 + lm;
        double d2 = fma(-2.0, acc[i][j][r], nam[i] + nbv[j][r]);
        d2 = fmax(d2, 0.0);
        double v = exp_nonpos_tab(d2 * neg_inv_sigma, etab);
        if (diag_tile && m == n) v = 1.0;
        e[i][j][r] = v;
      }
    }
  const bool interior = vec_ok && m0 + 32 <= U && n0 + 32 <= U;     // (wave-uniform)
  if (interior) { if (nt_stores) kbt_store_tile<false>(out, ldo, U, U, m0, n0, e, true); else kb_store_tile<false>(out, ldo, U, U, m0, n0, e, true); }
  else kb_store_tile<true>(out, ldo, U, U, m0, n0, e, vec_ok);
  if (!diag_tile) {
    // transpose through LDS: E[mloc][nloc], then the mirrored tile's accumulator layout reads
    // element (m' = n0 + 16 i + lm, n' = m0 + 16 j + lk + 4 r) = E[16 j + lk + 4 r][16 i + lm]
    double* tb = tbuf[wave];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) tb[(i * 16 + lm) * 33 + (j * 16 + lk + 4 * r)] = e[i][j][r];
    double et[2][2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) et[i][j][r] = tb[(j * 16 + lk + 4 * r) * 33 + (i * 16 + lm)];
    if (interior) { if (nt_stores) kbt_store_tile<false>(out, ldo, U, U, n0, m0, et, true); else kb_store_tile<false>(out, ldo, U, U, n0, m0, et, true); }
    else kb_store_tile<true>(out, ldo, U, U, n0, m0, et, vec_ok);
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Workgroup-tiled kernel build: one workgroup per 128 x 128 output tile. The two 128-row panels of X a tile needs
// are staged ONCE in LDS (k-major, in chunks of 32 columns of X) and shared by the four waves, each of which owns a
// 64 x 64 quarter (4 x 4 MFMA tiles); per byte written the tile reads 1/4 of what the one-wave-per-32x32 kernel
// fetches, which is what that kernel loses at large N (X no longer fits the XCDs' L2s: 4.96 TB/s written at
// N = 20 000, 3.7 at 50 000, 2.9 at 100 000) and at P > 32 (its fragments come straight from global memory).
// Tile order: XCD-aware (symmetric: the band-major order of the trailing update, SyrkMap; rectangular: xcd_remap).
// SYM: the lower tiles are computed, an off-diagonal 32 x 32 piece is stored a second time transposed through a
// wave-private LDS buffer (the panel space, free after the products). Stores are non-temporal 16-byte row pairs:
// the output is written once and must not displace X from the L2.
// ---------------------------------------------------------------------------------------------------------------
constexpr int KBT_KC = 32;            // columns of X staged per chunk
constexpr int KBT_LD = 128 + 8;       // panel row stride in doubles (the k rows of a fragment read half the banks apart)
constexpr size_t kbt_smem_bytes() { return (size_t)(2 * KBT_KC * KBT_LD + 256) * sizeof(double); }

template <bool SYM>
__global__ __launch_bounds__(NT, 2) void kernel_block_tiled_kernel(
    const double* __restrict__ A, int64_t lda, int U, const double* __restrict__ B, int64_t ldb, int V, int P,
    const double* __restrict__ na, const double* __restrict__ nb, double neg_inv_sigma, double* __restrict__ out,
    int64_t ldo, int64_t diag_shift, int tiles_m, int tiles_n, SyrkMap map) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  double* Xa = smem;
  double* Xb = smem + KBT_KC * KBT_LD;
  double* sna = smem + 2 * KBT_KC * KBT_LD;
  double* snb = sna + 128;
  int tm, tn;
  if (SYM) {
    // band-major, one contiguous eighth per XCD (see SyrkMap; 128-wide columns: CPT = 1)
    const int tiles = map.tiles;
    const int sq = xcd_remap(blockIdx.x, gridDim.x);
    int lo = 0, hi = map.nband;
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (map.band_start[mid] <= sq) lo = mid; else hi = mid;
    }
    int sp = sq - map.band_start[lo];
    const int r0 = (map.band0 + lo) * map.R, r1 = min(r0 + map.R, tiles);
    const int cfull = max(min(map.c1, r0 + 1), map.c0);
    const int nfull = (cfull - map.c0) * (r1 - r0);
    if (sp < nfull) {
      tn = map.c0 + sp / (r1 - r0);
      tm = r0 + sp % (r1 - r0);
    } else {
      sp -= nfull;
      tn = cfull;
      while (sp >= r1 - tn) { sp -= r1 - tn; ++tn; }
      tm = tn + sp;
    }
  } else {
    const int t = xcd_remap(blockIdx.x, tiles_m * tiles_n);
    tm = t % tiles_m;
    tn = t / tiles_m;
  }
  const int m0 = tm * 128, n0 = tn * 128;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = (wave & 1) * 64, wn = (wave >> 1) * 64;
  const int lm = lane & 15, lk = lane >> 4;
  const bool diag_wg = SYM && tm == tn;
  // a wave whose quarter lies entirely above the diagonal of a diagonal tile has nothing to produce (it still
  // stages and keeps the barriers)
  const bool idle = diag_wg && wm < wn;
  __shared__ double etab[32];
  if (tid < 128) sna[tid] = na[min(m0 + tid, U - 1)];
  else snb[tid - 128] = nb[min(n0 + tid - 128, V - 1)];
  if (tid < 32) etab[tid] = kExp2Tab32[tid];
  d4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (d4){0.0, 0.0, 0.0, 0.0};
  const int x = tid & 127, kh = tid >> 7;
  const double* ap = A + min(m0 + x, U - 1);
  const double* bp = B + min(n0 + x, V - 1);
  for (int kc0 = 0; kc0 < P; kc0 += KBT_KC) {
    const int cnt = min(KBT_KC, P - kc0);
    const int cnt4 = (cnt + 3) & ~3;
    double ra[KBT_KC / 2], rb[KBT_KC / 2];
#pragma unroll
    for (int q = 0; q < KBT_KC / 2; ++q) {
      const int k = kh + 2 * q;
      const int64_t kc = kc0 + min(k, cnt - 1);
      ra[q] = ap[kc * lda];
      rb[q] = bp[kc * ldb];
    }
    __syncthreads();                       // (the previous chunk's fragment reads are done)
#pragma unroll
    for (int q = 0; q < KBT_KC / 2; ++q) {
      const int k = kh + 2 * q;
      if (k < cnt4) {
        Xa[k * KBT_LD + x] = k < cnt ? ra[q] : 0.0;
        Xb[k * KBT_LD + x] = k < cnt ? rb[q] : 0.0;
      }
    }
    __syncthreads();
    if (!idle) {
      for (int kk = 0; kk < cnt4; kk += 4) {
        double af[4], bf[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) af[i] = Xa[(kk + lk) * KBT_LD + wm + i * 16 + lm];
#pragma unroll
        for (int j = 0; j < 4; ++j) bf[j] = Xb[(kk + lk) * KBT_LD + wn + j * 16 + lm];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(bf[j], af[i], acc[i][j], 0, 0, 0);
      }
    }
  }
  __syncthreads();                         // the panels are dead: their space becomes the transpose buffers
  if (idle) return;
  double* tb = smem + wave * (32 * 33);
  const bool vec_ok = ((ldo & 1) == 0) && ((reinterpret_cast<uintptr_t>(out) & 15) == 0);
  const int dsh = (int)diag_shift;
  const bool has_diag = diag_shift >= 0 && (int64_t)m0 < (int64_t)n0 + 128 + diag_shift &&
                        (int64_t)m0 + 128 > (int64_t)n0 + diag_shift;
#pragma unroll
  for (int si = 0; si < 2; ++si)
#pragma unroll
    for (int sj = 0; sj < 2; ++sj) {
      const int pm0 = m0 + wm + 32 * si, pn0 = n0 + wn + 32 * sj;       // this 32 x 32 piece
      if (diag_wg && pm0 < pn0) continue;                               // above the diagonal
      double e[2][2][4];
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int nl = wn + 32 * sj + j * 16 + lk + 4 * r;
          const double nbn = snb[nl];
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            const int ml = wm + 32 * si + i * 16 + lm;
            double d2 = fma(-2.0, acc[2 * si + i][2 * sj + j][r], sna[ml] + nbn);
            d2 = fmax(d2, 0.0);
            double v = exp_nonpos_tab(d2 * neg_inv_sigma, etab);
            if (has_diag && (m0 + ml) - (n0 + nl) == dsh) v = 1.0;
            e[i][j][r] = v;
          }
        }
      const bool interior = vec_ok && m0 + 128 <= U && n0 + 128 <= V;    // (workgroup-uniform)
      if (interior) kbt_store_tile<false>(out, ldo, U, V, pm0, pn0, e, true);
      else kbt_store_tile<true>(out, ldo, U, V, pm0, pn0, e, vec_ok);
      if (SYM && pm0 != pn0) {
        // E[mloc][nloc] through LDS; the mirrored piece's accumulator layout reads element
        // (m' = pn0 + 16 i + lm, n' = pm0 + 16 j + lk + 4 r) = E[16 j + lk + 4 r][16 i + lm]
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) tb[(i * 16 + lm) * 33 + (j * 16 + lk + 4 * r)] = e[i][j][r];
        double et[2][2][4];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) et[i][j][r] = tb[(j * 16 + lk + 4 * r) * 33 + (i * 16 + lm)];
        if (interior) kbt_store_tile<false>(out, ldo, V, U, pn0, pm0, et, true);
        else kbt_store_tile<true>(out, ldo, V, U, pn0, pm0, et, vec_ok);
      }
    }
}

// tile rows per band of the wave kernels' order: the band's rows of X (R x 32 x P doubles) take about 1 MB of an XCD's
// 4 MB L2 (BIGKRLS_KB_R overrides; a value >= the number of tile rows gives the plain column-by-column order)
static int kb_band_rows(int64_t p, int tiles) {
  static const int r_env = [] { const char* e = getenv("BIGKRLS_KB_R"); return e ? atoi(e) : 0; }();
  int64_t R = r_env > 0 ? r_env : std::max<int64_t>(8, (1 << 20) / (32 * 8 * std::max<int64_t>(p, 1)));
  R = std::max<int64_t>(1, std::min<int64_t>(R, 1024));
  return (int)std::min<int64_t>(R, std::max(tiles, 1));
}

// Non-temporal stores of the output tiles (the wave kernels; the workgroup-tiled kernel always uses them): the 8 N^2
// bytes of K pass through the write-back L2s and evict the rows of X the tiles are built from; with X larger than an
// XCD's 4 MB L2 the re-fetches stall the waves and non-temporal stores are 13 % faster (N = 50 000: 4.80 -> 4.24 ms,
// 4.17 -> 4.71 TB/s written). At N = 20 000, P = 20 (X = 3.2 MB) a loop over the launch alone prefers ordinary stores
// (0.730 vs 0.775 ms, round 4), but INSIDE the fit -- cold caches, the launch once per fit -- non-temporal stores are
// the faster ones (round 5, same-box A/B in fresh processes: 0.620 / 0.628 ms ordinary, 0.597 / 0.595 ms non-temporal;
// profiles/r05/r05b_knob_ab_C3.log): the threshold is X > 2 MB. BIGKRLS_KB_NT=0|1 overrides.
static int kb_nontemporal(int64_t rows, int64_t p) {
  static const int env = [] { const char* e = getenv("BIGKRLS_KB_NT"); return e ? atoi(e) : -1; }();
  if (env >= 0) return env != 0;
  return rows * p * (int64_t)sizeof(double) > (2ll << 20);    // 2 MB
}

int kernel_block(bigkrls_ctx* ctx, const double* A, int64_t u, int64_t lda, const double* B,
                 int64_t v, int64_t ldb, int64_t p, double sigma, double* out, int64_t ldo,
                 int64_t diag_shift) {
  BK_REQUIRE(u >= 0 && v >= 0 && p > 0, "kernel_block: bad dimensions");
  BK_REQUIRE(u < (1ll << 31) && v < (1ll << 31) && p < (1ll << 31), "kernel_block: too large");
  BK_REQUIRE(sigma > 0.0, "kernel_block: sigma must be > 0");
  if (u == 0 || v == 0) return BIGKRLS_OK;
  BK_REQUIRE(A && B && out, "kernel_block: null pointer");
  void *pna = nullptr, *pnb = nullptr;
  BK_TRY(ws_get(ctx, SLOT_NORMS_A, u * sizeof(double), &pna));
  BK_TRY(ws_get(ctx, SLOT_NORMS_B, v * sizeof(double), &pnb));
  BK_TRY(row_sqnorms(ctx, A, u, p, lda, (double*)pna));
  BK_TRY(row_sqnorms(ctx, B, v, p, ldb, (double*)pnb));
  // one workgroup per 128 x 128 tile with the X panels in LDS where it beats the one-wave-per-32x32 kernels: P > 32
  // (measured, TFLOP/s tiled vs wave: N = 100 000, P = 50: 44.7 vs 38.5; N = 30 000, P = 50: 37.2 vs 33.9;
  //  N = 50 000, P = 20: 21.1 vs 20.7; N = 20 000, P = 20: 17.8 vs 19.9; BIGKRLS_KB=wave|tiled forces either)
  static const int kb_force = [] {
    const char* e = getenv("BIGKRLS_KB");
    return e ? (std::string(e) == "tiled" ? 1 : (std::string(e) == "wave" ? -1 : 0)) : 0;
  }();
  const bool sym = A == B && u == v && lda == ldb && diag_shift == 0;
  const bool big = u >= 1024 && v >= 1024;
  if (kb_force > 0 || (kb_force == 0 && big && p > 32)) {
    const int tiles_m = (int)((u + 127) / 128), tiles_n = (int)((v + 127) / 128);
    SyrkMap mp{};
    int64_t nt = (int64_t)tiles_m * tiles_n;
    if (sym) {
      // rows per band: the A panels of a band are tiny (128 P doubles each): 32 rows keep them and the streamed B
      // panels far inside the L2
      static const int r_env = [] { const char* e = getenv("BIGKRLS_KB_R"); return e ? atoi(e) : 0; }();
      int R = r_env > 0 ? r_env : 32;
      while ((tiles_m + R - 1) / R > 159) ++R;
      mp.tiles = tiles_m; mp.c0 = 0; mp.c1 = tiles_m; mp.R = R; mp.band0 = 0; mp.t_off = 0;
      int64_t acc = 0;
      int nbands = 0;
      for (int b0 = 0; b0 * R < tiles_m; ++b0, ++nbands) {
        mp.band_start[nbands] = (int)acc;
        for (int tmr = b0 * R; tmr < std::min((b0 + 1) * R, tiles_m); ++tmr) acc += tmr + 1;
      }
      mp.band_start[nbands] = (int)acc;
      mp.nband = nbands;
      nt = acc;
    }
    BK_REQUIRE(nt < (1ll << 31), "kernel_block: too many tiles");
    BK_TRY(ensure_dyn_smem(ctx, sym ? (const void*)kernel_block_tiled_kernel<true> : (const void*)kernel_block_tiled_kernel<false>,
                           kbt_smem_bytes()));
    BK_TRY(prof_begin(ctx, "kernel_block", 2.0 * (double)u * (double)v * (double)p));
    if (sym)
      hipLaunchKernelGGL(kernel_block_tiled_kernel<true>, dim3((unsigned)nt), dim3(NT), kbt_smem_bytes(), ctx->stream, A, lda,
                         (int)u, B, ldb, (int)v, (int)p, (const double*)pna, (const double*)pnb, -1.0 / sigma, out, ldo,
                         diag_shift, tiles_m, tiles_n, mp);
    else
      hipLaunchKernelGGL(kernel_block_tiled_kernel<false>, dim3((unsigned)nt), dim3(NT), kbt_smem_bytes(), ctx->stream, A, lda,
                         (int)u, B, ldb, (int)v, (int)p, (const double*)pna, (const double*)pnb, -1.0 / sigma, out, ldo,
                         diag_shift, tiles_m, tiles_n, mp);
    BK_CHECK_LAUNCH();
    BK_TRY(prof_end(ctx, "kernel_block"));
    return BIGKRLS_OK;
  }
  if (p <= 128 && sym) {
    // symmetric Gram matrix (bGaussKernel): lower wave tiles + mirrored stores
    const int tiles = (int)((u + 31) / 32);
    const int64_t ntiles = (int64_t)tiles * (tiles + 1) / 2;
    BK_TRY(prof_begin(ctx, "kernel_block", 2.0 * (double)u * (double)v * (double)p));
    const int steps = (int)((p + 3) / 4);
    const int chunks = (steps + 7) / 8;
    const int ks = (steps + chunks - 1) / chunks;
    BK_REQUIRE((ntiles + 3) / 4 < (1ll << 31), "kernel_block: too many tiles");
    const dim3 grid((unsigned)((ntiles + 3) / 4));
    const int band_rows = kb_band_rows(p, tiles);
    const int kb_nt = kb_nontemporal(u, p);
#define BK_KBS(KS)                                                                                 \
  hipLaunchKernelGGL(kernel_block_sym_kernel<KS>, grid, dim3(NT), 0, ctx->stream, A, lda, (int)u,   \
                     (int)p, (const double*)pna, -1.0 / sigma, out, ldo, tiles, band_rows, kb_nt, ntiles)
    switch (ks) {
      case 1: BK_KBS(1); break;
      case 2: BK_KBS(2); break;
      case 3: BK_KBS(3); break;
      case 4: BK_KBS(4); break;
      case 5: BK_KBS(5); break;
      case 6: BK_KBS(6); break;
      case 7: BK_KBS(7); break;
      default: BK_KBS(8); break;
    }
#undef BK_KBS
    BK_CHECK_LAUNCH();
    BK_TRY(prof_end(ctx, "kernel_block"));
    return BIGKRLS_OK;
  }
  if (p <= 128) {
    const int tiles_m = (int)((u + 31) / 32), tiles_n = (int)((v + 31) / 32);
    const int64_t ntiles = (int64_t)tiles_m * tiles_n;
    BK_TRY(prof_begin(ctx, "kernel_block", 2.0 * (double)u * (double)v * (double)p));
    const int steps = (int)((p + 3) / 4);
    const int chunks = (steps + 7) / 8;
    const int ks = (steps + chunks - 1) / chunks;  // 1..8 MFMA steps per chunk, minimal padding
    BK_REQUIRE((ntiles + 3) / 4 < (1ll << 31), "kernel_block: too many tiles");
    const dim3 grid((unsigned)((ntiles + 3) / 4));
    const int band_rows = kb_band_rows(p, tiles_m);
    const int kb_nt = kb_nontemporal(std::max(u, v), p);
#define BK_KBW(KS)                                                                                  \
  hipLaunchKernelGGL(kernel_block_wave_kernel<KS>, grid, dim3(NT), 0, ctx->stream, A, lda, (int)u, B, \
                     ldb, (int)v, (int)p, (const double*)pna, (const double*)pnb, -1.0 / sigma, out,  \
                     ldo, diag_shift, tiles_m, tiles_n, band_rows, kb_nt, ntiles)
    switch (ks) {
      case 1: BK_KBW(1); break;
      case 2: BK_KBW(2); break;
      case 3: BK_KBW(3); break;
      case 4: BK_KBW(4); break;
      case 5: BK_KBW(5); break;
      case 6: BK_KBW(6); break;
      case 7: BK_KBW(7); break;
      default: BK_KBW(8); break;
    }
#undef BK_KBW
    BK_CHECK_LAUNCH();
    BK_TRY(prof_end(ctx, "kernel_block"));
    return BIGKRLS_OK;
  }
  GemmOperands g{A, B, lda, ldb, (int)u, (int)v, (int)p, nullptr};
  constexpr int KBN = 64;
  const int tiles_m = (int)((u + BM - 1) / BM), tiles_n = (int)((v + KBN - 1) / KBN);
  BK_TRY(ensure_dyn_smem(ctx, (const void*)kernel_block_kernel<KBN>, smem_bytes(KBN)));
  BK_TRY(prof_begin(ctx, "kernel_block", 2.0 * (double)u * (double)v * (double)p));
  hipLaunchKernelGGL(kernel_block_kernel<KBN>, dim3(tiles_m * tiles_n), dim3(NT), smem_bytes(KBN),
                     ctx->stream, g, (const double*)pna, (const double*)pnb, -1.0 / sigma, out, ldo,
                     tiles_m, tiles_n, diag_shift);
  BK_CHECK_LAUNCH();
  BK_TRY(prof_end(ctx, "kernel_block"));
  return BIGKRLS_OK;
}

}  // namespace bk
