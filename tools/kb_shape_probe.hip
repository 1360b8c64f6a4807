// The block-Lanczos product W = K B_j (n x 128 = (n x n) (n x 128)) in isolation (development tool): what bounds it?
//   * as in the fit: A = an n x n matrix streamed once from HBM;
//   * lda = 0: every k column of A is the same 8 n bytes (cache-resident): the same instruction stream without the
//     HBM stream;
//   * B of 64 columns (the 128 x 64 tile keeps two k-tiles in flight).
// BIGKRLS_GEMM_SPLITS=<n> (read once per process by csrc/gemm.hip) overrides the split-K count.
// hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/kb_shape_probe.hip -o tools/kb_shape_probe -Lbigkrls_amd -lbigkrls_hip -Wl,-rpath,'$ORIGIN/../bigkrls_amd'
#include "../bigkrls_amd/csrc/common.h"
#include <cstdio>
#include <cstdlib>
using namespace bk;
__global__ void fillr(double* p, int64_t n, unsigned seed) {
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) {
    unsigned x = (unsigned)e * 2654435761u + seed; x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;
    p[e] = (double)(x & 0xffff) / 65536.0 - 0.5;
  }
}
int main(int argc, char** argv) {
  bigkrls_ctx* ctx; if (bigkrls_ctx_create(0, &ctx)) { printf("%s\n", bigkrls_last_error()); return 1; }
  hipStream_t st = ctx->stream;
  const int64_t n = argc > 1 ? atoll(argv[1]) : 50000;
  double *A, *B, *W;
  if (hipMalloc(&A, n * n * 8) != hipSuccess || hipMalloc(&B, n * 128 * 8) != hipSuccess || hipMalloc(&W, n * 128 * 8) != hipSuccess) { printf("alloc failed\n"); return 1; }
  fillr<<<4096, 256, 0, st>>>(A, n * n, 1); fillr<<<2048, 256, 0, st>>>(B, n * 128, 2);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto timeit = [&](const char* name, double flops, auto fn) {
    fn(); hipStreamSynchronize(st);
    float best = 1e30f;
    for (int r = 0; r < 5; ++r) {
      hipEventRecord(e0, st); fn(); hipEventRecord(e1, st); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    printf("%-58s %9.1f us  %6.2f TFLOP/s  %.3f of 78.6\n", name, best * 1e3, flops / (best * 1e-3) / 1e12, flops / (best * 1e-3) / 78.6e12);
  };
  const char* sp = getenv("BIGKRLS_GEMM_SPLITS");
  printf("n = %lld, BIGKRLS_GEMM_SPLITS = %s\n", (long long)n, sp ? sp : "(library's choice)");
  timeit("W = A B, 128 columns, A streamed from HBM", 2.0 * n * n * 128, [&] { gemm(ctx, 0, 0, n, 128, n, 1.0, A, n, B, n, 0.0, W, n); });
  timeit("W = A B, 128 columns, lda = 0 (A cache-resident)", 2.0 * n * n * 128, [&] { gemm(ctx, 0, 0, n, 128, n, 1.0, A, 0, B, n, 0.0, W, n); });
  timeit("W = A B, 64 columns, A streamed from HBM", 2.0 * n * n * 64, [&] { gemm(ctx, 0, 0, n, 64, n, 1.0, A, n, B, n, 0.0, W, n); });
  timeit("W = A B, 64 columns, lda = 0", 2.0 * n * n * 64, [&] { gemm(ctx, 0, 0, n, 64, n, 1.0, A, 0, B, n, 0.0, W, n); });
  return 0;
}
