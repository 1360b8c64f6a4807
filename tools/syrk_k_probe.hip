// How does the trailing update scale with the rank of the update? (development probe)
// syrk_mirror at k = 128 / 256 / 512 for several m, both tile shapes: decides whether accumulating
// panels before touching A22 pays.
// hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/syrk_k_probe.hip -o tools/syrk_k_probe -Lbigkrls_amd -lbigkrls_hip -Wl,-rpath,'$ORIGIN/../bigkrls_amd'
#include "../bigkrls_amd/csrc/common.h"
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include <algorithm>
using namespace bk;
__global__ void fillr(double* p, int64_t n, unsigned seed) {
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) {
    unsigned x = (unsigned)e * 2654435761u + seed; x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;
    p[e] = ((double)(x & 0xffff) / 65536.0 - 0.5) * 1e-3;
  }
}
int main(int argc, char** argv) {
  bigkrls_ctx* ctx; if (bigkrls_ctx_create(0, &ctx)) { printf("%s\n", bigkrls_last_error()); return 1; }
  hipStream_t st = ctx->stream;
  const int64_t n = argc > 1 ? atoll(argv[1]) : 20000;
  double *C, *A, *B; hipMalloc(&C, n * n * 8); hipMalloc(&A, n * 512 * 8); hipMalloc(&B, n * 512 * 8);
  fillr<<<2048, 256, 0, st>>>(C, n * n, 1); fillr<<<2048, 256, 0, st>>>(A, n * 512, 2); fillr<<<2048, 256, 0, st>>>(B, n * 512, 3);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto timeit = [&](const char* name, int64_t m, int k, bool narrow, int tn0, int tn1) {
    auto fn = [&] { syrk_mirror(ctx, m, k, -1.0, A, n, B, n, C, n, tn0, tn1, narrow); };
    fn(); hipStreamSynchronize(st);
    hipEventRecord(e0, st); const int reps = 4;
    for (int r = 0; r < reps; ++r) fn();
    hipEventRecord(e1, st); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
    printf("%-22s m=%6lld k=%3d tiles=%s cols[%d,%d): %8.1f us  (%6.2f TFLOP/s if the whole triangle)  per 64 columns of rank: %7.1f us\n",
           name, (long long)m, k, narrow ? "128x64 " : "128x128", tn0, tn1, ms * 1e3,
           (double)m * (m + 1) * k / (ms * 1e-3) / 1e12, ms * 1e3 * 128.0 / k);
  };
  auto timecols = [&](const char* name, int64_t m, int k, int c0, int c1) {
    auto fn = [&] { syrk_mirror_cols(ctx, m, k, -1.0, A, n, B, n, C, n, c0, c1); };
    fn(); hipStreamSynchronize(st);
    hipEventRecord(e0, st); const int reps = 4;
    for (int r = 0; r < reps; ++r) fn();
    hipEventRecord(e1, st); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
    const double ca = 64.0 * c0, cb = c1 < 0 ? (double)m : 64.0 * c1;
    const double fl = 2.0 * k * ((cb - ca) * m - 0.5 * (cb * cb - ca * ca));
    printf("%-22s m=%6lld k=%3d tiles=128x64  c64[%d,%d): %8.1f us  %6.2f TFLOP/s\n", name, (long long)m, k, c0, c1,
           ms * 1e3, fl / (ms * 1e-3) / 1e12);
  };
  const bool quick = argc > 2;
  for (int64_t m : {n, (int64_t)(n * 0.7), n / 2}) {
    for (int k : {128, 256, 384, 512}) {
      if (!quick) timeit("syrk_mirror", m, k, false, 0, -1);
      timeit("syrk_mirror", m, k, true, 0, -1);
    }
    // the two equal-area pieces of a k = 256 update as the fit launches them (64-wide columns, the right part
    // starts at 1 - 1/sqrt 2 of the columns)
    const int ncol64 = (int)((m + 63) / 64), j1 = (int)(ncol64 * 0.2929 + 0.5);
    timecols("piece b (left cols)", m, 256, 0, j1);
    timecols("piece a (right cols)", m, 256, j1, -1);
  }
  return 0;
}
