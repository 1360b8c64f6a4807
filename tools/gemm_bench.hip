// GEMM core micro-benchmark (development tool)
// hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/gemm_bench.hip -o tools/gemm_bench -Lbigkrls_amd -lbigkrls_hip -Wl,-rpath,'$ORIGIN/../bigkrls_amd'
#include "../bigkrls_amd/csrc/common.h"
#include <cstdio>
#include <vector>
#include <algorithm>
#include <cstdlib>
#include <chrono>
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
  const int64_t n = argc > 1 ? atoll(argv[1]) : 20000;
  double *C, *A, *B; hipMalloc(&C, n * n * 8); hipMalloc(&A, n * 256 * 8); hipMalloc(&B, n * 256 * 8);
  fillr<<<2048, 256, 0, st>>>(C, n * n, 1); fillr<<<2048, 256, 0, st>>>(A, n * 256, 2); fillr<<<2048, 256, 0, st>>>(B, n * 256, 3);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto timeit = [&](const char* name, double flops, auto fn) {
    fn(); hipStreamSynchronize(st);
    hipEventRecord(e0, st); const int reps = 3;
    for (int r = 0; r < reps; ++r) fn();
    hipEventRecord(e1, st); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
    printf("%-44s %9.1f us  %7.2f TFLOP/s\n", name, ms * 1e3, flops / (ms * 1e-3) / 1e12);
  };
  timeit("syrk_mirror m=n k=128", (double)n * (n + 1) * 128, [&] { syrk_mirror(ctx, n, 128, -1.0, A, n, B, n, C, n); });
  timeit("syrk_mirror m=n k=16 (epilogue only)", (double)n * (n + 1) * 16, [&] { syrk_mirror(ctx, n, 16, -1.0, A, n, B, n, C, n); });
  timeit("gemm NT k=16 beta=1 (epilogue only)", 2.0 * n * n * 16, [&] { gemm(ctx, 0, 1, n, n, 16, -1.0, A, n, B, n, 1.0, C, n); });
  timeit("gemm NT k=16 beta=0 (epilogue only)", 2.0 * n * n * 16, [&] { gemm(ctx, 0, 1, n, n, 16, -1.0, A, n, B, n, 0.0, C, n); });
  timeit("syrk_lower  m=n k=128", (double)n * (n + 1) * 128, [&] { syrk_lower(ctx, n, 128, -1.0, A, n, B, n, C, n); });
  timeit("gemm NT m=n=n k=128 beta=1", 2.0 * n * n * 128, [&] { gemm(ctx, 0, 1, n, n, 128, -1.0, A, n, B, n, 1.0, C, n); });
  timeit("gemm NT m=n=n k=128 beta=0", 2.0 * n * n * 128, [&] { gemm(ctx, 0, 1, n, n, 128, -1.0, A, n, B, n, 0.0, C, n); });
  timeit("gemm NN (n x 64) = C(n x n) * A(n x 64)", 2.0 * n * n * 64, [&] { gemm(ctx, 0, 0, n, 64, n, 1.0, C, n, A, n, 0.0, B, n); });
  timeit("gemm NT (n x 64) = C(n x n) * At(64 x n)'", 2.0 * n * n * 64, [&] { gemm(ctx, 0, 1, n, 64, n, 1.0, C, n, A, 64, 0.0, B, n); });
  {
    // concurrency probe: trailing update and the A22 V product on two streams at once vs back to back
    bigkrls_ctx* ctx2; bigkrls_ctx_create(0, &ctx2);
    double* C2; hipMalloc(&C2, n * n * 8); hipMemcpy(C2, C, n * n * 8, hipMemcpyDeviceToDevice);
    double* Y; hipMalloc(&Y, n * 64 * 8);
    auto wall = [&](const char* name, auto fn) {
      fn(); hipDeviceSynchronize();
      auto t0 = std::chrono::steady_clock::now();
      for (int r = 0; r < 3; ++r) fn();
      hipDeviceSynchronize();
      double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / 3;
      printf("%-44s %9.1f us\n", name, us);
    };
    wall("syrk_mirror<64> alone", [&] { syrk_mirror(ctx, n, 128, -1.0, A, n, B, n, C, n, 0, -1, true); });
    wall("A22 V alone (other buffer)", [&] { gemm(ctx2, 0, 0, n, 64, n, 1.0, C2, n, A, n, 0.0, Y, n); });
    wall("syrk_mirror<64> || A22 V (two streams)", [&] {
      syrk_mirror(ctx, n, 128, -1.0, A, n, B, n, C, n, 0, -1, true);
      gemm(ctx2, 0, 0, n, 64, n, 1.0, C2, n, A, n, 0.0, Y, n); });
    // mixed tile shapes on two streams: do workgroups with different MFMA / epilogue phase lengths
    // overlap better than identical ones running in lockstep?
    const int tiles = (int)((n + 127) / 128);
    for (int cut : {tiles / 8, tiles / 5, tiles / 4, tiles / 3}) {
      char nm[96];
      snprintf(nm, sizeof nm, "syrk<128> cols [0,%d) || syrk<64> cols [%d,%d)", cut, cut, tiles);
      wall(nm, [&] {
        syrk_mirror(ctx, n, 128, -1.0, A, n, B, n, C, n, 0, cut, false);
        syrk_mirror(ctx2, n, 128, -1.0, A, n, B, n, C, n, cut, -1, true); });
      snprintf(nm, sizeof nm, "syrk<64> cols [0,%d) || syrk<128> cols [%d,%d)", cut, cut, tiles);
      wall(nm, [&] {
        syrk_mirror(ctx, n, 128, -1.0, A, n, B, n, C, n, 0, cut, true);
        syrk_mirror(ctx2, n, 128, -1.0, A, n, B, n, C, n, cut, -1, false); });
    }
    wall("syrk<128> cols [0,t/4) || syrk<128> rest", [&] {
      syrk_mirror(ctx, n, 128, -1.0, A, n, B, n, C, n, 0, tiles / 4, false);
      syrk_mirror(ctx2, n, 128, -1.0, A, n, B, n, C, n, tiles / 4, -1, false); });
    wall("syrk_mirror<128> alone", [&] { syrk_mirror(ctx, n, 128, -1.0, A, n, B, n, C, n, 0, -1, false); });
  }
  const int64_t q = 8192;
  if (n * n < 3 * q * q) return 0;
  timeit("gemm NN 8192^3", 2.0 * q * q * q, [&] { gemm(ctx, 0, 0, q, q, q, 1.0, C, q, C + q * q, q, 0.0, C + 2 * q * q, q); });
  timeit("gemm NT 8192^3", 2.0 * q * q * q, [&] { gemm(ctx, 0, 1, q, q, q, 1.0, C, q, C + q * q, q, 0.0, C + 2 * q * q, q); });
  timeit("gemm TN 8192^3", 2.0 * q * q * q, [&] { gemm(ctx, 1, 0, q, q, q, 1.0, C, q, C + q * q, q, 0.0, C + 2 * q * q, q); });
  return 0;
}
