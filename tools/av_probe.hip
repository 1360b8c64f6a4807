// The stage-1 product Y = A22 V (m x 64 = (m x m)(m x 64)) in isolation, read-only ceilings for the same bytes and the
// shader clock under load. With tools/experiments/av_tall64.patch applied to the library the second column is the
// dedicated tall-skinny kernel (BIGKRLS_NOAVTALL=1 selects the generic core); without it both columns are the generic core.
// hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/av_probe.hip -o tools/av_probe -Lbigkrls_amd -lbigkrls_hip -Wl,-rpath,'$ORIGIN/../bigkrls_amd'
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
    p[e] = (double)(x & 0xffff) / 65536.0 - 0.5;
  }
}
typedef double d2v __attribute__((ext_vector_type(2)));
// read ceilings for the same bytes: (a) every workgroup streams whole columns (m contiguous doubles) of the block;
// (b) the access pattern of av_tall64_kernel (a wave reads 256-byte column segments, 8 columns in flight per lane)
// with one add per element instead of the MFMA work
__global__ __launch_bounds__(256) void read_cols(const double* A, int64_t lda, int m, double* sink) {
  d2v s = {0, 0};
  for (int c = blockIdx.x; c < m; c += gridDim.x) {
    const d2v* col = (const d2v*)(A + (int64_t)c * lda);
    for (int r = threadIdx.x; r < m / 2; r += 256) s += col[r];
  }
  if (s.x + s.y == 1.2345e300) sink[0] = s.x;
}
__global__ __launch_bounds__(256, 2) void read_tall(const double* A, int64_t lda, int m, int k_chunk, double* sink) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l16 = lane & 15, lg = lane >> 4;
  const int r0 = blockIdx.x * 128 + wave * 32;
  const int kbeg = blockIdx.y * k_chunk, kend = min(m, kbeg + k_chunk);
  const double* ap = A + min(r0 + 2 * l16, m - 2);
  d2v s = {0, 0};
  for (int kc = kbeg; kc < kend; kc += 32) {
    d2v a[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) a[t] = *(const d2v*)(ap + (int64_t)min(kc + 4 * t + lg, m - 1) * lda);
#pragma unroll
    for (int t = 0; t < 8; ++t) s += a[t];
  }
  if (s.x + s.y == 1.2345e300) sink[0] = s.x;
}
// shader-clock monitor: one wave samples the core-clock counter (s_memtime) against the constant 100 MHz
// counter (s_memrealtime) every ~20 us for `dur_us`; run on a second stream next to the kernel under test
__global__ void clock_monitor(double* mhz, int nsamp, int dur_us) {
  if (threadIdx.x) return;
  const long long w0 = wall_clock64();
  long long wp = w0, cp = clock64();
  int i = 0;
  while (i < nsamp) {
    long long w = wall_clock64();
    if (w - wp >= 2000) {               // 20 us at 100 MHz
      long long c = clock64();
      mhz[i++] = (double)(c - cp) / (double)(w - wp) * 100.0;
      wp = w; cp = c;
    }
    if (w - w0 > (long long)dur_us * 100) break;
    __builtin_amdgcn_s_sleep(20);
  }
  for (; i < nsamp; ++i) mhz[i] = 0;
}
int main(int argc, char** argv) {
  bigkrls_ctx* ctx; if (bigkrls_ctx_create(0, &ctx)) { printf("%s\n", bigkrls_last_error()); return 1; }
  hipStream_t st = ctx->stream;
  const int64_t N = argc > 1 ? atoll(argv[1]) : 20000;
  double *A, *V, *Y0, *Y1; hipMalloc(&A, N * N * 8); hipMalloc(&V, N * 64 * 8); hipMalloc(&Y0, N * 64 * 8); hipMalloc(&Y1, N * 64 * 8);
  fillr<<<2048, 256, 0, st>>>(A, N * N, 1); fillr<<<2048, 256, 0, st>>>(V, N * 64, 2);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto timeit = [&](auto fn) {
    fn(); hipStreamSynchronize(st);
    hipEventRecord(e0, st); const int reps = 5;
    for (int r = 0; r < reps; ++r) fn();
    hipEventRecord(e1, st); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return (double)ms / reps * 1e3;
  };
  {
    hipStream_t s2; hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
    double* dm; const int ns = 400; hipMalloc(&dm, ns * 8); std::vector<double> hm(ns);
    const int64_t m = N - 64; const double* A22 = A + 64 + 64 * N;
    auto run = [&](const char* name, auto fn) {
      hipDeviceSynchronize();
      hipLaunchKernelGGL(clock_monitor, dim3(1), dim3(64), 0, s2, dm, ns, 6000);
      for (int r = 0; r < 5; ++r) fn();
      hipDeviceSynchronize();
      hipMemcpy(hm.data(), dm, ns * 8, hipMemcpyDeviceToHost);
      printf("clock while %-28s:", name);
      for (int i = 0; i < ns && hm[i] > 0; i += 12) printf(" %.0f", hm[i]);
      printf(" MHz\n");
    };
    run("idle", [&] {});
    run("A22 V (tall)", [&] { gemm(ctx, 0, 0, m, 64, m, 1.0, A22, N, V, m, 0.0, Y1, m); });
    run("read-only tall pattern", [&] { hipLaunchKernelGGL(read_tall, dim3(156, 4), dim3(256), 0, st, A22, N, (int)m, 5024, Y0); });
    {
      double *PA, *PB; hipMalloc(&PA, m * 512 * 8); hipMalloc(&PB, m * 512 * 8);
      fillr<<<2048, 256, 0, st>>>(PA, m * 512, 5); fillr<<<2048, 256, 0, st>>>(PB, m * 512, 6);
      run("syrk_mirror<64> k=128", [&] { syrk_mirror(ctx, m, 128, -1.0, PA, m, PB, m, A, N, 0, -1, true); });
      run("syrk_mirror<64> k=256", [&] { syrk_mirror(ctx, m, 256, -1.0, PA, m, PB, m, A, N, 0, -1, true); });
      hipFree(PA); hipFree(PB);
    }
    run("gemm NN 8192^3", [&] { gemm(ctx, 0, 0, 8192, 8192, 8192, 1.0, A, 8192, A + 8192 * 8192, 8192, 0.0, A + 2 * 8192 * 8192, 8192); });
    fillr<<<2048, 256, 0, st>>>(A, N * N, 1);
    hipDeviceSynchronize();
  }
  std::vector<double> h0, h1;
  // the trailing matrix of panel j: m = N - 64 (j + 1), at offset (N - m) (1 + N): every stage-1 shape and alignment
  for (int64_t m : {N - 64, N - 64 * 37, N * 3 / 4 / 64 * 64 + 2, N / 2 / 64 * 64, N / 4 / 64 * 64 - 2, (int64_t)4096, (int64_t)2048}) {
    if (m < 2048 || m > N - 64) continue;
    const int64_t off = N - m;
    const double* A22 = A + off + off * N;
    double flops = 2.0 * m * m * 64;
    int rc = 0;
    setenv("BIGKRLS_NOAVTALL", "1", 1);
    double t0 = timeit([&] { rc |= gemm(ctx, 0, 0, m, 64, m, 1.0, A22, N, V, m, 0.0, Y0, m); });
    unsetenv("BIGKRLS_NOAVTALL");
    double t1 = timeit([&] { rc |= gemm(ctx, 0, 0, m, 64, m, 1.0, A22, N, V, m, 0.0, Y1, m); });
    if (rc) { printf("error: %s\n", bigkrls_last_error()); return 1; }
    if (getenv("AV_ABLATE")) for (const char* ab : {"1", "2", "4", "6", "3", "7"}) {
      setenv("BIGKRLS_AVABL", ab, 1);
      double ta = timeit([&] { rc |= gemm(ctx, 0, 0, m, 64, m, 1.0, A22, N, V, m, 0.0, Y0, m); });
      unsetenv("BIGKRLS_AVABL");
      printf("          ablation %s (1 = no A loads, 2 = no V staging, 4 = no barriers): %8.1f us %6.2f TFLOP/s\n", ab, ta, flops / ta / 1e6);
    }
    double t2 = timeit([&] { hipLaunchKernelGGL(read_cols, dim3(2048), dim3(256), 0, st, A22, N, (int)m, Y0); });
    const int rb = (int)((m + 127) / 128), sp = std::max(1, (512 + rb - 1) / rb), kch = (int)(((m + sp - 1) / sp + 31) / 32 * 32);
    double t3 = timeit([&] { hipLaunchKernelGGL(read_tall, dim3(rb, (int)((m + kch - 1) / kch)), dim3(256), 0, st, A22, N, (int)m, kch, Y0); });
    printf("          read-only: whole columns %8.1f us %5.2f TB/s | tall pattern %8.1f us %5.2f TB/s\n", t2, m * m * 8.0 / t2 / 1e6, t3, m * m * 8.0 / t3 / 1e6);
    h0.resize(m * 64); h1.resize(m * 64);
    hipMemcpy(h0.data(), Y0, m * 64 * 8, hipMemcpyDeviceToHost); hipMemcpy(h1.data(), Y1, m * 64 * 8, hipMemcpyDeviceToHost);
    double md = 0, mx = 0;
    for (int64_t i = 0; i < m * 64; ++i) { md = fmax(md, fabs(h0[i] - h1[i])); mx = fmax(mx, fabs(h0[i])); }
    printf("m=%6lld  generic %8.1f us %6.2f TFLOP/s %5.2f TB/s | tall %8.1f us %6.2f TFLOP/s %5.2f TB/s | max diff %.2e (max |Y| %.2e)\n",
           (long long)m, t0, flops / t0 / 1e6, m * m * 8.0 / t0 / 1e6, t1, flops / t1 / 1e6, m * m * 8.0 / t1 / 1e6, md, mx);
  }
  return 0;
}
