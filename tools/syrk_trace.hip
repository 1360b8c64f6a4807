// Where does the time of the trailing update go? (development probe, needs the traced build of the library)
// Every workgroup of syrk_mirror_kernel<64> stamps the 100 MHz clock at its start, after its MFMA loop and at its
// end; this program prints the distribution of the two phase lengths and, in 5 us bins, how many workgroups of the
// whole GPU are in each phase -- a convoy (all workgroups in the same phase) shows as the two counts alternating.
// Build + run: tools/syrk_trace.sh
#include "../bigkrls_amd/csrc/common.h"
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <map>
using namespace bk;
extern "C" int bk_syrk_trace_set(unsigned long long* p);
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
  const int k = argc > 2 ? atoi(argv[2]) : 128;
  double *C, *A, *B; hipMalloc(&C, n * n * 8); hipMalloc(&A, n * 512 * 8); hipMalloc(&B, n * 512 * 8);
  fillr<<<2048, 256, 0, st>>>(C, n * n, 1); fillr<<<2048, 256, 0, st>>>(A, n * 512, 2); fillr<<<2048, 256, 0, st>>>(B, n * 512, 3);
  const int64_t tiles = (n + 127) / 128, nwg = tiles * (tiles + 1);   // 128 x 64 tiles of the lower triangle
  unsigned long long* tr; hipMalloc(&tr, (nwg + 1024) * 32); hipMemset(tr, 0, (nwg + 1024) * 32);
  syrk_mirror(ctx, n, k, -1.0, A, n, B, n, C, n, 0, -1, true); hipStreamSynchronize(st);   // warm
  if (bk_syrk_trace_set(tr)) { printf("trace_set failed\n"); return 1; }
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0, st);
  syrk_mirror(ctx, n, k, -1.0, A, n, B, n, C, n, 0, -1, true);
  hipEventRecord(e1, st); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> h((nwg + 1024) * 4);
  hipMemcpy(h.data(), tr, h.size() * 8, hipMemcpyDeviceToHost);
  int64_t cnt = 0; unsigned long long tmin = ~0ull, tmax = 0;
  for (int64_t w = 0; w < nwg + 1024; ++w) if (h[4 * w]) { ++cnt; tmin = std::min(tmin, h[4 * w]); tmax = std::max(tmax, h[4 * w + 2]); }
  printf("m=%lld k=%d: %.1f us by events, %lld workgroups stamped, span %.1f us\n", (long long)n, k, ms * 1e3, (long long)cnt, (tmax - tmin) * 0.01);
  std::vector<double> tk, te; std::map<unsigned long long, int> cus;
  for (int64_t w = 0; w < nwg + 1024; ++w) if (h[4 * w]) {
    tk.push_back((h[4 * w + 1] - h[4 * w]) * 0.01); te.push_back((h[4 * w + 2] - h[4 * w + 1]) * 0.01);
    cus[((h[4 * w + 3] >> 32) << 16) | ((h[4 * w + 3] >> 8) & 0xffff)]++;
  }
  auto q = [](std::vector<double> v, double p) { std::sort(v.begin(), v.end()); return v[(size_t)(p * (v.size() - 1))]; };
  double sk = 0, se = 0; for (double v : tk) sk += v; for (double v : te) se += v;
  printf("MFMA loop  (us): p10 %.1f  p50 %.1f  p90 %.1f  mean %.1f\n", q(tk, .1), q(tk, .5), q(tk, .9), sk / tk.size());
  printf("epilogue   (us): p10 %.1f  p50 %.1f  p90 %.1f  mean %.1f\n", q(te, .1), q(te, .5), q(te, .9), se / te.size());
  printf("distinct (xcc, se/sh/cu) ids: %zu; mean resident workgroups = (sum of lifetimes)/(span x ids) = %.2f\n", cus.size(),
         (sk + se) / ((tmax - tmin) * 0.01 * cus.size()));
  const double bin = 5.0; const int nb = (int)((tmax - tmin) * 0.01 / bin) + 1;
  std::vector<double> ink(nb, 0.0), ine(nb, 0.0);
  auto add = [&](std::vector<double>& v, unsigned long long a, unsigned long long b) {
    const double x0 = (a - tmin) * 0.01, x1 = (b - tmin) * 0.01;
    for (int i = (int)(x0 / bin); i <= (int)(x1 / bin) && i < nb; ++i) {
      const double lo = std::max(x0, i * bin), hi = std::min(x1, (i + 1) * bin);
      if (hi > lo) v[i] += (hi - lo) / bin;
    }
  };
  for (int64_t w = 0; w < nwg + 1024; ++w) if (h[4 * w]) { add(ink, h[4 * w], h[4 * w + 1]); add(ine, h[4 * w + 1], h[4 * w + 2]); }
  printf("workgroups of the GPU in each phase, %g us bins (first 60, then every 10th):\n  t_us   in_MFMA  in_epilogue\n", bin);
  for (int i = 0; i < nb; ++i) if (i < 60 || i % 10 == 0) printf("%6.0f  %7.1f  %7.1f\n", i * bin, ink[i], ine[i]);
  return 0;
}
