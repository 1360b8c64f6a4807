// Micro-benchmark of the tridiagonalisation symv kernels (development tool).
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/symv_bench.hip -o tools/symv_bench \
//        -Lbigkrls_amd -lbigkrls_hip -Wl,-rpath,$PWD/bigkrls_amd
#include "../bigkrls_amd/csrc/eigen.hip"
#include <cstdio>
#include <vector>
using namespace bk;

__global__ void fill_sym(double* W, int n) {
  const int64_t total = (int64_t)n * n;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int r = e % n, c = e / n;
    const int a = r < c ? r : c, b = r < c ? c : r;
    W[e] = 1.0 / (1.0 + ((a * 131 + b * 7) % 1000)) - 0.3;
  }
}
__global__ void read_bw(const double2* p, int64_t n2, double* out) {
  double s = 0;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n2; e += (int64_t)gridDim.x * blockDim.x) {
    double2 v = p[e]; s += v.x + v.y;
  }
  if (s == 1.2345) out[0] = s;
}

int main(int argc, char** argv) {
  int n = argc > 1 ? atoi(argv[1]) : 20000;
  int c = argc > 2 ? atoi(argv[2]) : 0;
  int reps = 5;
  bigkrls_ctx* ctx;
  if (bigkrls_ctx_create(0, &ctx)) { printf("ctx: %s\n", bigkrls_last_error()); return 1; }
  hipStream_t st = ctx->stream;
  const int64_t N = n;
  double *W, *P1, *P2, *scr, *U;
  hipMalloc(&W, N * N * 8); hipMalloc(&P1, 2 * N * TRD_NB * 8); hipMalloc(&P2, 2 * N * TRD_NB * 8);
  hipMalloc(&scr, (5 * N + 4096) * 8);
  const int64_t sv_prow = (N / SV_CW + 2) * N, sv_pcol = 10 * N, sv_prow2 = (N / (SV_CW * 32) + 2) * N;
  hipMalloc(&U, (sv_prow + sv_pcol + sv_prow2 + 4096) * 8);
  hipMemset(P1, 0, 2 * N * TRD_NB * 8); hipMemset(P2, 0, 2 * N * TRD_NB * 8);
  fill_sym<<<4096, 256, 0, st>>>(W, n);
  SymvWs sw{U, U + sv_prow, U + sv_prow + sv_pcol, U + sv_prow + sv_pcol + sv_prow2};
  double* y = scr; double* tvec = y + n; double* part1 = tvec + 2 * TRD_NB; double* e = part1 + 1024; double* tau = e + n; double* d = tau + n;
  double* part2 = d + n;
  const int i = argc > 3 ? atoi(argv[3]) : 0, pw = 64;
  const int nb1 = (n - c + 255) / 256;
  trd_k1<<<nb1, 256, 0, st>>>(W, n, c, i, pw, P1, P2, part2, 0, tau, d, part1);
  const int L = n - c - 1;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto timeit = [&](const char* name, double bytes, auto fn) {
    fn(); hipStreamSynchronize(st);
    hipEventRecord(e0, st);
    for (int r = 0; r < reps; ++r) fn();
    hipEventRecord(e1, st); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
    printf("%-28s %9.1f us  %8.1f GB/s\n", name, ms * 1e3, bytes / 1e9 / (ms / 1e3));
  };
  double* out; hipMalloc(&out, 8);
  timeit("stream read (full matrix)", 8.0 * N * N, [&] { read_bw<<<2048, 256, 0, st>>>((const double2*)W, N * N / 2, out); });
  timeit("stream read (half)", 4.0 * N * N, [&] { read_bw<<<2048, 256, 0, st>>>((const double2*)W, N * N / 4, out); });
  const int vec = 2, CH = 64 * vec;
  const int nstrips = (L + SV_CW - 1) / SV_CW;
  for (int div : {2, 4, 6, 12, 24}) {
    const int rsq = 4 * CH * BK_SV_NCH;
    int RS = ((L / div + rsq - 1) / rsq) * rsq;
    RS = std::max(rsq, std::min(RS, 16384));
    const int nsegmax = (L + 1 + RS - 1) / RS;
    char nm[64]; snprintf(nm, 64, "tiled symv RS=%d", RS);
    timeit(nm, 4.0 * L * (L + 1.0), [&] {
      trd_symv_tiles<2><<<dim3(nstrips + (2 * i + 15) / 16, nsegmax), 256, 0, st>>>(W, n, c, i, pw, P1, part1, nb1, RS, nstrips, sw.Prow, sw.Pcol, sw.Ppan, e, tau);
    });
    if (nsegmax > 9) break;
  }
  {
    const int rsq = 4 * CH * BK_SV_NCH; int RS = ((L / 6 + rsq - 1) / rsq) * rsq; RS = std::max(rsq, std::min(RS, 8192));
    const int nq = (nstrips + 31) / 32, nb3 = (L + 255) / 256;
    timeit("symv_reduce", 8.0 * L * nstrips / 2, [&] {
      trd_symv_reduce<<<dim3(nb3, nq), 256, 0, st>>>(W, n, c, i, pw, RS, vec, part1, nb1, sw.Prow, sw.Pcol, sw.Prow2, P1, P2);
    });
  }
  return 0;
}
