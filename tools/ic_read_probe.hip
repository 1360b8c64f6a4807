// Read-only ceiling for a buffer that stays in the Infinity Cache between passes (development probe): the solveforc
// probe kernel reads Q (N x K doubles: 40 MB at the bench size) once per probe, 39 probes back to back.
// hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/ic_read_probe.hip -o tools/ic_read_probe ; tools/ic_read_probe [MB]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef double d2 __attribute__((ext_vector_type(2)));
__global__ __launch_bounds__(256) void rd(const d2* __restrict__ p, size_t n2, double* __restrict__ out) {
  double acc = 0.0;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n2; i += (size_t)gridDim.x * 256) {
    const d2 v = p[i];
    acc += v.x + v.y;
  }
  if (acc == 1.2345e300) out[0] = acc;   // (keeps the loads alive)
}
int main(int argc, char** argv) {
  const size_t mb = argc > 1 ? atoi(argv[1]) : 40;
  const size_t bytes = mb << 20, n2 = bytes / 16;
  d2* p; double* out;
  if (hipMalloc(&p, bytes) != hipSuccess || hipMalloc(&out, 8) != hipSuccess) return 1;
  (void)hipMemset(p, 0, bytes);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int grid : {625, 1250, 2048, 4096, 8192}) {
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(rd, dim3(grid), dim3(256), 0, 0, p, n2, out);
    (void)hipEventRecord(e0);
    const int reps = 39;
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(rd, dim3(grid), dim3(256), 0, 0, p, n2, out);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%zu MB, grid %5d: %.2f us per pass = %.0f GB/s\n", mb, grid, 1e3 * ms / reps, bytes / (ms / reps * 1e-3) / 1e9);
  }
  return 0;
}
