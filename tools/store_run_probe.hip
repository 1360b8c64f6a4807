// HBM write rate against the length of the contiguous run a wave instruction writes (development probe).
// An n x n column-major fp64 matrix is written once; a workgroup of 256 threads owns a 128 x 128 tile and writes it
// either as 128-byte runs (16 rows of 8 columns per wave instruction: the pattern of an MFMA accumulator store with
// paired rows), 256-, 512-byte runs or whole 1-KB column pieces (128 rows of one column per wave instruction).
// hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/store_run_probe.hip -o tools/store_run_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef double d2v __attribute__((ext_vector_type(2)));
template <int RUN_ROWS, bool NT>   // rows per contiguous run: 16, 32, 64, 128
__global__ __launch_bounds__(256) void wr(double* out, long n, int tiles) {
  const int t = blockIdx.x, tm = t % tiles, tn = t / tiles;
  const long m0 = tm * 128L, n0 = tn * 128L;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  constexpr int PAIRS = RUN_ROWS / 2;          // lanes per run (16 B each)
  constexpr int RUNS = 64 / PAIRS;             // runs per wave instruction
  // the wave's 64 x 64... simply: the tile's 128 x 128 elements = 8192 row pairs; 256 lanes x 32 instructions
  for (int it = 0; it < 32; ++it) {
    // instruction `it` of wave `wave`: RUNS runs of RUN_ROWS rows
    const int run = lane / PAIRS, pr = lane % PAIRS;
    const int g = (it * 4 + wave) * RUNS + run;          // global run index inside the tile: 0 .. 128*128/RUN_ROWS
    const int runs_per_col = 128 / RUN_ROWS;
    const int col = g / runs_per_col, rr = (g % runs_per_col) * RUN_ROWS + 2 * pr;
    d2v v; v.x = (double)col; v.y = (double)rr;
    double* dst = out + (m0 + rr) + (n0 + col) * n;
    if (m0 + rr + 1 < n && n0 + col < n) {
      if (NT) __builtin_nontemporal_store(v, reinterpret_cast<d2v*>(dst));
      else *reinterpret_cast<d2v*>(dst) = v;
    }
  }
}
int main(int argc, char** argv) {
  const long n = argc > 1 ? atol(argv[1]) : 20000;
  double* d; hipMalloc(&d, n * n * 8);
  const int tiles = (int)((n + 127) / 128);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto run = [&](const char* name, auto kern) {
    hipLaunchKernelGGL(kern, dim3(tiles * tiles), dim3(256), 0, 0, d, n, tiles); hipDeviceSynchronize();
    hipEventRecord(e0); for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(kern, dim3(tiles * tiles), dim3(256), 0, 0, d, n, tiles);
    hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    printf("%-28s n=%ld: %8.1f us  %6.0f GB/s\n", name, n, ms * 1e3, 8.0 * n * n / ms / 1e6);
  };
  run("128-B runs", wr<16, false>); run("128-B runs, nt", wr<16, true>);
  run("256-B runs", wr<32, false>); run("256-B runs, nt", wr<32, true>);
  run("512-B runs", wr<64, false>); run("512-B runs, nt", wr<64, true>);
  run("1-KB runs", wr<128, false>); run("1-KB runs, nt", wr<128, true>);
  hipMemsetAsync(d, 0, n * n * 8); hipDeviceSynchronize();
  hipEventRecord(e0); for (int r = 0; r < 5; ++r) hipMemsetAsync(d, 0, n * n * 8);
  hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
  printf("%-28s n=%ld: %8.1f us  %6.0f GB/s\n", "hipMemsetAsync", n, ms * 1e3, 8.0 * n * n / ms / 1e6);
  return 0;
}
