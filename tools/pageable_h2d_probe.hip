// When does hipMemcpyAsync read a PAGEABLE host source: when the call returns, or when the copy executes on the stream?
// The stream is kept busy by a spin kernel; the source is overwritten right after hipMemcpyAsync returns; the device
// buffer tells which version travelled. (The library's divide & conquer uploaded descriptor vectors this way and one of
// them went out of scope before the stream was synchronised: DESIGN.md section 7.)
//   hipcc --offload-arch=gfx950 -O2 -o tools/pageable_h2d_probe tools/pageable_h2d_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void spin(long long ticks_100mhz) {
  const unsigned long long t0 = wall_clock64();
  while ((long long)(wall_clock64() - t0) < ticks_100mhz) {}
}
int main() {
  hipStream_t st;
  CK(hipStreamCreate(&st));
  const size_t sizes[] = {512, 2048, 6144, 24576, 65536, 1 << 20, 8 << 20, 64 << 20};
  for (size_t bytes : sizes) {
    int late = 0, reps = 5;
    for (int rep = 0; rep < reps; ++rep) {
      unsigned char* h = (unsigned char*)malloc(bytes);
      unsigned char* d = nullptr;
      CK(hipMalloc((void**)&d, bytes));
      memset(h, 1, bytes);
      hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, st, 300000LL);     // 3 ms ahead of the copy on the stream
      CK(hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, st));
      memset(h, 2, bytes);                                              // the source changes as soon as the call returns
      CK(hipStreamSynchronize(st));
      std::vector<unsigned char> back(bytes);
      CK(hipMemcpy(back.data(), d, bytes, hipMemcpyDeviceToHost));
      size_t twos = 0;
      for (size_t i = 0; i < bytes; ++i) twos += back[i] == 2;
      if (twos) ++late;
      if (rep == 0) printf("%9zu bytes: %zu of %zu bytes arrived with the value written AFTER hipMemcpyAsync returned\n", bytes, twos, bytes);
      CK(hipFree(d));
      free(h);
    }
    printf("%9zu bytes: source read late in %d of %d runs\n", bytes, late, reps);
  }
  // ---- a burst: many asynchronous copies from different pageable buffers queued behind one busy stream (what one level
  //      of the divide & conquer does: ~12 uploads of n doubles / ints back to back), every source overwritten at once
  for (size_t bytes : {(size_t)7200, (size_t)24576, (size_t)160000, (size_t)1 << 20}) {
    const int NB = 24;
    int wrong = 0;
    for (int rep = 0; rep < 20; ++rep) {
      std::vector<unsigned char*> h(NB), d(NB);
      for (int i = 0; i < NB; ++i) {
        h[i] = (unsigned char*)malloc(bytes);
        CK(hipMalloc((void**)&d[i], bytes));
        memset(h[i], 10 + i, bytes);
      }
      hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, st, 500000LL);     // 5 ms
      for (int i = 0; i < NB; ++i) CK(hipMemcpyAsync(d[i], h[i], bytes, hipMemcpyHostToDevice, st));
      for (int i = 0; i < NB; ++i) memset(h[i], 200, bytes);
      CK(hipStreamSynchronize(st));
      std::vector<unsigned char> back(bytes);
      for (int i = 0; i < NB; ++i) {
        CK(hipMemcpy(back.data(), d[i], bytes, hipMemcpyDeviceToHost));
        size_t bad = 0;
        for (size_t k = 0; k < bytes; ++k) bad += back[k] != (unsigned char)(10 + i);
        if (bad) {
          ++wrong;
          if (wrong <= 3) printf("burst of %d x %zu bytes: copy %d arrived with %zu wrong bytes (first byte %d)\n", NB, bytes, i, bad, (int)back[0]);
        }
        CK(hipFree(d[i]));
        free(h[i]);
      }
    }
    printf("burst of %d copies x %zu bytes behind a busy stream, 20 repetitions: %d copies arrived wrong\n", NB, bytes, wrong);
  }
  return 0;
}
