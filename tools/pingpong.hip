// Round-trip latency of an 8-byte message between two workgroups through global memory, by memory
// scope of the accesses and by placement (same XCD / different XCDs) -- development probe.
// hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/pingpong.hip -o tools/pingpong
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

template <int SCOPE>
__device__ __forceinline__ unsigned long long ld(const unsigned long long* p) {
  if (SCOPE == 0) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (SCOPE == 1) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  if (SCOPE == 2) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  return *(const volatile unsigned long long*)p;
}
template <int SCOPE>
__device__ __forceinline__ void st(unsigned long long* p, unsigned long long v) {
  if (SCOPE == 0) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  else if (SCOPE == 1) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  else if (SCOPE == 2) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  else *(volatile unsigned long long*)p = v;
}

// blocks a and b play; everybody else exits. box[0]: a -> b, box[16]: b -> a (separate cache lines)
template <int SCOPE>
__global__ void pingpong(unsigned long long* box, int a, int b, int rounds, long long* cycles, int* xcc,
                         int* fail) {
  const int bid = blockIdx.x;
  if (bid != a && bid != b) return;
  if (threadIdx.x != 0) return;
  const int me = (bid == a) ? 0 : 1;
  xcc[me] = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 15;   // HW_REG_XCC_ID[3:0]
  unsigned long long* out = box + (me == 0 ? 0 : 16);
  const unsigned long long* in = box + (me == 0 ? 16 : 0);
  const long long t0 = wall_clock64();
  for (int r = 1; r <= rounds; ++r) {
    if (me == 0) st<SCOPE>(out, (unsigned long long)r);
    long spins = 0;
    while (ld<SCOPE>(in) != (unsigned long long)r) {
      if (++spins > 20000000) { *fail = 1; return; }
    }
    if (me == 1) st<SCOPE>(out, (unsigned long long)r);
  }
  if (me == 0) *cycles = wall_clock64() - t0;
}

template <int SCOPE>
void run(const char* name, unsigned long long* box, long long* cyc, int* xcc, int* fail, int a, int b) {
  const int rounds = 20000;
  hipMemset(box, 0, 64 * 8);
  hipMemset(fail, 0, 4);
  hipLaunchKernelGGL(pingpong<SCOPE>, dim3(256), dim3(64), 0, 0, box, a, b, rounds, cyc, xcc, fail);
  hipDeviceSynchronize();
  long long h = 0; int hx[2] = {0, 0}, hf = 0;
  hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
  hipMemcpy(hx, xcc, 8, hipMemcpyDeviceToHost);
  hipMemcpy(&hf, fail, 4, hipMemcpyDeviceToHost);
  // wall_clock64 ticks at 100 MHz
  if (hf) printf("%-10s blocks %3d,%3d (XCC %d,%d): no progress (stale reads)\n", name, a, b, hx[0], hx[1]);
  else printf("%-10s blocks %3d,%3d (XCC %d,%d): %.3f us per round trip\n", name, a, b, hx[0], hx[1],
              (double)h / 100.0 / rounds);
}

int main() {
  unsigned long long* box; long long* cyc; int *xcc, *fail;
  hipMalloc(&box, 64 * 8); hipMalloc(&cyc, 8); hipMalloc(&xcc, 8); hipMalloc(&fail, 4);
  const int pairs[3][2] = {{0, 8}, {0, 1}, {3, 4}};
  for (auto& p : pairs) {
    run<0>("agent", box, cyc, xcc, fail, p[0], p[1]);
    run<1>("workgroup", box, cyc, xcc, fail, p[0], p[1]);
    run<2>("system", box, cyc, xcc, fail, p[0], p[1]);
    run<3>("volatile", box, cyc, xcc, fail, p[0], p[1]);
  }
  return 0;
}
