// Which compute units does a CU-masked stream (hipExtStreamCreateWithCUMask) run on? (development probe)
// Launches 2 x nbits workgroups that each hold 100 KB of LDS (one per CU) and spin ~100 us on a stream whose mask has
// the low `nbits` bits set, and prints per XCC the distinct (se, cu) pairs that ran a workgroup.
// hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/cumask_probe.hip -o tools/cumask_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <set>
#include <vector>
__global__ __launch_bounds__(256) void who(unsigned* out, long spin) {
  __shared__ double hold[12800];
  hold[threadIdx.x] = threadIdx.x;
  __syncthreads();
  unsigned xcc = __builtin_amdgcn_s_getreg((20) | (0 << 6) | (31 << 11));   // HW_REG_XCC_ID, offset 0, size 32
  unsigned hwid = __builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11));   // HW_REG_HW_ID
  long t0 = wall_clock64();
  while (wall_clock64() - t0 < spin) __builtin_amdgcn_s_sleep(8);
  if (threadIdx.x == 0) { out[2 * blockIdx.x] = xcc; out[2 * blockIdx.x + 1] = hwid + (unsigned)(hold[5] > 1e9); }
}
int main(int argc, char** argv) {
  const int nbits = argc > 1 ? atoi(argv[1]) : 64;
  const int hi = argc > 2 ? atoi(argv[2]) : 0;     // 1: the complement (bits nbits .. 255)
  uint32_t mask[8] = {0};
  for (int i = 0; i < 256; ++i) if ((i < nbits) != (hi != 0)) mask[i / 32] |= 1u << (i % 32);
  hipStream_t st;
  if (hipExtStreamCreateWithCUMask(&st, 8, mask) != hipSuccess) { printf("hipExtStreamCreateWithCUMask failed\n"); return 1; }
  const int nwg = 2 * (hi ? 256 - nbits : nbits);
  unsigned* d; hipMalloc(&d, nwg * 8);
  hipLaunchKernelGGL(who, dim3(nwg), dim3(256), 0, st, d, 10000L);   // 100 MHz clock: 100 us
  hipStreamSynchronize(st);
  std::vector<unsigned> h(2 * nwg); hipMemcpy(h.data(), d, nwg * 8, hipMemcpyDeviceToHost);
  std::map<unsigned, std::set<unsigned>> per;
  for (int i = 0; i < nwg; ++i) {
    const unsigned xcc = h[2 * i] & 0xf, id = h[2 * i + 1];
    const unsigned cu = (id >> 8) & 0xf, sh = (id >> 12) & 1, se = (id >> 13) & 0x7;
    per[xcc].insert(se * 100 + sh * 16 + cu);
  }
  printf("mask: %s %d bits; %d workgroups\n", hi ? "all but the low" : "low", nbits, nwg);
  int total = 0;
  for (auto& kv : per) {
    printf("  xcc %u: %zu CUs:", kv.first, kv.second.size());
    for (unsigned v : kv.second) printf(" se%u.cu%u", v / 100, v % 100);
    printf("\n");
    total += (int)kv.second.size();
  }
  printf("  distinct CUs: %d\n", total);
  return 0;
}
