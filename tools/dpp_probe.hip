// Lane-exchange primitives used by bc_regwin's column-sum butterfly, checked on the device (development probe):
// xor-1 / xor-2 (quad_perm), xor-4 (row_shl:4 / row_shr:4 under bank masks), xor-8 (row_ror:8), xor-16 / xor-32
// (ds_bpermute).   hipcc -O3 --offload-arch=gfx950 tools/dpp_probe.hip -o tools/dpp_probe
#include <hip/hip_runtime.h>
#include <cstdio>
template <int CTRL, int BANK = 0xF>
__device__ __forceinline__ double dppm(double v, double old) {
  const int lo = __builtin_amdgcn_update_dpp(__double2loint(old), __double2loint(v), CTRL, 0xF, BANK, false);
  const int hi = __builtin_amdgcn_update_dpp(__double2hiint(old), __double2hiint(v), CTRL, 0xF, BANK, false);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double bperm(double v, int srclane) {
  const int lo = __builtin_amdgcn_ds_bpermute(srclane << 2, __double2loint(v));
  const int hi = __builtin_amdgcn_ds_bpermute(srclane << 2, __double2hiint(v));
  return __hiloint2double(hi, lo);
}
__global__ void k(double* out) {
  const int l = threadIdx.x;
  const double v = (double)l;
  out[l] = dppm<0xB1>(v, v);
  out[64 + l] = dppm<0x4E>(v, v);
  double x = dppm<0x104, 0x5>(v, v);
  x = dppm<0x114, 0xA>(v, x);
  out[128 + l] = x;
  out[192 + l] = dppm<0x128>(v, v);
  out[256 + l] = bperm(v, l ^ 16);
  out[320 + l] = bperm(v, l ^ 32);
}
int main() {
  double* d; hipMalloc(&d, 384 * 8);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  double h[384]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
  const int x[6] = {1, 2, 4, 8, 16, 32};
  int bad = 0;
  for (int s = 0; s < 6; ++s)
    for (int l = 0; l < 64; ++l)
      if ((int)h[64 * s + l] != (l ^ x[s])) { if (bad < 8) printf("xor %d lane %d got %d\n", x[s], l, (int)h[64 * s + l]); ++bad; }
  printf(bad ? "dpp probe: %d MISMATCHES\n" : "dpp probe OK (%d)\n", bad);
  return bad != 0;
}
