// Diagnostic trace (off unless BIGKRLS_TRACE_DIR is set): a 64-bit position-dependent hash of a device or host
// buffer, appended with a tag to <dir>/pid<pid>.trace. The multi-GPU collectives log the hash of what they send and of
// what they deliver (csrc/dist.hip), the fit and the eigensolver log their replicated intermediate results, so that a
// wrong answer of a multi-rank run can be traced to the first buffer that differs between ranks -- or between two
// runs of the same fit in one process (tools/trace_diff.py). Tags: "C:" = result of a collective (equal on every rank),
// "R:" = replicated state (equal on every rank), "L:" = rank-local. Every call synchronises the stream it is given.
#include "common.h"

#include <unistd.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <mutex>

namespace bk {
namespace {

__host__ __device__ inline unsigned long long trace_mix(unsigned long long bits, unsigned long long index) {
  unsigned long long x = bits ^ (index * 0x9E3779B97F4A7C15ull + 1ull);
  x ^= x >> 30;
  x *= 0xBF58476D1CE4E5B9ull;
  x ^= x >> 27;
  x *= 0x94D049BB133111EBull;
  x ^= x >> 31;
  return x;
}

// sum over the elements of mix(bits, index) modulo 2^64: order-independent (integer addition), so the grid shape and
// the atomics do not matter
__global__ __launch_bounds__(256) void trace_hash_kernel(const unsigned long long* __restrict__ p, long long n,
                                                         unsigned long long* __restrict__ out) {
  __shared__ unsigned long long sh[256];
  unsigned long long acc = 0;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256)
    acc += trace_mix(p[i], (unsigned long long)i);
  sh[threadIdx.x] = acc;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) sh[threadIdx.x] += sh[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) atomicAdd(out, sh[0]);
}

struct Tracer {
  bool on = false;
  std::string dir;
  FILE* f = nullptr;
  long long seq = 0;
  std::mutex mu;
  struct Dev { unsigned long long* d = nullptr; unsigned long long* h = nullptr; };
  std::map<int, Dev> devs;
  Tracer() {
    if (const char* e = getenv("BIGKRLS_TRACE_DIR")) {
      if (*e) {
        dir = e;
        on = true;
      }
    }
  }
  FILE* file() {
    if (!f) {
      const std::string path = dir + "/pid" + std::to_string((long long)getpid()) + ".trace";
      f = fopen(path.c_str(), "a");
    }
    return f;
  }
};

Tracer& tracer() {
  static Tracer t;
  return t;
}

void trace_write(const char* tag, long long count, unsigned long long hash, long long extra) {
  Tracer& t = tracer();
  size_t free_b = 0, total_b = 0;
  (void)hipMemGetInfo(&free_b, &total_b);
  FILE* f = t.file();
  if (!f) return;
  fprintf(f, "%lld %s %lld %016llx %lld %lld\n", t.seq++, tag, count, hash, extra, (long long)(free_b >> 20));
  fflush(f);
}

}  // namespace

bool trace_on() { return tracer().on; }
bool trace_fine() {
  static const bool fine = tracer().on && getenv("BIGKRLS_TRACE_FINE") != nullptr;
  return fine;
}

int trace_point(bigkrls_ctx* ctx, hipStream_t st, const char* tag, const void* dev_ptr, int64_t count, int64_t extra) {
  Tracer& t = tracer();
  if (!t.on) return BIGKRLS_OK;
  std::lock_guard<std::mutex> lock(t.mu);
  if (!dev_ptr || count <= 0) {
    trace_write(tag, 0, 0, extra);
    return BIGKRLS_OK;
  }
  Tracer::Dev& dv = t.devs[ctx->device];
  if (!dv.d) {
    BK_HIP(hipMalloc((void**)&dv.d, sizeof(unsigned long long)));
    BK_HIP(hipHostMalloc((void**)&dv.h, sizeof(unsigned long long), hipHostMallocDefault));
  }
  BK_HIP(hipMemsetAsync(dv.d, 0, sizeof(unsigned long long), st));
  const int blocks = (int)std::min<int64_t>((count + 255) / 256, 1024);
  hipLaunchKernelGGL(trace_hash_kernel, dim3(blocks), dim3(256), 0, st, (const unsigned long long*)dev_ptr, (long long)count, dv.d);
  BK_CHECK_LAUNCH();
  BK_HIP(hipMemcpyAsync(dv.h, dv.d, sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
  BK_HIP(hipStreamSynchronize(st));
  trace_write(tag, count, *dv.h, extra);
  return BIGKRLS_OK;
}

int trace_host(const char* tag, const void* host_ptr, int64_t count, int64_t extra) {
  Tracer& t = tracer();
  if (!t.on) return BIGKRLS_OK;
  std::lock_guard<std::mutex> lock(t.mu);
  unsigned long long h = 0;
  const unsigned long long* p = (const unsigned long long*)host_ptr;
  for (int64_t i = 0; p && i < count; ++i) h += trace_mix(p[i], (unsigned long long)i);
  trace_write(tag, p ? count : 0, h, extra);
  return BIGKRLS_OK;
}

}  // namespace bk
