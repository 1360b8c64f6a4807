// Test infrastructure, not product code: a stand-in for librccl that lets SEVERAL RANKS SHARE ONE GPU.
//
// RCCL refuses two ranks on one device and this pool hands out single-GPU boxes, so the library's RCCL path
// (csrc/dist.hip: dlopen of the library named by BIGKRLS_RCCL_LIB, the function table, in-place all-reduce /
// equal-count all-gather / in-place broadcast of doubles on the context's stream) would otherwise never run with
// more than one rank. This library exports the seven entry points dist.hip binds and implements them over a POSIX
// shared-memory segment between the rank processes. Like RCCL's, its collectives are STREAM-ORDERED and
// asynchronous to the calling host thread: each one enqueues on the caller's stream
//     device -> pinned copy,  a host function that meets the peers in the segment,  pinned -> device copy,
// so code that reads a result without synchronising the stream fails here as it would with RCCL.
// Sums are formed in rank order by every rank (bit-identical results on all ranks).
//
// Build: make -C tests/mock_rccl   (g++ against the HIP runtime; no device code)
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <fcntl.h>
#include <sched.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

namespace {

constexpr size_t SLOT_BYTES = 16u << 20;      // per-rank slot of the segment = largest piece of one exchange
constexpr int MAX_RANKS = 16;

struct Header {
  std::atomic<int> ready;                     // set by rank 0 once the header is initialised
  std::atomic<int> arrived;                   // sense-reversing barrier
  std::atomic<int> sense;
  std::atomic<int> attached;
  int nranks;
};

struct Comm {
  int nranks = 0, rank = 0;
  std::string name;
  size_t map_bytes = 0;
  Header* hdr = nullptr;
  char* slots = nullptr;                      // nranks x SLOT_BYTES
  double* pin = nullptr;                      // pinned staging: MAX(nranks, 1) x SLOT_BYTES
  int local_sense = 0;
};

struct Op {                                    // one host-function invocation (owned by it)
  Comm* c;
  int kind;                                    // 0 all-reduce sum, 1 all-reduce min, 2 all-gather, 3 broadcast
  size_t count;                                // doubles per rank in this piece
  int root;
};

bool barrier(Comm* c) {
  Header* h = c->hdr;
  c->local_sense ^= 1;
  if (h->arrived.fetch_add(1, std::memory_order_acq_rel) + 1 == c->nranks) {
    h->arrived.store(0, std::memory_order_relaxed);
    h->sense.store(c->local_sense, std::memory_order_release);
    return true;
  }
  timespec t0;
  clock_gettime(CLOCK_MONOTONIC, &t0);
  long spins = 0;
  while (h->sense.load(std::memory_order_acquire) != c->local_sense) {
    if ((++spins & 1023) == 0) {
      sched_yield();
      timespec t1;
      clock_gettime(CLOCK_MONOTONIC, &t1);
      if (t1.tv_sec - t0.tv_sec > 300) {       // a peer died: do not hang the test session for ever
        fprintf(stderr, "[mock_rccl] rank %d: barrier timed out\n", c->rank);
        abort();
      }
    }
  }
  return true;
}

void exchange(void* arg) {
  Op* op = static_cast<Op*>(arg);
  Comm* c = op->c;
  const size_t n = op->count;
  double* mine = reinterpret_cast<double*>(c->slots + (size_t)c->rank * SLOT_BYTES);
  if (op->kind != 3 || c->rank == op->root) std::memcpy(mine, c->pin, n * sizeof(double));
  barrier(c);
  if (op->kind == 0 || op->kind == 1) {
    for (size_t i = 0; i < n; ++i) {
      double acc = reinterpret_cast<const double*>(c->slots)[i];
      for (int r = 1; r < c->nranks; ++r) {
        const double v = reinterpret_cast<const double*>(c->slots + (size_t)r * SLOT_BYTES)[i];
        acc = (op->kind == 0) ? acc + v : (v < acc ? v : acc);
      }
      c->pin[i] = acc;
    }
  } else if (op->kind == 2) {
    for (int r = 0; r < c->nranks; ++r)
      std::memcpy(c->pin + (size_t)r * n, c->slots + (size_t)r * SLOT_BYTES, n * sizeof(double));
  } else {
    std::memcpy(c->pin, c->slots + (size_t)op->root * SLOT_BYTES, n * sizeof(double));
  }
  barrier(c);                                  // nobody overwrites a slot before everybody has read it
  delete op;
}

#define MOCK_HIP(expr)                                                                   \
  do {                                                                                   \
    hipError_t e_ = (expr);                                                              \
    if (e_ != hipSuccess) {                                                              \
      fprintf(stderr, "[mock_rccl] %s: %s\n", #expr, hipGetErrorString(e_));             \
      return ncclUnhandledCudaError;                                                     \
    }                                                                                    \
  } while (0)

}  // namespace

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId* id) {
  if (!id) return ncclInvalidArgument;
  std::memset(id, 0, sizeof *id);
  timespec ts;
  clock_gettime(CLOCK_REALTIME, &ts);
  snprintf(id->internal, sizeof id->internal, "/bkmockrccl_%d_%ld_%ld", (int)getpid(), (long)ts.tv_sec, (long)ts.tv_nsec);
  return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t* out, int nranks, ncclUniqueId id, int rank) {
  if (!out || nranks < 1 || nranks > MAX_RANKS || rank < 0 || rank >= nranks) return ncclInvalidArgument;
  Comm* c = new Comm();
  c->nranks = nranks;
  c->rank = rank;
  c->name = std::string(id.internal, strnlen(id.internal, sizeof id.internal));
  c->map_bytes = 4096 + (size_t)nranks * SLOT_BYTES;
  int fd = -1;
  if (rank == 0) {
    fd = shm_open(c->name.c_str(), O_CREAT | O_EXCL | O_RDWR, 0600);
    if (fd < 0 || ftruncate(fd, (off_t)c->map_bytes) != 0) { perror("[mock_rccl] shm_open/ftruncate"); return ncclSystemError; }
  } else {
    for (int tries = 0; tries < 60000 && fd < 0; ++tries) {        // up to ~60 s for rank 0 to get here
      fd = shm_open(c->name.c_str(), O_RDWR, 0600);
      struct stat sb;
      if (fd >= 0 && (fstat(fd, &sb) != 0 || (size_t)sb.st_size < c->map_bytes)) { close(fd); fd = -1; }
      if (fd < 0) usleep(1000);
    }
    if (fd < 0) { fprintf(stderr, "[mock_rccl] rank %d: no segment %s\n", rank, c->name.c_str()); return ncclSystemError; }
  }
  void* p = mmap(nullptr, c->map_bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  if (p == MAP_FAILED) { perror("[mock_rccl] mmap"); return ncclSystemError; }
  c->hdr = static_cast<Header*>(p);
  c->slots = static_cast<char*>(p) + 4096;
  if (rank == 0) {
    c->hdr->arrived.store(0);
    c->hdr->sense.store(0);
    c->hdr->attached.store(0);
    c->hdr->nranks = nranks;
    c->hdr->ready.store(1, std::memory_order_release);
  } else {
    while (c->hdr->ready.load(std::memory_order_acquire) != 1) usleep(100);
    if (c->hdr->nranks != nranks) return ncclInvalidArgument;
  }
  MOCK_HIP(hipHostMalloc((void**)&c->pin, (size_t)nranks * SLOT_BYTES, hipHostMallocDefault));
  c->hdr->attached.fetch_add(1);
  barrier(c);                                  // (ncclCommInitRank is collective too)
  if (rank == 0) shm_unlink(c->name.c_str());  // everybody has it mapped: the name can go
  *out = reinterpret_cast<ncclComm_t>(c);
  return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm) {
  Comm* c = reinterpret_cast<Comm*>(comm);
  if (!c) return ncclSuccess;
  if (c->pin) (void)hipHostFree(c->pin);
  if (c->hdr) munmap(c->hdr, c->map_bytes);
  delete c;
  return ncclSuccess;
}

const char* ncclGetErrorString(ncclResult_t r) {
  switch (r) {
    case ncclSuccess: return "no error";
    case ncclUnhandledCudaError: return "HIP call failed (mock_rccl)";
    case ncclSystemError: return "system error (mock_rccl)";
    case ncclInvalidArgument: return "invalid argument (mock_rccl)";
    default: return "error (mock_rccl)";
  }
}

// one piece of at most SLOT_BYTES per rank: copy down, meet the peers, copy up -- all in stream order
static ncclResult_t piece(Comm* c, int kind, const double* send, double* recv, size_t count, size_t recv_stride,
                          int root, hipStream_t st) {
  if (kind != 3 || c->rank == root)
    MOCK_HIP(hipMemcpyAsync(c->pin, send, count * sizeof(double), hipMemcpyDeviceToHost, st));
  Op* op = new Op{c, kind, count, root};
  MOCK_HIP(hipLaunchHostFunc(st, exchange, op));
  if (kind == 2) {
    for (int r = 0; r < c->nranks; ++r)
      MOCK_HIP(hipMemcpyAsync(recv + (size_t)r * recv_stride, c->pin + (size_t)r * count, count * sizeof(double),
                              hipMemcpyHostToDevice, st));
  } else {
    MOCK_HIP(hipMemcpyAsync(recv, c->pin, count * sizeof(double), hipMemcpyHostToDevice, st));
  }
  return ncclSuccess;
}

static ncclResult_t run(Comm* c, int kind, const void* send, void* recv, size_t count, ncclDataType_t dt, int root,
                        hipStream_t st) {
  if (!c || dt != ncclFloat64 || !send || !recv) return ncclInvalidArgument;
  const size_t per = SLOT_BYTES / sizeof(double);
  for (size_t off = 0; off < count; off += per) {
    const size_t n = count - off < per ? count - off : per;
    // (the staging buffer is reused by the next piece: stream order keeps the pieces apart)
    ncclResult_t r = piece(c, kind, (const double*)send + off, (double*)recv + off, n, count, root, st);
    if (r != ncclSuccess) return r;
  }
  return ncclSuccess;
}

ncclResult_t ncclAllReduce(const void* send, void* recv, size_t count, ncclDataType_t dt, ncclRedOp_t op, ncclComm_t comm,
                           hipStream_t st) {
  if (op != ncclSum && op != ncclMin) return ncclInvalidArgument;
  return run(reinterpret_cast<Comm*>(comm), op == ncclSum ? 0 : 1, send, recv, count, dt, 0, st);
}

ncclResult_t ncclAllGather(const void* send, void* recv, size_t count, ncclDataType_t dt, ncclComm_t comm, hipStream_t st) {
  return run(reinterpret_cast<Comm*>(comm), 2, send, recv, count, dt, 0, st);
}

ncclResult_t ncclBroadcast(const void* send, void* recv, size_t count, ncclDataType_t dt, int root, ncclComm_t comm,
                           hipStream_t st) {
  Comm* c = reinterpret_cast<Comm*>(comm);
  if (!c || root < 0 || root >= c->nranks) return ncclInvalidArgument;
  return run(c, 3, send, recv, count, dt, root, st);
}

}  // extern "C"
