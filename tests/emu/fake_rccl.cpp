// TEST INFRASTRUCTURE ONLY.  A stand-in for librccl on the CPU tier: the nine nccl* entry points that
// strumpack_amd/csrc/host/Comm.cpp binds (dlsym), implemented over POSIX shared memory and a process-shared barrier, so
// that RcclComm -- the in-place ncclAllGather offsets, the grouped ncclReduce that serves as a reduce-scatter with
// per-rank counts, ncclAllReduce -- runs with N > 1 ranks (one process per rank, "device" memory = host memory of the
// emulator build, stream = none: every call completes before it returns).  Selected with STRUMPACK_AMD_RCCL_LIB; the
// product never loads it on its own.  Semantics follow rccl.h: AllGather receives rank r's block at recvbuff + r * count
// (send may alias that block), Reduce delivers the sum to `root` only, calls between GroupStart and GroupEnd are issued
// at GroupEnd, in order; every rank makes the same sequence of calls.  Sums run over the ranks in rank order.
#include <fcntl.h>
#include <sched.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

namespace {
constexpr size_t kSlot = size_t(4) << 20;   // staging bytes per rank and round
struct Header {
  std::atomic<int> ready, count, gen, attached, failed;
  int world;
};
struct Comm {
  Header* h = nullptr;
  char* slots = nullptr;
  size_t bytes = 0;
  int world = 0, rank = 0;
  std::string name;
};
struct Pending { const void* send; void* recv; size_t count; int dtype, root; Comm* c; };
thread_local bool grouping = false;
thread_local std::vector<Pending> queue;

size_t dsize(int dtype) { return dtype == 8 || dtype == 4 || dtype == 5 ? 8 : (dtype == 0 || dtype == 1 ? 1 : 4); }   // ncclFloat64 = 8, ncclInt8 / ncclChar = 0

bool barrier(Comm* c) {
  Header* h = c->h;
  const int g = h->gen.load();
  if (h->count.fetch_add(1) == c->world - 1) { h->count.store(0); h->gen.fetch_add(1); return true; }
  const auto t0 = std::chrono::steady_clock::now();
  while (h->gen.load() == g) {
    sched_yield();
    if (h->failed.load() || std::chrono::steady_clock::now() - t0 > std::chrono::seconds(120)) { h->failed.store(1); return false; }
  }
  return true;
}
// the sum of every rank's [send, send + count) doubles, delivered to root (root < 0: to every rank)
int reduce(Comm* c, const void* send, void* recv, size_t count, int dtype, int root) {
  if (dsize(dtype) != 8) return 4;   // ncclInvalidArgument: only double sums are needed
  const size_t per = kSlot / 8;
  for (size_t o = 0; o < count || o == 0; o += per) {
    const size_t n = count > o ? std::min(per, count - o) : 0;
    std::memcpy(c->slots + kSlot * c->rank, (const double*)send + o, 8 * n);
    if (!barrier(c)) return 1;
    if (root < 0 || root == c->rank) {
      double* out = (double*)recv + o;
      for (size_t e = 0; e < n; e++) {
        double s = 0.;
        for (int r = 0; r < c->world; r++) s += ((const double*)(c->slots + kSlot * r))[e];
        out[e] = s;
      }
    }
    if (!barrier(c)) return 1;
    if (count == 0) break;
  }
  return 0;
}
}  // namespace

extern "C" {
typedef struct { char internal[128]; } ncclUniqueId;

int ncclGetUniqueId(ncclUniqueId* id) {
  static std::atomic<int> serial{0};
  std::memset(id->internal, 0, sizeof(id->internal));
  std::snprintf(id->internal, sizeof(id->internal), "/spx_fake_rccl_%d_%d_%ld", (int)getpid(), serial.fetch_add(1),
                (long)std::chrono::steady_clock::now().time_since_epoch().count());
  return 0;
}

int ncclCommInitRank(void** out, int world, ncclUniqueId id, int rank) {
  if (world < 1 || rank < 0 || rank >= world || id.internal[0] != '/') return 4;
  Comm* c = new Comm();
  c->world = world; c->rank = rank; c->name = id.internal;
  c->bytes = 4096 + kSlot * (size_t)world;
  int fd = shm_open(c->name.c_str(), O_CREAT | O_EXCL | O_RDWR, 0600);
  const bool creator = fd >= 0;
  if (!creator) fd = shm_open(c->name.c_str(), O_RDWR, 0600);
  if (fd < 0) { delete c; return 2; }
  if (creator && ftruncate(fd, (off_t)c->bytes) != 0) { close(fd); delete c; return 2; }
  if (!creator) {   // (the creator may not have sized the segment yet)
    struct stat st;
    for (int i = 0; i < 20000; i++) { if (fstat(fd, &st) == 0 && (size_t)st.st_size >= c->bytes) break; usleep(100); }
  }
  void* p = mmap(nullptr, c->bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  if (p == MAP_FAILED) { delete c; return 2; }
  c->h = (Header*)p;
  c->slots = (char*)p + 4096;
  if (creator) { c->h->world = world; c->h->count.store(0); c->h->gen.store(0); c->h->attached.store(0); c->h->failed.store(0); c->h->ready.store(1); }
  else for (int i = 0; i < 200000 && !c->h->ready.load(); i++) usleep(100);
  if (!c->h->ready.load() || c->h->world != world) { munmap(p, c->bytes); delete c; return 2; }
  c->h->attached.fetch_add(1);
  if (!barrier(c)) return 1;   // (as in RCCL the call is collective)
  *out = c;
  return 0;
}

int ncclCommDestroy(void* comm) {
  Comm* c = (Comm*)comm;
  if (!c) return 0;
  const bool last = c->h->attached.fetch_sub(1) == 1;
  munmap((void*)c->h, c->bytes);
  if (last) shm_unlink(c->name.c_str());
  delete c;
  return 0;
}

int ncclAllGather(const void* send, void* recv, size_t count, int dtype, void* comm, void*) {
  Comm* c = (Comm*)comm;
  const size_t bytes = count * dsize(dtype);
  for (size_t o = 0; o < bytes || o == 0; o += kSlot) {
    const size_t n = bytes > o ? std::min(kSlot, bytes - o) : 0;
    std::memcpy(c->slots + kSlot * c->rank, (const char*)send + o, n);   // (out of the receive buffer first: send may be the rank's own block of it)
    if (!barrier(c)) return 1;
    for (int r = 0; r < c->world; r++) std::memcpy((char*)recv + bytes * r + o, c->slots + kSlot * r, n);
    if (!barrier(c)) return 1;
    if (bytes == 0) break;
  }
  return 0;
}

int ncclAllReduce(const void* send, void* recv, size_t count, int dtype, int op, void* comm, void*) {
  if (op != 0) return 4;
  return reduce((Comm*)comm, send, recv, count, dtype, -1);
}

int ncclReduce(const void* send, void* recv, size_t count, int dtype, int op, int root, void* comm, void*) {
  if (op != 0) return 4;
  if (grouping) { queue.push_back(Pending{send, recv, count, dtype, root, (Comm*)comm}); return 0; }
  return reduce((Comm*)comm, send, recv, count, dtype, root);
}

int ncclGroupStart() { grouping = true; return 0; }
int ncclGroupEnd() {
  grouping = false;
  int rc = 0;
  for (auto& p : queue)
    if (!rc) rc = reduce(p.c, p.send, p.recv, p.count, p.dtype, p.root);
  queue.clear();
  return rc;
}

const char* ncclGetErrorString(int r) {
  switch (r) {
    case 0: return "no error";
    case 1: return "fake rccl: a rank did not reach the collective (time-out)";
    case 2: return "fake rccl: shared memory segment unavailable";
    case 4: return "fake rccl: invalid argument";
    default: return "fake rccl: error";
  }
}
}
