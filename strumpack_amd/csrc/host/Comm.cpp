// RcclComm: RCCL bound through dlopen / dlsym (see Comm.hpp).
#include "Comm.hpp"

#include <dlfcn.h>

#include <cstdlib>
#include <cstring>
#include <mutex>
#include <stdexcept>
#include <string>

namespace strumpack {
namespace comm {

namespace {
// the slice of rccl.h this file needs (rccl/rccl.h:40-43, 187, 220, 260, 448-467, 550, 611, 678, 923, 933)
struct UniqueId { char internal[UNIQUE_ID_BYTES]; };
typedef void* Comm_t;
typedef int Result_t;   // ncclSuccess == 0
enum { kChar = 0, kDouble = 8 };
enum { kSum = 0 };
struct Api {
  Result_t (*GetUniqueId)(UniqueId*) = nullptr;
  Result_t (*CommInitRank)(Comm_t*, int, UniqueId, int) = nullptr;
  Result_t (*CommDestroy)(Comm_t) = nullptr;
  Result_t (*AllGather)(const void*, void*, size_t, int, Comm_t, void*) = nullptr;
  Result_t (*AllReduce)(const void*, void*, size_t, int, int, Comm_t, void*) = nullptr;
  Result_t (*Reduce)(const void*, void*, size_t, int, int, int, Comm_t, void*) = nullptr;
  Result_t (*GroupStart)() = nullptr;
  Result_t (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(Result_t) = nullptr;
};

const Api& api() {
  static Api a;
  static std::once_flag once;
  static std::string err;
  std::call_once(once, [] {
    void* h = nullptr;
    // a librccl that is already mapped (PyTorch's) must be the one used: two RCCL instances in one process do not share
    // their device bookkeeping
    const char* names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so", "/opt/rocm/lib/librccl.so.1"};
    std::string why;   // dlerror() hands out its message once: take it right behind the failing dlopen
    // STRUMPACK_AMD_RCCL_LIB names the library to bind instead (a differently installed RCCL; the CPU tests' stand-in,
    // tests/emu/fake_rccl.cpp): that one or none
    if (const char* forced = std::getenv("STRUMPACK_AMD_RCCL_LIB")) {
      if (!(h = dlopen(forced, RTLD_NOW | RTLD_LOCAL))) {
        const char* e = dlerror();
        err = std::string("cannot load STRUMPACK_AMD_RCCL_LIB = ") + forced + ": " + (e ? e : "?");
        return;
      }
    }
    if (!h)
      for (const char* n : names)
        if ((h = dlopen(n, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL))) break;
    if (!h)
      for (const char* n : names) {
        if ((h = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break;
        const char* e = dlerror();
        if (e && why.empty()) why = e;
      }
    if (!h) { err = "cannot load librccl: " + (why.empty() ? std::string("?") : why); return; }
    auto sym = [&](const char* s) -> void* {
      void* p = dlsym(h, s);
      if (!p && err.empty()) err = std::string("librccl lacks ") + s;
      return p;
    };
    a.GetUniqueId = (decltype(a.GetUniqueId))sym("ncclGetUniqueId");
    a.CommInitRank = (decltype(a.CommInitRank))sym("ncclCommInitRank");
    a.CommDestroy = (decltype(a.CommDestroy))sym("ncclCommDestroy");
    a.AllGather = (decltype(a.AllGather))sym("ncclAllGather");
    a.AllReduce = (decltype(a.AllReduce))sym("ncclAllReduce");
    a.Reduce = (decltype(a.Reduce))sym("ncclReduce");
    a.GroupStart = (decltype(a.GroupStart))sym("ncclGroupStart");
    a.GroupEnd = (decltype(a.GroupEnd))sym("ncclGroupEnd");
    a.GetErrorString = (decltype(a.GetErrorString))sym("ncclGetErrorString");
  });
  if (!err.empty()) throw std::runtime_error("RCCL: " + err);
  return a;
}

void ck(Result_t r, const char* what) {
  if (r != 0) {
    const char* s = api().GetErrorString ? api().GetErrorString(r) : "?";
    throw std::runtime_error(std::string("RCCL ") + what + ": " + s);
  }
}
}  // namespace

void RcclComm::unique_id(void* out128) {
  UniqueId id;
  ck(api().GetUniqueId(&id), "ncclGetUniqueId");
  std::memcpy(out128, id.internal, UNIQUE_ID_BYTES);
}

RcclComm::RcclComm(int world, int rank, const void* id128) : world_(world), rank_(rank) {
  if (world < 1 || rank < 0 || rank >= world) throw std::invalid_argument("RcclComm: bad world / rank");
  UniqueId id;
  std::memcpy(id.internal, id128, UNIQUE_ID_BYTES);
  Comm_t c = nullptr;
  ck(api().CommInitRank(&c, world, id, rank), "ncclCommInitRank");
  comm_ = c;
}

RcclComm::~RcclComm() {
  if (comm_) (void)api().CommDestroy((Comm_t)comm_);
}

void RcclComm::allgather(void* dbuf, long long bytes_per_rank, void* stream) {
  if (bytes_per_rank <= 0) return;
  const char* send = (const char*)dbuf + (size_t)bytes_per_rank * rank_;   // in place: send block inside the receive buffer
  ck(api().AllGather(send, dbuf, (size_t)bytes_per_rank, kChar, (Comm_t)comm_, stream), "ncclAllGather");
}

void RcclComm::allreduce_sum(double* buf, long long count, void* stream) {
  if (count <= 0) return;
  ck(api().AllReduce(buf, buf, (size_t)count, kDouble, kSum, (Comm_t)comm_, stream), "ncclAllReduce");
}

void RcclComm::reduce_scatter_sum(const double* send, const long long* offs, const long long* counts, double* recv,
                                  void* stream) {
  ck(api().GroupStart(), "ncclGroupStart");
  for (int g = 0; g < world_; g++)
    if (counts[g] > 0)
      ck(api().Reduce(send + offs[g], recv, (size_t)counts[g], kDouble, kSum, g, (Comm_t)comm_, stream), "ncclReduce");
  ck(api().GroupEnd(), "ncclGroupEnd");
}

void rccl_allgather_hook(void* user, void* dbuf, long long bytes_per_rank, void* stream) {
  ((RcclComm*)user)->allgather(dbuf, bytes_per_rank, stream);
}
void rccl_allreduce_hook(void* user, double* buf, long long count, void* stream) {
  ((RcclComm*)user)->allreduce_sum(buf, count, stream);
}
void rccl_reduce_scatter_hook(void* user, const double* send, const long long* offs, const long long* counts, double* recv,
                              void* stream) {
  ((RcclComm*)user)->reduce_scatter_sum(send, offs, counts, recv, stream);
}

}  // namespace comm
}  // namespace strumpack
