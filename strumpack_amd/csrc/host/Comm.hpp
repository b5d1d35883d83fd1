// Native RCCL communicator of the multi-GPU HSS engine (one process per GPU over xGMI).
//
// The engine's collectives are small in-place all-gathers of the cut nodes' reduced blocks (SURVEY.md section 8(e):
// r x d samples, r x r factor blocks, r x nrhs vectors), one sum-reduction of the replicated top nodes' coupling blocks
// per top level, and -- for a column-sharded input -- the reduction of the off-diagonal sample contributions
// Sr = sum_g A(:, cols_g) R(cols_g, :) to the rank that owns the rows.  They are issued on the ENGINE'S HIP stream, in
// order with its kernels: no host synchronisation, no Python, no torch.  (The reference's MPI code does the same
// exchanges with MPI_Allgather / ScaLAPACK on the host, HSSMatrixMPI.compress.hpp:187,494-509.)
//
// RCCL is bound at run time (dlopen of the librccl already in the process -- PyTorch ships one -- or of the ROCm
// installation), so the library has no link-time dependency on it and single-GPU users never load it.
#pragma once
#include <cstddef>

namespace strumpack {
namespace comm {

constexpr int UNIQUE_ID_BYTES = 128;   // NCCL_UNIQUE_ID_BYTES

class RcclComm {
 public:
  // rank 0 creates the id (ncclGetUniqueId) and hands it to the other ranks by any means (MPI_Bcast, a file, a socket,
  // torch.distributed.broadcast_object_list ...)
  static void unique_id(void* out128);
  // collective over all `world` ranks; binds to the calling thread's current HIP device
  RcclComm(int world, int rank, const void* id128);
  ~RcclComm();
  RcclComm(const RcclComm&) = delete;
  RcclComm& operator=(const RcclComm&) = delete;
  int world() const { return world_; }
  int rank() const { return rank_; }
  // dbuf holds world blocks of bytes_per_rank bytes, block `rank` valid on entry, all blocks valid once the stream
  // reaches this point (ncclAllGather, in place)
  void allgather(void* dbuf, long long bytes_per_rank, void* stream);
  // buf[0:count) <- sum over ranks (ncclAllReduce, in place)
  void allreduce_sum(double* buf, long long count, void* stream);
  // recv[0:counts[rank]) <- sum over ranks of send[offs[rank] : offs[rank] + counts[rank])  for every rank at once
  // (grouped ncclReduce: a reduce-scatter with per-rank counts)
  void reduce_scatter_sum(const double* send, const long long* offs, const long long* counts, double* recv, void* stream);

 private:
  void* comm_ = nullptr;
  int world_, rank_;
};

// engine hooks (EngineOptions::allgather_stream / allreduce_stream / reduce_scatter_stream) bound to an RcclComm*
void rccl_allgather_hook(void* user, void* dbuf, long long bytes_per_rank, void* stream);
void rccl_allreduce_hook(void* user, double* buf, long long count, void* stream);
void rccl_reduce_scatter_hook(void* user, const double* send, const long long* offs, const long long* counts, double* recv,
                              void* stream);

}  // namespace comm
}  // namespace strumpack
