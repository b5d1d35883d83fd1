"""Multi-GPU glue: one process per GPU, `torch.distributed` (backend "nccl" == RCCL over xGMI).

The HSS tree is partitioned by subtree: with 2^c ranks, rank g owns the g-th subtree at depth c -- its
rows of the sketch (98 % of the flops), its nodes' compression, ULV factors and solve / apply sweeps, all
without communication -- and the 2^c - 1 nodes above the cut are processed redundantly by every rank
after small all-gathers of the cut nodes' reduced blocks (r x d samples, r x r factor blocks, r x nrhs
vectors; a few hundred KB), plus one all-gather of the solution rows.  The native engine calls back
into `allgather` (SPXAllGatherFn: in-place all-gather of a device buffer); PyTorch is only plumbing
here (a zero-copy tensor view of the engine's device buffer + the collective).
"""
import ctypes as C
import os

import numpy as np

from . import capi


class _CudaView:
    """zero-copy __cuda_array_interface__ view of `count` doubles at a raw device pointer"""

    def __init__(self, ptr, count):
        self.__cuda_array_interface__ = {"shape": (count,), "typestr": "<f8", "data": (int(ptr), False),
                                         "version": 2, "strides": None}


class _CudaView32:
    def __init__(self, ptr, count):
        self.__cuda_array_interface__ = {"shape": (count,), "typestr": "<i4", "data": (int(ptr), False),
                                         "version": 2, "strides": None}


def _tensor32(ptr, count, on_device):
    import torch
    if on_device:
        return torch.as_tensor(_CudaView32(ptr, count), device=torch.device("cuda", torch.cuda.current_device()))
    buf = (C.c_int * count).from_address(int(ptr))
    return torch.from_numpy(np.ctypeslib.as_array(buf))


def _tensor(ptr, count, on_device):
    import torch
    if on_device:
        return torch.as_tensor(_CudaView(ptr, count), device=torch.device("cuda", torch.cuda.current_device()))
    buf = (C.c_double * count).from_address(int(ptr))
    return torch.from_numpy(np.ctypeslib.as_array(buf))


def make_exchange(lib, world, rank):
    """Returns the SPXAllGatherFn callback object (keep it alive as long as the matrix lives)."""
    import torch
    import torch.distributed as dist
    lib.hssk_is_device_pointer.argtypes = [C.c_void_p]

    debug = bool(os.environ.get("STRUMPACK_AMD_DEBUG_COMM"))
    counter = [0]

    def allgather(user, dbuf, bytes_per_rank):
        try:
            _allgather(dbuf, bytes_per_rank)
        except BaseException:  # an exception cannot propagate through the C caller: fail loudly
            import traceback
            traceback.print_exc()
            os._exit(3)

    def _allgather(dbuf, bytes_per_rank):
        counter[0] += 1
        if debug:
            print("[comm] rank %d call %d bytes_per_rank %d" % (rank, counter[0], bytes_per_rank), flush=True)
        n = int(bytes_per_rank) // 8
        rem = int(bytes_per_rank) % 8
        assert rem == 0 or int(bytes_per_rank) % 4 == 0
        on_dev = bool(lib.hssk_is_device_pointer(dbuf))
        if rem:  # int payloads of odd length: view as 4-byte words
            full = _tensor32(dbuf, int(bytes_per_rank) // 4 * world, on_dev)
            n = int(bytes_per_rank) // 4
        else:
            full = _tensor(dbuf, n * world, on_dev)
        mine = full[rank * n:(rank + 1) * n]
        if on_dev and dist.get_backend() == "nccl":
            # RCCL all-gather straight into the engine's buffer; the send block is a private copy (a few KB) so that
            # input and output never alias, whatever the backend's in-place rules are
            dist.all_gather_into_tensor(full, mine.clone())
            torch.cuda.synchronize()
        else:  # gloo (CPU tests, or several ranks sharing one GPU): list form through host memory
            src = mine.cpu() if on_dev else mine.clone()
            parts = [torch.empty_like(src) for _ in range(world)]
            dist.all_gather(parts, src)
            for r, prt in enumerate(parts):
                full[r * n:(r + 1) * n] = prt.to(full.device) if on_dev else prt
            if on_dev:
                torch.cuda.synchronize()

    return capi.ALLGATHER_CB(allgather)


class NativeComm:
    """The library's own RCCL communicator (SPX_comm_*, csrc/host/Comm.cpp): the engine issues its collectives on its own
    HIP stream -- no Python callback, no torch synchronisation.  torch.distributed is used once, to hand rank 0's unique
    id to the other ranks."""

    def __init__(self, lib):
        import torch.distributed as dist
        self.lib = lib
        self.world, self.rank = dist.get_world_size(), dist.get_rank()
        box = [None]
        if self.rank == 0:
            buf = C.create_string_buffer(128)
            if lib.SPX_comm_unique_id(buf):
                raise RuntimeError("SPX_comm_unique_id failed")
            box[0] = buf.raw
        dist.broadcast_object_list(box, src=0)
        self.h = C.c_void_p()
        if lib.SPX_comm_create(C.byref(self.h), self.world, self.rank, box[0]):
            raise RuntimeError("SPX_comm_create failed")

    def close(self):
        if self.h:
            self.lib.SPX_comm_destroy(C.byref(self.h))


def shard_range(lib, n, opts, world, rank):
    """rows == columns [lo, hi) of the n x n operand that `rank` has to hold"""
    lo, hi = C.c_int(), C.c_int()
    if lib.SPX_struct_shard_range(n, C.byref(opts), world, rank, C.byref(lo), C.byref(hi)):
        raise RuntimeError("SPX_struct_shard_range failed (world must be a power of two and the tree deep enough)")
    return lo.value, hi.value


def from_blocks_device(lib, d_rows, ldr, d_cols, ldc, n, opts, hss, comm=None, exchange_cb=None, world=None, rank=None):
    """HSS construction from this rank's shard of the operand: row block A[lo:hi, :] (device pointer or None for a
    column-sharded operator) and column block A[:, lo:hi]; `comm` = NativeComm, or an all-gather callback (gloo tests)."""
    h = C.c_void_p()
    if comm is not None:
        rc = lib.SPX_d_struct_from_blocks_device(C.byref(h), n, n, d_rows, ldr, d_cols, ldc, C.byref(opts), C.byref(hss), comm.h)
    else:
        import torch.distributed as dist
        world = dist.get_world_size() if world is None else world
        rank = dist.get_rank() if rank is None else rank
        rc = lib.SPX_d_struct_from_blocks_device_cb(C.byref(h), n, n, d_rows, ldr, d_cols, ldc, C.byref(opts), C.byref(hss),
                                                    world, rank, exchange_cb, None)
    if rc:
        raise RuntimeError("SPX_d_struct_from_blocks_device failed")
    return capi.StructuredMatrix(lib, h, n)


def from_generator(lib, n, kind, opts, hss, comm=None, exchange_cb=None, world=None, rank=None):
    """HSS construction of a matrix given by one of the library's formulas (kind 1: Toeplitz, 2: its upper triangle): no rank
    holds any part of the operand.  `comm` = NativeComm, or an all-gather callback (gloo tests), or neither (one process)."""
    h = C.c_void_p()
    if comm is not None:
        rc = lib.SPX_d_struct_from_generator_comm(C.byref(h), n, kind, C.byref(opts), C.byref(hss), comm.h)
    elif exchange_cb is not None:
        import torch.distributed as dist
        world = dist.get_world_size() if world is None else world
        rank = dist.get_rank() if rank is None else rank
        rc = lib.SPX_d_struct_from_generator_sharded(C.byref(h), n, kind, C.byref(opts), C.byref(hss), world, rank, exchange_cb, None)
    else:
        rc = lib.SPX_d_struct_from_generator(C.byref(h), n, kind, C.byref(opts), C.byref(hss))
    if rc:
        raise RuntimeError("SPX_d_struct_from_generator failed")
    return capi.StructuredMatrix(lib, h, n)


def from_dense_device_comm(lib, dptr, n, lda, opts, hss, comm):
    """replicated operand, native communicator"""
    h = C.c_void_p()
    if lib.SPX_d_struct_from_dense_device_comm(C.byref(h), n, n, dptr, lda, C.byref(opts), C.byref(hss), comm.h):
        raise RuntimeError("SPX_d_struct_from_dense_device_comm failed")
    return capi.StructuredMatrix(lib, h, n)


def from_dense_device(lib, dptr, n, lda, opts, hss, exchange_cb=None, world=None, rank=None):
    """HSS construction from a device-resident dense matrix; sharded over the process group when an
    exchange callback is given."""
    if exchange_cb is None:
        return capi.StructuredMatrix.from_dense_device(lib, dptr, n, lda, opts, hss)
    import torch.distributed as dist
    world = dist.get_world_size() if world is None else world
    rank = dist.get_rank() if rank is None else rank
    h = C.c_void_p()
    rc = lib.SPX_d_struct_from_dense_device_sharded(C.byref(h), n, n, dptr, lda, C.byref(opts), C.byref(hss),
                                                    world, rank, exchange_cb, None)
    if rc:
        raise RuntimeError("SPX_d_struct_from_dense_device_sharded failed")
    return capi.StructuredMatrix(lib, h, n)


CLUSTERING = {"natural": 0, "2means": 1, "kdtree": 2, "pca": 3, "cobble": 4}
KERNEL_TYPES = {"Gauss": 0, "Laplace": 1, "ANOVA": 2}


def from_kernel(lib, X, opts, kernel="Gauss", h=1.0, lam=0.0, degree=1, clustering="2means", neighbors=64,
                exchange_cb=None, world=None, rank=None, comm=None):
    """HSS approximation of the kernel matrix over the rows of X (n x d, host); sharded over the process group when an
    exchange callback or a NativeComm is given (every rank passes the same X).  Returns (matrix, points in cluster order,
    1-based perm)."""
    Xp = np.ascontiguousarray(X, dtype=np.float64).copy()
    n, d = Xp.shape
    perm = np.zeros(n, dtype=np.int32)
    hnd = C.c_void_p()
    if comm is not None:
        rc = lib.SPX_d_struct_from_kernel_comm(C.byref(hnd), n, d, Xp.ctypes.data, KERNEL_TYPES[kernel], h, lam, degree,
                                               C.byref(opts), CLUSTERING[clustering], neighbors, perm.ctypes.data, comm.h)
    elif exchange_cb is None:
        rc = lib.SPX_d_struct_from_kernel(C.byref(hnd), n, d, Xp.ctypes.data, KERNEL_TYPES[kernel], h, lam, degree,
                                          C.byref(opts), CLUSTERING[clustering], neighbors, perm.ctypes.data)
    else:
        import torch.distributed as dist
        world = dist.get_world_size() if world is None else world
        rank = dist.get_rank() if rank is None else rank
        rc = lib.SPX_d_struct_from_kernel_sharded(C.byref(hnd), n, d, Xp.ctypes.data, KERNEL_TYPES[kernel], h, lam, degree,
                                                  C.byref(opts), CLUSTERING[clustering], neighbors, perm.ctypes.data,
                                                  world, rank, exchange_cb, None)
    if rc:
        raise RuntimeError("SPX_d_struct_from_kernel failed")
    return capi.StructuredMatrix(lib, hnd, n), Xp, perm
