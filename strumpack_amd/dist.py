"""Multi-GPU glue: one process per GPU, `torch.distributed` (backend "nccl" == RCCL over xGMI).

The hot path shards where the flops are: rank g computes the sketch columns of its shard
(S^T[:, cols_g] = R^T A[cols_g, :]^T and R^T A[:, cols_g], 98 % of the work, no communication), the
shards are exchanged with ONE in-place all-gather per sample array and round (d x N doubles = 154 MB
at N = 1e5, d = 192), and the O(N r^2) tree phase runs replicated on every rank.  The native engine
calls back into `exchange` (SPXExchangeFn) at that point; PyTorch is only plumbing here (zero-copy
tensor views of the engine's device arrays + the collective).
"""
import ctypes as C

import numpy as np

from . import capi


class _CudaView:
    """zero-copy __cuda_array_interface__ view of `count` doubles at a raw device pointer"""

    def __init__(self, ptr, count):
        self.__cuda_array_interface__ = {"shape": (count,), "typestr": "<f8", "data": (int(ptr), False),
                                         "version": 2, "strides": None}


def _tensor(ptr, count, on_device):
    import torch
    if on_device:
        return torch.as_tensor(_CudaView(ptr, count), device=torch.device("cuda", torch.cuda.current_device()))
    buf = (C.c_double * count).from_address(int(ptr))
    return torch.from_numpy(np.ctypeslib.as_array(buf))


def make_exchange(lib, world, rank):
    """Returns the SPXExchangeFn callback object (keep it alive while the matrix is constructed)."""
    import torch
    import torch.distributed as dist
    lib.hssk_is_device_pointer.argtypes = [C.c_void_p]

    def exchange(user, dSrt, dSct, ld, cols_per_rank):
        shard = int(ld) * int(cols_per_rank)
        on_dev = bool(lib.hssk_is_device_pointer(dSrt))
        for ptr in (dSrt, dSct):
            full = _tensor(ptr, shard * world, on_dev)
            mine = full[rank * shard:(rank + 1) * shard]
            if on_dev:
                dist.all_gather_into_tensor(full, mine)
            else:  # gloo (CPU tests): list form, copy back
                parts = [torch.empty_like(mine) for _ in range(world)]
                dist.all_gather(parts, mine.clone())
                for r, prt in enumerate(parts):
                    full[r * shard:(r + 1) * shard] = prt
        if on_dev:
            torch.cuda.synchronize()

    return capi.EXCHANGE_CB(exchange)


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
