"""torch.distributed plumbing for G-sharded runs: one process per GPU, RCCL
(backend "nccl") on MI355X, gloo in CPU tests.  The library itself is
collective-agnostic: it hands the hook a raw pointer to int32 counters and the
hook sums them in place across shards."""
from __future__ import annotations

import ctypes

import numpy as np


class _RawDev:
    """Zero-copy view of a raw device pointer for torch (cuda array interface)."""

    def __init__(self, ptr: int, n: int):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": "<i4", "data": (ptr, False), "version": 2}


class _RawDevBytes:
    """The same for a run of bytes."""

    def __init__(self, ptr: int, n: int):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": "|u1", "data": (ptr, False), "version": 2}


def allreduce_hook(device=None, group=None, via_host: bool = False):
    """fn(ptr, count, stream): sum int32[count] at `ptr` across the process group, in place, ordered
    on the library's HIP stream.  `device` is a torch cuda device for device pointers: the collective
    (RCCL through torch.distributed) is enqueued under torch.cuda.stream(ExternalStream(stream)), so
    nothing blocks on the host.  device=None: `ptr` is a host pointer (gloo tests).  via_host=True
    stages a device buffer through the host so that a gloo group can reduce it (debugging the sharded
    flow with several ranks on one GPU, where RCCL refuses to run)."""
    import torch
    import torch.distributed as dist

    cache: dict = {}  # (ptr, count, stream) -> (stream object, tensor view): the library passes the same triple every pass

    def fn(ptr: int, count: int, stream: int = 0) -> None:
        if device is None:
            buf = (ctypes.c_int32 * count).from_address(ptr)
            t = torch.from_numpy(np.ctypeslib.as_array(buf))
            dist.all_reduce(t, group=group)
            return
        key = (ptr, count, stream)
        hit = cache.get(key)
        if hit is None:  # wrapping the raw pointer costs tens of microseconds; a step calls the hook 128 times
            ext = torch.cuda.ExternalStream(stream, device=device) if stream else torch.cuda.current_stream(device)
            with torch.cuda.stream(ext):
                t = torch.as_tensor(_RawDev(ptr, count), device=device)
            if len(cache) > 16:
                cache.clear()
            hit = cache[key] = (ext, t)
        ext, t = hit
        with torch.cuda.stream(ext):
            if via_host:
                ext.synchronize()
                h = t.cpu()
                dist.all_reduce(h, group=group)
                t.copy_(h)
                ext.synchronize()
            else:
                dist.all_reduce(t, group=group)  # enqueued; `ext` waits for it, the host does not

    return fn


def allgather_hook(device, group=None, via_host: bool = False):
    """fn(send_ptr, recv_ptr, bytes_per_rank, stream): all-gather of the shards' packed table words through
    torch.distributed (RCCL), enqueued on the library's HIP stream.  via_host=True stages the buffers through the host so
    that a gloo group can carry them (several ranks on one GPU, where RCCL refuses to run)."""
    import torch
    import torch.distributed as dist

    def fn(send: int, recv: int, nbytes: int, stream: int = 0) -> None:
        world = dist.get_world_size(group)
        ext = torch.cuda.ExternalStream(stream, device=device) if stream else torch.cuda.current_stream(device)
        with torch.cuda.stream(ext):
            src = torch.as_tensor(_RawDevBytes(send, nbytes), device=device)
            dst = torch.as_tensor(_RawDevBytes(recv, nbytes * world), device=device)
            if via_host:
                ext.synchronize()
                parts = [torch.empty(nbytes, dtype=torch.uint8) for _ in range(world)]
                dist.all_gather(parts, src.cpu(), group=group)
                dst.copy_(torch.cat(parts))
                ext.synchronize()
            else:
                dist.all_gather_into_tensor(dst, src, group=group)

    return fn


def shard_of_process():
    """(rank, world) of this process in the default group, (0, 1) without torch.distributed."""
    try:
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            return dist.get_rank(), dist.get_world_size()
    except ImportError:
        pass
    return 0, 1


def run_identify_degs_sharded(data, group, gene_names, pval_reo, pval_deg, padj_deg, ref_gene, n_iter, n_conv, *,
                              device_index: int, seed: int = 0, profile: bool = False):
    """identify_degs with the pair tiles split over the ranks of the default process group
    (every rank passes the same arguments and gets the same result)."""
    import torch

    from .hotpath import run_identify_degs
    rank, world = shard_of_process()
    hook = allgather_hook(torch.device("cuda", device_index)) if world > 1 else None  # the gather form: a quarter of the sum's bytes
    return run_identify_degs(data, group, gene_names, pval_reo, pval_deg, padj_deg, ref_gene, n_iter, n_conv,
                             seed=seed, device=device_index, shard=(rank, world), allgather=hook, profile=profile)


def table_trace_digest(ctx, G: int, ref0, n_iter: int = 8, rows_per_call: int = 1024) -> str:
    """64-bit digest (hex) of everything a sharded build must reproduce: the class of EVERY ordered gene pair (reo_get_codes over the
    whole G x G table, row blocks), then the iteration's trace and the nine tallies of every gene after `n_iter` forced passes.
    bench.py compares it across ranks and with an unsharded build before it times anything with more than one rank."""
    import hashlib
    import numpy as np
    h = hashlib.blake2b(digest_size=8)
    for i0 in range(0, G, rows_per_call):
        h.update(np.ascontiguousarray(ctx.get_codes(i0, min(G, i0 + rows_per_call), 0, G)).tobytes())
    res, iters, trace = ctx.identify_degs(ref0, 1.0, 0.05, n_iter, 0)
    h.update(np.asarray([iters], dtype=np.int64).tobytes())
    h.update(np.asarray(trace, dtype=np.int64).tobytes())
    h.update(np.ascontiguousarray(res[:, 2:11]).tobytes())
    return h.hexdigest()
