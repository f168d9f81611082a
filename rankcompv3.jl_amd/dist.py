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


class RankWatch:
    """A side channel for FAILURES of a multi-rank run (bench.py --gpus N), beside the collectives.

    A rank that fails between two collectives leaves its peers inside the next one (RCCL and gloo wait for ever), and the launcher
    then ends the job without rank 0 having printed anything: a timeout is all the caller sees.  With a watch every rank owns one
    key of the rendezvous store: `report(msg)` sets it (then the failing rank waits `grace` seconds before it exits, so that its
    peers can act), and a daemon thread on every rank polls the peers' keys; when one appears the thread calls
    `on_peer_failure(rank, msg)` -- rank 0 prints its JSON line with an `error` field there -- and ends the process with `exit_code`.
    The store is torch.distributed's own (the TCPStore of the rendezvous); nothing here touches a GPU or a collective."""

    def __init__(self, store, rank: int, world: int, on_peer_failure=None, poll_s: float = 0.2, grace_s: float = 3.0, exit_code: int = 4):
        import threading
        self.store, self.rank, self.world = store, rank, world
        self.on_peer_failure, self.poll_s, self.grace_s, self.exit_code = on_peer_failure, poll_s, grace_s, exit_code
        self._stop = threading.Event()
        self._t = threading.Thread(target=self._run, name="reo-rank-watch", daemon=True)
        self._t.start()

    @staticmethod
    def key(rank: int) -> str:
        return f"reo_rank_failed_{rank}"

    def _run(self):
        import os
        import sys
        while not self._stop.wait(self.poll_s):
            for r in range(self.world):
                if r == self.rank:
                    continue
                try:
                    if not self.store.check([self.key(r)]):
                        continue
                    msg = self.store.get(self.key(r)).decode("utf-8", "replace")
                except Exception:   # the store went away with rank 0: nothing left to watch
                    return
                if self._stop.is_set():
                    return
                try:
                    if self.on_peer_failure:
                        self.on_peer_failure(r, msg)
                finally:
                    sys.stdout.flush(); sys.stderr.flush()
                    os._exit(self.exit_code)   # the main thread may sit in a collective that will never complete

    def report(self, msg: str) -> None:
        """This rank has failed: tell the peers, give them time to act."""
        import time
        try:
            self.store.set(self.key(self.rank), msg[:2000])
        except Exception:
            return
        time.sleep(self.grace_s)

    def stop(self) -> None:
        self._stop.set()
