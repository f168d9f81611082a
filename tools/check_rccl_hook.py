"""Exercise dist.allreduce_hook with the real RCCL backend (world size 1 on the one-GPU box): a library-owned
stream and a hipMalloc'ed buffer, the collective enqueued through torch.cuda.ExternalStream."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, ".")
import __graft_entry__ as ge

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
pkg = ge.load_pkg()
hook = pkg.dist.allreduce_hook(dev)
side = torch.cuda.Stream(device=dev)            # stands in for the library's own stream
buf = torch.arange(160000, dtype=torch.int32, device=dev)
with torch.cuda.stream(side):
    buf.add_(1)
hook(buf.data_ptr(), buf.numel(), side.cuda_stream)   # (ptr, count, stream) exactly as the library calls it
with torch.cuda.stream(side):
    buf.mul_(2)
side.synchronize()
assert torch.equal(buf.cpu(), (torch.arange(160000, dtype=torch.int32) + 1) * 2)
# and through the library: a sharded context of world 1 never calls the hook, so call reo_tally on a forced 2-shard
# context whose hook all-reduces over the (single-rank) RCCL group: the result is the partial tally of shard 0
G, S, seed = 1500, 24, 5
X = pkg.synth.t1_counts(G, S, seed)
gid, lev = pkg.encode_groups(pkg.synth.groups(S))
with pkg.Context(device=0, seed=seed) as ctx:
    ctx.set_matrix(X); ctx.set_groups(gid, 2); ctx.compute_thresholds(0.01)
    ctx.set_shard(0, 2); ctx.set_allreduce(hook)
    ctx.build_pairs(0)
    ref = pkg.synth.ref_mask(G, 200, seed)
    cont = ctx.tally(ref)
    res, iters, trace = ctx.identify_degs(ref, 1.0, 0.05, 6, 1)
    print("rccl hook ok: tally rows", cont.shape, "passes", iters, "trace", trace[-1])
# cost of the hook per pass at bench size: shard 0 of a forced 2-shard context, 128 forced passes, against the
# same shard with a no-op hook (the single-rank RCCL all-reduce moves no data: what is measured is the enqueue path)
import time
G, S, seed = 20000, 1000, 0x5EED0003
X = pkg.synth.t0_ranks(G, S, seed); gid, lev = pkg.encode_groups(pkg.synth.groups(S)); ref = pkg.synth.ref_mask(G, 3000, seed)
for name, h in (("no-op hook", lambda ptr, count, stream=0: None), ("RCCL hook  ", hook)):
    with pkg.Context(device=0, seed=seed) as ctx:
        ctx.set_matrix(X); ctx.set_groups(gid, 2); ctx.compute_thresholds(0.01)
        ctx.set_shard(0, 2); ctx.set_allreduce(h)
        ctx.build_pairs(0)
        best = 1e9
        for rep in range(4):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            res, iters, trace = ctx.identify_degs(ref, 1.0, 0.05, 128, 0)
            torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
        print("%s: 128 passes %.2f ms = %.1f us per pass" % (name, best * 1e3, best * 1e6 / 128))
dist.destroy_process_group()
