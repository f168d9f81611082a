"""What the gather form of the table exchange costs next to the sum form, two shards on ONE GPU (thread barrier,
device-to-device copies stand in for the link): pack + unpack kernels are real, the wire time is not."""
import sys, threading, time, numpy as np
sys.path.insert(0, '.')
import __graft_entry__ as ge
pkg = ge.load_pkg()
import torch
G, S, seed, world = 20000, 1000, 0x5EED0003, int(sys.argv[1]) if len(sys.argv) > 1 else 2
X = pkg.synth.t0_ranks(G, S, seed)
gid, lev = pkg.encode_groups(np.asarray(pkg.synth.groups(S)))
dev = torch.device("cuda", 0)
for form in ("gather", "sum"):
    barrier = threading.Barrier(world); slots = [None] * world; out = [None] * world
    def run(rank):
        def gather(send, recv, nbytes, stream):
            torch.cuda.ExternalStream(stream, device=dev).synchronize()
            slots[rank] = torch.as_tensor(pkg.dist._RawDevBytes(send, nbytes), device=dev)
            mine = torch.as_tensor(pkg.dist._RawDevBytes(recv, nbytes * world), device=dev)
            barrier.wait()
            for r in range(world): mine[r * nbytes:(r + 1) * nbytes].copy_(slots[r])
            torch.cuda.synchronize(); barrier.wait()
            out[rank] = nbytes
        def total(ptr, count, stream):
            torch.cuda.ExternalStream(stream, device=dev).synchronize()
            slots[rank] = torch.as_tensor(pkg.dist._RawDev(ptr, count), device=dev)
            barrier.wait()
            if rank == 0:
                t = slots[0].clone()
                for r in range(1, world): t += slots[r]
                for r in range(world): slots[r].copy_(t)
                torch.cuda.synchronize()
            barrier.wait()
            out[rank] = count * 4
        with pkg.Context(device=0, seed=seed) as ctx:
            ctx.set_profiling(True)
            ctx.set_matrix(X); ctx.set_groups(gid, 2); ctx.compute_thresholds(0.01); ctx.set_shard(rank, world)
            (ctx.set_allgather(gather) if form == "gather" else ctx.set_allreduce(total))
            for rep in range(3):
                ctx.reset_timings(); ctx.build_pairs(0)
            tm = ctx.timings()
            if rank == 0: print("%s, %d shards: exchange stage %.2f ms (host-side stand-in included), K1 %.2f ms, bytes a shard sends %.1f MB" % (form, world, tm["exchange_ms"], tm["k1_ms"], out[0] / 1e6), flush=True)
    th = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    [t.start() for t in th]; [t.join() for t in th]
