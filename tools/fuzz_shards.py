"""Randomized check of the sharded table build: random gene / sample counts (so that every work-unit geometry -- panels of 1,
2 and 4 column chunks, tie-free and tie-rich widths -- comes up), random shard counts, the gather exchange through a
thread-barrier hook on one GPU; every shard's complete table must equal the unsharded one, code for code.
python tools/fuzz_shards.py [N] [seed]"""
import os, sys, threading, numpy as np
os.environ.setdefault("REO_DEBUG_SEGV", "1")   # a native backtrace if the host side ever crashes again (api.hip)
sys.path.insert(0, '.')
import __graft_entry__ as ge
pkg = ge.load_pkg()
import torch
N = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 99)
dev = torch.device("cuda", 0)
geoms = {}
for case in range(N):
    G = int(rng.integers(1100, 5200)); S = int(rng.choice([8, 24, 64, 200, 520, 1100])); world = int(rng.choice([2, 3, 4, 5, 7]))
    fam = str(rng.choice(["t0", "t1"])); seed = int(rng.integers(1, 1 << 30))
    X = (pkg.synth.t0_ranks if fam == "t0" else pkg.synth.t1_counts)(G, S, seed)
    gid, lev = pkg.encode_groups(np.asarray(pkg.synth.groups(S)))
    with pkg.Context(device=0, seed=seed) as ctx:
        ctx.set_matrix(X); ctx.set_groups(gid, 2); ctx.compute_thresholds(0.05); ctx.build_pairs(0)
        want = ctx.get_codes(0, G, 0, G); info0 = ctx.info()
    barrier = threading.Barrier(world); slots = [None] * world; got = [None] * world; errors = []
    def run(rank):
        try:
            def gather(send, recv, nbytes, stream):
                torch.cuda.ExternalStream(stream, device=dev).synchronize()
                slots[rank] = torch.as_tensor(pkg.dist._RawDevBytes(send, nbytes), device=dev)
                mine = torch.as_tensor(pkg.dist._RawDevBytes(recv, nbytes * world), device=dev)
                barrier.wait()
                for r in range(world): mine[r * nbytes:(r + 1) * nbytes].copy_(slots[r])
                torch.cuda.synchronize(); barrier.wait()
            with pkg.Context(device=0, seed=seed) as c:
                c.set_matrix(X); c.set_groups(gid, 2); c.compute_thresholds(0.05); c.set_shard(rank, world); c.set_allgather(gather)
                c.build_pairs(0)
                got[rank] = (c.get_codes(0, G, 0, G), c.info()["tiles_owned"])
        except Exception:
            import traceback; errors.append(traceback.format_exc()); barrier.abort()
    th = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    [t.start() for t in th]; [t.join(timeout=300) for t in th]
    assert not errors, errors
    assert sum(g[1] for g in got) == info0["tiles_total"]
    for r in range(world):
        assert np.array_equal(got[r][0], want), (case, G, S, world, fam, r)
    key = (fam, info0["chunk_j"], info0["chunks_per_panel"]); geoms[key] = geoms.get(key, 0) + 1
    print("case", case, "ok", G, S, world, fam, "chunk", info0["chunk_j"], "x", info0["chunks_per_panel"], flush=True)
print("fuzz shards ok:", N, "cases; geometries", geoms)
