"""Per-stage costs of an N-GPU run, measured on ONE GPU: shard 0 of N (every shard does the same amount of work; each
would have a GPU to itself) builds its share of the pair tiles, packs it, receives N - 1 packs (device copies of its own
pack stand in for the link, so the unpack kernels chew dense words -- the table is garbage, only times are taken) and
runs the passes.  The link time is NOT measured (one GPU): it is priced at an assumed all-gather bus bandwidth.
usage: python3 tools/scaling_model.py [3|4]   (BASELINE config)"""
import os, sys, time, numpy as np
os.environ["REO_CHECK_HOOK_TABLE"] = "0"   # the stand-in exchange makes a garbage table on purpose (and the passes never see it)
sys.path.insert(0, '.')
import __graft_entry__ as ge
pkg = ge.load_pkg()
import torch
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 3
G, S, seed = (20000, 1000, 0x5EED0003) if cfg == 3 else (30000, 4000, 0x5EED0004)
LINK_GBPS = 300.0   # assumed all-gather bus bandwidth per GPU over its 7 xGMI links (not measured here)
X = pkg.synth.t0_ranks(G, S, seed)
gid, lev = pkg.encode_groups(np.asarray(pkg.synth.groups(S)))
ref0 = pkg.synth.ref_mask(G, 3000, seed)
dev = torch.device("cuda", 0)
Xd = torch.from_numpy(np.ascontiguousarray(X.T)).to(dev)
P = G * (G - 1) // 2
rows = []
plain = {}   # the build with the whole exchange behind the pair kernel (REO_EXCHANGE_WAVES=1, round 3's form), per N
for world, waves in ((1, "4"), (2, "1"), (2, "4"), (4, "1"), (4, "4"), (8, "1"), (8, "4")):
    os.environ["REO_EXCHANGE_WAVES"] = waves   # (read when the context is created)
    sent = [0]
    def gather(send, recv, nbytes, stream):
        st = torch.cuda.ExternalStream(stream, device=dev)
        with torch.cuda.stream(st):
            src = torch.as_tensor(pkg.dist._RawDevBytes(send, nbytes), device=dev)
            dst = torch.as_tensor(pkg.dist._RawDevBytes(recv, nbytes * world), device=dev)
            for r in range(world):
                dst[r * nbytes:(r + 1) * nbytes].copy_(src, non_blocking=True)
        sent[0] += nbytes
    with pkg.Context(device=0, seed=seed) as ctx:
        ctx.set_profiling(True)
        if world > 1:
            ctx.set_shard(0, world); ctx.set_allgather(gather)
        best = None
        for rep in range(4):
            ctx.reset_timings(); sent[0] = 0
            torch.cuda.synchronize(); t0 = time.perf_counter()
            ctx.set_matrix_device(Xd.data_ptr(), G, S, G, "i64"); ctx.set_groups(gid, 2); ctx.compute_thresholds(0.01)
            ctx.build_pairs(0)
            torch.cuda.synchronize()   # (one GPU, no exchange: the call returns with the pair kernel still running)
            t1 = time.perf_counter()
            if world == 1:   # the passes are replicated: identical on every rank for every N -- and they must never see the
                res, iters, trace = ctx.identify_degs(ref0, 1.0, 0.05, 128, 0)   # garbage table of the stand-in exchange
                passes_ms = None
            t2 = time.perf_counter()
            tm = ctx.timings()
            cur = dict(build=(t1 - t0) * 1e3, passes=(t2 - t1) * 1e3 if world == 1 else rows[0][3]["passes"], k1=tm["k1_ms"], transform=tm["transform_ms"], exchange=tm["exchange_ms"], iter=tm["iter_ms"])
            if best is None or cur["build"] + cur["passes"] < best["build"] + best["passes"]:
                best = cur
        info = ctx.info()
    if world > 1 and waves == "1":
        plain[world] = best
        continue
    recv_mb = sent[0] * (world - 1) / 1e6
    link = recv_mb / 1e3 / LINK_GBPS * 1e3 if world > 1 else 0.0
    step = best["build"] + best["passes"] + link
    rows.append((world, info["tiles_owned"], info["tiles_total"], best, recv_mb, link, step))
print("BASELINE config %d: %d genes x %d samples, 128 forced passes; shard 0 of N alone on one MI355X" % (cfg, G, S))
print("pipelined exchange (4 waves: wave w packed, gathered and unpacked on a second stream while wave w + 1 is counted) against the plain form (everything behind the pair kernel)")
print("%2s %9s %9s %19s %12s %10s %10s %19s %21s %17s" % ("N", "K1 ms", "transf.", "exposed exch. ms", "recv MB/rank", "link* ms", "passes ms", "build wall ms", "pred. step ms", "speed-up"))
print("%2s %9s %9s %19s %12s %10s %10s %19s %21s %17s" % ("", "", "", "pipel. (plain)", "", "", "", "pipel. (plain)", "pipel. (plain)", "pipel. (plain)"))
base = rows[0][3]["build"] + rows[0][3]["passes"]
for world, own, tot, b, mb, link, step in rows:
    pl = plain.get(world)
    if not pl:
        print("%2d %9.3f %9.3f %19s %12.1f %10.3f %10.3f %19.3f %21.3f %17.2f" % (world, b["k1"], b["transform"], "-", mb, link, b["passes"], b["build"], base, 1.0))
        continue
    p_pipe = b["build"] + link / 4.0 + b["passes"]     # only the last wave's gather cannot hide behind the pair kernel
    p_plain = pl["build"] + link + pl["passes"]
    print("%2d %9.3f %9.3f %19s %12.1f %10.3f %10.3f %19s %21s %17s" % (world, b["k1"], b["transform"], "%.3f (%.3f)" % (b["exchange"], pl["exchange"]), mb, link, b["passes"],
          "%.3f (%.3f)" % (b["build"], pl["build"]), "%.3f (%.3f)" % (p_pipe, p_plain), "%.2f (%.2f)" % (base / p_pipe, base / p_plain)))
print("* link time = bytes received / %.0f GB/s (assumed all-gather bus bandwidth; no N > 1 hardware run exists)." % LINK_GBPS)
print("pred. step = measured build_pairs wall (transform + K1 share + pack + N-1 dense unpacks; on ONE GPU the exchange kernels of the pipelined form")
print("  compete with the pair kernel for the same CUs, so its build wall is not shorter here) + link* (plain: all of it; pipelined: the last wave's quarter) +")
print("  measured identify_degs wall.  K1 ms of the pipelined form = makespan of its four launches, exchange kernels of earlier waves included.")
