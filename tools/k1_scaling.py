"""K1 duration vs samples and genes: fits t = a*S + b per tile (where does the non-loop time go?)."""
import sys, numpy as np
sys.path.insert(0, '.')
import __graft_entry__ as ge
pkg = ge.load_pkg()
seed = 0x5EED0003
PEAK = 3.93e13
def run(G, S, family="t0"):
    X = pkg.synth.t0_ranks(G, S, seed) if family == "t0" else pkg.synth.t1_counts(G, S, seed)
    gid = np.asarray(pkg.synth.groups(S)); gid, _ = pkg.encode_groups(gid)
    with pkg.Context(device=0, seed=seed) as ctx:
        ctx.set_profiling(True)
        ctx.set_matrix(X); ctx.set_groups(gid, 2); ctx.compute_thresholds(0.01)
        best = 1e9
        for rep in range(4):
            ctx.reset_timings(); ctx.build_pairs(0); t = ctx.timings()
            best = min(best, t["k1_ms"] if "k1_ms" in t else list(t.values())[1])
    cmp_ = G * (G - 1) // 2 * S
    print("%s G=%6d S=%5d  k1 %.3f ms  %.3e cmp/s  frac %.3f" % (family, G, S, best, cmp_ / best * 1e3, cmp_ / best * 1e3 / PEAK), flush=True)
    return best
for S in (64, 128, 256, 512, 1000, 2000, 4000):
    run(20000, S)
for G in (5000, 10000, 40000):
    run(G, 1000)
run(20000, 1000, "t1")
