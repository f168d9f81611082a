"""K1 duration and fraction of its issue floor by shape: samples at 20 000 genes, genes at 1 000 samples, both families.
The floor is bench.py's: 1024 SIMDs x 2.4 GHz x 2048 comparisons per (2 * planes + 4) cycles, twice the cycles with ties."""
import sys, numpy as np
sys.path.insert(0, '.')
import __graft_entry__ as ge
pkg = ge.load_pkg()
seed = 0x5EED0003
def planes(G): return 12 if G <= 4095 else 15 if G <= 32767 else 16 if G <= 65535 else 17 if G <= 131071 else 18
def peak(G, ties): return 1024 * 2.4e9 * 2048 / ((2 * planes(G) + 4) * (2 if ties else 1))
def run(G, S, family="t0"):
    X = pkg.synth.t0_ranks(G, S, seed) if family == "t0" else pkg.synth.t1_counts(G, S, seed)
    gid, _ = pkg.encode_groups(np.asarray(pkg.synth.groups(S)))
    with pkg.Context(device=0, seed=seed) as ctx:
        ctx.set_profiling(True)
        ctx.set_matrix(X); ctx.set_groups(gid, 2); ctx.compute_thresholds(0.01)
        ts = []
        for rep in range(5):
            ctx.reset_timings(); ctx.build_pairs(0); ts.append(ctx.timings()["k1_ms"])
        ties = bool(ctx.info()["has_ties"])
    t = float(np.median(ts[1:]))
    cmp_ = G * (G - 1) // 2 * S
    print("%s G=%6d S=%5d planes %2d  k1 %8.3f ms  %.3e cmp/s  frac %.3f" % (family, G, S, planes(G), t, cmp_ / t * 1e3, cmp_ / t * 1e3 / peak(G, ties)), flush=True)
for S in (64, 128, 256, 512, 1000, 2000, 4000, 8000):
    run(20000, S)
for G in (3000, 5000, 10000, 32000, 40000, 60000):
    run(G, 1000)
for G, S in ((5000, 200), (20000, 1000), (20000, 4000), (40000, 1000)):
    run(G, S, "t1")
