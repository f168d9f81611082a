"""Transform and K1 stage times at config 3 for both data families (device timers)."""
import sys, numpy as np
sys.path.insert(0, ".")
import __graft_entry__ as ge
pkg = ge.load_pkg()
G, S, seed = 20000, 1000, 0x5EED0003
for fam in ("t1", "t0"):
    X = (pkg.synth.t1_counts if fam == "t1" else pkg.synth.t0_ranks)(G, S, seed)
    gid, _ = pkg.encode_groups(np.asarray(pkg.synth.groups(S)))
    with pkg.Context(device=0, seed=seed) as ctx:
        ctx.set_profiling(True)
        for r in range(4):
            ctx.reset_timings(); ctx.set_matrix(X); ctx.set_groups(gid, 2); ctx.compute_thresholds(0.01); ctx.build_pairs(0)
        print(fam, {k: round(v, 3) for k, v in ctx.timings().items() if k in ("transform_ms", "k1_ms")}, "varying key bits:",
              int(np.ceil(np.log2(float(X.max() - X.min() + 1)))))
