import os, sys, numpy as np
os.environ["REO_DEBUG_STAMPS"] = "1"; os.environ["REO_TRANSFORM"] = "wide"
sys.path.insert(0, ".")
import __graft_entry__ as ge
pkg = ge.load_pkg()
G, S, seed = 20000, 1000, 0x5EED0003
for fam in ("t0", "t1"):
    X = (pkg.synth.t0_ranks if fam == "t0" else pkg.synth.t1_counts)(G, S, seed)
    gid, _ = pkg.encode_groups(np.asarray(pkg.synth.groups(S)))
    with pkg.Context(device=0, seed=seed) as ctx:
        ctx.set_profiling(True)
        for r in range(2):
            ctx.reset_timings(); ctx.set_matrix(X); ctx.set_groups(gid, 2); ctx.compute_thresholds(0.01); ctx.build_pairs(0)
            print(fam, {k: round(v, 3) for k, v in ctx.timings().items() if k in ("transform_ms", "k1_ms")}, ctx.info()["transform_in_lds"], flush=True)
