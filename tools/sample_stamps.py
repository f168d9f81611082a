"""Phase marks of t_sample's histogram form (library built with -DREO_STAMPS; REO_DEBUG_STAMPS=1 prints them) and the transform's
stage time, Int64 ranks: python tools/sample_stamps.py G S"""
import os, sys, numpy as np
os.environ["REO_DEBUG_STAMPS"] = "1"
sys.path.insert(0, ".")
import __graft_entry__ as ge
pkg = ge.load_pkg()
G, S, seed = int(sys.argv[1]), int(sys.argv[2]), 0x5EED0003
X = pkg.synth.t0_ranks(G, S, seed)
gid, _ = pkg.encode_groups(np.asarray(pkg.synth.groups(S)))
with pkg.Context(device=0, seed=seed) as ctx:
    ctx.set_profiling(True)
    for r in range(3):
        ctx.reset_timings(); ctx.set_matrix(X); ctx.set_groups(gid, 2); ctx.compute_thresholds(0.01); ctx.build_pairs(0)
        print({k: round(v, 3) for k, v in ctx.timings().items() if k in ("transform_ms", "k1_ms")}, ctx.info()["transform_in_lds"], flush=True)
