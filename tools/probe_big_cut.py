import os, sys, time, numpy as np
sys.path.insert(0, '.')
import __graft_entry__ as ge
pkg = ge.load_pkg()
G, S, seed = 20000, 1000, 0x5EED0003
t0 = time.time()
X = pkg.synth.t1_counts(G, S, seed)
print("data", time.time() - t0, flush=True)
gid, _ = pkg.encode_groups(np.asarray(pkg.synth.groups(S))); ref0 = pkg.synth.ref_mask(G, 3000, seed)
padj = float(sys.argv[1]); n_iter = int(sys.argv[2])
with pkg.Context(device=0, seed=seed) as ctx:
    ctx.set_matrix(X); ctx.set_groups(gid, 2); ctx.compute_thresholds(0.01); ctx.build_pairs(0)
    print("built", time.time() - t0, flush=True)
    res, it, tr = ctx.identify_degs(ref0, 1.0, padj, n_iter, 0)
    print("done", time.time() - t0, it, tr[-1], flush=True)
