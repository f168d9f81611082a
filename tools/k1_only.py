"""Runs only transform + K1 (3 x build_pairs) on the bench workload; a target for rocprofv3 --pmc."""
import sys, numpy as np
sys.path.insert(0, '.')
import __graft_entry__ as ge
pkg = ge.load_pkg()
seed = 0x5EED0003
G, S = 20000, 1000
fam = sys.argv[1] if len(sys.argv) > 1 else "t0"
X = pkg.synth.t0_ranks(G, S, seed) if fam == "t0" else pkg.synth.t1_counts(G, S, seed)
gid, _ = pkg.encode_groups(np.asarray(pkg.synth.groups(S)))
with pkg.Context(device=0, seed=seed) as ctx:
    ctx.set_matrix(X); ctx.set_groups(gid, 2); ctx.compute_thresholds(0.01)
    for rep in range(3):
        ctx.build_pairs(0)
print("done")
