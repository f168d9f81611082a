"""Five groups at 20 000 x 1 000: one k1_group_counts + five k1_classify launches (target for rocprofv3)."""
import sys, numpy as np
sys.path.insert(0, '.')
import __graft_entry__ as ge
pkg = ge.load_pkg()
G, S, seed, C = 20000, 1000, 0x5EED0003, 5
X = pkg.synth.t0_ranks(G, S, seed); gid = (np.arange(S) % C).astype(np.int32)
with pkg.Context(device=0, seed=seed) as ctx:
    ctx.set_matrix(X); ctx.set_groups(gid, C); ctx.compute_thresholds(0.01)
    for k in range(C): ctx.build_pairs(k)
print("done")
