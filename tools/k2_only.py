"""Whole-table tally scans (reo_tally -> k2_scan, the kernel body of k2_tally's scan form) on the bench workload; a target for rocprofv3 --pmc."""
import sys, numpy as np
sys.path.insert(0, '.')
import __graft_entry__ as ge
pkg = ge.load_pkg()
seed, G, S = 0x5EED0003, 20000, 1000
X = pkg.synth.t0_ranks(G, S, seed)
gid, _ = pkg.encode_groups(np.asarray(pkg.synth.groups(S)))
ref0 = pkg.synth.ref_mask(G, 3000, seed)
with pkg.Context(device=0, seed=seed) as ctx:
    ctx.set_matrix(X); ctx.set_groups(gid, 2); ctx.compute_thresholds(0.01); ctx.build_pairs(0)
    for rep in range(5):
        ctx.tally(ref0)
print("done")
