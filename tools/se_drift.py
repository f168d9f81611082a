"""How much the trimmed standard deviation of delta1 (:409-411) and the BH cut move from one pass to the next
(config-3 shape by default): what a bracket around the previous pass's se has to cover."""
import sys, numpy as np
sys.path.insert(0, ".")
import __graft_entry__ as ge
pkg = ge.load_pkg()
G, S, fam = (int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]) if len(sys.argv) > 3 else (20000, 1000, "t0")
seed = 0x5EED0003
X = {"t0": pkg.synth.t0_ranks, "t1": pkg.synth.t1_counts, "float": pkg.synth.float_expr}[fam](G, S, seed)
gid, _ = pkg.encode_groups(np.asarray(pkg.synth.groups(S))); ref0 = pkg.synth.ref_mask(G, 3000, seed)
a0, b0 = int(np.rint(G * 0.05)) - 1, int(np.rint(G * 0.95)) - 1
prev = None
with pkg.Context(device=0, seed=seed) as ctx:
    ctx.set_matrix(X); ctx.set_groups(gid, 2); ctx.compute_thresholds(0.01); ctx.build_pairs(0)
    for n in list(range(1, 41)) + [64, 65, 127, 128]:
        res, iters, trace = ctx.identify_degs(ref0, 1.0, 0.05, n, 0)
        d1 = np.sort(res[:, 11])
        se = d1[a0:b0 + 1].std(ddof=1)
        print("pass %3d  se %.12g  rel change %+.3e  DEGs %d" % (n, se, (se / prev - 1.0) if prev else 0.0, trace[-1][0]), flush=True)
        prev = se
