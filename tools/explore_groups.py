"""Three groups (one-vs-rest comparisons on one context, shared per-group counts) at 20 000 genes: light passes against
sorting passes only, every comparison, calls repeated on the same context."""
import os, sys, signal, numpy as np
sys.path.insert(0, '.')
import __graft_entry__ as ge
pkg = ge.load_pkg()
G, S, seed, C = 20000, 96, 0x5EED0097, 3
X = pkg.synth.t1_counts(G, S, seed)
gid = (np.arange(S) * C // S).astype(np.int32)  # contiguous thirds: the synthetic effect sits in the second half of the samples
ref0 = pkg.synth.ref_mask(G, 3000, seed)
outs = {}
for mode in ("0", "1"):
    os.environ["REO_LIGHT"] = mode
    signal.alarm(120)
    with pkg.Context(device=0, seed=seed) as ctx:
        ctx.set_matrix(X); ctx.set_groups(gid, C); ctx.compute_thresholds(0.05)
        res = []
        for rep in range(2):
            for k in range(C):
                ctx.build_pairs(k)
                res.append(ctx.identify_degs(ref0, 1.0, 0.05, 20 + 7 * k, 0))
        outs[mode] = res
    signal.alarm(0)
bad = 0
for (r1, i1, t1), (r0, i0, t0) in zip(outs["1"], outs["0"]):
    ok = np.isfinite(r0).all(axis=1)
    same = i1 == i0 and t1 == t0 and np.array_equal(r1[:, 2:11], r0[:, 2:11]) and np.allclose(r1[ok][:, :2], r0[ok][:, :2], rtol=0, atol=1e-6)
    bad += not same
    print("passes", i0, "last", t0[-1], "light == sorting:", same, flush=True)
sys.exit(1 if bad else 0)
