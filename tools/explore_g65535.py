"""65 535 genes (the most the 16-bit positions allow) with light passes in both launch forms against sorting passes only."""
import os, sys, signal, time, numpy as np
sys.path.insert(0, '.')
import __graft_entry__ as ge
pkg = ge.load_pkg()
G, S, seed = 65535, 32, 0x5EED0098
X = pkg.synth.t1_counts(G, S, seed)
gid, _ = pkg.encode_groups(np.asarray(pkg.synth.groups(S)))
ref0 = pkg.synth.ref_mask(G, 9000, seed)
outs = {}
for mode in ("0", "1", "2"):
    os.environ["REO_LIGHT"] = mode
    signal.alarm(120)
    t0 = time.time()
    with pkg.Context(device=0, seed=seed) as ctx:
        ctx.set_matrix(X); ctx.set_groups(gid, 2); ctx.compute_thresholds(0.05); ctx.build_pairs(0)
        outs[mode] = [ctx.identify_degs(ref0, 1.0, padj, 24, 0) for padj in (0.05, 0.8)]
    signal.alarm(0)
    print("mode", mode, "%.1f s" % (time.time() - t0), [o[2][-1] for o in outs[mode]], flush=True)
bad = 0
for mode in ("1", "2"):
    for (r1, i1, t1), (r0, i0, t0_) in zip(outs[mode], outs["0"]):
        ok = np.isfinite(r0).all(axis=1)
        same = i1 == i0 and t1 == t0_ and np.array_equal(r1[:, 2:11], r0[:, 2:11]) and np.allclose(r1[ok][:, :2], r0[ok][:, :2], rtol=0, atol=1e-6)
        bad += not same
        print("mode", mode, "== sorting:", same, flush=True)
sys.exit(1 if bad else 0)
