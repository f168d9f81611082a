"""Float64 input (0.1 tie band, radix ranking in the transform) at 20 000 genes: light passes against sorting passes only,
and the wide pair kernels with more than 65 535 samples on a few thousand genes with light passes."""
import os, sys, signal, numpy as np
sys.path.insert(0, '.')
import __graft_entry__ as ge
pkg = ge.load_pkg()
bad = 0
for (G, S, nref, fam) in ((20000, 64, 3000, "float"), (4500, 66000, 900, "t1")):
    seed = 0x5EED0096
    X = pkg.synth.float_expr(G, S, seed) if fam == "float" else pkg.synth.t1_counts(G, S, seed)
    gid, _ = pkg.encode_groups(np.asarray(pkg.synth.groups(S)))
    ref0 = pkg.synth.ref_mask(G, nref, seed)
    outs = {}
    for mode in ("0", "1"):
        os.environ["REO_LIGHT"] = mode
        signal.alarm(200)
        with pkg.Context(device=0, seed=seed) as ctx:
            ctx.set_matrix(X); ctx.set_groups(gid, 2); ctx.compute_thresholds(0.05); ctx.build_pairs(0)
            outs[mode] = ctx.identify_degs(ref0, 1.0, 0.05, 30, 0)
            info = ctx.info()
        signal.alarm(0)
    (r1, i1, t1), (r0, i0, t0) = outs["1"], outs["0"]
    ok = np.isfinite(r0).all(axis=1)
    same = i1 == i0 and t1 == t0 and np.array_equal(r1[:, 2:11], r0[:, 2:11]) and np.allclose(r1[ok][:, :2], r0[ok][:, :2], rtol=0, atol=1e-6)
    bad += not same
    print(fam, G, S, "has_ties", info["has_ties"], "in_lds", info["transform_in_lds"], "passes", i0, "last", t0[-1], "light == sorting:", same, flush=True)
sys.exit(1 if bad else 0)
