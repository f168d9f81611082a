"""Exploratory runs of identify_degs at 20 000 genes with unusual parameters, light passes (two launches, persistent) against
sorting passes only.  Every case under an alarm: a call that does not return kills the process with the case on record."""
import os, sys, signal, numpy as np
sys.path.insert(0, '.')
import __graft_entry__ as ge
pkg = ge.load_pkg()
G, S, seed = 20000, 64, 0x5EED0099
X = pkg.synth.t1_counts(G, S, seed)
gid, _ = pkg.encode_groups(np.asarray(pkg.synth.groups(S)))
refs = {"3000": pkg.synth.ref_mask(G, 3000, seed), "all": np.ones(G, bool), "20": pkg.synth.ref_mask(G, 20, seed), "none": np.zeros(G, bool)}
cases = []
for n_iter in (1, 2, 3, 5, 40):
    cases.append(("3000", 1.0, 0.05, n_iter, 0))
cases += [("3000", 1.0, 0.05, 40, 5), ("3000", 1.0, 0.05, 40, 100000), ("3000", 0.01, 0.05, 30, 0), ("3000", 1.0, 1e-9, 30, 0),
          ("3000", 1.0, 1.0, 30, 0), ("all", 1.0, 0.05, 30, 0), ("20", 1.0, 0.05, 30, 0), ("none", 1.0, 0.05, 30, 0), ("3000", 0.2, 0.9, 30, 0)]
ctxs = {}
for mode in ("0", "1", "2"):
    os.environ["REO_LIGHT"] = mode
    ctx = pkg.Context(device=0, seed=seed)
    ctx.set_matrix(X); ctx.set_groups(gid, 2); ctx.compute_thresholds(0.05); ctx.build_pairs(0)
    ctxs[mode] = ctx
bad = 0
for case in cases:
    rk, pval, padj, n_iter, n_conv = case
    outs = {}
    for mode, ctx in ctxs.items():
        print("case", case, "mode", mode, flush=True)
        signal.alarm(60)
        outs[mode] = ctx.identify_degs(refs[rk], pval, padj, n_iter, n_conv)
        signal.alarm(0)
    r0, i0, t0 = outs["0"]
    ok = np.isfinite(r0).all(axis=1)
    for mode in ("1", "2"):
        r1, i1, t1 = outs[mode]
        same = i1 == i0 and t1 == t0 and np.array_equal(r1[:, 2:11], r0[:, 2:11]) and np.allclose(r1[ok][:, :2], r0[ok][:, :2], rtol=0, atol=1e-6) \
            and np.array_equal(np.isfinite(r1), np.isfinite(r0))
        if not same:
            bad += 1
            print("  MISMATCH mode", mode, "iters", i1, i0, "trace tail", t1[-2:], t0[-2:], flush=True)
    print("  passes", i0, "last", t0[-1] if t0 else None, flush=True)
print("mismatches:", bad)
sys.exit(1 if bad else 0)
