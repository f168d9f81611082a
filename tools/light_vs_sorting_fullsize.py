"""Light passes against sorting passes at config-3 size (tie-rich family) for cut-offs that put many more genes inside the
BH cut: the histogram tiles beyond the first two / four.  Same process, REO_LIGHT read at context creation."""
import os, sys, numpy as np
sys.path.insert(0, '.')
import __graft_entry__ as ge
pkg = ge.load_pkg()
G, S, seed = 20000, 1000, 0x5EED0003
X = pkg.synth.t1_counts(G, S, seed)
gid, _ = pkg.encode_groups(np.asarray(pkg.synth.groups(S))); ref0 = pkg.synth.ref_mask(G, 3000, seed)
outs = {}
for mode in ("1", "0"):
    os.environ["REO_LIGHT"] = mode
    with pkg.Context(device=0, seed=seed) as ctx:
        ctx.set_matrix(X); ctx.set_groups(gid, 2); ctx.compute_thresholds(0.01); ctx.build_pairs(0)
        outs[mode] = {padj: ctx.identify_degs(ref0, 1.0, padj, 24, 0) for padj in (0.05, 0.3, 0.6, 0.9)}
    print("mode", mode, "done", flush=True)
for padj in outs["0"]:
    (r1, i1, t1), (r0, i0, t0) = outs["1"][padj], outs["0"][padj]
    ok = np.isfinite(r0).all(axis=1)
    same = i1 == i0 and t1 == t0 and np.array_equal(r1[:, 2:11], r0[:, 2:11]) and np.allclose(r1[ok][:, :2], r0[ok][:, :2], rtol=0, atol=1e-6)
    print("padj_deg", padj, "DEGs at the end", t0[-1][0], "light == sorting:", same, flush=True)
    assert same
