"""Light passes against sorting passes at config-3 size for cut-offs that put many more genes inside the BH cut (the
histogram tiles beyond the first two, and the hand-over to the sorting path when more than 8192 ranks are finite)."""
import os, sys, subprocess, json, numpy as np
if len(sys.argv) > 1:
    sys.path.insert(0, '.')
    import __graft_entry__ as ge
    pkg = ge.load_pkg()
    G, S, seed = 20000, 1000, 0x5EED0003
    X = pkg.synth.t1_counts(G, S, seed)
    gid, _ = pkg.encode_groups(np.asarray(pkg.synth.groups(S))); ref0 = pkg.synth.ref_mask(G, 3000, seed)
    out = {}
    with pkg.Context(device=0, seed=seed) as ctx:
        ctx.set_matrix(X); ctx.set_groups(gid, 2); ctx.compute_thresholds(0.01); ctx.build_pairs(0)
        for padj in (0.05, 0.3, 0.6, 0.9):
            res, it, tr = ctx.identify_degs(ref0, 1.0, padj, 24, 0)
            out[str(padj)] = {"trace": tr, "sum": float(np.nansum(res)), "hash": int(np.frombuffer(np.ascontiguousarray(res).tobytes(), dtype=np.uint64).sum() % (1 << 61))}
    print(json.dumps(out))
else:
    outs = {}
    for mode in ("0", "1"):
        env = dict(os.environ, REO_LIGHT=mode, REO_DEBUG_PASSES="1")
        p = subprocess.run([sys.executable, __file__, "child"], env=env, capture_output=True, text=True)
        outs[mode] = json.loads(p.stdout.strip().splitlines()[-1])
        if mode == "1":
            print("\n".join(l for l in p.stderr.splitlines() if "light" in l and " 0 light" not in l)[:1500])
    for k in outs["0"]:
        same = outs["0"][k] == outs["1"][k]
        print("padj_deg", k, "DEGs at the end", outs["0"][k]["trace"][-1][0], "identical:", same)
        assert same
