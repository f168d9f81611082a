"""Long randomized parity run (not part of the test suite): N random cases of the same generator the GPU
test uses, larger G range, every comparison checked against the oracle.  python tools/fuzz_gpu.py [N] [seed]"""
import os, sys, time, faulthandler, numpy as np
faulthandler.enable()
os.environ.setdefault("REO_DEBUG_SEGV", "1")   # a native backtrace if the host side ever crashes again (api.hip)
sys.path.insert(0, '.')
import __graft_entry__ as ge
pkg = ge.load_pkg(); oracle = ge.load_oracle()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 777)
P_ATOL, STAT_RTOL = 1e-6, 1e-7

def case():
    G = int(rng.choice([rng.integers(12, 200), rng.integers(200, 1400), rng.integers(1400, 2600)]))
    ng = int(rng.choice([2, 2, 3, 5]))
    sizes = rng.integers(2, 40, size=ng)
    S = int(sizes.sum())
    labels = np.concatenate([[f"grp{g}"] * int(n) for g, n in enumerate(sizes)])
    if rng.random() < 0.5:
        labels = labels[rng.permutation(S)]
    kind = str(rng.choice(["small_int", "wide_int", "float_band", "ranks", "float_cont", "big_int", "lu_corner", "infinities"]))
    if kind == "lu_corner":
        # genes whose table is N = [[b b][b b]]: integer-singular, non-singular for :242's float test when b * (1.0 / b) != 1
        # (kernels.hip mccullagh3; synth.lu_corner).  Half of the cases take such a b, the reference set is all genes or a subset.
        b = int(rng.choice([49, 98, 103, 107, 161, 187, 196, 197, 206, 214, 237, 239])) if rng.random() < 0.5 else int(rng.integers(12, 400))
        X, labels, _ = pkg.synth.lu_corner(b, int(rng.integers(0, b + 1)), int(rng.integers(2, 6)), int(rng.integers(1, 4)),
                                           int(rng.integers(0, 6)), int(rng.integers(0, 4)), seed=int(rng.integers(0, 2 ** 30)))
        if rng.random() < 0.5: X = X.astype(np.float64)
        G, S = X.shape
        return dict(G=G, S=S, ng=2, labels=labels, X=X, kind=kind, pval_reo=float(rng.choice([0.01, 0.05, 0.3])),
                    n_conv=int(rng.choice([0, 1, 5])), n_iter=int(rng.integers(1, 9)), seed=int(rng.integers(0, 2 ** 40)),
                    nref=G if rng.random() < 0.5 else int(rng.integers(max(3, G // 2), G + 1)))
    if kind == "small_int": X = rng.integers(0, int(rng.integers(2, 12)), size=(G, S))
    elif kind == "wide_int": X = rng.integers(-50000, 50000, size=(G, S))
    elif kind == "big_int": X = rng.integers(0, 2 ** 31, size=(G, S))
    elif kind == "float_band": X = np.round(rng.normal(5, 1.0, size=(G, S)), 1) + rng.choice([0.0, 0.04, 0.099, 0.1], size=(G, S))
    elif kind == "infinities":   # log(0) = -Inf, a few +Inf, sometimes whole samples or genes: is_greater on equal infinities is false both ways (:72-76)
        X = pkg.synth.with_infinities(np.round(rng.normal(5, 1.0, size=(G, S)), 1) + rng.choice([0.0, 0.04, 0.1], size=(G, S)), int(rng.integers(0, 2 ** 30)),
                                      str(rng.choice(["log0", "column", "group", "rows"])))
    elif kind == "ranks": X = np.argsort(np.argsort(rng.random((G, S)), axis=0), axis=0)
    else: X = rng.normal(0, 3, size=(G, S))
    return dict(G=G, S=S, ng=ng, labels=labels, X=X, kind=kind, pval_reo=float(rng.choice([0.01, 0.05, 0.3])),
                n_conv=int(rng.choice([1, 5])), n_iter=int(rng.integers(1, 9)), seed=int(rng.integers(0, 2 ** 40)), nref=int(rng.integers(3, G)))

t0 = time.time(); kinds = {}
for n in range(N):
    cs = case(); G = cs["G"]
    gid, lev = pkg.encode_groups(cs["labels"])
    ref0 = pkg.synth.ref_mask(G, cs["nref"], cs["seed"])
    tag = (n, cs["kind"], G, cs["S"], cs["ng"], cs["n_iter"], cs["n_conv"], cs["pval_reo"], cs["nref"])
    if len(sys.argv) > 3: print("start", tag, flush=True)
    # (round 6) the two DEG thresholds vary as well -- drawn here, behind the case's own draws, so that a seed still names the same problems
    pval_deg, padj_deg = float(rng.choice([1.0, 1.0, 0.05, 0.01])), float(rng.choice([0.05, 0.05, 0.2, 1.0]))
    run = pkg.run_identify_degs(cs["X"], cs["labels"], list(range(G)), cs["pval_reo"], pval_deg, padj_deg, ref0, cs["n_iter"], cs["n_conv"], seed=cs["seed"], device=0)
    Xf = np.asarray(cs["X"], dtype=np.float64)
    for cm in run.comparisons:
        if len(sys.argv) > 3: print("  oracle k", cm["k"], flush=True)
        exp, iters, trace = oracle.identify_degs(Xf, gid, len(lev), cs["pval_reo"], pval_deg, padj_deg, ref0, cs["n_iter"], cs["n_conv"], cs["seed"], k=cm["k"])
        assert cm["iters_run"] == iters and cm["trace"] == trace, tag
        assert np.array_equal(cm["result"][:, 2:11], exp[:, 2:11]), tag
        ok = np.isfinite(exp).all(axis=1)
        assert np.allclose(cm["result"][ok][:, :2], exp[ok][:, :2], rtol=0, atol=P_ATOL), tag
        assert np.allclose(cm["result"][ok][:, 11:], exp[ok][:, 11:], rtol=STAT_RTOL, atol=1e-9), tag
    kinds[cs["kind"]] = kinds.get(cs["kind"], 0) + 1
    if n % 25 == 24: print("case %d ok (%.0f s) %s" % (n + 1, time.time() - t0, kinds), flush=True)
print("fuzz ok:", N, "cases", kinds)
