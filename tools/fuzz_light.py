"""Randomized parity run of the LIGHT iteration passes (not part of the test suite): small random problems with the light passes
forced on (REO_LIGHT_MIN_G=64), 20-90 passes, random window / band / histogram depth / cycle watch switches per case, every
comparison checked against the oracle -- trace of every pass, tallies bit for bit, p-values to 1e-6.
python tools/fuzz_light.py [N] [seed]"""
import os, sys, time, faulthandler, numpy as np
faulthandler.enable()
os.environ.setdefault("REO_DEBUG_SEGV", "1")   # a native backtrace if the host side ever crashes again (api.hip)
sys.path.insert(0, '.')
import __graft_entry__ as ge
pkg = ge.load_pkg(); oracle = ge.load_oracle()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 4711)
P_ATOL, STAT_RTOL = 1e-6, 1e-7
os.environ["REO_LIGHT_MIN_G"] = "64"

def case():
    G = int(rng.choice([rng.integers(120, 400), rng.integers(400, 1400), rng.integers(1400, 2600)]))
    ng = int(rng.choice([2, 2, 2, 3]))
    sizes = rng.integers(4, 40, size=ng)
    S = int(sizes.sum())
    labels = np.concatenate([[f"grp{g}"] * int(n) for g, n in enumerate(sizes)])
    if rng.random() < 0.5:
        labels = labels[rng.permutation(S)]
    kind = str(rng.choice(["small_int", "wide_int", "float_band", "ranks", "float_cont"]))
    if kind == "small_int": X = rng.integers(0, int(rng.integers(3, 12)), size=(G, S))
    elif kind == "wide_int": X = rng.integers(-50000, 50000, size=(G, S))
    elif kind == "float_band": X = np.round(rng.normal(5, 1.0, size=(G, S)), 1) + rng.choice([0.0, 0.04, 0.099, 0.1], size=(G, S))
    elif kind == "ranks": X = np.argsort(np.argsort(rng.random((G, S)), axis=0), axis=0)
    else: X = rng.normal(0, 3, size=(G, S))
    # a block of genes with a group effect, so that the cut is not at zero
    if rng.random() < 0.7:
        gsel = rng.random(G) < rng.choice([0.05, 0.2, 0.5])
        first = labels == labels[0]
        X = X.copy(); X[np.ix_(gsel, first)] += (3 if X.dtype.kind == "f" else max(1, int(np.ptp(X) // 8)))
    return dict(G=G, S=S, ng=ng, labels=labels, X=X, kind=kind, pval_reo=float(rng.choice([0.01, 0.05, 0.3])),
                n_conv=int(rng.choice([0, 0, 1, 5])), n_iter=int(rng.integers(20, 91)), seed=int(rng.integers(0, 2 ** 40)),
                nref=int(rng.integers(3, G)), pval_deg=float(rng.choice([1.0, 1.0, 0.2])), padj_deg=float(rng.choice([0.05, 0.3, 0.9])),
                env=dict(REO_LIGHT_WINDOW=str(rng.choice([1, 3, 12, 24])), REO_LIGHT_BAND=str(rng.choice([0, 2, 32])),
                         REO_HIST_BELOW=str(rng.choice([0, 3, 12, 256])), REO_CYCLE=str(rng.choice([0, 1, 1, 1])),
                         REO_LIGHT=str(rng.choice([1, 1, 1, 3, 2]))))

t0 = time.time(); kinds = {}; cycles = 0; skipped = 0; light = 0
for n in range(N):
    cs = case(); G = cs["G"]
    os.environ.update(cs["env"])
    gid, lev = pkg.encode_groups(cs["labels"])
    ref0 = pkg.synth.ref_mask(G, cs["nref"], cs["seed"])
    tag = (n, cs["kind"], G, cs["S"], cs["ng"], cs["n_iter"], cs["n_conv"], cs["pval_reo"], cs["nref"], cs["pval_deg"], cs["padj_deg"], cs["env"])
    if len(sys.argv) > 3: print("start", tag, flush=True)
    run = pkg.run_identify_degs(cs["X"], cs["labels"], list(range(G)), cs["pval_reo"], cs["pval_deg"], cs["padj_deg"], ref0, cs["n_iter"], cs["n_conv"],
                                seed=cs["seed"], device=0, profile=True)
    Xf = np.asarray(cs["X"], dtype=np.float64)
    for cm in run.comparisons:
        exp, iters, trace = oracle.identify_degs(Xf, gid, len(lev), cs["pval_reo"], cs["pval_deg"], cs["padj_deg"], ref0, cs["n_iter"], cs["n_conv"], cs["seed"], k=cm["k"])
        assert cm["iters_run"] == iters and cm["trace"] == trace, tag
        assert np.array_equal(cm["result"][:, 2:11], exp[:, 2:11]), tag
        ok = np.isfinite(exp).all(axis=1)
        assert np.allclose(cm["result"][ok][:, :2], exp[ok][:, :2], rtol=0, atol=P_ATOL), tag
        assert np.allclose(cm["result"][ok][:, 11:], exp[ok][:, 11:], rtol=STAT_RTOL, atol=1e-9), tag
    kinds[cs["kind"]] = kinds.get(cs["kind"], 0) + 1
    if run.info["cycle_period"] > 0: cycles += 1; skipped += run.info["cycle_passes_skipped"]
    if run.timings["k2_launches"] < sum(c["iters_run"] for c in run.comparisons): light += 1
    if (n + 1) % 25 == 0: print(f"{n + 1} cases ok, {time.time() - t0:.0f} s, cycles found in {cycles}, passes skipped {skipped}, cases with light passes {light}", flush=True)
print("OK", N, "cases", kinds, "cycles found in", cycles, "cases, passes skipped", skipped, "; cases with light passes", light)
