"""Randomized parity run of the rank/band transform at LARGE gene counts (not part of the test suite): random G up to 65 535 (with a
third argument `big`: 65 536 ... 262 143, the records-in-L2 form t_sample_big, form 3),
few samples, every kind of data (ranks, small counts, long-tailed counts, wide and negative integers, log-expression floats,
floats on a 0.1 grid), random pair blocks compared with the oracle's literal comparator; which form of the transform ran is
tallied (1 histogram forms, 2 bucket form, 0 segmented sort).  python tools/fuzz_transform.py [N] [seed]"""
import os, sys, time, faulthandler, numpy as np
faulthandler.enable()
os.environ.setdefault("REO_DEBUG_SEGV", "1")
sys.path.insert(0, '.')
import __graft_entry__ as ge
pkg = ge.load_pkg(); oracle = ge.load_oracle()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 31337)

BIG = len(sys.argv) > 3 and sys.argv[3] == "big"

def case():
    G = int(rng.choice([rng.integers(1000, 20481), rng.integers(20481, 32769), rng.integers(32769, 41473), rng.integers(41473, 58369), rng.integers(58369, 65536), 65535]))
    if BIG: G = int(rng.choice([65536, rng.integers(65536, 90000), rng.integers(90000, 180000), rng.integers(180000, 262144), 262143]))
    S = int(rng.integers(6, 15)) if not BIG else int(rng.integers(2, 7))
    kind = str(rng.choice(["ranks", "small_int", "tail", "wide_int", "big_int", "float_expr", "float_band", "float_cont", "float_zeros", "infinities"]))
    if kind == "ranks": X = np.argsort(np.argsort(rng.random((G, S)), axis=0), axis=0)
    elif kind == "small_int": X = rng.integers(0, int(rng.integers(2, 40)), size=(G, S))
    elif kind == "tail":
        X = np.minimum(np.floor(np.exp(rng.normal(3.0, float(rng.uniform(1.5, 3.0)), size=(G, S)))).astype(np.int64), (1 << int(rng.integers(17, 25))) - 1)
        X[rng.random((G, S)) < 0.3] = 0
    elif kind == "wide_int": X = rng.integers(-50000, 50000, size=(G, S))
    elif kind == "big_int": X = rng.integers(0, 2 ** 40, size=(G, S))
    elif kind == "float_expr": X = np.log2(1.0 + np.floor(np.exp(rng.normal(2.0, 2.0, size=(G, S))))) + rng.uniform(0, 0.05, (G, S))
    elif kind == "float_band": X = np.round(rng.normal(5, 1.0, size=(G, S)), 1) + rng.choice([0.0, 0.04, 0.099, 0.1], size=(G, S))
    elif kind == "infinities":   # log(0) = -Inf where counts are zero, a few +Inf, sometimes whole samples / whole genes (is_greater on them: :72-76)
        cnt = np.floor(np.exp(rng.normal(1.0, 2.0, size=(G, S))))
        with np.errstate(divide="ignore"): X = np.log2(cnt) + np.where(cnt > 0, rng.uniform(0, 0.05, (G, S)), 0.0)
        X[rng.random((G, S)) < float(rng.choice([0.0, 0.001, 0.05]))] = np.inf
        if rng.random() < 0.3: X[:, int(rng.integers(0, S))] = -np.inf if rng.random() < 0.5 else np.inf
        if rng.random() < 0.3: X[rng.random(G) < 0.01, :] = -np.inf
        if rng.random() < 0.3: X[[0, G - 1], :] = rng.choice([-np.inf, np.inf], size=(2, 1))
    elif kind == "float_zeros": X = np.where(rng.random((G, S)) < 0.5, 0.0, rng.lognormal(1.0, 1.0, (G, S)))
    else: X = rng.normal(0, 3, size=(G, S))
    return G, S, kind, X

t0 = time.time(); forms = {}
for n in range(N):
    G, S, kind, X = case()
    gid = (np.arange(S) % 2).astype(np.int32)
    blocks = []
    for _ in range(4):
        i0 = int(rng.integers(0, G - 24)); j0 = int(rng.integers(0, G - 48))
        blocks.append((i0, i0 + 24, j0, j0 + 48))
    blocks.append((G - 24, G, 0, 48)); blocks.append((0, 24, G - 48, G))
    with pkg.Context(device=0, seed=n) as ctx:
        ctx.set_matrix(X); ctx.set_groups(gid, 2)
        out = [ctx.pair_counts(*b) for b in blocks]
        form = ctx.info()["transform_in_lds"]
    Xf = np.asarray(X, dtype=np.float64)
    for blk, (gt, eq) in zip(blocks, out):
        egt, eeq = (oracle.pair_counts_as_evaluated if kind == "infinities" else oracle.pair_counts)(Xf, gid, 2, *blk)
        assert np.array_equal(gt, egt) and np.array_equal(eq, eeq), (n, kind, G, S, blk, form)
    forms[(kind, form)] = forms.get((kind, form), 0) + 1
    if n % 10 == 9: print("case %d ok (%.0f s)" % (n + 1, time.time() - t0), flush=True)
print("fuzz transform ok:", N, "cases; (kind, form) counts:", dict(sorted(forms.items())))
