"""Randomized check of the pipelined, narrowed upload (round 5; not part of the test suite): random shapes, group layouts, value ranges
(so that chunks travel as 16-bit, 32-bit and raw numbers, and change width on the way), Float64, views with a leading dimension, random
chunk sizes and thread counts -- the whole class table, the tallies and a 6-pass run must equal those of the same problem uploaded in
one copy before the groups are known (REO_EAGER_UPLOAD=0, REO_UPLOAD_THREADS=0).  python tools/fuzz_upload.py [N] [seed]"""
import os, sys, time, hashlib, faulthandler, numpy as np
faulthandler.enable()
os.environ.setdefault("REO_DEBUG_SEGV", "1")
sys.path.insert(0, '.')
import __graft_entry__ as ge
pkg = ge.load_pkg()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 555)


def digest(ctx, G, ref0, ng):
    h = hashlib.blake2b(digest_size=8)
    for k in range(1 if ng == 2 else ng):
        ctx.build_pairs(k)
        for i0 in range(0, G, 2048):
            h.update(ctx.get_codes(i0, min(G, i0 + 2048), 0, G).tobytes())
        res, it, tr = ctx.identify_degs(ref0, 1.0, 0.05, 6, 0)
        h.update(np.asarray(tr, dtype=np.int64).tobytes()); h.update(np.ascontiguousarray(res).tobytes())
    h.update(bytes([ctx.info()["has_ties"]]))
    return h.hexdigest()


t0 = time.time(); kinds = {}; nranges = 0
for n in range(N):
    G = int(rng.choice([rng.integers(300, 3000), rng.integers(3000, 12000), rng.integers(12000, 40000)]))
    S = int(rng.choice([rng.integers(8, 64), rng.integers(64, 300), rng.integers(300, 700)]))
    ng = int(rng.choice([2, 2, 2, 3, 4]))
    layout = str(rng.choice(["contiguous", "interleaved", "unequal"]))
    if layout == "contiguous": labels = np.repeat(np.arange(ng), -(-S // ng))[:S]
    elif layout == "interleaved": labels = rng.integers(0, ng, S); labels[:ng] = np.arange(ng)
    else:
        cut = np.sort(rng.choice(np.arange(2, S - 2), ng - 1, replace=False)); labels = np.searchsorted(cut, np.arange(S), side="right")
    kind = str(rng.choice(["ranks", "counts16", "counts32", "huge", "growing", "negative", "float", "float_band", "float_counts", "float_f32", "float_mixed"]))
    if kind == "ranks": X = np.argsort(np.argsort(rng.random((G, S)), axis=0), axis=0).astype(np.int64)
    elif kind == "counts16": X = rng.integers(0, int(rng.integers(3, 30000)), size=(G, S))
    elif kind == "counts32": X = np.floor(np.exp(rng.normal(4, 3, size=(G, S)))).astype(np.int64) % (2 ** 31)
    elif kind == "huge": X = rng.integers(-2 ** 45, 2 ** 45, size=(G, S))
    elif kind == "growing":
        X = rng.integers(0, 20000, size=(G, S)); c1, c2 = sorted(rng.integers(1, S, 2)); X[:, c1:] += 50000; X[rng.integers(0, G), c2:] = 2 ** 33
    elif kind == "negative": X = rng.integers(-32768, 32768, size=(G, S))
    elif kind == "float_counts": X = rng.integers(-100, int(rng.integers(3, 100000)), size=(G, S)).astype(np.float64)
    elif kind == "float_f32": X = rng.lognormal(1, 2, size=(G, S)).astype(np.float32).astype(np.float64)
    elif kind == "float_mixed":
        X = rng.integers(0, 500, size=(G, S)).astype(np.float64); c1, c2 = sorted(rng.integers(1, S, 2)); X[:, c1:] *= 0.5; X[:, c2:] += rng.normal(0, 1e-9, size=(G, S - c2))
        X[rng.integers(0, G), rng.integers(0, S)] = -0.0
    elif kind == "float": X = np.log2(1.0 + np.floor(np.exp(rng.normal(2.0, 2.0, size=(G, S))))) + rng.uniform(0, 0.05, (G, S))
    else: X = np.round(rng.normal(5, 1.0, size=(G, S)), 1) + rng.choice([0.0, 0.04, 0.099, 0.1], size=(G, S))
    X = np.asfortranarray(X)
    if rng.random() < 0.3:   # a view with a leading dimension
        big = np.asfortranarray(np.full((G + 13, S), 7, dtype=X.dtype)); big[5:5 + G, :] = X; X = big[5:5 + G, :]
    gid, lev = pkg.encode_groups(labels)
    ref0 = pkg.synth.ref_mask(G, max(3, G // 5), n)
    pval = float(rng.choice([0.01, 0.05, 0.3]))
    os.environ.update(REO_EAGER_UPLOAD="0", REO_UPLOAD_THREADS="0"); os.environ.pop("REO_EAGER_CHUNK", None); os.environ.pop("REO_EAGER_RANGES", None)
    with pkg.Context(device=0, seed=n) as ctx:
        ctx.set_matrix(X); ctx.set_groups(gid, len(lev)); ctx.compute_thresholds(pval)
        want = digest(ctx, G, ref0, len(lev))
    os.environ.update(REO_EAGER_UPLOAD=str(rng.choice(["2", "2", "1"])), REO_UPLOAD_THREADS=str(rng.choice(["12", "1", "5", "0"])))
    if rng.random() < 0.5: os.environ["REO_EAGER_CHUNK"] = str(int(rng.integers(1, S + 5)))
    if rng.random() < 0.7: os.environ["REO_EAGER_RANGES"] = str(int(rng.integers(1, 7)))   # (round 6) ranges of sample blocks per side; unset: by shape
    tag = (n, kind, layout, G, S, ng, os.environ["REO_EAGER_UPLOAD"], os.environ["REO_UPLOAD_THREADS"], os.environ.get("REO_EAGER_CHUNK"), os.environ.get("REO_EAGER_RANGES"))
    with pkg.Context(device=0, seed=n) as ctx:
        ctx.set_groups(gid, len(lev)); ctx.compute_thresholds(pval); ctx.set_matrix(X)
        got = digest(ctx, G, ref0, len(lev))
        assert got == want, tag
        nranges += ctx.info()["eager_range_launches"]
        ctx.set_matrix(X)                      # the same context again (staging ring, work lists and streams reused)
        assert digest(ctx, G, ref0, len(lev)) == want, tag
    kinds[kind] = kinds.get(kind, 0) + 1
    if n % 20 == 19: print("case %d ok (%.0f s) %s" % (n + 1, time.time() - t0, kinds), flush=True)
print("fuzz upload ok:", N, "cases", kinds, "launches over ranges of a side:", nranges)
