"""Generates tests/golden/*.json.

Run in the build container (needs /root/reference for the bundled test data):
    python tests/golden/make_golden.py
The reference is Julia and cannot run here, so expected values come from the
CPU oracle (oracle/reo_oracle.c), every one cross-checked against the
independent numpy/scipy restatement (oracle/reo_numpy.py) before it is
written; the McCullagh known-answer values are the reference's own
(src/RankCompV3.jl:207-221).  Fixtures are data only: inputs + expected outputs.
"""
from __future__ import annotations

import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402
from oracle import reo_numpy as rn  # noqa: E402

REF = "/root/reference"
SEED = 0x5EED0001


def dump(name, obj):
    with open(os.path.join(HERE, name), "w") as f:
        json.dump(obj, f, separators=(",", ":"))
    print("wrote", name, os.path.getsize(os.path.join(HERE, name)), "bytes")


def case(X, gid, ngroups, ref0, pval_reo, pval_deg, padj_deg, n_iter, n_conv, seed, k=0):
    G = X.shape[0]
    sizes = np.bincount(gid, minlength=ngroups)
    thr = [oracle.threshold(int(sizes[k]), pval_reo), oracle.threshold(int(sizes.sum() - sizes[k]), pval_reo)]
    assert thr == [rn.threshold(int(sizes[k]), pval_reo), rn.threshold(int(sizes.sum() - sizes[k]), pval_reo)]
    gt, eq = oracle.pair_counts(X, gid, ngroups, 0, G, 0, G)
    gt2, eq2 = rn.pair_counts(X, gid, ngroups)
    assert np.array_equal(gt, gt2) and np.array_equal(eq, eq2)
    code = oracle.build_codes(X, gid, ngroups, k, thr, seed)
    assert np.array_equal(code, rn.build_codes(X, gid, ngroups, k, thr, seed))
    cont = oracle.tally(code, ref0)
    assert np.array_equal(cont, rn.tally(code, ref0))
    res, iters, trace = oracle.iterate(code, ref0, pval_deg, padj_deg, n_iter, n_conv)
    res2, iters2, trace2 = rn.iterate(code, ref0, pval_deg, padj_deg, n_iter, n_conv)
    assert iters == iters2 and trace == trace2 and np.allclose(res, res2, rtol=1e-10, atol=1e-12)
    return {
        "X": X.tolist(), "gid": gid.tolist(), "ngroups": ngroups, "ref0": np.asarray(ref0, dtype=int).tolist(),
        "pval_reo": pval_reo, "pval_deg": pval_deg, "padj_deg": padj_deg, "n_iter": n_iter, "n_conv": n_conv,
        "seed": seed, "k": k, "thr": thr, "n_gt": gt.tolist(), "n_eq": eq.tolist(), "code": code.tolist(),
        "cont": cont.tolist(), "result": res.tolist(), "iters_run": iters, "trace": [list(t) for t in trace],
        "labels": rn.labels(res, pval_deg, padj_deg).tolist(),
    }


def main():
    # (1) McCullagh KAT: the reference's own numbers
    dump("mccullagh_kat.json", {
        "source": "src/RankCompV3.jl:207-221, test/McCullagh_test.jl:39",
        "mat": [[43, 8, 3, 0], [2, 2, 5, 3], [1, 0, 7, 2], [0, 0, 1, 5]],
        "expected": [0.005469174895116946, 1.4504988072997458, 1.502600073417028, 0.5221345956920705, 2.778017046308073],
        "N": [[14, 4, 0], [4, 12, 3], [0, 3, 6]], "R": [11, 11, 5],
        "paper_rounded": {"delta1": 1.45, "delta2": 1.50, "se": 0.53},
    })
    # (2) stable-REO threshold table m(n), pval_reo = 0.01 and 0.05
    ns = [2, 3, 5, 7, 8, 10, 11, 32, 64, 100, 500, 2000, 4000]
    tab = {}
    for p in (0.01, 0.05):
        row = {}
        for n in ns:
            m = oracle.threshold(n, p)
            assert m == rn.threshold(n, p), (n, p)
            row[str(n)] = m
        tab[str(p)] = row
    assert [tab["0.01"][str(n)] for n in (5, 7, 8, 10, 32, 64, 100, 500, 2000, 4000)] == \
        [5, 7, 8, 10, 24, 43, 64, 280, 1059, 2082]  # SURVEY.md §7 [analysis]
    dump("thresholds.json", tab)

    # (3) 64-gene slice of the reference's bundled test data (test/fn_expr.txt, test/fn_meta.txt)
    rows = []
    with open(os.path.join(REF, "test", "fn_expr.txt")) as f:
        header = f.readline().rstrip("\n").split("\t")
        for line in f:
            rows.append(line.rstrip("\n").split("\t"))
    names = [r[0] for r in rows]
    X = np.array([[int(v) for v in r[1:]] for r in rows], dtype=np.int64)
    meta = [l.rstrip("\n").split("\t") for l in open(os.path.join(REF, "test", "fn_meta.txt"))][1:]
    assert [m[0] for m in meta] == header[1:]
    gid, lev = rn.group_ids([m[1] for m in meta])
    pick = list(range(0, 40)) + list(range(10000, 10024))  # DE.. and EE.. genes
    Xs = X[pick].astype(np.float64)
    ref0 = np.zeros(64, dtype=bool)
    ref0[::3] = True
    c = case(Xs, gid, 2, ref0, 0.01, 1.0, 0.05, 16, 1, SEED)
    c["gene_names"] = [names[i] for i in pick]
    c["source"] = "rows 1-40 and 10001-10024 of test/fn_expr.txt; groups from test/fn_meta.txt"
    dump("bundled_slice64.json", c)

    # (4) hand-checkable example: all nine classes + the singular-N branch.
    # 6 samples per group -> m = 6: a pair is stable only if all 6 samples agree.
    # gene:      A        B        C        D   (ctrl samples | treat samples)
    Xh = np.array([
        [1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1],        # A low everywhere
        [5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5],        # B mid everywhere
        [9, 9, 9, 9, 9, 9, 0, 0, 0, 0, 0, 0],        # C high in ctrl, lowest in treat
        [0, 0, 0, 0, 0, 0, 9, 9, 9, 9, 9, 9],        # D lowest in ctrl, high in treat
        [3, 7, 3, 7, 3, 7, 5, 5, 5, 5, 5, 6],        # E unstable vs B in ctrl, tied with B in treat
        [5, 5, 5, 5, 5, 5, 3, 7, 3, 7, 3, 7],        # F tied with B in ctrl, unstable vs B in treat
        [2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2],
        [8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8],
        [4, 6, 4, 6, 4, 6, 4, 6, 4, 6, 4, 6],
        [6, 4, 6, 4, 6, 4, 6, 4, 6, 4, 6, 4],
        [1, 9, 1, 9, 1, 9, 9, 1, 9, 1, 9, 1],
        [7, 7, 7, 7, 7, 7, 1, 1, 1, 1, 1, 1],
    ], dtype=np.float64)
    gidh = np.array([0] * 6 + [1] * 6, dtype=np.int32)
    refh = np.ones(12, dtype=bool)
    ch = case(Xh, gidh, 2, refh, 0.05, 1.0, 0.05, 4, 1, SEED)
    classes = set(np.asarray(ch["code"]).ravel().tolist()) - {255}
    assert classes == set(range(9)), classes
    sing = [i for i in range(12) if oracle.mccullagh(np.asarray(ch["cont"][i]).reshape(3, 3))[0][0] == 1.0]
    assert sing, "example must exercise the singular-N branch"
    ch["singular_rows"] = sing
    dump("hand12.json", ch)

    # (4b) three groups, one-vs-rest (:375-390,396-436): 48 genes of the bundled data, samples relabelled a/b/c
    gid3 = np.array([0, 1, 2, 0, 1, 2, 0, 1, 2, 1], dtype=np.int32)
    X3 = X[list(range(0, 30)) + list(range(12000, 12018))].astype(np.float64)
    ref3 = np.ones(48, dtype=bool)
    ref3[5::7] = False
    multi = {"X": X3.tolist(), "gid": gid3.tolist(), "ngroups": 3, "ref0": ref3.astype(int).tolist(),
             "comparisons": [case(X3, gid3, 3, ref3, 0.3, 1.0, 0.05, 6, 1, SEED, k=k) for k in range(3)]}
    for cm in multi["comparisons"]:
        for key in ("X", "gid", "ngroups", "ref0", "n_gt", "n_eq"):
            cm.pop(key)
    gt3, eq3 = oracle.pair_counts(X3, gid3, 3, 0, 48, 0, 48)
    multi["n_gt"], multi["n_eq"] = gt3.tolist(), eq3.tolist()
    dump("three_groups48.json", multi)

    # (5) BH + trimmed-std vectors, incl. half-even rounding and the G=10 error path
    rng = np.random.default_rng(12345)
    vec = {}
    for G in (10, 20, 30, 50, 101, 250):
        d = np.round(rng.normal(0, 1.5, G), 6)
        p = np.round(rng.uniform(0, 1, G) ** 2, 6)
        p[: G // 5] = p[0]  # ties
        entry = {"d": d.tolist(), "p": p.tolist(), "bh": oracle.bh(p).tolist()}
        assert np.allclose(oracle.bh(p), rn.bh(p), rtol=1e-14, atol=0)
        try:
            entry["trimmed_std"] = oracle.trimmed_std(d)
            assert abs(entry["trimmed_std"] - rn.trimmed_std(d)) < 1e-13
            entry["slice"] = [rn.jl_round(G * 0.05), rn.jl_round(G * 0.95)]
        except IndexError:
            entry["trimmed_std"] = None  # reference: BoundsError (index 0)
        vec[str(G)] = entry
    assert vec["10"]["trimmed_std"] is None and vec["30"]["slice"] == [2, 28] and vec["50"]["slice"] == [2, 48]
    dump("bh_trimmed_std.json", vec)


if __name__ == "__main__":
    main()
