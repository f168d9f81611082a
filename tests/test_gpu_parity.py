"""Parity of the HIP path (through the C ABI) against the CPU oracle.

Integer outputs (pair counts, class codes, tallies, iteration trace) must be
bit-exact; floating-point statistics within the tolerances written below
(north star: p-values within 1e-6).
"""
import os
import zlib

import numpy as np

import sharding_mirror
import pytest

pytestmark = pytest.mark.gpu

P_ATOL = 1e-6      # north-star tolerance on pval / padj
STAT_RTOL = 1e-7   # delta1, delta2, se, z1 (fp64 log/sqrt, closed-form vs Gauss-Jordan 2x2 inverse)


def _setup(pkg, X, group, seed=0, pval_reo=0.01):
    gid, lev = pkg.encode_groups(group)
    ctx = pkg.Context(device=0, seed=seed)
    ctx.set_matrix(X)
    ctx.set_groups(gid, len(lev))
    ctx.compute_thresholds(pval_reo)
    return ctx, gid, len(lev)


def _check_result(res, exp):
    assert np.array_equal(res[:, 2:11], exp[:, 2:11]), "tallies differ"
    assert np.allclose(res[:, :2], exp[:, :2], rtol=0, atol=P_ATOL), np.abs(res[:, :2] - exp[:, :2]).max()
    assert np.allclose(res[:, 11:], exp[:, 11:], rtol=STAT_RTOL, atol=1e-9), np.abs(res[:, 11:] - exp[:, 11:]).max()


@pytest.mark.parametrize("name", ["bundled_slice64.json", "hand12.json"])
def test_golden_fixtures(pkg, golden, name):
    g = golden(name)
    X = np.array(g["X"], dtype=np.float64)
    G = X.shape[0]
    for dtype in (np.float64, np.int64):
        with pkg.Context(device=0, seed=g["seed"]) as ctx:
            ctx.set_matrix(X.astype(dtype))
            ctx.set_groups(g["gid"], g["ngroups"])
            thr = ctx.compute_thresholds(g["pval_reo"])
            assert thr[:, 0].tolist() == g["thr"]
            gt, eq = ctx.pair_counts(0, G, 0, G)
            assert np.array_equal(gt, g["n_gt"]) and np.array_equal(eq, g["n_eq"])
            ctx.build_pairs(0)
            assert np.array_equal(ctx.get_codes(0, G, 0, G), g["code"])
            assert np.array_equal(ctx.tally(np.array(g["ref0"])), g["cont"])
            res, iters, trace = ctx.identify_degs(np.array(g["ref0"]), g["pval_deg"], g["padj_deg"], g["n_iter"], g["n_conv"])
            assert iters == g["iters_run"] and [list(t) for t in trace] == g["trace"]
            _check_result(res, np.array(g["result"]))
            assert pkg.label_genes(res, g["pval_deg"], g["padj_deg"]).tolist() == g["labels"]


def test_mccullagh_device_routine(pkg, oracle, golden):
    """Device McCullagh (3x3) against the oracle's general k x k routine, incl. the singular branch."""
    rng = np.random.default_rng(3)
    cont = rng.integers(0, 400, size=(2000, 9)).astype(np.int32)
    cont[:50, [1, 2, 3, 5, 6, 7]] = 0           # N == 0 -> singular
    cont[50:60] = [5, 3, 0, 3, 9, 0, 0, 0, 7]   # a*d == b*b with a = d = b? (n13 = n31 = 0 -> b = 0, d = 0)
    cont[60:70] = [0, 0, 4, 0, 1, 0, 4, 0, 0]   # a = d = b = 8
    with pkg.Context(device=0) as ctx:
        out = ctx.mccullagh(cont)
    sing = 0
    for i in range(cont.shape[0]):
        exp, _, _ = oracle.mccullagh(cont[i].reshape(3, 3))
        if exp[0] == 1.0 and exp[1] == 0.0:
            sing += 1
            assert out[i].tolist() == [1.0, 0.0, 0.0, 0.0, 0.0]
        else:
            assert abs(out[i, 0] - exp[0]) <= P_ATOL
            assert np.allclose(out[i, 1:], exp[1:], rtol=STAT_RTOL, atol=1e-12)
    assert sing >= 70


def _abd_tables(bs, splits=None):
    """3x3 tables with n12 = n21 = n23 = n32 = 0: N = [[b b][b b]], integer-singular for every b."""
    rows = []
    for b in bs:
        for n13 in (sorted({0, 1, b // 2, b - 1, b}) if splits is None else splits(b)):
            for diag in ((5, 3, 2), (0, 0, 0)):
                rows.append([diag[0], 0, n13, 0, diag[1], 0, b - n13, 0, diag[2]])
    return np.array(rows, dtype=np.int32)


def test_mccullagh_integer_singular_tables_follow_the_float_elimination(pkg, oracle):
    """:242 tests abs(det(N)) <= eps() on the LU of Float64.(N), not a*d == b*b.  For a = b = d > 0 the
    elimination's l21 = b * (1.0 / b) is not 1 for 286 of the b in 1..2999 (49, 98, 103, 107, 161, 187, ...):
    det = b * u22 > eps and the reference computes on with that factorisation's inverse.  The device takes the
    oracle's arithmetic there (kernels.hip, mccullagh3): every such table equal to oracle.mccullagh."""
    named = [8, 49, 98, 103, 107, 161, 187]
    cont = np.concatenate([_abd_tables(named), _abd_tables(range(1, 3001), lambda b: sorted({0, b // 3, b // 2, b}))])
    with pkg.Context(device=0) as ctx:
        out = ctx.mccullagh(cont)
    live, live_b = 0, set()
    for i in range(cont.shape[0]):
        exp, _, _ = oracle.mccullagh(cont[i].reshape(3, 3))
        b = int(cont[i, 2] + cont[i, 6])
        if exp[3] == 0.0:   # the singular branch: (1, 0, 0, 0, 0)
            assert out[i].tolist() == [1.0, 0.0, 0.0, 0.0, 0.0], (b, out[i])
            assert b * (1.0 / b) == 1.0
        else:
            live += 1; live_b.add(b)
            assert b * (1.0 / b) != 1.0
            assert abs(out[i, 0] - exp[0]) <= P_ATOL, (b, out[i], exp)
            assert np.allclose(out[i, 1:], exp[1:], rtol=STAT_RTOL, atol=1e-12), (b, out[i], exp)
            n13 = float(cont[i, 2])     # delta1 is robust in this corner: both log terms are equal, the weights sum to 1
            assert np.isclose(out[i, 1], np.log((n13 + 0.5) / (b - n13 + 0.5)), rtol=1e-9, atol=1e-12)
    assert live_b >= {49, 98, 103, 107, 161, 187} and 8 not in live_b
    assert len([b for b in live_b if b < 3000]) == 286 and live > 2000


@pytest.mark.parametrize("b,n13,spg,corners,dtype", [(49, 24, 2, 1, np.int64), (49, 24, 2, 3, np.float64), (107, 30, 2, 2, np.int64),
                                                     (161, 80, 3, 1, np.int64), (98, 49, 4, 2, np.float64), (8, 4, 2, 1, np.int64)])
def test_identify_degs_on_integer_singular_tables(pkg, oracle, b, n13, spg, corners, dtype):
    """A whole run in which genes HAVE such tables: two to four samples per group, the corner genes' partners
    are all n11 / n13 / n22 / n31 / n33 with n13 + n31 = b (synth.lu_corner).  First pass and a run of passes."""
    X, group, ci = pkg.synth.lu_corner(b, n13, spg, corners, seed=5)
    G = X.shape[0]
    gid, lev = pkg.encode_groups(group)
    ref0 = np.ones(G, dtype=bool)
    names = list(range(G))
    Xf = X.astype(np.float64)
    hit = 0
    for n_iter, n_conv in ((1, 0), (6, 0), (6, 1)):
        exp, iters, trace = oracle.identify_degs(Xf, gid, 2, 0.01, 1.0, 0.05, ref0, n_iter, n_conv, 7)
        run = pkg.run_identify_degs(X.astype(dtype), group, names, 0.01, 1.0, 0.05, ref0, n_iter, n_conv, seed=7, device=0)
        assert run.iters_run == iters and run.trace == trace
        _check_result(run.result, exp)
        if n_iter == 1:
            for c in ci:
                t = exp[c, 2:11].astype(int)
                assert t[1] == t[3] == t[5] == t[7] == 0 and t[2] + t[6] == b and t[2] == n13
                if b * (1.0 / b) != 1.0:
                    assert exp[c, 13] > 0 and run.result[c, 13] > 0       # se: the non-singular branch ran
                    hit += 1
                else:
                    assert exp[c, 13] == 0 and run.result[c, 13] == 0
    assert hit == (corners if b != 8 else 0)


@pytest.mark.parametrize("family,G,S", [("t0", 700, 40), ("t1", 700, 40), ("float", 500, 31), ("t1", 333, 9)])
def test_counts_codes_tallies_bit_exact(pkg, oracle, family, G, S):
    seed = 0x5EED0002
    X = {"t0": pkg.synth.t0_ranks, "t1": pkg.synth.t1_counts, "float": pkg.synth.float_expr}[family](G, S, seed)
    group = pkg.synth.groups(S)
    ctx, gid, ng = _setup(pkg, X, group, seed)
    with ctx:
        Xf = X.astype(np.float64)
        # blocks on and off the diagonal, ragged edges, i > j
        for (i0, i1, j0, j1) in [(0, 70, 0, 70), (0, 33, G - 300, G), (G - 37, G, 5, 130), (250, 290, 250, 330)]:
            gt, eq = ctx.pair_counts(i0, i1, j0, j1)
            egt, eeq = oracle.pair_counts(Xf, gid, ng, i0, i1, j0, j1)
            assert np.array_equal(gt, egt) and np.array_equal(eq, eeq), (family, i0, i1, j0, j1)
        info_ties = (family != "t0")
        ctx.build_pairs(0)
        assert bool(ctx.info()["has_ties"]) == info_ties
        thr = ctx.get_thresholds()[:, 0]
        code = oracle.build_codes(Xf, gid, ng, 0, thr, seed)
        assert np.array_equal(ctx.get_codes(0, G, 0, G), code)
        rng = np.random.default_rng(1)
        for frac in (0.0, 0.3, 1.0):
            ref = rng.random(G) < frac
            assert np.array_equal(ctx.tally(ref), oracle.tally(code, ref))


@pytest.mark.parametrize("family,G,S,n_conv", [("t0", 900, 60, 1), ("t1", 900, 60, 5), ("float", 400, 24, 1)])
def test_identify_degs_matches_oracle(pkg, oracle, family, G, S, n_conv):
    seed = 0x5EED0003
    X = {"t0": pkg.synth.t0_ranks, "t1": pkg.synth.t1_counts, "float": pkg.synth.float_expr}[family](G, S, seed)
    group = pkg.synth.groups(S)
    ref0 = pkg.synth.ref_mask(G, 150, seed)
    names = [f"g{i}" for i in range(G)]
    run = pkg.run_identify_degs(X, group, names, 0.01, 1.0, 0.05, ref0, 16, n_conv, seed=seed, device=0)
    gid, lev = pkg.encode_groups(group)
    exp, iters, trace = oracle.identify_degs(X.astype(np.float64), gid, len(lev), 0.01, 1.0, 0.05, ref0, 16, n_conv, seed)
    assert run.iters_run == iters and run.trace == trace
    _check_result(run.result, exp)
    assert run.res.shape == (G, 17) and run.res[0, 0] == "g0"
    from oracle import reo_numpy as rn
    assert run.labels.tolist() == rn.labels(exp, 1.0, 0.05).tolist()


def test_forced_iterations_and_zero_iterations(pkg, oracle):
    G, S, seed = 500, 30, 11
    X = pkg.synth.t1_counts(G, S, seed)
    group = pkg.synth.groups(S)
    ref0 = pkg.synth.ref_mask(G, 80, seed)
    gid, lev = pkg.encode_groups(group)
    names = list(range(G))
    # n_conv = 0 makes :419 never true -> exactly n_iter passes
    run = pkg.run_identify_degs(X, group, names, 0.01, 1.0, 0.05, ref0, 7, 0, seed=seed, device=0)
    exp, iters, trace = oracle.identify_degs(X.astype(np.float64), gid, 2, 0.01, 1.0, 0.05, ref0, 7, 0, seed)
    assert run.iters_run == iters == 7 and run.trace == trace
    _check_result(run.result, exp)
    # n_iter = 0: the loop never runs, result stays zeros(r,15) (:398) and every label is "no change"
    run0 = pkg.run_identify_degs(X, group, names, 0.01, 1.0, 0.05, ref0, 0, 5, seed=seed, device=0)
    assert run0.iters_run == 0 and not run0.result.any() and set(run0.labels) == {"no change"}


def test_group_order_and_interleaved_samples(pkg, oracle):
    """Groups need not be contiguous; ctrl is the group of the FIRST sample (:353,358)."""
    G, S, seed = 300, 22, 21
    X = pkg.synth.t1_counts(G, S, seed)
    rng = np.random.default_rng(5)
    group = np.array(["B", "A"])[rng.integers(0, 2, S)]
    group[0] = "B"
    ref0 = pkg.synth.ref_mask(G, 60, seed)
    gid, lev = pkg.encode_groups(group)
    assert lev[0] == "B"
    run = pkg.run_identify_degs(X, group, list(range(G)), 0.05, 1.0, 0.05, ref0, 6, 1, seed=seed, device=0)
    exp, iters, trace = oracle.identify_degs(X.astype(np.float64), gid, 2, 0.05, 1.0, 0.05, ref0, 6, 1, seed)
    assert run.iters_run == iters and run.trace == trace
    _check_result(run.result, exp)


def test_error_paths(pkg):
    X = pkg.synth.t1_counts(9, 10, 1)
    group = pkg.synth.groups(10)
    with pytest.raises(pkg.DimensionMismatch):  # G < 10: slice of :411 starts at index 0 (BoundsError)
        pkg.identify_degs(X, group, list(range(9)), 0.01, 1.0, 0.05, np.ones(9, bool), 4, 1, device=0)
    Xn = pkg.synth.float_expr(50, 10, 1)
    Xn[3, 4] = np.nan          # refused: every comparison with a NaN is false -- row order, not an ordering (transform.hip, Codec<double>)
    with pytest.raises(pkg.DimensionMismatch, match="contains NaN"):
        pkg.identify_degs(Xn, group, list(range(50)), 0.01, 1.0, 0.05, np.ones(50, bool), 4, 1, device=0)
    with pkg.Context(device=0) as ctx:      # matrix first: the call that ranks reports it
        ctx.set_matrix(Xn)
        with pytest.raises(pkg.DimensionMismatch, match="contains NaN"):
            ctx.set_groups((np.arange(10) >= 5).astype(np.int32), 2) or ctx.pair_counts(0, 4, 0, 4)
    with pkg.Context(device=0) as ctx:
        with pytest.raises(pkg.DimensionMismatch):
            ctx.build_pairs(0)  # nothing set
        ctx.set_matrix(pkg.synth.t1_counts(40, 10, 1))
        with pytest.raises(pkg.DimensionMismatch):
            ctx.set_groups([0, 1, 0], 2) or ctx.pair_counts(0, 4, 0, 4)  # 3 labels for 10 columns (:355)
        with pytest.raises(pkg.DimensionMismatch):
            ctx.set_groups([1, 0] * 5, 2)  # ids must follow first appearance


@pytest.mark.parametrize("kind", ["log0", "column", "group", "rows"])
def test_infinities_are_compared_as_the_reference_compares_them(pkg, oracle, kind, monkeypatch):
    """is_greater on +-Inf (:72-76): equal infinities are neither tied nor greater (abs(Inf - Inf) = NaN), deterministic; an
    infinity against anything else compares as usual.  log(0) = -Inf tables, a whole sample of -Inf, -Inf in one group only,
    whole genes infinite: counts, class table, tallies and the run equal to the oracle's literal comparator -- in both orders of
    calls (matrix first; groups first = the pipelined upload), and as float32-representable data (the narrowed upload)."""
    G, S, seed = 500, 24, 0x5EED0062
    X = pkg.synth.with_infinities(pkg.synth.float_expr(G, S, seed), seed, kind)
    group = pkg.synth.groups(S)
    gid, lev = pkg.encode_groups(group)
    ref0 = pkg.synth.ref_mask(G, 150, seed)
    blocks = [(0, 70, 0, 70), (0, 33, G - 300, G), (G - 37, G, 5, 130), (250, 290, 250, 330)]
    for form in ("f64", "f32"):
        Xv = X if form == "f64" else X.astype(np.float32).astype(np.float64)
        thr = [oracle.threshold(12), oracle.threshold(12)]
        code = oracle.build_codes(Xv, gid, 2, 0, thr, seed)
        exp, iters, trace = oracle.identify_degs(Xv, gid, 2, 0.01, 1.0, 0.05, ref0, 8, 1, seed)
        for order in ("matrix_first", "groups_first"):
            with pkg.Context(device=0, seed=seed) as ctx:
                if order == "matrix_first":
                    ctx.set_matrix(Xv); ctx.set_groups(gid, 2); ctx.compute_thresholds(0.01)
                else:
                    ctx.set_groups(gid, 2); ctx.compute_thresholds(0.01); ctx.set_matrix(Xv)
                for blk in blocks:
                    gt, eq = ctx.pair_counts(*blk)
                    egt, eeq = oracle.pair_counts_as_evaluated(Xv, gid, 2, *blk)
                    assert np.array_equal(gt, egt) and np.array_equal(eq, eeq), (kind, form, order, blk)
                ctx.build_pairs(0)
                assert ctx.info()["transform_in_lds"] == 2
                assert np.array_equal(ctx.get_codes(0, G, 0, G), code), (kind, form, order)
                assert np.array_equal(ctx.tally(ref0), oracle.tally(code, ref0))
                res, it, tr = ctx.identify_degs(ref0, 1.0, 0.05, 8, 1)
                assert it == iters and tr == trace
                _check_result(res, exp)
    run = pkg.run_identify_degs(X, group, list(range(G)), 0.01, 1.0, 0.05, ref0, 8, 1, seed=seed, device=0)   # the drop-in call
    exp, iters, trace = oracle.identify_degs(X, gid, 2, 0.01, 1.0, 0.05, ref0, 8, 1, seed)
    assert run.iters_run == iters and run.trace == trace
    _check_result(run.result, exp)


@pytest.mark.parametrize("G", [9000, 21000, 30000, 61000, 70000])
def test_infinities_in_every_form_of_the_float64_ranking(pkg, oracle, G):
    """The same at the gene counts of every form of the bucket ranking (t_sample_wide <4, true> / <3, true> / <4, false> /
    <3, false>, t_sample_big): random and corner pair blocks against the oracle; three groups as well (the one-vs-rest counts)."""
    S, seed = 9, 0x5EED0063 + G
    X = pkg.synth.with_infinities(pkg.synth.float_expr(G, S, seed), seed, "column")
    X[G - 1, :] = np.inf; X[0, :] = -np.inf; X[1, :4] = -np.inf          # the first and the last gene: the ends of the code ranges
    rng = np.random.default_rng(G)
    inf_rows = np.flatnonzero(np.isinf(X[:, 0]))[:24]
    blocks = [(0, 24, 0, 48), (G - 24, G, G - 48, G), (G - 24, G, 0, 48), (0, 24, G - 48, G)]
    for _ in range(3):
        i0 = int(rng.integers(0, G - 24)); j0 = int(rng.integers(0, G - 48))
        blocks.append((i0, i0 + 24, j0, j0 + 48))
    for ng in (2, 3):
        gid = (np.arange(S) % ng).astype(np.int32)
        got, info = _counts_blocks(pkg, X, gid, ng, blocks)
        assert info["transform_in_lds"] == (2 if G <= 65535 else 3)
        for blk, (gt, eq) in zip(blocks, got):
            egt, eeq = oracle.pair_counts_as_evaluated(X, gid, ng, *blk)
            assert np.array_equal(gt, egt) and np.array_equal(eq, eeq), (G, ng, blk)
    assert len(inf_rows) == 24


def test_all_unstable_degenerate_se_zero(pkg, oracle):
    """Every pair unstable -> every N singular -> all delta1 = 0 -> trimmed std = 0 -> p = 0 (Normal(0,0))."""
    rng = np.random.default_rng(9)
    G, S = 60, 16
    X = rng.integers(0, 1000, size=(G, S))
    group = pkg.synth.groups(S)
    gid, lev = pkg.encode_groups(group)
    ref0 = np.ones(G, bool)
    run = pkg.run_identify_degs(X, group, list(range(G)), 1e-9, 1.0, 0.05, ref0, 3, 1, seed=1, device=0)
    exp, iters, trace = oracle.identify_degs(X.astype(np.float64), gid, 2, 1e-9, 1.0, 0.05, ref0, 3, 1, 1)
    assert run.iters_run == iters and run.trace == trace
    _check_result(run.result, exp)


def test_config2_5000x200_one_iteration(pkg, oracle):
    """BASELINE config 2: synthetic 5,000 x 200, 2 groups, 1 REO iteration, bit-exact counts/codes/tallies."""
    G, S, seed = 5000, 200, 0x5EED0002
    for family in ("t0", "t1"):
        X = (pkg.synth.t0_ranks if family == "t0" else pkg.synth.t1_counts)(G, S, seed)
        group = pkg.synth.groups(S)
        ref0 = pkg.synth.ref_mask(G, 3000, seed)
        ctx, gid, ng = _setup(pkg, X, group, seed)
        with ctx:
            Xf = X.astype(np.float64)
            assert ctx.get_thresholds()[:, 0].tolist() == [64, 64]
            for (i0, i1, j0, j1) in [(0, 64, 0, 256), (4900, 5000, 4700, 5000), (2500, 2532, 100, 400)]:
                gt, eq = ctx.pair_counts(i0, i1, j0, j1)
                egt, eeq = oracle.pair_counts(Xf, gid, ng, i0, i1, j0, j1)
                assert np.array_equal(gt, egt) and np.array_equal(eq, eeq)
            ctx.build_pairs(0)
            code = oracle.build_codes(Xf, gid, ng, 0, [64, 64], seed)
            got = ctx.get_codes(0, G, 0, G)
            assert np.array_equal(got, code)
            assert np.array_equal(ctx.tally(ref0), oracle.tally(code, ref0))
            res, iters, trace = ctx.identify_degs(ref0, 1.0, 0.05, 1, 5)
            exp, eit, etr = oracle.iterate(code, ref0, 1.0, 0.05, 1, 5)
            assert iters == eit == 1 and trace == etr
            _check_result(res, exp)


def test_full_size_properties_20000x1000(pkg):
    """BASELINE config 3 size: size-independent properties of the domain (mirror rule, tally identity, count symmetry, BH
    monotonicity) -- beside, not instead of, the whole-table comparison with the oracle in tests/test_gpu_fullsize.py."""
    G, S, seed = 20000, 1000, 0x5EED0003
    X = pkg.synth.t0_ranks(G, S, seed)
    group = pkg.synth.groups(S)
    ref0 = pkg.synth.ref_mask(G, 3000, seed)
    ctx, gid, ng = _setup(pkg, X, group, seed)
    with ctx:
        assert ctx.get_thresholds()[:, 0].tolist() == [280, 280]
        ctx.build_pairs(0)
        # mirror rule (:386) across distant blocks and on the diagonal
        for (a, b) in [(0, 19000), (5000, 5000), (12345, 300), (19744, 19744)]:
            n = 256
            c1 = ctx.get_codes(a, a + n, b, b + n).astype(np.int16)
            c2 = ctx.get_codes(b, b + n, a, a + n).astype(np.int16)
            off = (np.arange(a, a + n)[:, None] != np.arange(b, b + n)[None, :])
            assert np.array_equal(c1[off], (8 - c2.T)[off])
            assert ((c1 == 255) == ~off).all()
        # tallies of gene i sum to |ref| - [i in ref] (:403)
        cont = ctx.tally(ref0)
        assert np.array_equal(cont.sum(axis=1), ref0.sum() - ref0.astype(np.int64))
        assert cont.min() >= 0
        # counts of a sample of pairs: n_gt(i,j) + n_gt(j,i) = group size when there are no ties
        gt, eq = ctx.pair_counts(100, 164, 9000, 9300)
        gt2, eq2 = ctx.pair_counts(9000, 9300, 100, 164)
        assert not eq.any() and np.array_equal(gt + gt2.transpose(1, 0, 2), np.full(gt.shape, 500))
        # a direct numpy count on a few pairs
        for (i, j) in [(100, 9000), (163, 9299), (120, 9123)]:
            for g, sl in enumerate((slice(0, 500), slice(500, 1000))):
                assert gt[i - 100, j - 9000, g] == int((X[i, sl] > X[j, sl]).sum())
        res, iters, trace = ctx.identify_degs(ref0, 1.0, 0.05, 128, 5)
        assert 1 <= iters <= 128 and all(d + n == G for d, n in trace)
        assert np.isfinite(res).all() and (res[:, 0] >= 0).all() and (res[:, 1] <= 1).all()
        # BH monotonicity: padj is a non-decreasing function of pval
        o = np.argsort(res[:, 0], kind="stable")
        assert (np.diff(res[o, 1]) >= -1e-15).all() and (res[:, 1] >= res[:, 0] - 1e-15).all()


def _expected_block_codes(oracle, X, gid, thr, seed, i0, i1, j0, j1, ngroups=2, k=0):
    """Class codes of a block of ordered pairs from the oracle's counts and tie coins: comparison k =
    group k against every other sample (:374-377); thr is the 2 x ngroups threshold matrix (:362)."""
    gt, eq = oracle.pair_counts(X, gid, ngroups, i0, i1, j0, j1)
    sizes = np.bincount(gid, minlength=ngroups)
    S = int(sizes.sum())
    m1, m2 = int(thr[0, k]), int(thr[1, k])
    out = np.empty((i1 - i0, j1 - j0), dtype=np.uint8)
    for a in range(i1 - i0):
        for b in range(j1 - j0):
            i, j = i0 + a, j0 + b
            if i == j:
                out[a, b] = 255
                continue
            lo, hi = (i, j) if i < j else (j, i)  # the reference evaluates the pair with the smaller gene first (:366-392)
            g2 = gt[a, b].astype(np.int64) if i < j else sizes - gt[a, b] - eq[a, b]  # counts of the ordered pair (lo, hi)
            e2 = eq[a, b]
            nre = [int(g2[g]) + (oracle.tie_wins(seed, lo, hi, g, int(e2[g])) if e2[g] else 0) for g in range(ngroups)]
            nk, nt = nre[k], sum(nre) - nre[k]
            sk, st_ = int(sizes[k]), S - int(sizes[k])
            ic = 2 if nk >= m1 else (0 if sk - nk >= m1 else 1)
            it = 2 if nt >= m2 else (0 if st_ - nt >= m2 else 1)
            c = 3 * ic + it
            out[a, b] = c if i < j else 8 - c
    return out


@pytest.mark.parametrize("family", ["t0", "t1"])
def test_full_size_class_codes_on_sampled_blocks(pkg, oracle, family):
    """BASELINE config 3 size: the class table of the production pair kernel against codes derived from the
    oracle's counts and tie coins, on blocks far from the diagonal, on it, and in the padded last chunk."""
    G, S, seed = 20000, 1000, 0x5EED0003
    X = (pkg.synth.t0_ranks if family == "t0" else pkg.synth.t1_counts)(G, S, seed)
    group = pkg.synth.groups(S)
    ctx, gid, ng = _setup(pkg, X, group, seed)
    Xf = np.asfortranarray(X.astype(np.float64))
    with ctx:
        thr = ctx.get_thresholds()
        ctx.build_pairs(0)
        assert ctx.info()["has_ties"] == (0 if family == "t0" else 1)
        for (i0, j0, n) in [(0, 19960, 40), (10000, 10000, 48), (19952, 19952, 48), (777, 15000, 32), (15000, 777, 32), (19968, 31, 32)]:
            got = ctx.get_codes(i0, i0 + n, j0, j0 + n)
            exp = _expected_block_codes(oracle, Xf, gid, thr, seed, i0, i0 + n, j0, j0 + n)
            assert np.array_equal(got, exp), (family, i0, j0)


def test_full_size_one_vs_rest_codes_on_sampled_blocks(pkg, oracle):
    """Three groups at 20 000 genes: the shared per-group counts + classification kernels against codes derived
    from the oracle's counts and tie coins, every comparison, blocks far from / on the diagonal and in the padding."""
    G, S, seed, C = 20000, 150, 0x5EED0041, 3
    X = pkg.synth.t1_counts(G, S, seed)
    gid = (np.arange(S) % C).astype(np.int32)
    Xf = np.asfortranarray(X.astype(np.float64))
    with pkg.Context(device=0, seed=seed) as ctx:
        ctx.set_matrix(X); ctx.set_groups(gid, C); thr = ctx.compute_thresholds(0.05)
        for k in (1, 0, 2):
            ctx.build_pairs(k)
            assert ctx.info()["shared_group_counts"] == 1
            for (i0, j0, n) in [(0, 19968, 32), (10016, 10016, 40), (19960, 19960, 40), (15000, 777, 24)]:
                got = ctx.get_codes(i0, i0 + n, j0, j0 + n)
                exp = _expected_block_codes(oracle, Xf, gid, thr, seed, i0, i0 + n, j0, j0 + n, ngroups=C, k=k)
                assert np.array_equal(got, exp), (k, i0, j0)


@pytest.mark.parametrize("exchange,world,family", [("sum", 2, "t1"), ("gather", 2, "t1"), ("gather", 3, "t1"), ("gather", 8, "t1"), ("gather", 3, "t0")])
def test_shards_on_one_gpu_reproduce_the_unsharded_run(pkg, oracle, exchange, world, family):
    """G-sharding: `world` contexts (shards r / world) on one GPU, their exchange hooks joined by a thread barrier.
    Before the exchange owned pairs carry the oracle's codes and the others are empty; after the one exchange of the
    class table -- "sum": in-place sum of the whole table (reo_set_allreduce); "gather": all-gather of the shards' own
    forward words, mirror words derived on arrival (reo_set_allgather: what the in-library RCCL path does; three shards
    = uneven unit counts, eight = more shards than work units) -- every shard holds the whole table, and tallies / the whole iteration equal the unsharded
    result bit for bit."""
    import threading
    import torch
    G, S, seed = 2300, 24, 0x5EED0005
    ties = family == "t1"   # (tie-free data: work units four times as wide as the padded table)
    X = (pkg.synth.t1_counts if ties else pkg.synth.t0_ranks)(G, S, seed)
    group = pkg.synth.groups(S)
    gid, lev = pkg.encode_groups(group)
    ref0 = pkg.synth.ref_mask(G, 200, seed)
    exp, eit, etr = oracle.identify_degs(X.astype(np.float64), gid, 2, 0.01, 1.0, 0.05, ref0, 8, 1, seed)
    thr = [oracle.threshold(12), oracle.threshold(12)]
    code = oracle.build_codes(X.astype(np.float64), gid, 2, 0, thr, seed)
    barrier = threading.Barrier(world)
    slots = [None] * world
    dev = torch.device("cuda", 0)
    results, errors = [None] * world, []

    def run(rank):
        try:
            def hook(ptr, count, stream):
                torch.cuda.ExternalStream(stream, device=dev).synchronize()  # a host-side hook syncs the stream itself
                slots[rank] = torch.as_tensor(pkg.dist._RawDev(ptr, count), device=dev)
                barrier.wait()
                if rank == 0:
                    total = slots[0] + slots[1]
                    slots[0].copy_(total)
                    slots[1].copy_(total)
                    torch.cuda.synchronize()
                barrier.wait()

            def gather(send, recv, nbytes, stream):
                torch.cuda.ExternalStream(stream, device=dev).synchronize()
                slots[rank] = torch.as_tensor(pkg.dist._RawDevBytes(send, nbytes), device=dev)
                mine = torch.as_tensor(pkg.dist._RawDevBytes(recv, nbytes * world), device=dev)
                barrier.wait()
                for r in range(world):
                    mine[r * nbytes:(r + 1) * nbytes].copy_(slots[r])
                torch.cuda.synchronize()
                barrier.wait()

            with pkg.Context(device=0, seed=seed) as ctx:
                ctx.set_matrix(X)
                ctx.set_groups(gid, 2)
                ctx.compute_thresholds(0.01)
                ctx.set_shard(rank, world)
                ctx.build_pairs(0)          # no exchange configured yet: the shard's own part of the table
                info = ctx.info()
                got = ctx.get_codes(0, G, 0, G)
                mask = sharding_mirror.owned_pair_mask(G, info["sample_slots"], ties, rank, world)
                off = ~np.eye(G, dtype=bool)
                assert np.array_equal(got[mask], code[mask])
                assert (got[off & ~mask] == 4).all()
                with pytest.raises(pkg.ReoError):
                    ctx.tally(ref0)         # REO_ECOMM: the table has not been exchanged
                if exchange == "sum":
                    ctx.set_allreduce(hook)
                else:
                    ctx.set_allgather(gather)
                ctx.build_pairs(0)          # one exchange of the class table over the shards
                got = ctx.get_codes(0, G, 0, G)
                assert np.array_equal(got[off], code[off])
                cont = ctx.tally(ref0)
                res = ctx.identify_degs(ref0, 1.0, 0.05, 8, 1)
                results[rank] = (info, cont, res)
        except Exception:
            import traceback
            errors.append(traceback.format_exc())
            barrier.abort()

    threads = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=300)
    assert not errors, errors
    owned = [results[r][0]["tiles_owned"] for r in range(world)]
    assert sum(owned) == results[0][0]["tiles_total"] and (min(owned) > 0 or world == 8 or not ties)  # (five work units, three with tie-free data: some shards own none)
    for r in range(world):
        info, cont, (res, iters, trace) = results[r]
        assert np.array_equal(cont, oracle.tally(code, ref0))
        assert iters == eit and trace == etr
        _check_result(res, exp)


@pytest.mark.parametrize("world,waves,family", [(2, "4", "t0"), (3, "4", "t1"), (3, "8", "t0"), (2, "1", "t0")])
def test_pipelined_exchange_equals_the_unsharded_table(pkg, monkeypatch, world, waves, family):
    """Several shards with a gather exchange count their units in waves and exchange wave w (pack, all-gather, unpack on a
    second stream) while wave w + 1 is being counted (kernels.hip, launch_k1; REO_EXCHANGE_WAVES, 1 = the whole exchange
    behind the pair kernel).  9 000 genes: 21 work units, i.e. several per shard and wave.  Every shard's table, tallies
    and iteration must equal the unsharded context's, bit for bit; the hook is called once per wave."""
    import threading
    import torch
    monkeypatch.setenv("REO_EXCHANGE_WAVES", waves)
    G, S, seed = 9000, 32, 0x5EED0072
    X = (pkg.synth.t1_counts if family == "t1" else pkg.synth.t0_ranks)(G, S, seed)
    gid, lev = pkg.encode_groups(pkg.synth.groups(S))
    ref0 = pkg.synth.ref_mask(G, 900, seed)
    with pkg.Context(device=0, seed=seed) as ctx:
        ctx.set_matrix(X); ctx.set_groups(gid, 2); ctx.compute_thresholds(0.01); ctx.build_pairs(0)
        want_codes = ctx.get_codes(0, G, 0, G)
        want_cont = ctx.tally(ref0)
        want_res, want_it, want_tr = ctx.identify_degs(ref0, 1.0, 0.05, 6, 0)
    barrier = threading.Barrier(world)
    slots, calls = [None] * world, [0] * world
    dev = torch.device("cuda", 0)
    results, errors = [None] * world, []

    def run(rank):
        try:
            def gather(send, recv, nbytes, stream):
                torch.cuda.ExternalStream(stream, device=dev).synchronize()
                calls[rank] += 1
                slots[rank] = torch.as_tensor(pkg.dist._RawDevBytes(send, nbytes), device=dev)
                mine = torch.as_tensor(pkg.dist._RawDevBytes(recv, nbytes * world), device=dev)
                barrier.wait()
                for r in range(world):
                    mine[r * nbytes:(r + 1) * nbytes].copy_(slots[r])
                torch.cuda.synchronize()
                barrier.wait()

            with pkg.Context(device=0, seed=seed) as ctx:
                ctx.set_matrix(X); ctx.set_groups(gid, 2); ctx.compute_thresholds(0.01)
                ctx.set_shard(rank, world)
                ctx.set_allgather(gather)
                for rep in range(2):   # (the second build reuses the waves' item lists)
                    ctx.build_pairs(0)
                results[rank] = (ctx.get_codes(0, G, 0, G), ctx.tally(ref0), ctx.identify_degs(ref0, 1.0, 0.05, 6, 0), ctx.info())
        except Exception:
            import traceback
            errors.append(traceback.format_exc())
            barrier.abort()

    threads = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=300)
    assert not errors, errors
    total_units = 21
    per_shard = -(-total_units // world)
    assert calls == [2 * min(int(waves), per_shard)] * world, calls
    for r in range(world):
        codes, cont, (res, it, tr), info = results[r]
        assert np.array_equal(codes, want_codes) and np.array_equal(cont, want_cont)
        assert it == want_it and tr == want_tr and np.array_equal(res[:, 2:11], want_res[:, 2:11])
        assert np.allclose(res[:, :2], want_res[:, :2], rtol=0, atol=1e-12)


def test_three_group_golden_and_synthetic_one_vs_rest(pkg, oracle, golden):
    """More than two groups: one comparison per group against every other sample, 16 columns each."""
    g = golden("three_groups48.json")
    X = np.array(g["X"], dtype=np.float64)
    with pkg.Context(device=0, seed=g["comparisons"][0]["seed"]) as ctx:
        ctx.set_matrix(X)
        ctx.set_groups(g["gid"], 3)
        ctx.compute_thresholds(g["comparisons"][0]["pval_reo"])
        gt, eq = ctx.pair_counts(0, 48, 0, 48)
        assert np.array_equal(gt, g["n_gt"]) and np.array_equal(eq, g["n_eq"])
        for cm in g["comparisons"]:
            assert ctx.get_thresholds()[:, cm["k"]].tolist() == cm["thr"]
            ctx.build_pairs(cm["k"])
            assert np.array_equal(ctx.get_codes(0, 48, 0, 48), cm["code"])
            assert np.array_equal(ctx.tally(np.array(g["ref0"])), cm["cont"])
            res, iters, trace = ctx.identify_degs(np.array(g["ref0"]), cm["pval_deg"], cm["padj_deg"], cm["n_iter"], cm["n_conv"])
            assert iters == cm["iters_run"] and [list(t) for t in trace] == cm["trace"]
            _check_result(res, np.array(cm["result"]))
    # synthetic: three groups of unequal size, tie-rich / tie-free / float
    for family, gen in (("t1", pkg.synth.t1_counts), ("t0", pkg.synth.t0_ranks), ("float", pkg.synth.float_expr)):
        G, S, seed = 520, 47, 0x5EED0006
        X = gen(G, S, seed)
        group = np.array(["a"] * 14 + ["b"] * 17 + ["c"] * 16, dtype=object)
        group = group[np.random.default_rng(3).permutation(S)]  # interleaved samples
        gid, lev = pkg.encode_groups(group)
        ref0 = pkg.synth.ref_mask(G, 120, seed)
        run = pkg.run_identify_degs(X, group, list(range(G)), 0.01, 1.0, 0.05, ref0, 8, 1, seed=seed, device=0)
        assert run.res.shape == (G, 1 + 16 * 3) and len(run.comparisons) == 3
        for cm in run.comparisons:
            exp, iters, trace = oracle.identify_degs(X.astype(np.float64), gid, 3, 0.01, 1.0, 0.05, ref0, 8, 1, seed, k=cm["k"])
            assert cm["iters_run"] == iters and cm["trace"] == trace, (family, cm["k"])
            _check_result(cm["result"], exp)
            assert np.array_equal(run.res[:, 1 + 16 * cm["k"]: 16 + 16 * cm["k"]].astype(float), cm["result"])


def test_more_than_64_groups(pkg, oracle):
    """One-vs-rest with 70 groups of two or three samples each (the reference takes any number of levels, :353-362; the
    threshold falls back to the group size with its warning, :87-90): class tables of a few comparisons against the oracle."""
    G, S, seed, C = 700, 170, 0x5EED0064, 70
    X = pkg.synth.t1_counts(G, S, seed)
    gid = (np.arange(S) % C).astype(np.int32)
    with pkg.Context(device=0, seed=seed) as ctx:
        ctx.set_matrix(X); ctx.set_groups(gid, C); thr = ctx.compute_thresholds(0.01)
        assert thr.shape == (2, C)
        for k in (0, 37, 69):
            ctx.build_pairs(k)
            code = oracle.build_codes(X.astype(np.float64), gid, C, k, [int(thr[0, k]), int(thr[1, k])], seed)
            assert np.array_equal(ctx.get_codes(0, G, 0, G), code), k


@pytest.mark.parametrize("family", ["t0", "t1"])
def test_shared_group_counts_equal_recounting(pkg, oracle, family, monkeypatch):
    """> 2 groups: the comparisons classify from per-group counts kept in HBM (counted once); the class
    codes and tallies must equal both the per-comparison recount and the oracle, also on two shards."""
    G, S, seed, C = 1100, 83, 0x5EED0011, 5
    X = (pkg.synth.t0_ranks if family == "t0" else pkg.synth.t1_counts)(G, S, seed)
    gid = np.random.default_rng(5).integers(0, C, S).astype(np.int32)
    gid[:C] = np.arange(C)  # every group present
    ref0 = pkg.synth.ref_mask(G, 300, seed)
    out = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("REO_SHARE_GROUP_COUNTS", mode)
        with pkg.Context(device=0, seed=seed) as ctx:
            ctx.set_matrix(X); ctx.set_groups(gid, C); thr = ctx.compute_thresholds(0.05)
            for k in (3, 0, 4, 1, 2):  # any order, the counts are cached after the first
                ctx.build_pairs(k)
                info = ctx.info()
                assert info["shared_group_counts"] == int(mode)
                assert (info["group_count_bytes"] > 0) == (mode == "1")
                out[mode, k] = (ctx.get_codes(0, G, 0, G), ctx.tally(ref0))
    for k in range(C):
        assert np.array_equal(out["1", k][0], out["0", k][0]) and np.array_equal(out["1", k][1], out["0", k][1])
        code = oracle.build_codes(X.astype(np.float64), gid, C, k, [int(thr[0, k]), int(thr[1, k])], seed)
        assert np.array_equal(out["1", k][0], code)
    # two shards: their tallies add up to the unsharded ones
    monkeypatch.setenv("REO_SHARE_GROUP_COUNTS", "1")
    for k in (2, 0):
        tot = 0
        for rank in range(2):
            with pkg.Context(device=0, seed=seed) as ctx:
                ctx.set_matrix(X); ctx.set_groups(gid, C); ctx.compute_thresholds(0.05); ctx.set_shard(rank, 2)
                ctx.build_pairs(1); ctx.build_pairs(k)
                tot = tot + _shard_raw(pkg, ctx, ref0, G)
        assert np.array_equal(sharding_mirror.derive_tallies(tot, ref0), out["1", k][1])


def _shard_raw(pkg, ctx, ref0, G):
    """raw counters of one shard's partial class table (codes it does not own decode as 'unstable/unstable')."""
    code = ctx.get_codes(0, G, 0, G)
    return sharding_mirror.raw_counters(code, ref0)


def _eager_ctx(pkg, X, group, seed, pval_reo=0.01):
    """groups and thresholds BEFORE the matrix: reo_set_matrix_* pipelines the upload with the ranking and the pair kernel's sides"""
    gid, lev = pkg.encode_groups(group)
    ctx = pkg.Context(device=0, seed=seed)
    ctx.set_groups(gid, len(lev))
    ctx.compute_thresholds(pval_reo)
    ctx.set_matrix(X)
    return ctx


@pytest.mark.parametrize("kind", ["ranks", "counts", "float", "big_int", "three_groups", "interleaved", "unequal", "view_ld", "small",
                                  "negative", "huge_int", "growing", "float_counts", "float_from_f32", "float_mixed",
                                  "ties_second_group", "flag_second_group", "ties_late_in_first_group"])
def test_pipelined_upload_equals_matrix_first(pkg, kind, monkeypatch):
    """reo_set_matrix_i64 / _f64 from host memory with the groups already set (round 5): the columns travel in chunks, samples are
    ranked as they arrive, a group's blocks are sliced when its last sample is in, and (two groups, one GPU, thresholds set) the
    pair kernel's side of that group starts while the other group is still on its way.  Everything must be BIT-EQUAL to the
    matrix-first order of rounds 1-4: whole class table, tallies, trace, every statistic -- for contiguous and interleaved labels,
    unequal groups (padding slots), three groups (ranking only), a view with a leading dimension, data that the first choice of
    ranking kernel flags (falls back to the resident copy), every form a chunk can take on the link (int16 / int32 / float32 / the
    caller's array, for Int64 and for Float64 input, changing on the way), and with the pipelining switched down (REO_EAGER_UPLOAD=1, 0)."""
    seed = 0x5EED0051
    rng = np.random.default_rng(11)
    G, S = 9000, 300
    group = pkg.synth.groups(S)
    if kind == "ranks": X = pkg.synth.t0_ranks(G, S, seed)
    elif kind == "counts": X = pkg.synth.t1_counts(G, S, seed)
    elif kind == "float": X = pkg.synth.float_expr(G, S, seed)
    elif kind == "big_int": X = rng.integers(0, 2 ** 31, size=(G, S))            # more varying key bits than the histogram form takes
    elif kind == "three_groups": X = pkg.synth.t1_counts(G, S, seed); group = ["a"] * 90 + ["b"] * 110 + ["c"] * 100
    elif kind == "interleaved": X = pkg.synth.t1_counts(G, S, seed); group = [("u", "v")[int(b)] for b in rng.integers(0, 2, S)]
    elif kind == "unequal": X = pkg.synth.t0_ranks(G, S, seed); group = ["u"] * 37 + ["v"] * (S - 37)
    elif kind == "view_ld":
        big = np.asfortranarray(np.full((G + 24, S), -7, dtype=np.int64)); big[8:8 + G, :] = pkg.synth.t1_counts(G, S, seed); X = big[8:8 + G, :]
    elif kind == "negative": X = rng.integers(-30000, 30000, size=(G, S)); X[5, 7] = -32768; X[6, 299] = 32767     # the 16-bit limits exactly
    elif kind == "huge_int": X = rng.integers(-2 ** 40, 2 ** 40, size=(G, S))                                      # no narrow width fits: the caller's array itself
    elif kind == "growing":                                                                                        # widths grow on the way: 16 bits, then 32, then 64
        X = rng.integers(0, 30000, size=(G, S)); X[:, 130:] += 40000; X[17, 135] = 2 ** 31 - 1; X[G - 1, 280] = -2 ** 31 - 1
    elif kind == "float_counts": X = rng.integers(0, 3000, size=(G, S)).astype(np.float64); X[3, 3] = -5.0                  # integers in a Float64 matrix: 16-bit on the link
    elif kind == "float_from_f32": X = np.round(rng.lognormal(1, 1, size=(G, S)), 2).astype(np.float32).astype(np.float64)   # single-precision data: float32 on the link
    elif kind == "float_mixed":                                                                                              # the form climbs: int16, int32, float32, the array itself
        X = rng.integers(0, 900, size=(G, S)).astype(np.float64); X[:, 100:] += 70000.0; X[:, 180:] += np.float32(0.25); X[11, 190] = -0.0
        X[:, 250:] = rng.normal(8, 2, size=(G, 50))
    # what the device-side gate of the side launches (k1w_pairs_gated) has to get right: the transform's flags CHANGE while sides and
    # ranges of the pair kernel are already queued or running
    elif kind == "ties_second_group":         # side 0 is launched tie-free; ties appear only in group 2's columns
        X = pkg.synth.t0_ranks(G, S, seed); X[:, S // 2:] = pkg.synth.t1_counts(G, S, seed)[:, S // 2:]
    elif kind == "flag_second_group":         # a sample of group 2 needs another form of the ranking: everything is done again at reo_build_pairs
        X = pkg.synth.t0_ranks(G, S, seed); X[:, S - 20:] = rng.integers(0, 2 ** 31, size=(G, 20))
    elif kind == "ties_late_in_first_group":  # the first RANGE of side 0 runs tie-free and parks n_gt only; the tie form of its next range takes n_ge = n_gt for those blocks
        X = pkg.synth.t0_ranks(G, S, seed); X[:, 100:150] = pkg.synth.t1_counts(G, S, seed)[:, 100:150]
    else: G, S = 700, 41; X = pkg.synth.t1_counts(G, S, seed); group = pkg.synth.groups(S)
    ref0 = pkg.synth.ref_mask(G, G // 5, seed)
    ng = len(set(group))

    def outputs(ctx):
        with ctx:
            out = []
            for k in range(1 if ng == 2 else ng):
                ctx.build_pairs(k)
                codes = ctx.get_codes(0, G, 0, G)
                out.append((codes, ctx.tally(ref0), ctx.identify_degs(ref0, 1.0, 0.05, 6, 0), ctx.info()["has_ties"]))
            return out

    want = outputs(_setup(pkg, X, group, seed)[0])
    # (REO_UPLOAD_THREADS: Int64 chunks cross the link as 16- or 32-bit numbers when they fit, converted by that many host threads and
    #  widened on the device; 0 = the caller's array as it is.  REO_EAGER_CHUNK=37: many small chunks, the staging ring goes round.)
    # (REO_EAGER_RANGES, round 6: the pair kernel's items of a side run over RANGES of its sample blocks as the chunks arrive, counts
    #  parked in between, the last range classifies -- 1 = whole sides as in round 5, n = n ranges per side where the blocks allow.)
    for mode, threads, chunk, ranges in (("2", None, None, None), ("1", None, None, None), ("0", None, None, None), ("2", "0", None, "1"), ("2", "3", "37", "1"),
                                         ("2", None, "37", "4"), ("2", "3", "64", "2"), ("2", None, "32", "6")):
        monkeypatch.setenv("REO_EAGER_UPLOAD", mode)
        for name, val in (("REO_UPLOAD_THREADS", threads), ("REO_EAGER_CHUNK", chunk), ("REO_EAGER_RANGES", ranges)):
            if val is None: monkeypatch.delenv(name, raising=False)
            else: monkeypatch.setenv(name, val)
        ctxe = _eager_ctx(pkg, X, group, seed)
        link = ctxe.info()["upload_link_bytes"]
        nr = ctxe.info()["eager_range_launches"]
        if ranges in (None, "1") or ng != 2 or mode != "2" or kind == "small": assert nr == 0, (kind, mode, ranges, nr)    # (default: 5 blocks per side are too few to cut)
        elif kind not in ("big_int", "flag_second_group"): assert nr >= (2 if kind == "unequal" else 4), (kind, ranges, nr)                           # both sides in at least two ranges (37 samples: one block)
        if mode != "0" and threads != "0" and kind in ("ranks", "negative", "big_int", "float_counts", "float_from_f32"):
            assert link == X.size * (2 if kind in ("ranks", "negative", "float_counts") else 4), (kind, link)     # what the link carried
        elif mode != "0" and threads != "0" and kind in ("counts", "growing", "float_mixed"):
            assert X.size * 2 < link < X.size * 8, (kind, link)                                   # chunk by chunk: the narrowest form that fits
        elif mode != "0" and (kind in ("float", "huge_int") or threads == "0"):
            assert link == X.size * 8, (kind, link)
        elif mode != "0":
            assert X.size * 2 <= link <= X.size * 8, (kind, link)
        got = outputs(ctxe)
        for (c0, t0, (r0, i0, tr0), h0), (c1, t1, (r1, i1, tr1), h1) in zip(want, got):
            assert np.array_equal(c0, c1), (kind, mode, "class table")
            assert np.array_equal(t0, t1) and i0 == i1 and tr0 == tr1 and h0 == h1, (kind, mode)
            assert np.array_equal(r0, r1, equal_nan=True), (kind, mode, "statistics")
    for name in ("REO_EAGER_UPLOAD", "REO_UPLOAD_THREADS", "REO_EAGER_CHUNK", "REO_EAGER_RANGES"):
        monkeypatch.delenv(name, raising=False)
    # rebuilds on the same context, a second matrix, new thresholds
    if ng == 2:
        with _eager_ctx(pkg, X, group, seed) as ctx:
            ctx.build_pairs(0)
            r = ctx.identify_degs(ref0, 1.0, 0.05, 6, 0)
            assert np.array_equal(r[0], want[0][2][0], equal_nan=True) and r[2] == want[0][2][2]
            ctx.build_pairs(0); ctx.build_pairs(0)                                # (the second and third really rebuild)
            assert np.array_equal(ctx.get_codes(0, G, 0, G), want[0][0])
            ctx.set_matrix(X)                                                     # a second matrix on the same context, pipelined again
            ctx.build_pairs(0)
            assert np.array_equal(ctx.get_codes(0, G, 0, G), want[0][0])
            ctx.compute_thresholds(0.2)                                           # new thresholds drop the table that reo_set_matrix made
            with pytest.raises(pkg.DimensionMismatch):
                ctx.get_codes(0, 4, 0, 4)
    # a NaN is reported by the call that reads the matrix (infinities are accepted: test_infinities_*)
    if kind == "float":
        Xn = X.copy(); Xn[G // 2, S - 3] = np.nan
        gid, lev = pkg.encode_groups(group)
        with pkg.Context(device=0, seed=seed) as ctx:
            ctx.set_groups(gid, len(lev)); ctx.compute_thresholds(0.01)
            with pytest.raises(pkg.DimensionMismatch, match="contains NaN"):
                ctx.set_matrix(Xn)
            ctx.set_matrix(X); ctx.build_pairs(0)                                 # the context works again
            assert np.array_equal(ctx.get_codes(0, 64, 0, G), want[0][0][:64])


def test_host_matrix_view_with_leading_dimension(pkg, oracle):
    """A column-major host view with ld > G (rows of a taller matrix, like a Julia view) must arrive
    intact: counts of blocks at both ends and in the middle of a 20 000-gene matrix."""
    G, S, seed = 20000, 260, 0x5EED0012
    big = np.asfortranarray(np.zeros((G + 24, S), dtype=np.int64))
    big[:] = -7  # rows outside the view must never be read as data
    Xc = pkg.synth.t1_counts(G, S, seed)
    big[8:8 + G, :] = Xc
    view = big[8:8 + G, :]
    assert view.strides == (8, 8 * (G + 24))
    gid, _ = pkg.encode_groups(pkg.synth.groups(S))
    with pkg.Context(device=0, seed=seed) as ctx:
        ctx.set_matrix(view)
        ctx.set_groups(gid, 2)
        for (i0, i1, j0, j1) in ((0, 40, 0, 40), (G - 40, G, G - 40, G), (9990, 10030, 40, 80)):
            gt, eq = ctx.pair_counts(i0, i1, j0, j1)
            egt, eeq = oracle.pair_counts(Xc.astype(np.float64), gid, 2, i0, i1, j0, j1)
            assert np.array_equal(gt, egt) and np.array_equal(eq, eeq)
    for dt in (np.float64,):  # the same through the Float64 entry point, contiguous
        Xf = np.asfortranarray(Xc.astype(dt))
        with pkg.Context(device=0, seed=seed) as ctx:
            ctx.set_matrix(Xf)
            ctx.set_groups(gid, 2)
            gt, eq = ctx.pair_counts(100, 140, 19900, 19940)
            egt, eeq = oracle.pair_counts(Xf, gid, 2, 100, 140, 19900, 19940)
            assert np.array_equal(gt, egt) and np.array_equal(eq, eeq)


def _counts_blocks(pkg, X, gid, ngroups, blocks, seed=3):
    """Counts of pair blocks as the library ranks the data.  With REO_TRANSFORM=segmented in the environment the caller asks for the
    SECOND implementation of the transform (rocprim's segmented sort, rounds 1-4).  Since round 5 the shipped library carries no
    rocprim (`make ROCPRIM=1` builds the A/B variant): then the second opinion is the oracle's literal comparator itself."""
    try:
        with pkg.Context(device=0, seed=seed) as ctx:
            ctx.set_matrix(X)
            ctx.set_groups(gid, ngroups)
            out = [ctx.pair_counts(*b) for b in blocks]
            return out, ctx.info()
    except pkg.DimensionMismatch as e:
        if os.environ.get("REO_TRANSFORM") != "segmented" or "no segmented sort" not in str(e):
            raise
        import __graft_entry__ as ge
        orc = ge.load_oracle()
        Xf = np.asarray(X, dtype=np.float64)
        return [orc.pair_counts(Xf, np.asarray(gid, dtype=np.int32), ngroups, *b) for b in blocks], {"transform_in_lds": 0, "has_ties": None}


def _varying_key_bits(X):
    """the widest window of key bits that varies inside one sample (what t_sample sorts by): bit length of the OR of
    x ^ x[0] over the sample, minus its trailing zeros"""
    worst = 0
    for s in range(X.shape[1]):
        col = X[:, s].astype(np.int64)
        d = int(np.bitwise_or.reduce(col ^ col[0]))
        if d:
            worst = max(worst, d.bit_length() - ((d & -d).bit_length() - 1))
    return worst


@pytest.mark.parametrize("G,family", [(32769, "t0"), (50001, "t0"), (65535, "t0"), (40000, "small"), (65535, "t1"), (45000, "tail"), (65535, "tail")])
def test_histogram_ranking_up_to_65535_genes(pkg, oracle, G, family, monkeypatch):
    """Above 32 768 genes the per-sample ranking in LDS is the histogram form with 16-bit bins (Int64 keys of at most 16
    varying bits: ranks of up to 65 535 genes, small counts) or the compressed histogram with its low-bit rows in L2 (17-24
    bits: counts); everything else still goes through the segmented sort.  All must give the oracle's counts -- at the first
    gene count above the 32-bit-bin form, in the middle, and at the u16 limit."""
    S, seed = 12, 0x5EED0017
    if family == "t0":
        X = pkg.synth.t0_ranks(G, S, seed)
    elif family == "t1":
        X = pkg.synth.t1_counts(G, S, seed)
    elif family == "tail":   # counts with a long tail (up to 22 bits): dense and repeated at small values, a few genes per octave far out
        rng = np.random.default_rng(seed)
        X = np.floor(np.exp(rng.normal(3.0, 2.6, size=(G, S)))).astype(np.int64)
        X = np.minimum(X, (1 << 22) - 1)
        X[rng.random((G, S)) < 0.3] = 0
    else:
        X = np.random.default_rng(seed).integers(0, 9, size=(G, S))   # nine values: bands of thousands of genes
    gid = np.array([0, 1, 0, 1, 1, 0, 0, 1, 0, 1, 1, 0], dtype=np.int32)
    blocks = [(0, 40, 0, 40), (G - 40, G, G - 40, G), (G // 2, G // 2 + 24, 8, 40), (min(32760, G - 24), min(32760, G - 24) + 24, G - 64, G)]
    monkeypatch.delenv("REO_TRANSFORM", raising=False)
    a, info_a = _counts_blocks(pkg, X, gid, 2, blocks)
    # ranks and small counts: the 16-bit-bin histogram; T1 counts (17-24 varying bits): the compressed histogram, unless a sample
    # has a crowded lossy bucket (then every sample takes the bucket form)
    assert _varying_key_bits(X) <= (16 if family not in ("t1", "tail") else 24), _varying_key_bits(X)
    assert 17 <= _varying_key_bits(X) or family != "tail"
    assert info_a["transform_in_lds"] == 1 or (family == "t1" and info_a["transform_in_lds"] == 2)
    print("transform_in_lds", info_a["transform_in_lds"], "varying key bits", _varying_key_bits(X))
    monkeypatch.setenv("REO_TRANSFORM", "segmented")
    b, info_b = _counts_blocks(pkg, X, gid, 2, blocks)
    assert info_b["transform_in_lds"] == 0
    Xf = X.astype(np.float64)
    for blk, (ga, ea), (gb, eb) in zip(blocks, a, b):
        egt, eeq = oracle.pair_counts(Xf, gid, 2, *blk)
        assert np.array_equal(ga, gb) and np.array_equal(ea, eb)
        assert np.array_equal(ga, egt) and np.array_equal(ea, eeq)


@pytest.mark.parametrize("G", [8192, 8193, 20480, 20481, 24576, 24577, 32768, 32769])
def test_transform_in_lds_and_segmented_agree_at_the_size_limits(pkg, oracle, G, monkeypatch):
    """The per-sample ranking in LDS (<= 32 768 genes) and the device-wide segmented sort must give the same counts,
    and the oracle's, on both sides of every items-per-thread limit -- for the histogram forms (Int64 counts) and for
    the bucket form (the same data as Float64, REO_TRANSFORM=wide for the integers)."""
    S, seed = 11, 0x5EED0013
    X = pkg.synth.t1_counts(G, S, seed)
    gid = np.array([0, 1, 0, 1, 1, 0, 0, 1, 0, 1, 1], dtype=np.int32)
    blocks = [(0, 40, 0, 40), (G - 40, G, G - 40, G), (G // 2, G // 2 + 24, 8, 40)]
    monkeypatch.delenv("REO_TRANSFORM", raising=False)
    a, info_a = _counts_blocks(pkg, X, gid, 2, blocks)
    assert info_a["transform_in_lds"] in (1, 2)   # (2: a crowded lossy bucket sent every sample to the bucket form)
    monkeypatch.setenv("REO_TRANSFORM", "segmented")
    b, info_b = _counts_blocks(pkg, X, gid, 2, blocks)
    assert info_b["transform_in_lds"] == 0
    Xf = X.astype(np.float64)
    monkeypatch.setenv("REO_TRANSFORM", "wide")
    w, info_w = _counts_blocks(pkg, X, gid, 2, blocks)
    assert info_w["transform_in_lds"] == 2
    monkeypatch.delenv("REO_TRANSFORM", raising=False)
    f, info_f = _counts_blocks(pkg, np.log2(1.0 + Xf), gid, 2, blocks)   # a monotone map: ties stay equalities or widen into 0.1 bands
    assert info_f["transform_in_lds"] == 2
    for blk, (ga, ea), (gb, eb), (gw, ew), (gf, ef) in zip(blocks, a, b, w, f):
        egt, eeq = oracle.pair_counts(Xf, gid, 2, *blk)
        assert np.array_equal(ga, gb) and np.array_equal(ea, eb)
        assert np.array_equal(ga, egt) and np.array_equal(ea, eeq)
        assert np.array_equal(gw, egt) and np.array_equal(ew, eeq)
        fgt, feq = oracle.pair_counts(np.log2(1.0 + Xf), gid, 2, *blk)
        assert np.array_equal(gf, fgt) and np.array_equal(ef, feq)


def test_transform_key_width_decides_the_path(pkg, oracle, monkeypatch):
    """Int64 keys of at most 24 varying bits rank by histogram (transform_in_lds 1); wider integers, negative values and
    Float64 by buckets (2); all of them match the oracle."""
    monkeypatch.delenv("REO_TRANSFORM", raising=False)
    rng = np.random.default_rng(99)
    G, S = 700, 12
    gid = np.array([0, 1] * 6, dtype=np.int32)
    cases = {
        "24 bits": (rng.integers(0, 2 ** 24, (G, S), dtype=np.int64), 1),
        "31 bits": (rng.integers(0, 2 ** 31, (G, S), dtype=np.int64), 2),
        "33 bits": (rng.integers(0, 2 ** 33, (G, S), dtype=np.int64), 2),
        "63 bits": (rng.integers(-2 ** 62, 2 ** 62, (G, S), dtype=np.int64), 2),
        "negatives": (rng.integers(-50, 50, (G, S), dtype=np.int64), 2),
        "offset": (rng.integers(0, 1000, (G, S), dtype=np.int64) + (1 << 40), 1),   # high bits constant: still narrow
        "floats": (rng.lognormal(2.0, 1.5, (G, S)), 2),
        "float ranks": (rng.integers(0, 64, (G, S)).astype(np.float64) * 0.25, 2),
    }
    for name, (X, want) in cases.items():
        X[5] = X[6]  # some exact ties
        (out,), info = _counts_blocks(pkg, X, gid, 2, [(0, G, 0, G)])
        if want is not None:
            assert info["transform_in_lds"] == want, name
        if name == "63 bits":   # (the oracle takes Float64: integers this wide are not exactly representable -- compare with numpy)
            gi = np.arange(0, G, 7)
            egt = np.stack([(X[gi][:, None, gid == g] > X[None, :, gid == g]).sum(axis=2) for g in (0, 1)], axis=2)
            eeq = np.stack([(X[gi][:, None, gid == g] == X[None, :, gid == g]).sum(axis=2) for g in (0, 1)], axis=2)
            assert np.array_equal(out[0][gi], egt) and np.array_equal(out[1][gi], eeq), name
            continue
        egt, eeq = oracle.pair_counts(np.asarray(X, dtype=np.float64), gid, 2, 0, G, 0, G)
        assert np.array_equal(out[0], egt) and np.array_equal(out[1], eeq), name


@pytest.mark.parametrize("G", [700, 9000, 20000, 32768, 32769, 47000, 50000, 65535])
def test_transform_bucket_ranking_of_float64(pkg, oracle, G, monkeypatch):
    """t_sample_wide on Float64 (transform.hip): the 0.1 band of is_greater (:72) found in code space with the
    reference's own predicate.  Samples built to hit its corners: many exact zeros (an equality bucket of one value),
    values on both sides of x - y = 0.1 to the last ulp, x slightly above 0.1 (x - 0.1 cancels to something tiny),
    negative values and signed zeros, denormals, a sample of one value, huge dynamic range, and ordinary
    log-expression.  Against the oracle's literal comparator and against the segmented path."""
    rng = np.random.default_rng(1000 + G)
    S = 12
    gid = np.array([0, 1] * 6, dtype=np.int32)
    X = np.log2(1.0 + pkg.synth.t1_counts(G, S, 0x5EED0067).astype(np.float64)) + rng.uniform(0, 0.05, (G, S))
    X[:, 1] = np.where(rng.random(G) < 0.3, 0.0, rng.lognormal(1.0, 1.0, G))              # 30 % exact zeros
    base = rng.uniform(0.5, 8.0, G)
    X[:, 2] = np.where(rng.random(G) < 0.5, base, np.nextafter(base + 0.1, np.where(rng.random(G) < 0.5, 0.0, 100.0)))
    h = G // 4
    X[: 2 * h, 2] = np.concatenate([base[:h], base[:h] + 0.1])                           # x and x + 0.1 both present: the band edge itself
    X[:, 3] = 0.1 + rng.uniform(0, 1e-9, G) * (rng.random(G) < 0.5)                      # x - 0.1 cancels; plus exact 0.1
    X[: G // 3, 3] = rng.uniform(0, 3e-10, G // 3)
    X[:, 4] = rng.normal(0, 0.2, G); X[:5, 4] = [0.0, -0.0, 0.05, -0.05, 0.1]            # around zero, both signs
    X[:, 5] = 5e-324 * rng.integers(0, 1000, G)                                          # denormals and zeros
    X[:, 6] = 3.25                                                                        # one value
    X[:, 7] = 10.0 ** rng.uniform(-300, 300, G) * rng.choice([-1.0, 1.0], G)              # 600 decades, both signs
    X[:, 8] = np.round(rng.lognormal(0, 2, G), 1)                                         # a 0.1 grid: band edges meet grid values
    blocks = [(0, 48, 0, 48), (G - 40, G, G - 40, G), (100, 124, G // 2, G // 2 + 64), (G // 4, G // 4 + 16, 0, 64)]
    monkeypatch.delenv("REO_TRANSFORM", raising=False)
    a, info_a = _counts_blocks(pkg, X, gid, 2, blocks)
    assert info_a["transform_in_lds"] == 2
    monkeypatch.setenv("REO_TRANSFORM", "segmented")
    b, info_b = _counts_blocks(pkg, X, gid, 2, blocks)
    assert info_b["transform_in_lds"] == 0
    for blk, (ga, ea), (gb, eb) in zip(blocks, a, b):
        egt, eeq = oracle.pair_counts(X, gid, 2, *blk)
        assert np.array_equal(ga, egt) and np.array_equal(ea, eeq), blk
        assert np.array_equal(gb, egt) and np.array_equal(eb, eeq), blk
    # values a few hundred ulps apart next to one far outlier, and a column sorted by gene index: the splitters follow the data
    Y = X.copy()
    Y[:, 1] = 1.0 + rng.uniform(0, 1e-13, G)
    Y[0, 1] = 1e6
    Y[:, 2] = np.sort(X[:, 0])
    monkeypatch.delenv("REO_TRANSFORM", raising=False)
    (c,), info_c = _counts_blocks(pkg, Y, gid, 2, [blocks[0]])
    assert info_c["transform_in_lds"] == 2
    egt, eeq = oracle.pair_counts(Y, gid, 2, *blocks[0])
    assert np.array_equal(c[0], egt) and np.array_equal(c[1], eeq)


@pytest.mark.parametrize("G", [3000, 20000, 32768])
def test_transform_compressed_histogram_ranking(pkg, oracle, G, monkeypatch):
    """Integer keys with 16 to 24 varying bits rank by a histogram of a monotone compression of the key (transform.hip,
    t_sample): exact codes below 2^13, octave + mantissa above, ranks inside a lossy bucket by scanning its members.
    Values on octave boundaries, equal values inside lossy buckets, moderately crowded buckets (scanned) and crowded
    ones (every sample then takes the bucket form, t_sample_wide) -- all against the oracle's counts and against the segmented path."""
    rng = np.random.default_rng(G)
    S = 12
    gid = np.array([0, 1] * 6, dtype=np.int32)
    X = pkg.synth.t1_counts(G, S, 0x5EED0066).copy()
    # sample 1: a long tail to 2^23 with values exactly on octave boundaries and repeated values in the tail
    X[:, 1] = (rng.lognormal(6.0, 3.0, G)).astype(np.int64) % (1 << 23)
    X[: G // 50, 1] = 1 << rng.integers(13, 23, G // 50)
    X[G // 50: G // 25, 1] = (1 << 20) + rng.integers(0, 40, G // 25 - G // 50)       # one moderately crowded region, many ties
    # sample 2: 200 genes inside one lossy bucket (more than the scan limit): the radix sort takes that sample
    X[:, 2] = rng.integers(0, 1 << 18, G)
    X[:200, 2] = (1 << 17) + rng.integers(0, 16, 200)
    # sample 3: 20 bits, uniformly spread (every bucket nearly empty)
    X[:, 3] = rng.integers(0, 1 << 20, G)
    blocks = [(0, 48, 0, 48), (0, 40, G // 50, G // 50 + 64), (G - 40, G, G - 40, G), (100, 124, 150, 214)]
    monkeypatch.delenv("REO_TRANSFORM", raising=False)
    a, info_a = _counts_blocks(pkg, X, gid, 2, blocks)
    assert info_a["transform_in_lds"] == 2      # sample 2's crowded bucket sends every sample to the bucket form (t_sample_wide)
    X1 = X.copy(); X1[:, 2] = X[::-1, 3]        # without it: the compressed histogram ranks all samples
    a1, info_a1 = _counts_blocks(pkg, X1, gid, 2, blocks)
    assert info_a1["transform_in_lds"] == (1 if G == 3000 else 2)   # (at the larger sizes sample 1's crowded region passes the scan limit too)
    monkeypatch.setenv("REO_TRANSFORM", "segmented")
    b, info_b = _counts_blocks(pkg, X, gid, 2, blocks)
    Xf = X.astype(np.float64)
    for blk, (ga, ea), (gb, eb), (g1, e1) in zip(blocks, a, b, a1):
        egt, eeq = oracle.pair_counts(Xf, gid, 2, *blk)
        assert np.array_equal(ga, egt) and np.array_equal(ea, eeq), blk
        assert np.array_equal(ga, gb) and np.array_equal(ea, eb), blk
        egt1, eeq1 = oracle.pair_counts(X1.astype(np.float64), gid, 2, *blk)
        assert np.array_equal(g1, egt1) and np.array_equal(e1, eeq1), blk


def test_plain_c_client_of_the_abi(pkg, oracle, tmp_path):
    """include/reo_hip.h from a C program (no Python, no torch in that process): compile tests/abi/abi_client.c
    with gcc, run it, and compare what it prints with the oracle and with the ctypes path."""
    import shutil
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    libdir = os.path.join(root, "rankcompv3.jl_amd")
    exe = str(tmp_path / "abi_client")
    gcc = shutil.which("gcc")
    assert gcc, "gcc is part of the image"
    subprocess.run([gcc, "-O1", "-Wall", "-I" + os.path.join(root, "include"), os.path.join(root, "tests", "abi", "abi_client.c"),
                    "-L" + libdir, "-lreo_hip", "-Wl,-rpath," + libdir, "-o", exe], check=True)
    G, S, seed = 300, 26, 0x5EED0031
    X = pkg.synth.t1_counts(G, S, seed)
    gid = np.array([0, 1] * 13, dtype=np.int32)
    ref0 = pkg.synth.ref_mask(G, 90, seed)
    text = "%d %d 2 %d 0.05 1.0 0.05 9 1\n" % (G, S, seed)
    text += " ".join(map(str, gid)) + "\n" + " ".join(str(int(v)) for v in ref0) + "\n"
    text += " ".join(map(str, np.asfortranarray(X).ravel(order="F"))) + "\n"
    out = subprocess.run([exe], input=text, capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    # the same through the in-library RCCL path (a communicator of one rank: reo_build_pairs ends with an ncclAllReduce of
    # the class table) and through reo_create_multi over the GPUs of this box: identical output
    for mode in ("rccl", "multi"):
        alt = subprocess.run([exe, mode], input=text, capture_output=True, text=True, timeout=300)
        assert alt.returncode == 0, (mode, alt.stderr)
        assert alt.stdout[alt.stdout.index("passes "):] == out.stdout, mode   # (RCCL prints a version banner first)
    lines = out.stdout.strip().splitlines()
    passes = int(lines[0].split()[1])
    trace = [tuple(int(v) for v in ln.split()[1:]) for ln in lines[1:1 + passes]]
    res = np.array([[float(v) for v in ln.split()] for ln in lines[1 + passes:]])
    exp, iters, etrace = oracle.identify_degs(X.astype(np.float64), gid, 2, 0.05, 1.0, 0.05, ref0, 9, 1, seed)
    assert passes == iters and trace == etrace
    _check_result(res, exp)
    run = pkg.run_identify_degs(X, gid, list(range(G)), 0.05, 1.0, 0.05, ref0, 9, 1, seed=seed, device=0)
    assert np.array_equal(res, run.result), "the C client and the ctypes path must agree to the last bit"


def test_reoa_bundled_test_data_end_to_end(pkg, oracle, tmp_path):
    """BASELINE config 1: reoa(use_testdata="yes") on the reference's bundled files (README.md:26-56)."""
    seed = 0x5EED0001
    df = pkg.reoa(use_testdata="yes", work_dir=str(tmp_path), seed=seed, device=0)
    assert df.shape == (19999, 2) and list(df.columns) == ["gene_name", "group1_vs_group2"]  # README.md:39-41
    assert set(df["group1_vs_group2"]) <= {"up", "down", "no change"} and df["gene_name"][0] == "DE1"
    for f in ("fn_expr_group1_group2_result.tsv", "fn_expr_df_expr.tsv", "fn_expr_df_meta.tsv", "fn_expr_gene_up_down.tsv"):
        assert (tmp_path / f).stat().st_size > 0
    run = df.attrs["run"]
    import importlib
    R = importlib.import_module(pkg.__name__ + ".reoa")
    prep = R.prepare(
        os.path.join(os.path.dirname(__file__), "golden", "fn_expr.txt"),
        os.path.join(os.path.dirname(__file__), "golden", "fn_meta.txt"), seed=seed)
    gid, lev = pkg.encode_groups(prep["sample_groups"])
    assert run.thresholds[:, 0].tolist() == [5, 5]  # the WARN branch of :87-90 for 5-vs-5 samples
    exp, iters, trace = oracle.identify_degs(prep["data"].astype(np.float64), gid, 2, 0.01, 1.0, 0.05, prep["ref"], 128, 5, seed)
    assert run.iters_run == iters and run.trace == trace
    _check_result(run.result, exp)
    from oracle import reo_numpy as rn
    assert df["group1_vs_group2"].tolist() == rn.labels(exp, 1.0, 0.05).tolist()
    # README.md:38-55 displays eleven genes, all "up": made by a predecessor of :412-416 (see
    # tests/test_oracle.py::test_readme_quick_start_display_against_the_oracle), so not these labels -- but "up" needs z1 > 0
    # (:428) in either version, and the GPU's z1 of those eleven genes is positive (chance: 0.49^11 = 4e-4)
    shown = [prep["gene_names"].index(n) for n in ("DE1", "DE2", "DE3", "DE4", "DE5", "DE6", "EE19996", "EE19997", "EE19998", "EE19999", "EE20000")]
    assert (run.result[shown, 14] > 3.0).all() and 0.45 < (run.result[:, 14] > 0).mean() < 0.55


def test_pseudobulk_kernels_dense_and_csc(pkg, rn):
    """pseudobulk_group (src/RankCompV3.jl:56-67) on the GPU: exact for counts, bit-exact for Float64
    (cells are added in the shuffled order, left to right, like sum(eachrow(...)) at :63)."""
    import importlib
    import scipy.sparse as sp
    R = importlib.import_module(pkg.__name__ + ".reoa")
    rng = np.random.default_rng(11)
    G, C = 2500, 900
    dense_i = (rng.random((G, C)) < 0.08) * rng.integers(1, 400, size=(G, C))
    dense_f = dense_i * rng.random((G, C)) * 1.37
    orders, ptrs, base = [], [0], 0
    for gi, (lo, hi) in enumerate(((0, 500), (500, 900))):  # two groups of cells, 64 and 7 profiles
        o, p = R.pseudobulk_partition(hi - lo, 64 if gi == 0 else 7, seed=9, stream=gi)
        orders.append(o + lo)
        ptrs += (p[1:] + base).tolist()
        base += hi - lo
    order, ptr = np.concatenate(orders), np.asarray(ptrs, dtype=np.int32)
    assert len(ptr) - 1 == 63 + 7  # ceil(500/64) = 8 cells per chunk -> 63 chunks; ceil(400/7) = 58 -> 7 chunks
    with pkg.Context(device=0) as ctx:
        for X in (dense_i.astype(np.int64), dense_f):
            exp = rn.pseudobulk(X, order, ptr)
            assert np.array_equal(ctx.pseudobulk(X, order, ptr), exp)                  # dense
            assert np.array_equal(ctx.pseudobulk(sp.csc_matrix(X), order, ptr), exp)   # CSC
        assert np.array_equal(exp.sum(axis=1), dense_f[:, order].sum(axis=1)) or np.allclose(exp.sum(axis=1), dense_f.sum(axis=1))
        # more gene rows than one LDS tile, empty cells, a dropped cell, an empty profile
        G2 = 40000
        X2 = sp.random(G2, 300, density=0.01, random_state=3, format="csc", dtype=np.float64)
        X2.data = np.round(X2.data * 50)
        o2 = np.arange(299, dtype=np.int32)[::-1].copy()
        p2 = np.array([0, 100, 100, 250, 299], dtype=np.int32)
        exp2 = rn.pseudobulk(X2.toarray(), o2, p2)
        assert np.array_equal(ctx.pseudobulk(X2, o2, p2), exp2) and not exp2[:, 1].any()
        assert np.array_equal(ctx.pseudobulk(X2.toarray(), o2, p2), exp2)
        with pytest.raises(pkg.DimensionMismatch):
            ctx.pseudobulk(X2, np.array([5, 300], dtype=np.int32), np.array([0, 2], dtype=np.int32))


@pytest.mark.parametrize("kind", ["small_counts", "wide_counts", "huge_value", "float", "many_genes"])
def test_pseudobulk_csc_chunked_narrowed_upload(pkg, kind, monkeypatch):
    """CSC input from host memory (round 5): the entries travel in chunks of 4 M, a pool of host threads checks the row indices and
    narrows them (16 bits when G <= 65 536) and the Int64 values (16, else 32 bits, else the caller's array itself; the widths only
    grow) into pinned staging, kernels widen them on the device.  Exact sums for every width, for Float64 values (not narrowed),
    above 65 536 genes (32-bit row indices), with the narrowing switched off; a row index outside the matrix is refused."""
    import scipy.sparse as sp
    rng = np.random.default_rng(21)
    G, C, dens = (30000, 3200, 0.06) if kind != "many_genes" else (70000, 1100, 0.06)     # 5.8 M / 4.6 M entries: two chunks
    nnz_per = rng.binomial(G, dens, size=C)
    indptr = np.concatenate([[0], np.cumsum(nnz_per)]).astype(np.int64)
    rows = np.concatenate([np.sort(rng.choice(G, n, replace=False)) for n in nnz_per]).astype(np.int32)
    nnz = int(indptr[-1])
    assert nnz > (4 << 20)
    if kind == "float": vals = np.round(rng.lognormal(1.0, 1.0, nnz), 3)
    else:
        vals = rng.integers(1, 300, size=nnz).astype(np.int64)
        if kind == "wide_counts": vals[rng.integers(0, nnz, 50)] = 100000           # 32 bits from the first chunk on
        if kind == "huge_value": vals[nnz - 7] = 2 ** 40; vals[5] = -40000           # 32 bits in chunk one, the raw array in chunk two
    X = sp.csc_matrix((vals, rows, indptr), shape=(G, C))
    order = rng.permutation(C).astype(np.int32)
    ptr = np.array([0, 700, 700, 1500, C], dtype=np.int32) if C > 1500 else np.array([0, 400, 400, 900, C], dtype=np.int32)
    exp = np.stack([np.asarray(X[:, order[ptr[o]:ptr[o + 1]]].sum(axis=1)).ravel() for o in range(len(ptr) - 1)], axis=1)
    if kind == "float":   # the reference's left-to-right order of the row sums (:63)
        from oracle import reo_numpy as rn
        exp = None
    for threads in (None, "0", "3"):
        if threads is None: monkeypatch.delenv("REO_UPLOAD_THREADS", raising=False)
        else: monkeypatch.setenv("REO_UPLOAD_THREADS", threads)
        with pkg.Context(device=0) as ctx:
            got = ctx.pseudobulk(X, order, ptr)
            if exp is None:
                exp = got                        # Float64: bit-equal across the settings; against numpy within rounding below
                dense = np.stack([np.asarray(X[:, order[ptr[o]:ptr[o + 1]]].sum(axis=1)).ravel() for o in range(len(ptr) - 1)], axis=1)
                assert np.allclose(got, dense, rtol=1e-12, atol=0)
            assert np.array_equal(got, exp), (kind, threads)
            bad = X.copy(); bad.indices = bad.indices.copy(); bad.indices[nnz - 3] = G        # one row index outside the matrix
            bad.has_sorted_indices = True
            with pytest.raises(pkg.DimensionMismatch, match="row index"):
                ctx.pseudobulk(bad, order, ptr)
            assert np.array_equal(ctx.pseudobulk(X, order, ptr), exp)                        # the context works again


def test_reoa_pseudobulk_mode(pkg, oracle, tmp_path):
    """reoa(n_pseudo=...) (src/RankCompV3.jl:608-612): cells -> pseudo-bulk profiles -> identify_degs."""
    import importlib
    R = importlib.import_module(pkg.__name__ + ".reoa")
    G, C, seed = 400, 240, 0x5EED0007
    X = pkg.synth.t1_counts(G, C, seed)
    names = [f"g{i}" for i in range(G)]
    cells = [f"c{j}" for j in range(C)]
    with open(tmp_path / "cells.tsv", "w") as f:
        f.write("Name\t" + "\t".join(cells) + "\n")
        for i in range(G):
            f.write(names[i] + "\t" + "\t".join(str(v) for v in X[i]) + "\n")
    with open(tmp_path / "meta.tsv", "w") as f:
        f.write("Name\tGroup\n" + "".join(f"{c}\t{'ctrl' if j < C // 2 else 'case'}\n" for j, c in enumerate(cells)))
    df = pkg.reoa(str(tmp_path / "cells.tsv"), str(tmp_path / "meta.tsv"), n_pseudo=12, use_hk_genes="no",
                  ref_gene_max=100, n_iter=8, n_conv=1, work_dir=str(tmp_path), seed=seed, device=0)
    prep = R.prepare(str(tmp_path / "cells.tsv"), str(tmp_path / "meta.tsv"), n_pseudo=12, use_hk_genes="no",
                     ref_gene_max=100, seed=seed)  # host sums: the same matrix
    assert prep["data"].shape[1] == 24 and prep["sample_names"][0] == "ctrl_x1" and prep["sample_groups"][-1] == "case"
    meta_out = (tmp_path / "cells_df_meta.tsv").read_text().splitlines()
    assert meta_out[1] == "ctrl_x1\tctrl" and len(meta_out) == 25
    gid, lev = pkg.encode_groups(prep["sample_groups"])
    exp, iters, trace = oracle.identify_degs(prep["data"].astype(np.float64), gid, 2, 0.01, 1.0, 0.05, prep["ref"], 8, 1, seed)
    run = df.attrs["run"]
    assert run.iters_run == iters and run.trace == trace
    _check_result(run.result, exp)


def _random_case(rng):
    G = int(rng.integers(12, 700))
    ng = int(rng.choice([2, 2, 2, 3, 4]))
    sizes = rng.integers(2, 24, size=ng)
    S = int(sizes.sum())
    labels = np.concatenate([[f"grp{g}"] * int(n) for g, n in enumerate(sizes)])
    if rng.random() < 0.5:
        labels = labels[rng.permutation(S)]
    kind = rng.choice(["small_int", "wide_int", "float_band", "ranks", "float_cont"])
    if kind == "small_int":
        X = rng.integers(0, int(rng.integers(2, 12)), size=(G, S))
    elif kind == "wide_int":
        X = rng.integers(-50000, 50000, size=(G, S))
    elif kind == "float_band":  # many values closer than 0.1 apart, some exactly 0.1 apart
        X = np.round(rng.normal(5, 1.0, size=(G, S)), 1) + rng.choice([0.0, 0.04, 0.099, 0.1], size=(G, S))
    elif kind == "ranks":
        X = np.argsort(np.argsort(rng.random((G, S)), axis=0), axis=0)
    else:
        X = rng.normal(0, 3, size=(G, S))
    return dict(G=G, S=S, ng=ng, labels=labels, X=X, kind=str(kind), pval_reo=float(rng.choice([0.01, 0.05, 0.3])),
                n_conv=int(rng.choice([1, 5])), n_iter=int(rng.integers(1, 7)), seed=int(rng.integers(0, 2 ** 40)),
                nref=int(rng.integers(3, G)))


def test_randomized_sweep_against_oracle(pkg, oracle):
    """Random shapes, group layouts (2-4 groups, contiguous or interleaved), dtypes and tie structures."""
    rng = np.random.default_rng(20261003)
    kinds = set()
    for case_no in range(40):
        cs = _random_case(rng)
        G, S = cs["G"], cs["S"]
        if G < 10:
            continue
        gid, lev = pkg.encode_groups(cs["labels"])
        ref0 = pkg.synth.ref_mask(G, cs["nref"], cs["seed"])
        tag = (case_no, cs["kind"], G, S, cs["ng"])
        run = pkg.run_identify_degs(cs["X"], cs["labels"], list(range(G)), cs["pval_reo"], 1.0, 0.05, ref0, cs["n_iter"],
                                    cs["n_conv"], seed=cs["seed"], device=0)
        Xf = np.asarray(cs["X"], dtype=np.float64)
        for cm in run.comparisons:
            exp, iters, trace = oracle.identify_degs(Xf, gid, len(lev), cs["pval_reo"], 1.0, 0.05, ref0, cs["n_iter"],
                                                     cs["n_conv"], cs["seed"], k=cm["k"])
            assert cm["iters_run"] == iters and cm["trace"] == trace, tag
            assert np.array_equal(cm["result"][:, 2:11], exp[:, 2:11]), tag
            ok = np.isfinite(exp).all(axis=1)
            assert np.allclose(cm["result"][ok][:, :2], exp[ok][:, :2], rtol=0, atol=P_ATOL), tag
            assert np.allclose(cm["result"][ok][:, 11:], exp[ok][:, 11:], rtol=STAT_RTOL, atol=1e-9), tag
        if case_no % 4 == 0:  # raw counts on the whole pair matrix
            with pkg.Context(device=0, seed=cs["seed"]) as ctx:
                ctx.set_matrix(cs["X"])
                ctx.set_groups(gid, len(lev))
                gt, eq = ctx.pair_counts(0, G, 0, G)
                egt, eeq = oracle.pair_counts(Xf, gid, len(lev), 0, G, 0, G)
                assert np.array_equal(gt, egt) and np.array_equal(eq, eeq), tag
        kinds.add(cs["kind"])
    assert len(kinds) == 5


def test_one_context_through_many_problems(pkg, oracle):
    """A context is reused the way a long-lived caller would: problems of different sizes, element types, tie structure and
    group layouts one after the other (buffers grow and are kept, the cached item list / transform / table of the previous
    problem must never leak into the next), small -> large -> small, across the size where the light passes start, and
    the first problem once more at the end."""
    rng = np.random.default_rng(424242)
    seed = 0x5EED0123
    sizes = [(300, 3), (5000, 2), (40, 2), (9000, 2), (700, 4), (5000, 2), (12, 2), (2600, 3)]
    problems = []
    for G, ng in sizes:
        per = rng.integers(3, 20, size=ng)
        S = int(per.sum())
        gid = np.concatenate([[g] * int(n) for g, n in enumerate(per)]).astype(np.int32)
        if rng.random() < 0.5:
            gid = gid[rng.permutation(S)]
        gid, _ = pkg.encode_groups([f"grp{g}" for g in gid])   # ids in order of first appearance (unique(), :353)
        kind = str(rng.choice(["small_int", "ranks", "float_band", "wide_int"]))
        if kind == "small_int":
            X = rng.integers(0, 9, size=(G, S))
        elif kind == "ranks":
            X = np.argsort(np.argsort(rng.random((G, S)), axis=0), axis=0)
        elif kind == "float_band":
            X = np.round(rng.normal(5, 1.0, size=(G, S)), 1) + rng.choice([0.0, 0.04, 0.099, 0.1], size=(G, S))
        else:
            X = rng.integers(-50000, 50000, size=(G, S))
        problems.append(dict(G=G, S=S, ng=ng, gid=gid, X=X, kind=kind, ref0=pkg.synth.ref_mask(G, int(rng.integers(3, G)), seed),
                             n_iter=int(rng.integers(1, 12)), n_conv=int(rng.choice([0, 5])), pval=float(rng.choice([0.01, 0.05]))))
    problems.append(problems[0])
    first = None
    with pkg.Context(device=0, seed=seed) as ctx:
        for no, pr in enumerate(problems):
            tag = (no, pr["kind"], pr["G"], pr["S"], pr["ng"])
            ctx.set_matrix(pr["X"])
            ctx.set_groups(pr["gid"], pr["ng"])
            ctx.compute_thresholds(pr["pval"])
            Xf = np.asarray(pr["X"], dtype=np.float64)
            for k in (range(pr["ng"]) if pr["ng"] > 2 else [0]):
                ctx.build_pairs(k)
                res, iters, trace = ctx.identify_degs(pr["ref0"], 1.0, 0.05, pr["n_iter"], pr["n_conv"])
                exp, eit, etr = oracle.identify_degs(Xf, pr["gid"], pr["ng"], pr["pval"], 1.0, 0.05, pr["ref0"], pr["n_iter"], pr["n_conv"], seed, k=k)
                assert iters == eit and trace == etr, tag
                assert np.array_equal(res[:, 2:11], exp[:, 2:11]), tag
                ok = np.isfinite(exp).all(axis=1)
                assert np.allclose(res[ok][:, :2], exp[ok][:, :2], rtol=0, atol=P_ATOL), tag
                assert np.allclose(res[ok][:, 11:], exp[ok][:, 11:], rtol=STAT_RTOL, atol=1e-9), tag
                if no == 0 and k == 0:
                    first = res.copy()
                if no == len(problems) - 1 and k == 0:
                    assert np.array_equal(res, first, equal_nan=True), "the same problem on the same context gave a different result the second time"


def test_contexts_in_concurrent_host_threads(pkg, oracle):
    """Three callers, each with its own context on the same device, running at the same time (ctypes drops the GIL during a
    call): contexts share the device, the per-device self-test verdict and -- since round 5 -- the process-wide block cache, the
    host thread pool of the narrowed upload and the store of pair-kernel work lists; reo_last_error is per thread."""
    import threading
    seed = 77
    jobs = []
    for t, (G, S, fam) in enumerate([(5200, 40, "t1"), (4700, 56, "t0"), (900, 30, "float")]):
        X = {"t0": pkg.synth.t0_ranks, "t1": pkg.synth.t1_counts, "float": pkg.synth.float_expr}[fam](G, S, seed + t)
        gid, lev = pkg.encode_groups(pkg.synth.groups(S))
        jobs.append(dict(X=X, gid=gid, ng=len(lev), ref0=pkg.synth.ref_mask(G, 200, seed + t), out=[], err=[]))

    def work(job):
        try:
            for rep in range(4):
                with pkg.Context(device=0, seed=seed) as ctx:
                    if rep % 2 == 0:   # the matrix first (one copy), or groups and thresholds first: the pipelined, narrowed upload, whose
                        ctx.set_matrix(job["X"])          # host thread pool, staging blocks and work-list store the contexts of all threads share
                    ctx.set_groups(job["gid"], job["ng"])
                    ctx.compute_thresholds(0.01)
                    if rep % 2 == 1:
                        ctx.set_matrix(job["X"])
                    ctx.build_pairs(0)
                    job["out"].append(ctx.identify_degs(job["ref0"], 1.0, 0.05, 24, 0))
        except Exception as e:  # noqa: BLE001 -- reported by the main thread
            job["err"].append(e)

    threads = [threading.Thread(target=work, args=(j,)) for j in jobs]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    for j in jobs:
        assert not j["err"], j["err"]
        exp, eit, etr = oracle.identify_degs(np.asarray(j["X"], dtype=np.float64), j["gid"], j["ng"], 0.01, 1.0, 0.05, j["ref0"], 24, 0, seed)
        for res, iters, trace in j["out"]:
            assert iters == eit and trace == etr
            _check_result(res, exp)


def test_maximum_gene_count_65535(pkg, oracle):
    """G = 65535 (the u16 position limit), S = 16: sampled pair blocks and the mirror rule at full size."""
    G, S, seed = 65535, 16, 77
    rng = np.random.default_rng(seed)
    X = rng.integers(0, 40, size=(G, S))  # tie-rich
    group = pkg.synth.groups(S)
    gid, lev = pkg.encode_groups(group)
    with pkg.Context(device=0, seed=seed) as ctx:
        ctx.set_matrix(X)
        ctx.set_groups(gid, 2)
        thr = ctx.compute_thresholds(0.01)
        assert thr[:, 0].tolist() == [8, 8]
        Xf = X.astype(np.float64)
        for (i0, i1, j0, j1) in [(0, 40, 65400, 65535), (65500, 65535, 0, 300), (32760, 32790, 32700, 32900)]:
            gt, eq = ctx.pair_counts(i0, i1, j0, j1)
            egt, eeq = oracle.pair_counts(Xf, gid, 2, i0, i1, j0, j1)
            assert np.array_equal(gt, egt) and np.array_equal(eq, eeq)
        ctx.build_pairs(0)
        for (a, b) in [(0, 65279), (65279, 65279), (40000, 123)]:
            c1 = ctx.get_codes(a, a + 256, b, b + 256).astype(np.int16)
            c2 = ctx.get_codes(b, b + 256, a, a + 256).astype(np.int16)
            off = (np.arange(a, a + 256)[:, None] != np.arange(b, b + 256)[None, :])
            assert np.array_equal(c1[off], (8 - c2.T)[off]) and ((c1 == 255) == ~off).all()
        # codes of a block against the oracle's classification of the same counts (tie coins included)
        i0, i1, j0, j1 = 1000, 1032, 60000, 60256
        gt, eq = oracle.pair_counts(Xf, gid, 2, i0, i1, j0, j1)
        exp = np.empty((i1 - i0, j1 - j0), dtype=np.uint8)
        for a in range(i1 - i0):
            for b in range(j1 - j0):
                nre = [int(gt[a, b, g]) + (oracle.tie_wins(seed, i0 + a, j0 + b, g, int(eq[a, b, g])) if eq[a, b, g] else 0) for g in (0, 1)]
                ic = 2 if nre[0] >= 8 else (0 if 8 - nre[0] >= 8 else 1)
                it = 2 if nre[1] >= 8 else (0 if 8 - nre[1] >= 8 else 1)
                exp[a, b] = 3 * ic + it
        assert np.array_equal(ctx.get_codes(i0, i1, j0, j1), exp)
        ref0 = pkg.synth.ref_mask(G, 3000, seed)
        cont = ctx.tally(ref0)
        assert np.array_equal(cont.sum(axis=1), ref0.sum() - ref0.astype(np.int64)) and cont.min() >= 0
        res, iters, trace = ctx.identify_degs(ref0, 1.0, 0.05, 4, 5)
        assert 1 <= iters <= 4 and np.isfinite(res).all()


def test_full_identify_degs_at_65535_genes(pkg, oracle):
    """The whole path at the u16 limit (G = 65535, tie-rich counts, 16 samples): transform (the 16-bit-bin histogram form
    if the counts have at most 16 varying bits, else the segmented path), tie-rich K1, 64 sort chunks in K3; trace and tallies bit-exact, statistics within tolerance."""
    G, S, seed = 65535, 16, 0x5EED0021
    X = pkg.synth.t1_counts(G, S, seed)
    group = np.array(["a", "b"] * 8, dtype=object)
    gid, lev = pkg.encode_groups(group)
    ref0 = pkg.synth.ref_mask(G, 3000, seed)
    run = pkg.run_identify_degs(X, group, list(range(G)), 0.05, 1.0, 0.05, ref0, 12, 5, seed=seed, device=0)
    exp, iters, trace = oracle.identify_degs(X.astype(np.float64), gid, 2, 0.05, 1.0, 0.05, ref0, 12, 5, seed)
    assert run.iters_run == iters and run.trace == trace and iters >= 2
    assert run.info["transform_in_lds"] in (1, 2) and run.info["has_ties"] == 1
    _check_result(run.result, exp)


@pytest.mark.parametrize("G,S,family,n_iter", [(70000, 24, "t1", 8), pytest.param(140000, 16, "t0", 6, marks=pytest.mark.gpu_slow),
                                                     (70000, 24, "float", 4), (66000, 24, "big_int", 4)])
def test_more_than_65535_genes(pkg, oracle, G, S, family, n_iter, monkeypatch):
    """Above 65 535 genes positions take 17 or 18 bit planes (round 3): 32-bit transform rows, the big plane layout, the
    generated count loop for NB = 17 / 18 at two waves per SIMD, the sorting passes with their splitter tables in dynamic LDS.
    Round 5: every sample is ranked by ONE workgroup here too (t_sample_big: the bucket ranking with its by-slot records in L2;
    Int64 of any width and Float64 with the 0.1 band) -- rocprim's segmented sort is no longer on the path (REO_TRANSFORM=segmented
    still runs it: compared below).  Sampled blocks of the class table against the oracle's counts -- first / last columns, the
    diagonal, across the 65 536 boundary, the padded tail -- and the whole run (trace, tallies, statistics) against the oracle."""
    seed = 0x5EED0065
    X = {"t1": pkg.synth.t1_counts, "t0": pkg.synth.t0_ranks, "float": pkg.synth.float_expr,
         "big_int": lambda G_, S_, seed_: np.random.default_rng(seed_).integers(-2 ** 40, 2 ** 40, size=(G_, S_))}[family](G, S, seed)
    group = np.array(["a", "b"] * (S // 2), dtype=object)      # interleaved groups: the transform re-orders the samples
    gid, lev = pkg.encode_groups(group)
    ref0 = pkg.synth.ref_mask(G, 3000, seed)
    Xf = np.asfortranarray(X.astype(np.float64))
    with pkg.Context(device=0, seed=seed) as ctx:
        ctx.set_matrix(X); ctx.set_groups(gid, 2)
        thr = ctx.compute_thresholds(0.05)
        ctx.build_pairs(0)
        info = ctx.info()
        assert info["has_ties"] == (1 if family in ("t1", "float") else 0) and info["transform_in_lds"] == 3
        for (i0, j0, n) in [(0, G - 40, 40), (0, 0, 40), (65520, 65520, 48), (65500, 100, 40), (100, 65530, 40), (G - 48, G - 48, 48),
                            (G - 33, 17, 32), (40000, G - 300, 32)]:
            exp = _expected_block_codes(oracle, Xf, gid, thr, seed, i0, i0 + n, j0, j0 + n)
            assert np.array_equal(ctx.get_codes(i0, i0 + n, j0, j0 + n), exp), (i0, j0)
        cont = ctx.tally(ref0)
        assert np.array_equal(cont.sum(axis=1), ref0.sum() - ref0.astype(np.int64)) and cont.min() >= 0
        res, iters, trace = ctx.identify_degs(ref0, 1.0, 0.05, n_iter, 3)
        for (i0, i1, j0, j1) in [(0, 24, G - 40, G), (65530, 65560, 65500, 65580)]:
            gt, eq = ctx.pair_counts(i0, i1, j0, j1)              # the raw counts (deterministic part) on the big plane layout
            egt, eeq = oracle.pair_counts(Xf, gid, 2, i0, i1, j0, j1)
            assert np.array_equal(gt, egt) and np.array_equal(eq, eeq)
    exp, eit, etr = oracle.identify_degs(Xf, gid, 2, 0.05, 1.0, 0.05, ref0, n_iter, 3, seed)
    assert iters == eit and trace == etr, (iters, eit, trace, etr)
    _check_result(res, exp)
    # the library's segmented sort (rounds 3-4) gives the same counts, everywhere sampled
    blocks = [(0, 64, G - 64, G), (65500, 65564, 0, 64), (G - 64, G, 30000, 30064), (12345, 12409, 54321, 54385)]
    with pkg.Context(device=0, seed=seed) as ctx:
        ctx.set_matrix(X); ctx.set_groups(gid, 2)
        mine = [ctx.pair_counts(*b) for b in blocks]
    monkeypatch.setenv("REO_TRANSFORM", "segmented")
    other, info_o = _counts_blocks(pkg, X, gid, 2, blocks, seed=seed)   # (the A/B build's segmented sort, or the oracle)
    for b, (gt, eq), (sgt, seq) in zip(blocks, mine, other):
        assert np.array_equal(gt, sgt) and np.array_equal(eq, seq), b
    assert info_o["transform_in_lds"] == 0


def test_gene_count_limits(pkg, oracle, monkeypatch):
    """G <= 262 143 (18 position planes).  Above 65 535 genes one-vs-rest works too (the wave kernels have loops for 17 / 18
    planes; with the count planes shared, or recounted per comparison); only more than 65 535 samples are refused there."""
    rng = np.random.default_rng(1)
    with pkg.Context(device=0, seed=1) as ctx:
        with pytest.raises(pkg.ReoError):
            ctx.set_matrix(np.zeros((262144, 2), dtype=np.int64))
    G, S, seed = 66000, 9, 0x5EED0067
    X = rng.integers(0, 1000, size=(G, S))
    gid = np.array([0, 1, 2, 0, 1, 2, 0, 1, 2], dtype=np.int32)
    Xf = np.asfortranarray(X.astype(np.float64))
    for share in ("1", "0"):
        monkeypatch.setenv("REO_SHARE_GROUP_COUNTS", share)
        with pkg.Context(device=0, seed=seed) as ctx:
            ctx.set_matrix(X); ctx.set_groups(gid, 3)
            thr = ctx.compute_thresholds(0.5)
            for k in (1, 0):
                ctx.build_pairs(k)
                assert ctx.info()["shared_group_counts"] == int(share)
                for (i0, j0, n) in [(0, G - 40, 40), (65520, 65520, 40), (100, 65530, 32), (G - 33, 17, 32)]:
                    exp = _expected_block_codes(oracle, Xf, gid, thr, seed, i0, i0 + n, j0, j0 + n, ngroups=3, k=k)
                    assert np.array_equal(ctx.get_codes(i0, i0 + n, j0, j0 + n), exp), (share, k, i0, j0)


@pytest.mark.gpu_slow
def test_largest_gene_count_262143(pkg, oracle):
    """The largest gene count, ranked by t_sample_big (counts with many repeated values, then Float64): sampled counts against the
    oracle.  (A 34 GB class table's worth of allocations: a long case, REO_RUN_SLOW=1.)"""
    rng = np.random.default_rng(1)
    Gm, Sm = 262143, 4
    for Xm in (rng.integers(0, 3000, size=(Gm, Sm)), np.round(rng.normal(8, 2, size=(Gm, Sm)), 3)):
        with pkg.Context(device=0, seed=1) as ctx:
            ctx.set_matrix(Xm); ctx.set_groups([0, 1, 0, 1], 2)
            for (i0, i1, j0, j1) in ((0, 32, Gm - 64, Gm), (Gm - 32, Gm, 0, 64), (131000, 131032, 200000, 200064)):
                gt, eq = ctx.pair_counts(i0, i1, j0, j1)
                egt, eeq = oracle.pair_counts(Xm.astype(np.float64), np.array([0, 1, 0, 1], dtype=np.int32), 2, i0, i1, j0, j1)
                assert np.array_equal(gt, egt) and np.array_equal(eq, eeq), (Xm.dtype, i0, j0)
            assert ctx.info()["transform_in_lds"] == 3
    pkg._ffi.trim_memory()   # (a 34 GB class table went back to the block cache: return it to the driver)


@pytest.mark.parametrize("case", ["two_samples", "one_vs_nine", "empty_ref", "full_ref", "g11", "constant", "one_group_all_ties"])
def test_edge_cases_against_oracle(pkg, oracle, case):
    rng = np.random.default_rng(zlib.crc32(case.encode()))
    G, seed, n_iter, n_conv = 64, 5, 5, 1
    X = rng.integers(0, 30, size=(G, 10))
    group = ["a"] * 5 + ["b"] * 5
    ref0 = pkg.synth.ref_mask(G, 20, seed)
    if case == "two_samples":          # one sample per group: m(1) = 1 (the WARN branch)
        X, group = X[:, :2], ["a", "b"]
    elif case == "one_vs_nine":
        group = ["a"] + ["b"] * 9
    elif case == "empty_ref":          # no reference genes: every tally is 0, every N singular
        ref0 = np.zeros(G, dtype=bool)
    elif case == "full_ref":
        ref0 = np.ones(G, dtype=bool)
    elif case == "g11":                # smallest G the reference's slice (:411) accepts: round(Int, 0.5) = 0 for G = 10
        X, ref0 = X[:11], np.ones(11, dtype=bool)
        with pytest.raises(pkg.DimensionMismatch):
            pkg.identify_degs(X[:10], group, list(range(10)), 0.01, 1.0, 0.05, np.ones(10, bool), 3, 1, device=0)
        with pytest.raises(RuntimeError):
            oracle.identify_degs(X[:10].astype(np.float64), pkg.encode_groups(group)[0], 2, 0.01, 1.0, 0.05, np.ones(10, bool), 3, 1, seed)
    elif case == "constant":           # every pair tied in every sample: classes come from the coins alone
        X = np.full((G, 10), 7)
    elif case == "one_group_all_ties":
        X = X.copy(); X[:, :5] = 3
    Gx = X.shape[0]
    gid, lev = pkg.encode_groups(group)
    run = pkg.run_identify_degs(X, group, list(range(Gx)), 0.01, 1.0, 0.05, ref0, n_iter, n_conv, seed=seed, device=0)
    exp, iters, trace = oracle.identify_degs(X.astype(np.float64), gid, 2, 0.01, 1.0, 0.05, ref0, n_iter, n_conv, seed)
    assert run.iters_run == iters and run.trace == trace, case
    _check_result(run.result, exp)
    with pkg.Context(device=0, seed=seed) as ctx:
        ctx.set_matrix(X)
        ctx.set_groups(gid, 2)
        gt, eq = ctx.pair_counts(0, Gx, 0, Gx)
        egt, eeq = oracle.pair_counts(X.astype(np.float64), gid, 2, 0, Gx, 0, Gx)
        assert np.array_equal(gt, egt) and np.array_equal(eq, eeq), case


@pytest.mark.parametrize("G,S", [(17409, 35), (8193, 70), (20481, 33)])
def test_rank_search_and_transform_index_edges(pkg, oracle, G, S):
    """Regression for the GPU memory access fault of round 1 ('Faults'): gene counts whose number of
    1024-gene sort chunks is not a multiple of the 16 search lanes (the lockstep rank searches then have lanes with no
    second chunk: their loads must stay inside the chunk array), one gene past a transform bucket edge, and sample
    counts that are no multiple of 32 or 64.  The whole run against the oracle."""
    seed = 0x5EED0077
    X = pkg.synth.t1_counts(G, S, seed)
    group = pkg.synth.groups(S)
    ref0 = pkg.synth.ref_mask(G, 2500, seed)
    run = pkg.run_identify_degs(X, group, list(range(G)), 0.01, 1.0, 0.05, ref0, 6, 1, seed=seed, device=0)
    gid, lev = pkg.encode_groups(group)
    exp, eit, etr = oracle.identify_degs(X.astype(np.float64), gid, 2, 0.01, 1.0, 0.05, ref0, 6, 1, seed)
    assert run.iters_run == eit and run.trace == etr
    _check_result(run.result, exp)


def test_in_library_rccl_and_multi_context_from_python(pkg, oracle):
    """reo_comm_init_rank (world of one rank) and reo_create_multi (all GPUs of this box) through ctypes: same results
    as the plain context; the exchange stage timer shows that the RCCL call was made."""
    G, S, seed = 2100, 40, 0x5EED0051
    X = pkg.synth.t1_counts(G, S, seed)
    group = pkg.synth.groups(S)
    gid, lev = pkg.encode_groups(group)
    ref0 = pkg.synth.ref_mask(G, 500, seed)

    def run(ctx):
        with ctx:
            ctx.set_profiling(True)
            ctx.set_matrix(X); ctx.set_groups(gid, 2); ctx.compute_thresholds(0.01)
            ctx.build_pairs(0)
            return ctx.get_codes(0, G, 0, G), ctx.identify_degs(ref0, 1.0, 0.05, 6, 1), ctx.timings(), ctx.info()

    code0, (res0, it0, tr0), tm0, info0 = run(pkg.Context(device=0, seed=seed))
    ctx = pkg.Context(device=0, seed=seed)
    ctx.comm_init_rank(pkg._ffi.comm_unique_id(), 0, 1)
    code1, (res1, it1, tr1), tm1, info1 = run(ctx)
    code2, (res2, it2, tr2), tm2, info2 = run(pkg.Context(seed=seed, n_gpus=0))
    for code, res, it, tr in ((code1, res1, it1, tr1), (code2, res2, it2, tr2)):
        assert np.array_equal(code, code0) and np.array_equal(res, res0) and it == it0 and tr == tr0
    assert info2["tiles_owned"] <= info0["tiles_owned"] == info0["tiles_total"]
    exp, eit, etr = oracle.identify_degs(X.astype(np.float64), gid, 2, 0.01, 1.0, 0.05, ref0, 6, 1, seed)
    assert it0 == eit and tr0 == etr
    _check_result(res0, exp)


@pytest.mark.parametrize("shards,family", [(3, "t1"), (2, "t0"), (5, "t1")])
def test_multi_context_orchestration_on_one_device(pkg, oracle, monkeypatch, shards, family):
    """reo_create_multi with REO_MULTI_ONE_DEVICE=1 (comm.hip): the shard contexts share this box's one GPU and their
    packs travel by device copies, everything else is the code an N-GPU node runs -- one host thread per shard building
    its own pair tiles, x_pack on the peers, hand-over to the leader, x_expand_* there, groups / thresholds / matrix
    (host and device source, the latter by ONE 2-D copy) handed to the peers.  Equal to the one-context run, bit for bit."""
    import torch
    monkeypatch.setenv("REO_MULTI_ONE_DEVICE", "1")
    G, S, seed = 3300, 72, 0x5EED0061
    X = (pkg.synth.t1_counts if family == "t1" else pkg.synth.t0_ranks)(G, S, seed)
    gid, lev = pkg.encode_groups(pkg.synth.groups(S))
    ref0 = pkg.synth.ref_mask(G, 700, seed)
    with pkg.Context(device=0, seed=seed) as ctx:
        ctx.set_matrix(X); ctx.set_groups(gid, 2); ctx.compute_thresholds(0.01)
        ctx.build_pairs(0)
        code0 = ctx.get_codes(0, G, 0, G)
        res0, it0, tr0 = ctx.identify_degs(ref0, 1.0, 0.05, 6, 1)
    ld = G + 24   # a device-resident source with a leading dimension
    Xd = torch.zeros((S, ld), dtype=torch.int64, device="cuda:0")
    Xd[:, :G] = torch.from_numpy(np.ascontiguousarray(X.T)).to("cuda:0")
    torch.cuda.synchronize()
    for source in ("host", "device"):
        with pkg.Context(seed=seed, n_gpus=shards) as ctx:
            if source == "host":
                ctx.set_matrix(X)
            else:
                ctx.set_matrix_device(Xd.data_ptr(), G, S, ld, "i64")
            ctx.set_groups(gid, 2); ctx.compute_thresholds(0.01)
            ctx.build_pairs(0)
            info = ctx.info()
            assert info["tiles_owned"] < info["tiles_total"]       # the leader built a share only
            assert np.array_equal(ctx.get_codes(0, G, 0, G), code0)
            res, it, tr = ctx.identify_degs(ref0, 1.0, 0.05, 6, 1)
            assert it == it0 and tr == tr0 and np.array_equal(res, res0)
            ctx.build_pairs(1)                                     # a second table on the same contexts
            code_b = ctx.get_codes(0, 64, 0, G)
        with pkg.Context(device=0, seed=seed) as ctx:
            ctx.set_matrix(X); ctx.set_groups(gid, 2); ctx.compute_thresholds(0.01)
            ctx.build_pairs(1)
            assert np.array_equal(ctx.get_codes(0, 64, 0, G), code_b)
    exp, eit, etr = oracle.identify_degs(X.astype(np.float64), gid, 2, 0.01, 1.0, 0.05, ref0, 6, 1, seed)
    assert it0 == eit and tr0 == etr
    _check_result(res0, exp)


def test_a_hook_that_delivers_wrong_words_is_refused(pkg):
    """A caller-supplied collective that hands back something that is not the other shards' words (here: all ones) makes
    an inconsistent class table -- pairs in two states at once, bits on the diagonal.  reo_build_pairs scans the table it
    got from a hook and answers REO_ECOMM; the passes, whose kernels rely on the tally identity, never run on it."""
    import torch
    G, S, seed, world = 3000, 64, 0x5EED0063, 2
    X = pkg.synth.t0_ranks(G, S, seed)
    gid, lev = pkg.encode_groups(pkg.synth.groups(S))
    dev = torch.device("cuda", 0)

    def gather(send, recv, nbytes, stream):
        torch.cuda.ExternalStream(stream, device=dev).synchronize()
        dst = torch.as_tensor(pkg.dist._RawDevBytes(recv, nbytes * world), device=dev)
        dst.fill_(0xFF)
        torch.cuda.synchronize()

    with pkg.Context(device=0, seed=seed) as ctx:
        ctx.set_matrix(X); ctx.set_groups(gid, 2); ctx.compute_thresholds(0.01)
        ctx.set_shard(1, world)
        ctx.set_allgather(gather)
        with pytest.raises(pkg.ReoError, match="inconsistent class table"):
            ctx.build_pairs(0)
        with pytest.raises(pkg.ReoError):
            ctx.tally(pkg.synth.ref_mask(G, 300, seed))   # still no complete table


@pytest.mark.parametrize("garbage", ["ones", "random"])
def test_passes_on_an_inconsistent_table_answer_an_error_not_a_fault(pkg, oracle, monkeypatch, garbage):
    """Round 3's GPU memory fault, as a regression test: with the scan of a hook's table switched off
    (REO_CHECK_HOOK_TABLE=0, a timing knob) the iteration passes get a table in which pairs hold two states.  The
    tallies derived from its counters go negative, McCullagh's logarithms would give NaN, and a NaN delta1 made
    k3_abs_rank store sorted_p[G + r].  The pass kernels now refuse such tallies (IterState.fault ->
    REO_EHIP), keep delta1 finite and clamp every rank; the context drops the table and works again after a rebuild."""
    import torch
    monkeypatch.setenv("REO_CHECK_HOOK_TABLE", "0")
    monkeypatch.setenv("REO_LIGHT_MIN_G", "64")
    G, S, seed, world = 5000, 64, 0x5EED0071, 2
    X = pkg.synth.t0_ranks(G, S, seed)
    gid, lev = pkg.encode_groups(pkg.synth.groups(S))
    ref0 = pkg.synth.ref_mask(G, 600, seed)
    dev = torch.device("cuda", 0)

    def gather(send, recv, nbytes, stream):
        torch.cuda.ExternalStream(stream, device=dev).synchronize()
        dst = torch.as_tensor(pkg.dist._RawDevBytes(recv, nbytes * world), device=dev)
        if garbage == "ones":
            dst.fill_(0xFF)
        else:
            gen = torch.Generator(device=dev); gen.manual_seed(7)
            dst.copy_(torch.randint(0, 256, (nbytes * world,), dtype=torch.uint8, device=dev, generator=gen))
        torch.cuda.synchronize()

    with pkg.Context(device=0, seed=seed) as ctx:
        ctx.set_matrix(X); ctx.set_groups(gid, 2); ctx.compute_thresholds(0.01)
        ctx.set_shard(1, world)
        ctx.set_allgather(gather)
        ctx.build_pairs(0)   # not scanned: accepted
        with pytest.raises(pkg.ReoError, match="no class table can produce"):
            ctx.identify_degs(ref0, 1.0, 0.05, 12, 0)
        with pytest.raises(pkg.ReoError, match="no class table"):
            ctx.identify_degs(ref0, 1.0, 0.05, 12, 0)   # the table was dropped
        ctx.set_shard(0, 1)
        ctx.set_allgather(None)
        ctx.build_pairs(0)
        res, it, tr = ctx.identify_degs(ref0, 1.0, 0.05, 12, 0)
    exp, eit, etr = oracle.identify_degs(X.astype(np.float64), gid, 2, 0.01, 1.0, 0.05, ref0, 12, 0, seed)
    assert it == eit and tr == etr
    _check_result(res, exp)


@pytest.mark.parametrize("family", ["t0", "t1"])
def test_workgroup_form_of_the_pair_kernel_still_matches(pkg, oracle, monkeypatch, family):
    """REO_K1_WAVE=0 selects round 2's workgroup form of K1 (kept for > 2 groups and as a cross-check): same table.  So does
    the wave form without half-height items (REO_K1_HALF=0; at this size the default deals every item as two halves)."""
    G, S, seed = 2603, 96, 0x5EED0062
    X = (pkg.synth.t1_counts if family == "t1" else pkg.synth.t0_ranks)(G, S, seed)
    gid, lev = pkg.encode_groups(pkg.synth.groups(S))
    codes = []
    for wave, half in (("1", "1"), ("1", "0"), ("0", "1")):
        monkeypatch.setenv("REO_K1_WAVE", wave)
        monkeypatch.setenv("REO_K1_HALF", half)
        with pkg.Context(device=0, seed=seed) as ctx:
            ctx.set_matrix(X); ctx.set_groups(gid, 2); ctx.compute_thresholds(0.01)
            ctx.build_pairs(0)
            codes.append(ctx.get_codes(0, G, 0, G))
    assert np.array_equal(codes[0], codes[1]) and np.array_equal(codes[0], codes[2])
    thr = np.array([oracle.threshold(int((gid == 0).sum())), oracle.threshold(int((gid == 1).sum()))], dtype=np.int32)
    assert np.array_equal(codes[0], oracle.build_codes(X.astype(np.float64), gid, 2, 0, thr, seed))


@pytest.mark.parametrize("window,light,band,xcc,below", [
    ("3", "3", "32", "1", "256"), ("1", "3", "32", "1", "256"), ("12", "3", "2", "1", "256"), ("12", "3", "0", "0", "256"), ("3", "3", "32", "0", "256"),
    ("3", "1", "32", "1", "256"), ("1", "1", "32", "1", "256"), ("12", "1", "2", "1", "256"), ("12", "1", "0", "0", "256"),
    ("3", "1", "32", "0", "256"), ("3", "2", "32", "1", "256"), ("1", "2", "32", "1", "256"),
    ("12", "1", "32", "1", "0"), ("12", "1", "2", "0", "3"), ("3", "1", "32", "1", "12")])
def test_light_passes_on_random_problems_incl_window_failures(pkg, oracle, monkeypatch, window, light, band, xcc, below):
    """The light iteration passes (quantile windows + BH cut from the histogram of step-up ranks) on many small random
    problems: REO_LIGHT_MIN_G lets small gene counts use them, and a window of 1-3 ranks makes windows lose their order
    statistic now and then, so the fall-back to the sorting path and the hand-over of the tally state between the two
    kinds of pass are exercised too.  More passes than usual (n_conv = 0 forces them); different padj / pval cut-offs;
    tie-heavy data puts many equal delta1 values around the quantiles.  light = 1: two launches per pass (the default), 3: one
    launch per pass (round 4: bracketed p-values, the cut from the listed genes' exact ranks), 2: the persistent form.
    below: the two-launch form keeps its rank histogram relative to the last cut, with this many ranks under it (256 in
    production); 0, 3 and 12 make cuts drop out of the histogram, which must send the pass to the sorting path."""
    monkeypatch.setenv("REO_LIGHT_MIN_G", "64")
    monkeypatch.setenv("REO_LIGHT_WINDOW", window)
    monkeypatch.setenv("REO_LIGHT", light)
    monkeypatch.setenv("REO_XCC_LOCAL", xcc)   # 1: rank histogram per XCD with atomics that stay in its L2 (if the self-test passes); 0: device-coherent atomics
    monkeypatch.setenv("REO_LIGHT_BAND", band)  # half width of the list of genes near the BH cut: 0 and 2 make the cut leave it
    monkeypatch.setenv("REO_HIST_BELOW", below)
    rng = np.random.default_rng(4242 + int(window) * 7 + int(light))
    light_batches = 0
    for case_no in range(14):
        cs = _random_case(rng)
        G = max(cs["G"], 120) if cs["G"] >= 120 else 120 + cs["G"]
        X = cs["X"] if cs["X"].shape[0] == G else np.vstack([cs["X"], rng.permutation(cs["X"], axis=0)])[:G] if cs["X"].shape[0] * 2 >= G else None
        if X is None:
            X = rng.integers(0, 9, size=(G, cs["S"]))
        labels = cs["labels"]
        gid, lev = pkg.encode_groups(labels)
        ref0 = pkg.synth.ref_mask(G, max(3, G // 3), cs["seed"])
        n_iter, n_conv = int(rng.integers(6, 14)), int(rng.choice([0, 0, 1]))
        pval_deg, padj_deg = float(rng.choice([1.0, 1.0, 0.2])), float(rng.choice([0.05, 0.3, 0.9]))
        tag = (window, light, band, xcc, below, case_no, cs["kind"], G, cs["S"], cs["ng"], n_iter, n_conv, pval_deg, padj_deg)
        run = pkg.run_identify_degs(X, labels, list(range(G)), cs["pval_reo"], pval_deg, padj_deg, ref0, n_iter, n_conv,
                                    seed=cs["seed"], device=0, profile=True)
        Xf = np.asarray(X, dtype=np.float64)
        for cm in run.comparisons:
            exp, iters, trace = oracle.identify_degs(Xf, gid, len(lev), cs["pval_reo"], pval_deg, padj_deg, ref0, n_iter, n_conv,
                                                     cs["seed"], k=cm["k"])
            assert cm["iters_run"] == iters and cm["trace"] == trace, tag
            assert np.array_equal(cm["result"][:, 2:11], exp[:, 2:11]), tag
            ok = np.isfinite(exp).all(axis=1)
            assert np.allclose(cm["result"][ok][:, :2], exp[ok][:, :2], rtol=0, atol=P_ATOL), tag
            assert np.allclose(cm["result"][ok][:, 11:], exp[ok][:, 11:], rtol=STAT_RTOL, atol=1e-9), tag
        # a pass that ran light needed no K2 launch: fewer K2 launches than passes means light passes happened
        if run.timings["k2_launches"] < sum(c["iters_run"] for c in run.comparisons) + 2 * len(run.comparisons):
            light_batches += 1
    assert light_batches > 0
    if xcc == "1" and light in ("1", "3"):
        assert run.info["xcc_local_histograms"] == 1  # (the self-test of reo_create passes on an MI355X)


@pytest.mark.parametrize("window,below", [("12", "256"), ("3", "256"), ("12", "2")])
def test_cycle_watch_skips_whole_periods_and_changes_nothing(pkg, oracle, monkeypatch, window, below):
    """A loop that does not converge ends in a cycle of reference sets; the light passes notice the first set that returns
    (Brent's search, compared bit for bit) and the host skips whole periods.  iters_run, the trace of every pass and the
    result must be what the oracle gets by executing all n_iter passes -- for cycles of any period (fixed points are
    period 1: an even number of passes is skipped all the same), for n_iter values that leave every remainder, with
    window failures and lost cuts sending passes to the sorting path in between (which drops the snapshot), and with
    n_conv = 1 (a fixed point converges before it is a cycle)."""
    monkeypatch.setenv("REO_LIGHT_MIN_G", "64")
    monkeypatch.setenv("REO_LIGHT_WINDOW", window)
    monkeypatch.setenv("REO_HIST_BELOW", below)
    rng = np.random.default_rng(77001 + int(window))
    found, skipped, periods = 0, 0, set()
    for case_no in range(16):
        cs = _random_case(rng)
        G = max(cs["G"], 120) if cs["G"] >= 120 else 120 + cs["G"]
        X = cs["X"] if cs["X"].shape[0] == G else np.vstack([cs["X"], rng.permutation(cs["X"], axis=0)])[:G] if cs["X"].shape[0] * 2 >= G else None
        if X is None:
            X = rng.integers(0, 9, size=(G, cs["S"]))
        labels = cs["labels"]
        gid, lev = pkg.encode_groups(labels)
        ref0 = pkg.synth.ref_mask(G, max(3, G // 3), cs["seed"])
        n_iter, n_conv = int(rng.integers(30, 71)), int(rng.choice([0, 0, 0, 1]))
        pval_deg, padj_deg = float(rng.choice([1.0, 1.0, 0.2])), float(rng.choice([0.05, 0.3, 0.9]))
        tag = (window, below, case_no, cs["kind"], G, cs["S"], cs["ng"], n_iter, n_conv, pval_deg, padj_deg)
        run = pkg.run_identify_degs(X, labels, list(range(G)), cs["pval_reo"], pval_deg, padj_deg, ref0, n_iter, n_conv,
                                    seed=cs["seed"], device=0, profile=True)
        Xf = np.asarray(X, dtype=np.float64)
        for cm in run.comparisons:
            exp, iters, trace = oracle.identify_degs(Xf, gid, len(lev), cs["pval_reo"], pval_deg, padj_deg, ref0, n_iter, n_conv,
                                                     cs["seed"], k=cm["k"])
            assert cm["iters_run"] == iters and cm["trace"] == trace, tag
            assert np.array_equal(cm["result"][:, 2:11], exp[:, 2:11]), tag
            ok = np.isfinite(exp).all(axis=1)
            assert np.allclose(cm["result"][ok][:, :2], exp[ok][:, :2], rtol=0, atol=P_ATOL), tag
            assert np.allclose(cm["result"][ok][:, 11:], exp[ok][:, 11:], rtol=STAT_RTOL, atol=1e-9), tag
        info = run.info   # (of the last comparison)
        if info["cycle_period"] > 0:
            found += 1; skipped += info["cycle_passes_skipped"]; periods.add(info["cycle_period"])
            assert info["cycle_passes_skipped"] % info["cycle_period"] == 0 and info["cycle_passes_skipped"] % 2 == 0, tag
            assert info["cycle_found_at_pass"] + info["cycle_passes_skipped"] <= n_iter, tag
    assert found > 0 and skipped > 0, (found, skipped, periods)


def test_cycle_watch_at_twenty_thousand_genes_equals_every_pass_executed(pkg, monkeypatch):
    """20 000 genes (the production size of the light passes), 45 forced passes, both data families: with the cycle watch the
    call returns the trace and the result of the call that executes every pass (REO_CYCLE=0), bit for bit."""
    G, S = 20000, 64
    for fam, seed in (("t0", 0x5EED00A1), ("t1", 0x5EED00A2)):
        X = (pkg.synth.t1_counts if fam == "t1" else pkg.synth.t0_ranks)(G, S, seed)
        gid, lev = pkg.encode_groups(pkg.synth.groups(S))
        ref0 = pkg.synth.ref_mask(G, 3000, seed)
        out, info = {}, {}
        for cyc in ("1", "0"):
            monkeypatch.setenv("REO_CYCLE", cyc)
            with pkg.Context(device=0, seed=seed) as ctx:
                ctx.set_matrix(X); ctx.set_groups(gid, 2); ctx.compute_thresholds(0.01); ctx.build_pairs(0)
                out[cyc] = [ctx.identify_degs(ref0, 1.0, 0.05, n_iter, 0) for n_iter in (45, 46, 47)]
                info[cyc] = ctx.info()
        assert info["0"]["cycle_period"] == 0
        assert info["1"]["cycle_period"] > 0 and info["1"]["cycle_passes_skipped"] > 0, (fam, info["1"])
        for (r1, i1, t1), (r0, i0, t0) in zip(out["1"], out["0"]):
            assert i1 == i0 and t1 == t0, fam
            assert np.array_equal(r1, r0, equal_nan=True), fam


def test_sorting_passes_only_and_large_cuts_at_a_size_that_uses_light_passes(pkg, monkeypatch):
    """REO_LIGHT=0 (sorting passes only) on a problem large enough for light passes -- the sorting path must then keep
    asking for itself (a call once waited for ever for light passes that nobody enqueued) -- and cut-offs that put
    thousands of genes inside or near the BH cut (histogram tiles beyond the first four): the same results as the default."""
    G, S, seed = 20000, 64, 0x5EED0093
    X = pkg.synth.t1_counts(G, S, seed)
    group = pkg.synth.groups(S)
    ref0 = pkg.synth.ref_mask(G, 3000, seed)
    out = {}
    for mode in ("3", "1", "1/2", "0"):   # "1/2": the two-launch form with two ranks of histogram under the last cut (cuts drop out of it)
        monkeypatch.setenv("REO_LIGHT", mode[0])
        monkeypatch.setenv("REO_HIST_BELOW", mode[2:] or "256")
        with pkg.Context(device=0, seed=seed) as ctx:
            gid, lev = pkg.encode_groups(group)
            ctx.set_matrix(X); ctx.set_groups(gid, 2); ctx.compute_thresholds(0.01); ctx.build_pairs(0)
            out[mode] = [ctx.identify_degs(ref0, 1.0, padj, 16, 0) for padj in (0.05, 0.9)]
    for mode in ("3", "1", "1/2"):
        for (r1, i1, t1), (r0, i0, t0) in zip(out[mode], out["0"]):
            assert i1 == i0 == 16 and t1 == t0, mode
            assert np.array_equal(r1[:, 2:11], r0[:, 2:11]), mode
            ok = np.isfinite(r0).all(axis=1)
            assert np.allclose(r1[ok][:, :2], r0[ok][:, :2], rtol=0, atol=P_ATOL), mode


@pytest.mark.parametrize("family,ngroups", [("t1", 2), ("t0", 2), ("float", 2), ("t1", 3), ("t1", -2), ("t0", -2)])
def test_more_than_65535_samples(pkg, oracle, family, ngroups):
    """Single-cell mode without pseudo-bulking (src/RankCompV3.jl:608-616 with n_pseudo = 0) can hand over more samples
    than a 16-bit count holds: the wide forms of the pair kernel (two groups: the count loop in runs of 2 047 blocks with
    32-bit totals; more groups: the workgroup form with 32-bit counts).  66 100 samples, the whole run; ngroups = -2: one
    group of 66 000 samples, i.e. a side of more than one run."""
    G, seed = 260, 0x5EED0091
    sizes = [33100, 33000] if ngroups == 2 else ([66000, 100] if ngroups == -2 else [33000, 33040, 60])
    ngroups = abs(ngroups)
    S = sum(sizes)
    gen = {"t0": pkg.synth.t0_ranks, "t1": pkg.synth.t1_counts, "float": pkg.synth.float_expr}[family]
    X = gen(G, S, seed)
    gid = np.concatenate([[g] * n for g, n in enumerate(sizes)]).astype(np.int32)
    if ngroups == 3:  # interleave the two small groups into the big one (group ids stay in order of first appearance)
        perm = np.concatenate([[0, 33000, 66040], np.random.default_rng(1).permutation(np.setdiff1d(np.arange(S), [0, 33000, 66040]))])
        X, gid = X[:, perm], gid[perm]
    ref0 = pkg.synth.ref_mask(G, 80, seed)
    run = pkg.run_identify_degs(X, gid, list(range(G)), 0.01, 1.0, 0.05, ref0, 5, 1, seed=seed, device=0)
    assert run.info["S"] == S
    for cm in run.comparisons:
        exp, eit, etr = oracle.identify_degs(X.astype(np.float64), gid, ngroups, 0.01, 1.0, 0.05, ref0, 5, 1, seed, k=cm["k"])
        assert cm["iters_run"] == eit and cm["trace"] == etr, (family, cm["k"])
        _check_result(cm["result"], exp)
    with pkg.Context(device=0, seed=seed) as ctx:
        ctx.set_matrix(X); ctx.set_groups(gid, ngroups)
        with pytest.raises(pkg.DimensionMismatch):
            ctx.pair_counts(0, 8, 0, 8)          # 16-bit outputs cannot hold these counts
