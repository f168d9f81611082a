"""Parity at the full sizes of BASELINE.json's configs 3, 4 and 5 (through the C ABI, against the CPU oracle).

Config 3 (20 000 x 1 000): the WHOLE class table and the whole iteration loop (128 forced passes for the tie-free
family, the converging run for the tie-rich one) against the oracle -- about a minute of oracle time on the GPU
box's host cores.  Config 4 (30 000 x 4 000, the one-chunk-per-panel geometry), both data families: the WHOLE class table, the tallies and
the whole loop (128 forced passes and the converging run) against the tuned oracle R2 (bit-identical to the plain one,
tests/test_oracle.py), sampled blocks against the plain oracle's counts, and two shards == unsharded.  Config 5 (sparse
20 000 x 50 000 cells -> n_pseudo = 64 -> identify_degs): pseudo-bulk sums and the whole run against the oracle.
Same tolerances as tests/test_gpu_parity.py."""
import threading

import numpy as np
import pytest

from test_gpu_parity import _check_result, _expected_block_codes, _setup

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("family,n_iter,n_conv", [("t0", 128, 0), ("t1", 128, 5)])
def test_config3_whole_table_and_loop_against_oracle(pkg, oracle, family, n_iter, n_conv):
    """BASELINE config 3: 20 000 genes x 1 000 samples, 3 000 initial reference genes (src/RankCompV3.jl:363-425)."""
    G, S, seed = 20000, 1000, 0x5EED0003
    X = (pkg.synth.t0_ranks if family == "t0" else pkg.synth.t1_counts)(G, S, seed)
    group = pkg.synth.groups(S)
    ref0 = pkg.synth.ref_mask(G, 3000, seed)
    ctx, gid, ng = _setup(pkg, X, group, seed)
    Xf = np.asfortranarray(X.astype(np.float64))
    with ctx:
        thr = ctx.get_thresholds()
        assert thr[:, 0].tolist() == [280, 280]
        ctx.build_pairs(0)
        assert ctx.info()["has_ties"] == (0 if family == "t0" else 1)
        if family == "t0":
            code = oracle.build_codes(Xf, gid, ng, 0, [280, 280], seed)     # the reference's pair loop, every pair (36 s on the box's 128 threads)
        else:
            # the tie-rich family against the tuned restatement R2 (bit-identical to the plain one: tests/test_oracle.py) -- 5 s instead of
            # 36 keeps the plain `-m gpu` run inside the driver's limit -- plus sampled blocks against the plain oracle's pair loop
            code = oracle.tuned_decode(oracle.tuned_build_table(Xf, gid, 2, 0.01, seed), 0, G, 0, G)
            for (i0, j0, n) in [(0, 0, 96), (0, G - 96, 96), (G - 96, G - 96, 96), (7000, 13000, 96), (19000, 40, 96)]:
                assert np.array_equal(code[i0:i0 + n, j0:j0 + n], _expected_block_codes(oracle, Xf, gid, thr, seed, i0, i0 + n, j0, j0 + n)), (i0, j0)
        got = ctx.get_codes(0, G, 0, G)
        assert np.array_equal(got, code), "class table differs from the oracle"
        del got
        assert np.array_equal(ctx.tally(ref0), oracle.tally(code, ref0))
        res, iters, trace = ctx.identify_degs(ref0, 1.0, 0.05, n_iter, n_conv)
        exp, eit, etr = oracle.iterate(code, ref0, 1.0, 0.05, n_iter, n_conv)
        assert iters == eit and trace == etr, (iters, eit, [a for a in zip(trace, etr) if a[0] != a[1]][:3])
        if n_conv == 0:
            assert iters == 128
        _check_result(res, exp)
    # the drop-in order of calls (groups and thresholds first, then the matrix from pageable host memory: the upload is pipelined with
    # the ranking and the pair kernel's sides, transform.hip eager_upload) against the same oracle table and loop
    from test_gpu_parity import _eager_ctx
    with _eager_ctx(pkg, np.asfortranarray(X), group, seed) as ctx2:
        ctx2.build_pairs(0)  # (returns at once: reo_set_matrix has launched the pair kernel of both sides)
        res2, it2, tr2 = ctx2.identify_degs(ref0, 1.0, 0.05, n_iter, n_conv)
        assert it2 == eit and tr2 == etr
        _check_result(res2, exp)
        assert np.array_equal(ctx2.get_codes(0, G, 0, G), code), "class table of the pipelined upload differs from the oracle"


def _two_shard_run(pkg, X, gid, seed, pval_reo, ref0, n_iter, n_conv):
    """Two contexts (shards 0/2 and 1/2) on one GPU joined by a host-side all-gather hook; returns what each shard computed."""
    import torch
    world = 2
    barrier = threading.Barrier(world)
    slots = [None] * world
    dev = torch.device("cuda", 0)
    results, errors = [None] * world, []

    def run(rank):
        try:
            def gather(send, recv, nbytes, stream):  # reo_set_allgather: what the in-library RCCL path does
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
                ctx.compute_thresholds(pval_reo)
                ctx.set_shard(rank, world)
                ctx.set_allgather(gather)
                ctx.build_pairs(0)
                info = ctx.info()
                cont = ctx.tally(ref0)
                res = ctx.identify_degs(ref0, 1.0, 0.05, n_iter, n_conv)
                results[rank] = (info, cont, res)
        except Exception:
            import traceback
            errors.append(traceback.format_exc())
            barrier.abort()

    threads = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=600)
    assert not errors, errors
    return results


@pytest.mark.parametrize("family", ["t0", pytest.param("t1", marks=pytest.mark.gpu_slow)])
def test_config4_30000x4000_one_chunk_per_panel(pkg, oracle, family):
    """BASELINE config 4 on one GPU: the geometry with one j-chunk per panel (Q = 1), never reached by the small tests.
    Whole class table (9e8 ordered pairs), tallies, 128 forced passes and the converging run against the oracle."""
    G, S, seed = 30000, 4000, 0x5EED0004
    X = (pkg.synth.t0_ranks if family == "t0" else pkg.synth.t1_counts)(G, S, seed)
    group = pkg.synth.groups(S)
    ref0 = pkg.synth.ref_mask(G, 3000, seed)
    ctx, gid, ng = _setup(pkg, X, group, seed)
    Xf = np.asfortranarray(X.astype(np.float64))
    T = oracle.tuned_build_table(Xf, gid, 2, 0.01, seed)            # R2: about a minute on the box's host cores
    with ctx:
        thr = ctx.get_thresholds()
        assert thr[:, 0].tolist() == [1059, 1059]
        ctx.build_pairs(0)
        info = ctx.info()
        assert info["chunks_per_panel"] == 1 and info["has_ties"] == (0 if family == "t0" else 1) and info["Gp"] == 30720
        for i0 in range(0, G, 5000):                                # the whole table, 5 000 rows at a time
            got = ctx.get_codes(i0, i0 + 5000, 0, G)
            assert np.array_equal(got, oracle.tuned_decode(T, i0, i0 + 5000, 0, G)), ("class table differs from the oracle", i0)
        del got
        # the plain oracle's literal comparator on sampled blocks: first / last panel, diagonal, padded tail, mirrored
        for (i0, j0, n) in [(0, 29960, 40), (0, 0, 40), (15000, 15000, 48), (29952, 29952, 48), (1023, 1000, 40), (777, 22000, 32),
                            (22000, 777, 32), (29968, 31, 32), (29690, 29700, 40)]:
            exp = _expected_block_codes(oracle, Xf, gid, thr, seed, i0, i0 + n, j0, j0 + n)
            assert np.array_equal(ctx.get_codes(i0, i0 + n, j0, j0 + n), exp), (i0, j0)
        cont = ctx.tally(ref0)
        exp1, _, _ = oracle.tuned_iterate(T, ref0, 1.0, 0.05, 1, 0)  # the first pass's tallies are those of ref0
        assert np.array_equal(cont, exp1[:, 2:11].astype(np.int64))
        res, iters, trace = ctx.identify_degs(ref0, 1.0, 0.05, 128, 0)
        exp, eit, etr = oracle.tuned_iterate(T, ref0, 1.0, 0.05, 128, 0)
        assert iters == eit == 128 and trace == etr, (iters, eit, [a for a in zip(trace, etr) if a[0] != a[1]][:3])
        _check_result(res, exp)
        res5, it5, tr5 = ctx.identify_degs(ref0, 1.0, 0.05, 128, 5)
        exp5, eit5, etr5 = oracle.tuned_iterate(T, ref0, 1.0, 0.05, 128, 5)
        assert it5 == eit5 and tr5 == etr5
        _check_result(res5, exp5)
        res, iters, trace = ctx.identify_degs(ref0, 1.0, 0.05, 8, 0)
    # the drop-in order of calls at this size (round 6): the matrix from pageable host memory with groups and thresholds already set -- the
    # pair kernel's items of a group run over THREE ranges of its 63 sample blocks as the chunks arrive, counts parked in between
    # (kernels.hip, K1Args::park) -- against the same oracle table and 8-pass run
    from test_gpu_parity import _eager_ctx
    with _eager_ctx(pkg, np.asfortranarray(X), group, seed) as ctx2:
        assert ctx2.info()["eager_range_launches"] == 6
        ctx2.build_pairs(0)
        for i0 in range(0, G, 5000):
            assert np.array_equal(ctx2.get_codes(i0, i0 + 5000, 0, G), oracle.tuned_decode(T, i0, i0 + 5000, 0, G)), ("class table of the pipelined upload differs", i0)
        res2, it2, tr2 = ctx2.identify_degs(ref0, 1.0, 0.05, 8, 0)
        assert it2 == iters and tr2 == trace and np.array_equal(res2, res)
    del T
    if family == "t1":
        return
    shards = _two_shard_run(pkg, X, gid, seed, 0.01, ref0, 8, 0)
    owned = [s[0]["tiles_owned"] for s in shards]
    assert sum(owned) == shards[0][0]["tiles_total"] and min(owned) > 0.4 * max(owned)
    for info_s, cont_s, (res_s, it_s, tr_s) in shards:
        assert np.array_equal(cont_s, cont)
        assert it_s == iters and tr_s == trace
        assert np.array_equal(res_s, res)      # same kernels on the same complete table: bit for bit


def _synthetic_cells(G, C, seed, dens=0.06):
    """Sparse single-cell counts, CSC: zero-inflated, heavy-tailed gene scales; cells of group 2 shift 10 % of the genes."""
    import scipy.sparse as sp
    rng = np.random.default_rng(seed)
    scale = 2.0 ** rng.integers(0, 9, size=G)
    nnz_per_cell = rng.binomial(G, dens, size=C)
    indptr = np.concatenate([[0], np.cumsum(nnz_per_cell)]).astype(np.int64)
    rows = np.concatenate([np.sort(rng.choice(G, n, replace=False)) for n in nnz_per_cell]).astype(np.int32)
    eff = np.where(rng.random(G) < 0.1, rng.choice([0.5, 2.0], size=G), 1.0)
    cell_of = np.repeat(np.arange(C), nnz_per_cell)
    vals = 1 + rng.poisson(scale[rows] * np.where(cell_of >= C // 2, eff[rows], 1.0))
    return sp.csc_matrix((vals.astype(np.int64), rows, indptr), shape=(G, C))


def test_config5_sparse_cells_to_pseudobulk_to_identify_degs(pkg, oracle, rn):
    """BASELINE config 5 on one GPU: 20 000 genes x 50 000 sparse cells -> n_pseudo = 64 per group (src/RankCompV3.jl:56-67,
    608-612) -> identify_degs on the 20 000 x 128 pseudo-bulk matrix, every stage against the oracle."""
    import importlib
    R = importlib.import_module(pkg.__name__ + ".reoa")
    G, C, seed = 20000, 50000, 0x5EED0005
    X = _synthetic_cells(G, C, seed)
    assert 0.05 < X.nnz / (G * C) < 0.07
    orders, ptrs, base = [], [0], 0
    for gi, (lo, hi) in enumerate(((0, C // 2), (C // 2, C))):
        o, p = R.pseudobulk_partition(hi - lo, 64, seed, gi)     # shuffle + Iterators.partition(…, ceil(c / n_pseudo)), :60-62
        assert len(p) - 1 == 64 and int(np.diff(p).max()) == 391
        orders.append(o + lo); ptrs += (p[1:] + base).tolist(); base += hi - lo
    order, ptr = np.concatenate(orders), np.asarray(ptrs, dtype=np.int32)
    with pkg.Context(device=0, seed=seed) as ctx:
        pb = ctx.pseudobulk(X, order, ptr)
    assert pb.shape == (G, 128) and pb.dtype == np.int64
    exp_pb = np.stack([np.asarray(X[:, order[ptr[o]:ptr[o + 1]]].sum(axis=1)).ravel() for o in range(128)], axis=1)
    assert np.array_equal(pb, exp_pb)
    group = np.array(["g1"] * 64 + ["g2"] * 64, dtype=object)
    keep = (pb > 0).sum(axis=1) > 0                                # the all-zero-row filter of :626
    pbk = np.ascontiguousarray(pb[keep])
    Gk = pbk.shape[0]
    ref0 = pkg.synth.ref_mask(Gk, 3000, seed)
    run = pkg.run_identify_degs(pbk, group, list(range(Gk)), 0.01, 1.0, 0.05, ref0, 128, 5, seed=seed, device=0)
    assert run.thresholds[:, 0].tolist() == [43, 43] and run.info["has_ties"] == 1
    gid, lev = pkg.encode_groups(group)
    exp, eit, etr = oracle.identify_degs(pbk.astype(np.float64), gid, 2, 0.01, 1.0, 0.05, ref0, 128, 5, seed)
    assert run.iters_run == eit and run.trace == etr
    _check_result(run.result, exp)
