"""Multi-GPU protocol without a GPU: ownership rule, disjointness of the shards' class-table bits, and the
all-reduce hook summing the table over gloo with world_size = 2."""
import os
import re
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_python_mirror_constants_match_the_library_source(pkg):
    hdr = open(os.path.join(ROOT, "rankcompv3.jl_amd", "csrc", "reo_internal.h")).read()
    val = {k: int(v) for k, v in re.findall(r"constexpr int (k\w+) = (\d+);", hdr)}
    import sharding_mirror as sh
    assert (val["kTileI"], val["kTileJ"], val["kRJ"], val["kRJTies"], val["kUnitH"]) == (sh.TILE_I, sh.TILE_J, sh.RJ, sh.RJ_TIES, sh.UNIT_H)


@pytest.mark.parametrize("G,slots,world", [(1400, 32, 2), (5000, 208, 2), (5000, 208, 8), (20000, 1008, 8), (900, 16, 3)])
def test_tiles_are_partitioned(pkg, G, slots, world):
    import sharding_mirror as sh
    owner = sh.tile_owner(G, slots, False, world)
    Gp, CJ, Q = sh.geometry(G, slots, False)
    it, jc = np.meshgrid(np.arange(owner.shape[0]), np.arange(owner.shape[1]), indexing="ij")
    below = (jc * CJ + CJ - 1) // 64 < (it * sh.TILE_I) // 64
    assert ((owner == -1) == below).all()          # every tile on or above the diagonal has exactly one owner
    assert owner.max() == min(world, owner.max() + 1) - 1 or owner.max() < world
    counts = np.bincount(owner[owner >= 0], minlength=world)
    if G >= 5000 and world == 2:
        assert counts.min() > 0
    if G >= 20000:
        assert counts.min() > 0 and counts.max() <= 1.3 * counts.mean()   # units are dealt round-robin


def test_pair_masks_cover_every_pair_once(pkg):
    import sharding_mirror as sh
    G, slots, world = 1400, 32, 2
    masks = [sh.owned_pair_mask(G, slots, False, r, world) for r in range(world)]
    tot = sum(m.astype(np.int32) for m in masks)
    off = ~np.eye(G, dtype=bool)
    assert (tot[off] == 1).all() and (tot[~off] == 0).all()
    assert all(np.array_equal(m, m.T) for m in masks) and all(m.any() for m in masks)


@pytest.mark.parametrize("G,ties,world", [(1400, True, 3), (2300, False, 3), (2300, True, 8), (3100, True, 2)])
def test_gather_exchange_mirror_assembles_the_table(pkg, G, ties, world):
    """pack_units / expand_units (the numpy mirror of x_pack / x_expand_fwd / x_expand_mirror): every shard packs the forward
    rectangles of its own work units, all packs go to everybody, and every shard ends with the complete table -- uneven
    unit counts, more shards than units, units wider than the table."""
    import sharding_mirror as sh
    rng = np.random.default_rng(G + world)
    code = rng.integers(0, 9, size=(G, G)).astype(np.uint8)
    iu = np.triu_indices(G, 1)
    code[iu[1], iu[0]] = (2 - code[iu] // 3) * 3 + (2 - code[iu] % 3)   # the pair seen from the other gene: low <-> high on both sides
    np.fill_diagonal(code, 255)
    slots = 64
    whole = sh.class_planes(code)
    parts = [np.ascontiguousarray(sh.class_planes(code, sh.owned_pair_mask(G, slots, ties, r, world))) for r in range(world)]
    assert sum(int(p.sum()) for p in parts) == int(whole.sum())
    packs = np.stack([sh.pack_units(parts[r], G, slots, ties, r, world) for r in range(world)])
    for r in range(world):
        t = parts[r].copy()
        sh.expand_units(t, packs, G, slots, ties, r, world)
        assert np.array_equal(t, whole), (r, world)


@pytest.mark.parametrize("G,ties,world,waves", [(5200, False, 3, 4), (5200, True, 2, 4), (4100, False, 2, 8), (3100, True, 3, 4)])
def test_pipelined_exchange_mirror_assembles_the_table_wave_by_wave(pkg, G, ties, world, waves):
    """The pipelined form of the exchange (launch_k1): units are counted in waves; wave w is packed from a table that holds the
    shard's own bits of the waves up to w -- and, because the pair kernel runs on beside the pack, ANY part of its later
    waves' bits -- gathered and OR-ed in before the later waves exist.  Every shard must end with the complete table, whatever
    the later waves had written when an earlier one was packed (bits that travel early arrive again: everything is OR-ed)."""
    import sharding_mirror as sh
    rng = np.random.default_rng(G * 7 + world)
    code = rng.integers(0, 9, size=(G, G)).astype(np.uint8)
    iu = np.triu_indices(G, 1)
    code[iu[1], iu[0]] = (2 - code[iu] // 3) * 3 + (2 - code[iu] % 3)
    np.fill_diagonal(code, 255)
    slots = 64
    whole = sh.class_planes(code)
    owner = sh.tile_owner(G, slots, ties, world)                     # per pair tile
    nwaves, mw, maxu = sh.wave_plan(G, slots, ties, world, waves)
    assert nwaves > 1
    # the bits of every shard by wave: mask of the pairs of its units counted in that wave
    units, W = sh.unit_list(G, slots, ties)
    wave_of = sh.wave_of_unit(G, slots, ties, world, waves)
    H = sh.UNIT_H * sh.TILE_I
    unit_of_pair = np.full((G, G), -1, dtype=np.int64)
    for u, (p_, r_) in enumerate(units):
        unit_of_pair[r_ * H:(r_ + 1) * H, p_ * W:(p_ + 1) * W] = u
    upper = np.triu(np.ones((G, G), dtype=bool), 1)
    unit_of_pair = np.where(upper, unit_of_pair, unit_of_pair.T)     # a pair and its mirror belong to the unit of the (i < j) pair
    tables = []
    for r in range(world):
        mine = sh.owned_pair_mask(G, slots, ties, r, world)
        assert np.array_equal(mine, (unit_of_pair % world == r) & (unit_of_pair >= 0) & ~np.eye(G, dtype=bool))
        tables.append(np.zeros_like(whole))
    for w in range(nwaves):
        packs = []
        for r in range(world):
            mine_w = (unit_of_pair >= 0) & (unit_of_pair % world == r) & (wave_of[np.maximum(unit_of_pair, 0)] == w) & ~np.eye(G, dtype=bool)
            tables[r] |= sh.class_planes(code, mine_w)               # wave w is counted ...
            later = (unit_of_pair >= 0) & (unit_of_pair % world == r) & (wave_of[np.maximum(unit_of_pair, 0)] > w) & ~np.eye(G, dtype=bool)
            seen = tables[r] | sh.class_planes(code, later & (rng.random((G, G)) < 0.3))   # ... and the pack may catch bits of the waves still running
            mc = max(0, min(mw, maxu - w * mw))                      # (the last waves can be empty: no exchange then, on any shard)
            if mc:
                packs.append(sh.pack_units(seen, G, slots, ties, r, world, m0=w * mw, mcnt=mc))
        if packs:
            packs = np.stack(packs)
            for r in range(world):
                sh.expand_units(tables[r], packs, G, slots, ties, r, world, m0=w * mw)
    for r in range(world):
        assert np.array_equal(tables[r], whole), (r, world)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, G, S, seed, out):
    import torch.distributed as dist
    import __graft_entry__ as ge
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        pkg, oracle = ge.load_pkg(), ge.load_oracle()
        import sharding_mirror as sh
        X = pkg.synth.t1_counts(G, S, seed).astype(np.float64)
        group = pkg.synth.groups(S)
        gid, lev = pkg.encode_groups(group)
        sizes = np.bincount(gid)
        thr = [oracle.threshold(int(sizes[0])), oracle.threshold(int(sizes[1]))]
        code = oracle.build_codes(X, gid, 2, 0, thr, seed)
        slots = sh.sample_slots(sizes)
        mask = sh.owned_pair_mask(G, slots, True, rank, world)
        hook = pkg.dist.allreduce_hook(None)
        assert pkg.dist.shard_of_process() == (rank, world)
        # what reo_build_pairs does with world > 1: own bits into a zeroed table, one sum over the shards
        planes = np.ascontiguousarray(sh.class_planes(code, mask))
        assert planes.sum() < sh.class_planes(code).sum()         # a proper part
        hook(planes.ctypes.data, planes.size, 0)                  # in place
        assert planes.max() == 1                                  # the shards' bits were disjoint
        whole = sh.codes_from_planes(planes)
        assert np.array_equal(whole, code)
        # the cheaper exchange, what the library's RCCL path does: all-gather of the shards' own forward rectangles,
        # mirror words derived on arrival (x_pack / x_expand_fwd / x_expand_mirror)
        import torch
        part = np.ascontiguousarray(sh.class_planes(code, mask))
        pack = torch.from_numpy(sh.pack_units(part, G, slots, True, rank, world))
        recv = [torch.empty_like(pack) for _ in range(world)]
        dist.all_gather(recv, pack)
        sh.expand_units(part, np.stack([r.numpy() for r in recv]), G, slots, True, rank, world)
        assert part.max() == 1 and np.array_equal(sh.codes_from_planes(part), code)
        # the pipelined form of it (launch_k1): one all-gather per wave of units, every shard calling it the same number of times
        nwaves, mw, maxu = sh.wave_plan(G, slots, True, world)
        part = np.ascontiguousarray(sh.class_planes(code, mask))
        for w in range(nwaves):
            mc = max(0, min(mw, maxu - w * mw))
            if mc == 0:
                continue
            pack = torch.from_numpy(sh.pack_units(part, G, slots, True, rank, world, m0=w * mw, mcnt=mc))
            recv = [torch.empty_like(pack) for _ in range(world)]
            dist.all_gather(recv, pack)
            sh.expand_units(part, np.stack([r.numpy() for r in recv]), G, slots, True, rank, world, m0=w * mw)
        assert part.max() == 1 and np.array_equal(sh.codes_from_planes(part), code)
        # ... after which tallies and the loop of src/RankCompV3.jl:396-425 run unsharded on every rank
        ref0 = pkg.synth.ref_mask(G, 200, seed)
        assert np.array_equal(sh.derive_tallies(sh.raw_counters(whole, ref0), ref0), oracle.tally(code, ref0))
        res, iters, trace = oracle.iterate(whole, ref0, 1.0, 0.05, 6, 1)
        exp, eiters, etrace = oracle.iterate(code, ref0, 1.0, 0.05, 6, 1)
        assert trace == etrace and np.array_equal(res, exp)
        out.put((rank, "ok", len(trace)))
    except Exception as e:  # pragma: no cover
        import traceback
        out.put((rank, "fail", traceback.format_exc()))
    finally:
        dist.destroy_process_group()


def test_gloo_world2_sharded_iteration_matches_unsharded():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, 1400, 24, 0x5EED0004, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert all(g[1] == "ok" for g in got), got
    assert got[0][2] == got[1][2] >= 1


def _watch_worker(rank, world, port, outdir):
    """rank 1 fails before its first collective; rank 0 is already inside one (a barrier that will never complete)"""
    import json
    import torch.distributed as dist
    import __graft_entry__ as ge
    pkg = ge.load_pkg()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)

    def peer_failed(r, msg):
        if rank == 0:
            with open(os.path.join(outdir, "rank0_line.json"), "w") as f:
                json.dump({"value": None, "error": f"rank {r} failed: {msg}", "failed_rank": r}, f)

    watch = pkg.dist.RankWatch(dist.distributed_c10d._get_default_store(), rank, world, on_peer_failure=peer_failed, poll_s=0.05, grace_s=1.5)
    try:
        if rank == 1:
            raise RuntimeError("injected failure on rank 1")
        dist.barrier()          # rank 1 never arrives
        open(os.path.join(outdir, "rank0_passed_the_barrier"), "w").close()
    except RuntimeError as e:
        watch.report(repr(e))
        os._exit(7)


def test_a_failing_rank_surfaces_in_rank_zeros_line_not_as_a_timeout(tmp_path):
    """bench.py --gpus N (round 6): a rank that fails between two collectives used to leave its peers waiting inside the next one
    until the launcher's timeout.  dist.RankWatch: the failing rank sets a key in the rendezvous store and waits a grace period;
    rank 0's watch thread sees it, writes the line with an `error` field and ends the process with a non-zero code."""
    import json
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    port = _free_port()
    procs = [ctx.Process(target=_watch_worker, args=(r, 2, port, str(tmp_path))) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=120)
    assert [p.exitcode for p in procs] == [4, 7], [p.exitcode for p in procs]
    assert not os.path.exists(tmp_path / "rank0_passed_the_barrier")
    line = json.load(open(tmp_path / "rank0_line.json"))
    assert line["failed_rank"] == 1 and "injected failure on rank 1" in line["error"] and line["value"] is None


def test_mirror_geometry_of_the_three_kernel_forms(pkg):
    """launch_k1's unit geometry depends on which pair kernel it selects (ADVICE, round 2): the wave form (two groups, and the
    shared per-group counts of more than two) uses kRJ genes per lane for both data families, the workgroup form kRJ /
    kRJTies, the wide form (S > 65535) kRJWide / kRJWideTies.  The mirror's constants against the sources, and the chunk widths that follow."""
    src = open(os.path.join(ROOT, "rankcompv3.jl_amd", "csrc", "kernels.hip")).read()
    m = re.search(r"constexpr int kRJWide = (\d+), kRJWideTies = (\d+);", src)
    import sharding_mirror as sh
    assert (int(m.group(1)), int(m.group(2))) == (sh.RJ_WIDE, sh.RJ_WIDE_TIES)
    assert "const int RJ = (wave || wcounts || wmulti) ? kRJ : (wide ? (c->has_ties ? kRJWideTies : kRJWide) : (c->has_ties ? kRJTies : kRJ));" in src
    for ties in (False, True):
        assert sh.geometry(5000, 208, ties, "wave")[1] == sh.TILE_J * sh.RJ
        assert sh.geometry(5000, 208, ties, "wg")[1] == sh.TILE_J * (sh.RJ_TIES if ties else sh.RJ)
        assert sh.geometry(300, 66112, ties, "wide")[1] == sh.TILE_J * (sh.RJ_WIDE_TIES if ties else sh.RJ_WIDE)
    assert sh.geometry(5000, 208, True)[1] == sh.geometry(5000, 208, True, "wave")[1]   # the default form
    # every form still partitions the tiles
    for form in ("wave", "wg", "wide"):
        owner = sh.tile_owner(3000, 64, True, 3, form)
        Gp, CJ, Q = sh.geometry(3000, 64, True, form)
        it, jc = np.meshgrid(np.arange(owner.shape[0]), np.arange(owner.shape[1]), indexing="ij")
        assert ((owner == -1) == ((jc * CJ + CJ - 1) // 64 < (it * sh.TILE_I) // 64)).all()


def test_generated_count_loops_are_current(pkg):
    """csrc/k1_loop_gen.inc is committed next to its generator: it must be what gen_k1_loop.py prints."""
    import subprocess, sys
    csrc = os.path.join(ROOT, "rankcompv3.jl_amd", "csrc")
    out = subprocess.run([sys.executable, os.path.join(csrc, "gen_k1_loop.py")], capture_output=True, text=True, check=True).stdout
    assert out == open(os.path.join(csrc, "k1_loop_gen.inc")).read()


def test_digest_of_table_and_trace_sees_every_difference(pkg):
    """bench.py --gpus N refuses to time a sharded run whose class table, trace or tallies differ from the unsharded build's: the
    64-bit digest it compares (dist.table_trace_digest) must change when ONE pair's class, one trace entry or one tally changes, must
    not depend on how the table is read in row blocks, and must be the same for equal inputs.  (A stand-in context: no GPU.)"""
    rng = np.random.default_rng(5)
    G = 300
    codes = rng.integers(0, 9, size=(G, G)).astype(np.uint8)
    res = np.zeros((G, 15)); res[:, 2:11] = rng.integers(0, 50, size=(G, 9))
    trace = [(int(a), G - int(a)) for a in rng.integers(0, 40, size=8)]

    class Ctx:
        def __init__(self, codes, res, trace): self.codes, self.res, self.trace = codes, res, trace
        def get_codes(self, i0, i1, j0, j1): return self.codes[i0:i1, j0:j1]
        def identify_degs(self, ref0, pval_deg, padj_deg, n_iter, n_conv): return self.res, len(self.trace), self.trace

    ref0 = np.ones(G, dtype=bool)
    base = pkg.dist.table_trace_digest(Ctx(codes, res, trace), G, ref0)
    assert base == pkg.dist.table_trace_digest(Ctx(codes.copy(), res.copy(), list(trace)), G, ref0) and len(base) == 16
    assert base == pkg.dist.table_trace_digest(Ctx(codes, res, trace), G, ref0, rows_per_call=37)       # block size of the read: irrelevant
    c2 = codes.copy(); c2[G - 1, 0] ^= 1
    r2 = res.copy(); r2[G // 2, 10] += 1
    t2 = list(trace); t2[-1] = (t2[-1][0] + 1, t2[-1][1] - 1)
    others = {pkg.dist.table_trace_digest(Ctx(c2, res, trace), G, ref0), pkg.dist.table_trace_digest(Ctx(codes, r2, trace), G, ref0),
              pkg.dist.table_trace_digest(Ctx(codes, res, t2), G, ref0), pkg.dist.table_trace_digest(Ctx(codes, res, trace[:-1]), G, ref0)}
    assert base not in others and len(others) == 4
