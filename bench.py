#!/usr/bin/env python3
"""Headline benchmark: gene-pair·sample comparisons/s of the REO hot path.

One "step" = one full pass of the hot path over the BASELINE.json config-3 workload (synthetic 20,000 genes x 1,000
samples, tie-free T0 family, 2 groups, 3,000 initial reference genes, n_iter = 128): rank/band transform + pair kernel
K1 + 128 iteration passes (tallies + statistics), with the expression matrix already resident in HBM when the timed
region starts and the G x 15 result copied back to the host inside it.  The primary `value` uses n_conv = 0, which
makes the convergence test of src/RankCompV3.jl:419 never true, i.e. exactly 128 passes (deterministic, worst case);
the run with the reference's default n_conv = 5 is reported beside it as `converged`, and the tie-rich family (count
data: two comparisons per pair and sample) as `tie_rich`.

More blocks ride on the same line: `float64` (the same shape as Float64 input: the reference's other eltype, ranked
through the 0.1 tie band), `config4` (BASELINE config 4, 30 000 x 4 000 -- the shape BASELINE.json names for the
1/2/4/8-GPU scaling -- forced and converging run, K1 roofline, per-rank stage times when N > 1), `cycle_watch` (the
library's default: whole periods of a loop that repeats itself are skipped) and, since round 5, `from_host`: the DROP-IN
call, whose matrix starts in pageable host memory as the Julia shim hands it over (upload alone, the pipelined call
sequence, the sequence of rounds 1-4, a context per call) for config 3 as Int64 and Float64 and for config 4.  `value`
never includes an upload.  `cpu_baseline` times the reference-faithful C restatement on the WHOLE config-3 workload
(about 40 s on the GPU box's 128 threads) and checks its trace against the GPU's.
With N > 1 nothing is timed before every rank's sharded build of a small problem has reproduced rank 0's unsharded
class table, trace and tallies (`sharded_equals_unsharded`; exit status 3 otherwise).

`dtype` is "u16": the pair kernel compares 16-bit sorted positions, bit-sliced over 32-sample blocks (v_bitop3_b32
borrow chains + v_bcnt_u32_b32); the tallies are integer popcounts and the per-gene statistics fp64.
`scaling` is "strong": with N GPUs the same 20k x 1k problem is split over the ranks (pair tiles; one RCCL all-gather
of the ranks' own class-table words inside reo_build_pairs, in-library: reo_comm_init_rank).

python bench.py [--gpus N --steps K --warmup W].  With N > 1 and no WORLD_SIZE in the environment the script launches
itself under torch.distributed.run (one rank per GPU, rendezvous on 127.0.0.1) BEFORE anything touches the GPU, passes
rank 0's JSON line through and exits with the launcher's status; under an external torch.distributed.run it is a rank.
"""
from __future__ import annotations

import argparse
import gc
import json
import os
import shutil
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

HBM_PEAK = 8.0e12          # B/s, MI355X spec (MI355X_MICROARCH.md: 8 TB/s; 6.29 TB/s measured copy)
CLOCK = 2.4e9              # Hz, the nominal shader clock the issue peak is priced at (the chip holds about 2.0 GHz under this load)
SIMDS = 256 * 4


def valu_peak(nbits: int, ties: bool) -> float:
    """Issue floor of the bit-sliced pair loop: per 32 samples of one pair and lane, `nbits` v_bitop3_b32 at 2 cycles
    (full rate, three VGPR sources) + one v_bcnt_u32_b32 at 4 cycles (half rate), twice that with ties (lo and hi);
    64 lanes per wave instruction, 4 SIMDs per CU, 256 CUs.  Measured rates: tools/microbench_bitop.hip."""
    cycles = (2 * nbits + 4) * (2 if ties else 1)
    return SIMDS * CLOCK * (64 * 32) / cycles


# What the instructions of that loop cost on this chip, measured with the kernel's own clock at K1's launch shape (one wave per
# workgroup, three per SIMD; tools/microbench_issue.hip -> profiles/r6_j_microbench_issue.txt): a VALU instruction with THREE register
# operands (v_bitop3_b32, v_fma_f32, v_fmac_f32 alike) issues at 1.48 x the rate of a two-operand one, not at the same rate.
ISSUE_CYCLES = {"two_operand_vop2": 1.73, "v_bitop3_b32": 2.56, "v_bcnt_u32_b32": 2.78, "v_add_u32_sdwa": 2.78}


def valu_peak_measured(nbits: int, ties: bool, clock: float = CLOCK) -> float:
    """The same floor priced with the MEASURED issue costs: nbits v_bitop3_b32 + one v_bcnt_u32_b32 + half a v_add_u32_sdwa (the odd
    rows' second accumulate) per 2 048 comparisons."""
    cycles = (nbits * ISSUE_CYCLES["v_bitop3_b32"] + ISSUE_CYCLES["v_bcnt_u32_b32"] + 0.5 * ISSUE_CYCLES["v_add_u32_sdwa"]) * (2 if ties else 1)
    return SIMDS * clock * (64 * 32) / cycles


def plane_bits(G: int) -> int:
    return 12 if G <= 4095 else (15 if G <= 32767 else 16)


def cpu_model() -> str:
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--genes", type=int, default=20000)
    ap.add_argument("--samples", type=int, default=1000)
    ap.add_argument("--family", default="t0", choices=["t0", "t1"])
    ap.add_argument("--n-iter", type=int, default=128)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-tie-rich", action="store_true")
    ap.add_argument("--no-float64", action="store_true", help="skip the Float64-input block")
    ap.add_argument("--no-config4", action="store_true", help="skip the BASELINE config 4 block (30 000 x 4 000)")
    ap.add_argument("--no-cycle-watch", action="store_true", help="skip the blocks that time the library's default (cycle watch on)")
    ap.add_argument("--no-from-host", action="store_true", help="skip the from_host block (the drop-in call: the matrix starts in pageable host memory)")
    ap.add_argument("--cpu-genes", type=int, default=8000, help="R1 sample size when the host has too few cores for the whole workload")
    ap.add_argument("--debug-gloo-one-gpu", action="store_true",
                    help="debug only: every rank uses cuda:0 and the table exchange goes through gloo via the host")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: become the launcher.  Nothing here has imported torch or made a HIP call, and
        # the ranks are CHILD processes (never an exec of a process that has initialised the GPU).
        import socket
        with socket.socket() as so:
            so.bind(("127.0.0.1", 0))
            port = so.getsockname()[1]
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.run(cmd, env=env).returncode)

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    if args.debug_gloo_one_gpu:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    # RCCL and gloo print banners to stdout when a communicator comes up; stdout carries the one JSON line only
    class _StdoutToStderr:
        def __enter__(self):
            sys.stdout.flush()
            self.saved = os.dup(1)
            os.dup2(2, 1)
        def __exit__(self, *exc):
            sys.stdout.flush()
            os.dup2(self.saved, 1)
            os.close(self.saved)

    force_comm = bool(os.environ.get("REO_BENCH_FORCE_COMM"))  # rehearse the N > 1 plumbing with a world of one rank
    if world > 1 or force_comm:
        with _StdoutToStderr():
            if args.debug_gloo_one_gpu:
                dist.init_process_group("gloo")
            else:
                dist.init_process_group("nccl", device_id=dev)
            dist.barrier()

    pkg = ge.load_pkg()
    G, S, seed = args.genes, args.samples, 0x5EED0003
    watch = None
    if world > 1:
        # a rank that fails between two collectives must not turn into a launcher timeout: its message reaches rank 0 through the
        # rendezvous store, rank 0 prints the line with an `error` field and every rank exits non-zero (dist.RankWatch)
        def peer_failed(r, msg):
            if rank == 0:
                print(json.dumps({"metric": "gene-pair·sample comparisons/sec at 20k genes × 1k samples", "value": None, "n_gpus": world,
                                  "error": f"rank {r} failed: {msg}", "failed_rank": r}), flush=True)
        try:
            watch = pkg.dist.RankWatch(dist.distributed_c10d._get_default_store(), rank, world, on_peer_failure=peer_failed)
        except Exception as e:   # (no such store in this torch: the run goes on without the side channel, as before round 6)
            print(f"bench.py: no failure watch ({e!r})", file=sys.stderr)
            watch = None
    try:
        run(args, pkg, torch, dist, rank, world, local, dev, force_comm, _StdoutToStderr, G, S, seed)
    except BaseException as e:
        if isinstance(e, SystemExit):   # (exit 3, the sharded table differs from the unsharded one: every rank leaves together, rank 0 has printed its line)
            raise
        if watch is not None:
            if rank == 0:
                print(json.dumps({"metric": "gene-pair·sample comparisons/sec at 20k genes × 1k samples", "value": None, "n_gpus": world,
                                  "error": f"rank 0 failed: {e!r}", "failed_rank": 0}), flush=True)
            watch.report(repr(e))
        raise
    finally:
        if watch is not None:
            watch.stop()


def run(args, pkg, torch, dist, rank, world, local, dev, force_comm, _StdoutToStderr, G, S, seed):
    if os.environ.get("REO_BENCH_FAIL_RANK") == str(rank):   # rehearsal of the failure path (tests, tools/final_r6.sh)
        raise RuntimeError("REO_BENCH_FAIL_RANK: a failure injected on this rank before its first collective")
    group = pkg.synth.groups(S)
    gid, lev = pkg.encode_groups(group)
    ref0 = pkg.synth.ref_mask(G, 3000, seed)              # ref_gene_max = 3000

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def make_ctx():
        ctx = pkg.Context(device=local, seed=seed)
        ctx.set_profiling(True)
        if world > 1 or force_comm:
            if args.debug_gloo_one_gpu:
                ctx.set_shard(rank, world)
                ctx.set_allgather(pkg.dist.allgather_hook(dev, via_host=True))  # the protocol of the RCCL path, carried by gloo
            else:  # in-library RCCL: rank 0 makes the id, torch.distributed only carries its 128 bytes
                with _StdoutToStderr():
                    box = [pkg._ffi.comm_unique_id() if rank == 0 else None]
                    dist.broadcast_object_list(box, src=0)
                    ctx.comm_init_rank(box[0], rank, world)
        return ctx

    sharded_ok = None
    if world > 1 or force_comm:
        # The first run with more than one rank must not produce a number from a wrong table: before anything is timed every rank
        # builds a small problem sharded (the exchange that the timed steps use), rank 0 builds it unsharded as well, and the hashes
        # of the whole class table + an 8-pass trace + the tallies must agree across ranks and with the unsharded run.
        Gs_, Ss_, seeds_ = 9000, 64, 0x5EED00C4
        Xs_ = pkg.synth.t1_counts(Gs_, Ss_, seeds_)
        gids_, levs_ = pkg.encode_groups(pkg.synth.groups(Ss_))
        refs_ = pkg.synth.ref_mask(Gs_, 2000, seeds_)
        def small(ctx_):
            ctx_.set_matrix(Xs_); ctx_.set_groups(gids_, len(levs_)); ctx_.compute_thresholds(0.01); ctx_.build_pairs(0)
            return pkg.dist.table_trace_digest(ctx_, Gs_, refs_)   # 64-bit digest of every pair's class + 8-pass trace + tallies
        ctxs_ = make_ctx()
        mine_ = small(ctxs_)
        ctxs_.close()
        if rank == 0:
            ctxu_ = pkg.Context(device=local, seed=seed)
            whole_ = small(ctxu_)
            ctxu_.close()
        else:
            whole_ = None
        allh_ = [None] * world
        dist.all_gather_object(allh_, (mine_, whole_))
        sharded_ok = all(h[0] == allh_[0][1] for h in allh_)
        if not sharded_ok:
            if rank == 0:
                print(json.dumps({"metric": "gene-pair·sample comparisons/sec at 20k genes × 1k samples", "value": None, "n_gpus": world,
                                  "sharded_equals_unsharded": False, "hashes_sharded_by_rank": [h[0] for h in allh_], "hash_unsharded": allh_[0][1],
                                  "error": "the sharded class table / trace differs from the unsharded one: nothing was timed"}))
            dist.destroy_process_group()
            raise SystemExit(3)

    step_walls = {}   # per-step wall times of every timed leg (ms): the blocks report their median beside the mean

    def run_family(family: str, steps: int, warmup: int, n_conv_list, G=G, S=S, seed=seed, gid=gid, ref0=ref0, cycle="0"):
        # cycle="0": every pass of the loop is executed (REO_CYCLE=0: the headline and every block that earlier rounds reported);
        # "1": the library's default -- a reference set that returns lets the call skip whole periods (block "cycle_watch")
        os.environ["REO_CYCLE"] = cycle   # (read by reo_create)
        gen = {"t0": pkg.synth.t0_ranks, "t1": pkg.synth.t1_counts, "float": pkg.synth.float_expr}[family]
        X = gen(G, S, seed)                                   # Int64, like Matrix(df_expr) of count data (Float64: log-like expression)
        Xd = torch.from_numpy(np.ascontiguousarray(X.T)).to(dev)  # (S, G) row-major == G x S column-major, ld = G
        torch.cuda.synchronize()
        ctx = make_ctx()
        dt_name = "f64" if X.dtype == np.float64 else "i64"

        def step(n_conv: int):
            if os.environ.get("REO_BENCH_DEBUG_WALLS"):   # where inside a step the wall time goes (stderr)
                t = [time.perf_counter()]
                ctx.set_matrix_device(Xd.data_ptr(), G, S, G, dt_name); t.append(time.perf_counter())
                ctx.set_groups(gid, len(lev)); ctx.compute_thresholds(0.01); t.append(time.perf_counter())
                ctx.build_pairs(0); t.append(time.perf_counter())
                r = ctx.identify_degs(ref0, 1.0, 0.05, args.n_iter, n_conv); t.append(time.perf_counter())
                print(family, "set_matrix %.2f groups+thresholds %.2f build_pairs %.2f identify_degs %.2f ms" % tuple((b - a) * 1e3 for a, b in zip(t[:-1], t[1:])), file=sys.stderr)
                return r
            ctx.set_matrix_device(Xd.data_ptr(), G, S, G, dt_name)
            ctx.set_groups(gid, len(lev))
            ctx.compute_thresholds(0.01)
            ctx.build_pairs(0)
            return ctx.identify_degs(ref0, 1.0, 0.05, args.n_iter, n_conv)

        def timed(n_conv: int, steps: int):
            ctx.reset_timings()
            # the interpreter's cyclic garbage collector is off inside a timed region, as in the standard library's timeit: a
            # generation-2 collection walks every container object torch has created (40-60 ms, measured: one step of the Float64 leg
            # took 52 ms instead of 8.9 -- inside the Python stub, not in the library: REO_DEBUG_PASSES shows the library's waits)
            gc.collect()
            gc.disable()
            barrier()
            t0 = time.perf_counter()
            walls = []
            for _ in range(steps):
                ts = time.perf_counter()
                res, iters, trace = step(n_conv)
                walls.append((time.perf_counter() - ts) * 1e3)   # (a step ends with a blocking copy of the result: its wall time is the step's)
            barrier()
            dt = time.perf_counter() - t0
            gc.enable()
            step_walls[(family, n_conv, cycle, G)] = walls
            if world > 1:
                t = torch.tensor([dt], dtype=torch.float64, device="cpu" if args.debug_gloo_one_gpu else dev)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                dt = float(t.item())
            return dt, iters, trace, ctx.timings(), res

        for _ in range(warmup):
            step(n_conv_list[0])
        out = [timed(nc, steps) for nc in n_conv_list]
        info = ctx.info()
        ctx.close()
        return X, out, info

    X, (forced, conv), info = run_family(args.family, args.steps, args.warmup, [0, 5])
    dt, iters, trace, tm, res = forced
    dtc, iters_c, trace_c, tmc, _ = conv

    P = G * (G - 1) // 2
    units = P * S                                          # comparisons per step (whole job, all ranks together)
    value = units * args.steps / dt
    share = info["tiles_owned"] / max(info["tiles_total"], 1)

    def k1_roofline(tm_, info_, ties: bool, G=G, S=S):
        k1_ms = tm_["k1_ms"] / max(tm_["k1_launches"], 1)
        share_ = info_["tiles_owned"] / max(info_["tiles_total"], 1)
        # algorithmic bytes: K1 reads G*S*2 B of 16-bit positions and writes the 4-bit class table
        k1_bytes = (G * S * 2 + G * G / 2) * share_
        peak = valu_peak(plane_bits(G), ties)
        rate = (G * (G - 1) // 2) * S * share_ / (k1_ms * 1e-3)
        return {"bound": "valu", "achieved": rate / 1e12, "peak": peak / 1e12, "unit": "Tcmp/s", "frac": rate / peak,
                "peak_definition": "256 CU x 4 SIMD x 2.4 GHz x 2048 comparisons per (2*bits+4)%s cycles, bits=%d" % (" x 2" if ties else "", plane_bits(G)),
                "hbm_achieved_GBps": k1_bytes / (k1_ms * 1e-3) / 1e9, "hbm_frac": k1_bytes / (k1_ms * 1e-3) / HBM_PEAK,
                "ms_per_launch": k1_ms, "traffic": None,
                "frac_of_measured_issue_floor": rate / valu_peak_measured(plane_bits(G), ties),
                "measured_issue_floor_note": "the loop's own instructions at the issue costs measured on this chip (shader-clock cycles per wave instruction "
                                             "and SIMD: %s; profiles/r6_j_microbench_issue.txt), still at the nominal 2.4 GHz; the chip holds about 2.0 GHz in this "
                                             "kernel (GRBM_GUI_ACTIVE / duration, profiles/r6_h_k1_sq_counters_config3.csv): x 1.2 again" % json.dumps(ISSUE_CYCLES),
                "kernel": ("k1w_pairs<%d,%s>" if os.environ.get("REO_K1_WAVE", "1") != "0" else "k1_pairs<%d,%s,false>") % (plane_bits(G), "true" if ties else "false")}

    k1 = k1_roofline(tm, info, bool(info["has_ties"]))
    k2_ms = tm["k2_full_ms"] / max(tm["k2_full_launches"], 1)   # whole-table scans only
    # K2 streams the class table + mask and writes int32[8] per gene
    k2_bytes = G * G / 2 + G / 8 + 36 * G
    k2 = {"bound": "hbm", "achieved": k2_bytes / (k2_ms * 1e-3) / 1e9 if k2_ms else None, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
          "frac": k2_bytes / (k2_ms * 1e-3) / HBM_PEAK if k2_ms else None, "ms_per_launch": k2_ms, "launches": tm["k2_full_launches"],
          "traffic": None, "kernel": "k2_tally"}
    pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(pmc):
        with open(pmc) as f:
            tr = json.load(f)
        k1["traffic"], k2["traffic"] = tr.get("k1_pairs"), tr.get("k2_tally")
        k1["traffic_source"], k2["traffic_source"] = tr.get("k1_pairs_source"), tr.get("k2_tally_source")
        traffic_tie = (tr.get("k1_pairs_tie_rich"), tr.get("k1_pairs_tie_rich_source"))
    else:
        traffic_tie = (None, None)

    out = {
        "metric": "gene-pair·sample comparisons/sec at 20k genes × 1k samples",
        "value": value, "unit": "comparisons/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
        "scaling": "strong", "vs_baseline": None, "dtype": "u16",
        "dtype_note": "pair kernel: 16-bit sorted positions as bit planes (exact); tallies: integer popcounts; statistics: fp64",
        "data": "synthetic",
        "config": {"workload": f"BASELINE config 3: synthetic {G} genes x {S} samples ({args.family.upper()} family, Int64 input), "
                               f"2 groups, ref_gene_max=3000, n_iter={args.n_iter}, n_conv=0 (exactly {iters} iterations)",
                   "genes": G, "samples": S, "iterations": iters, "sharding": f"pair tiles over {world} GPU(s)",
                   "cycle_watch": "off for this line (REO_CYCLE=0): all %d passes are executed; the library's default is in the block cycle_watch" % iters},
        "converged": {"value": units * args.steps / dtc, "ms_per_step": dtc / args.steps * 1e3,
                      "n_conv": 5, "iterations": iters_c, "final_trace": list(trace_c[-1]) if trace_c else None},
        "stages_ms_per_step": {k: tm[k] / args.steps for k in ("transform_ms", "k1_ms", "k2_full_ms", "iter_ms", "exchange_ms")},
        "roofline": k1, "roofline_k1": k1, "roofline_k2": k2,
        "final_trace": list(trace[-1]) if trace else None,
        "has_ties": info["has_ties"],
    }
    out["step_walls_ms"] = {"%s n_conv=%d cycle=%s G=%d" % k: [round(w, 3) for w in v] for k, v in step_walls.items()}
    out["stages_ms_per_step"]["iteration_passes_us_each"] = tm["iter_ms"] / args.steps / max(iters, 1) * 1e3

    if world > 1 or force_comm:  # what each rank did (the driver computes the scaling efficiency from the per-N values itself)
        mine = {"rank": rank, "k1_ms": tm["k1_ms"] / args.steps, "tiles_owned": info["tiles_owned"], "tiles_total": info["tiles_total"],
                "exchange_ms_per_build": tm["exchange_ms"] / args.steps, "transform_ms": tm["transform_ms"] / args.steps,
                "iter_ms": tm["iter_ms"] / args.steps}
        allr = [None] * world
        dist.all_gather_object(allr, mine)
        rep = (mine["transform_ms"] + mine["iter_ms"]) / (dt / args.steps * 1e3)
        out["ranks"] = allr
        out["sharded_equals_unsharded"] = sharded_ok   # checked before timing on 9000 x 64 (class table + 8-pass trace + tallies, every rank)
        out["replicated_stage_share"] = rep     # transform + iteration passes run identically on every rank
        out["collective"] = ("one ncclAllGather per reo_build_pairs, in-library RCCL: every rank sends the forward words of its own pair "
                             "tiles (about %d MB over all ranks; the whole table is %d MB), mirror words are derived on arrival; "
                             "exchange_ms_per_build = pack + collective + unpack") % (G * info["Gp"] // 4 // 1000000, G * info["Gp"] // 2 // 1000000)

    if not args.no_cycle_watch:
        # the library as it ships: the forced loop ends in a cycle of reference sets (period 4 here), the light passes find it and the
        # call skips whole periods -- the same iters_run, trace and result (compared below), fewer executed passes
        stc = max(3, args.steps // 2)
        _, (fc,), infoc = run_family(args.family, stc, 1, [0], cycle="1")
        dtw, itw, trw, tmw, resw = fc
        out["cycle_watch"] = {"workload": "the headline workload with the cycle watch on (the default)", "ms_per_step": dtw / stc * 1e3, "value": units * stc / dtw,
                              "steps": stc, "iterations_reported": itw, "period": infoc["cycle_period"], "found_in_front_of_pass": infoc["cycle_found_at_pass"],
                              "passes_skipped": infoc["cycle_passes_skipped"], "iter_ms": tmw["iter_ms"] / stc,
                              "same_trace_as_all_passes_executed": bool(trw == trace and itw == iters),
                              "same_result_as_all_passes_executed": bool(np.array_equal(resw, res, equal_nan=True))}

    if not args.no_tie_rich and args.family == "t0":
        st = max(3, args.steps // 4)
        _, (f1,), info1 = run_family("t1", st, 1, [0])
        dt1, it1, tr1, tm1, _ = f1
        r1 = k1_roofline(tm1, info1, True)
        out["tie_rich"] = {"workload": "T1 family (zero-inflated counts, ~2 % tied cells): lo and hi band edges, two borrow chains per pair",
                           "ms_per_step": dt1 / st * 1e3, "value": units * st / dt1, "k1_ms": r1["ms_per_launch"], "frac": r1["frac"],
                           "peak": r1["peak"], "achieved": r1["achieved"], "kernel": r1["kernel"], "steps": st,
                           "traffic": traffic_tie[0], "traffic_source": traffic_tie[1]}

    def per_rank(tm_, info_, steps_):
        """what every rank did in a block, gathered (N > 1 only)"""
        if world == 1 and not force_comm:
            return None
        mine = {"rank": rank, "k1_ms": tm_["k1_ms"] / steps_, "tiles_owned": info_["tiles_owned"], "tiles_total": info_["tiles_total"],
                "exchange_ms_per_build": tm_["exchange_ms"] / steps_, "transform_ms": tm_["transform_ms"] / steps_, "iter_ms": tm_["iter_ms"] / steps_}
        allr = [None] * world
        dist.all_gather_object(allr, mine)
        return allr

    if not args.no_float64 and args.family == "t0":
        # the reference's other input type (Matrix(df_expr) of Float64, :652): the same shape with log-like expression values,
        # ranked through the 0.1 band of is_greater (:72) -- every pair needs both band edges (the tie-rich pair kernel)
        st = max(3, args.steps // 4)
        _, (ff,), infof = run_family("float", st, 1, [0])
        dtf, itf, trf, tmf, _ = ff
        rf = k1_roofline(tmf, infof, bool(infof["has_ties"]))
        out["float64"] = {"workload": f"synthetic {G} x {S} Float64 (log2(1 + counts) + jitter: values closer than 0.1 apart are ties, :72), "
                                      f"n_iter={args.n_iter}, n_conv=0", "ms_per_step": dtf / st * 1e3, "value": units * st / dtf, "iterations": itf,
                          "transform_ms": tmf["transform_ms"] / st, "k1_ms": rf["ms_per_launch"], "iter_ms": tmf["iter_ms"] / st,
                          "frac": rf["frac"], "kernel": rf["kernel"], "steps": st, "transform_in_lds": infof["transform_in_lds"],
                          "final_trace": list(trf[-1]) if trf else None}

    if not args.no_config4 and args.family == "t0" and (G, S) == (20000, 1000):
        # BASELINE config 4 (30 000 x 4 000, the shape BASELINE.json names for the 1/2/4/8-GPU scaling): the same step on
        # it -- forced 128 passes and the converging run -- for every N; its pair stage is 19-20 ms on one GPU, so the
        # replicated pass stage weighs a sixth of the step instead of half
        G4, S4, seed4 = 30000, 4000, 0x5EED0004
        gid4, _lev4 = pkg.encode_groups(pkg.synth.groups(S4))
        ref4 = pkg.synth.ref_mask(G4, 3000, seed4)
        st4 = max(2, args.steps // 5)
        _, (f4, c4), info4 = run_family("t0", st4, 1, [0, 5], G=G4, S=S4, seed=seed4, gid=gid4, ref0=ref4)
        dt4, it4, tr4, tm4, _ = f4
        dt4c, it4c, tr4c, tm4c, _ = c4
        units4 = (G4 * (G4 - 1) // 2) * S4
        r4 = k1_roofline(tm4, info4, bool(info4["has_ties"]), G=G4, S=S4)
        out["config4"] = {"workload": f"BASELINE config 4: synthetic {G4} genes x {S4} samples (T0 family, Int64 input), 2 groups, ref_gene_max=3000, "
                                      f"n_iter={args.n_iter}, n_conv=0 (exactly {it4} iterations); pair tiles over {world} GPU(s)",
                          "value": units4 * st4 / dt4, "unit": "comparisons/s", "ms_per_step": dt4 / st4 * 1e3, "steps": st4, "iterations": it4,
                          "converged": {"value": units4 * st4 / dt4c, "ms_per_step": dt4c / st4 * 1e3, "n_conv": 5, "iterations": it4c,
                                        "final_trace": list(tr4c[-1]) if tr4c else None},
                          "stages_ms_per_step": {k: tm4[k] / st4 for k in ("transform_ms", "k1_ms", "k2_full_ms", "iter_ms", "exchange_ms")},
                          "roofline": r4, "final_trace": list(tr4[-1]) if tr4 else None,
                          "thresholds": [pkg._ffi.threshold(S4 // 2), pkg._ffi.threshold(S4 - S4 // 2)]}
        if not args.no_cycle_watch:
            _, (w4,), infow4 = run_family("t0", st4, 1, [0], G=G4, S=S4, seed=seed4, gid=gid4, ref0=ref4, cycle="1")
            dtw4, itw4, trw4, tmw4, _ = w4
            out["config4"]["cycle_watch"] = {"ms_per_step": dtw4 / st4 * 1e3, "value": units4 * st4 / dtw4, "iter_ms": tmw4["iter_ms"] / st4,
                                             "period": infow4["cycle_period"], "found_in_front_of_pass": infow4["cycle_found_at_pass"],
                                             "passes_skipped": infow4["cycle_passes_skipped"], "same_trace_as_all_passes_executed": bool(trw4 == tr4 and itw4 == it4)}
        pr4 = per_rank(tm4, info4, st4)
        if pr4:
            out["config4"]["ranks"] = pr4
            out["config4"]["replicated_stage_share"] = (tm4["transform_ms"] + tm4["iter_ms"]) / st4 / (dt4 / st4 * 1e3)   # transform + passes: the same on every rank
            if "cycle_watch" in out["config4"]:
                out["config4"]["cycle_watch"]["replicated_stage_share"] = (tmw4["transform_ms"] + tmw4["iter_ms"]) / st4 / (dtw4 / st4 * 1e3)

    if not args.no_from_host and world == 1 and args.family == "t0" and (G, S) == (20000, 1000):
        # The DROP-IN call: what julia/RankCompV3HIP.jl (and tests/abi/abi_client.c) does at :652 -- the matrix is a pageable
        # column-major HOST array handed to reo_set_matrix_i64 / _f64.  `value` above never includes this (inputs resident in HBM);
        # this block says what it costs and how much of it the library hides behind the transform and the pair kernel (groups and
        # thresholds are set BEFORE the matrix, so reo_set_matrix can rank samples and start the pair kernel's group-1 side while
        # the rest of the matrix is still on its way).  PCIe spec of the box: 63 GB/s.
        def from_host(family, Gh, Sh, seedh, steps_h, compute_ms):
            os.environ["REO_CYCLE"] = "0"
            gen = {"t0": pkg.synth.t0_ranks, "float": pkg.synth.float_expr}[family]
            Xh = np.asfortranarray(gen(Gh, Sh, seedh))                       # column-major, pageable: what a Julia Matrix is
            gidh, levh = pkg.encode_groups(pkg.synth.groups(Sh))
            refh = pkg.synth.ref_mask(Gh, 3000, seedh)
            nbytes = Xh.nbytes
            def call(ctx_):
                ctx_.set_groups(gidh, len(levh)); ctx_.compute_thresholds(0.01); ctx_.set_matrix(Xh)
                ctx_.build_pairs(0)
                return ctx_.identify_degs(refh, 1.0, 0.05, args.n_iter, 0)
            def timed_walls(fn, n):
                gc.collect(); gc.disable()
                w = []
                for _ in range(n):
                    torch.cuda.synchronize(); t0 = time.perf_counter(); r = fn(); w.append((time.perf_counter() - t0) * 1e3)
                gc.enable()
                return w, r
            os.environ["REO_UPLOAD_THREADS"] = "0"                             # the link itself: the caller's array as it is
            plain = pkg.Context(device=local, seed=seedh)                      # (no groups set: reo_set_matrix is the upload alone)
            plain.set_matrix(Xh)
            up_w, _ = timed_walls(lambda: plain.set_matrix(Xh), max(3, steps_h))
            plain.close()
            del os.environ["REO_UPLOAD_THREADS"]
            plain = pkg.Context(device=local, seed=seedh)                      # the upload alone as the library does it (Int64: narrowed)
            plain.set_matrix(Xh)
            upn_w, _ = timed_walls(lambda: plain.set_matrix(Xh), max(3, steps_h))
            plain.close()
            os.environ["REO_EAGER_UPLOAD"] = "0"; os.environ["REO_UPLOAD_THREADS"] = "0"   # rounds 1-4: ONE copy of the caller's array, THEN transform, pair kernel, passes
            ctxs = pkg.Context(device=local, seed=seedh)
            call(ctxs)
            seq_w, rseq = timed_walls(lambda: call(ctxs), steps_h)
            ctxs.close()
            del os.environ["REO_EAGER_UPLOAD"], os.environ["REO_UPLOAD_THREADS"]
            ctxh = pkg.Context(device=local, seed=seedh)
            call(ctxh)
            pip_w, rpip = timed_walls(lambda: call(ctxh), steps_h)
            link_bytes = ctxh.info()["upload_link_bytes"]
            range_launches = ctxh.info()["eager_range_launches"]
            ctxh.close()
            os.environ["REO_EAGER_RANGES"] = "1"                                # round 5's form on the same box: a side of the pair kernel waits for its whole group
            ctxw = pkg.Context(device=local, seed=seedh)
            call(ctxw)
            whole_w, rwhole = timed_walls(lambda: call(ctxw), steps_h)
            ctxw.close()
            del os.environ["REO_EAGER_RANGES"]
            def whole_call():                                                  # a context per call, as the Julia shim does it
                with pkg.Context(device=local, seed=seedh) as c_:
                    return call(c_)
            whole_call()
            all_w, _ = timed_walls(whole_call, steps_h)
            med = lambda v: float(np.median(v))
            up = med(up_w)
            return {"workload": f"{Gh} x {Sh} {'Float64' if Xh.dtype == np.float64 else 'Int64'} from a pageable column-major host array, n_iter={args.n_iter}, n_conv=0, every pass executed",
                    "upload_ms": up, "upload_GBps": nbytes / up / 1e6, "upload_frac_of_pcie_63GBps": nbytes / up / 1e6 / 63.0, "matrix_MB": nbytes / 1e6,
                    "upload_ms_as_the_library_does_it": med(upn_w),   # (Int64: 16- / 32-bit numbers on the link, widened on the device)
                    "compute_ms_device_resident": compute_ms,
                    "ms_per_step": med(pip_w), "ms_per_step_not_pipelined": med(seq_w), "ms_per_call_with_create_and_destroy": med(all_w),
                    "ms_per_step_whole_sides_as_in_round_5": med(whole_w), "pair_kernel_launches_over_ranges_of_a_side": range_launches,
                    "same_result_ranges_and_whole_sides": bool(np.array_equal(rwhole[0], rpip[0], equal_nan=True) and rwhole[2] == rpip[2]),
                    "link_MB": link_bytes / 1e6,
                    "sum_upload_compute_ms": up + compute_ms, "max_upload_compute_ms": max(up, compute_ms),
                    "ratio_to_max": med(pip_w) / max(up, compute_ms), "ratio_to_sum": med(pip_w) / (up + compute_ms),
                    "value": (Gh * (Gh - 1) // 2) * Sh / (med(pip_w) * 1e-3), "steps": steps_h, "walls_ms": [round(x, 3) for x in pip_w],
                    "same_result_pipelined_and_not": bool(np.array_equal(rseq[0], rpip[0], equal_nan=True) and rseq[2] == rpip[2])}
        sth = max(3, args.steps // 4)
        out["from_host"] = {"config3_int64": from_host("t0", G, S, seed, sth, out["ms_per_step"])}
        if "float64" in out:
            out["from_host"]["config3_float64"] = from_host("float", G, S, seed, sth, out["float64"]["ms_per_step"])
        if "config4" in out:
            out["from_host"]["config4_int64"] = from_host("t0", 30000, 4000, 0x5EED0004, 3, out["config4"]["ms_per_step"])

    out["step_walls_ms"] = {"%s n_conv=%d cycle=%s G=%d" % k: [round(w, 3) for w in v] for k, v in step_walls.items()}   # (every timed leg, per step)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        oracle = ge.load_oracle()
        ncores = oracle.num_threads()
        if ncores >= 32:
            # R1 on the WHOLE config-3 workload (about 45 s on 128 threads): no sub-sampling, and its trace must be the GPU's
            t0 = time.perf_counter()
            r1res, it1r, tr1r = oracle.identify_degs(X.astype(np.float64), gid, len(lev), 0.01, 1.0, 0.05, ref0, args.n_iter, 0, seed)
            tc = time.perf_counter() - t0
            out["cpu_baseline"] = {"value": units / tc, "unit": "comparisons/s", "cores": ncores, "kind": "port", "seconds": tc, "cpu": cpu_model(),
                                   "julia": shutil.which("julia") or "not found",
                                   "sample": f"whole workload: all {G} genes x {S} samples, full identify_degs (n_iter={args.n_iter}, n_conv=0), "
                                             "C restatement of the reference's loop nest with OpenMP",
                                   "same_trace_as_gpu": bool(tr1r == trace and it1r == iters),
                                   "same_tallies_as_gpu": bool(np.array_equal(r1res[:, 2:11], res[:, 2:11]))}
        else:
            refs_all = ref0.copy()
            samples = []
            for Gs in (args.cpu_genes // 2, args.cpu_genes):   # two sizes: the rate per comparison must not depend on G (G^2 S scaling)
                Gs = min(Gs, G)
                Xs = X[:Gs].astype(np.float64)
                refs = refs_all[:Gs].copy(); refs[:10] = True
                t0 = time.perf_counter()
                oracle.identify_degs(Xs, gid, len(lev), 0.01, 1.0, 0.05, refs, args.n_iter, 0, seed)
                tc = time.perf_counter() - t0
                samples.append({"genes": Gs, "seconds": tc, "rate": (Gs * (Gs - 1) // 2) * S / tc})
            big = samples[-1]
            out["cpu_baseline"] = {"value": big["rate"], "unit": "comparisons/s", "cores": ncores, "kind": "port",
                                   "seconds": big["seconds"], "cpu": cpu_model(), "julia": shutil.which("julia") or "not found",
                                   "sample": f"first {big['genes']} genes x {S} samples of the same matrix (fewer than 32 host threads: the whole workload "
                                             f"would take minutes), full identify_degs (n_iter={args.n_iter}, n_conv=0), C restatement of the reference's loop nest with OpenMP",
                                   "scaling_check": {"sizes": samples, "rate_ratio_big_over_small": big["rate"] / samples[0]["rate"]}}
        if hasattr(oracle, "tuned_identify_degs"):  # R2: fast enough for the whole workload, no sub-sampling
            t0 = time.perf_counter()
            r2, it2, tr2 = oracle.tuned_identify_degs(X.astype(np.float64), gid, len(lev), 0.01, 1.0, 0.05, ref0, args.n_iter, 0, seed)
            tc = time.perf_counter() - t0
            out["cpu_baseline"]["tuned"] = {"value": units / tc, "unit": "comparisons/s", "seconds": tc, "genes": G, "kind": "port (tuned)",
                                            "cores": oracle.num_threads(), "same_trace_as_gpu": tr2 == trace,
                                            "what": "R2 of SURVEY.md §8d on the WHOLE workload: per-sample rank transform, gene-major 16-bit "
                                                    "positions, SIMD compares (AVX2), bit-plane class table, popcount tallies"}
    if rank == 0:
        print(json.dumps(out))
    if world > 1 or force_comm:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
