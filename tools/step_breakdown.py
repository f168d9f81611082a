"""Host wall time of each call of a bench step (matrix resident in HBM), averaged; and the same step end to end."""
import sys, time, numpy as np
sys.path.insert(0, '.')
import __graft_entry__ as ge
pkg = ge.load_pkg()
import torch
G, S, seed = 20000, 1000, 0x5EED0003
X = pkg.synth.t0_ranks(G, S, seed)
Xd = torch.from_numpy(np.ascontiguousarray(X.T)).to("cuda:0")
gid, lev = pkg.encode_groups(np.asarray(pkg.synth.groups(S))); ref0 = pkg.synth.ref_mask(G, 3000, seed)
n_iter = int(sys.argv[1]) if len(sys.argv) > 1 else 128
with pkg.Context(device=0, seed=seed) as ctx:
    names = ["set_matrix_device", "set_groups", "compute_thresholds", "build_pairs", "identify_degs"]
    for prof in (False, True, False, True):   # (the stage timers of bench.py: HIP events around every stage, read at the end of a call)
        ctx.set_profiling(prof)
        acc = np.zeros(len(names)); reps = 20
        for rep in range(reps + 3):
            t = [time.perf_counter()]
            ctx.set_matrix_device(Xd.data_ptr(), G, S, G, "i64"); t.append(time.perf_counter())
            ctx.set_groups(gid, len(lev)); t.append(time.perf_counter())
            ctx.compute_thresholds(0.01); t.append(time.perf_counter())
            ctx.build_pairs(0); t.append(time.perf_counter())
            ctx.identify_degs(ref0, 1.0, 0.05, n_iter, 0); t.append(time.perf_counter())
            if rep >= 3: acc += np.diff(t)
        print("stage timers %s, per call (ms): " % ("on " if prof else "off") + ", ".join("%s %.3f" % (n, v / reps * 1e3) for n, v in zip(names, acc)) + " | step %.3f" % (acc.sum() / reps * 1e3), flush=True)
    ctx.set_profiling(True); ctx.reset_timings()
    for rep in range(5):
        ctx.set_matrix_device(Xd.data_ptr(), G, S, G, "i64"); ctx.set_groups(gid, len(lev)); ctx.compute_thresholds(0.01); ctx.build_pairs(0)
        ctx.identify_degs(ref0, 1.0, 0.05, n_iter, 0)
    tm = ctx.timings()
    print("device stage timers per step (ms):", {k: round(v / 5, 3) for k, v in tm.items() if k.endswith("_ms")})
