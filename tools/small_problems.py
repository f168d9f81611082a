"""BASELINE config 2 (5 000 genes x 200 samples, 1 iteration) and a tiny problem: host wall per call, matrix resident."""
import sys, time, numpy as np
sys.path.insert(0, '.')
import __graft_entry__ as ge
pkg = ge.load_pkg()
import torch
for (G, S, n_iter, fam) in ((5000, 200, 1, "t1"), (5000, 200, 128, "t1"), (1000, 60, 16, "t0")):
    seed = 0x5EED0003
    X = (pkg.synth.t1_counts if fam == "t1" else pkg.synth.t0_ranks)(G, S, seed)
    Xd = torch.from_numpy(np.ascontiguousarray(X.T)).to("cuda:0")
    gid, lev = pkg.encode_groups(np.asarray(pkg.synth.groups(S))); ref0 = pkg.synth.ref_mask(G, min(3000, G // 4), seed)
    with pkg.Context(device=0, seed=seed) as ctx:
        names = ["set_matrix_device", "set_groups", "compute_thresholds", "build_pairs", "identify_degs"]
        acc = np.zeros(len(names)); reps = 30
        for rep in range(reps + 3):
            t = [time.perf_counter()]
            ctx.set_matrix_device(Xd.data_ptr(), G, S, G, "i64"); t.append(time.perf_counter())
            ctx.set_groups(gid, len(lev)); t.append(time.perf_counter())
            ctx.compute_thresholds(0.01); t.append(time.perf_counter())
            ctx.build_pairs(0); t.append(time.perf_counter())
            res, iters, trace = ctx.identify_degs(ref0, 1.0, 0.05, n_iter, 5); t.append(time.perf_counter())
            if rep >= 3: acc += np.diff(t)
        ctx.set_profiling(True); ctx.reset_timings()
        for rep in range(5):
            ctx.set_matrix_device(Xd.data_ptr(), G, S, G, "i64"); ctx.set_groups(gid, len(lev)); ctx.compute_thresholds(0.01); ctx.build_pairs(0)
            ctx.identify_degs(ref0, 1.0, 0.05, n_iter, 5)
        tm = ctx.timings()
        print("%d x %d %s n_iter %d (ran %d): " % (G, S, fam, n_iter, iters) + ", ".join("%s %.3f" % (n, v / reps * 1e3) for n, v in zip(names, acc)) + " | step %.3f ms | device: " % (acc.sum() / reps * 1e3) +
              ", ".join("%s %.3f" % (k, tm[k] / 5) for k in ("transform_ms", "k1_ms", "iter_ms")), flush=True)
