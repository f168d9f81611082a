"""A/B of a K1 switch inside ONE process (box-to-box and run-to-run differences are larger than most effects): two contexts
created under different values of an environment variable, bench-like steps alternating between them, K1's HIP-event time.
usage: python tools/k1_ab.py ENV_VAR value_a value_b [family] [reps]"""
import os, sys, time, numpy as np
sys.path.insert(0, '.')
import __graft_entry__ as ge
pkg = ge.load_pkg()
import torch
var, va, vb = sys.argv[1], sys.argv[2], sys.argv[3]
fam = sys.argv[4] if len(sys.argv) > 4 else "t0"
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 12
G, S, seed = 20000, 1000, 0x5EED0003
X = pkg.synth.t0_ranks(G, S, seed) if fam == "t0" else pkg.synth.t1_counts(G, S, seed)
Xd = torch.from_numpy(np.ascontiguousarray(X.T)).to("cuda:0")
gid, lev = pkg.encode_groups(np.asarray(pkg.synth.groups(S))); ref0 = pkg.synth.ref_mask(G, 3000, seed)
ctxs = {}
for v in (va, vb):
    os.environ[var] = v
    ctxs[v] = pkg.Context(device=0, seed=seed)
    ctxs[v].set_profiling(True)
times = {va: [], vb: []}; walls = {va: [], vb: []}
for rep in range(reps + 2):
    for v in ((va, vb) if rep % 2 == 0 else (vb, va)):
        ctx = ctxs[v]
        os.environ[var] = v   # (switches that are read per launch)
        ctx.reset_timings()
        t0 = time.perf_counter()
        ctx.set_matrix_device(Xd.data_ptr(), G, S, G, "i64"); ctx.set_groups(gid, len(lev)); ctx.compute_thresholds(0.01)
        ctx.build_pairs(0)
        res, iters, trace = ctx.identify_degs(ref0, 1.0, 0.05, 128, 0)
        w = time.perf_counter() - t0
        if rep >= 2: times[v].append(ctx.timings()["k1_ms"]); walls[v].append(w * 1e3)
for v in (va, vb):
    t = np.array(times[v]); print("%s=%s: step wall median %.4f ms, mean %.4f" % (var, v, np.median(walls[v]), np.mean(walls[v])))
    print("%s=%s: K1 median %.4f ms, mean %.4f, min %.4f, max %.4f (%d launches), trace end %s" % (var, v, np.median(t), t.mean(), t.min(), t.max(), len(t), trace[-1]))
for c in ctxs.values(): c.close()
