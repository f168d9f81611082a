"""BASELINE config 4 on ONE GPU (the 8-GPU sharded run is the driver's): synthetic 30,000 genes x 4,000 samples."""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import __graft_entry__ as ge

pkg = ge.load_pkg()
G, S, seed = 30000, 4000, 0x5EED0004
t = time.perf_counter()
X = pkg.synth.t0_ranks(G, S, seed)
print("generated %d x %d in %.1f s" % (G, S, time.perf_counter() - t))
group = pkg.synth.groups(S)
gid, lev = pkg.encode_groups(group)
ref0 = pkg.synth.ref_mask(G, 3000, seed)
import torch
Xd = torch.from_numpy(np.ascontiguousarray(X.T)).cuda()
with pkg.Context(device=0, seed=seed) as ctx:
    ctx.set_profiling(True)
    for rep in range(2):
        ctx.reset_timings()
        torch.cuda.synchronize()
        t = time.perf_counter()
        ctx.set_matrix_device(Xd.data_ptr(), G, S, G, "i64")
        ctx.set_groups(gid, 2)
        thr = ctx.compute_thresholds(0.01)
        ctx.build_pairs(0)
        res, iters, trace = ctx.identify_degs(ref0, 1.0, 0.05, 128, 0)  # exactly 128 passes
        dt = time.perf_counter() - t
    tm = ctx.timings()
    P = G * (G - 1) // 2
    print("thresholds %s passes %d trace %s" % (thr[:, 0].tolist(), iters, trace[-1]))
    print("step %.1f ms -> %.3e comparisons/s; K1 %.1f ms = %.3e cmp/s; transform %.1f ms; iterations %.2f ms"
          % (dt * 1e3, P * S / dt, tm["k1_ms"], P * S / (tm["k1_ms"] * 1e-3), tm["transform_ms"], tm["iter_ms"]))
    cont = ctx.tally(ref0)
    assert np.array_equal(cont.sum(axis=1), ref0.sum() - ref0.astype(np.int64))
    gt, eq = ctx.pair_counts(10, 42, 20000, 20256)
    for (i, j) in [(10, 20000), (41, 20255)]:
        for g, sl in enumerate((slice(0, 2000), slice(2000, 4000))):
            assert gt[i - 10, j - 20000, g] == int((X[i, sl] > X[j, sl]).sum())
    print("property checks ok")
