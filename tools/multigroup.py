"""One-vs-rest over C groups at 20 000 x 1 000: shared per-group counts vs recounting per comparison."""
import os, sys, time, numpy as np
sys.path.insert(0, '.')
import __graft_entry__ as ge
pkg = ge.load_pkg()
G, S, seed = 20000, 1000, 0x5EED0003
C = int(sys.argv[1]) if len(sys.argv) > 1 else 10
fam = sys.argv[2] if len(sys.argv) > 2 else "t0"
X = pkg.synth.t0_ranks(G, S, seed) if fam == "t0" else pkg.synth.t1_counts(G, S, seed)
gid = (np.arange(S) % C).astype(np.int32)
ref0 = pkg.synth.ref_mask(G, 3000, seed)
import torch
for mode in ("0", "1"):
    os.environ["REO_SHARE_GROUP_COUNTS"] = mode
    with pkg.Context(device=0, seed=seed) as ctx:
        ctx.set_profiling(True)
        ctx.set_matrix(X); ctx.set_groups(gid, C); ctx.compute_thresholds(0.01)
        per = []
        t0 = time.perf_counter()
        for k in range(C):
            ctx.reset_timings(); ctx.build_pairs(k); per.append(ctx.timings()["k1_ms"])
            res, iters, trace = ctx.identify_degs(ref0, 1.0, 0.05, 128, 5)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        info = ctx.info()
    print("%s C=%d shared=%s: total %.1f ms (%d comparisons incl. iterations), pair stage per comparison (ms): %s, counts held %.2f GB" %
          (fam, C, mode, dt * 1e3, C, " ".join("%.2f" % v for v in per), info["group_count_bytes"] / 1e9), flush=True)
