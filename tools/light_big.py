"""Pass time above 65 535 genes: forced passes with the light passes on (two-launch form) and off (sorting passes only).
usage: python3 tools/light_big.py [G S family n_iter]"""
import os, sys, time, numpy as np
sys.path.insert(0, ".")
import __graft_entry__ as ge
pkg = ge.load_pkg()
G, S, fam, n_iter = (int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], int(sys.argv[4])) if len(sys.argv) > 4 else (70000, 24, "t1", 64)
seed = 0x5EED0081
X = {"t0": pkg.synth.t0_ranks, "t1": pkg.synth.t1_counts}[fam](G, S, seed)
gid, _ = pkg.encode_groups(np.asarray(pkg.synth.groups(S))); ref0 = pkg.synth.ref_mask(G, 3000, seed)
out = {}
for mode in ("1", "0"):
    os.environ["REO_LIGHT"] = mode
    with pkg.Context(device=0, seed=seed) as ctx:
        ctx.set_profiling(True)
        ctx.set_matrix(X); ctx.set_groups(gid, 2); ctx.compute_thresholds(0.01); ctx.build_pairs(0)
        for rep in range(3):
            ctx.reset_timings()
            t0 = time.perf_counter()
            res, iters, trace = ctx.identify_degs(ref0, 1.0, 0.05, n_iter, 0)
            dt = time.perf_counter() - t0
        tm = ctx.timings()
        out[mode] = (res, iters, trace)
        print("REO_LIGHT=%s  %d x %d %s: %d passes in %.2f ms (device %.2f ms) = %.1f us per pass; table scans %d; last trace %s" %
              (mode, G, S, fam, iters, dt * 1e3, tm["iter_ms"], tm["iter_ms"] / max(iters, 1) * 1e3, tm["k2_full_launches"], trace[-1]), flush=True)
r1, r0 = out["1"], out["0"]
assert r1[1] == r0[1] and r1[2] == r0[2] and np.array_equal(r1[0][:, 2:11], r0[0][:, 2:11]) and np.allclose(r1[0][:, :2], r0[0][:, :2], rtol=0, atol=1e-9)
print("light == sorting: same trace, tallies, p-values")
