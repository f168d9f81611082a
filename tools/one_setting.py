"""The drop-in call (groups, thresholds, matrix from pageable host memory, build, 128 forced passes) with ONE context in the process --
the settings come from the environment.  (Several contexts alive in one process share HIP's four hardware queues: the streams of the
third and later contexts alias, and an A/B that alternates contexts measures that -- profiles/r6_n_numa_probe.txt.)
usage: [REO_... env] python tools/one_setting.py [t0|t1|float] G S [reps]"""
import os, sys, time, numpy as np
sys.path.insert(0, '.')
import __graft_entry__ as ge
pkg = ge.load_pkg()
import torch
fam, G, S = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 8
X = np.asfortranarray({"t0": pkg.synth.t0_ranks, "t1": pkg.synth.t1_counts, "float": pkg.synth.float_expr}[fam](G, S, 3))
gid, _ = pkg.encode_groups(np.asarray(pkg.synth.groups(S))); ref0 = pkg.synth.ref_mask(G, 3000, 3)
os.environ["REO_CYCLE"] = "0"
ctx = pkg.Context(device=0, seed=3)
w = []
for rep in range(reps + 2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    ctx.set_groups(gid, 2); ctx.compute_thresholds(0.01); ctx.set_matrix(X); ctx.build_pairs(0)
    res, it, tr = ctx.identify_degs(ref0, 1.0, 0.05, 128, 0)
    w.append((time.perf_counter() - t0) * 1e3)
tag = " ".join("%s=%s" % (k, v) for k, v in sorted(os.environ.items()) if k.startswith("REO_EAGER") or k == "GPU_MAX_HW_QUEUES")
print("%-45s %s %d x %d: median %.2f ms  %s  (range launches %d, trace end %s)" % (tag or "defaults", fam, G, S, float(np.median(w[2:])), " ".join("%.2f" % x for x in w[2:]), ctx.info()["eager_range_launches"], tr[-1]))
