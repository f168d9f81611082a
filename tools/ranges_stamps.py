import os, sys, time, numpy as np
sys.path.insert(0, '.')
import __graft_entry__ as ge
pkg = ge.load_pkg()
import torch
G, S = int(sys.argv[1]), int(sys.argv[2])
X = np.asfortranarray(pkg.synth.t0_ranks(G, S, 3))
gid, _ = pkg.encode_groups(np.asarray(pkg.synth.groups(S))); ref0 = pkg.synth.ref_mask(G, 3000, 3)
os.environ["REO_CYCLE"] = "0"
for r in sys.argv[3].split(","):
    os.environ["REO_EAGER_RANGES"] = r
    os.environ.pop("REO_DEBUG_PASSES", None)
    ctx = pkg.Context(device=0, seed=3)
    for rep in range(3):
        ctx.set_groups(gid, 2); ctx.compute_thresholds(0.01); ctx.set_matrix(X); ctx.build_pairs(0); ctx.identify_degs(ref0, 1.0, 0.05, 128, 0)
    ctx.close()
    os.environ["REO_DEBUG_PASSES"] = "1"
    ctx = pkg.Context(device=0, seed=3)
    for rep in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        print("=== ranges", r, "call", rep, file=sys.stderr, flush=True)
        ctx.set_groups(gid, 2); ctx.compute_thresholds(0.01)
        t1 = time.perf_counter(); ctx.set_matrix(X); t2 = time.perf_counter(); ctx.build_pairs(0); t3 = time.perf_counter()
        torch.cuda.synchronize(); t4 = time.perf_counter()
        ctx.identify_degs(ref0, 1.0, 0.05, 128, 0); t5 = time.perf_counter()
        print("ranges %s: groups+thr %.2f set_matrix %.2f build %.2f device idle after %.2f identify %.2f total %.2f ms" % (r, (t1-t0)*1e3, (t2-t1)*1e3, (t3-t2)*1e3, (t4-t3)*1e3, (t5-t4)*1e3, (t5-t0)*1e3), file=sys.stderr, flush=True)
    ctx.close()
