"""set_matrix (pipelined, ranking only: REO_EAGER_UPLOAD=1) wall time against the number of narrowing threads.  python tools/upload_sweep.py [family] [G] [S]"""
import os, sys, time, numpy as np
sys.path.insert(0, '.')
import __graft_entry__ as ge
pkg = ge.load_pkg()
import torch
fam = sys.argv[1] if len(sys.argv) > 1 else "t0"
G = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
S = int(sys.argv[3]) if len(sys.argv) > 3 else 1000
X = np.asfortranarray({"t0": pkg.synth.t0_ranks, "t1": pkg.synth.t1_counts}[fam](G, S, 3))
gid, _ = pkg.encode_groups(np.asarray(pkg.synth.groups(S)))
for mode in ("1", "2"):
    for thr in ("0", "2", "4", "8", "12", "16", "32"):
        os.environ["REO_EAGER_UPLOAD"] = mode; os.environ["REO_UPLOAD_THREADS"] = thr
        ctx = pkg.Context(device=0, seed=3); ctx.set_groups(gid, 2); ctx.compute_thresholds(0.01)
        w = []
        for rep in range(6):
            torch.cuda.synchronize(); t0 = time.perf_counter(); ctx.set_matrix(X); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
            w.append(((t1 - t0) * 1e3, (t2 - t0) * 1e3))
        m = np.median(np.array(w[1:]), axis=0)
        print("eager %s, %2s threads: set_matrix returns after %.2f ms, GPU idle after %.2f ms; link bytes %.0f MB" % (mode, thr, m[0], m[1], ctx.info()["upload_link_bytes"] / 1e6), flush=True)
        ctx.close()
