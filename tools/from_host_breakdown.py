"""Where the drop-in call (matrix in pageable host memory) spends its time: wall clock of every call and the library's stage timers,
for the three settings of REO_EAGER_UPLOAD and a few chunk sizes.  python tools/from_host_breakdown.py [family] [G] [S]"""
import os, sys, time, numpy as np
sys.path.insert(0, '.')
import __graft_entry__ as ge
pkg = ge.load_pkg()
import torch
fam = sys.argv[1] if len(sys.argv) > 1 else "t0"
G = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
S = int(sys.argv[3]) if len(sys.argv) > 3 else 1000
seed = 0x5EED0003
X = np.asfortranarray({"t0": pkg.synth.t0_ranks, "t1": pkg.synth.t1_counts, "float": pkg.synth.float_expr}[fam](G, S, seed))
gid, _ = pkg.encode_groups(np.asarray(pkg.synth.groups(S))); ref0 = pkg.synth.ref_mask(G, 3000, seed)
os.environ["REO_CYCLE"] = "0"
def T(): return time.perf_counter()
for mode, chunk in (("0", None), ("1", None), ("2", None), ("2", "16"), ("2", "32"), ("2", "128"), ("2", "250"), ("2", "500")):
    os.environ["REO_EAGER_UPLOAD"] = mode
    if chunk: os.environ["REO_EAGER_CHUNK"] = chunk
    else: os.environ.pop("REO_EAGER_CHUNK", None)
    ctx = pkg.Context(device=0, seed=seed); ctx.set_profiling(os.environ.get("FHB_PROFILING", "1") == "1")
    rows = []
    for rep in range(5):
        ctx.reset_timings(); torch.cuda.synchronize()
        t = [T()]
        ctx.set_groups(gid, 2); ctx.compute_thresholds(0.01); t.append(T())
        ctx.set_matrix(X); t.append(T())
        ctx.build_pairs(0); t.append(T())
        torch.cuda.synchronize(); t.append(T())
        res, it, tr = ctx.identify_degs(ref0, 1.0, 0.05, 128, 0); t.append(T())
        tm = ctx.timings()
        rows.append([(b - a) * 1e3 for a, b in zip(t[:-1], t[1:])] + [(t[-1] - t[0]) * 1e3, tm["transform_ms"], tm["k1_ms"], tm["iter_ms"]])
    r = np.median(np.array(rows[1:]), axis=0)
    print("eager %s chunk %-4s: groups+thr %.2f | set_matrix %.2f | build_pairs %.2f | wait for K1 %.2f | identify_degs %.2f | TOTAL %.2f ms   (GPU timers: transform %.2f, K1 %.2f, passes %.2f)"
          % ((mode, chunk or "auto") + tuple(r)), flush=True)
    ctx.close()
