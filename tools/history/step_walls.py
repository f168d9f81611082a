import os, sys, time, numpy as np, torch
sys.path.insert(0, ".")
import __graft_entry__ as ge
pkg = ge.load_pkg()
G, S, seed = 20000, 1000, 0x5EED0003
os.environ["REO_CYCLE"] = sys.argv[1] if len(sys.argv) > 1 else "0"
gid, lev = pkg.encode_groups(pkg.synth.groups(S)); ref0 = pkg.synth.ref_mask(G, 3000, seed)
for fam in ("float", "t0"):
    X = {"t0": pkg.synth.t0_ranks, "float": pkg.synth.float_expr}[fam](G, S, seed)
    Xd = torch.from_numpy(np.ascontiguousarray(X.T)).to("cuda:0"); torch.cuda.synchronize()
    ctx = pkg.Context(device=0, seed=seed); ctx.set_profiling(True)
    for i in range(8):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        ctx.set_matrix_device(Xd.data_ptr(), G, S, G, "f64" if X.dtype == np.float64 else "i64"); t1 = time.perf_counter()
        ctx.set_groups(gid, 2); ctx.compute_thresholds(0.01); t2 = time.perf_counter()
        ctx.build_pairs(0); t3 = time.perf_counter()
        r = ctx.identify_degs(ref0, 1.0, 0.05, 128, 0); torch.cuda.synchronize(); t4 = time.perf_counter()
        print(fam, i, "set_matrix %.2f groups+thr %.2f build_pairs %.2f identify %.2f total %.2f ms" % ((t1-t0)*1e3, (t2-t1)*1e3, (t3-t2)*1e3, (t4-t3)*1e3, (t4-t0)*1e3), flush=True)
    ctx.close()
