"""Where the host-buffer (PCIe-inclusive) call spends its time, stage by stage."""
import sys, time, numpy as np
sys.path.insert(0, '.')
import __graft_entry__ as ge
pkg = ge.load_pkg()
import torch
G, S, seed = 20000, 1000, 0x5EED0003
X = np.asfortranarray(pkg.synth.t0_ranks(G, S, seed))  # column-major, what a Julia Matrix is
gid, _ = pkg.encode_groups(np.asarray(pkg.synth.groups(S))); ref0 = pkg.synth.ref_mask(G, 3000, seed)
def T(): torch.cuda.synchronize(); return time.perf_counter()
for rep in range(4):
    t = [T()]
    ctx = pkg.Context(device=0, seed=seed); t.append(T())
    ctx.set_groups(gid, 2); thr = ctx.compute_thresholds(0.01); t.append(T())
    ctx.set_matrix(X); t.append(T())                      # (groups first: the upload is pipelined with the ranking and the pair kernel)
    ctx.build_pairs(0); t.append(T())
    res, iters, trace = ctx.identify_degs(ref0, 1.0, 0.05, 128, 5); t.append(T())
    ctx.close(); t.append(T())
    names = ["create", "groups+thresholds", "set_matrix (upload + ranking + pair kernel)", "build_pairs", "identify_degs", "destroy"]
    print("rep %d: " % rep + ", ".join("%s %.2f" % (n, (b - a) * 1e3) for n, a, b in zip(names, t[:-1], t[1:])) + " | total %.2f ms" % ((t[-1] - t[0]) * 1e3), flush=True)
