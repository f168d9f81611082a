"""A/B inside one process: the sides of the pair kernel launched by k1w_pairs_gated behind an event (REO_EAGER_GATE=1) against the host
reading the transform's flags first (=0); the drop-in call sequence WITHOUT anything between the calls (as bench.py's from_host block
and the shims make it) and with a device wait between reo_set_matrix and reo_identify_degs (as tools/from_host_breakdown.py does)."""
import os, sys, time, numpy as np
sys.path.insert(0, '.')
import __graft_entry__ as ge
pkg = ge.load_pkg()
import torch
fam = sys.argv[1] if len(sys.argv) > 1 else "t0"
G = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
S = int(sys.argv[3]) if len(sys.argv) > 3 else 1000
X = np.asfortranarray({"t0": pkg.synth.t0_ranks, "t1": pkg.synth.t1_counts, "float": pkg.synth.float_expr}[fam](G, S, 3))
gid, _ = pkg.encode_groups(np.asarray(pkg.synth.groups(S))); ref0 = pkg.synth.ref_mask(G, 3000, 3)
os.environ["REO_CYCLE"] = "0"
ctxs = {}
for gate in ("1", "0"):
    os.environ["REO_EAGER_GATE"] = gate
    ctxs[gate] = pkg.Context(device=0, seed=3)
for rnd in range(3):
    for gate in ("1", "0"):
        ctx = ctxs[gate]
        for between in (False, True):
            w = []
            for rep in range(6):
                torch.cuda.synchronize(); t0 = time.perf_counter()
                ctx.set_groups(gid, 2); ctx.compute_thresholds(0.01); ctx.set_matrix(X); ctx.build_pairs(0)
                if between: torch.cuda.synchronize()
                ctx.identify_degs(ref0, 1.0, 0.05, 128, 0)
                w.append((time.perf_counter() - t0) * 1e3)
            print("gate %s, %s: median %.2f ms  %s" % (gate, "device wait between" if between else "calls back to back ", float(np.median(w[1:])), " ".join("%.2f" % x for x in w[1:])), flush=True)
