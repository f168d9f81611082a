"""A/B inside one process (round 6): the drop-in call (groups, thresholds, matrix from pageable host memory, build, 128 forced passes)
with the pair kernel's sides launched WHOLE (REO_EAGER_RANGES=1, round 5) against launched over RANGES of a side's sample blocks as the
chunks arrive (2, 3, 4 ranges per side; `d` = the library's default by shape).  Alternating contexts, medians; the result digest of
every setting must be the first one's.   python tools/ranges_ab.py [t0|t1|float] [G] [S] [settings, e.g. 1,d,2,3,4]"""
import os, sys, time, hashlib, numpy as np
sys.path.insert(0, '.')
import __graft_entry__ as ge
pkg = ge.load_pkg()
import torch
fam = sys.argv[1] if len(sys.argv) > 1 else "t0"
G = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
S = int(sys.argv[3]) if len(sys.argv) > 3 else 1000
settings = (sys.argv[4] if len(sys.argv) > 4 else "1,d,2,3,4").split(",")
X = np.asfortranarray({"t0": pkg.synth.t0_ranks, "t1": pkg.synth.t1_counts, "float": pkg.synth.float_expr}[fam](G, S, 3))
gid, _ = pkg.encode_groups(np.asarray(pkg.synth.groups(S))); ref0 = pkg.synth.ref_mask(G, 3000, 3)
os.environ["REO_CYCLE"] = "0"
ctxs = {}
for r in settings:
    if r.startswith("d"): os.environ.pop("REO_EAGER_RANGES", None)   # (d, d2, d3 ...: several contexts at the default)
    else: os.environ["REO_EAGER_RANGES"] = r
    ctxs[r] = pkg.Context(device=0, seed=3)
os.environ.pop("REO_EAGER_RANGES", None)
first = None
reps = 7 if G * S <= 3e7 else 4
for rnd in range(3):
    for r in settings:
        ctx = ctxs[r]; w = []
        for rep in range(reps):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            ctx.set_groups(gid, 2); ctx.compute_thresholds(0.01); ctx.set_matrix(X); ctx.build_pairs(0)
            res, it, tr = ctx.identify_degs(ref0, 1.0, 0.05, 128, 0)
            w.append((time.perf_counter() - t0) * 1e3)
        dg = hashlib.blake2b(np.ascontiguousarray(res).tobytes() + np.asarray(tr, dtype=np.int64).tobytes(), digest_size=8).hexdigest()
        first = first or dg
        print("ranges %s (%d launches over ranges): median %.2f ms  %s  %s" % (r, ctx.info()["eager_range_launches"], float(np.median(w[1:])),
              " ".join("%.2f" % x for x in w[1:]), "same result" if dg == first else "RESULT DIFFERS"), flush=True)
        assert dg == first
