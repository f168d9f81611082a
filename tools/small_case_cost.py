import os, sys, time, numpy as np
os.environ.update(REO_LIGHT_MIN_G="64", REO_LIGHT_WINDOW="3")
sys.path.insert(0, '.')
import __graft_entry__ as ge
pkg = ge.load_pkg()
rng = np.random.default_rng(1)
G, S = 300, 24
X = rng.integers(0, 9, size=(G, S))
labels = np.array(["a"] * 12 + ["b"] * 12)
ref0 = pkg.synth.ref_mask(G, 100, 3)
for rep in range(4):
    t0 = time.perf_counter()
    run = pkg.run_identify_degs(X, labels, list(range(G)), 0.05, 1.0, 0.05, ref0, 10, 0, seed=3, device=0, profile=True)
    print("run %.1f ms" % ((time.perf_counter() - t0) * 1e3), run.timings["iter_ms"], run.trace[-1], flush=True)
