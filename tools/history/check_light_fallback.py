import os, sys
os.environ.update(REO_LIGHT_MIN_G="64", REO_LIGHT_WINDOW="1", REO_DEBUG_PASSES="1")
sys.path.insert(0, ".")
import numpy as np, __graft_entry__ as ge
pkg = ge.load_pkg()
rng = np.random.default_rng(5)
for G in (150, 400, 900):
    X = rng.integers(0, 9, size=(G, 30)); labels = np.array(["a"] * 15 + ["b"] * 15, dtype=object)
    ref0 = pkg.synth.ref_mask(G, G // 3, 1)
    print("G", G, file=sys.stderr)
    pkg.run_identify_degs(X, labels, list(range(G)), 0.05, 1.0, 0.3, ref0, 12, 0, seed=1, device=0)
