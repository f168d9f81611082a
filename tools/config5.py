"""BASELINE config 5 on one GPU: synthetic scRNA-seq 20,000 genes x 50,000 sparse cells, n_pseudo = 64
per group -> 128 pseudo-bulk profiles -> identify_degs.  Prints timings and the pseudo-bulk kernel's
effective bandwidth.  (The 8-GPU version of this config is the driver's to run.)"""
import sys
import time

import numpy as np
import scipy.sparse as sp

sys.path.insert(0, ".")
import __graft_entry__ as ge

pkg = ge.load_pkg()
import importlib
R = importlib.import_module(pkg.__name__ + ".reoa")

G, C, seed, dens = 20000, 50000, 0x5EED0005, 0.06
rng = np.random.default_rng(seed)
t0 = time.perf_counter()
# zero-inflated counts with heavy-tailed gene scales; cells of group 2 shift 10 % of the genes
scale = 2.0 ** rng.integers(0, 9, size=G)
nnz_per_cell = rng.binomial(G, dens, size=C)
indptr = np.concatenate([[0], np.cumsum(nnz_per_cell)]).astype(np.int64)
rows = np.concatenate([np.sort(rng.choice(G, n, replace=False)) for n in nnz_per_cell]).astype(np.int32)
eff = np.where(rng.random(G) < 0.1, rng.choice([0.5, 2.0], size=G), 1.0)
cell_of = np.repeat(np.arange(C), nnz_per_cell)
vals = 1 + rng.poisson(scale[rows] * np.where(cell_of >= C // 2, eff[rows], 1.0))
X = sp.csc_matrix((vals.astype(np.int64), rows, indptr), shape=(G, C))
print("generated %d nnz (%.1f %%) in %.1f s" % (X.nnz, 100 * X.nnz / (G * C), time.perf_counter() - t0))

orders, ptrs, base = [], [0], 0
for gi, (lo, hi) in enumerate(((0, C // 2), (C // 2, C))):
    o, p = R.pseudobulk_partition(hi - lo, 64, seed, gi)
    orders.append(o + lo); ptrs += (p[1:] + base).tolist(); base += hi - lo
order, ptr = np.concatenate(orders), np.asarray(ptrs, dtype=np.int32)
with pkg.Context(device=0, seed=seed) as ctx:
    ctx.set_profiling(True)
    for rep in range(2):
        ctx.reset_timings()
        t = time.perf_counter()
        pb = ctx.pseudobulk(X, order, ptr)
        wall = time.perf_counter() - t
        ms = ctx.timings()["pseudobulk_ms"]
    in_bytes = X.nnz * 12 + (C + 1) * 8
    print("pseudobulk CSC: %d profiles, kernel %.3f ms (%.0f GB/s of CSC input), call wall %.1f ms (H2D of %.0f MB inclusive)"
          % (pb.shape[1], ms, in_bytes / ms / 1e6, wall * 1e3, in_bytes / 1e6))
    assert np.array_equal(pb.sum(axis=1), np.asarray(X.sum(axis=1)).ravel())
group = np.array(["g1"] * 64 + ["g2"] * 64, dtype=object)
keep = (pb > 0).sum(axis=1) > 0
pbk = pb[keep]
ref0 = pkg.synth.ref_mask(pbk.shape[0], 3000, seed)
for rep in range(2):
    t = time.perf_counter()
    run = pkg.run_identify_degs(pbk, group, list(range(pbk.shape[0])), 0.01, 1.0, 0.05, ref0, 128, 5, seed=seed, device=0, profile=True)
    dt = time.perf_counter() - t
Gk = pbk.shape[0]
print("identify_degs on %d x 128: %.1f ms, passes %d, trace %s, thresholds %s, timings %s"
      % (Gk, dt * 1e3, run.iters_run, run.trace[-1], run.thresholds[:, 0].tolist(), {k: round(float(v), 3) for k, v in run.timings.items()}))
print("comparisons/s (host-buffer boundary): %.3e" % (Gk * (Gk - 1) // 2 * 128 / dt))
