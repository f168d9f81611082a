import sys, time, numpy as np
sys.path.insert(0, '.')
import __graft_entry__ as ge
pkg = ge.load_pkg(); oracle = ge.load_oracle()
G, S, seed = 65535, 16, 0x5EED0021
X = pkg.synth.t1_counts(G, S, seed)
group = np.array(["a", "b"] * 8, dtype=object)
gid, lev = pkg.encode_groups(group)
ref0 = pkg.synth.ref_mask(G, 3000, seed)
t = time.perf_counter()
run = pkg.run_identify_degs(X, group, list(range(G)), 0.05, 1.0, 0.05, ref0, 12, 5, seed=seed, device=0)
print("gpu %.2f s, passes %d trace %s" % (time.perf_counter() - t, run.iters_run, run.trace[-1]), flush=True)
t = time.perf_counter()
exp, iters, trace = oracle.identify_degs(X.astype(np.float64), gid, 2, 0.05, 1.0, 0.05, ref0, 12, 5, seed)
print("oracle %.1f s, passes %d trace %s" % (time.perf_counter() - t, iters, trace[-1]), flush=True)
assert iters == run.iters_run and trace == run.trace
assert np.array_equal(run.result[:, 2:11], exp[:, 2:11])
ok = np.isfinite(exp).all(axis=1)
print("max |dp|", np.abs(run.result[ok][:, :2] - exp[ok][:, :2]).max(), "max rel stat", np.nanmax(np.abs(run.result[ok][:, 11:] - exp[ok][:, 11:]) / (np.abs(exp[ok][:, 11:]) + 1e-9)))
print("OK")
