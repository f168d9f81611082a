import sys, time, numpy as np
sys.path.insert(0, '.')
import __graft_entry__ as ge
pkg = ge.load_pkg()
G, S, seed = 20000, 1000, 0x5EED0003
X = np.asfortranarray(pkg.synth.t0_ranks(G, S, seed))  # column-major like a Julia Matrix (a C-order numpy array costs a 30 ms transposing copy first)
group = pkg.synth.groups(S); ref0 = pkg.synth.ref_mask(G, 3000, seed)
names = list(range(G))
for n_conv in (0, 5):
    for rep in range(3):
        t = time.perf_counter()
        run = pkg.run_identify_degs(X, group, names, 0.01, 1.0, 0.05, ref0, 128, n_conv, seed=seed, device=0)
        dt = time.perf_counter() - t
    print("host-buffer boundary (PCIe + context create inclusive): n_conv=%d passes=%d  %.1f ms  %.3e cmp/s" % (n_conv, run.iters_run, dt*1e3, G*(G-1)//2*S/dt))
Xf = np.asfortranarray(X.astype(np.float64))
t = time.perf_counter(); run = pkg.run_identify_degs(Xf, group, names, 0.01, 1.0, 0.05, ref0, 128, 5, seed=seed, device=0); dt = time.perf_counter() - t
print("float64 input: %.1f ms" % (dt*1e3))
Xt = np.asfortranarray(pkg.synth.t1_counts(G, S, seed))
for rep in range(2):
    t = time.perf_counter(); run = pkg.run_identify_degs(Xt, group, names, 0.01, 1.0, 0.05, ref0, 128, 5, seed=seed, device=0, profile=True); dt = time.perf_counter() - t
print("tie-rich T1 counts: %.1f ms, passes %d, timings %s" % (dt*1e3, run.iters_run, run.timings))
