"""The C oracle under AddressSanitizer + UBSan on the CPU (no GPU): the randomized cases of tools/fuzz_gpu.py,
every entry point the GPU tests use, cross-checked against the numpy restatement.  Run as
    python tools/asan_oracle.py [N]
(re-executes itself with the sanitizer runtime preloaded)."""
import os, subprocess, sys
HERE = os.path.dirname(os.path.abspath(__file__)); ROOT = os.path.dirname(HERE)
SO = "/tmp/liboracle_asan.so"
if os.environ.get("REO_ASAN_CHILD") != "1":
    subprocess.check_call(["gcc", "-O1", "-g", "-fPIC", "-fopenmp", "-ffp-contract=off", "-fsanitize=address,undefined",
                           "-fno-omit-frame-pointer", "-shared", "-o", SO, os.path.join(ROOT, "oracle", "reo_oracle.c"), "-lm"])
    asan = subprocess.check_output(["gcc", "-print-file-name=libasan.so"]).decode().strip()
    env = dict(os.environ, REO_ASAN_CHILD="1", LD_PRELOAD=asan, ASAN_OPTIONS="detect_leaks=0:halt_on_error=1",
               UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1", OMP_NUM_THREADS="4")
    sys.exit(subprocess.call([sys.executable] + sys.argv, env=env))

import ctypes
import numpy as np
sys.path.insert(0, ROOT)
import oracle
from oracle import reo_numpy as rn
oracle.build = lambda force=False: SO          # load the instrumented library instead of oracle/liboracle.so
L = oracle.lib()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 150
rng = np.random.default_rng(777)
for n in range(N):
    G = int(rng.choice([rng.integers(10, 60), rng.integers(60, 260)]))
    ng = int(rng.choice([2, 2, 3, 5]))
    sizes = rng.integers(2, 24, size=ng); S = int(sizes.sum())
    gid = np.concatenate([[g] * int(k) for g, k in enumerate(sizes)]).astype(np.int32)
    if rng.random() < 0.5:
        perm = rng.permutation(S); gid = gid[perm]
        _, first = np.unique(gid, return_index=True)   # renumber in order of first appearance
        remap = {g: r for r, g in enumerate(gid[np.sort(first)])}; gid = np.array([remap[g] for g in gid], dtype=np.int32)
    kind = str(rng.choice(["small_int", "wide_int", "float_band", "ranks", "float_cont"]))
    if kind == "small_int": X = rng.integers(0, int(rng.integers(2, 12)), size=(G, S)).astype(np.float64)
    elif kind == "wide_int": X = rng.integers(-50000, 50000, size=(G, S)).astype(np.float64)
    elif kind == "float_band": X = np.round(rng.normal(5, 1.0, size=(G, S)), 1) + rng.choice([0.0, 0.04, 0.099, 0.1], size=(G, S))
    elif kind == "ranks": X = np.argsort(np.argsort(rng.random((G, S)), axis=0), axis=0).astype(np.float64)
    else: X = rng.normal(0, 3, size=(G, S))
    seed = int(rng.integers(0, 2 ** 40)); pval_reo = float(rng.choice([0.01, 0.05, 0.3]))
    ref0 = np.zeros(G, dtype=bool); ref0[rng.choice(G, int(rng.integers(3, G)), replace=False)] = True
    n_iter, n_conv = int(rng.integers(0, 9)), int(rng.choice([0, 1, 5]))
    gt, eq = oracle.pair_counts(X, gid, ng, 0, G, 0, G)
    gtn, eqn = rn.pair_counts(X, gid, ng)
    assert np.array_equal(gt, gtn) and np.array_equal(eq, eqn), (n, kind)
    for k in range(ng if ng > 2 else 1):
        sz = np.bincount(gid, minlength=ng)
        thr = [oracle.threshold(int(sz[k]), pval_reo), oracle.threshold(int(S - sz[k]), pval_reo)]
        assert thr == [rn.threshold(int(sz[k]), pval_reo), rn.threshold(int(S - sz[k]), pval_reo)]
        if min(thr) < 0: continue
        code = oracle.build_codes(X, gid, ng, k, thr, seed)
        assert np.array_equal(code, rn.build_codes(X, gid, ng, k, thr, seed)), (n, kind, k)
        assert np.array_equal(oracle.tally(code, ref0), rn.tally(code, ref0))
        try:
            res, it, tr = oracle.iterate(code, ref0, 1.0, 0.05, n_iter, n_conv)
        except IndexError:            # the reference's BoundsError at :411 for tiny G: both restatements must agree on it
            try:
                rn.iterate(code, ref0, 1.0, 0.05, n_iter, n_conv)
                raise AssertionError("numpy restatement did not raise")
            except IndexError:
                continue
        resn, itn, trn = rn.iterate(code, ref0, 1.0, 0.05, n_iter, n_conv)
        assert it == itn and tr == trn and np.allclose(res, resn, rtol=1e-9, atol=1e-12, equal_nan=True), (n, kind, k)
    if n % 25 == 24: print("case", n + 1, "ok", flush=True)
print("asan/ubsan clean:", N, "cases")
