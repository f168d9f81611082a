"""The ORACLE half of tools/fuzz_gpu.py under AddressSanitizer + UBSan, on the CPU: the same generator, the same seed, the same
order of random draws, so case n here is case n of the GPU run -- without the GPU call.  Written in round 5 to answer one
question about profiles/faults/r4_fuzz_gpu_host_crash.log (a host SIGSEGV with ONE Python frame, `<module>`: a damaged heap, not a
fault inside a library call): fuzz_gpu.py calls TWO native libraries per case, libreo_hip.so and the C oracle; if the oracle
overruns a caller's numpy buffer, the crash surfaces later in free()/malloc() exactly like that.
    python tools/replay_fuzz_oracle_asan.py [first] [last] [seed]     (default: cases 0..230 of seed 2026)
(re-executes itself with the sanitizer runtime preloaded)."""
import os, subprocess, sys
HERE = os.path.dirname(os.path.abspath(__file__)); ROOT = os.path.dirname(HERE)
SO = "/tmp/liboracle_asan_full.so"
if os.environ.get("REO_ASAN_CHILD") != "1":
    subprocess.check_call(["gcc", "-O1", "-g", "-fPIC", "-fopenmp", "-ffp-contract=off", "-fsanitize=address,undefined",
                           "-fno-omit-frame-pointer", "-shared", "-o", SO, os.path.join(ROOT, "oracle", "reo_oracle.c"),
                           os.path.join(ROOT, "oracle", "reo_tuned.c"), "-lm"])
    asan = subprocess.check_output(["gcc", "-print-file-name=libasan.so"]).decode().strip()
    env = dict(os.environ, REO_ASAN_CHILD="1", LD_PRELOAD=asan, ASAN_OPTIONS="detect_leaks=0:halt_on_error=1",
               UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1", OMP_NUM_THREADS=os.environ.get("OMP_NUM_THREADS", "8"))
    sys.exit(subprocess.call([sys.executable] + sys.argv, env=env))

import time
import numpy as np
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
import oracle
oracle.build = lambda force=False: SO          # the instrumented library instead of oracle/liboracle.so
pkg = ge.load_pkg()
FIRST = int(sys.argv[1]) if len(sys.argv) > 1 else 0
LAST = int(sys.argv[2]) if len(sys.argv) > 2 else 230
rng = np.random.default_rng(int(sys.argv[3]) if len(sys.argv) > 3 else 2026)


def case():   # tools/fuzz_gpu.py, verbatim order of draws
    G = int(rng.choice([rng.integers(12, 200), rng.integers(200, 1400), rng.integers(1400, 2600)]))
    ng = int(rng.choice([2, 2, 3, 5]))
    sizes = rng.integers(2, 40, size=ng)
    S = int(sizes.sum())
    labels = np.concatenate([[f"grp{g}"] * int(n) for g, n in enumerate(sizes)])
    if rng.random() < 0.5:
        labels = labels[rng.permutation(S)]
    kind = str(rng.choice(["small_int", "wide_int", "float_band", "ranks", "float_cont", "big_int"]))
    if kind == "small_int": X = rng.integers(0, int(rng.integers(2, 12)), size=(G, S))
    elif kind == "wide_int": X = rng.integers(-50000, 50000, size=(G, S))
    elif kind == "big_int": X = rng.integers(0, 2 ** 31, size=(G, S))
    elif kind == "float_band": X = np.round(rng.normal(5, 1.0, size=(G, S)), 1) + rng.choice([0.0, 0.04, 0.099, 0.1], size=(G, S))
    elif kind == "ranks": X = np.argsort(np.argsort(rng.random((G, S)), axis=0), axis=0)
    else: X = rng.normal(0, 3, size=(G, S))
    return dict(G=G, S=S, ng=ng, labels=labels, X=X, kind=kind, pval_reo=float(rng.choice([0.01, 0.05, 0.3])),
                n_conv=int(rng.choice([1, 5])), n_iter=int(rng.integers(1, 9)), seed=int(rng.integers(0, 2 ** 40)), nref=int(rng.integers(3, G)))


t0 = time.time()
for n in range(LAST + 1):
    cs = case(); G = cs["G"]
    if n < FIRST:
        continue
    gid, lev = pkg.encode_groups(cs["labels"])
    ref0 = pkg.synth.ref_mask(G, cs["nref"], cs["seed"])
    Xf = np.asarray(cs["X"], dtype=np.float64)
    ncomp = 1 if len(lev) == 2 else len(lev)
    for k in range(ncomp):
        oracle.identify_degs(Xf, gid, len(lev), cs["pval_reo"], 1.0, 0.05, ref0, cs["n_iter"], cs["n_conv"], cs["seed"], k=k)
    print("case %d ok: %s G=%d S=%d groups=%d n_iter=%d (%.0f s)" % (n, cs["kind"], G, cs["S"], cs["ng"], cs["n_iter"], time.time() - t0), flush=True)
print("oracle asan/ubsan clean on cases %d..%d" % (FIRST, LAST))
