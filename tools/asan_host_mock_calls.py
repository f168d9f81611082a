"""Driver of tools/asan_host_mock.sh: the host side of libreo_hip.so (ASan-instrumented, linked against tools/mockhip) through its
call sequences.  No GPU; results are garbage (no kernel runs) -- what is checked is every host write and every copy length.
    tools/asan_host_mock.sh [last_fuzz_case]      (default 230: the cases of profiles/faults/r4_fuzz_gpu_host_crash.log and before)"""
import ctypes, os, sys, time
sys.modules["torch"] = None          # _ffi.lib() would import torch to share its HIP runtime: there is no runtime here
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
pkg = ge.load_pkg()
ffi = pkg._ffi
assert "asan_mock" in ffi.LIB_PATH, "run through tools/asan_host_mock.sh (REO_LIB_PATH)"
L = ffi.lib()
mock = ctypes.CDLL(os.environ["REO_MOCK_LIB"])
mock.mockhip_launches.restype = ctypes.c_long
LAST = int(sys.argv[1]) if len(sys.argv) > 1 else 230
t0 = time.time()


def note(msg):
    print("%s  (%.0f s, %d launches so far)" % (msg, time.time() - t0, mock.mockhip_launches()), flush=True)


# ---- 1. the library half of tools/fuzz_gpu.py, same generator, same seed as the run that crashed --------------------------------
rng = np.random.default_rng(2026)


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


run = None
for n in range(LAST + 1):
    cs = case(); G = cs["G"]
    ref0 = pkg.synth.ref_mask(G, cs["nref"], cs["seed"])
    os.environ["MOCKHIP_N_ITER"] = str(cs["n_iter"])     # the mock "executes" n_iter passes per wait: the loop ends, trace and result are copied out
    run = pkg.run_identify_degs(cs["X"], cs["labels"], list(range(G)), cs["pval_reo"], 1.0, 0.05, ref0, cs["n_iter"], cs["n_conv"],
                                seed=cs["seed"], device=0, profile=(n % 3 == 0))
    assert run.iters_run == cs["n_iter"]
    if n % 50 == 49:
        note("fuzz case %d" % (n + 1))
note("library half of fuzz_gpu.py, seed 2026, cases 0..%d" % LAST)

# ---- 2. every other entry point on one context ------------------------------------------------------------------------------------
os.environ["MOCKHIP_N_ITER"] = "6"
G, S = 3000, 40
X = pkg.synth.t1_counts(G, S, 7)
gid, lev = pkg.encode_groups(np.asarray(pkg.synth.groups(S)))
with pkg.Context(device=0, seed=1) as ctx:
    ctx.set_profiling(True)
    ctx.set_matrix(X)
    ctx.set_matrix(np.asfortranarray(X.astype(np.float64))[:G - 7])      # a view with a leading dimension
    ctx.set_matrix(X)
    ctx.set_groups(gid, 2); ctx.compute_thresholds(0.01); ctx.set_thresholds([20, 20, 20, 20]); ctx.get_thresholds()
    ctx.build_pairs(0)
    ctx.pair_counts(0, 37, 100, 611); ctx.get_codes(5, 100, 0, G); ctx.tally(pkg.synth.ref_mask(G, 500, 3))
    for n_iter in (0, 1, 6):
        os.environ["MOCKHIP_N_ITER"] = str(max(n_iter, 1))
        ctx.identify_degs(pkg.synth.ref_mask(G, 500, 3), 1.0, 0.05, n_iter, 5)
    ctx.mccullagh(np.arange(90)); ctx.timings(); ctx.info(); ctx.reset_timings()
    # pseudo-bulk, dense and CSC
    cells = np.random.default_rng(1).integers(0, 5, size=(300, 90))
    order = np.random.default_rng(2).permutation(90).astype(np.int32); ptr = np.array([0, 30, 60, 90], dtype=np.int32)
    ctx.pseudobulk(cells, order, ptr); ctx.pseudobulk(cells.astype(np.float64), order, ptr)
    import scipy.sparse as sp
    ctx.pseudobulk(sp.csc_matrix(cells), order, ptr); ctx.pseudobulk(sp.csc_matrix(cells.astype(np.float64)), order, ptr)
    # errors after a context exists
    for bad in (lambda: ctx.pair_counts(0, G + 1, 0, 5), lambda: ctx.get_codes(3, 3, 0, 5), lambda: ctx.set_groups([0, 1], 2) or ctx.build_pairs(0)):
        try:
            bad(); raise SystemExit("no error raised")
        except ffi.ReoError:
            pass
note("entry points on one context")

# ---- 3. shapes that take the other kernels' host paths: light passes, more than 65 535 genes, more than 65 535 samples, many groups --
def one(G, S, ngroups=2, n_iter=3, ints=True, seed=5, shard=(0, 1), allgather=None, allreduce=None, kind=None):
    r = np.random.default_rng(seed)
    X = r.integers(0, 1000, size=(G, S)) if ints else r.normal(size=(G, S))
    if kind == "float_counts": X = r.integers(0, 100000, size=(G, S)).astype(np.float64); X[:, S // 2:] += 0.5; X[:, 3 * S // 4:] += 1e-9   # int32, float32, raw on the way
    if kind == "wide_ints": X = r.integers(0, 30000, size=(G, S)); X[:, S // 3:] += 2 ** 20; X[7, S - 2] = 2 ** 50                          # int16, int32, raw
    labels = np.repeat(np.arange(ngroups), -(-S // ngroups))[:S]
    os.environ["MOCKHIP_N_ITER"] = str(n_iter)
    return pkg.run_identify_degs(X, labels, list(range(G)), 0.05, 1.0, 0.05, pkg.synth.ref_mask(G, G // 4, seed), n_iter, 0, seed=seed, device=0,
                                 shard=shard, allgather=allgather, allreduce=allreduce)


one(9000, 64, n_iter=40)                  # light passes (G >= 4096): batches, the replay
for ch in ("7", "16", None):              # every form of a chunk on the link, climbing on the way, small and default chunks
    if ch: os.environ["REO_EAGER_CHUNK"] = ch
    one(5000, 64, kind="float_counts"); one(5000, 64, kind="wide_ints")
    os.environ.pop("REO_EAGER_CHUNK", None)
one(9000, 64, n_iter=40, ints=False)
one(30000, 16)
one(70000, 8)                             # 32-bit positions, the segmented sort's host side, k1w_pairs<17>
one(300, 66000)                           # more than 65 535 samples: the wide pair kernels
one(5000, 70, ngroups=7)                  # one-vs-rest with shared group counts
os.environ["REO_SHARE_GROUP_COUNTS"] = "0"; one(3000, 70, ngroups=7); del os.environ["REO_SHARE_GROUP_COUNTS"]
# (REO_LIGHT=2 and REO_STATE_MIRROR=0 copy the DEVICE copy of the loop state back, which no kernel has written here: only with the block
#  cache off, when the mock's fresh "device" blocks are zeros)
unmirrored = (("REO_LIGHT", "2"), ("REO_STATE_MIRROR", "0")) if os.environ.get("REO_DEVICE_CACHE_MB") == "0" else ()
for e, v in (("REO_LIGHT", "0"), ("REO_LIGHT", "3"), ("REO_K1_WAVE", "0"), ("REO_TRANSFORM", "wide"),
             ("REO_EAGER_UPLOAD", "0"), ("REO_EAGER_UPLOAD", "1"), ("REO_EAGER_CHUNK", "7")) + unmirrored:
    os.environ[e] = v; one(9000, 40, n_iter=12); one(9000, 41, n_iter=3, ints=False); del os.environ[e]
# (round 6) the pair kernel's sides launched over RANGES of a group's sample blocks as the chunks arrive: park buffers, the range
# bookkeeping of eager_upload, interleaved and unequal groups, every number of ranges, with Int64 and Float64 chunks
for rg, ch in (("2", "64"), ("4", "33"), ("6", "32"), (None, None), ("3", None)):
    if rg: os.environ["REO_EAGER_RANGES"] = rg
    if ch: os.environ["REO_EAGER_CHUNK"] = ch
    one(3000, 700, n_iter=2); one(3000, 700, n_iter=2, ints=False); one(2000, 901, n_iter=2, kind="wide_ints")
    for k in ("REO_EAGER_RANGES", "REO_EAGER_CHUNK"): os.environ.pop(k, None)
one(9000, 4100, n_iter=2)                 # (default ranges by shape: 65 blocks per side -> three ranges)
L.reo_trim_memory()
note("other shapes and switches")

# ---- 4. shards: both hooks, the pipelined exchange, the in-library communicator, the multi-GPU context -------------------------------
def ag_hook(send, recv, nbytes, stream):   # (one process stands for every shard: the own pack into the own slot is all it can deliver)
    ctypes.memmove(recv, send, nbytes)


def ar_hook(ptr, count, stream):
    pass


for world in (2, 3, 8):
    for rank in (0, world - 1):
        for waves in ("1", "4", "8"):
            os.environ["REO_EXCHANGE_WAVES"] = waves
            try:
                one(9000, 40, shard=(rank, world), allgather=ag_hook)
            except ffi.ReoError as e:     # the mock's zero table passes the consistency scan; anything else is a finding
                raise
        one(2300, 40, shard=(rank, world), allreduce=ar_hook)
del os.environ["REO_EXCHANGE_WAVES"]
with pkg.Context(device=0, seed=1) as ctx:
    ctx.comm_init_rank(ffi.comm_unique_id(), 0, 1)
    ctx.set_matrix(X); ctx.set_groups(gid, 2); ctx.compute_thresholds(0.01); ctx.build_pairs(0)
    os.environ["MOCKHIP_N_ITER"] = "2"; ctx.identify_degs(pkg.synth.ref_mask(G, 500, 3), 1.0, 0.05, 2, 0)
os.environ["MOCKHIP_NDEV"] = "4"
for seam in ("0", "1"):
    os.environ["REO_MULTI_ONE_DEVICE"] = seam
    with pkg.Context(seed=1, n_gpus=3) as ctx:
        ctx.set_matrix(X); ctx.set_groups(gid, 2); ctx.compute_thresholds(0.01); ctx.build_pairs(0)
        ctx.identify_degs(pkg.synth.ref_mask(G, 500, 3), 1.0, 0.05, 2, 0)
del os.environ["REO_MULTI_ONE_DEVICE"], os.environ["MOCKHIP_NDEV"]
note("shards, hooks, communicator, multi-GPU context")
print("asan_host_mock_calls: all call sequences done in %.0f s" % (time.time() - t0))
