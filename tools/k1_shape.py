"""transform + pair kernel (3 x build_pairs) on a given shape; a target for rocprofv3.
usage: python3 tools/k1_shape.py G S family [group sizes...]   family: t0 | t1 | float"""
import sys, time, numpy as np
sys.path.insert(0, '.')
import __graft_entry__ as ge
pkg = ge.load_pkg()
G, S, fam = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
seed = 0x5EED0003
X = {"t0": pkg.synth.t0_ranks, "t1": pkg.synth.t1_counts, "float": pkg.synth.float_expr}[fam](G, S, seed)
gid, _ = pkg.encode_groups(np.asarray(pkg.synth.groups(S)))
with pkg.Context(device=0, seed=seed) as ctx:
    ctx.set_profiling(True)
    for rep in range(3):   # (the matrix is set again every time: a build on an unchanged matrix reuses the planes; the first pass allocates)
        ctx.set_matrix(X); ctx.set_groups(gid, 2); ctx.compute_thresholds(0.01)
        ctx.reset_timings()
        ctx.build_pairs(0)
        tm = ctx.timings()
        transform_ms = tm["transform_ms"]
    P = G * (G - 1) // 2
    nb = 12 if G <= 4095 else (15 if G <= 32767 else (16 if G <= 65535 else (17 if G <= 131071 else 18)))
    ties = bool(ctx.info()["has_ties"])
    floor = 1024 * 2.4e9 * 2048 / ((2 * nb + 4) * (2 if ties else 1))
    print("%d x %d %s: transform %.3f ms, K1 %.3f ms = %.3e cmp/s = %.3f of its issue floor (%d planes%s)" %
          (G, S, fam, transform_ms, tm["k1_ms"], P * S / (tm["k1_ms"] * 1e-3), P * S / (tm["k1_ms"] * 1e-3) / floor, nb, ", two chains" if ties else ""))
