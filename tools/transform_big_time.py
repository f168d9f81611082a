"""Transform time above 65 535 genes: t_sample_big (round 5) against the library's segmented sort (REO_TRANSFORM=segmented), 70 000 x 1 000
and 140 000 x 400, counts / ranks / Float64.  python tools/transform_big_time.py"""
import os, subprocess, sys
if len(sys.argv) > 1:
    import numpy as np
    sys.path.insert(0, ".")
    import __graft_entry__ as ge
    pkg = ge.load_pkg()
    for G, S in ((70000, 1000), (140000, 400)):
        for fam in ("t1", "t0", "float"):
            X = {"t1": pkg.synth.t1_counts, "t0": pkg.synth.t0_ranks, "float": pkg.synth.float_expr}[fam](G, S, 5)
            gid, _ = pkg.encode_groups(np.asarray(pkg.synth.groups(S)))
            with pkg.Context(device=0, seed=5) as ctx:
                ctx.set_profiling(True)
                ts = []
                for r in range(4):
                    ctx.reset_timings(); ctx.set_matrix(X); ctx.set_groups(gid, 2); ctx.pair_counts(0, 8, 0, 8); ts.append(ctx.timings()["transform_ms"])
                print("%-9s %6d x %4d %-5s transform %.2f ms (form %d)" % (sys.argv[1], G, S, fam, min(ts[1:]), ctx.info()["transform_in_lds"]), flush=True)
        pkg._ffi.trim_memory()
else:
    for name, env in (("big", {}), ("segmented", {"REO_TRANSFORM": "segmented"})):
        subprocess.check_call([sys.executable, __file__, name], env=dict(os.environ, REO_EAGER_UPLOAD="0", **env))
