import sys, time
sys.path.insert(0, '.')
import __graft_entry__ as ge
pkg = ge.load_pkg()
import os
for env in ("1", "0"):
    os.environ["REO_XCC_LOCAL"] = env
    t = []
    for k in range(6):
        t0 = time.perf_counter()
        ctx = pkg.Context(device=0, seed=1)
        t1 = time.perf_counter()
        ctx.__exit__(None, None, None)
        t.append((t1 - t0) * 1e3)
    print("REO_XCC_LOCAL=%s: create ms" % env, ["%.2f" % v for v in t])
