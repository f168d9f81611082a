"""Two pipelined reo_set_matrix calls (the second one is the one to look at) for a kernel / copy timeline under rocprofv3:
   rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d DIR -o t -- python3 tools/upload_trace.py [family] [G] [S]
then tools/upload_trace.py --show DIR prints the last call's timeline."""
import os, sys, glob, csv
if len(sys.argv) > 2 and sys.argv[1] == "--show":
    rows = []
    for f in glob.glob(os.path.join(sys.argv[2], "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:60], r.get("Stream_Id", r.get("Queue_Id", ""))))
    for f in glob.glob(os.path.join(sys.argv[2], "**", "*memory_copy_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", "") , ""))
    rows.sort()
    # the last table clear (a fill kernel or memset) marks the start of the last call: find the last t_widen/t_sample burst instead
    last_k1 = [i for i, r in enumerate(rows) if "k1w_pairs" in r[2]]
    end = rows[last_k1[-1]][1]
    # the last call begins behind the device wait between the two calls: the longest idle gap in front of the last pair-kernel launch
    # (round 6: a call launches the pair kernel several times -- ranges of a side's blocks -- so counting launches no longer finds it)
    busy_until, gap, start_idx = rows[0][1], -1, 0
    for i in range(1, last_k1[-1] + 1):
        if i > len(rows) // 3 and rows[i][0] - busy_until > gap: gap, start_idx = rows[i][0] - busy_until, i
        busy_until = max(busy_until, rows[i][1])
    t0 = rows[start_idx][0]
    for r in rows[start_idx:]:
        if r[0] > end: break
        print("%9.3f ms  +%8.3f ms  %s  q%s" % ((r[0] - t0) / 1e6, (r[1] - r[0]) / 1e6, r[2], r[3]))
    sys.exit(0)
import numpy as np
sys.path.insert(0, '.')
import __graft_entry__ as ge
pkg = ge.load_pkg()
import torch
fam = sys.argv[1] if len(sys.argv) > 1 else "t0"
G = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
S = int(sys.argv[3]) if len(sys.argv) > 3 else 1000
X = np.asfortranarray({"t0": pkg.synth.t0_ranks, "t1": pkg.synth.t1_counts, "float": pkg.synth.float_expr}[fam](G, S, 3))
gid, _ = pkg.encode_groups(np.asarray(pkg.synth.groups(S)))
ctx = pkg.Context(device=0, seed=3); ctx.set_groups(gid, 2); ctx.compute_thresholds(0.01)
for rep in range(2):
    ctx.set_matrix(X); torch.cuda.synchronize()
ctx.close()
