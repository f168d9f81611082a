"""Average rocprofv3 --pmc counters per kernel from a counter_collection.csv (python tools/pmc_agg.py DIR)."""
import collections, csv, glob, sys
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:60]; agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
    for k, v in agg.items():
        print(k, "launches", len(n[k]))
        for c, x in sorted(v.items()): print("    %-24s %.4g per launch" % (c, x / len(n[k])))
