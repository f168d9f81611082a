"""Generates tests/golden/julia_float_vectors.json and tests/golden/bundled_slice64_result.tsv.

julia_float_vectors.json: Float64 values (as hex bit patterns) with the text Julia prints for them -- `print(x)` /
CSV.write use Base.Ryu.writeshortest: the shortest digits that round-trip, fixed notation with at least one decimal
for 1e-4 <= |x| < 1e6, otherwise d.ddde[-]X.  No Julia runs in this pipeline, so the expected strings are written
down here from that rule; the five McCullagh numbers are the reference's own printed output
(/root/reference/src/RankCompV3.jl:207-209) and pin the rule on real Julia output.
bundled_slice64_result.tsv: the result file (:665-671) for the golden 64-gene slice, from its committed result matrix
(tests/golden/bundled_slice64.json) -- a byte-level regression fixture of the writer.
Run from the repository root:  python tests/golden/make_tsv_golden.py"""
import json
import os
import struct
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

VECTORS = [
    # (value, text)
    (0.005469174895116946, "0.005469174895116946"), (1.4504988072997458, "1.4504988072997458"),   # :207-209, printed by Julia
    (1.502600073417028, "1.502600073417028"), (0.5221345956920705, "0.5221345956920705"), (2.778017046308073, "2.778017046308073"),
    (1e-5, "1.0e-5"), (1e-4, "0.0001"), (9.999e-5, "9.999e-5"), (0.00012345, "0.00012345"),
    (999999.0, "999999.0"), (999999.9, "999999.9"), (1e6, "1.0e6"), (1234567.0, "1.234567e6"), (123456.789, "123456.789"),
    (0.1 + 0.2, "0.30000000000000004"), (0.1, "0.1"), (1.0 / 3.0, "0.3333333333333333"), (2.0 / 3.0, "0.6666666666666666"),
    (5e-324, "5.0e-324"), (2.2250738585072014e-308, "2.2250738585072014e-308"), (1.7976931348623157e308, "1.7976931348623157e308"),
    (-0.0, "-0.0"), (0.0, "0.0"), (-3.25, "-3.25"), (1e22, "1.0e22"), (1e15, "1.0e15"), (2.5e21, "2.5e21"),
    (0.0, "0.0"), (1.0, "1.0"), (17074.0, "17074.0"), (2926.0, "2926.0"), (100000.0, "100000.0"), (65535.0, "65535.0"),  # tallies are written as Float64
    (1e-310, "1.0e-310"), (4.9406564584124654e-324, "5.0e-324"), (1.0000000000000002, "1.0000000000000002"),
    (float("nan"), "NaN"), (float("inf"), "Inf"), (float("-inf"), "-Inf"),
]


def main():
    import numpy as np
    import __graft_entry__ as ge
    pkg = ge.load_pkg()
    import importlib
    R = importlib.import_module(pkg.__name__ + ".reoa")
    with open(os.path.join(HERE, "julia_float_vectors.json"), "w") as f:
        json.dump([{"bits": "%016x" % struct.unpack("<Q", struct.pack("<d", v))[0], "text": t} for v, t in VECTORS], f, indent=0)
    g = json.load(open(os.path.join(HERE, "bundled_slice64.json")))
    res = np.array(g["result"], dtype=np.float64)
    labels = pkg.label_genes(res, g["pval_deg"], g["padj_deg"])
    names = ["slice%02d" % i for i in range(res.shape[0])]
    prep = {"g_name": ["group1", "group2"], "gene_names": names, "sample_names": ["s%d" % s for s in range(len(g["gid"]))],
            "sample_groups": ["group%d" % (1 + v) for v in g["gid"]], "data": np.array(g["X"], dtype=np.int64)}

    class Run:
        comparisons = [{"k": 0, "result": res, "labels": labels}]

    import tempfile
    with tempfile.TemporaryDirectory() as d:
        R.write_outputs("bundled_slice64", prep, Run, d)
        os.replace(os.path.join(d, "bundled_slice64_group1_group2_result.tsv"), os.path.join(HERE, "bundled_slice64_result.tsv"))


if __name__ == "__main__":
    main()
