"""reoa()'s preprocessing and writers (src/RankCompV3.jl:557-651, 663-683) -- host logic, no GPU."""
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def R(pkg):
    import importlib
    return importlib.import_module(pkg.__name__ + ".reoa")


def test_julia_float_formatting(R):
    f = R.julia_float
    assert [f(v) for v in (0.0, 1.0, 12.0, 0.5, 100000.0, 999999.0, 1e6, 1234567.0, 1e-4, 1e-5, 0.00001234,
                           1.5e-7, 2.5e21, -3.25, 0.1, 1.4504988072997458, float("nan"), float("inf"))] == \
        ["0.0", "1.0", "12.0", "0.5", "100000.0", "999999.0", "1.0e6", "1.234567e6", "0.0001", "1.0e-5", "1.234e-5",
         "1.5e-7", "2.5e21", "-3.25", "0.1", "1.4504988072997458", "NaN", "Inf"]
    rng = np.random.default_rng(0)
    for v in np.concatenate([rng.normal(0, 1, 200), 10.0 ** rng.uniform(-12, 12, 200)]):
        assert float(f(float(v))) == float(v)  # round-trips


def test_prepare_bundled_data_matches_the_readme(R):
    """README.md:39-41: 19999 genes survive (EP5057 is all-zero and dropped by the row filter of :626)."""
    p = R.prepare(os.path.join(GOLD, "fn_expr.txt"), os.path.join(GOLD, "fn_meta.txt"), seed=3)
    assert p["data"].shape == (19999, 10) and "EP5057" not in p["gene_names"] and p["gene_names"][0] == "DE1"
    assert p["g_name"] == ["group1", "group2"] and p["sample_groups"] == ["group1"] * 5 + ["group2"] * 5
    assert p["sample_names"][0] == "Sample1" and p["ref"].sum() == 3000  # no HK table -> random 3000 (:635)
    assert np.issubdtype(p["data"].dtype, np.integer)


def _write(tmp_path, name, text):
    path = tmp_path / name
    path.write_text(text)
    return str(path)


def test_argument_errors_mirror_the_reference(R, tmp_path):
    e = _write(tmp_path, "e.tsv", "gene\ts1\ts2\ts3\ts4\nA\t1\t2\t3\t4\nB\t0\t0\t5\t6\nC\t0\t0\t0\t0\n")
    m = _write(tmp_path, "m.tsv", "sample\tgrp\ns1\tx\ns2\tx\ns3\ty\ns4\ty\n")
    p = R.prepare(e, m)
    assert p["gene_names"] == ["A", "B"] and p["data"].tolist() == [[1, 2, 3, 4], [0, 0, 5, 6]]  # C filtered (:626)
    with pytest.raises(R.ArgumentError):  # :565
        R.prepare(e, str(tmp_path / "missing.tsv"))
    with pytest.raises(R.ArgumentError):  # :566
        R.prepare(e, _write(tmp_path, "empty.tsv", ""))
    with pytest.raises(R.ArgumentError):  # :574
        R.prepare(e, _write(tmp_path, "m1.tsv", "sample\ns1\n"))
    with pytest.raises(R.ArgumentError):  # :580-582
        R.prepare(e, _write(tmp_path, "m2.tsv", "sample\tgrp\ns1\tx\nzz\ty\n"))
    with pytest.raises(R.ArgumentError):  # :589-591
        R.prepare(e, _write(tmp_path, "m3.tsv", "sample\tgrp\ns1\tx\ns2\tx\ns3\tx\ns4\tx\n"))
    with pytest.raises(R.ArgumentError):
        R.prepare(str(tmp_path / "x.rds"), m)
    # min_profiles filters samples, min_features filters genes (:618,626)
    p2 = R.prepare(e, m, min_profiles=1)
    assert p2["sample_names"] == ["s3", "s4"] and p2["sample_groups"] == ["y", "y"]


def test_infinities_in_the_expression_file_are_kept(R, tmp_path):
    """A log-transformed table holds log(0) = -Inf; CSV.jl reads "-Inf" as a Float64 and dropmissing! (:601) drops `missing`, not
    infinities: the rows stay and reach identify_degs, where is_greater compares them (:71-77).  The loader keeps them too; the
    row filter of :626 counts the POSITIVE entries of a row (`sum(row .> 0) > min_features`): -Inf is not positive, +Inf is."""
    e = _write(tmp_path, "e.tsv", "gene\ts1\ts2\ts3\ts4\nA\t1.5\t-Inf\t3\tInf\nB\t-Inf\t-Inf\t-Inf\t-Inf\nC\t0.5\t0.25\tinf\t2\nD\t1\t2\t3\t4\n")
    m = _write(tmp_path, "m.tsv", "sample\tgrp\ns1\tx\ns2\tx\ns3\ty\ns4\ty\n")
    p = R.prepare(e, m)
    assert p["gene_names"] == ["A", "C", "D"]                       # B has no positive entry (:626); A has three (1.5, 3, Inf)
    data = np.asarray(p["data"], dtype=np.float64)
    assert np.isinf(data).sum() == 3 and np.isneginf(data[0, 1]) and np.isposinf(data[0, 3]) and np.isposinf(data[1, 2])
    assert R.julia_float(float("-inf")) == "-Inf" and R.julia_float(float("inf")) == "Inf"


def test_hk_table_selects_reference_genes(R, tmp_path):
    genes = [f"ENSG{i:05d}" for i in range(300)]
    rows = "\n".join(f"{g}\t" + "\t".join(str((i * 7 + s) % 11 + 1) for s in range(4)) for i, g in enumerate(genes))
    e = _write(tmp_path, "e.tsv", "Name\ts1\ts2\ts3\ts4\n" + rows + "\n")
    m = _write(tmp_path, "m.tsv", "Name\tGroup\ns1\tx\ns2\tx\ns3\ty\ns4\ty\n")
    hk = _write(tmp_path, "hk.tsv", "Name\tENSEMBL\n" + "\n".join(f"n{i}\t{g}" for i, g in enumerate(genes[:150])) + "\n")
    p = R.prepare(e, m, hk_file=hk)
    assert p["ref"].sum() == 150 and p["ref"][:150].all()
    few = _write(tmp_path, "hk2.tsv", "Name\tENSEMBL\n" + "\n".join(f"n{i}\t{g}" for i, g in enumerate(genes[:20])) + "\n")
    assert R.prepare(e, m, hk_file=few, ref_gene_max=50)["ref"].sum() == 50  # < ref_gene_min -> random (:645-648)
    with pytest.raises(R.ArgumentError):
        R.prepare(e, m, hk_file=str(tmp_path / "nope.tsv"))


def test_pseudobulk_group_semantics(R):
    """src/RankCompV3.jl:56-67: chunks of ceil(c/n_pseudo) shuffled cells, row sums, names <g>_x<k>."""
    rng = np.random.default_rng(1)
    vals = rng.integers(0, 33, size=(10, 6))
    out, names = R.pseudobulk_group(vals, 3, "group1", seed=5, stream=0)
    assert out.shape == (10, 3) and names == ["group1_x1", "group1_x2", "group1_x3"]
    assert np.array_equal(out.sum(axis=1), vals.sum(axis=1))
    out2, names2 = R.pseudobulk_group(rng.integers(0, 9, size=(4, 50000 // 2)), 64, "g", seed=5, stream=1)
    assert out2.shape == (4, 64) and names2[-1] == "g_x64"  # ceil(25000/64) = 391 cells per chunk -> 64 chunks
    out3, _ = R.pseudobulk_group(vals, 4, "g", seed=5, stream=0)  # ceil(6/4) = 2 -> only 3 chunks
    assert out3.shape == (10, 3)
    order, ptr = R.pseudobulk_partition(6, 3, seed=5, stream=0)
    assert sorted(order.tolist()) == list(range(6)) and ptr.tolist() == [0, 2, 4, 6]
    from oracle import reo_numpy as rn
    assert np.array_equal(R.host_sums(vals, order, ptr), rn.pseudobulk(vals, order, ptr))


def test_writers(R, pkg, tmp_path):
    prep = {"data": np.array([[1, 2], [3, 4]]), "sample_names": ["s1", "s2"], "sample_groups": ["x", "y"],
            "gene_names": ["A", "B"], "g_name": ["x", "y"], "ref": np.array([True, True])}
    res = np.zeros((2, 15))
    res[0] = [0.001, 0.01, 3, 0, 1, 0, 5, 0, 0, 2, 1, 1.5, 1.25, 0.5, 3.0]
    res[1] = [0.5, 1.0, 0, 0, 0, 0, 9, 0, 0, 0, 0, 0, 0, 0, 0]
    run = pkg.DegRun(result=res, labels=np.array(["up", "no change"], dtype=object), levels=["x", "y"],
                     thresholds=None, iters_run=1, comparisons=[{"k": 0, "result": res, "labels": np.array(["up", "no change"], dtype=object)}])
    df = R.write_outputs("fn_expr", prep, run, str(tmp_path))
    assert list(df.columns) == ["gene_name", "x_vs_y"] and df["x_vs_y"].tolist() == ["up", "no change"]
    lines = (tmp_path / "fn_expr_x_y_result.tsv").read_text().splitlines()
    assert lines[0].split("\t") == ["genename"] + pkg.HEADER
    assert lines[1] == "A\t0.001\t0.01\t3.0\t0.0\t1.0\t0.0\t5.0\t0.0\t0.0\t2.0\t1.0\t1.5\t1.25\t0.5\t3.0\tup"
    assert (tmp_path / "fn_expr_df_expr.tsv").read_text().splitlines()[:2] == ["genename\ts1\ts2", "A\t1\t2"]
    assert (tmp_path / "fn_expr_df_meta.tsv").read_text() == "Name\tGroup\ns1\tx\ns2\ty\n"
    assert (tmp_path / "fn_expr_gene_up_down.tsv").read_text().splitlines()[0] == "gene_name\tx_vs_y"


def test_julia_float_vector_table(R):
    """Byte-level parity of the float printer with the committed vector table (tests/golden/julia_float_vectors.json:
    Julia's shortest round-trip rule, the edge cases of the fixed / scientific switch, subnormals, signed zero, the
    tallies written as Float64, and the five numbers the reference itself prints at src/RankCompV3.jl:207-209)."""
    import json
    import struct
    table = json.load(open(os.path.join(GOLD, "julia_float_vectors.json")))
    assert len(table) >= 35
    for row in table:
        v = struct.unpack("<d", struct.pack("<Q", int(row["bits"], 16)))[0]
        assert R.julia_float(v) == row["text"], (row, R.julia_float(v))


def test_result_tsv_bytes_for_the_golden_slice(R, pkg, tmp_path):
    """The result file of :665-671 for the golden 64-gene slice, byte for byte against the committed fixture
    (header genename + 15 statistics + up_down, tab-delimited, tallies as Float64)."""
    import json
    g = json.load(open(os.path.join(GOLD, "bundled_slice64.json")))
    res = np.array(g["result"], dtype=np.float64)
    labels = pkg.label_genes(res, g["pval_deg"], g["padj_deg"])
    prep = {"g_name": ["group1", "group2"], "gene_names": ["slice%02d" % i for i in range(res.shape[0])],
            "sample_names": ["s%d" % s for s in range(len(g["gid"]))],
            "sample_groups": ["group%d" % (1 + v) for v in g["gid"]], "data": np.array(g["X"], dtype=np.int64)}

    class Run:
        comparisons = [{"k": 0, "result": res, "labels": labels}]

    R.write_outputs("bundled_slice64", prep, Run, str(tmp_path))
    got = (tmp_path / "bundled_slice64_group1_group2_result.tsv").read_bytes()
    assert got == open(os.path.join(GOLD, "bundled_slice64_result.tsv"), "rb").read()
    first = got.split(b"\n")[0].decode()
    assert first == "genename\tpval\tpadj\tn11\tn12\tn13\tn21\tn22\tn23\tn31\tn32\tn33\tΔ1\tΔ2\tse\tz1\tup_down"
    meta = (tmp_path / "bundled_slice64_df_meta.tsv").read_text().splitlines()
    assert meta[0] == "Name\tGroup" and len(meta) == 1 + len(g["gid"])


def test_df_meta_keeps_every_meta_column(R, tmp_path):
    """CSV.write(..., meta_group) at :680 writes the whole meta table, not just Name and Group."""
    e = _write(tmp_path, "e.tsv", "gene\ts1\ts2\ts3\ts4\nA\t1\t2\t3\t4\nB\t0\t0\t5\t6\n")
    m = _write(tmp_path, "m.tsv", "sample\tgrp\tbatch\tage\ns1\tx\tb1\t30\ns2\tx\tb2\t41\ns3\ty\tb1\t52\ns4\ty\tb2\t63\n")
    p = R.prepare(e, m, use_hk_genes="no")
    assert list(p["meta"].columns) == ["Name", "Group", "batch", "age"] and p["meta"]["age"].tolist() == [30, 41, 52, 63]


def test_df_meta_keeps_the_meta_files_own_row_and_column_order(R, tmp_path):
    """meta_group is `meta` minus the rows of the profiles that the min_profiles filter dropped (:612-623): the meta
    file's own row order (here not the expression matrix's column order) and column order survive."""
    e = _write(tmp_path, "e.tsv", "gene\ts1\ts2\ts3\ts4\nA\t1\t0\t3\t4\nB\t1\t0\t5\t6\nC\t2\t0\t1\t1\n")
    m = _write(tmp_path, "m.tsv", "sample\tgrp\tbatch\ns4\ty\tb2\ns2\tx\tb2\ns1\tx\tb1\ns3\ty\tb1\n")
    with pytest.raises(R.ArgumentError, match="align_meta"):   # the silent-wrong-answer case must be decided by the caller
        R.prepare(e, m, use_hk_genes="no")
    p = R.prepare(e, m, use_hk_genes="no", align_meta=False)   # s2 has no expressed gene: dropped by the profile filter
    # like the reference (:653) the groups are the meta rows' in THEIR order -- row t labels column t, names unchecked --
    # so this meta table mislabels s1 and s4 there and, with align_meta=False, here (a warning says so) ...
    assert p["sample_names"] == ["s1", "s3", "s4"] and p["sample_groups"] == ["y", "x", "y"]
    assert list(p["meta"].columns) == ["Name", "Group", "batch"]
    assert p["meta"]["Name"].tolist() == ["s4", "s1", "s3"] and p["meta"]["batch"].tolist() == ["b2", "b1", "b1"]
    # ... and align_meta=True (this build's extra) matches by sample name instead
    q = R.prepare(e, m, use_hk_genes="no", align_meta=True)
    assert q["sample_names"] == ["s1", "s3", "s4"] and q["sample_groups"] == ["x", "y", "y"]
    assert q["meta"]["Name"].tolist() == ["s4", "s1", "s3"]


def test_meta_rows_in_another_order_must_be_decided_and_fewer_rows_than_columns_fail(R, pkg, tmp_path, caplog):
    """:653 hands meta_group.Group to identify_degs as it stands: with a permuted meta table that mislabels samples without a
    word.  Here the caller has to choose (align_meta=None raises; False = the reference's behaviour + a warning; True = by name).
    A meta table that describes fewer profiles than the matrix has columns fails in identify_degs with DimensionMismatch (:355)."""
    import logging
    e = _write(tmp_path, "e.tsv", "gene\ts1\ts2\ts3\ts4\nA\t1\t2\t3\t4\nB\t1\t1\t5\t6\n")
    m = _write(tmp_path, "m.tsv", "sample\tgrp\ns2\tx\ns1\ty\ns3\tx\ns4\ty\n")
    with pytest.raises(R.ArgumentError, match="align_meta=True"):
        R.prepare(e, m, use_hk_genes="no")
    with caplog.at_level(logging.WARNING):
        p = R.prepare(e, m, use_hk_genes="no", align_meta=False)
    assert p["sample_groups"] == ["x", "y", "x", "y"] and any("not in the order" in r.message for r in caplog.records)
    assert R.prepare(e, m, use_hk_genes="no", align_meta=True)["sample_groups"] == ["y", "x", "x", "y"]   # s1 -> y, s2 -> x, by name
    ordered = _write(tmp_path, "m_ordered.tsv", "sample\tgrp\ns1\ty\ns2\tx\ns3\tx\ns4\ty\n")
    assert R.prepare(e, ordered, use_hk_genes="no")["sample_groups"] == ["y", "x", "x", "y"]             # in order: nothing to decide
    m2 = _write(tmp_path, "m2.tsv", "sample\tgrp\ns1\tx\ns3\ty\n")   # s2, s4 undescribed: 2 labels for 4 columns
    p2 = R.prepare(e, m2, use_hk_genes="no")
    assert len(p2["sample_groups"]) == 2 and p2["data"].shape[1] == 4
    with pytest.raises(pkg.DimensionMismatch):
        pkg.run_identify_degs(p2["data"], p2["sample_groups"], p2["gene_names"], 0.01, 1.0, 0.05, p2["ref"], 1, 1)  # (fails before any GPU call)


def test_hk_table_is_found_in_a_reference_checkout(R, tmp_path, monkeypatch):
    """use_hk_genes="yes" without hk_file: REO_HK_FILE, then hk_gene_file/HK_genes_info.tsv under REO_REFERENCE_DIR (:546)."""
    genes = [f"ENSG{i:05d}" for i in range(200)]
    rows = "\n".join(f"{g}\t" + "\t".join(str((i * 7 + s) % 11 + 1) for s in range(4)) for i, g in enumerate(genes))
    e = _write(tmp_path, "e.tsv", "Name\ts1\ts2\ts3\ts4\n" + rows + "\n")
    m = _write(tmp_path, "m.tsv", "Name\tGroup\ns1\tx\ns2\tx\ns3\ty\ns4\ty\n")
    (tmp_path / "ref" / "hk_gene_file").mkdir(parents=True)
    (tmp_path / "ref" / "hk_gene_file" / "HK_genes_info.tsv").write_text("Name\tENSEMBL\n" + "\n".join(f"n{i}\t{g}" for i, g in enumerate(genes[:120])) + "\n")
    monkeypatch.delenv("REO_HK_FILE", raising=False)
    monkeypatch.setenv("REO_REFERENCE_DIR", str(tmp_path / "ref"))
    p = R.prepare(e, m)
    assert p["ref"].sum() == 120 and p["ref"][:120].all()
