"""`reoa(...)`: the reference's user entry point (/root/reference/src/RankCompV3.jl:536-685) around the HIP hot path.

Same positional arguments and keywords, same return value (a `gene_name | <g1>_vs_<g2>` DataFrame, :682-684)
and the same TSV files in `work_dir` (:663-683).  Everything numeric between the preprocessing and the writers
is `identify_degs` on the GPU.  Differences, all deliberate and listed in DESIGN.md:
  * plots (`plot_result`, `plot_heatmap`, :672-675) are not produced;
  * `.rds` / `.RData` inputs are refused (no R reader here); delimited text is read with pandas;
  * the reference's unseeded `sample(...)` calls (:62,635,647) use a counter-based RNG keyed by `seed`;
  * the house-keeping gene table is a data asset of the reference package and is not shipped: pass
    `hk_file=` (or set REO_HK_FILE); without it `use_hk_genes="yes"` falls back to the random reference set with a WARN.
"""
from __future__ import annotations

import logging
import math
import os

import numpy as np

from . import _ffi, synth
from .hotpath import HEADER, run_identify_degs

log = logging.getLogger("reo_hip")


class ArgumentError(ValueError):
    """The reference's ArgumentError paths (src/RankCompV3.jl:565-566,574,581,584,590,603,638-639)."""


def julia_float(x: float) -> str:
    """Shortest round-trip decimal the way Julia prints a Float64 (CSV.write, :671): fixed notation for
    1e-4 <= |x| < 1e6 with at least one decimal, else `d.ddde[-]X`."""
    if x != x:
        return "NaN"
    if math.isinf(x):
        return "Inf" if x > 0 else "-Inf"
    if x == 0:
        return "-0.0" if math.copysign(1.0, x) < 0 else "0.0"
    r = repr(float(x))
    mant, _, exp = r.partition("e")
    sign = "-" if mant.startswith("-") else ""
    mant = mant.lstrip("-")
    ip, _, fp = mant.partition(".")
    digits = (ip + fp).lstrip("0")
    # decimal exponent of the first significant digit
    if exp:
        e10 = int(exp) + len(ip) - 1
    elif ip.strip("0"):
        e10 = len(ip.lstrip("0")) - 1
    else:
        e10 = -(len(fp) - len(fp.lstrip("0")) + 1)
    digits = digits.rstrip("0") or "0"
    if -4 <= e10 < 6:
        if e10 >= 0:
            whole = digits[: e10 + 1].ljust(e10 + 1, "0")
            frac = digits[e10 + 1:] or "0"
            return f"{sign}{whole}.{frac}"
        return f"{sign}0.{'0' * (-e10 - 1)}{digits}"
    frac = digits[1:] or "0"
    return f"{sign}{digits[0]}.{frac}e{e10}"


def _read_table(path: str):
    import pandas as pd
    if ".rds" in path or ".RData" in path:
        raise ArgumentError(f"{path}: .rds/.RData inputs are not supported by this build (no R reader); use delimited text")
    with open(path) as f:
        head = f.readline()
    sep = "\t" if head.count("\t") >= head.count(",") and "\t" in head else ("," if "," in head else r"\s+")
    return pd.read_csv(path, sep=sep, engine="python" if sep == r"\s+" else "c")


def shuffled(n: int, seed: int, stream: int) -> np.ndarray:
    """Permutation of 0..n-1 from the counter RNG (stands in for sample(1:n, n, replace=false), :62)."""
    h = synth.u64(seed + 0x100 * (stream + 1), np.arange(n), 7)
    return np.argsort(h, kind="stable")


def pseudobulk_partition(c: int, n_pseudo: int, seed: int, stream: int):
    """The reference's shuffled partition of one group's c cells (src/RankCompV3.jl:60-62): chunks of
    ceil(c / n_pseudo) cells of sample(1:c, c, replace=false).  Returns (order, chunk_ptr)."""
    cp = math.ceil(c / n_pseudo)
    if cp <= 1:
        log.info("WARN: too few profiles to generate %d pseudo-bulk profiles for the 'group' group", n_pseudo)
    order = shuffled(c, seed, stream)
    ptr = list(range(0, c, cp)) + [c]  # Iterators.partition
    return order.astype(np.int32), np.asarray(ptr, dtype=np.int32)


def host_sums(values, order, chunk_ptr) -> np.ndarray:
    """Left-to-right row sums of groups of cells on the host (used by the CPU-side unit tests only;
    `reoa` itself runs the HIP kernel, Context.pseudobulk)."""
    values = np.asarray(values)
    out = np.zeros((values.shape[0], len(chunk_ptr) - 1), dtype=values.dtype)
    for o in range(len(chunk_ptr) - 1):
        for c in order[chunk_ptr[o]: chunk_ptr[o + 1]]:
            out[:, o] = out[:, o] + values[:, c]
    return out


def pseudobulk_group(values: np.ndarray, n_pseudo: int, g_name: str, seed: int, stream: int, sums=host_sums):
    """pseudobulk_group, src/RankCompV3.jl:56-67: shuffle the group's cells, cut into chunks of
    ceil(c / n_pseudo) cells, sum each chunk.  Returns (matrix r x chunks, column names `<g>_x<k>`)."""
    order, ptr = pseudobulk_partition(values.shape[1], n_pseudo, seed, stream)
    out = sums(values, order, ptr)
    return out, [f"{g_name}_x{k + 1}" for k in range(len(ptr) - 1)]


def prepare(fn_expr: str, fn_meta: str, *, min_profiles: int = 0, min_features: int = 0, n_pseudo: int = 0,
            use_hk_genes: str = "yes", hk_file: str | None = None, gene_name_type: str = "ENSEMBL",
            ref_gene_max: int = 3000, ref_gene_min: int = 100, seed: int = 0, sums=host_sums, align_meta: bool | None = None):
    """Everything `reoa` does before `identify_degs` (:565-651).  Returns a dict with the expression
    matrix (genes x samples), the sample table (Name, Group), gene names, group levels and the reference mask.
    `align_meta` (not in the reference) only matters when the rows of the meta table are NOT in the order of the matrix's columns.
    The reference then labels column t with the group of meta row t whatever the names say (:653) -- mislabelled samples, wrong
    DEGs, no message.  Here that case must be decided by the caller: None (the default) raises ArgumentError and names the two
    choices; False = the reference's behaviour (row t labels column t), with a warning; True = look every column's group up by
    sample name."""
    import pandas as pd
    if not (os.path.isfile(fn_expr) and os.path.isfile(fn_meta)):  # :565
        raise ArgumentError(f"{fn_expr}, or {fn_meta}, does not exist or is not a regular file.")
    if not (os.path.getsize(fn_expr) > 0 and os.path.getsize(fn_meta) > 0):  # :566
        raise ArgumentError(f"{fn_expr}, or {fn_meta}, has size 0.")
    expr, meta = _read_table(fn_expr), _read_table(fn_meta)
    if meta.shape[1] < 2:  # :574
        raise ArgumentError(f"{fn_meta} the file for meta data, has only 0 or 1 column.")
    ecols = list(expr.columns)
    if not {"Name", "Group"} <= set(meta.columns) and set(meta.iloc[:, 0]) <= set(ecols):  # :575-578
        meta = meta.rename(columns={meta.columns[0]: "Name", meta.columns[1]: "Group"})
    if not {"Name", "Group"} <= set(meta.columns) or not set(meta["Name"]) <= set(ecols):  # :580-582
        raise ArgumentError(f"Meta data file, {fn_meta}, does not fit with the expression file, {fn_expr}. Some sample "
                            "names in the meta are not found in the column names of the expression matrix")
    if len(set(ecols)) != len(ecols):  # :583-585
        raise ArgumentError("Duplicate column names exist in the representation matrix.")
    meta = meta.copy()
    meta["Group"] = meta["Group"].astype(str)  # string.(categorical(...)), :586
    g_name = list(dict.fromkeys(meta["Group"]))  # unique, first appearance (:587)
    if len(g_name) < 2:  # :589-591
        raise ArgumentError(f"Meta data file, {fn_meta} has only 0 or 1 group. It must consist of two 'Group' levels")
    log.info("INFO: According to the meta information, there are %d groups of data and each group will be analyzed "
             "with the rest of the sample.", len(g_name))
    if "Name" not in ecols and ecols[0] not in set(meta["Name"]):  # :595-598
        expr = expr.rename(columns={ecols[0]: "Name"})
    expr = expr.dropna()  # dropmissing!, :601
    numeric = [c for c in expr.columns if pd.api.types.is_numeric_dtype(expr[c])]
    if not set(meta["Name"]) <= set(numeric):  # :602-604
        raise ArgumentError(f"{fn_expr} expression matrix contains non-numeric (Number) profiles.")
    gene_names = [str(v) for v in expr["Name"]] if "Name" in expr.columns else [str(v) for v in expr.iloc[:, 0]]
    if n_pseudo > 0:  # :608-612
        mats, names, groups = [], [], []
        for gi, g in enumerate(g_name):
            cols = list(meta["Name"][meta["Group"] == g])
            m, nm = pseudobulk_group(expr[cols].to_numpy(), n_pseudo, g, seed, gi, sums)
            mats.append(m); names += nm; groups += [g] * len(nm)
        data = np.concatenate(mats, axis=1)
        sample_names, sample_groups = names, groups
    else:
        data_cols = list(expr.columns[1:])  # expr[:, 2:end], :614
        data = expr[data_cols].to_numpy()
        sample_names = data_cols
        sample_groups = None  # from the meta rows, below (:613,619-624,653)
    data = np.asarray(data)
    all_names = list(sample_names)
    s_inds = (data > 0).sum(axis=0) > min_profiles  # :618
    data = data[:, s_inds]
    sample_names = [n for n, k in zip(sample_names, s_inds) if k]
    if sample_groups is not None:
        sample_groups = [g for g, k in zip(sample_groups, s_inds) if k]
    else:
        # meta_group = meta minus the rows whose FIRST column names a dropped profile (:619-624), and identify_degs gets
        # meta_group.Group as it stands (:653): row t of the meta table labels column t of the matrix, whatever the names
        # say.  A meta table in another order than the columns therefore mislabels samples in the reference; this build
        # does the same and says so (align_meta=True looks the groups up by name instead).
        dropped = set(all_names) - set(sample_names)
        kept = meta[~meta.iloc[:, 0].isin(dropped)]
        if align_meta:
            by_name = dict(zip(meta["Name"], meta["Group"]))
            sample_groups = [by_name.get(c) for c in sample_names]
            if any(g is None for g in sample_groups):
                raise ArgumentError("Expression matrix has sample columns that the meta data does not describe")
        else:
            sample_groups = list(kept["Group"])
            if list(kept["Name"]) != sample_names and len(kept) == len(sample_names) and set(kept["Name"]) == set(sample_names):
                # the same samples in another order: the reference would silently label column t with meta row t (:653)
                if align_meta is None:
                    raise ArgumentError("the rows of the meta table are not in the order of the expression matrix's columns.  The reference "
                                        "(src/RankCompV3.jl:653) labels column t with the group of meta row t whatever the sample names say, "
                                        "which mislabels samples here.  Pass align_meta=True to match samples by name, or align_meta=False "
                                        "to do exactly what the reference does.")
                log.warning("WARN: the rows of the meta table are not in the order of the expression matrix's columns; like the "
                            "reference (src/RankCompV3.jl:653) the group of row t labels column t (align_meta=False).")
            elif list(kept["Name"]) != sample_names:
                log.warning("WARN: the meta table does not describe the expression matrix's columns one to one; like the reference "
                            "(src/RankCompV3.jl:653) the group of row t labels column t.  Pass align_meta=True to match samples by name instead.")
    inds = (data > 0).sum(axis=1) > min_features  # :626
    gene_names = [n for n, k in zip(gene_names, inds) if k]
    data = data[inds, :]
    log.info("INFO: size after filtering lowly expressed genes and profiles and pseudo-bulk sampling, %s", data.shape)
    G = len(gene_names)

    def random_ref():  # sample(gene_names, min(length, ref_gene_max), replace=false), :635,647
        return synth.ref_mask(G, min(G, ref_gene_max), seed)

    ref = random_ref()
    if use_hk_genes == "yes":  # :636-650
        hk = hk_file or os.environ.get("REO_HK_FILE")
        if hk is None:  # the reference's default: hk_gene_file/HK_genes_info.tsv of its own checkout (:546)
            for root in (os.environ.get("REO_REFERENCE_DIR"), os.path.dirname(os.path.dirname(os.path.abspath(__file__)))):
                cand = os.path.join(root, "hk_gene_file", "HK_genes_info.tsv") if root else None
                if cand and os.path.isfile(cand):
                    hk = cand
                    break
        if hk is None:
            log.warning("no house-keeping gene table given (hk_file= / REO_HK_FILE / REO_REFERENCE_DIR): the reference reads its bundled "
                        "hk_gene_file/HK_genes_info.tsv here (src/RankCompV3.jl:546,641-649), a data asset that is not shipped with "
                        "this build.  Using the random reference set instead (what the reference does when fewer than "
                        "ref_gene_min house-keeping genes match); pass use_hk_genes=\"no\" to silence this.")
        else:
            if not os.path.isfile(hk):  # :638
                raise ArgumentError(f"{hk} does not exist or is not a regular file.")
            if os.path.getsize(hk) == 0:  # :639
                raise ArgumentError(f"{hk} for house-keeping genes has size 0.")
            tab = pd.read_csv(hk, sep="\t", dtype=str)
            if gene_name_type in tab.columns:  # :642-649
                hkset = set(tab[gene_name_type].dropna())
                mask = np.array([g in hkset for g in gene_names], dtype=bool)
                if mask.sum() < ref_gene_min:
                    log.info("WARN: only %d house-keeping genes are available, we just ignore this.", int(mask.sum()))
                else:
                    ref = mask
    # meta_group as the reference writes it (:680): every column of the meta table for the kept samples; after
    # pseudo-bulking the table is rebuilt from the new profile names (:612-613) and has the two columns only
    if n_pseudo > 0:
        meta_out = pd.DataFrame({"Name": sample_names, "Group": sample_groups})
    else:  # `meta` itself, minus the rows whose first column names a dropped profile (:612-623): row and column order kept
        dropped = set(all_names) - set(sample_names)
        meta_out = meta[~meta.iloc[:, 0].isin(dropped)].reset_index(drop=True)
    return {"data": data, "sample_names": sample_names, "sample_groups": sample_groups, "gene_names": gene_names,
            "g_name": g_name, "ref": ref, "meta": meta_out}


def write_outputs(stem: str, prep: dict, run, work_dir: str = "."):
    """The reference's result files (:663-683) minus the plots.  Returns the gene_up_down DataFrame (:682-684)."""
    import pandas as pd
    g_name, genes = prep["g_name"], prep["gene_names"]
    mg = len(g_name)
    for cm in run.comparisons:  # :666-677
        fg = "_".join([g_name[0], g_name[1]]) if mg == 2 else g_name[cm["k"]]
        path = os.path.join(work_dir, f"{stem}_{fg}_result.tsv")
        with open(path, "w") as f:
            f.write("\t".join(["genename"] + HEADER) + "\n")
            res, lab = cm["result"], cm["labels"]
            for i, gname in enumerate(genes):
                f.write(gname + "\t" + "\t".join(julia_float(v) for v in res[i]) + "\t" + lab[i] + "\n")
    with open(os.path.join(work_dir, f"{stem}_df_expr.tsv"), "w") as f:  # :678-679
        f.write("\t".join(["genename"] + prep["sample_names"]) + "\n")
        data = prep["data"]
        isint = np.issubdtype(data.dtype, np.integer)
        for i, gname in enumerate(genes):
            f.write(gname + "\t" + "\t".join(str(int(v)) if isint else julia_float(float(v)) for v in data[i]) + "\n")
    meta_out = prep.get("meta")
    if meta_out is None:
        meta_out = pd.DataFrame({"Name": prep["sample_names"], "Group": prep["sample_groups"]})
    meta_out.to_csv(os.path.join(work_dir, f"{stem}_df_meta.tsv"), sep="\t", index=False)  # :680, every meta column
    cols = [f"{g_name[0]}_vs_{g_name[1]}"] if mg == 2 else [f"{g}_vs_other" for g in g_name]  # :683
    df = pd.DataFrame({"gene_name": genes})
    for cname, cm in zip(cols, run.comparisons):
        df[cname] = cm["labels"]
    df.to_csv(os.path.join(work_dir, f"{stem}_gene_up_down.tsv"), sep="\t", index=False)
    return df


def reoa(fn_expr: str = "fn_expr.txt", fn_meta: str = "fn_meta.txt", *, expr_threshold=0, min_profiles: int = 0,
         min_features: int = 0, pval_reo: float = 0.01, pval_deg: float = 1.0, padj_deg: float = 0.05,
         n_pseudo: int = 0, use_hk_genes: str = "yes", hk_file: str | None = None, gene_name_type: str = "ENSEMBL",
         ref_gene_max: int = 3000, ref_gene_min: int = 100, n_iter: int = 128, n_conv: int = 5, work_dir: str = "./",
         use_testdata: str = "no", seed: int = 0, device: int = -1, testdata_dir: str | None = None, align_meta: bool | None = None):
    """reoa(fn_expr, fn_meta; kwargs...) -- src/RankCompV3.jl:536-555.  `expr_threshold` is accepted and
    unused, as in the reference (:539).  Extra keywords: `seed` (the reference's RNG is unseeded),
    `device`, `testdata_dir` (where fn_expr.txt / fn_meta.txt of the reference's test/ directory live), `align_meta`
    (match samples to meta rows by name; the reference and the default take the meta rows in their own order, :653)."""
    work_dir = os.path.abspath(work_dir)
    if use_testdata == "yes":  # :559-562
        d = testdata_dir or os.environ.get("REO_TESTDATA_DIR") or os.path.join(
            os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
        fn_expr, fn_meta = os.path.join(d, "fn_expr.txt"), os.path.join(d, "fn_meta.txt")
    elif not os.path.isabs(fn_expr):
        fn_expr, fn_meta = os.path.join(work_dir, fn_expr), os.path.join(work_dir, fn_meta)  # cd(work_dir), :557
    stem = os.path.splitext(os.path.basename(fn_expr))[0]  # :567

    def gpu_sums(values, order, chunk_ptr):  # pseudobulk_group's sums on the GPU (:63)
        with _ffi.Context(device=device, seed=seed) as pctx:
            return pctx.pseudobulk(values, order, chunk_ptr)

    prep = prepare(fn_expr, fn_meta, min_profiles=min_profiles, min_features=min_features, n_pseudo=n_pseudo,
                   use_hk_genes=use_hk_genes, hk_file=hk_file, gene_name_type=gene_name_type,
                   ref_gene_max=ref_gene_max, ref_gene_min=ref_gene_min, seed=seed, sums=gpu_sums, align_meta=align_meta)
    run = run_identify_degs(prep["data"], prep["sample_groups"], prep["gene_names"], pval_reo, pval_deg, padj_deg,
                            prep["ref"], n_iter, n_conv, seed=seed, device=device)  # :652-662
    for p, (d, n) in enumerate(run.trace):
        log.info("INFO: iteration %d,  # DEGs %d, # non-DEGs %d", p, d, n)  # :418
    df = write_outputs(stem, prep, run, work_dir)
    df.attrs["run"] = run
    return df
