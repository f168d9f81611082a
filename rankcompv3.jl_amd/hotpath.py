"""Host-side mirror of the reference's hot-path interface.

`identify_degs` has the reference's positional signature
(/root/reference/src/RankCompV3.jl:339-350) and returns the same G x 17
matrix of [gene_name, 15 Float64 statistics, label] (:430,437); everything
numeric happens in libreo_hip.so on the GPU.  The Julia shim a maintainer
would drop into the reference (julia/RankCompV3HIP.jl) is the same ~60 lines
in Julia.
"""
from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np

from . import _ffi

HEADER = ["pval", "padj", "n11", "n12", "n13", "n21", "n22", "n23", "n31", "n32", "n33",
          "Δ1", "Δ2", "se", "z1", "up_down"]  # src/RankCompV3.jl:665


def encode_groups(group):
    """unique(group) in first-appearance order (src/RankCompV3.jl:353,357) -> (ids, levels)."""
    levels: list = []
    index: dict = {}
    ids = np.empty(len(group), dtype=np.int32)
    for s, g in enumerate(group):
        key = g.item() if isinstance(g, np.generic) else g
        if key not in index:
            index[key] = len(levels)
            levels.append(key)
        ids[s] = index[key]
    return ids, levels


def label_genes(result: np.ndarray, pval_deg: float, padj_deg: float) -> np.ndarray:
    """up / down / no change, src/RankCompV3.jl:426-429."""
    sig = (result[:, 0] <= pval_deg) & (result[:, 1] <= padj_deg)
    out = np.full(result.shape[0], "no change", dtype=object)
    out[sig & (result[:, 14] > 0)] = "up"
    out[sig & (result[:, 14] < 0)] = "down"
    return out


@dataclass
class DegRun:
    """Everything the comparisons produced (the reference only keeps `res`)."""
    result: np.ndarray              # G x 15 Float64 of the first comparison
    labels: np.ndarray              # labels of the first comparison
    levels: list
    thresholds: np.ndarray          # 2 x ngroups (:362)
    iters_run: int
    trace: list = field(default_factory=list)   # (#DEG, #non-DEG) per pass (:418), first comparison
    timings: dict = field(default_factory=dict)
    info: dict = field(default_factory=dict)
    comparisons: list = field(default_factory=list)  # per comparison: dict(k, result, labels, iters_run, trace)
    gene_names: object = None
    _res: object = None

    @property
    def res(self) -> np.ndarray:
        """G x (1 + 16 C) object matrix, the reference's return value (:430,437).  Built on first use: boxing
        20 000 x 15 floats costs 10 ms of host time, more than the GPU spends on a small problem."""
        if self._res is None:
            r = self.result.shape[0]
            res = np.empty((r, 1 + 16 * len(self.comparisons)), dtype=object)  # hcat(res, result, gene_up_down) per comparison
            res[:, 0] = np.asarray(self.gene_names, dtype=object)
            for q, cm in enumerate(self.comparisons):
                res[:, 1 + 16 * q: 16 + 16 * q] = cm["result"]
                res[:, 16 + 16 * q] = cm["labels"]
            self._res = res
        return self._res


def run_identify_degs(data, group, gene_names, pval_reo, pval_deg, padj_deg, ref_gene, n_iter, n_conv, *,
                      seed: int = 0, device: int = -1, shard=(0, 1), allreduce=None, allgather=None, profile: bool = False) -> DegRun:
    """identify_degs with the extras (trace, timings) kept.  Two groups: one comparison, group 1 vs
    group 2 (the reference's `gnum == 2` path, :387-389,431-434).  More groups: one comparison per
    group, that group vs every other sample (:375-390,396-436), 16 more columns each."""
    data = np.asarray(data)
    if data.ndim != 2:
        raise _ffi.DimensionMismatch(_ffi.REO_EINVAL, "'data' must be a genes x samples matrix")
    r, c = data.shape
    if c != len(group):  # :355
        raise _ffi.DimensionMismatch(_ffi.REO_EINVAL, "'data' and 'group' do not have compatiable sizes")
    gid, levels = encode_groups(group)
    if len(levels) < 2:  # :356
        raise _ffi.DimensionMismatch(_ffi.REO_EINVAL, "Only 1 level in 'group1, at least 2 levels!")
    if len(gene_names) != r or len(ref_gene) != r:
        raise _ffi.DimensionMismatch(_ffi.REO_EINVAL, "gene_names / ref_gene length != number of rows of 'data'")
    ncomp = 1 if len(levels) == 2 else len(levels)
    comps = []
    with _ffi.Context(device=device, seed=seed) as ctx:
        ctx.set_profiling(profile)
        # groups, thresholds and sharding BEFORE the matrix: reo_set_matrix then ranks the samples as their columns arrive and, on
        # one GPU with two groups, starts the pair kernel's first side while the second group is still crossing PCIe
        ctx.set_groups(gid, len(levels))
        thr = ctx.compute_thresholds(pval_reo)
        if shard[1] > 1:
            ctx.set_shard(*shard)
            if allgather is not None:
                ctx.set_allgather(allgather)   # all-gather of the shards' own table words (what the in-library RCCL path does)
            else:
                ctx.set_allreduce(allreduce)   # in-place sum of the whole table
        ctx.set_matrix(data)
        for k in range(ncomp):  # `for k=1:gnum ... if gnum==2 break` (:396,431-434)
            ctx.build_pairs(k)
            result, iters, trace = ctx.identify_degs(np.asarray(ref_gene, dtype=bool), pval_deg, padj_deg, n_iter, n_conv)
            comps.append({"k": k, "result": result, "labels": label_genes(result, pval_deg, padj_deg),
                          "iters_run": iters, "trace": trace})
        timings = ctx.timings() if profile else {}
        info = ctx.info()
    first = comps[0]
    return DegRun(result=first["result"], labels=first["labels"], levels=levels, thresholds=thr, iters_run=first["iters_run"],
                  trace=first["trace"], timings=timings, info=info, comparisons=comps, gene_names=list(gene_names))


def identify_degs(data, group, gene_names, pval_reo, pval_deg, padj_deg, ref_gene, n_iter, n_conv, **kw) -> np.ndarray:
    """Drop-in for identify_degs (src/RankCompV3.jl:339-350): same arguments in the same order, same
    G x (1 + 16 C) return (C = 1 for two groups, else the number of groups).  `seed=` keys the tie
    coins that the reference draws from its unseeded global RNG (:73)."""
    return run_identify_degs(data, group, gene_names, pval_reo, pval_deg, padj_deg, ref_gene, n_iter, n_conv, **kw).res
