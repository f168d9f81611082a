"""Independent numpy/scipy restatement of the REO hot path (small cases only).

TEST INFRASTRUCTURE ONLY -- see oracle/reo_oracle.c.  This second restatement
exists so that the C oracle is cross-checked by code that shares nothing with
it except the reference lines both follow (all citations are relative to
/root/reference).  Vectorised over pairs; sizes of a few hundred genes.
"""
from __future__ import annotations

import numpy as np
from scipy import stats

M64 = (1 << 64) - 1


def mix64(z: int) -> int:
    z = (z + 0x9E3779B97F4A7C15) & M64
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & M64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & M64
    return z ^ (z >> 31)


def tie_wins(seed: int, i: int, j: int, g: int, n_eq: int) -> int:
    """Binomial(n_eq, 1/2) draw of the keyed coin stream (is_greater ties,
    src/RankCompV3.jl:72-73)."""
    base = (mix64(seed ^ mix64((i << 32) | j)) + (g << 40)) & M64
    wins, w = 0, 0
    while n_eq > 0:
        bits = mix64((base + w) & M64)
        take = min(64, n_eq)
        if take < 64:
            bits &= (1 << take) - 1
        wins += bin(bits).count("1")
        n_eq -= take
        w += 1
    return wins


def threshold(n: int, pval_reo: float = 0.01) -> int:
    """get_major_reo_lower_count, src/RankCompV3.jl:81-92."""
    d = stats.binom(n, 0.5)

    def pv(x):  # HypothesisTests two-sided p for a discrete distribution
        return min(1.0, 2.0 * min(d.sf(x - 1), d.cdf(x)))

    if pv(0) < pval_reo:
        for x in range(0, n // 2 + 1):
            if pv(x) > pval_reo:
                return n - x + 1
        raise ValueError("findfirst returned nothing")
    return n


def group_ids(group) -> tuple[np.ndarray, list]:
    """unique() in first-appearance order, src/RankCompV3.jl:353,357."""
    lev: list = []
    ids = np.empty(len(group), dtype=np.int32)
    for s, g in enumerate(group):
        if g not in lev:
            lev.append(g)
        ids[s] = lev.index(g)
    return ids, lev


def pair_counts(X: np.ndarray, gid: np.ndarray, ngroups: int):
    """Deterministic (n_gt, n_eq) per ordered pair and group; G x G x ngroups.
    Comparator of src/RankCompV3.jl:71-77 without the coin."""
    X = np.asarray(X, dtype=np.float64)
    G, S = X.shape
    n_gt = np.zeros((G, G, ngroups), dtype=np.int32)
    n_eq = np.zeros((G, G, ngroups), dtype=np.int32)
    for s in range(S):
        d = X[:, None, s] - X[None, :, s]
        tie = np.abs(d) < 0.1
        gt = (~tie) & (X[:, None, s] > X[None, :, s])
        n_gt[:, :, gid[s]] += gt
        n_eq[:, :, gid[s]] += tie
    return n_gt, n_eq


def build_codes(X, gid, ngroups, k, thr, seed):
    """REO table, src/RankCompV3.jl:363-392; 0..8 per ordered pair, 255 on
    the diagonal."""
    n_gt, n_eq = pair_counts(X, gid, ngroups)
    G = n_gt.shape[0]
    sizes = np.bincount(gid, minlength=ngroups)
    s1, s2 = int(sizes[k]), int(sizes.sum() - sizes[k])
    code = np.full((G, G), 255, dtype=np.uint8)
    for i in range(G):
        for j in range(i + 1, G):
            nre = [int(n_gt[i, j, g]) + (tie_wins(seed, i, j, g, int(n_eq[i, j, g])) if n_eq[i, j, g] else 0)
                   for g in range(ngroups)]
            a, b = nre[k], sum(nre) - nre[k]
            ic = 3 if a >= thr[0] else (1 if s1 - a >= thr[0] else 2)
            it = 3 if b >= thr[1] else (1 if s2 - b >= thr[1] else 2)
            c = 3 * (ic - 1) + (it - 1)
            code[i, j] = c
            code[j, i] = 8 - c
    return code


def tally(code: np.ndarray, ref: np.ndarray) -> np.ndarray:
    """src/RankCompV3.jl:403."""
    G = code.shape[0]
    out = np.zeros((G, 9), dtype=np.int32)
    sel = code[:, np.asarray(ref, dtype=bool)]
    for c in range(9):
        out[:, c] = (sel == c).sum(axis=1)
    return out


def mccullagh(mat: np.ndarray):
    """src/RankCompV3.jl:225-259; returns ((pval,d1,d2,se,z1), N, R)."""
    mat = np.asarray(mat, dtype=np.int64)
    k = mat.shape[0]
    m = k - 1
    N = np.zeros((m, m), dtype=np.int64)
    for i in range(m):
        for j in range(i, m):
            N[i, j] = N[j, i] = mat[: i + 1, j + 1:].sum() + mat[j + 1:, : i + 1].sum()
    n = np.diag(N).astype(np.float64)
    R = np.array([mat[: i + 1, i + 1:].sum() for i in range(m)], dtype=np.int64)
    Nf = N.astype(np.float64)
    if abs(np.linalg.det(Nf)) <= np.finfo(np.float64).eps:
        return (1.0, 0.0, 0.0, 0.0, 0.0), N, R
    w2 = np.linalg.inv(Nf) @ n
    nu = 1.0 / (n @ w2)
    w1 = (n * w2) * nu
    Rf = R.astype(np.float64)
    d1 = float(w1 @ np.log((Rf + 0.5) / (n - Rf + 0.5)))
    d2 = float(np.log((0.5 + w2 @ Rf) / (0.5 + w2 @ (n - Rf))))
    v1 = 4 * (1 + 0.25 * d1 ** 2) * nu
    v2 = 4 * (1 + 0.25 * d2 ** 2) * nu
    se = float(np.sqrt((v1 + v2) * 0.5))
    z1 = d1 / se
    p = min(1.0, 2.0 * min(stats.norm.cdf(z1), stats.norm.sf(z1)))
    return (p, d1, d2, se, z1), N, R


def jl_round(x: float) -> int:
    return int(np.rint(x))  # half-even, like Julia's round(Int, x)


def trimmed_std(d: np.ndarray) -> float:
    """src/RankCompV3.jl:409-411."""
    G = len(d)
    a, b = jl_round(G * 0.05), jl_round(G * 0.95)
    if a < 1 or b > G:
        raise IndexError("BoundsError in the reference")
    return float(np.std(np.sort(d)[a - 1: b], ddof=1))


def bh(p: np.ndarray) -> np.ndarray:
    """MultipleTesting BenjaminiHochberg (call site src/RankCompV3.jl:413)."""
    p = np.asarray(p, dtype=np.float64)
    n = len(p)
    if n <= 1:
        return p.copy()
    order = np.argsort(p, kind="stable")
    adj = p[order] * (n / np.arange(1, n + 1))
    adj = np.minimum.accumulate(adj[::-1])[::-1]
    out = np.empty(n)
    out[order] = np.minimum(adj, 1.0)
    return out


def iterate(code, ref0, pval_deg, padj_deg, n_iter, n_conv):
    """src/RankCompV3.jl:396-425; returns (result G x 15, passes, trace)."""
    G = code.shape[0]
    ref = np.asarray(ref0, dtype=bool).copy()
    result = np.zeros((G, 15))
    trace = []
    i_iter = 0
    while i_iter < n_iter:
        cont = tally(code, ref)
        for i in range(G):
            (p, d1, d2, se, z1), _, _ = mccullagh(cont[i].reshape(3, 3))
            result[i, 0] = p
            result[i, 1] = 1.0
            result[i, 2:11] = cont[i]
            result[i, 11:15] = (d1, d2, se, z1)
        se = trimmed_std(result[:, 11])
        with np.errstate(divide="ignore", invalid="ignore"):
            z = result[:, 11] / se
        if se == 0.0:  # StatsFuns normcdf/normccdf: sigma == 0 and x == mu -> z = +Inf
            z = np.where(result[:, 11] == 0.0, np.inf, z)
        pval = np.minimum(1.0, 2.0 * np.minimum(stats.norm.cdf(z), stats.norm.sf(z)))
        padj = bh(pval)
        result[:, 0], result[:, 1] = pval, padj
        inds = ~((pval <= pval_deg) & (padj <= padj_deg))
        trace.append((int(G - inds.sum()), int(inds.sum())))
        if abs(int(ref.sum()) - int(inds.sum())) < n_conv:
            break
        i_iter += 1
        ref = inds
    return result, len(trace), trace


def identify_degs(X, group, pval_reo, pval_deg, padj_deg, ref0, n_iter, n_conv, seed, k=0):
    gid, lev = group_ids(group)
    ng = len(lev)
    sizes = np.bincount(gid, minlength=ng)
    thr = (threshold(int(sizes[k]), pval_reo), threshold(int(sizes.sum() - sizes[k]), pval_reo))
    code = build_codes(X, gid, ng, k, thr, seed)
    return iterate(code, ref0, pval_deg, padj_deg, n_iter, n_conv)


def labels(result, pval_deg, padj_deg):
    """src/RankCompV3.jl:426-429."""
    sig = (result[:, 0] <= pval_deg) & (result[:, 1] <= padj_deg)
    out = np.array(["no change"] * result.shape[0], dtype=object)
    out[sig & (result[:, 14] > 0)] = "up"
    out[sig & (result[:, 14] < 0)] = "down"
    return out


def pseudobulk(values: np.ndarray, order, chunk_ptr) -> np.ndarray:
    """pseudobulk_group, src/RankCompV3.jl:56-67: profile o = row sums of the cells
    order[chunk_ptr[o]:chunk_ptr[o+1]], added left to right like `sum(eachrow(...))` (:63)."""
    values = np.asarray(values)
    n_out = len(chunk_ptr) - 1
    out = np.zeros((values.shape[0], n_out), dtype=values.dtype)
    for o in range(n_out):
        acc = np.zeros(values.shape[0], dtype=values.dtype)
        for c in order[chunk_ptr[o]: chunk_ptr[o + 1]]:
            acc = acc + values[:, c]
        out[:, o] = acc
    return out
