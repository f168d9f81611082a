"""Synthetic expression matrices for parity tests and bench.py.

Integer-only and counter-based (SplitMix64 of (seed, gene, sample)), so any
implementation reproduces the same matrix without sharing a library RNG.
Two families: T0 tie-free ranks, T1 tie-rich counts.  Groups
are contiguous: the first S//2 samples are "group1", the rest "group2".
"""
from __future__ import annotations

import numpy as np

_U = np.uint64


def mix64(z: np.ndarray) -> np.ndarray:
    z = np.asarray(z, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = z + _U(0x9E3779B97F4A7C15)
        z = (z ^ (z >> _U(30))) * _U(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> _U(27))) * _U(0x94D049BB133111EB)
        return z ^ (z >> _U(31))


def u64(seed: int, g: np.ndarray, s: np.ndarray) -> np.ndarray:
    g = np.asarray(g, dtype=np.uint64)
    s = np.asarray(s, dtype=np.uint64)
    return mix64(_U(seed & 0xFFFFFFFFFFFFFFFF) ^ mix64((g << _U(32)) | s))


def groups(S: int):
    g = np.array(["group1"] * (S // 2) + ["group2"] * (S - S // 2), dtype=object)
    return g


def _effects(seed: int, G: int) -> np.ndarray:
    """10 % of genes shifted in group2 (half up, half down), magnitude 1..4 base units."""
    h = u64(seed + 2, np.arange(G), 0)
    sel = (h % _U(20)).astype(np.int64)
    mag = (1 + ((h >> _U(8)) % _U(4))).astype(np.int64)
    return np.where(sel == 0, mag, np.where(sel == 1, -mag, 0))


def t0_ranks(G: int, S: int, seed: int) -> np.ndarray:
    """Tie-free: every sample column is a permutation of 0..G-1 (Int64)."""
    g = np.arange(G)[:, None]
    s = np.arange(S)[None, :]
    base = (u64(seed + 1, np.arange(G), 0) % _U(64)).astype(np.int64)[:, None]
    eff = _effects(seed, G)[:, None] * (s >= S // 2)
    noise = (u64(seed, g, s) >> _U(44)).astype(np.int64)  # 20 bits = 4 base units
    v = (base + eff) * (1 << 18) + noise
    # rank by (v, gene): stable argsort over the gene axis
    order = np.argsort(v, axis=0, kind="stable")
    X = np.empty((G, S), dtype=np.int64)
    np.put_along_axis(X, order, np.broadcast_to(np.arange(G, dtype=np.int64)[:, None], (G, S)), axis=0)
    return X


def t1_counts(G: int, S: int, seed: int) -> np.ndarray:
    """Tie-rich zero-inflated integer counts (Int64): ~9 % zeros, heavy-tailed gene scales."""
    g = np.arange(G)[:, None]
    s = np.arange(S)[None, :]
    h = u64(seed, g, s)
    zero = (h & _U(0xFF)) < _U(23)
    shift = (u64(seed + 1, np.arange(G), 0) % _U(12)).astype(np.int64)[:, None]
    eff = _effects(seed, G)[:, None] * (s >= S // 2)
    sh = np.clip(shift + np.sign(eff), 0, 13)
    frac = ((h >> _U(8)) & _U(0xFFFF)).astype(np.int64)
    X = (frac << sh) >> 12
    X[zero] = 0
    return X.astype(np.int64)


def float_expr(G: int, S: int, seed: int) -> np.ndarray:
    """Float64 log-like expression with values closer than 0.1 apart (exercises the tie band)."""
    X = t1_counts(G, S, seed).astype(np.float64)
    frac = (u64(seed + 7, np.arange(G)[:, None], np.arange(S)[None, :]) >> _U(54)).astype(np.float64) / 1024.0
    return np.log2(1.0 + X) + 0.05 * frac


def ref_mask(G: int, n: int, seed: int) -> np.ndarray:
    """n reference genes chosen by the counter RNG (stands in for sample(), src/RankCompV3.jl:635)."""
    h = u64(seed + 3, np.arange(G), 0)
    order = np.argsort(h, kind="stable")
    m = np.zeros(G, dtype=bool)
    m[order[: min(n, G)]] = True
    return m


def lu_corner(b: int, n13: int, spg: int = 2, corners: int = 1, n_always: int = 3, n_unstable: int = 2, seed: int = 0):
    """A small problem whose `corners` first genes (before the shuffle) have a 3x3 table with
    n12 = n21 = n23 = n32 = 0 and n13 + n31 = b against the full reference set: N = [[b b][b b]], the
    integer-singular table that the reference's float test abs(det(N)) <= eps() (src/RankCompV3.jl:242)
    calls NON-singular whenever b * (1.0 / b) != 1 (b = 49, 98, 103, 107, 161, ...).
    `spg` samples per group; every designed ordering is unanimous, hence stable at any threshold.
    Returns (X Int64 G x 2*spg, group labels, indices of the corner genes after the shuffle)."""
    assert 0 <= n13 <= b and spg >= 2 and corners >= 1
    S = 2 * spg
    treat = np.arange(S) >= spg
    rows = []
    for c in range(corners):                   # below the n13 partners in ctrl, above them in treat
        rows.append(np.where(treat, 100000 + 10 * c, 10 * c))
    for j in range(n13):                       # constant: class (1, 3) against a corner gene
        rows.append(np.full(S, 40000 + j))
    for j in range(b - n13):                   # below every corner in ctrl, above in treat: class (3, 1)
        rows.append(np.where(treat, 200000 + j, -100000 - j))
    for j in range(n_always):                  # always above / always below: n11 / n33
        rows.append(np.full(S, (500000 + j) if j % 2 == 0 else (-500000 - j)))
    if spg % 2 == 0:
        for j in range(n_unstable):            # alternating around a corner gene in both groups: n22
            alt = np.where(np.arange(S) % 2 == 0, -(5 + j), 5 + j)
            rows.append(np.where(treat, 100000, 0) + 10 * (corners - 1) * (np.arange(S) % 2) + alt * (10 * corners))
    X = np.array(rows, dtype=np.int64)
    G = X.shape[0]
    perm = np.argsort(u64(seed + 11, np.arange(G), 0), kind="stable")
    inv = np.empty(G, dtype=np.int64)
    inv[perm] = np.arange(G)
    group = np.array(["ctrl"] * spg + ["treat"] * spg, dtype=object)
    return X[perm], group, inv[:corners]


def with_infinities(X: np.ndarray, seed: int, kind: str = "log0") -> np.ndarray:
    """Float64 copy of X with infinities where log-transformed tables have them (is_greater on them: src/RankCompV3.jl:71-77).
    log0: the smallest values of every sample become -Inf (log(0)), a few entries +Inf;  column: one whole sample -Inf too;
    group: -Inf in the first half of the samples only;  rows: whole genes -Inf / +Inf."""
    X = np.array(X, dtype=np.float64)
    G, S = X.shape
    h = u64(seed + 21, np.arange(G)[:, None], np.arange(S)[None, :])
    lowq = np.quantile(X, 0.12, axis=0, keepdims=True)
    if kind in ("log0", "column", "rows"):
        X[X <= lowq] = -np.inf
    elif kind == "group":
        X[:, : S // 2][(X <= lowq)[:, : S // 2]] = -np.inf
    X[(h % _U(97)) == _U(0)] = np.inf
    if kind == "column":
        X[:, S // 3] = -np.inf
        X[:, S - 1] = np.inf
    if kind == "rows":
        X[(h[:, 0] % _U(23)) == _U(1), :] = -np.inf
        X[(h[:, 0] % _U(23)) == _U(2), :] = np.inf
    return X
