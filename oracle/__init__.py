"""CPU oracle for the REO hot path -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
import this package.  See oracle/reo_oracle.c for the parity statement.
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

_u8p = ctypes.POINTER(ctypes.c_uint8)
_u16p = ctypes.POINTER(ctypes.c_uint16)
_i32p = ctypes.POINTER(ctypes.c_int32)
_i64p = ctypes.POINTER(ctypes.c_int64)
_f64p = ctypes.POINTER(ctypes.c_double)


def build(force: bool = False) -> str:
    so = os.path.join(_HERE, "liboracle.so")
    srcs = [os.path.join(_HERE, "reo_oracle.c"), os.path.join(_HERE, "reo_tuned.c")]
    if force or not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(f) for f in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-B", "liboracle.so"], stdout=subprocess.DEVNULL)
    return so


def lib() -> ctypes.CDLL:
    global _LIB
    if _LIB is None:
        L = ctypes.CDLL(build())
        L.oracle_tie_wins.restype = ctypes.c_uint32
        L.oracle_tie_wins.argtypes = [ctypes.c_uint64, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32]
        L.oracle_threshold.restype = ctypes.c_int32
        L.oracle_threshold.argtypes = [ctypes.c_int32, ctypes.c_double]
        L.oracle_pair_counts.restype = None
        L.oracle_pair_counts.argtypes = [_f64p, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, _i32p, ctypes.c_int32,
                                         ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, _u16p, _u16p]
        L.oracle_build_codes.restype = None
        L.oracle_build_codes.argtypes = [_f64p, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, _i32p, ctypes.c_int32,
                                         ctypes.c_int32, _i32p, ctypes.c_uint64, _u8p]
        L.oracle_tally.restype = None
        L.oracle_tally.argtypes = [_u8p, ctypes.c_int64, _u8p, _i32p]
        L.oracle_mccullagh.restype = ctypes.c_int
        L.oracle_mccullagh.argtypes = [ctypes.c_int32, _i64p, _f64p, _i64p, _i64p]
        L.oracle_mccullagh9.restype = ctypes.c_int
        L.oracle_mccullagh9.argtypes = [_i32p, _f64p]
        L.oracle_trimmed_std.restype = ctypes.c_double
        L.oracle_trimmed_std.argtypes = [_f64p, ctypes.c_int64, _i32p]
        L.oracle_bh.restype = None
        L.oracle_bh.argtypes = [_f64p, ctypes.c_int64, _f64p]
        L.oracle_iter_stats.restype = ctypes.c_int64
        L.oracle_iter_stats.argtypes = [_i32p, ctypes.c_int64, ctypes.c_double, ctypes.c_double, _f64p, _u8p]
        L.oracle_iterate.restype = ctypes.c_int32
        L.oracle_iterate.argtypes = [_u8p, ctypes.c_int64, _u8p, ctypes.c_double, ctypes.c_double, ctypes.c_int32,
                                     ctypes.c_int32, _f64p, _i32p, _i32p]
        L.oracle_identify_degs.restype = ctypes.c_int32
        L.oracle_identify_degs.argtypes = [_f64p, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, _i32p, ctypes.c_int32,
                                           ctypes.c_int32, ctypes.c_double, ctypes.c_double, ctypes.c_double, _u8p,
                                           ctypes.c_int32, ctypes.c_int32, ctypes.c_uint64, _f64p, _i32p, _i32p]
        L.oracle_num_threads.restype = ctypes.c_int32
        _LIB = L
    return _LIB


def _p(a, t):
    return a.ctypes.data_as(t)


def _colmajor(X):
    X = np.asfortranarray(np.asarray(X, dtype=np.float64))
    return X, X.shape[0], X.shape[1]


def tie_wins(seed, i, j, g, n_eq):
    return int(lib().oracle_tie_wins(seed, i, j, g, n_eq))


def threshold(n, pval_reo=0.01):
    return int(lib().oracle_threshold(n, pval_reo))


def pair_counts(X, gid, ngroups, i0, i1, j0, j1):
    X, G, S = _colmajor(X)
    gid = np.ascontiguousarray(gid, dtype=np.int32)
    gt = np.zeros((i1 - i0, j1 - j0, ngroups), dtype=np.uint16)
    eq = np.zeros_like(gt)
    lib().oracle_pair_counts(_p(X, _f64p), G, S, G, _p(gid, _i32p), ngroups, i0, i1, j0, j1, _p(gt, _u16p), _p(eq, _u16p))
    return gt, eq


def pair_counts_as_evaluated(X, gid, ngroups, i0, i1, j0, j1):
    """The counts of pair (i, j) as the reference comes by them: it evaluates is_greater(x_i, x_j) for i < j only
    (src/RankCompV3.jl:366-372) and MIRRORS the class for (j, i) (:386), i.e. for i > j the counts are
    n_gt(i, j) = S_g - n_gt(j, i) - n_eq(j, i), n_eq(i, j) = n_eq(j, i).  Equal to pair_counts() -- the comparator applied to
    the ordered pair -- unless both values are the same infinity (is_greater(Inf, Inf) is false BOTH ways: neither tied
    nor greater).  The diagonal, which the reference never evaluates, is reported as tied in every sample."""
    gid = np.ascontiguousarray(gid, dtype=np.int32)
    gt, eq = pair_counts(X, gid, ngroups, i0, i1, j0, j1)
    tgt, teq = pair_counts(X, gid, ngroups, j0, j1, i0, i1)
    sizes = np.bincount(gid, minlength=ngroups).astype(np.int64)
    ii = np.arange(i0, i1)[:, None]
    jj = np.arange(j0, j1)[None, :]
    mgt = (sizes[None, None, :] - tgt.transpose(1, 0, 2).astype(np.int64) - teq.transpose(1, 0, 2)).astype(np.uint16)
    meq = teq.transpose(1, 0, 2)
    low = (ii > jj)[:, :, None]
    dia = (ii == jj)[:, :, None]
    gt = np.where(low, mgt, gt)
    eq = np.where(low, meq, eq)
    gt = np.where(dia, np.uint16(0), gt)
    eq = np.where(dia, sizes.astype(np.uint16)[None, None, :], eq)
    return gt.astype(np.uint16), eq.astype(np.uint16)


def build_codes(X, gid, ngroups, k, thr, seed):
    X, G, S = _colmajor(X)
    gid = np.ascontiguousarray(gid, dtype=np.int32)
    thr = np.ascontiguousarray(thr, dtype=np.int32)
    code = np.empty((G, G), dtype=np.uint8)
    lib().oracle_build_codes(_p(X, _f64p), G, S, G, _p(gid, _i32p), ngroups, k, _p(thr, _i32p), seed, _p(code, _u8p))
    return code


def tally(code, ref):
    code = np.ascontiguousarray(code, dtype=np.uint8)
    ref = np.ascontiguousarray(ref, dtype=np.uint8)
    G = code.shape[0]
    cont = np.empty((G, 9), dtype=np.int32)
    lib().oracle_tally(_p(code, _u8p), G, _p(ref, _u8p), _p(cont, _i32p))
    return cont


def mccullagh(mat):
    mat = np.ascontiguousarray(mat, dtype=np.int64)
    k = mat.shape[0]
    out = np.zeros(5)
    N = np.zeros((k - 1, k - 1), dtype=np.int64)
    R = np.zeros(k - 1, dtype=np.int64)
    lib().oracle_mccullagh(k, _p(mat, _i64p), _p(out, _f64p), _p(N, _i64p), _p(R, _i64p))
    return tuple(out), N, R


def trimmed_std(d):
    d = np.ascontiguousarray(d, dtype=np.float64)
    err = ctypes.c_int32(0)
    v = lib().oracle_trimmed_std(_p(d, _f64p), len(d), ctypes.byref(err))
    if err.value:
        raise IndexError("BoundsError in the reference")
    return float(v)


def bh(p):
    p = np.ascontiguousarray(p, dtype=np.float64)
    out = np.empty_like(p)
    lib().oracle_bh(_p(p, _f64p), len(p), _p(out, _f64p))
    return out


def iterate(code, ref0, pval_deg, padj_deg, n_iter, n_conv):
    code = np.ascontiguousarray(code, dtype=np.uint8)
    G = code.shape[0]
    ref0 = np.ascontiguousarray(ref0, dtype=np.uint8)
    res = np.zeros((G, 15), order="F")
    iters = ctypes.c_int32(0)
    trace = np.zeros((max(n_iter, 1), 2), dtype=np.int32)
    rc = lib().oracle_iterate(_p(code, _u8p), G, _p(ref0, _u8p), pval_deg, padj_deg, n_iter, n_conv,
                              _p(res, _f64p), ctypes.byref(iters), _p(trace, _i32p))
    if rc:
        raise IndexError("BoundsError in the reference")
    return res, iters.value, [tuple(int(v) for v in t) for t in trace[: iters.value]]


def identify_degs(X, gid, ngroups, pval_reo, pval_deg, padj_deg, ref0, n_iter, n_conv, seed, k=0):
    X, G, S = _colmajor(X)
    gid = np.ascontiguousarray(gid, dtype=np.int32)
    ref0 = np.ascontiguousarray(ref0, dtype=np.uint8)
    res = np.zeros((G, 15), order="F")
    iters = ctypes.c_int32(0)
    trace = np.zeros((max(n_iter, 1), 2), dtype=np.int32)
    rc = lib().oracle_identify_degs(_p(X, _f64p), G, S, G, _p(gid, _i32p), ngroups, k, pval_reo, pval_deg, padj_deg,
                                    _p(ref0, _u8p), n_iter, n_conv, seed, _p(res, _f64p), ctypes.byref(iters),
                                    _p(trace, _i32p))
    if rc:
        raise RuntimeError(f"oracle_identify_degs rc={rc}")
    return res, iters.value, [tuple(int(v) for v in t) for t in trace[: iters.value]]


def tuned_identify_degs(X, gid, ngroups, pval_reo, pval_deg, padj_deg, ref0, n_iter, n_conv, seed):
    """R2 of SURVEY.md section 8d (oracle/reo_tuned.c): the tuned CPU implementation, two groups; same results as
    identify_degs.  For the reported CPU baseline only."""
    X, G, S = _colmajor(X)
    gid = np.ascontiguousarray(gid, dtype=np.int32)
    ref0 = np.ascontiguousarray(ref0, dtype=np.uint8)
    res = np.zeros((G, 15), order="F")
    iters = ctypes.c_int32(0)
    trace = np.zeros((max(n_iter, 1), 2), dtype=np.int32)
    L = lib()
    L.tuned_identify_degs.restype = ctypes.c_int32
    L.tuned_identify_degs.argtypes = [_f64p, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, _i32p, ctypes.c_int32, ctypes.c_double,
                                      ctypes.c_double, ctypes.c_double, _u8p, ctypes.c_int32, ctypes.c_int32, ctypes.c_uint64,
                                      _f64p, ctypes.POINTER(ctypes.c_int32), _i32p]
    rc = L.tuned_identify_degs(_p(X, _f64p), G, S, G, _p(gid, _i32p), ngroups, pval_reo, pval_deg, padj_deg, _p(ref0, _u8p),
                               n_iter, n_conv, seed, _p(res, _f64p), ctypes.byref(iters), _p(trace, _i32p))
    if rc:
        raise RuntimeError(f"tuned_identify_degs rc={rc}")
    return res, iters.value, [tuple(int(v) for v in t) for t in trace[: iters.value]]


def _tuned_sigs():
    L = lib()
    u64p = ctypes.POINTER(ctypes.c_uint64)
    L.tuned_build_table.restype = ctypes.c_int32
    L.tuned_build_table.argtypes = [_f64p, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, _i32p, ctypes.c_int32, ctypes.c_double, ctypes.c_uint64, u64p]
    L.tuned_iterate.restype = ctypes.c_int32
    L.tuned_iterate.argtypes = [u64p, ctypes.c_int64, ctypes.c_double, ctypes.c_double, _u8p, ctypes.c_int32, ctypes.c_int32, _f64p,
                                ctypes.POINTER(ctypes.c_int32), _i32p]
    L.tuned_decode.restype = None
    L.tuned_decode.argtypes = [u64p, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, _u8p]
    return L, u64p


def tuned_build_table(X, gid, ngroups, pval_reo, seed):
    """R2's class table of comparison 0 as bit planes [G][cL cH tL tH][ceil(G / 64)] uint64 (reo_tuned.c; two groups)."""
    L, u64p = _tuned_sigs()
    X, G, S = _colmajor(X)
    gid = np.ascontiguousarray(gid, dtype=np.int32)
    T = np.zeros((G, 4, (G + 63) // 64), dtype=np.uint64)
    rc = L.tuned_build_table(_p(X, _f64p), G, S, G, _p(gid, _i32p), ngroups, pval_reo, seed, _p(T, u64p))
    if rc:
        raise RuntimeError(f"tuned_build_table rc={rc}")
    return T


def tuned_iterate(T, ref0, pval_deg, padj_deg, n_iter, n_conv):
    """The iteration driver on a table of tuned_build_table: (result G x 15, passes, trace)."""
    L, u64p = _tuned_sigs()
    G = T.shape[0]
    ref0 = np.ascontiguousarray(ref0, dtype=np.uint8)
    res = np.zeros((G, 15), order="F")
    iters = ctypes.c_int32(0)
    trace = np.zeros((max(n_iter, 1), 2), dtype=np.int32)
    rc = L.tuned_iterate(_p(T, u64p), G, pval_deg, padj_deg, _p(ref0, _u8p), n_iter, n_conv, _p(res, _f64p), ctypes.byref(iters), _p(trace, _i32p))
    if rc:
        raise RuntimeError(f"tuned_iterate rc={rc}")
    return res, iters.value, [tuple(int(v) for v in t) for t in trace[: iters.value]]


def tuned_decode(T, i0, i1, j0, j1):
    """Class codes 0..8 (255 on the diagonal) of a block of a table of tuned_build_table."""
    L, u64p = _tuned_sigs()
    code = np.empty((i1 - i0, j1 - j0), dtype=np.uint8)
    L.tuned_decode(_p(T, u64p), T.shape[0], i0, i1, j0, j1, _p(code, _u8p))
    return code


def num_threads():
    return int(lib().oracle_num_threads())
