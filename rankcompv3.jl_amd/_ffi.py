"""ctypes binding of libreo_hip.so (include/reo_hip.h).

There is no CPU fallback: if the shared library is missing or no MI355X is
visible the calls raise, they never route anywhere else.
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("REO_LIB_PATH") or os.path.join(_HERE, "libreo_hip.so")  # (REO_LIB_PATH: an instrumented build, tools/asan_host.sh)
CSRC = os.path.join(_HERE, "csrc")

REO_OK, REO_EINVAL, REO_EHIP, REO_ECOMM, REO_ENOMEM = 0, -1, -2, -3, -4
NTIMINGS = 12

# every symbol include/reo_hip.h declares
SYMBOLS = [
    "reo_version", "reo_last_error", "reo_create", "reo_destroy", "reo_trim_memory", "reo_set_shard", "reo_set_allreduce", "reo_set_allgather",
    "reo_create_multi", "reo_comm_unique_id", "reo_comm_init_rank",
    "reo_set_matrix_f64", "reo_set_matrix_i64", "reo_set_matrix_dev_f64", "reo_set_matrix_dev_i64",
    "reo_set_groups", "reo_compute_thresholds", "reo_set_thresholds", "reo_get_thresholds", "reo_threshold",
    "reo_build_pairs", "reo_pair_counts", "reo_get_codes", "reo_tally", "reo_identify_degs", "reo_mccullagh",
    "reo_set_profiling", "reo_reset_timings", "reo_get_timings", "reo_get_info",
    "reo_pseudobulk_dense_f64", "reo_pseudobulk_dense_i64", "reo_pseudobulk_csc_f64", "reo_pseudobulk_csc_i64",
]

ALLREDUCE_FN = ctypes.CFUNCTYPE(ctypes.c_int32, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p)
ALLGATHER_FN = ctypes.CFUNCTYPE(ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p)


class LibraryMissing(RuntimeError):
    """libreo_hip.so has not been built (run __graft_entry__.build())."""


class ReoError(RuntimeError):
    def __init__(self, status: int, message: str):
        super().__init__(f"libreo_hip status {status}: {message}")
        self.status = status
        self.message = message


class DimensionMismatch(ReoError, ValueError):
    """REO_EINVAL: the reference's DimensionMismatch / ArgumentError / BoundsError paths."""


def build_library(force: bool = False) -> str:
    """Compile libreo_hip.so for gfx950 with hipcc (cross-compiles without a GPU)."""
    args = ["make", "-C", CSRC, "-j4"]
    if force:
        args.append("-B")
    subprocess.check_call(args, stdout=subprocess.DEVNULL)
    return LIB_PATH


_LIB = None
SIGNATURES = {}   # name -> (restype, argtypes) of every bound entry point, filled by lib()


def lib() -> ctypes.CDLL:
    global _LIB
    if _LIB is not None:
        return _LIB
    if not os.path.exists(LIB_PATH):
        raise LibraryMissing(f"{LIB_PATH} not found: build it with __graft_entry__.build(); there is no CPU fallback")
    try:
        # torch ships its own copy of the HIP runtime; two copies in one process do not share the GPU
        # (the second one reports "no HIP GPUs").  Importing torch first makes libreo_hip.so bind to
        # the copy torch already loaded, so both see the same devices and streams.
        import torch  # noqa: F401
    except ImportError:
        pass
    L = ctypes.CDLL(LIB_PATH)
    vp, i32, i64, u64, f64 = ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64, ctypes.c_uint64, ctypes.c_double
    sig = {
        "reo_version": (i32, []),
        "reo_last_error": (ctypes.c_char_p, []),
        "reo_create": (i32, [ctypes.POINTER(vp), i32, u64]),
        "reo_destroy": (None, [vp]),
        "reo_trim_memory": (i32, []),
        "reo_set_shard": (i32, [vp, i32, i32]),
        "reo_create_multi": (i32, [ctypes.POINTER(vp), i32, u64]),
        "reo_comm_unique_id": (i32, [vp]),
        "reo_comm_init_rank": (i32, [vp, vp, i32, i32]),
        "reo_set_allreduce": (i32, [vp, ALLREDUCE_FN, vp]),
        "reo_set_allgather": (i32, [vp, ALLGATHER_FN, vp]),
        "reo_set_matrix_f64": (i32, [vp, vp, i64, i64, i64]),
        "reo_set_matrix_i64": (i32, [vp, vp, i64, i64, i64]),
        "reo_set_matrix_dev_f64": (i32, [vp, vp, i64, i64, i64]),
        "reo_set_matrix_dev_i64": (i32, [vp, vp, i64, i64, i64]),
        "reo_set_groups": (i32, [vp, vp, i64, i32]),
        "reo_compute_thresholds": (i32, [vp, f64]),
        "reo_set_thresholds": (i32, [vp, vp]),
        "reo_get_thresholds": (i32, [vp, vp]),
        "reo_threshold": (i32, [i32, f64]),
        "reo_build_pairs": (i32, [vp, i32]),
        "reo_pair_counts": (i32, [vp, i64, i64, i64, i64, vp, vp]),
        "reo_get_codes": (i32, [vp, i64, i64, i64, i64, vp]),
        "reo_tally": (i32, [vp, vp, vp]),
        "reo_identify_degs": (i32, [vp, vp, f64, f64, i32, i32, vp, vp, vp]),
        "reo_mccullagh": (i32, [vp, vp, i64, vp]),
        "reo_set_profiling": (i32, [vp, i32]),
        "reo_reset_timings": (i32, [vp]),
        "reo_get_timings": (i32, [vp, vp, i32]),
        "reo_get_info": (i32, [vp, vp, i32]),
        "reo_pseudobulk_dense_f64": (i32, [vp, vp, i64, i64, i64, vp, i64, vp, i32, vp]),
        "reo_pseudobulk_dense_i64": (i32, [vp, vp, i64, i64, i64, vp, i64, vp, i32, vp]),
        "reo_pseudobulk_csc_f64": (i32, [vp, i64, i64, vp, vp, vp, vp, i64, vp, i32, vp]),
        "reo_pseudobulk_csc_i64": (i32, [vp, i64, i64, vp, vp, vp, vp, i64, vp, i32, vp]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)
        fn.restype = res
        fn.argtypes = args
    global SIGNATURES
    SIGNATURES = sig
    _LIB = L
    return L


def check(status: int) -> None:
    if status == REO_OK:
        return
    msg = lib().reo_last_error().decode("utf-8", "replace")
    if status == REO_EINVAL:
        raise DimensionMismatch(status, msg)
    raise ReoError(status, msg)


def trim_memory() -> None:
    """reo_trim_memory: give the cached device and pinned blocks of destroyed contexts back to the driver (other allocators of the
    process -- PyTorch's -- cannot see what the library's block cache holds; REO_DEVICE_CACHE_MB bounds it)."""
    check(lib().reo_trim_memory())


def threshold(sample_size: int, pval_reo: float = 0.01) -> int:
    """get_major_reo_lower_count (src/RankCompV3.jl:81-92) as the library computes it (host arithmetic)."""
    return int(lib().reo_threshold(int(sample_size), float(pval_reo)))


def _ptr(a: np.ndarray) -> int:
    return a.ctypes.data


UNIQUE_ID_BYTES = 128


def comm_unique_id() -> bytes:
    """reo_comm_unique_id: the 128 bytes rank 0 hands to every rank of a one-process-per-GPU run."""
    buf = ctypes.create_string_buffer(UNIQUE_ID_BYTES)
    check(lib().reo_comm_unique_id(buf))
    return buf.raw


class Context:
    """One reo_ctx: one GPU (or, with n_gpus, all GPUs of this process behind one handle), one expression matrix."""

    def __init__(self, device: int = -1, seed: int = 0, n_gpus: int | None = None):
        self._h = ctypes.c_void_p()
        self._L = lib()
        if n_gpus is None:
            check(self._L.reo_create(ctypes.byref(self._h), int(device), int(seed) & 0xFFFFFFFFFFFFFFFF))
        else:  # reo_create_multi: 0 = all visible devices
            check(self._L.reo_create_multi(ctypes.byref(self._h), int(n_gpus), int(seed) & 0xFFFFFFFFFFFFFFFF))
        self._keep = []  # keeps callbacks / device tensors alive
        self.G = self.S = 0
        self.ngroups = 0

    def close(self) -> None:
        if self._h:
            self._L.reo_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- problem definition -------------------------------------------------
    def set_matrix(self, X: np.ndarray) -> None:
        """X is genes x samples (host); Float64 or any integer dtype (-> Int64), like Matrix(df_expr)."""
        X = np.asarray(X)
        if X.ndim != 2:
            raise DimensionMismatch(REO_EINVAL, "expression matrix must be 2-D (genes x samples)")
        integer = np.issubdtype(X.dtype, np.integer) or X.dtype == np.bool_
        fn = self._L.reo_set_matrix_i64 if integer else self._L.reo_set_matrix_f64
        want = np.int64 if integer else np.float64
        G, S = X.shape
        if X.dtype == want and G > 0 and S > 1 and X.strides[0] == 8 and X.strides[1] % 8 == 0 and X.strides[1] >= 8 * G:
            Xf, ld = X, X.strides[1] // 8  # a column-major view (rows of a taller matrix): passed as is, like a Julia view
        else:
            Xf = np.asfortranarray(X, dtype=want)
            ld = max(G, 1)
        check(fn(self._h, _ptr(Xf), G, S, ld))
        self.G, self.S = G, S

    def set_matrix_device(self, dev_ptr: int, G: int, S: int, ld: int, dtype: str, keepalive=None) -> None:
        """Column-major matrix already resident in HBM (dtype 'f64' or 'i64')."""
        fn = self._L.reo_set_matrix_dev_f64 if dtype == "f64" else self._L.reo_set_matrix_dev_i64
        check(fn(self._h, ctypes.c_void_p(dev_ptr), G, S, ld))
        self._keep.append(keepalive)
        self.G, self.S = G, S

    def set_groups(self, group_id, ngroups: int) -> None:
        gid = np.ascontiguousarray(group_id, dtype=np.int32)
        check(self._L.reo_set_groups(self._h, _ptr(gid), gid.size, int(ngroups)))
        self.ngroups = int(ngroups)

    def compute_thresholds(self, pval_reo: float) -> np.ndarray:
        check(self._L.reo_compute_thresholds(self._h, float(pval_reo)))
        return self.get_thresholds()

    def set_thresholds(self, m) -> None:
        m = np.ascontiguousarray(m, dtype=np.int32)
        if m.size != 2 * self.ngroups:
            raise DimensionMismatch(REO_EINVAL, "thresholds must be 2 x ngroups")
        check(self._L.reo_set_thresholds(self._h, _ptr(m)))

    def get_thresholds(self) -> np.ndarray:
        m = np.zeros(2 * self.ngroups, dtype=np.int32)
        check(self._L.reo_get_thresholds(self._h, _ptr(m)))
        return m.reshape(self.ngroups, 2).T.copy()  # 2 x ngroups like :362

    def set_shard(self, rank: int, world: int) -> None:
        check(self._L.reo_set_shard(self._h, int(rank), int(world)))

    def comm_init_rank(self, unique_id: bytes, rank: int, world: int) -> None:
        """In-library RCCL: join the communicator of `unique_id` as shard `rank` of `world` (sets the shard too)."""
        if len(unique_id) != UNIQUE_ID_BYTES:
            raise DimensionMismatch(REO_EINVAL, "unique id must be 128 bytes")
        check(self._L.reo_comm_init_rank(self._h, ctypes.c_char_p(unique_id), int(rank), int(world)))

    def set_allreduce(self, fn) -> None:
        """fn(dev_ptr: int, count: int, stream: int) -> None: sum int32[count] (the class table) in place across
        shards, ordered on the HIP stream `stream` (see include/reo_hip.h).  None removes the hook."""
        if fn is None:
            check(self._L.reo_set_allreduce(self._h, ALLREDUCE_FN(), None))
            return

        def _cb(ptr, count, stream, _user):
            try:
                fn(int(ptr), int(count), int(stream or 0))
                return 0
            except Exception:  # never let an exception cross the C boundary
                import traceback
                traceback.print_exc()
                return 1
        cb = ALLREDUCE_FN(_cb)
        self._keep.append(cb)
        check(self._L.reo_set_allreduce(self._h, cb, None))

    def set_allgather(self, fn) -> None:
        """fn(send_ptr: int, recv_ptr: int, bytes_per_rank: int, stream: int) -> None: gather `bytes_per_rank` bytes of
        every shard's `send` into `recv + shard * bytes_per_rank` on every shard, ordered on the HIP stream `stream`
        (the cheaper form of the table exchange, see include/reo_hip.h).  None removes the hook."""
        if fn is None:
            check(self._L.reo_set_allgather(self._h, ALLGATHER_FN(), None))
            return

        def _cb(send, recv, nbytes, stream, _user):
            try:
                fn(int(send), int(recv), int(nbytes), int(stream or 0))
                return 0
            except Exception:  # never let an exception cross the C boundary
                import traceback
                traceback.print_exc()
                return 1
        cb = ALLGATHER_FN(_cb)
        self._keep.append(cb)
        check(self._L.reo_set_allgather(self._h, cb, None))

    # -- hot path -------------------------------------------------------------
    def build_pairs(self, k: int = 0) -> None:
        check(self._L.reo_build_pairs(self._h, int(k)))

    def pair_counts(self, i0: int, i1: int, j0: int, j1: int):
        shape = (i1 - i0, j1 - j0, self.ngroups)
        gt = np.zeros(shape, dtype=np.uint16)
        eq = np.zeros(shape, dtype=np.uint16)
        check(self._L.reo_pair_counts(self._h, i0, i1, j0, j1, _ptr(gt), _ptr(eq)))
        return gt, eq

    def get_codes(self, i0: int, i1: int, j0: int, j1: int) -> np.ndarray:
        code = np.zeros((i1 - i0, j1 - j0), dtype=np.uint8)
        check(self._L.reo_get_codes(self._h, i0, i1, j0, j1, _ptr(code)))
        return code

    def tally(self, ref_mask) -> np.ndarray:
        ref = np.ascontiguousarray(np.asarray(ref_mask) != 0, dtype=np.uint8)
        if ref.size != self.G:
            raise DimensionMismatch(REO_EINVAL, "reference mask length != number of genes")
        cont = np.zeros((self.G, 9), dtype=np.int32)
        check(self._L.reo_tally(self._h, _ptr(ref), _ptr(cont)))
        return cont

    def identify_degs(self, ref0, pval_deg: float, padj_deg: float, n_iter: int, n_conv: int):
        ref = np.ascontiguousarray(np.asarray(ref0) != 0, dtype=np.uint8)
        if ref.size != self.G:
            raise DimensionMismatch(REO_EINVAL, "reference mask length != number of genes")
        result = np.zeros((self.G, 15), dtype=np.float64, order="F")
        iters = ctypes.c_int32(0)
        trace = np.zeros((max(int(n_iter), 1), 2), dtype=np.int32)
        check(self._L.reo_identify_degs(self._h, _ptr(ref), float(pval_deg), float(padj_deg), int(n_iter), int(n_conv),
                                        _ptr(result), ctypes.byref(iters), _ptr(trace)))
        return result, iters.value, [tuple(int(v) for v in t) for t in trace[: iters.value]]

    def mccullagh(self, cont) -> np.ndarray:
        cont = np.ascontiguousarray(cont, dtype=np.int32).reshape(-1, 9)
        out = np.zeros((cont.shape[0], 5), dtype=np.float64)
        check(self._L.reo_mccullagh(self._h, _ptr(cont), cont.shape[0], _ptr(out)))
        return out

    # -- pseudo-bulk front end ------------------------------------------------------
    def pseudobulk(self, cells, order, chunk_ptr) -> np.ndarray:
        """Row-wise sums of groups of cells (src/RankCompV3.jl:56-67).  `cells` is a genes x cells
        ndarray (Int64 / Float64) or a scipy.sparse matrix (converted to CSC); `order` lists the cells
        of all output profiles back to back and chunk_ptr delimits them.  Returns genes x profiles."""
        order = np.ascontiguousarray(order, dtype=np.int32)
        chunk_ptr = np.ascontiguousarray(chunk_ptr, dtype=np.int32)
        n_out = chunk_ptr.size - 1
        if hasattr(cells, "tocsc"):
            m = cells.tocsc()
            m.sort_indices()
            G, C = m.shape
            isint = np.issubdtype(m.dtype, np.integer)
            val = np.ascontiguousarray(m.data, dtype=np.int64 if isint else np.float64)
            colptr = np.ascontiguousarray(m.indptr, dtype=np.int64)
            rowidx = np.ascontiguousarray(m.indices, dtype=np.int32)
            out = np.zeros((G, n_out), dtype=val.dtype, order="F")
            fn = self._L.reo_pseudobulk_csc_i64 if isint else self._L.reo_pseudobulk_csc_f64
            check(fn(self._h, G, C, _ptr(colptr), _ptr(rowidx), _ptr(val), _ptr(order), order.size, _ptr(chunk_ptr), n_out, _ptr(out)))
            return out
        X = np.asarray(cells)
        isint = np.issubdtype(X.dtype, np.integer)
        Xf = np.asfortranarray(X, dtype=np.int64 if isint else np.float64)
        G, C = Xf.shape
        out = np.zeros((G, n_out), dtype=Xf.dtype, order="F")
        fn = self._L.reo_pseudobulk_dense_i64 if isint else self._L.reo_pseudobulk_dense_f64
        check(fn(self._h, _ptr(Xf), G, C, G, _ptr(order), order.size, _ptr(chunk_ptr), n_out, _ptr(out)))
        return out

    # -- instrumentation -------------------------------------------------------
    def set_profiling(self, on: bool) -> None:
        check(self._L.reo_set_profiling(self._h, 1 if on else 0))

    def reset_timings(self) -> None:
        check(self._L.reo_reset_timings(self._h))

    def timings(self) -> dict:
        ms = np.zeros(NTIMINGS, dtype=np.float64)
        check(self._L.reo_get_timings(self._h, _ptr(ms), NTIMINGS))
        return {"transform_ms": ms[0], "k1_ms": ms[1], "k2_ms": ms[2], "iter_ms": ms[3], "k3_ms": max(ms[3] - ms[2], 0.0), "k2_launches": int(ms[4]),
                "k1_launches": int(ms[5]), "exchange_ms": ms[6], "pseudobulk_ms": ms[7], "k2_full_ms": ms[8],
                "k2_full_launches": int(ms[9]), "k2_delta_ms": ms[10], "set_matrix_host_wall_ms": ms[11]}

    def info(self) -> dict:
        v = np.zeros(21, dtype=np.int64)
        check(self._L.reo_get_info(self._h, _ptr(v), 21))
        return {"G": int(v[0]), "S": int(v[1]), "Gp": int(v[2]), "table_bytes": int(v[3]), "has_ties": int(v[4]),
                "tiles_owned": int(v[5]), "tiles_total": int(v[6]), "tile_i": int(v[7]), "chunk_j": int(v[8]),
                "chunks_per_panel": int(v[9]), "unit_h": int(v[10]), "sample_slots": int(v[11]),
                "shared_group_counts": int(v[12]), "group_count_bytes": int(v[13]), "transform_in_lds": int(v[14]), "xcc_local_histograms": int(v[15]),
                "cycle_period": int(v[16]), "cycle_found_at_pass": int(v[17]), "cycle_passes_skipped": int(v[18]), "upload_link_bytes": int(v[19]),
                "eager_range_launches": int(v[20])}
