"""Host-side logic and the C-ABI surface.  No GPU, no compute calls."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol(pkg):
    hdr = open(os.path.join(ROOT, "include", "reo_hip.h")).read()
    declared = set(re.findall(r"\b(reo_[a-z0-9_]+)\s*\(", hdr)) - {"reo_allreduce_fn"}
    assert declared == set(pkg._ffi.SYMBOLS), declared ^ set(pkg._ffi.SYMBOLS)
    L = ctypes.CDLL(pkg._ffi.LIB_PATH)
    for s in declared:
        assert hasattr(L, s), s
    assert pkg._ffi.lib().reo_version() >= 100


def test_library_is_gfx950_code(pkg):
    blob = open(pkg._ffi.LIB_PATH, "rb").read()
    assert b"gfx950" in blob and b"k1_pairs" in blob and b"k2_tally" in blob


def test_threshold_function_matches_oracle(pkg, oracle, golden):
    for p, row in golden("thresholds.json").items():
        for n, m in row.items():
            assert pkg.threshold(int(n), float(p)) == m
    for n in range(2, 200, 7):
        assert pkg.threshold(n, 0.01) == oracle.threshold(n, 0.01)


def test_no_gpu_means_loud_failure(pkg):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(pkg.ReoError) as e:
        pkg.Context()
    assert e.value.status == pkg._ffi.REO_EHIP and "no CPU fallback" in e.value.message
    with pytest.raises(pkg.ReoError):
        pkg.identify_degs(np.zeros((12, 4)), ["a", "a", "b", "b"], list("abcdefghijkl"), 0.01, 1.0, 0.05,
                          np.ones(12, bool), 2, 1)


def test_host_argument_checks_mirror_the_reference(pkg):
    X = np.zeros((12, 4))
    names = list("abcdefghijkl")
    with pytest.raises(pkg.DimensionMismatch):  # :355
        pkg.identify_degs(X, ["a", "b", "b"], names, 0.01, 1.0, 0.05, np.ones(12, bool), 2, 1)
    with pytest.raises(pkg.DimensionMismatch):  # :356
        pkg.identify_degs(X, ["a"] * 4, names, 0.01, 1.0, 0.05, np.ones(12, bool), 2, 1)


def test_encode_groups_first_appearance_order(pkg):
    ids, lev = pkg.encode_groups(["t", "c", "t", "c", "x"])
    assert lev == ["t", "c", "x"] and ids.tolist() == [0, 1, 0, 1, 2]


def test_labels(pkg):
    res = np.zeros((4, 15))
    res[:, 0] = [0.001, 0.001, 0.5, 0.001]
    res[:, 1] = [0.01, 0.01, 0.9, 0.2]
    res[:, 14] = [2.0, -2.0, 3.0, 3.0]
    assert pkg.label_genes(res, 1.0, 0.05).tolist() == ["up", "down", "no change", "no change"]


def test_synthetic_generators(pkg):
    X = pkg.synth.t0_ranks(300, 20, 5)
    assert X.dtype == np.int64 and (np.sort(X, axis=0) == np.arange(300)[:, None]).all()
    assert np.array_equal(X, pkg.synth.t0_ranks(300, 20, 5))
    Y = pkg.synth.t1_counts(300, 20, 5)
    assert 0.04 < (Y == 0).mean() < 0.2 and Y.min() == 0
    m = pkg.synth.ref_mask(300, 40, 5)
    assert m.sum() == 40
    # scalar restatement of the counter RNG agrees with the vectorised one
    from oracle import reo_numpy as rn
    z = int(pkg.synth.u64(5, np.array([7]), np.array([3]))[0])
    assert z == rn.mix64(5 ^ rn.mix64((7 << 32) | 3))


def test_host_thread_pool_of_the_narrowed_upload_under_thread_sanitizer():
    """HostPool (csrc/transform.hip): the pool of host threads that narrows Int64 / Float64 chunks for the upload and checks CSC row
    indices.  The class is cut out of the source verbatim and stressed on the CPU under ThreadSanitizer (tools/hostpool_tsan.py): four
    caller threads, 80 000 short jobs with task counts around the worker count -- every task exactly once, no data race reported."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "hostpool_tsan.py")], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "hostpool_tsan: clean" in out.stdout, out.stdout + out.stderr
