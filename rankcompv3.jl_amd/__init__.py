"""MI355X-native REO pair-comparison engine behind RankCompV3's identify_degs().

The directory name carries a dot, so import it through `__graft_entry__.load_pkg()`
(or add an alias with importlib); see INTEGRATION.md.
"""
from . import _ffi, dist, synth  # noqa: F401
from ._ffi import Context, DimensionMismatch, LibraryMissing, ReoError, build_library, threshold  # noqa: F401
from .hotpath import HEADER, DegRun, encode_groups, identify_degs, label_genes, run_identify_degs  # noqa: F401
from .reoa import ArgumentError, reoa  # noqa: F401

__all__ = ["Context", "DimensionMismatch", "LibraryMissing", "ReoError", "build_library", "threshold", "HEADER",
           "DegRun", "encode_groups", "identify_degs", "label_genes", "run_identify_degs", "synth", "dist", "reoa", "ArgumentError"]
