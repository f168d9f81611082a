"""What version of the reference printed README.md:38-55 (eleven genes, all "up")?  Not the current one: with the p-values
recalibrated from the trimmed spread of delta1 (src/RankCompV3.jl:409-416) DE1..DE6 of the bundled data are "no change".  The
commented-out line :414 (`# padj = adjust(result[:,1], BenjaminiHochberg())`) is the trace of a predecessor that adjusted
McCullagh's OWN p-value (:255).  This script runs that predecessor's whole iteration on the bundled files with the oracle's parts
(class table, tallies, McCullagh) and prints, per pass, how many genes it calls DEGs and how many of the eleven displayed genes
are "up".  CPU only (test infrastructure: it uses oracle/).    python tools/readme_predecessor_rule.py [seed ...]
Result (round 5): about 90 % of the 19 999 genes are DEGs under that rule; the eleven displayed genes end 10-11 of 11 "up"
depending on the seed of the draws the reference leaves unseeded -- consistent with the display, which the current rule is not."""
import ctypes, importlib, logging, os, sys
import numpy as np
from scipy import stats
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
pkg = ge.load_pkg(); oracle = ge.load_oracle()
from oracle import reo_numpy as rn
R = importlib.import_module(pkg.__name__ + ".reoa")
logging.disable(logging.WARNING)
NAMES = ["DE1", "DE2", "DE3", "DE4", "DE5", "DE6", "EE19996", "EE19997", "EE19998", "EE19999", "EE20000"]
gold = os.path.join(ROOT, "tests", "golden")
for seed in [int(a, 0) for a in sys.argv[1:]] or [0x5EED0001, 1, 2]:
    prep = R.prepare(os.path.join(gold, "fn_expr.txt"), os.path.join(gold, "fn_meta.txt"), seed=seed, use_hk_genes="no")
    gid, lev = pkg.encode_groups(prep["sample_groups"])
    X = prep["data"].astype(np.float64); G = X.shape[0]
    idx = [prep["gene_names"].index(n) for n in NAMES]
    code = oracle.build_codes(X, gid, 2, 0, [oracle.threshold(5, 0.01)] * 2, seed)
    ref = prep["ref"].astype(np.uint8).copy()
    L = oracle.lib()
    for it in range(128):
        cont = oracle.tally(code, ref)
        res = np.zeros((G, 15), order="F"); inds = np.zeros(G, np.uint8)
        L.oracle_iter_stats(cont.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)), G, 1.0, 0.05,
                            res.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), inds.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8)))
        z1 = res[:, 14]
        padj = rn.bh(np.minimum(1.0, 2 * np.minimum(stats.norm.cdf(z1), stats.norm.sf(z1))))   # :255, then :414 as it was
        nondeg = ~(padj <= 0.05)
        up = (padj <= 0.05) & (z1 > 0)
        print("seed %#x pass %d: reference genes %d, DEGs %d (up %d), of the eleven displayed: %d up, smallest z1 %.2f"
              % (seed, it, int(ref.sum()), int((~nondeg).sum()), int(up.sum()), int(up[idx].sum()), z1[idx].min()), flush=True)
        if abs(int(ref.sum()) - int(nondeg.sum())) < 5:   # :419
            break
        ref = nondeg.astype(np.uint8)
