"""How many reference genes change from pass to pass at config 3 (REO_DEBUG_PASSES prints the count behind every batch)."""
import os, sys, numpy as np
os.environ["REO_DEBUG_PASSES"]="1"
sys.path.insert(0,'.')
import __graft_entry__ as ge
pkg=ge.load_pkg()
G,S,seed=20000,1000,0x5EED0003
X=pkg.synth.t0_ranks(G,S,seed)
gid,_=pkg.encode_groups(np.asarray(pkg.synth.groups(S))); ref0=pkg.synth.ref_mask(G,3000,seed)
with pkg.Context(device=0,seed=seed) as ctx:
    ctx.set_matrix(X); ctx.set_groups(gid,2); ctx.compute_thresholds(0.01); ctx.build_pairs(0)
    for n in (1,2,3,4,5,6,20):
        ctx.identify_degs(ref0,1.0,0.05,n,0)
