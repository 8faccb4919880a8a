"""Test infrastructure (CPU, uses oracle/): is the reference's nearest-occupied-cell label (emptyCells, reference jly_3ddt.cpp:1012-1132) a function of
the LOCAL tie -- the set of occupied cells at exactly the minimal squared distance, as offsets from the cell? For every cell of a grid built by the
oracle's reference-order builder the exact nearest seeds are found by brute force; cells with two or more are grouped by their set of tied offsets.
If the same set of tied offsets gets different labels in different cells, no rule that looks only at the tied seeds can reproduce the reference map
(the winner depends on which propagated vector reached the cell first, i.e. on the scan history of its neighbourhood).
    python tools/dt_tie_probe.py"""
import sys, numpy as np, collections
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from oracle.refbind import Oracle, cfg_vector
from helpers import load_clouds, synth_pair
def probe(model, S, tag):
    v,_ = cfg_vector(distTransSize=S)
    o = Oracle(); o.set_config(v); o.set_clouds(model, model[:50]); o.build_dt()
    dist, near3, cc, _ = o.dt_dump()
    g = o.geom  # xMin,xMax,yMin,yMax,zMin,zMax,scale,S
    scale = g[6]
    cells = np.floor((model.astype(np.float64) - np.array([g[0],g[2],g[4]]))*scale + 0.5).astype(int)
    cells = np.unique(cells[(cells>=0).all(1) & (cells<S).all(1)], axis=0)   # occupied cells (x,y,z)
    zz,yy,xx = np.meshgrid(np.arange(S),np.arange(S),np.arange(S), indexing='ij')
    P = np.stack([xx.ravel(),yy.ravel(),zz.ravel()],1)
    choice = collections.defaultdict(collections.Counter); where = collections.defaultdict(list)
    ntied = 0; nonmin = 0
    near3 = near3.reshape(-1,3)
    for i in range(0, len(P), 4096):
        p = P[i:i+4096]
        d2 = ((p[:,None,:]-cells[None,:,:])**2).sum(-1)
        mn = d2.min(1)
        for r in range(len(p)):
            if mn[r] == 0: continue
            tied = cells[d2[r]==mn[r]] - p[r]
            lab = near3[i+r] - p[r]
            if (lab**2).sum() != mn[r]: nonmin += 1; continue
            if len(tied) < 2: continue
            ntied += 1
            key = tuple(sorted(map(tuple, tied.tolist())))
            choice[key][tuple(lab.tolist())] += 1; where[(key, tuple(lab.tolist()))].append(tuple(p[r].tolist()))
    amb = {k:c for k,c in choice.items() if len(c)>1}
    print(tag, "S",S,"occupied",len(cells),"tied cells",ntied,"label not at min distance",nonmin,"distinct tie sets",len(choice),"tie sets with MORE THAN ONE label chosen",len(amb))
    for k, c in list(amb.items())[:5]:
        print("   tied seeds at offsets", k, "-> the reference's nearest-occupied cell:", ", ".join("offset %s in cell(s) %s" % (l, where[(k, l)][:2]) for l in c))
    return amb
P = load_clouds("pair1")
probe(P["t"], 20, "pair1 target")
Q = synth_pair(300, seed=5)
probe(Q["t"], 20, "synthetic 300")
probe(Q["t"], 30, "synthetic 300")
