"""Regenerates tests/golden/icp_ties.npz from the COMPILED REFERENCE (oracle/_ref/libgoicp_ref.so): GoICP::ICP in/out on clouds
built to produce exact float distance ties in the kd-tree search (duplicated model points, queries on lattice midpoints), which
pins the tie rule of the 1-NN step (reference nanoflann.hpp:1137-1184: the first point the traversal meets wins).
Run in the build container only (needs /root/reference):  python tests/golden/make_icp_ties.py"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle.refbind import Ref, cfg_vector  # noqa: E402


def tie_clouds(seed, n_dup=40):
    rng = np.random.default_rng(seed)
    g = np.arange(6) * 0.25 - 0.625                       # exactly representable lattice
    L = np.stack(np.meshgrid(g, g, g, indexing="ij"), -1).reshape(-1, 3)
    m = np.concatenate([L, L[rng.choice(len(L), n_dup, replace=False)]])        # duplicated model points
    m = m[rng.permutation(len(m))].astype(np.float32)
    mids = []
    for _ in range(120):                                   # queries equidistant from 2, 4 or 8 lattice points
        c = L[rng.integers(len(L))].copy()
        ax = rng.choice(3, rng.integers(1, 4), replace=False)
        c[ax] += 0.125
        mids.append(c)
    d = np.concatenate([np.array(mids), L[rng.choice(len(L), 40)], rng.uniform(-0.7, 0.7, (60, 3))]).astype(np.float32)
    return m, d[rng.permutation(len(d))]


def main():
    out = {}
    for name, seed, over in (("plain", 1, {}), ("trim", 2, dict(trimFraction=0.1)), ("geo", 3, dict(regularization=0.0, ponderation=0))):
        m, d = tie_clouds(seed)
        v, _ = cfg_vector(**over)
        r = Ref(); r.set_config(v); r.set_clouds(m, d); r.build_dt(); r.initialize()
        rng = np.random.default_rng(50 + seed)
        rows = []
        for k in range(5):
            R = np.eye(3); t = np.array([0.125, 0.0, -0.25]) * (k % 2) if k < 2 else rng.uniform(-0.1, 0.1, 3)
            e, Ro, to, corr, dis = r.icp(R.copy(), np.asarray(t, np.float64).copy())
            rows.append((R, np.asarray(t, np.float64), e, Ro, to, corr, dis))
        out.update({name + "_cfg": v, name + "_model": m, name + "_data": d,
                    name + "_Rin": np.stack([x[0] for x in rows]), name + "_tin": np.stack([x[1] for x in rows]),
                    name + "_err": np.array([x[2] for x in rows], np.float32), name + "_Rout": np.stack([x[3] for x in rows]),
                    name + "_tout": np.stack([x[4] for x in rows]), name + "_corr": np.stack([x[5] for x in rows]),
                    name + "_dis": np.stack([x[6] for x in rows])})
    np.savez_compressed(os.path.join(HERE, "icp_ties.npz"), **out)
    print("icp_ties.npz written")


if __name__ == "__main__":
    main()
