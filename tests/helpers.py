"""Shared test helpers (test infrastructure)."""
import json
import os

import numpy as np

import goicp_amd  # noqa: F401
from oracle.refbind import Oracle, cfg_vector, CFG_KEYS

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def load_clouds(name):
    return dict(np.load(os.path.join(GOLDEN, name + "_clouds.npz")))


def load_hotpath(name):
    return dict(np.load(os.path.join(GOLDEN, "hotpath_%s.npz" % name)))


def load_register():
    return json.load(open(os.path.join(GOLDEN, "register.json")))


def rot_cloud(R, d):
    """pDataTemp = R*p in float32, left-to-right, no FMA (jly_goicp.cpp:750-756)."""
    R = np.asarray(R, np.float32); d = np.asarray(d, np.float32)
    return np.stack([(R[k, 0] * d[:, 0] + R[k, 1] * d[:, 1]) + R[k, 2] * d[:, 2] for k in range(3)], 1).astype(np.float32)


def oracle_for(P, cfgvec, H=None, nd=None):
    """Oracle context on clouds P with config vector; adopts the fixture's DT tables when H is given."""
    o = Oracle()
    o.set_config(cfgvec)
    o.set_clouds(P["t"], P["s"], P["tc"], P["sc"], P["tf"], P["sf"])
    if nd:
        o.set_nd(nd)
    if H is not None:
        o.load_dt(int(H["geom"][7]), H["geom"], H["dist"], H["nearest3"], H["cellc"], H["cell_ptr"], H["cell_idx"])
    else:
        o.build_dt()
    return o


from goicp_amd.synth import synth_pair  # noqa: E402,F401  (the seeded generator lives in the package: bench.py uses it too)
