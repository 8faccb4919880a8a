"""GPU (-m gpu): the opt-in exact distance transform built on the device (csrc/dt_gpu.hip, GoICP dt_mode=1).

It is NOT bit-identical to the reference's DT3D::Build (a sequential vector propagation that approximates the Euclidean
transform, reference jly_3ddt.cpp:897-1137), so it is checked (a) against an independent exact transform (scipy's
distance_transform_edt) on 20^3..60^3 grids, (b) by counting the cells in which it differs from the reference-order builder,
and (c) end to end: the known answers of the compiled reference must still be met within the north-star tolerance (1e-5)."""
import ctypes as C
import json
import os

import numpy as np
import pytest

import goicp_amd
from oracle.refbind import cfg_vector
from helpers import GOLDEN, load_clouds, load_register

pytestmark = pytest.mark.gpu


def _vp(a):
    return a.ctypes.data_as(C.c_void_p)


def build_exact(S, occ_cells, scale):
    L = goicp_amd.load_library()
    L.goicp_dtgpu_dist.restype = C.c_void_p; L.goicp_dtgpu_nearest.restype = C.c_void_p
    nt = 3 * (S - 1) * (S - 1) + 1
    table = (np.sqrt(np.arange(nt, dtype=np.float64)).astype(np.float32).astype(np.float64) / scale).astype(np.float32)
    occ = np.ascontiguousarray(occ_cells, np.int32)
    h = C.c_void_p()
    assert L.goicp_dtgpu_build(0, S, _vp(occ), len(occ), _vp(table), nt, C.byref(h)) == 0
    dist = np.zeros(S ** 3, np.float32); near = np.zeros(S ** 3, np.int32)
    assert L.goicp_dtgpu_download(h, _vp(dist), _vp(near)) == 0
    L.goicp_dtgpu_destroy(h)
    return dist.reshape(S, S, S), near.reshape(S, S, S), table


@pytest.mark.parametrize("S,nseed,seed", [(20, 1, 1), (20, 30, 2), (33, 200, 3), (60, 500, 4), (47, 3000, 5)])
def test_exact_transform_against_scipy(S, nseed, seed):
    from scipy import ndimage
    rng = np.random.default_rng(seed)
    occ = np.unique(rng.integers(0, S ** 3, nseed)).astype(np.int32)
    if seed == 4:                                       # a planar sheet of seeds: many exact ties
        zz, yy = np.meshgrid(np.arange(10, 50, 3), np.arange(5, 55, 2), indexing="ij")
        occ = np.unique(((zz * S + yy) * S + 30).ravel()).astype(np.int32)
    scale = 37.25
    dist, near, table = build_exact(S, occ, scale)
    mask = np.ones(S ** 3, bool); mask[occ] = False
    ed = ndimage.distance_transform_edt(mask.reshape(S, S, S))             # exact Euclidean distance in cells, [z][y][x]
    d2 = np.rint(ed ** 2).astype(np.int64)
    assert np.array_equal(dist, table[d2])                                  # the stored value of every cell
    # the nearest-occupied map points at an occupied cell at exactly that distance
    z, y, x = np.meshgrid(np.arange(S), np.arange(S), np.arange(S), indexing="ij")
    nz, ny, nx = near // (S * S), (near // S) % S, near % S
    assert np.isin(near.ravel(), occ).all()
    assert np.array_equal((z - nz) ** 2 + (y - ny) ** 2 + (x - nx) ** 2, d2)


@pytest.mark.parametrize("pair,S", [("pair1", 20), ("pair1", 100), ("pair2", 50)])
def test_cells_that_differ_from_the_reference_propagation(pair, S):
    """Same geometry and seeds as the reference-order host builder; the propagated grid over-estimates a few cells and carries
    the phantom seed of the reference's uninitialised first candidate in the low corner -- the exact grid does not."""
    P = load_clouds(pair)
    v = goicp_amd.cfg_vector(distTransSize=S)
    a = goicp_amd.GoICP(v); a.set_clouds(P["t"], P["s"], P["tc"], P["sc"], P["tf"], P["sf"]); a.build_dt()
    b = goicp_amd.GoICP(v, dt_mode=1); b.set_clouds(P["t"], P["s"], P["tc"], P["sc"], P["tf"], P["sf"]); b.build_dt()
    assert np.array_equal(a.geom, b.geom)
    da, na, ca = a.dt_dump(); db, nb, cb = b.dt_dump()
    assert np.array_equal(ca, cb)                                           # same occupied cells, same cell colours
    assert np.array_equal(da == 0, db == 0)
    diff = da != db
    lower = db < da; higher = db > da
    print("%s S=%d: %d of %d cells differ (exact lower in %d, higher in %d; max |delta| %.4f)" % (pair, S, diff.sum(), diff.size, lower.sum(), higher.sum(), np.abs(da - db).max()))
    # exact <= propagated everywhere except under the reference's phantom seed at cell (0,0,0), where the propagated value is too small
    zz, yy, xx = np.unravel_index(np.nonzero(higher)[0], (S, S, S))
    if higher.any():
        assert max(zz.max(), yy.max(), xx.max()) < 0.7 * S
    assert diff.mean() < 0.05
    # nearest-occupied map: where the two builders name different cells, the exact one is at least as close (ties or propagation slack)
    dn = np.nonzero(na != nb)[0]
    zz, yy, xx = np.unravel_index(dn, (S, S, S))
    def d2(n):
        return (zz - n // (S * S)) ** 2 + (yy - (n // S) % S) ** 2 + (xx - n % S) ** 2
    assert (d2(nb[dn]) <= d2(na[dn])).all()
    print("   nearest-occupied map: %d of %d cells name a different cell (%d of them at exactly the same distance)" % (len(dn), na.size, int((d2(nb[dn]) == d2(na[dn])).sum())))
    a.close(); b.close()


def _demo(name):
    from oracle.refbind import read_txt_cloud
    D = os.path.join(GOLDEN, "data", "demo")
    if name.startswith("rand"):
        return read_txt_cloud(os.path.join(D, "model_rand.txt")), read_txt_cloud(os.path.join(D, "data_rand.txt"))
    B = np.load(os.path.join(D, "bunny.npz"))
    return B["model"], B["data"][:1000]


@pytest.mark.parametrize("name", ["pair1_S100_geo", "rand_untrimmed", "rand_trim", "bunny_S300"])
def test_geometric_known_answers_hold_on_the_exact_grid(name):
    """End to end in dt_mode=1 on the geometric configurations (distance grid only): the compiled reference's optimal R, t within
    1e-5 and its error within 1e-5 relative -- bunny_S300 is the reference's own shipped demo/output.txt. Configurations with
    penalty terms are NOT comparable in this mode: they read the nearest-occupied-cell map, where exact distance ties (frequent
    on the default 20^3 grid) are resolved by the reference's sign-guessing order (jly_3ddt.cpp:1012-1132), by the exact
    transform in index order; the map itself is checked in test_cells_that_differ_from_the_reference_propagation."""
    G = load_register()[name]
    g = goicp_amd.GoICP(cfg_vector(**G["cfg"])[0], dt_mode=1)
    if name.startswith("pair1"):
        P = load_clouds("pair1"); g.set_clouds(P["t"], P["s"], P["tc"], P["sc"], P["tf"], P["sf"])
    else:
        m, d = _demo(name); g.set_clouds(m, d)
    g.build_dt()
    out = g.register()
    assert np.abs(out["R"] - np.array(G["R"])).max() < 1e-5 and np.abs(out["t"] - np.array(G["t"])).max() < 1e-5
    assert abs(float(out["error"]) - G["error"]) <= 1e-5 * G["error"]
    g.close()


def test_c3_pair_on_the_exact_grid_and_build_time():
    """BASELINE configs[2] with the grid built on the device: the reference's known answer within tolerance, BuildDT() <= 0.05 s."""
    import time
    from goicp_amd.synth import synth_pair
    G = json.load(open(os.path.join(GOLDEN, "c3_reference.json")))
    P = synth_pair(5000, seed=20260101, theta_range=(2.2, 3.1))
    v = goicp_amd.cfg_vector(MSEThresh=0.001, trimFraction=0.1, regularization=0.0, regularizationFPFH=0.0, cfpfh=0, ponderation=0, norm=2, distTransSize=300)
    g = goicp_amd.GoICP(v, dt_mode=1); g.set_clouds(P["t"], P["s"])
    g.build_dt()                                       # (first build in the process pays for the code-object load)
    t0 = time.perf_counter(); g.build_dt(); t_dt = time.perf_counter() - t0
    out = g.register()
    print("device DT build at 300^3: %.4f s" % t_dt)
    assert abs(float(out["error"]) - G["error"]) <= 1e-5 * G["error"]
    assert np.abs(out["R"] - np.array(G["R"])).max() < 1e-5 and np.abs(out["t"] - np.array(G["t"])).max() < 1e-5
    assert t_dt <= 0.05
    g.close()
