"""CPU (-m "not gpu"): the plain-C oracle against the committed known answers that
tests/golden/make_golden.py produced with the compiled reference, and against the reference's own
shipped goldens (output/similar1.txt, demo/output.txt, cavitiesN/*.xyz)."""
import os

import numpy as np
import pytest

from oracle.refbind import Oracle, cfg_vector, read_mol2, normalise_pair, read_cfpfh, error_trace, read_txt_cloud
from helpers import GOLDEN, load_clouds, load_hotpath, load_register, rot_cloud, oracle_for

DATA = os.path.join(GOLDEN, "data")
HOT = ["pair1_default", "pair1_trim", "pair1_fpfh", "pair1_norm1", "pair2_S50"]


def test_normalised_clouds_match_shipped_xyz():
    # jly_main.cpp:75-93 / transformation.cpp:340-350 known answer: cavitiesN/*_sim1N.xyz
    sx, sc = read_mol2(os.path.join(DATA, "cavities", "2x86_3_cavity6.mol2"))
    tx, tc = read_mol2(os.path.join(DATA, "cavities", "1eq2_6_cavity6.mol2"))
    s, t, scale, _, _ = normalise_pair(sx, tx)
    for arr, col, f in ((s, sc, "2x86_3_cavity6_sim1N.xyz"), (t, tc, "1eq2_6_cavity6_sim1N.xyz")):
        g = np.loadtxt(os.path.join(DATA, "cavitiesN", f), skiprows=1)
        assert np.array_equal(g[:, :3].astype(np.float32), arr)
        assert np.array_equal(g[:, 3].astype(np.int32), col)


@pytest.mark.parametrize("name", HOT)
def test_dt_build_matches_reference_tables(name):
    H = load_hotpath(name)
    P = load_clouds("pair2" if name.startswith("pair2") else "pair1")
    o = oracle_for(P, H["cfg"])
    dist, near, cc, _ = o.dt_dump()
    assert np.array_equal(o.geom, H["geom"])
    assert np.array_equal(dist, H["dist"])            # bit-exact float grid
    assert np.array_equal(near, H["nearest3"])
    assert np.array_equal(cc, H["cellc"])


@pytest.mark.parametrize("name", HOT)
def test_initialize_tables(name):
    H = load_hotpath(name)
    P = load_clouds("pair2" if name.startswith("pair2") else "pair1")
    o = oracle_for(P, H["cfg"], H)
    o.initialize()
    assert np.array_equal(o.weights(), H["weights"])
    assert np.array_equal(o.maxrotdis(), H["maxrotdis"])
    assert o.inlier_num() == int(H["inlierNum"])
    assert np.float32(o.sse_thresh()) == np.float32(H["sse"])


@pytest.mark.parametrize("name", HOT)
def test_cube_bounds_and_inner_bnb(name):
    """(cube)->(ub,lb) rows pushed by the reference's InnerBnB, and InnerBnB's return value / best node: bit-exact."""
    H = load_hotpath(name)
    P = load_clouds("pair2" if name.startswith("pair2") else "pair1")
    o = oracle_for(P, H["cfg"], H)
    o.initialize()
    cubes = H["cubes"]
    for k in range(len(H["call_level"])):
        o.set_rotated(rot_cloud(H["call_R"][k], P["s"]))
        lvl = int(H["call_level"][k])
        rows = cubes[cubes[:, 0] == k]
        for row in rows:
            ub, lb = o.eval_cube(lvl, row[1], row[2], row[3], row[4])
            assert ub == np.float32(row[5]) and lb == np.float32(row[6])
        o.set_opt_error(H["call_E"][k])
        val, node = o.inner_bnb(lvl)
        assert val == H["call_val"][k]
        assert np.array_equal(node[:4], H["call_node"][k])
    for row in H["corners"]:
        k = int(row[0])
        o.set_rotated(rot_cloud(H["call_R"][k], P["s"]))
        assert o.check_compatibilities(row[1], row[2], row[3]) == int(row[4])
        if int(H["cfg"][13]) != 0:
            assert o.sum_fpfh(row[1], row[2], row[3]) == np.float32(row[5])


@pytest.mark.parametrize("name", ["pair1_default", "pair1_trim", "pair1_fpfh"])
def test_icp_known_answers(name):
    I = dict(np.load(os.path.join(GOLDEN, "icp_%s.npz" % name)))
    P = load_clouds("pair1")
    o = oracle_for(P, I["cfg"])
    o.initialize()
    for k in range(len(I["err"])):
        e, R, t, corr, dis = o.icp(I["Rin"][k], I["tin"][k])
        assert e == I["err"][k]
        assert np.abs(R - I["Rout"][k]).max() < 1e-12 and np.abs(t - I["tout"][k]).max() < 1e-12
        assert np.array_equal(corr, I["corr"][k])       # 1-NN ids incl. trimmed ordering
        assert np.array_equal(dis, I["dis"][k])


@pytest.mark.parametrize("name", ["plain", "trim", "geo"])
def test_icp_tie_rule_known_answers(name):
    """Exact float distance ties in the 1-NN step (duplicated model points, queries on lattice midpoints): the compiled
    reference's correspondences (tests/golden/make_icp_ties.py) pin the kd-tree visiting order the oracle restates
    (nanoflann.hpp:927-1184). A lowest-index rule fails this test on most rows."""
    I = dict(np.load(os.path.join(GOLDEN, "icp_ties.npz")))
    o = Oracle(); o.set_config(I[name + "_cfg"]); o.set_clouds(I[name + "_model"], I[name + "_data"]); o.build_dt(); o.initialize()
    for k in range(len(I[name + "_err"])):
        e, R, t, corr, dis = o.icp(I[name + "_Rin"][k], I[name + "_tin"][k])
        assert e == I[name + "_err"][k]
        assert np.abs(R - I[name + "_Rout"][k]).max() < 1e-12 and np.abs(t - I[name + "_tout"][k]).max() < 1e-12
        assert np.array_equal(corr, I[name + "_corr"][k]) and np.array_equal(dis, I[name + "_dis"][k])


def test_host_kdorder_equals_the_oracle_tree():
    """csrc/host_kdorder.hpp (the product's tie-breaker, pure host code) against the oracle's restatement of the same tree, on
    tie-heavy and on random clouds incl. leaf-sized and single-point models."""
    import ctypes as C
    import goicp_amd
    from oracle import refbind
    L = goicp_amd.load_library(); O = C.CDLL(refbind.ORACLE_SO)
    I = dict(np.load(os.path.join(GOLDEN, "icp_ties.npz")))
    rng = np.random.default_rng(77)
    cases = [(I["plain_model"], I["plain_data"]), (I["trim_model"], I["trim_data"])]
    for nm in (1, 7, 10, 11, 64, 500, 3000):
        m = rng.uniform(-1, 1, (nm, 3)).astype(np.float32)
        if nm >= 64:
            m[rng.choice(nm, nm // 8)] = m[rng.choice(nm, nm // 8)]          # duplicates
        m = (np.round(m * 16) / 16).astype(np.float32) if nm in (64, 500) else m     # coarse lattice: many ties
        q = np.concatenate([rng.uniform(-1.2, 1.2, (300, 3)), (np.round(rng.uniform(-1, 1, (300, 3)) * 16) / 16 + 1 / 32)]).astype(np.float32)
        cases.append((m, q))
    for m, q in cases:
        m = np.ascontiguousarray(m, np.float32); q = np.ascontiguousarray(q, np.float32); n = len(q)
        a = np.zeros(n, np.int32); da = np.zeros(n, np.float32); b = np.zeros(n, np.int32); db = np.zeros(n, np.float32)
        vp = lambda x: x.ctypes.data_as(C.c_void_p)
        assert L.goicp_kdorder_nn(vp(m), len(m), vp(q), n, vp(a), vp(da)) == 0
        O.orc_kd_nn(vp(m), len(m), vp(q), n, vp(b), vp(db))
        assert np.array_equal(a, b) and np.array_equal(da, db)
        d = q[:, None, :] - m[None, :, :]
        d2 = ((d[..., 0] * d[..., 0] + d[..., 1] * d[..., 1]) + d[..., 2] * d[..., 2]).astype(np.float32)
        assert np.array_equal(da, d2.min(1))                                  # the search is exact; only the tie choice is tree-dependent


REG_FAST = ["pair1_default", "pair1_fpfh", "pair1_fpfh33", "pair1_color8", "pair1_norm1", "pair1_trim", "pair1_S50", "pair1_S100_geo"]


@pytest.mark.parametrize("name", REG_FAST)
def test_register_known_answers(name):
    G = load_register()[name]
    P = load_clouds("pair1")
    v, _ = cfg_vector(**G["cfg"])
    o = oracle_for(P, v)
    out = o.register()
    assert float(out["error"]) == G["error"] and out["comp"] == G["comp"]
    assert np.abs(out["R"] - np.array(G["R"])).max() < 1e-12 and np.abs(out["t"] - np.array(G["t"])).max() < 1e-12
    assert error_trace(out["log"]) == [tuple(x) for x in G["trace"]]


def test_register_pair1_matches_shipped_output_file():
    # output/similar1.txt:1-11 (R,t printed %12.7f; Error 6 significant digits; Compatibilities)
    txt = open(os.path.join(DATA, "output", "similar1.txt")).read().split()
    nums = [float(x) for x in txt if x.replace("-", "").replace(".", "").isdigit()]
    Rg = np.array(nums[1:10]).reshape(3, 3); tg = np.array(nums[10:13]); err = nums[13]; comp = int(nums[14])
    P = load_clouds("pair1")
    v, _ = cfg_vector()
    out = oracle_for(P, v).register()
    assert np.abs(out["R"] - Rg).max() < 5.1e-8 and np.abs(out["t"] - tg).max() < 5.1e-8
    assert float("%g" % out["error"]) == err and out["Nd"] - out["comp"] == comp


def test_register_demo_rand_known_answers():
    regs = load_register()
    mr = read_txt_cloud(os.path.join(DATA, "demo", "model_rand.txt")); dr = read_txt_cloud(os.path.join(DATA, "demo", "data_rand.txt"))
    for nm in ("rand_untrimmed", "rand_trim"):
        G = regs[nm]
        v, _ = cfg_vector(**G["cfg"])
        o = Oracle(); o.set_config(v); o.set_clouds(mr, dr); o.build_dt()
        out = o.register()
        assert float(out["error"]) == G["error"]
        assert np.abs(out["R"] - np.array(G["R"])).max() < 1e-12 and np.abs(out["t"] - np.array(G["t"])).max() < 1e-12


@pytest.mark.slow
def test_register_pair1_on_300_cubed_known_answer():
    """BASELINE configs[1]: the shipped cavity pair on a 300^3 DT (error 8.30483, 132 incompatibilities)."""
    G = load_register()["pair1_S300"]
    P = load_clouds("pair1")
    v, _ = cfg_vector(**G["cfg"])
    out = oracle_for(P, v).register()
    assert float(out["error"]) == G["error"] and out["comp"] == G["comp"] and float("%g" % G["error"]) == 8.30483 and G["Nd"] - G["comp"] == 132
    assert np.abs(out["R"] - np.array(G["R"])).max() < 1e-12 and np.abs(out["t"] - np.array(G["t"])).max() < 1e-12


def test_register_streamed_pair_known_answer():
    """The oracle on the repo's seeded 9 000-point synthetic pair against the compiled reference's result
    (tests/golden/stream9k_reference.json, generator tests/golden/make_synth_reference.py)."""
    import json
    from helpers import synth_pair
    G = json.load(open(os.path.join(GOLDEN, "stream9k_reference.json")))
    P = synth_pair(G["n"], seed=G["seed"], theta_range=tuple(G["pair"]["theta_range"]), outlier_frac=G["pair"]["outlier_frac"])
    v, _ = cfg_vector(**G["cfg"])
    o = Oracle(); o.set_config(v); o.set_clouds(P["t"], P["s"]); o.build_dt()
    out = o.register()
    assert float(out["error"]) == G["error"]
    assert np.abs(out["R"] - np.array(G["R"])).max() < 1e-12 and np.abs(out["t"] - np.array(G["t"])).max() < 1e-12
    assert [(a, float("%g" % b)) for a, b in error_trace(out["log"])] == [tuple(x) for x in G["trace"]]


@pytest.mark.slow
def test_register_pair2_known_answer():
    G = load_register()["pair2_default"]
    P = load_clouds("pair2")
    v, _ = cfg_vector(**G["cfg"])
    out = oracle_for(P, v).register()
    assert float(out["error"]) == G["error"] and out["comp"] == G["comp"]
    assert np.abs(out["R"] - np.array(G["R"])).max() < 1e-12 and np.abs(out["t"] - np.array(G["t"])).max() < 1e-12
    # Error 16.1742, Compatibilities 118, t=(0.0429688,-0.1132812,0.0039062)
    assert "%g" % out["error"] == "16.1742" and out["Nd"] - out["comp"] == 118
