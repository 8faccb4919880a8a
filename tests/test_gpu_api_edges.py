"""GPU (-m gpu): boundary behaviour of the C ABI -- empty and ragged inputs, argument errors (status codes, never exits),
pop budgets, queue-slab regrowth."""
import ctypes as C
import os

import numpy as np
import pytest

import goicp_amd
from goicp_amd.binding import BnbJob, BnbResult
from helpers import GOLDEN, load_clouds, load_hotpath, synth_pair
from test_gpu_parity import engine_from_hotpath

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    H = load_hotpath("pair1_default"); P = load_clouds("pair1")
    e = engine_from_hotpath(H, P)
    e.set_rotation(0, H["call_R"][0])
    yield e, H, P
    e.close()


def test_empty_and_ragged_cube_lists(eng):
    e, H, P = eng
    ub, lb = e.eval_cubes(0, -1, np.zeros((0, 4), np.float32))
    assert len(ub) == 0 and len(lb) == 0
    cubes = H["cubes"][:23, 1:5].astype(np.float32)          # 23 = 2 full tiles of 8 + a tail of 7
    ub_all, lb_all = e.eval_cubes(0, 2, cubes)
    for n in (1, 7, 8, 9, 17):
        ub, lb = e.eval_cubes(0, 2, cubes[:n])
        assert np.array_equal(ub, ub_all[:n]) and np.array_equal(lb, lb_all[:n])
    # order independence: a cube's bounds do not depend on its tile mates
    perm = np.random.default_rng(0).permutation(23)
    ubp, lbp = e.eval_cubes(0, 2, cubes[perm])
    assert np.array_equal(ubp, ub_all[perm]) and np.array_equal(lbp, lb_all[perm])


def test_argument_errors_are_status_codes(eng):
    e, H, P = eng
    L = e.L
    cubes = np.zeros((8, 4), np.float32); ub = np.zeros(8, np.float32); lb = np.zeros(8, np.float32)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    assert L.goicp_engine_eval_cubes(e.h, 99, -1, p(cubes), 8, p(ub), p(lb)) == 1          # bad slot
    assert L.goicp_engine_eval_cubes(e.h, 0, 20, p(cubes), 8, p(ub), p(lb)) == 1           # rot level >= MAXROTLEVEL
    assert L.goicp_engine_eval_cubes(e.h, 0, -1, None, 8, p(ub), p(lb)) == 1               # null buffer
    assert L.goicp_engine_set_rotation(e.h, -1, p(np.eye(3, dtype=np.float32))) == 1
    J = (BnbJob * 1)(); R = (BnbResult * 1)()
    J[0].rot_level = 25
    assert L.goicp_engine_inner_bnb_batch(e.h, J, 1, R) == 1
    assert L.goicp_engine_inner_bnb_batch(e.h, J, 0, R) == 0                               # empty batch is fine


def test_unsupported_configurations_fail_at_create():
    H = load_hotpath("pair1_default"); P = load_clouds("pair1")
    base = dict(S=int(H["geom"][7]), geom4=(H["geom"][0], H["geom"][2], H["geom"][4], H["geom"][6]), dist=H["dist"], data_xyz=P["s"], weights=H["weights"],
                maxrotdis=H["maxrotdis"], model_xyz=P["t"], inlier_num=int(H["inlierNum"]))
    with pytest.raises(goicp_amd.EngineError):          # penalties requested without cell tables
        goicp_amd.Engine(reg=0.0005, **base)
    with pytest.raises(goicp_amd.EngineError):          # inlierNum out of range for a trimmed run
        goicp_amd.Engine(do_trim=True, **dict(base, inlier_num=0))
    with pytest.raises(goicp_amd.EngineError):          # no such device
        goicp_amd.Engine(device_id=63, **base)
    # clouds above 8192 points are streamed and support the geometric error only: penalties on such a cloud are refused
    S = int(H["geom"][7]); nb = 9000
    near = (H["nearest3"][:, 2].astype(np.int64) * S + H["nearest3"][:, 1]) * S + H["nearest3"][:, 0]
    big = dict(base, data_xyz=np.zeros((nb, 3), np.float32), weights=np.ones(nb, np.float32), maxrotdis=np.zeros(20 * nb, np.float32), inlier_num=nb)
    with pytest.raises(goicp_amd.EngineError):
        goicp_amd.Engine(reg=0.0005, nearest=near, cell_color=H["cellc"], cell_ptr=H["cell_ptr"], cell_idx=H["cell_idx"], model_color=P["tc"],
                         data_color=np.zeros(nb, np.int32), **big)
    e = goicp_amd.Engine(**big)                         # the same cloud without penalties is accepted
    e.close()
    # grid sides above GOICP_MAX_GRID = 1290 overflow the int32 linear cell index: refused before anything is read or allocated
    with pytest.raises(goicp_amd.EngineError):
        goicp_amd.Engine(**dict(base, S=1291))
    g = goicp_amd.GoICP(goicp_amd.cfg_vector(distTransSize=1291), dt_mode=1); g.set_clouds(P["t"], P["s"])
    with pytest.raises(goicp_amd.EngineError):
        g.build_dt()
    g.close()
    g = goicp_amd.GoICP(goicp_amd.cfg_vector(distTransSize=1001)); g.set_clouds(P["t"], P["s"])      # host builder: proven up to 1000
    with pytest.raises(goicp_amd.EngineError):
        g.build_dt()
    g.close()


def test_more_than_16_colour_codes_is_rejected():
    H = load_hotpath("pair1_default"); P = load_clouds("pair1")
    S = int(H["geom"][7])
    near = (H["nearest3"][:, 2].astype(np.int64) * S + H["nearest3"][:, 1]) * S + H["nearest3"][:, 0]
    dc = np.arange(len(P["s"]), dtype=np.int32) % 40 + 100
    with pytest.raises(goicp_amd.EngineError):
        goicp_amd.Engine(S=S, geom4=(H["geom"][0], H["geom"][2], H["geom"][4], H["geom"][6]), dist=H["dist"], data_xyz=P["s"], weights=H["weights"],
                         maxrotdis=H["maxrotdis"], model_xyz=P["t"], inlier_num=int(H["inlierNum"]), reg=0.0005, nearest=near, cell_color=H["cellc"],
                         cell_ptr=H["cell_ptr"], cell_idx=H["cell_idx"], model_color=P["tc"], data_color=dc)


def test_pop_budget_and_rerun(eng):
    e, H, P = eng
    cfg = H["cfg"]; init = tuple(np.float32(cfg[5:9]))
    jobs = [dict(R=H["call_R"][k], level=int(H["call_level"][k]), E=H["call_E"][k], sse=np.float32(H["sse"]), init=init) for k in range(len(H["call_level"]))]
    full = e.inner_bnb_batch(jobs)
    k = int(np.argmax([r["pops"] for r in full]))
    assert full[k]["status"] == 0 and full[k]["pops"] > 20
    capped = e.inner_bnb(max_pops=10, **jobs[k])
    assert capped["status"] == 5 and capped["pops"] == 11          # GOICP_ERR_LIMIT, stopped right after the budget
    again = e.inner_bnb(**jobs[k])                                  # unbounded again: same search, same answer
    assert again["status"] == 0 and again["val"] == full[k]["val"] and again["pops"] == full[k]["pops"] and np.array_equal(again["node"], full[k]["node"])


def test_queue_slab_regrowth_gives_the_same_answer(monkeypatch):
    """With a 32-node queue slab (test knob) and only 8 nodes in LDS, every longer search overflows and is re-run alone on
    16x larger slabs; the results must still be the reference's known answers, bit for bit."""
    monkeypatch.setenv("GOICP_HEAP_CAP", "32"); monkeypatch.setenv("GOICP_HEAP_LDS", "8")
    H = load_hotpath("pair1_default"); P = load_clouds("pair1")
    e = engine_from_hotpath(H, P)
    cfg = H["cfg"]; init = tuple(np.float32(cfg[5:9]))
    jobs = [dict(R=H["call_R"][k], level=int(H["call_level"][k]), E=H["call_E"][k], sse=np.float32(H["sse"]), init=init) for k in range(len(H["call_level"]))]
    res = e.inner_bnb_batch(jobs)
    assert max(r["pops"] for r in res) > 40            # long enough that queues exceed 24 nodes and the regrowth path runs
    for k, r in enumerate(res):
        assert r["status"] == 0 and r["val"] == H["call_val"][k]
        if r["found"]:
            assert np.array_equal(r["node"], H["call_node"][k])
    e.close()


def test_bench_prints_the_contract_line():
    """bench.py --gpus 1: one JSON line with the driver's keys, the roofline and (here skipped) cpu_baseline objects."""
    import json, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "2", "--warmup", "1", "--no-cpu", "--no-register"],
                         capture_output=True, text=True, timeout=180, cwd=root)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    J = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in J, k
    assert J["n_gpus"] == 1 and J["steps"] == 2 and J["warmup"] == 1 and J["higher_is_better"] is True and J["scaling"] == "weak" and J["vs_baseline"] is None
    assert J["unit"] == "cube-evals/s" and J["value"] > 1e6 and "workload" in J["config"]
    R = J["roofline"]
    assert R["bound"] == "hbm" and R["unit"] == "GB/s" and abs(R["frac"] - R["achieved"] / R["peak"]) < 1e-12 and 0 < R["frac"] < 1
    # value and the event-timed kernel agree: the step is one launch
    assert abs(J["ms_per_step"] - R["avg_launch_ms"]) < 0.15 * J["ms_per_step"]


def test_bench_refuses_more_ranks_than_gpus():
    """bench.py --gpus N must start N ranks or fail loudly: on this one-GPU box --gpus 2 exits non-zero without a JSON line."""
    import subprocess, sys, torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("two GPUs visible")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--no-cpu", "--no-register"],
                         capture_output=True, text=True, timeout=180, cwd=root)
    assert out.returncode != 0 and "--gpus 2" in out.stderr
    assert not [ln for ln in out.stdout.splitlines() if ln.startswith("{")]


def test_non_finite_coordinates_are_refused_and_tiny_clouds_register():
    """NaN / inf coordinates come back as an error from set_clouds (they would walk the reference-order selection loops out
    of their arrays); degenerate but finite inputs (one point, identical points, a 2^3 grid, far outside the grid) register."""
    from goicp_amd.synth import synth_pair
    P = synth_pair(200, seed=5)
    geo = dict(ponderation=0, regularization=0.0)
    for bad in (np.nan, np.inf):
        s = P["s"].copy(); s[3, 1] = bad
        g = goicp_amd.GoICP(goicp_amd.cfg_vector(**geo))
        with pytest.raises(goicp_amd.EngineError):
            g.set_clouds(P["t"], s)
        g.close()
    for t, s, cfg in ((P["t"], P["s"][:1], geo), (P["t"], np.repeat(P["s"][:1], 50, 0), geo), (P["t"], P["s"], dict(geo, distTransSize=2)),
                      (P["t"][:1], P["s"], geo), (P["t"], P["s"] * 50, dict(geo, MSEThresh=10.0)), (P["t"], P["s"], dict(geo, trimFraction=0.99)),
                      (P["t"], P["s"][:5], dict(regularization=0.0))):
        g = goicp_amd.GoICP(goicp_amd.cfg_vector(**cfg)); g.set_clouds(t, s); g.build_dt()
        out = g.register()
        assert np.isfinite(out["error"]) and np.all(np.isfinite(out["R"])) and np.all(np.isfinite(out["t"]))
        g.close()


def test_pooled_device_resources_can_be_released_and_switched_off():
    """Engines park their device blocks, stream/event triples and pinned buffer for the next engine; releasing the pool, and
    running with GOICP_CACHE_GB=0 (every engine allocates and frees for itself), must not change a result."""
    import json, subprocess, sys
    from helpers import load_register
    G = load_register()["pair1_default"]; P = load_clouds("pair1")
    from oracle.refbind import cfg_vector

    def reg():
        g = goicp_amd.GoICP(cfg_vector(**G["cfg"])[0]); g.set_clouds(P["t"], P["s"], P["tc"], P["sc"], P["tf"], P["sf"]); g.build_dt()
        out = g.register(); g.close()
        return float(out["error"]), out["comp"]
    first = reg()
    goicp_amd.load_library().goicp_engine_release_cached_memory()
    assert reg() == first == (G["error"], G["comp"])
    code = ("import sys; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
            "import goicp_amd\nfrom helpers import load_register, load_clouds\nfrom oracle.refbind import cfg_vector\n"
            "G = load_register()['pair1_default']; P = load_clouds('pair1')\n"
            "for _ in range(3):\n"
            "    g = goicp_amd.GoICP(cfg_vector(**G['cfg'])[0]); g.set_clouds(P['t'], P['s'], P['tc'], P['sc'], P['tf'], P['sf']); g.build_dt(); o = g.register(); g.close()\n"
            "    assert float(o['error']) == G['error'] and o['comp'] == G['comp']\n"
            "print('ok')\n") % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120, env=dict(os.environ, GOICP_CACHE_GB="0"))
    assert out.returncode == 0 and out.stdout.strip().endswith("ok"), out.stderr[-1500:]


def test_concurrent_engines_of_different_sizes_share_kernel_instantiations():
    """Two registrations in flight on two host threads whose clouds (1000 and 600 points, parity mode) run the SAME k_inner_bnb
    instantiation with different dynamic-LDS needs (76 KB and 50 KB). The dynamic-LDS ceiling of a kernel is a per-process
    attribute of the instantiation, so it is raised once to what the CU offers instead of per engine (a smaller engine created
    in between must not lower it under a larger engine's launch). Results equal the sequential runs bit for bit."""
    from concurrent.futures import ThreadPoolExecutor
    from helpers import synth_pair
    cfg = goicp_amd.cfg_vector(MSEThresh=0.001, trimFraction=0.0, regularization=0.0, regularizationFPFH=0.0, cfpfh=0, ponderation=0, norm=2, distTransSize=50)
    # (no outliers: an untrimmed search on a cloud with 10 % outliers runs for hours; these four take 0.2-11 s on the CPU oracle)
    pairs = [synth_pair(n, seed=sd, outlier_frac=0.0, noise=0.005) for n, sd in ((1000, 4101), (600, 4102), (1000, 4103), (600, 4104))]
    known = [0.6682798, 0.32610527, 0.5772468, 0.3744664]      # oracle (plain-C restatement), same pairs and config

    def reg(P):
        g = goicp_amd.GoICP(cfg); g.set_clouds(P["t"], P["s"]); g.build_dt()
        o = g.register(); g.close()
        return o
    seq = [reg(P) for P in pairs]
    for o, want in zip(seq, known):
        assert float(o["error"]) == float(np.float32(want)), (o["error"], want)
    for rep in range(3):
        with ThreadPoolExecutor(max_workers=2) as ex:
            outs = list(ex.map(reg, pairs))
        for a, b in zip(seq, outs):
            assert a["error"] == b["error"] and np.array_equal(a["R"], b["R"]) and np.array_equal(a["t"], b["t"])
            assert a["stats"]["trans_pops"] == b["stats"]["trans_pops"]


def test_register_budgets_return_the_best_pose_so_far():
    """GoICP::wall_budget_s / search_pop_budget (goicp_set_budget): a search that cannot run to convergence returns the incumbent with budget_hit set
    and status OK; without budgets the same pair converges to the reference's known answer (tests/golden/c3_reference.json)."""
    import json
    G = json.load(open(os.path.join(GOLDEN, "c3_reference.json")))
    P = synth_pair(5000, seed=20260101, theta_range=(2.2, 3.1))
    v = goicp_amd.cfg_vector(**G["cfg"])
    g = goicp_amd.GoICP(v, dt_mode=1); g.set_clouds(P["t"], P["s"]); g.build_dt()
    g.set_budget(0.0, 50)                       # 50 translation-queue pops per InnerBnB call: the first search already needs more
    out = g.register()
    assert out["budget_hit"] and np.isfinite(out["error"]) and out["error"] > G["error"] * 1.5
    assert [a for a, _ in out["trace"]][:2] == ["Init", "ICP"] and out["stats"]["trans_pops"] >= 50
    g.set_budget(1e-6, 0)                       # wall budget: checked before every rotation-queue pop
    out = g.register()
    assert out["budget_hit"] and out["stats"]["rot_pops"] <= 1
    g.set_budget(0.0, 0)
    out = g.register()
    assert not out["budget_hit"] and abs(float(out["error"]) - G["error"]) <= 1e-5 * G["error"] and out["stats"]["trans_pops"] == G["pops"]
    assert out["search"]["rounds"] > 0 and out["search"]["cube_evals_launched"] >= out["stats"]["cube_evals"]
    g.close()
