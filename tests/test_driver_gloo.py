"""CPU (-m "not gpu"): the N>1 path of the batch driver -- pair sharding + the single all_gather of result
records -- with world_size 2 over gloo; plus the host-side file plumbing (tsv rows, mol2 atom count)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import goicp_amd
from goicp_amd import driver
from helpers import GOLDEN

DATA = os.path.join(GOLDEN, "data")


def test_tsv_rows_and_atom_count():
    pairs = driver.read_tsv_pairs(os.path.join(DATA, "cavities_similar_BO1_clean.tsv"))
    assert len(pairs) == 383                       # 383 rows
    assert pairs[0] == ("2x86_3", "1eq2_6") and pairs[1] == ("2ktd_1", "4imo_2")      # source = field 3, target = field 4
    assert driver.mol2_atom_count(os.path.join(DATA, "cavities", "2x86_3_cavity6.mol2")) == 238
    assert driver.mol2_atom_count(os.path.join(DATA, "cavities", "2ktd_1_cavity6.mol2")) == 247


def test_shard_is_a_partition():
    for world in (1, 2, 3, 8):
        got = sorted(sum((driver.shard(383, r, world) for r in range(world)), []))
        assert got == list(range(383))


def _fake_register(tf, sf, nd, config, output, k):
    # deterministic stand-in for the GPU registration: the driver logic under test is scheduling + gather
    R = np.eye(3) * k
    return dict(R=R, t=np.array([k, k + 1, k + 2], float), error=10.0 * k, compat=nd, Nd=nd, seconds=0.01 * k, dt_seconds=0.0, cube_evals=100 * k, status=0)


def test_lpt_assignment_is_a_balanced_partition():
    rng = np.random.default_rng(3)
    costs = list(rng.integers(150, 401, 383) * rng.integers(150, 401, 383).astype(float))
    for world in (1, 2, 3, 8):
        parts = driver.lpt_assign(costs, world)
        assert sorted(sum(parts, [])) == list(range(383))
        loads = [sum(costs[i] for i in p) for p in parts]
        assert max(loads) - min(loads) <= max(costs)                   # greedy LPT: no rank exceeds another by more than one item
        for p in parts:
            assert [costs[i] for i in p] == sorted((costs[i] for i in p), reverse=True)   # longest first inside a rank
    assert driver.lpt_order([5.0, 9.0, 5.0, 1.0]) == [1, 0, 2, 3]


@pytest.fixture(scope="module")
def small_batch(tmp_path_factory):
    """14 rows of the configs[3] batch (2 real + 12 synthetic cavity pairs) as files."""
    d = str(tmp_path_factory.mktemp("c4"))
    tsv, cfgp, truth = goicp_amd.c4_batch.materialise(d, DATA, rows=14)
    return d, tsv, cfgp, truth


def test_c4_batch_files_are_readable_and_consistent(small_batch):
    from oracle.refbind import read_mol2, read_cfpfh, read_config
    d, tsv, cfgp, truth = small_batch
    jobs, n = driver.list_jobs(d, tsv)
    assert n == 14 and [j[0] for j in jobs] == list(range(1, 15))
    assert jobs[0][5:] == ("2x86_3", "1eq2_6") and jobs[0][3] == 238 and jobs[1][3] == 247
    cfg = read_config(cfgp)
    assert cfg["cfpfh"] == 1 and abs(cfg["regularizationFPFH"] - 5e-6) < 1e-12 and cfg["regularization"] == 0.0005 and cfg["distTransSize"] == 20
    for k, tf, sf, nd, cost, src, tgt in jobs[2:]:
        sx, sc = read_mol2(os.path.join(d, sf)); tx, tc = read_mol2(os.path.join(d, tf))
        assert len(sx) == nd and 150 <= nd <= 400 and cost == float(len(sx)) * len(tx)
        assert read_cfpfh(os.path.join(d, "cfpfh", src + "_cavity6.cfpfh"), nd).shape == (nd, 41)
        R, t = truth[k - 1]
        # the planted pose maps most source atoms (the inliers) onto the target cloud
        moved = sx @ R.T + t
        nn = np.sqrt(((moved[:, None, :] - tx[None, :, :]) ** 2).sum(-1)).min(1)
        assert np.median(nn) < 0.6                                   # Angstrom; noise 0.01 x scale 9-14
        assert driver.protein_paths(d, src, tgt) is not None
    assert driver.protein_paths(d, "2x86_3", "1eq2_6") is not None and driver.protein_paths(d, "2ktd_1", "4imo_2") is not None


def test_pair_rmsd_of_the_planted_pose_is_zero(small_batch, tmp_path):
    """The RMSD slot of a record: protein transformed with the run's rescaled R, t (applyTransformationProtein) against the aligned
    reference protein (computeRMSD); feeding the planted pose must give ~0 (4-decimal file rounding)."""
    d, tsv, cfgp, truth = small_batch
    jobs, _ = driver.list_jobs(d, tsv)
    k, tf, sf, nd, cost, src, tgt = jobs[5]
    R, t = truth[k - 1]
    out = os.path.join(str(tmp_path), "run.1", "similar%d.txt" % k); os.makedirs(os.path.dirname(out), exist_ok=True)   # a dot in a DIRECTORY name
    assert driver.rescaled_name(out) == os.path.join(str(tmp_path), "run.1", "similar%d_rescaled.txt" % k)
    assert driver.rescaled_name("output/similar7.txt") == "output/similar7_rescaled.txt"           # the reference's own layout: same name as its rule gives
    _write_rescaled(driver.rescaled_name(out), R, t)
    assert driver.pair_rmsd(d, src, tgt, out, k) < 2e-3
    assert np.isnan(driver.pair_rmsd(d, "nosuch", tgt, out))
    assert os.path.exists(os.path.join(d, "rot", "rot_%s_protein.mol2" % src))                    # the reference's name holds the finished file
    assert not [f for f in os.listdir(os.path.join(d, "rot")) if f.endswith(".tmp")]


def _write_rescaled(path, R, t):
    with open(path, "w") as f:
        f.write("Time: 0\nRotation Matrix:\n" + "\n".join("   %.9g   %.9g   %.9g" % tuple(r) for r in R) + "\nTranslation Vector:\n" + "\n".join("   %.9g" % v for v in t) + "\nError: 0\n")


def test_rows_that_share_a_source_do_not_share_the_transformed_protein(small_batch, tmp_path):
    """A source id repeats in up to ten rows of the shipped .tsv (3lzs_1 x10) and longest-first ordering runs them side by side: every
    row must measure ITS OWN transformed protein. Eight threads evaluate the same source at once, half with the planted pose (RMSD ~ 0)
    and half with a pose 5 Angstrom off; each must get its own answer."""
    import threading
    d, tsv, cfgp, truth = small_batch
    jobs, _ = driver.list_jobs(d, tsv)
    k, tf, sf, nd, cost, src, tgt = jobs[6]
    R, t = truth[k - 1]
    outs = []
    for i in range(8):
        o = os.path.join(str(tmp_path), "similar%d.txt" % (100 + i)); outs.append(o)
        _write_rescaled(driver.rescaled_name(o), R, t + (5.0 if i % 2 else 0.0))
    got = [None] * 8

    def run(i):
        for _ in range(6):
            got[i] = driver.pair_rmsd(d, src, tgt, outs[i], 100 + i)
            assert (got[i] > 4.0) if i % 2 else (got[i] < 2e-3), (i, got[i])
    errs = []

    def guarded(i):
        try:
            run(i)
        except BaseException as e:
            errs.append(e)
    ts = [threading.Thread(target=guarded, args=(i,)) for i in range(8)]
    [x.start() for x in ts]; [x.join() for x in ts]
    assert not errs, errs


def _slow_fake_register(tf, sf, nd, config, output, k):
    import time
    time.sleep(0.002 * (nd % 7))
    return _fake_register(tf, sf, nd, config, output, k)


def _worker(rank, world, port, tmp, base, tsv, schedule):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    recs, n_total = driver.run_batch(base, tsv, "unused", os.path.join(tmp, "out%d" % rank), _slow_fake_register, rank=rank, world=world, threads=2,
                                     schedule=schedule, dist=dist, counter_key="next_" + schedule)
    allr = driver.gather_records(recs, n_total, dist=dist)
    if rank == 0:
        np.save(os.path.join(tmp, "gathered_%s.npy" % schedule), allr)
    np.save(os.path.join(tmp, "local_%s_%d.npy" % (schedule, rank)), np.array(recs).reshape(-1, driver.REC))
    # a SECOND batch in the same process group, same counter key (bench.py: warm-up pass, then the timed pass): it must hand out every row again
    recs2, _ = driver.run_batch(base, tsv, "unused", os.path.join(tmp, "out%d" % rank), _slow_fake_register, rank=rank, world=world, threads=2,
                                schedule=schedule, dist=dist, counter_key="next_" + schedule)
    allr2 = driver.gather_records(recs2, n_total, dist=dist)
    if rank == 0:
        np.save(os.path.join(tmp, "gathered2_%s.npy" % schedule), allr2)
    # a batch in which ONE rank's registration fails: every rank must raise (nobody left waiting in the gather)
    def failing(tf, sf, nd, config, output, k):
        if rank == 1:
            raise RuntimeError("planted failure")
        return _slow_fake_register(tf, sf, nd, config, output, k)
    try:
        driver.run_batch(base, tsv, "unused", os.path.join(tmp, "out%d" % rank), failing, rank=rank, world=world, threads=2, schedule=schedule, dist=dist,
                         counter_key="next_" + schedule)
        raised = ""
    except RuntimeError as e:
        raised = str(e)
    open(os.path.join(tmp, "raised_%s_%d.txt" % (schedule, rank)), "w").write(raised)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("schedule", ["dynamic", "lpt", "rr"])
def test_two_rank_gloo_batch(tmp_path, small_batch, schedule):
    """world_size 2 over gloo: every row registered exactly once whatever the scheduler, one all_gather, records ordered by pair id."""
    base, tsv, cfgp, truth = small_batch
    world = 2
    port = 29500 + (os.getpid() % 2000) + {"dynamic": 0, "lpt": 1, "rr": 2}[schedule]
    mp.spawn(_worker, args=(world, port, str(tmp_path), base, tsv, schedule), nprocs=world, join=True)
    allr = np.load(os.path.join(str(tmp_path), "gathered_%s.npy" % schedule))
    assert allr.shape == (14, driver.REC)
    assert list(allr[:, 0]) == [float(k) for k in range(1, 15)]
    assert np.array_equal(np.load(os.path.join(str(tmp_path), "gathered2_%s.npy" % schedule))[:, :21], allr[:, :21], equal_nan=True)    # second batch of the group: all rows again
    r0 = open(os.path.join(str(tmp_path), "raised_%s_0.txt" % schedule)).read(); r1 = open(os.path.join(str(tmp_path), "raised_%s_1.txt" % schedule)).read()
    assert "planted failure" in r1 and "rank(s) failed" in r0
    assert allr[0, 15] == 238 and allr[1, 15] == 247            # Nd read from line 6 of the source mol2
    assert allr[1, 13] == 20.0 and np.array_equal(allr[1, 10:13], [2, 3, 4])
    l0 = np.load(os.path.join(str(tmp_path), "local_%s_0.npy" % schedule)); l1 = np.load(os.path.join(str(tmp_path), "local_%s_1.npy" % schedule))
    assert len(l0) + len(l1) == 14 and not set(l0[:, 0]) & set(l1[:, 0])
    assert set(l0[:, 21]) <= {0.0} and set(l1[:, 21]) <= {1.0}
    jobs, _ = driver.list_jobs(base, tsv)
    if schedule == "lpt":
        parts = driver.lpt_assign([j[4] for j in jobs], world)
        assert sorted(l0[:, 0]) == sorted(jobs[i][0] for i in parts[0]) and sorted(l1[:, 0]) == sorted(jobs[i][0] for i in parts[1])
    if schedule == "rr":
        assert sorted(l0[:, 0]) == [1, 3, 5, 7, 9, 11, 13]
    if schedule == "dynamic":
        assert len(l0) >= 1 and len(l1) >= 1                    # both ranks pulled work from the shared counter


def test_library_exports_every_declared_symbol():
    """include/*.h vs the built C-ABI library: every declared entry point is exported (no compute calls here)."""
    import re
    L = goicp_amd.load_library()
    names = []
    for hdr in ("goicp_engine.h", "goicp_c.h"):
        txt = open(os.path.join(os.path.dirname(GOLDEN), "..", "include", hdr)).read()
        txt = re.sub(r"#ifdef GOICP_PROBES.*?#endif", "", txt, flags=re.S)     # entry points of the diagnosis build (make PROBES=1) only
        txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
        names += re.findall(r"\b(goicp_[a-z0-9_]+)\s*\(", txt)
    names = sorted(set(names))
    assert len(names) >= 40
    for n in names:
        assert hasattr(L, n), n


def test_engine_fails_loudly_without_gpu():
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(goicp_amd.EngineError):
        goicp_amd.Engine(S=2, geom4=(0, 0, 0, 1), dist=np.zeros(8), data_xyz=np.zeros((4, 3)), weights=np.ones(4), maxrotdis=np.zeros(80),
                         model_xyz=np.zeros((4, 3)), inlier_num=4)
    g = goicp_amd.GoICP(goicp_amd.cfg_vector())
    g.set_clouds(np.random.rand(30, 3), np.random.rand(30, 3))
    g.build_dt()
    with pytest.raises(goicp_amd.EngineError):
        g.register()


def test_threaded_batch_returns_the_same_records_in_row_order(tmp_path):
    """run_batch(threads=N) keeps N registrations of a rank in flight; the records must equal the sequential run's, in row order."""
    tsv = os.path.join(DATA, "cavities_similar_BO1_clean.tsv")
    seq, n1 = driver.run_batch(DATA, tsv, "unused", str(tmp_path), _fake_register, rmsd=False)
    par, n2 = driver.run_batch(DATA, tsv, "unused", str(tmp_path), _fake_register, threads=4, rmsd=False)
    assert n1 == n2 == 383 and len(seq) == len(par) == 2
    assert all(np.array_equal(a, b, equal_nan=True) for a, b in zip(seq, par))


def test_worker_threads_leave_two_cpus_of_the_budget_free():
    """driver.worker_threads: spinning worker threads never fill this rank's share of the CPU budget (affinity set or cgroup quota)."""
    from goicp_amd import driver
    assert driver.host_cpu_budget() >= 1
    assert driver.worker_threads(16, 1, 16) == 14 and driver.worker_threads(16, 8, 128) == 14
    assert driver.worker_threads(4, 1, 16) == 4 and driver.worker_threads(16, 1, 2) == 1 and driver.worker_threads(16, 2, 16) == 6


def test_batch_driver_tells_the_library_how_many_registrations_are_in_flight(small_batch, tmp_path, monkeypatch):
    """run_batch hands its thread count to the registration function's configure_batch; the GPU one (driver.gpu_register_fn) turns it into the
    two process-wide settings of the file pipeline: the speculation depth (128 alone ... 16 from 16 in flight on) and ICP3D::Run in one launch
    from three registrations in flight on (include/goicp_c.h: goicp_set_batch_speculation, goicp_set_batch_icp_one_launch)."""
    d, tsv, cfgp, _ = small_batch
    seen = []

    def reg(tf, sf, nd, config, output, k):
        return _fake_register(tf, sf, nd, config, output, k)
    reg.configure_batch = seen.append
    recs, n = driver.run_batch(d, tsv, cfgp, str(tmp_path / "o"), reg, threads=5, rmsd=False)
    assert seen == [5] and len(recs) == 14
    assert [driver.batch_speculation(t) for t in (1, 2, 3, 8, 14, 16, 64)] == [128, 128, 85, 32, 18, 16, 16]
    calls = []
    from goicp_amd import binding
    monkeypatch.setattr(binding, "set_batch_speculation", lambda depth: calls.append(("spec", depth)))
    monkeypatch.setattr(binding, "set_batch_icp_one_launch", lambda on: calls.append(("icp", on)))
    for t in (1, 2, 3, 14):
        driver.gpu_register_fn(d).configure_batch(t)
    assert calls == [("spec", 128), ("icp", 0), ("spec", 128), ("icp", 0), ("spec", 85), ("icp", 1), ("spec", 18), ("icp", 1)]
