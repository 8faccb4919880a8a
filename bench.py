#!/usr/bin/env python3
"""bench.py -- translation-BnB cube-evaluations/s (+ end-to-end pair registrations/s) on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Workload (BASELINE.json configs[2], the one the metric is quoted on): seeded synthetic 5 000-point shell cloud vs a
rotated + noised copy with 10 % outliers, trimFraction 0.1, 300^3 distance transform, MSEThresh 1e-3, penalties off.
One "step" = one launch group of the cube sweep: 16 rotations x 2 bound variants (rho = 0 and rho = maxRotDis[3]) x all
4 680 cubes of octree levels 1-4 of [-0.5,0.5]^3 = 149 760 cube evaluations, every input resident in HBM beforehand.
Per rank the work is fixed (weak scaling): pairs are independent, there is no data-path collective.
The same run also registers the pair end to end (Register(), and BuildDT()+Register()) and, on rank 0 at N=1, times
the CPU reference (oracle/_ref, the unmodified reference compiled in place) on a bounded replay of the same searches.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

N_POINTS = 5000
DT_SIZE = 300
TRIM = 0.1
N_ROT = 16
PEAK_HBM_GBPS = 8000.0          # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)
TILE_CUBES = 8                  # cubes that share one register-resident copy of the cloud (TC in engine.hip)


def level_cubes():
    rows = []
    for L in (1, 2, 3, 4):
        n = 2 ** L; w = np.float32(1.0 / n)
        for k in range(n):
            for j in range(n):
                for i in range(n):          # x fastest: adjacent cubes of a tile are x-neighbours
                    rows.append((np.float32(-0.5) + i * w, np.float32(-0.5) + j * w, np.float32(-0.5) + k * w, w))
    return np.array(rows, np.float32)


def algorithmic_bytes_per_cube_eval(nd):
    # Nd x 4 B (one fp32 DT value per point) + 8 B (ub, lb) + Nd x 20 B / T (xyz + weight + rho per tile)
    return nd * 4 + 8 + nd * 20 / TILE_CUBES


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--points", type=int, default=N_POINTS)
    ap.add_argument("--dt", type=int, default=DT_SIZE)
    ap.add_argument("--trim", type=float, default=TRIM, help="trimFraction (configs[4], the 50 000-point stress shape, is untrimmed)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--no-register", action="store_true", help="skip the end-to-end registration leg")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--cavity-pairs", type=int, default=8, help="cavity-like pairs registered per rank")
    ap.add_argument("--cavity-threads", type=int, default=8, help="host threads (= concurrent registrations, one engine + stream each) per rank")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1")); local = int(os.environ.get("LOCAL_RANK", "0"))
    # under torch.distributed.run (RANK/MASTER_* set) the collectives are exercised even with one rank
    use_dist = world > 1 or ("RANK" in os.environ and "MASTER_PORT" in os.environ)
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl")       # RCCL on ROCm
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    import goicp_amd
    from goicp_amd.synth import synth_pair
    goicp_amd.load_library()                  # fails loudly if the HIP extension is missing

    nd = args.points
    P = synth_pair(nd, seed=20260101, theta_range=(2.2, 3.1))
    cfg = goicp_amd.cfg_vector(MSEThresh=0.001, trimFraction=args.trim, regularization=0.0, regularizationFPFH=0.0, cfpfh=0,
                               ponderation=0, norm=2, distTransSize=args.dt)
    g = goicp_amd.GoICP(cfg, device_id=local)
    g.set_clouds(P["t"], P["s"])
    t0 = time.time(); g.build_dt(); t_dt = time.time() - t0
    g.initialize()
    eng = g.engine()

    # ---- micro-batch inputs, resident in HBM before the timed region ------------------------------------
    rng = np.random.default_rng(20260102)
    rots = []
    for _ in range(N_ROT):
        aa = rng.normal(size=3); aa = aa / np.linalg.norm(aa) * rng.uniform(0, np.pi)
        th = np.linalg.norm(aa); k = aa / th
        K = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
        rots.append((np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * K @ K).astype(np.float32))
    for s, R in enumerate(rots):
        eng.set_rotation(s, R)
    cubes_h = level_cubes()
    ncube = len(cubes_h)
    cubes = torch.from_numpy(cubes_h).to(dev)
    ub = torch.empty((N_ROT * 2, ncube), dtype=torch.float32, device=dev); lb = torch.empty_like(ub)
    torch.cuda.synchronize()
    stream = torch.cuda.ExternalStream(eng.stream(), device=dev)     # the engine's own HIP stream

    g_slots = np.repeat(np.arange(N_ROT, dtype=np.int32), 2)
    g_levels = np.tile(np.array([-1, 3], np.int32), N_ROT)

    def step():
        # ONE launch: 32 (rotation, rho-level) groups x 585 tiles of 8 cubes = 149 760 cube evaluations
        eng.eval_cubes_multi_dev(g_slots, g_levels, cubes.data_ptr(), ncube, ub.data_ptr(), lb.data_ptr())

    for _ in range(args.warmup):
        step()
    eng.synchronize()
    ev0 = torch.cuda.Event(enable_timing=True); ev1 = torch.cuda.Event(enable_timing=True)
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    with torch.cuda.stream(stream):
        ev0.record(stream)
    for _ in range(args.steps):
        step()
    with torch.cuda.stream(stream):
        ev1.record(stream)
    eng.synchronize(); torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    ev_ms = ev0.elapsed_time(ev1)
    if use_dist:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    evals_per_step = N_ROT * 2 * ncube
    value = world * evals_per_step * args.steps / elapsed
    launches = args.steps
    kern_ms = ev_ms / launches
    bytes_per_launch = algorithmic_bytes_per_cube_eval(nd) * evals_per_step
    achieved = bytes_per_launch / (kern_ms * 1e-3) / 1e9
    # sanity: a finite checksum of the outputs (work was really done)
    checksum = float(ub.double().sum().item() + lb.double().sum().item())
    assert np.isfinite(checksum)

    # ---- end-to-end registration of the same pair ------------------------------------------------------
    reg = None
    if not args.no_register:
        g.clear()
        g.record_jobs(True)
        if use_dist:
            dist.barrier()
        t0 = time.perf_counter(); out = g.register(); t_reg = time.perf_counter() - t0
        if use_dist:
            tt = torch.tensor([t_reg], dtype=torch.float64, device=dev); dist.all_reduce(tt, op=dist.ReduceOp.MAX); t_reg_max = float(tt.item())
        else:
            t_reg_max = t_reg
        reg = dict(register_seconds=t_reg_max, dt_build_seconds=t_dt, pair_registrations_per_sec=world / t_reg_max,
                   pair_registrations_per_sec_incl_dt_build=world / (t_reg_max + t_dt), error=float(out["error"]),
                   cube_evals=out["stats"]["cube_evals"], rot_pops=out["stats"]["rot_pops"], inner_calls=out["stats"]["inner_calls"],
                   inner_launches=out["stats"]["inner_launches"], engine_kernel_ms=out["kernel_ms"],
                   dR_vs_planted=float(np.abs(out["R"] - P["Rtrue"]).max()), dt_vs_planted=float(np.abs(out["t"] - P["ttrue"]).max()))

    # ---- cavity-pair registrations/s: BASELINE configs[3] shape (cavity-sized clouds, default config + c-FPFH), sharded ----
    cav = None
    if not args.no_register:
        n_pairs = args.cavity_pairs
        cfg4 = goicp_amd.cfg_vector(cfpfh=1, regularizationFPFH=0.000005)      # shipped defaults otherwise (20^3 DT, compat penalty, ponderation)
        rngc = np.random.default_rng(20260103)
        sizes = rngc.integers(150, 401, size=n_pairs * world)
        mine = list(range(rank, n_pairs * world, world))                       # static round-robin shard, no data-path collective
        pairs = [synth_pair(int(sizes[k]), seed=777000 + k, noise=0.01, outlier_frac=0.05, theta_range=(0.3, np.pi), corr_features=True) for k in mine]
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()

        def one(Pc):
            # one engine + one HIP stream per registration; ctypes releases the GIL, so several pairs are in flight on the GPU
            gc = goicp_amd.GoICP(cfg4, device_id=local)
            gc.set_clouds(Pc["t"], Pc["s"], Pc["tc"], Pc["sc"], Pc["tf"], Pc["sf"])
            gc.build_dt()
            oc = gc.register()
            gc.close()
            return float(oc["error"]), oc["stats"]["cube_evals"]
        if args.cavity_threads > 1:
            from concurrent.futures import ThreadPoolExecutor
            with ThreadPoolExecutor(max_workers=args.cavity_threads) as ex:
                outs_c = list(ex.map(one, pairs))
        else:
            outs_c = [one(Pc) for Pc in pairs]
        errs = [o[0] for o in outs_c]; cev = sum(o[1] for o in outs_c)
        t_cav = time.perf_counter() - t0
        rec = torch.tensor([t_cav, float(cev)], dtype=torch.float64, device=dev)
        if use_dist:
            tmax = rec.clone(); dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            tsum = rec.clone(); dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
            t_cav = float(tmax[0].item()); cev = int(tsum[1].item())
            # the one exchange of the batch driver: all_gather of the per-pair results (RCCL)
            loc = torch.tensor(errs, dtype=torch.float64, device=dev)
            outs = [torch.empty_like(loc) for _ in range(world)]
            dist.all_gather(outs, loc)
        cav = dict(pairs=n_pairs * world, seconds=t_cav, cavity_pair_registrations_per_sec=n_pairs * world / t_cav, cube_evals=cev, concurrent_registrations_per_gpu=args.cavity_threads,
                   config="default config.txt + cfpfh=1, regularizationFPFH=5e-6; synthetic cavity-like pairs, N in [150,400]")

    # ---- CPU baseline: the unmodified reference (oracle/_ref) replaying the same InnerBnB calls, bounded -------
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu:
        cpu = cpu_baseline(g, P, cfg, args, reg)
        if cpu is not None and cav is not None:
            cpu.update(cpu_cavity_baseline(pairs[:8], cfg4, cav))

    # the DT / engine can go now
    g.close()

    if rank == 0:
        traffic = None
        pj = os.path.join(ROOT, "profiles", "r01_summary.json")
        if os.path.exists(pj):
            try:
                traffic = json.load(open(pj)).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        default_shape = nd == N_POINTS and args.dt == DT_SIZE and abs(args.trim - TRIM) < 1e-9
        if not default_shape:
            traffic = None                      # the committed PMC summary belongs to the default shape only
        geom = [(256, 1, 256, 0), (512, 2, 256, 0), (1024, 1, 1024, 0), (2048, 4, 512, 1), (4096, 8, 512, 1), (5120, 10, 512, 1)]
        kname = next(("k_eval_cubes<%d,%d,8,%s>" % (b, p, "true" if j else "false") for lim, p, b, j in geom if nd <= lim), "k_eval_cubes_stream")
        label = "configs[2]" if default_shape else ("configs[4] shape" if nd >= 50000 else "non-default shape")
        line = {
            "metric": "translation_bnb_cube_evals_per_sec", "value": value, "unit": "cube-evals/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%s: synthetic %d-pt shell cloud vs rotated+noised copy (10%% outliers), trimFraction=%.1f, %d^3 DT; "
                                   "step = 16 rotations x 2 rho-variants x 4680 cubes (octree levels 1-4)" % (label, nd, args.trim, args.dt),
                       "points": nd, "dt_size": args.dt, "trimFraction": args.trim, "cube_evals_per_step": evals_per_step, "parallelism": "pairs sharded, 1 rank per GPU"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": PEAK_HBM_GBPS, "unit": "GB/s", "frac": achieved / PEAK_HBM_GBPS, "traffic": traffic,
                         "kernel": kname, "avg_launch_ms": kern_ms, "algorithmic_bytes_per_launch": bytes_per_launch,
                         "point_lookups_per_sec": value / world * nd},
            "cpu_baseline": cpu,
            "registration": reg,
            "cavity_batch": cav,
        }
        print(json.dumps(line))
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


def cpu_baseline(g, P, cfg, args, reg):
    """Times the reference's own InnerBnB (oracle/_ref) on the host cores of this box, replaying the first InnerBnB calls of the
    GPU registration (same rotations, levels and incumbents => identical searches) until ~cpu_seconds of CPU time are spent."""
    from oracle import refbind
    from goicp_amd.synth import rot_cloud
    jobs = g.job_log() if reg is not None else []
    if not jobs:
        return None
    kind = "reference" if os.path.exists(refbind.REF_SO) else "port"
    dist_grid, _, _ = g.dt_dump()
    if kind == "reference":
        r = refbind.Ref(); r.set_config(cfg); r.set_clouds(P["t"], P["s"]); r.load_dt(g.S, g.geom, dist_grid)
    else:
        r = refbind.Oracle(); r.set_config(cfg); r.set_clouds(P["t"], P["s"]); r.load_dt(g.S, g.geom, dist_grid)
    r.initialize()
    r.counters(reset=True)
    spent = 0.0; evals = 0; ncalls = 0; mismatches = 0
    for j in jobs:
        r.set_rotated(rot_cloud(j["R"], P["s"]))
        r.set_opt_error(j["E"])
        t0 = time.perf_counter()
        val, _ = r.inner_bnb(j["level"], want_node=False)
        spent += time.perf_counter() - t0
        ncalls += 1
        evals += j["cube_evals"]           # the replayed search is the same search (checked below)
        if abs(float(val) - float(j["val"])) > 1e-5 * max(1.0, abs(float(val))):
            mismatches += 1
        if spent >= args.cpu_seconds:
            break
    pops = r.counters(reset=True)[0]
    if kind == "reference":
        evals_counted = 8 * (pops - ncalls + r.trans_empty)       # every pop expands 8 cubes except the final threshold break
    else:
        evals_counted = evals
    return {"value": evals_counted / spent, "unit": "cube-evals/s", "cores": 1, "kind": kind,
            "sample": "first %d InnerBnB calls of this registration replayed by the reference's GoICP::InnerBnB (%d cube-evals, %.1f s, "
                      "%d result mismatches > 1e-5)" % (ncalls, evals_counted, spent, mismatches),
            "host_cpus": os.cpu_count()}


def cpu_cavity_baseline(pairs, cfg4, cav):
    """The reference's whole Register() (BuildDT included) on the same cavity-like pairs, one host core."""
    from oracle import refbind
    if not os.path.exists(refbind.REF_SO):
        return {}
    t0 = time.perf_counter(); n = 0; mism = 0
    for Pc in pairs:
        r = refbind.Ref(); r.set_config(cfg4); r.set_clouds(Pc["t"], Pc["s"], Pc["tc"], Pc["sc"], Pc["tf"], Pc["sf"])
        r.build_dt(); r.register(); n += 1
        if time.perf_counter() - t0 > 40.0:
            break
    dt = time.perf_counter() - t0
    return {"cavity_pair_registrations_per_sec": n / dt, "cavity_sample": "%d of the %d cavity-like pairs of rank 0, reference BuildDT()+Register(), %.1f s" % (n, len(pairs), dt)}


if __name__ == "__main__":
    main()
