#!/usr/bin/env python3
"""bench.py -- translation-BnB cube-evaluations/s (+ end-to-end pair registrations/s) on MI355X.

    python bench.py --gpus N --steps K --warmup W
        N > 1 without RANK in the environment: bench.py starts the N ranks itself (torch.distributed.run, one per GPU, before
        anything in this process touches a GPU) and relays rank 0's line; it fails loudly if fewer than N GPUs are visible.
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Workload (BASELINE.json configs[2], the one the metric is quoted on): seeded synthetic 5 000-point shell cloud vs a
rotated + noised copy with 10 % outliers, trimFraction 0.1, 300^3 distance transform, MSEThresh 1e-3, penalties off.
One "step" = one launch group of the cube sweep: 16 rotations x 2 bound variants (rho = 0 and rho = maxRotDis[3]) x all
4 680 cubes of octree levels 1-4 of [-0.5,0.5]^3 = 149 760 cube evaluations, every input resident in HBM beforehand.
Per rank the work is fixed (weak scaling): pairs are independent, there is no data-path collective.
The same run also (a) registers the pair end to end (Register(), BuildDT()+Register(), and several such pairs with their DT
builds overlapped on host threads), (b) runs BASELINE configs[3] -- the 383-row cavity batch under the c-FPFH weighted error --
through the batch driver (rows drawn longest-first from one shared counter, one all_gather of the result records incl. RMSD), and
(c) on rank 0 at N=1, times the CPU reference (oracle/_ref, the unmodified reference compiled in place) on a bounded replay of
the same searches, on one core and on all host cores.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

N_POINTS = 5000
DT_SIZE = 300
TRIM = 0.1
N_ROT = 16
PEAK_HBM_GBPS = 8000.0          # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)
TILE_CUBES = 8                  # cubes that share one register-resident copy of the cloud (TC in csrc/engine_dev.hpp)
PMC_SUMMARY = os.path.join("profiles", "r03_summary.json")     # written by tools/profile_round.sh from rocprofv3 --pmc passes of this command
PMC_SUMMARY_STRESS = os.path.join("profiles", "r03_c5_summary.json")   # the same for the configs[4]-shaped launch (python bench.py --points 50000 --dt 500 --trim 0)


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


def issued_bytes_per_launch(nd, evals_per_launch, fused):
    """Bytes one launch actually requests: per tile of 8 cubes the cloud once (16 B per point + 4 B radius where a radius row is read) and one
    4-B DT value per point and cube -- once per PAIR of cube-evals when (rho = 0, rho) group pairs are evaluated from one pass of gathers."""
    if not fused:
        return algorithmic_bytes_per_cube_eval(nd) * evals_per_launch
    pair_evals = evals_per_launch / 2
    return (nd * 4 + 16 + nd * 20 / TILE_CUBES) * pair_evals


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=600, help="timed launches (2.0 ms each at the default shape: ~1.2 s)")
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--points", type=int, default=N_POINTS)
    ap.add_argument("--dt", type=int, default=DT_SIZE)
    ap.add_argument("--trim", type=float, default=TRIM, help="trimFraction (configs[4], the 50 000-point stress shape, is untrimmed)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--no-register", action="store_true", help="skip the end-to-end registration legs")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--cpu-workers", type=int, default=0, help="all-cores CPU baseline: worker processes (0 = min(host cpus, 32))")
    ap.add_argument("--cavity-rows", type=int, default=383, help="rows of the configs[3] batch (whole job, sharded over the ranks)")
    ap.add_argument("--cavity-threads", type=int, default=16, help="host threads (= concurrent registrations, one engine + stream each) per rank")
    ap.add_argument("--overlap-pairs", type=int, default=16, help="configs[2]-shaped pairs registered with their DT builds overlapped on host threads (a box gives one GPU 16 host cores)")
    ap.add_argument("--no-stress", action="store_true", help="skip the configs[4]-shaped sweep leg (50 000 points, 500^3 grid, untrimmed)")
    ap.add_argument("--stress-steps", type=int, default=10)
    return ap.parse_args()


def spawn_ranks(args):
    """--gpus N without a launcher: start N ranks (one per GPU) as a child torch.distributed.run and relay its output."""
    import torch                                    # importing torch and counting devices does not initialise a GPU
    ndev = torch.cuda.device_count()
    if ndev < args.gpus:
        sys.stderr.write("bench.py: --gpus %d requested but only %d GPU(s) are visible on this node; refusing to report a smaller run as N=%d\n" % (args.gpus, ndev, args.gpus))
        sys.exit(2)
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ); env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.exit(subprocess.call(cmd, env=env))


def main():
    args = parse_args()
    if args.gpus > 1 and "RANK" not in os.environ:
        spawn_ranks(args)
    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1")); local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.stderr.write("bench.py: --gpus %d does not match WORLD_SIZE %d\n" % (args.gpus, world))
        sys.exit(2)

    import torch
    import torch.distributed as dist
    # under torch.distributed.run (RANK/MASTER_* set) the collectives are exercised even with one rank
    use_dist = world > 1 or ("RANK" in os.environ and "MASTER_PORT" in os.environ)
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # RCCL on ROCm. Rehearsal of the N > 1 path on a ONE-GPU box (never the measured configuration): GOICP_BENCH_BACKEND=gloo with
        # GOICP_BENCH_ONE_GPU=1 puts every rank on device 0 (RCCL refuses two ranks on one device); the JSON line then says so in `rehearsal`.
        if os.environ.get("GOICP_BENCH_ONE_GPU") == "1":
            local = 0
        torch.cuda.set_device(local)
        backend = os.environ.get("GOICP_BENCH_BACKEND", "nccl")
        # device_id: the rank's communicator is bound to ITS GPU at once (no guessing the device from the global rank at the first barrier)
        dist.init_process_group(backend, **({"device_id": torch.device("cuda", local)} if backend == "nccl" else {}))
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
    kname = eng.eval_kernel_name()
    fused = eng.last_eval_fused()
    # `frac` counts SURVEY 8(d)'s bytes for every cube-eval; a fused (rho = 0, rho) pair is served by ONE pass of gathers, so the bytes the
    # kernel actually asks the memory system for are fewer: frac_issued prices only the gathers made (memory-system utilisation)
    issued_per_launch = issued_bytes_per_launch(nd, evals_per_step, fused)
    # The launch group holds every rotation twice (rho = 0 and rho = maxRotDis[3], as OuterBnB's ub / lb calls do), and the engine
    # evaluates such pairs from ONE pass of gathers (k_eval_cubes_geo<.., true>). For transparency the same launch group is also timed
    # with the fusion switched off (a second engine on the same grid): reported beside the line, never as `value`.
    unfused = None
    if fused and rank == 0:
        os.environ["GOICP_FUSE"] = "0"
        try:
            g.initialize(); eng_u = g.engine()
            for s_, R in enumerate(rots):
                eng_u.set_rotation(s_, R)
            eng_u.eval_cubes_multi_dev(g_slots, g_levels, cubes.data_ptr(), ncube, ub.data_ptr(), lb.data_ptr()); eng_u.synchronize()
            t0 = time.perf_counter()
            for _ in range(20):
                eng_u.eval_cubes_multi_dev(g_slots, g_levels, cubes.data_ptr(), ncube, ub.data_ptr(), lb.data_ptr())
            eng_u.synchronize()
            unfused = dict(kernel=eng_u.eval_kernel_name(), avg_launch_ms=(time.perf_counter() - t0) / 20 * 1e3)
            unfused["frac"] = bytes_per_launch / (unfused["avg_launch_ms"] * 1e-3) / 1e9 / PEAK_HBM_GBPS
        finally:
            del os.environ["GOICP_FUSE"]

    # ---- BASELINE configs[4] shape: 50 000 points, 500^3 grid (beyond the Infinity Cache), untrimmed: the HBM-bound sweep. Right behind the metric
    #      leg (behind the registration and batch legs the same launches once measured 29.8 ms instead of 19.1-19.6) -------
    stress = None
    default_shape_run = nd == N_POINTS and args.dt == DT_SIZE and abs(args.trim - TRIM) < 1e-9
    if rank == 0 and default_shape_run and not args.no_stress:
        stress = stress_leg(goicp_amd, synth_pair, local, torch, dev, args.stress_steps)

    # ---- BASELINE configs[3]: the cavity batch under the c-FPFH weighted error, through the batch driver (ahead of the registration legs; its rate follows
    #      the host cores the box has free: 241-557 pairs/s over this round's runs of the same build, tools/order_effect.py) ----
    cav = None
    if not args.no_register and args.cavity_rows > 0:
        cav = cavity_batch(goicp_amd, args, rank, world, local, use_dist, dist, dev, torch)
        # the batch's pooled streams go back to the runtime here: while the 16 CU-masked search streams of its engines stay alive (idle, in the
        # engine pool), every later launch of the process runs slower -- the registration leg below 60-72 ms instead of 43 ms, its search
        # kernels 20-30 ms instead of 17.8 (tools: BENCH_KEEP_POOL=1 python bench.py --no-cpu --no-stress --overlap-pairs 1)
        if not os.environ.get("BENCH_KEEP_POOL"):
            goicp_amd.lib().goicp_engine_release_cached_memory()

    # ---- end-to-end registration of the same pair ------------------------------------------------------
    reg = None
    if not args.no_register:
        g.clear()
        g.register()                                      # untimed: the first registration of a process loads the search / ICP code objects (10-20 ms)
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
        # the same pair with the grid built on the device (opt-in exact transform, csrc/dt_gpu.hip): BuildDT() + Register()
        g2 = goicp_amd.GoICP(cfg, device_id=local, dt_mode=1); g2.set_clouds(P["t"], P["s"])
        g2.build_dt()                                     # (first build of the process: code-object load)
        if use_dist:
            dist.barrier()
        t0 = time.perf_counter(); g2.build_dt(); t1 = time.perf_counter(); out2 = g2.register(); t2 = time.perf_counter()
        g2.close()
        td = torch.tensor([t2 - t0], dtype=torch.float64, device=dev)
        if use_dist:
            dist.all_reduce(td, op=dist.ReduceOp.MAX)
        reg.update(device_dt=dict(dt_build_seconds=t1 - t0, register_seconds=t2 - t1, pair_registrations_per_sec_incl_dt_build=world / float(td.item()),
                                  error=float(out2["error"]), dR_vs_planted=float(np.abs(out2["R"] - P["Rtrue"]).max()),
                                  note="dt_mode=1: exact Euclidean transform built on the GPU (not bit-identical to the reference's propagated grid; opt-in)"))
        if args.overlap_pairs > 1:
            from goicp_amd import driver
            # host-built grids: the builds are CPU work that fills the CPU budget whatever the thread count, one pair per budgeted CPU
            reg.update(overlapped_registrations(goicp_amd, synth_pair, cfg, nd, max(2, min(args.overlap_pairs, driver.host_cpu_budget() // max(1, world))), local, world, rank, use_dist, dist, dev, torch))
            nov = driver.worker_threads(args.overlap_pairs, world)          # spinning waits: two CPUs of the budget stay free (14 of a 16-CPU quota)
            # the same with the device-built grid: 4 x as many pairs, the same number of registrations in flight
            reg["device_dt"].update(overlapped_registrations(goicp_amd, synth_pair, cfg, nd, 4 * nov, local, world, rank, use_dist, dist, dev, torch,
                                                             dt_mode=1, threads=nov, tag="in_flight"))

    # ---- CPU baseline: the unmodified reference (oracle/_ref) replaying the same InnerBnB calls, bounded -------
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu:
        cpu = cpu_baseline(g, P, cfg, args, reg)
        if cpu is not None and cav is not None:
            cpu.update(cpu_cavity_baseline(cav.pop("_base"), cav.pop("_tsv"), cav.pop("_cfg")))
    if cav is not None:
        for k in ("_base", "_tsv", "_cfg"):
            cav.pop(k, None)

    # the DT / engine can go now
    g.close()

    if rank == 0:
        traffic = None; traffic_source = None
        default_shape = nd == N_POINTS and args.dt == DT_SIZE and abs(args.trim - TRIM) < 1e-9
        pj = os.path.join(ROOT, PMC_SUMMARY)
        if default_shape and os.path.exists(pj):     # the committed PMC summary belongs to the default shape only
            try:
                S = json.load(open(pj))
                traffic = S.get("hbm_bytes_per_launch_by_kernel", {}).get(kname)
                if traffic is not None:
                    traffic_source = "%s (separate rocprofv3 --pmc passes of this command on this build; not measured by this run)" % PMC_SUMMARY
            except Exception:
                traffic = None
        label = "configs[2]" if default_shape else ("configs[4] shape" if nd >= 50000 else "non-default shape")
        line = {
            "metric": "translation_bnb_cube_evals_per_sec", "value": value, "unit": "cube-evals/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%s: synthetic %d-pt shell cloud vs rotated+noised copy (10%% outliers), trimFraction=%.1f, %d^3 DT; "
                                   "step = 16 rotations x 2 rho-variants x 4680 cubes (octree levels 1-4)" % (label, nd, args.trim, args.dt),
                       "points": nd, "dt_size": args.dt, "trimFraction": args.trim, "cube_evals_per_step": evals_per_step, "parallelism": "pairs sharded, 1 rank per GPU"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": PEAK_HBM_GBPS, "unit": "GB/s", "frac": achieved / PEAK_HBM_GBPS,
                         "frac_issued": issued_per_launch / (kern_ms * 1e-3) / 1e9 / PEAK_HBM_GBPS, "issued_bytes_per_launch": issued_per_launch, "traffic": traffic,
                         "traffic_source": traffic_source, "kernel": kname, "avg_launch_ms": kern_ms, "algorithmic_bytes_per_launch": bytes_per_launch,
                         "point_lookups_per_sec": value / world * nd,
                         "fused_group_pairs": fused, "unfused": unfused},
            "cpu_baseline": cpu,
            "registration": reg,
            "cavity_batch": cav,
            "stress": stress,
        }
        if os.environ.get("GOICP_BENCH_ONE_GPU") == "1" or os.environ.get("GOICP_BENCH_BACKEND", "nccl") != "nccl":
            line["rehearsal"] = "NOT a measurement: %d ranks, backend %s, GOICP_BENCH_ONE_GPU=%s (all ranks share device 0)" % (
                world, os.environ.get("GOICP_BENCH_BACKEND", "nccl"), os.environ.get("GOICP_BENCH_ONE_GPU", "0"))
        print(json.dumps(line))
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


def stress_leg(goicp_amd, synth_pair, local, torch, dev, steps):
    """BASELINE configs[4] shape through the same launch group (16 rotations x 2 rho-variants x 4 680 cubes = 149 760 cube-evals per launch):
    50 000 points streamed from HBM per tile, 500^3 grid (250 MB coded + 500 MB float: beyond the 256 MB Infinity Cache), untrimmed. The grid is
    built on the device (dt_mode=1, milliseconds) -- the host builder would take seconds and the sweep does not care which builder made the grid."""
    nd, S = 50000, 500
    P = synth_pair(nd, seed=20260101, theta_range=(2.2, 3.1))
    cfg = goicp_amd.cfg_vector(MSEThresh=1e-5, trimFraction=0.0, regularization=0.0, regularizationFPFH=0.0, cfpfh=0, ponderation=0, norm=2, distTransSize=S)
    g = goicp_amd.GoICP(cfg, device_id=local, dt_mode=1); g.set_clouds(P["t"], P["s"]); g.build_dt(); g.initialize()
    eng = g.engine()
    rng = np.random.default_rng(20260102)
    for s_ in range(N_ROT):
        aa = rng.normal(size=3); aa = aa / np.linalg.norm(aa) * rng.uniform(0, np.pi)
        th = np.linalg.norm(aa); k = aa / th
        K = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
        eng.set_rotation(s_, (np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * K @ K).astype(np.float32))
    cubes_h = level_cubes(); ncube = len(cubes_h)
    cubes = torch.from_numpy(cubes_h).to(dev)
    ub = torch.empty((N_ROT * 2, ncube), dtype=torch.float32, device=dev); lb = torch.empty_like(ub)
    g_slots = np.repeat(np.arange(N_ROT, dtype=np.int32), 2); g_levels = np.tile(np.array([-1, 3], np.int32), N_ROT)
    stream = torch.cuda.ExternalStream(eng.stream(), device=dev)
    for _ in range(2):
        eng.eval_cubes_multi_dev(g_slots, g_levels, cubes.data_ptr(), ncube, ub.data_ptr(), lb.data_ptr())
    eng.synchronize(); torch.cuda.synchronize()
    ev0 = torch.cuda.Event(enable_timing=True); ev1 = torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(stream):
        ev0.record(stream)
    t0 = time.perf_counter()
    for _ in range(steps):
        eng.eval_cubes_multi_dev(g_slots, g_levels, cubes.data_ptr(), ncube, ub.data_ptr(), lb.data_ptr())
    with torch.cuda.stream(stream):
        ev1.record(stream)
    eng.synchronize(); torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    kern_ms = ev0.elapsed_time(ev1) / steps
    evals = N_ROT * 2 * ncube
    fused = eng.last_eval_fused(); kname = eng.eval_kernel_name()
    bytes_per_launch = algorithmic_bytes_per_cube_eval(nd) * evals
    issued = issued_bytes_per_launch(nd, evals, fused)
    assert np.isfinite(float(ub.double().sum().item() + lb.double().sum().item()))
    g.close()
    traffic = None; src = None
    pj = os.path.join(ROOT, PMC_SUMMARY_STRESS)
    if os.path.exists(pj):
        try:
            traffic = json.load(open(pj)).get("hbm_bytes_per_launch_by_kernel", {}).get(kname)
            if traffic is not None:
                src = "%s (separate rocprofv3 --pmc passes of `python bench.py --points 50000 --dt 500 --trim 0` on this build; not measured by this run)" % PMC_SUMMARY_STRESS
        except Exception:
            traffic = None
    return {"workload": "configs[4] shape: synthetic 50000-pt cloud (10% outliers), untrimmed, 500^3 DT built on the device; step = 16 rotations x 2 rho-variants x 4680 cubes",
            "value": evals * steps / wall, "unit": "cube-evals/s", "steps": steps, "ms_per_step": wall / steps * 1e3, "cube_evals_per_step": evals,
            "point_lookups_per_sec": evals * steps / wall * nd,
            "roofline": {"bound": "hbm", "achieved": bytes_per_launch / (kern_ms * 1e-3) / 1e9, "peak": PEAK_HBM_GBPS, "unit": "GB/s",
                         "frac": bytes_per_launch / (kern_ms * 1e-3) / 1e9 / PEAK_HBM_GBPS, "frac_issued": issued / (kern_ms * 1e-3) / 1e9 / PEAK_HBM_GBPS,
                         "traffic": traffic, "traffic_source": src, "kernel": kname, "avg_launch_ms": kern_ms, "algorithmic_bytes_per_launch": bytes_per_launch,
                         "issued_bytes_per_launch": issued, "fused_group_pairs": fused}}


def overlapped_registrations(goicp_amd, synth_pair, cfg, nd, n_pairs, local, world, rank, use_dist, dist, dev, torch, dt_mode=0, threads=None, tag="overlapped"):
    """pairs/s including the DT build when several registrations are in flight on one GPU: n_pairs configs[2]-shaped pairs per rank on host
    threads, each BuildDT() -> Register(). dt_mode 0: the host builder's 0.9 s overlap the other pairs' GPU work;
    dt_mode 1: the grid is built on the device in milliseconds and the registrations' own launch gaps (search rounds, ICP iterations) overlap."""
    from concurrent.futures import ThreadPoolExecutor
    pairs = [synth_pair(nd, seed=20260201 + 17 * (rank * n_pairs + k), theta_range=(2.2, 3.1)) for k in range(n_pairs)]

    def one(Pk):
        gk = goicp_amd.GoICP(cfg, device_id=local, dt_mode=dt_mode); gk.set_clouds(Pk["t"], Pk["s"])
        t0 = time.perf_counter(); gk.build_dt(); t1 = time.perf_counter(); o = gk.register(); t2 = time.perf_counter()
        gk.close()
        return t1 - t0, t2 - t1, float(np.abs(o["R"] - Pk["Rtrue"]).max())
    if use_dist:
        dist.barrier()
    t0 = time.perf_counter()
    with ThreadPoolExecutor(max_workers=threads or n_pairs) as ex:
        outs = list(ex.map(one, pairs))
    t = time.perf_counter() - t0
    if use_dist:
        tt = torch.tensor([t], dtype=torch.float64, device=dev); dist.all_reduce(tt, op=dist.ReduceOp.MAX); t = float(tt.item())
    return {tag + "_pairs": n_pairs * world, tag + "_seconds": t, "pair_registrations_per_sec_incl_dt_build_" + tag: n_pairs * world / t,
            tag + "_mean_dt_build_seconds": float(np.mean([o[0] for o in outs])), tag + "_mean_register_seconds": float(np.mean([o[1] for o in outs])),
            tag + "_max_dR_vs_planted": float(max(o[2] for o in outs))}


def cavity_batch(goicp_amd, args, rank, world, local, use_dist, dist, dev, torch):
    from goicp_amd import driver, c4_batch
    base = os.path.join(tempfile.gettempdir(), "goicp_c4_%d_%s" % (os.getuid(), os.environ.get("MASTER_PORT", str(os.getpid()))))
    if rank == 0:                               # ranks of one node share the filesystem: one writer
        tsv, cfgp, truth = c4_batch.materialise(base, os.path.join(ROOT, "tests", "golden", "data"), rows=args.cavity_rows)
    if use_dist:
        dist.barrier()
    tsv = os.path.join(base, "c4_batch.tsv"); cfgp = os.path.join(base, "config_c4.txt")
    outd = os.path.join(base, "output_r%d" % rank)
    # untimed warm-up: the first rows through the same driver with the same number of registrations in flight (code objects of the search and ICP
    # kernels, pooled streams / pinned buffers / device blocks of 16 engines), as the metric leg has its warm-up launches
    warm = min(32, max(1, args.cavity_rows // world))
    # host threads (= registrations in flight) per rank: the waits spin, so never more threads than this rank's share of the host cores
    cores = driver.host_cpu_budget()                      # affinity set or cgroup CPU quota, whichever is smaller (16 on a one-GPU box)
    threads = driver.worker_threads(args.cavity_threads, world, cores)
    driver.run_batch(base, tsv, cfgp, os.path.join(base, "warm_r%d" % rank), driver.gpu_register_fn(base, device_id=local), rank=rank, world=world,
                     threads=threads, limit=min(args.cavity_rows, warm * world), schedule="rr", rmsd=False)      # (disjoint rows per rank: the pipeline writes per-row files)
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    goicp_amd.batch_search_totals(reset=True)
    t0 = time.perf_counter()
    recs, n_rows = driver.run_batch(base, tsv, cfgp, outd, driver.gpu_register_fn(base, device_id=local), rank=rank, world=world, threads=threads,
                                    dist=dist if use_dist else None)
    t_local = time.perf_counter() - t0
    allr = driver.gather_records(recs, n_rows, dist=dist if use_dist else None, device=dev if use_dist and dist.get_backend() == "nccl" else None)   # THE exchange: one all_gather (RCCL; host tensors under the gloo rehearsal)
    t_cav = time.perf_counter() - t0
    t_ranks = [t_local]
    if use_dist:
        tt = torch.tensor([t_cav], dtype=torch.float64, device=dev); dist.all_reduce(tt, op=dist.ReduceOp.MAX); t_cav = float(tt.item())
        t_ranks = [None] * world; dist.all_gather_object(t_ranks, t_local)       # (after the timed region: how long each rank registered before the gather)
    tot = goicp_amd.batch_search_totals()             # this rank's registrations of the timed batch
    search = dict(tot, pops_launched_per_consumed=tot["pops_launched"] / max(1, tot["pops_consumed"]),
                  jobs_launched_per_consumed=tot["jobs_launched"] / max(1, tot["inner_calls_consumed"]), scope="rank 0's registrations of the timed batch")
    rm = allr[:, 19]
    per_rank = [int((allr[:, 21] == r).sum()) for r in range(world)]
    weak = None
    if world > 1:
        # weak-scaling leg: EVERY rank registers the whole batch on files of its own (the strong leg above is bounded by the longest pair once
        # the rows are spread over 8 GPUs); no exchange, max over ranks
        wbase = base + "_weak_r%d" % rank
        wtsv, wcfg, _ = c4_batch.materialise(wbase, os.path.join(ROOT, "tests", "golden", "data"), rows=args.cavity_rows)
        dist.barrier(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        wrecs, _ = driver.run_batch(wbase, wtsv, wcfg, os.path.join(wbase, "output"), driver.gpu_register_fn(wbase, device_id=local), rank=0, world=1, threads=threads, rmsd=False)
        t_weak_local = time.perf_counter() - t0
        tw = torch.tensor([t_weak_local], dtype=torch.float64, device=dev); dist.all_reduce(tw, op=dist.ReduceOp.MAX)
        nw = torch.tensor([float(len(wrecs))], dtype=torch.float64, device=dev); dist.all_reduce(nw, op=dist.ReduceOp.SUM)
        tw_ranks = [None] * world; dist.all_gather_object(tw_ranks, t_weak_local)
        weak = dict(pairs=int(nw.item()), seconds=float(tw.item()), cavity_pair_registrations_per_sec=float(nw.item()) / float(tw.item()),
                    scaling="weak (every rank registers all %d rows)" % args.cavity_rows, seconds_per_rank=[float(x) for x in tw_ranks])
    return dict(pairs=int(allr.shape[0]), seconds=t_cav, cavity_pair_registrations_per_sec=allr.shape[0] / t_cav, cube_evals=int(np.nansum(allr[:, 18])),
                concurrent_registrations_per_gpu=threads, host_cores=cores, warmup_pairs=warm, scaling="strong (the whole batch is divided over the ranks)",
                schedule="longest-first, rows drawn from one shared counter" if world > 1 else "longest-first", pairs_per_rank=per_rank,
                failed=int((allr[:, 20] != 0).sum()), rmsd_gathered=int((~np.isnan(rm)).sum()), rmsd_median=float(np.nanmedian(rm)), rmsd_below_2A=int((rm < 2.0).sum()),
                rank0_seconds_before_gather=t_local, seconds_before_gather_per_rank=[float(x) for x in t_ranks], search=search, weak=weak,
                config="BASELINE configs[3]: rows 1-2 shipped cavity pairs, rows 3-%d seeded synthetic cavity-like pairs (N in [150,400]); shipped config.txt + cfpfh=1, regularizationFPFH=5e-6; file pipeline incl. protein transform + RMSD" % args.cavity_rows,
                _base=base, _tsv=tsv, _cfg=cfgp)


def cpu_baseline(g, P, cfg, args, reg):
    """Times the reference's own InnerBnB (oracle/_ref) on the host cores of this box, replaying the first InnerBnB calls of the
    GPU registration (same rotations, levels and incumbents => identical searches) until ~cpu_seconds of CPU time are spent:
    once on one core (the reference is single-threaded) and once as W independent processes over disjoint shares (all cores)."""
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
    out = {"value": evals_counted / spent, "unit": "cube-evals/s", "cores": 1, "kind": kind,
           "sample": "first %d InnerBnB calls of this registration replayed by the reference's GoICP::InnerBnB (%d cube-evals, %.1f s, "
                     "%d result mismatches > 1e-5)" % (ncalls, evals_counted, spent, mismatches),
           "host_cpus": os.cpu_count()}
    # all host cores: W worker processes (each its own copy of the reference and of the grid) over disjoint shares of the calls
    # (as many as the CPU budget carries: a box shows 256 CPUs but its cgroup has a quota of 16 -- 32 workers would only share those)
    from goicp_amd import driver
    budget = driver.host_cpu_budget()
    W = args.cpu_workers if args.cpu_workers > 0 else min(budget, 32)
    out["cpu_budget"] = budget
    if W > 1:
        try:
            path = os.path.join(tempfile.gettempdir(), "goicp_cpu_jobs_%d.npz" % os.getpid())
            np.savez(path, cfg=cfg, model=P["t"], data=P["s"], S=g.S, geom=g.geom, dist=dist_grid, R=np.stack([j["R"] for j in jobs]),
                     level=np.array([j["level"] for j in jobs], np.int32), E=np.array([j["E"] for j in jobs], np.float32))
            res = subprocess.run([sys.executable, "-m", "oracle.cpu_allcores", path, str(W), str(args.cpu_seconds)], cwd=ROOT, capture_output=True, text=True, timeout=600)
            os.remove(path)
            A = json.loads(res.stdout.strip().splitlines()[-1])
            out["all_cores"] = {"value": A["value"], "unit": "cube-evals/s", "cores": A["cores"], "kind": A["kind"],
                                "sample": "%d worker processes, each replaying its share of the same InnerBnB calls for %.0f s (%d calls, %d cube-evals)" % (A["cores"], args.cpu_seconds, A["calls"], A["cube_evals"])}
        except Exception as e:                  # the one-core figure stands on its own
            out["all_cores"] = {"error": repr(e)[:200]}
    return out


def cpu_cavity_baseline(base, tsv, cfgp):
    """The reference's whole BuildDT()+Register() on the first rows of the same cavity batch (same files), one host core."""
    from oracle import refbind
    from goicp_amd import driver
    if not os.path.exists(refbind.REF_SO):
        return {}
    cfg = refbind.read_config(cfgp); v = np.array([cfg[k] for k in refbind.CFG_KEYS], np.float64)
    jobs, _ = driver.list_jobs(base, tsv)
    t0 = time.perf_counter(); n = 0
    for k, tf, sf, nd, cost, src, tgt in jobs[2:]:              # (row 2 alone is 35 s on the reference: start behind the shipped pairs)
        sx, sc = refbind.read_mol2(os.path.join(base, sf)); tx, tc = refbind.read_mol2(os.path.join(base, tf))
        sn, tn = refbind.normalise_pair(sx, tx)[:2]
        r = refbind.Ref(); r.set_config(v)
        r.set_clouds(tn, sn, tc, sc, refbind.read_cfpfh(os.path.join(base, "cfpfh", tgt + "_cavity6.cfpfh"), len(tn)), refbind.read_cfpfh(os.path.join(base, "cfpfh", src + "_cavity6.cfpfh"), len(sn)))
        r.build_dt(); r.register(); n += 1
        if time.perf_counter() - t0 > 30.0:
            break
    dt = time.perf_counter() - t0
    return {"cavity_pair_registrations_per_sec": n / dt, "cavity_sample": "rows 3-%d of the same cavity batch, reference BuildDT()+Register() on one core, %.1f s" % (n + 2, dt)}


if __name__ == "__main__":
    main()
