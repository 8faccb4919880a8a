"""[needs the diagnosis build: make -C go-icp-protein-cavities_amd/csrc probe; run with GOICP_LIB=go-icp-protein-cavities_amd/libgoicp_hip_probe.so] Diagnosis (GPU box): where k_fr_select spends its cycles. Replays the InnerBnB jobs of a configs[2]-shaped registration as one
batch with GOICP_DBG=32 and prints the per-phase cycle sums. python tools/fr_phases.py [points] [dt] [trim]"""
import os, sys, time, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import goicp_amd
from goicp_amd.synth import synth_pair

n = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
S = int(sys.argv[2]) if len(sys.argv) > 2 else 300
trim = float(sys.argv[3]) if len(sys.argv) > 3 else 0.1
P = synth_pair(n, seed=20260101, theta_range=(2.2, 3.1))
cfg = goicp_amd.cfg_vector(MSEThresh=0.001, trimFraction=trim, regularization=0.0, regularizationFPFH=0.0, cfpfh=0, ponderation=0, norm=2, distTransSize=S)
g = goicp_amd.GoICP(cfg); g.set_clouds(P["t"], P["s"]); g.build_dt()
g.record_jobs(True); out = g.register(); jobs = g.job_log()
print("registration: %d consumed jobs, %d pops" % (len(jobs), sum(j["pops"] for j in jobs)), flush=True)
os.environ["GOICP_DBG"] = "32"
g.initialize(); e = g.engine()
J = [dict(R=j["R"], level=j["level"], E=float(j["E"]), sse=float(j["sse"]), init=(-0.5, -0.5, -0.5, 1.0)) for j in jobs]
for rep in range(2):
    t0 = time.perf_counter(); res = e.inner_bnb_batch(J); t = time.perf_counter() - t0
    dbg = np.zeros(16, np.uint64); e.L.goicp_engine_debug_times(e.h, dbg.ctypes.data_as(C.c_void_p))
    st = e.search_stats()
    print("batch of %d jobs: %.2f ms wall, pops %d (max %d), rounds %d, launched %d" % (len(J), t * 1e3, sum(r["pops"] for r in res), max(r["pops"] for r in res), st["rounds"], st["cube_evals_launched"]))
    print("  cycles: stage-in %d | loop ctl %d | pop %d | E+pushes %d | select %d | write-back %d | replayed pops %d (prefetch misses %d) | wall ticks(100MHz) %d" % tuple(int(v) for v in dbg[:9]))
    tot = float(dbg[:6].sum()); wall_us = float(dbg[8]) / 100.0
    print("  => shader clock ~ %.2f GHz; per replayed pop: pop %.0f cyc, E+pushes %.0f cyc" % (tot / wall_us / 1e3 if wall_us else 0, float(dbg[2]) / max(1, float(dbg[6])), float(dbg[3]) / max(1, float(dbg[6]))))
