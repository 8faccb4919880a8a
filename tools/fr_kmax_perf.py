"""Round-based search: nodes expanded per search and round (GOICP_FR_KMAX) against Register() time (GPU box).
python tools/fr_kmax_perf.py      -> configs[2] pair (16 searches side by side) and a budgeted configs[4] search (one deep search)"""
import os, sys, time, subprocess, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import numpy as np
    import goicp_amd
    from goicp_amd.synth import synth_pair
    P = synth_pair(5000, seed=20260101, theta_range=(2.2, 3.1))
    cfg = goicp_amd.cfg_vector(MSEThresh=0.001, trimFraction=0.1, regularization=0.0, regularizationFPFH=0.0, cfpfh=0, ponderation=0, norm=2, distTransSize=300)
    g = goicp_amd.GoICP(cfg, dt_mode=1); g.set_clouds(P["t"], P["s"]); g.build_dt()
    for k in range(3):
        t0 = time.perf_counter(); out = g.register(); t = time.perf_counter() - t0
    print("  configs[2]: register %.4f s kernels %.1f ms rounds %d cube evals launched %d error %.9g pops %d" % (t, out["kernel_ms"], out["search"]["rounds"], out["search"]["cube_evals_launched"], out["error"], out["stats"]["trans_pops"]), flush=True)
    g.close()
    P = synth_pair(50000, seed=20260101, theta_range=(2.2, 3.1))
    cfg = goicp_amd.cfg_vector(MSEThresh=1e-5, trimFraction=0.0, regularization=0.0, regularizationFPFH=0.0, cfpfh=0, ponderation=0, norm=2, distTransSize=500)
    g = goicp_amd.GoICP(cfg, dt_mode=1); g.set_clouds(P["t"], P["s"]); g.build_dt(); g.set_budget(0.0, 60000)
    t0 = time.perf_counter(); out = g.register(); t = time.perf_counter() - t0
    print("  configs[4], 60 000-pop budget: register %.3f s kernels %.1f ms rounds %d pops %d cube evals launched %d consumed %d" % (t, out["kernel_ms"], out["search"]["rounds"], out["stats"]["trans_pops"],
          out["search"]["cube_evals_launched"], out["stats"]["cube_evals"]), flush=True)
    g.close()
else:
    for k in (16, 32, 63):
        env = dict(os.environ); env["GOICP_FR_KMAX"] = str(k)
        print("GOICP_FR_KMAX=%d" % k, flush=True)
        subprocess.call([sys.executable, os.path.abspath(__file__), "child"], env=env)
