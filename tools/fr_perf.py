"""Timing probe (GPU box): Register() of configs[2]-shaped pairs with the round-based search on and off, plus search statistics.
python tools/fr_perf.py [points] [dt] [trim]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import goicp_amd
from goicp_amd.synth import synth_pair

n = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
S = int(sys.argv[2]) if len(sys.argv) > 2 else 300
trim = float(sys.argv[3]) if len(sys.argv) > 3 else 0.1
mse = float(sys.argv[4]) if len(sys.argv) > 4 else 0.001
P = synth_pair(n, seed=20260101, theta_range=(2.2, 3.1))
cfg = goicp_amd.cfg_vector(MSEThresh=mse, trimFraction=trim, regularization=0.0, regularizationFPFH=0.0, cfpfh=0, ponderation=0, norm=2, distTransSize=S)
g = goicp_amd.GoICP(cfg); g.set_clouds(P["t"], P["s"])
t0 = time.time(); g.build_dt(); print("dt build %.3f s" % (time.time() - t0), flush=True)
for fr in ("0", "1", "1"):
    os.environ["GOICP_FR"] = fr
    g.initialize(); eng = g.engine()
    t0 = time.perf_counter(); out = g.register(); t = time.perf_counter() - t0
    print("GOICP_FR=%s register %.4f s kernel %.2f ms error %.6f cube_evals %d trans_pops %d rot_pops %d launches %d dR %.2e" % (
        fr, t, out["kernel_ms"], out["error"], out["stats"]["cube_evals"], out["stats"]["trans_pops"], out["stats"]["rot_pops"], out["stats"]["inner_launches"],
        np.abs(out["R"] - P["Rtrue"]).max()), flush=True)
