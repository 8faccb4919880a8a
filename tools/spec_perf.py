"""Speculation depth sweep on cavity pair 2 (2.6 M cube evaluations) and a hard synthetic cavity pair."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import goicp_amd
from helpers import load_clouds, load_register, synth_pair
G = load_register(); P2 = load_clouds("pair2")
cases = [("pair2", P2, goicp_amd.cfg_vector(**G["pair2_default"]["cfg"]))]
rng = np.random.default_rng(9000 + 7); n = int(rng.integers(120, 420))
cases.append(("color8_hard", synth_pair(n, seed=31007, noise=0.01, outlier_frac=float(rng.choice([0.0, 0.05, 0.15])), theta_range=(0.3, np.pi), corr_features=True),
              goicp_amd.cfg_vector(cfpfh=3, regularizationFPFH=0.00005, regularization=0.0)))
for name, P, v in cases:
    for spec in (128,):
        g = goicp_amd.GoICP(v, speculate=spec)
        g.set_clouds(P["t"], P["s"], P["tc"], P["sc"], P["tf"], P["sf"]); g.build_dt()
        t0 = time.perf_counter(); o = g.register(); dt = time.perf_counter() - t0
        st = o["stats"]
        print("%s over=%s spec=%d reg %.3fs kern %.1fms err %g launches %d jobs %d cube_evals %d" % (name, os.environ.get("GOICP_SPEC_OVER", "1"), spec, dt, o["kernel_ms"], o["error"], st["inner_launches"], st["inner_jobs_launched"], st["cube_evals"]), flush=True)
        g.close()
