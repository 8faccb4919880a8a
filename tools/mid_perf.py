"""Mid-size geometric registrations (1025..5120 data points): where the 512-lane just-in-time search kernel applies.
GOICP_JIT_BNB=0 selects the 1024-lane register-resident kernel for comparison."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import goicp_amd
from goicp_amd.synth import synth_pair
geo = dict(regularization=0.0, ponderation=0, norm=2, cfpfh=0, regularizationFPFH=0.0)
b = np.load(os.path.join(ROOT, "tests", "golden", "data", "demo", "bunny.npz"))
rng = np.random.default_rng(3)
cases = [("bunny_model2000_vs_model", b["model"], b["model"][rng.choice(len(b["model"]), 2000, replace=False)] @ np.array([[0.36, 0.48, -0.8], [-0.8, 0.6, 0.0], [0.48, 0.64, 0.6]], np.float32).T, dict(MSEThresh=0.001, trimFraction=0.0, distTransSize=100, **geo))]
for n, th in ((2000, (2.4, 3.0)), (4000, (2.4, 3.0)), (5000, (2.2, 3.1))):
    P = synth_pair(n, seed=20260101 if n == 5000 else 4242 + n, theta_range=th, outlier_frac=0.1 if n == 5000 else 0.0)
    cases.append(("synth%d" % n, P["t"], P["s"], dict(MSEThresh=0.001, trimFraction=0.1 if n == 5000 else 0.0, distTransSize=100 if n < 5000 else 300, **geo)))
for name, t, s, cfg in cases:
    g = goicp_amd.GoICP(goicp_amd.cfg_vector(**cfg)); g.set_clouds(np.ascontiguousarray(t, np.float32), np.ascontiguousarray(s, np.float32)); g.build_dt()
    t0 = time.time(); o = g.register(); dt = time.time() - t0
    st = o["stats"]
    print("%-26s Nd=%5d reg %.3fs kern %.1fms err %.6g pops %d launches %d jobs %d" % (name, len(s), dt, o["kernel_ms"], o["error"], st["trans_pops"], st["inner_launches"], st["inner_jobs_launched"]), flush=True)
    g.close()
