import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["GOICP_TIMING"] = "1"
import numpy as np
import goicp_amd
from helpers import synth_pair
cfg4 = goicp_amd.cfg_vector(cfpfh=1, regularizationFPFH=0.000005)
rngc = np.random.default_rng(20260103)
sizes = rngc.integers(150, 401, size=8)
for rep in range(2):
    for k in range(8):
        P = synth_pair(int(sizes[k]), seed=777000 + k, noise=0.01, outlier_frac=0.05, theta_range=(0.3, np.pi), corr_features=True)
        t0 = time.perf_counter()
        g = goicp_amd.GoICP(cfg4); g.set_clouds(P["t"], P["s"], P["tc"], P["sc"], P["tf"], P["sf"])
        t1 = time.perf_counter(); g.build_dt(); t2 = time.perf_counter()
        o = g.register(); t3 = time.perf_counter(); g.close(); t4 = time.perf_counter()
        st = o["stats"]
        print("pair %d N=%d total %.1f ms: setup %.1f dt %.1f register %.1f close %.1f | kern %.1f ms launches %d jobs %d pops %d icp %d/%d" % (
            k, sizes[k], (t4 - t0) * 1e3, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t4 - t3) * 1e3, o["kernel_ms"], st["inner_launches"], st["inner_jobs_launched"], st["trans_pops"], st["icp_calls"], st["icp_iters"]), flush=True)
