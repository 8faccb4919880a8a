"""Speculation depth / LDS-heap size sweep on cavity pair 2 (2.6 M cube evaluations). Run with GOICP_HEAP_LDS in the env."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import goicp_amd
from helpers import load_clouds, load_register
G = load_register(); P2 = load_clouds("pair2")
v = goicp_amd.cfg_vector(**G["pair2_default"]["cfg"])
for spec in (8, 16, 32, 64):
    for rep in range(2):
        g = goicp_amd.GoICP(v, speculate=spec)
        g.set_clouds(P2["t"], P2["s"], P2["tc"], P2["sc"], P2["tf"], P2["sf"]); g.build_dt()
        t0 = time.perf_counter(); o = g.register(); dt = time.perf_counter() - t0
        st = o["stats"]
        print("heap_lds=%s spec=%d reg %.3fs kern %.1fms err %g launches %d jobs %d" % (os.environ.get("GOICP_HEAP_LDS", "2048"), spec, dt, o["kernel_ms"], o["error"], st["inner_launches"], st["inner_jobs_launched"]), flush=True)
        g.close()
