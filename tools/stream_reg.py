"""Register streamed-cloud candidates for the known-answer pair (tests/golden/make_synth_reference.py) on the GPU."""
import sys, time
sys.path.insert(0, '/root/repo')
import numpy as np
import goicp_amd
from goicp_amd.synth import synth_pair
for theta, mse in (((2.4, 3.0), 0.003), ((2.4, 3.0), 0.001)):
    P = synth_pair(9000, seed=4242, theta_range=theta, outlier_frac=0.0)
    v = goicp_amd.cfg_vector(MSEThresh=mse, trimFraction=0.0, regularization=0.0, regularizationFPFH=0.0, cfpfh=0, ponderation=0, norm=2, distTransSize=60)
    g = goicp_amd.GoICP(v); g.set_clouds(P["t"], P["s"]); g.build_dt()
    t0 = time.time(); out = g.register(); dt = time.time() - t0
    print(theta, mse, "error", out["error"], "seconds %.3f" % dt, "stats", out["stats"], "trace", out["trace"], flush=True)
    g.close()
