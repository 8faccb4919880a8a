"""Trimmed cavity-sized registrations with ICP3D::Run in one launch (default) and with one launch per iteration (GOICP_DEVICE_ICP=0):
same results, wall time and ICP share. Run on the GPU box: python tools/icp_trim_perf.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import goicp_amd
from helpers import load_clouds, load_register, synth_pair

G = load_register()
cases = [("pair1_trim", load_clouds("pair1"), goicp_amd.cfg_vector(**G["pair1_trim"]["cfg"]))]
cases.append(("pair1_default", load_clouds("pair1"), goicp_amd.cfg_vector(**G["pair1_default"]["cfg"])))
cases.append(("pair1_fpfh", load_clouds("pair1"), goicp_amd.cfg_vector(**G["pair1_fpfh"]["cfg"])))
for n, trim, seed in ((238, 0.2, 1), (400, 0.1, 2), (800, 0.3, 3), (1024, 0.1, 4), (150, 0.0, 5), (238, 0.0, 6), (400, 0.0, 7), (800, 0.0, 8), (1024, 0.0, 9)):
    P = synth_pair(n, seed=99100 + seed, noise=0.01, outlier_frac=trim * 0.8 if trim > 0 else 0.0, theta_range=(0.3, np.pi))
    cases.append(("synth n=%d trim %.1f" % (n, trim), P, goicp_amd.cfg_vector(trimFraction=trim, regularization=0.0, regularizationFPFH=0.0, cfpfh=0, ponderation=0, distTransSize=100)))
for name, P, v in cases:
    res = {}
    for mode in ("1", "0", "1", "0"):
        os.environ["GOICP_DEVICE_ICP"] = mode
        g = goicp_amd.GoICP(v); g.set_clouds(P["t"], P["s"], P.get("tc"), P.get("sc"), P.get("tf"), P.get("sf")); g.build_dt()
        t0 = time.perf_counter(); o = g.register(); dt = time.perf_counter() - t0
        g.close()
        res[mode] = (dt, o)
    a, b = res["1"], res["0"]
    same = a[1]["error"] == b[1]["error"] and np.array_equal(a[1]["R"], b[1]["R"]) and np.array_equal(a[1]["t"], b[1]["t"]) and a[1]["stats"] == b[1]["stats"]
    st = a[1]["stats"]
    print("%-22s one launch %.4f s | per-iteration launches %.4f s | identical %s | icp calls %d iters %d trans_pops %d error %g" % (
        name, a[0], b[0], same, st["icp_calls"], st["icp_iters"], st["trans_pops"], a[1]["error"]), flush=True)
