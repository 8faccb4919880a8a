"""[needs the diagnosis build: make -C go-icp-protein-cavities_amd/csrc probe; run with GOICP_LIB=go-icp-protein-cavities_amd/libgoicp_hip_probe.so] Diagnosis (GPU box): cycles per iteration of the device-resident ICP loop (k_icp_loop) by phase, on the shipped cavity pairs."""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["GOICP_DBG"] = "32"
import numpy as np
import goicp_amd
from helpers import load_clouds
for name in ("pair1", "pair2"):
    P = load_clouds(name)
    g = goicp_amd.GoICP(goicp_amd.cfg_vector()); g.set_icp_one_launch(True); g.set_clouds(P["t"], P["s"], P["tc"], P["sc"], P["tf"], P["sf"]); g.build_dt(); g.initialize()
    e = g.engine()
    rng = np.random.default_rng(3)
    for k in range(6):
        aa = rng.normal(size=3); aa = aa / np.linalg.norm(aa) * rng.uniform(0, 1.0); th = np.linalg.norm(aa); kk = aa / th
        K = np.array([[0, -kk[2], kk[1]], [kk[2], 0, -kk[0]], [-kk[1], kk[0], 0]])
        g.icp(np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * K @ K, rng.uniform(-0.1, 0.1, 3))
    d = np.zeros(16, np.uint64); e.L.goicp_engine_debug_times(e.h, d.ctypes.data_as(C.c_void_p))
    it = max(1, int(d[12])); c = d[8:12].astype(float) / it
    print("%s (Nd %d, Nm %d): %d iterations in 6 calls; cycles per iteration: search %.0f | 7 running sums + stop test %.0f | covariance %.0f | 3x3 algebra (Jacobi SVD, pose) %.0f | total %.0f" % (
        name, g.Nd, g.Nm, it, c[0], c[1], c[2], c[3], c.sum()), flush=True)
    g.close()
