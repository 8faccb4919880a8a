"""[needs the diagnosis build: make -C go-icp-protein-cavities_amd/csrc probe; run with GOICP_LIB=go-icp-protein-cavities_amd/libgoicp_hip_probe.so] Diagnosis (GPU box): the metric launch under the GOICP_DBG ablations (1 = no inlier selection, 2 = gathers folded onto a 4 KB
window, 3 = both; results are wrong by construction, timing only), fused and unfused.  python tools/geo_ablate.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import goicp_amd
from goicp_amd.synth import synth_pair
import bench
P = synth_pair(5000, seed=20260101, theta_range=(2.2, 3.1))
cfg = goicp_amd.cfg_vector(MSEThresh=0.001, trimFraction=0.1, regularization=0.0, regularizationFPFH=0.0, cfpfh=0, ponderation=0, norm=2, distTransSize=300)
g = goicp_amd.GoICP(cfg); g.set_clouds(P["t"], P["s"]); g.build_dt()
cubes_h = bench.level_cubes(); n = len(cubes_h)
cubes = torch.from_numpy(cubes_h).cuda(); ub = torch.empty((32, n), device="cuda"); lb = torch.empty_like(ub)
sl = np.repeat(np.arange(16, dtype=np.int32), 2); lv = np.tile(np.array([-1, 3], np.int32), 16)
for fuse in ("1", "0"):
    for dbg in ("0", "1", "2", "3"):
        os.environ["GOICP_FUSE"] = fuse; os.environ["GOICP_DBG"] = dbg
        g.initialize(); e = g.engine()
        rng = np.random.default_rng(20260102)
        for s in range(16):
            aa = rng.normal(size=3); aa = aa / np.linalg.norm(aa) * rng.uniform(0, np.pi); th = np.linalg.norm(aa); k = aa / th
            K = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
            e.set_rotation(s, (np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * K @ K).astype(np.float32))
        for _ in range(3):
            e.eval_cubes_multi_dev(sl, lv, cubes.data_ptr(), n, ub.data_ptr(), lb.data_ptr())
        e.synchronize(); t0 = time.perf_counter()
        for _ in range(30):
            e.eval_cubes_multi_dev(sl, lv, cubes.data_ptr(), n, ub.data_ptr(), lb.data_ptr())
        e.synchronize()
        print("fuse %s dbg %s: %.3f ms per launch (%s)" % (fuse, dbg, (time.perf_counter() - t0) / 30 * 1e3, e.eval_kernel_name()), flush=True)
