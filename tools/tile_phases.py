"""[needs the diagnosis build: make -C go-icp-protein-cavities_amd/csrc probe; run with GOICP_LIB=go-icp-protein-cavities_amd/libgoicp_hip_probe.so] Diagnosis (GPU box): per-phase cycles of the metric kernel's tile sweep (GOICP_DBG=64 stamps in sweep_tile).
Needs a library built with the stamps compiled in: make -C go-icp-protein-cavities_amd/csrc clean all STAMPS=1
python tools/tile_phases.py"""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["GOICP_DBG"] = "64"
import numpy as np, torch
import goicp_amd
from goicp_amd.synth import synth_pair
import bench
P = synth_pair(5000, seed=20260101, theta_range=(2.2, 3.1))
cfg = goicp_amd.cfg_vector(MSEThresh=0.001, trimFraction=0.1, regularization=0.0, regularizationFPFH=0.0, cfpfh=0, ponderation=0, norm=2, distTransSize=300)
g = goicp_amd.GoICP(cfg); g.set_clouds(P["t"], P["s"]); g.build_dt(); g.initialize(); e = g.engine()
rng = np.random.default_rng(20260102)
for s in range(16):
    aa = rng.normal(size=3); aa = aa / np.linalg.norm(aa) * rng.uniform(0, np.pi); th = np.linalg.norm(aa); k = aa / th
    K = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    e.set_rotation(s, (np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * K @ K).astype(np.float32))
cubes_h = bench.level_cubes(); n = len(cubes_h)
cubes = torch.from_numpy(cubes_h).cuda(); ub = torch.empty((32, n), device="cuda"); lb = torch.empty_like(ub)
sl = np.repeat(np.arange(16, dtype=np.int32), 2); lv = np.tile(np.array([-1, 3], np.int32), 16)
for _ in range(3):
    e.eval_cubes_multi_dev(sl, lv, cubes.data_ptr(), n, ub.data_ptr(), lb.data_ptr())
e.synchronize()
d = np.zeros(16, np.uint64); e.L.goicp_engine_debug_times(e.h, d.ctypes.data_as(C.c_void_p))
tiles = float(d[7]); c = d[:7].astype(float) / max(tiles, 1)
wall_us = float(d[8]) / max(tiles, 1) / 100.0
import time
t0 = time.perf_counter()
for _ in range(20):
    e.eval_cubes_multi_dev(sl, lv, cubes.data_ptr(), n, ub.data_ptr(), lb.data_ptr())
e.synchronize(); ms = (time.perf_counter() - t0) / 20 * 1e3
ntile = 585 * 16
print("kernel %s: %.3f ms per launch; a stamped tile takes %.1f us of wall time (%.0f cycles -> %.2f GHz); tiles in flight = tiles x tile time / launch time = %.0f (2 x 256 = 512 if two workgroups share every CU)"
      % (e.eval_kernel_name(), ms, wall_us, c.sum(), c.sum() / max(wall_us, 1e-9) / 1e3, ntile * wall_us / (ms * 1e3)))
print("stamped tiles %d; cycles per tile (fused pairs: both selections summed): gather+residuals %.0f | max %.0f | histogram %.0f | scan %.0f | collect+sums %.0f | finish %.0f | radius pass %.0f | total %.0f" % (tiles, c[0], c[1], c[2], c[3], c[4], c[5], c[6], c.sum()))

for variant in ():
    ms = C.c_float()
    rc = e.L.goicp_engine_debug_gather_probe(e.h, C.c_void_p(cubes.data_ptr()), n, C.c_void_p(ub.data_ptr()), variant, C.byref(ms))
    print("gather-only probe on the real access pattern (%s): rc %d, %.3f ms per launch-equivalent (%.0f G lookups/s)" % ("512x10" if variant == 0 else "1024x5", rc, ms.value, 149760 * 5000 / ms.value / 1e6))
