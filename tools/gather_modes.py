"""[needs the diagnosis build: make -C go-icp-protein-cavities_amd/csrc probe; run with GOICP_LIB=go-icp-protein-cavities_amd/libgoicp_hip_probe.so] Diagnosis (GPU box): what the memory system charges the metric kernel's gathers for, and the workgroup -> XCD map.
(a) goicp_engine_debug_gather_probe variants on the real tiles: all 8 lookups per point / only 2 of them (the misses) / 8 loads of ONE
cell (1 line, 7 loads that find it requested) / 8 loads on 8 grid rows (8 lines);  (b) the metric launch under GOICP_XCD = 0, 1, 2.
python tools/gather_modes.py"""
import os, sys, time, ctypes as C
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


def engine():
    g.initialize(); e = g.engine()
    rng = np.random.default_rng(20260102)
    for s in range(16):
        aa = rng.normal(size=3); aa = aa / np.linalg.norm(aa) * rng.uniform(0, np.pi); th = np.linalg.norm(aa); k = aa / th
        K = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
        e.set_rotation(s, (np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * K @ K).astype(np.float32))
    return e


def launch_ms(e, reps=30):
    for _ in range(3):
        e.eval_cubes_multi_dev(sl, lv, cubes.data_ptr(), n, ub.data_ptr(), lb.data_ptr())
    e.synchronize(); t0 = time.perf_counter()
    for _ in range(reps):
        e.eval_cubes_multi_dev(sl, lv, cubes.data_ptr(), n, ub.data_ptr(), lb.data_ptr())
    e.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


names = {0: "8 lookups per point (real tile)", 2: "2 of the 8 (cubes 0 and 4: the misses alone)", 3: "8 loads of one cell (1 line per point)", 4: "8 loads on 8 grid rows (8 lines per point)", 5: "real lookups, lane = (point, cube)", 6: "same, y/z cells computed once per point and shared"}
for coded in ("1", "0"):
    os.environ["GOICP_CODED"] = coded
    e = engine(); launch_ms(e, 2)
    for variant in (0, 2, 3, 4, 5, 6, 0, 5, 6):
        ms = C.c_float()
        rc = e.L.goicp_engine_debug_gather_probe(e.h, C.c_void_p(cubes.data_ptr()), n, C.c_void_p(ub.data_ptr()), variant, C.byref(ms))
        print("coded %s probe %d %-48s rc %d  %.3f ms" % (coded, variant, names[variant], rc, ms.value), flush=True)
os.environ["GOICP_CODED"] = "1"
ref = None
for fuse in ("1", "0"):
    os.environ["GOICP_FUSE"] = fuse
    for xm in ("0", "1"):
        os.environ["GOICP_XCD"] = xm
        e = engine(); ms = launch_ms(e)
        u = ub.clone(); l = lb.clone()
        if ref is None:
            ref = (u, l)
        same = bool(torch.equal(u, ref[0]) and torch.equal(l, ref[1]))
        print("fuse %s GOICP_XCD %s: %.3f ms per launch (%s) outputs equal to the first run: %s" % (fuse, xm, ms, e.eval_kernel_name(), same), flush=True)
