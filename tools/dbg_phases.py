# needs the diagnosis build: make -C go-icp-protein-cavities_amd/csrc probe; run with GOICP_LIB=go-icp-protein-cavities_amd/libgoicp_hip_probe.so
import os, sys, ctypes as C
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
os.environ["GOICP_DBG"] = os.environ.get("GOICP_DBG", "32")      # (160 = 32 + 128: phases with the fused pop evaluation switched off)
import numpy as np
import goicp_amd
from oracle.refbind import cfg_vector
from helpers import load_clouds, load_hotpath
from test_gpu_parity import engine_from_hotpath
for name, pn in (("pair1_fpfh","pair1"),("pair1_default","pair1"),("pair1_trim","pair1")):
    H = load_hotpath(name); P = load_clouds(pn)
    e = engine_from_hotpath(H, P)
    cfg = H["cfg"]; init = tuple(np.float32(cfg[5:9]))
    tot = np.zeros(8, np.uint64)
    for k in range(len(H["call_level"])):
        r = e.inner_bnb(R=H["call_R"][k], level=int(H["call_level"][k]), E=H["call_E"][k], sse=np.float32(H["sse"]), init=init)
    out = np.zeros(16, np.uint64)
    e.L.goicp_engine_debug_times(e.h, out.ctypes.data_as(C.c_void_p))
    pops = int(out[7]); cyc = out[:6].astype(float)
    print(name, "pops", pops, "cycles/pop by phase [pop+ctl, setup, sweep, compat, fpfh, decide+push]:", np.round(cyc / pops).astype(int), "total", int(cyc.sum() / pops), "(s_memtime ticks @100MHz? see note)")
    if out[8]:
        print("   selection: rows %d, partition rounds per row %.1f, median-of-medians rounds per row %.2f (%.0f cycles each), final insertion sort %.0f cycles per row" % (
            out[8], out[9] / out[8], out[10] / out[8], out[11] / max(1, int(out[10])), out[12] / out[8]))
        print("   cycles per partition round: pivot (incl. median of medians) %.0f | ranks + index lists %.0f | swaps %.0f" % (out[13] / out[9], out[14] / out[9], out[15] / out[9]))
    e.close()
