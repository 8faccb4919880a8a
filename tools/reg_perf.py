#!/usr/bin/env python3
"""Registration timing on the GPU for the committed cases (run on the GPU box)."""
import os, sys, time, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import goicp_amd
from oracle.refbind import cfg_vector, read_txt_cloud
from helpers import load_clouds, load_register, GOLDEN, synth_pair

def run(name, P, cfgd, spec=128):
    v = cfg_vector(**cfgd)[0]
    g = goicp_amd.GoICP(v, speculate=spec)
    g.set_clouds(P["t"], P["s"], P.get("tc"), P.get("sc"), P.get("tf"), P.get("sf"))
    t0 = time.time(); g.build_dt(); tdt = time.time() - t0
    t0 = time.time(); out = g.register(); tr = time.time() - t0
    st = out["stats"]
    print("%-16s Nd=%5d reg %.3fs dt %.2fs kern %.1fms err %.6g pops %d cubes %d calls %d launches %d jobs %d rotpops %d icp %d/%d  us/pop(kernel) %.2f" % (
        name, g.Nd, tr, tdt, out["kernel_ms"], out["error"], st["trans_pops"], st["cube_evals"], st["inner_calls"], st["inner_launches"],
        st["inner_jobs_launched"], st["rot_pops"], st["icp_calls"], st["icp_iters"], out["kernel_ms"] * 1e3 / max(1, st["trans_pops"])))
    g.close()

G = load_register()
P1 = load_clouds("pair1"); P2 = load_clouds("pair2")
for _ in range(3):
    run("pair1_default", P1, G["pair1_default"]["cfg"])
run("pair2_default", P2, G["pair2_default"]["cfg"])
run("pair2_spec1", P2, G["pair2_default"]["cfg"], spec=1)
run("pair2_spec16", P2, G["pair2_default"]["cfg"], spec=16)
run("pair1_fpfh", P1, G["pair1_fpfh"]["cfg"])
D = os.path.join(GOLDEN, "data", "demo")
mr = read_txt_cloud(os.path.join(D, "model_rand.txt")); dr = read_txt_cloud(os.path.join(D, "data_rand.txt"))
run("rand_untrimmed", dict(t=mr, s=dr), G["rand_untrimmed"]["cfg"])
run("rand_trim", dict(t=mr, s=dr), G["rand_trim"]["cfg"])
b = np.load(os.path.join(D, "bunny.npz"))
run("bunny_S100", dict(t=b["model"], s=b["data"]), G["bunny_S100"]["cfg"])
P = synth_pair(1000, seed=20260101, theta_range=(2.2, 3.1))
run("synth1000_S100", P, dict(distTransSize=100, trimFraction=0.1, regularization=0.0, ponderation=0, MSEThresh=0.001))
