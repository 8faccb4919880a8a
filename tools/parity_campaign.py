"""Randomised parity campaign (run on the GPU box): GoICP::Register() on the HIP path vs the plain-C oracle on seeded cavity-like
pairs over a cycle of configurations. Prints one line per pair and a summary; exit status 1 on any mismatch.
Test infrastructure (uses oracle/): python tools/parity_campaign.py [n_pairs] [first_pair]   (the pair number seeds the pair, so a second
campaign with first_pair = the first one's n_pairs covers fresh clouds)"""
import os, sys, time
from concurrent.futures import ThreadPoolExecutor
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import goicp_amd
from oracle.refbind import cfg_vector, error_trace
from helpers import synth_pair, oracle_for

CONFIGS = [
    ("default", {}),
    ("cfpfh1", dict(cfpfh=1, regularizationFPFH=0.000005)),
    ("trim", dict(trimFraction=0.1, MSEThresh=0.02)),
    ("norm1", dict(norm=1)),
    ("S50", dict(distTransSize=50, MSEThresh=0.02)),
    ("cfpfh2", dict(cfpfh=2, regularizationFPFH=0.000005)),
    ("geo_trim", dict(trimFraction=0.2, MSEThresh=0.03, regularization=0.0, ponderation=0)),
    ("color8", dict(cfpfh=3, regularizationFPFH=0.00005, regularization=0.0)),
]


def one(k):
    name, over = CONFIGS[k % len(CONFIGS)]
    rng = np.random.default_rng(9000 + k)
    n = int(rng.integers(120, 420))
    P = synth_pair(n, seed=31000 + k, noise=0.01, outlier_frac=float(rng.choice([0.0, 0.05, 0.15])), theta_range=(0.3, np.pi), corr_features=True)
    v, _ = cfg_vector(**over)
    t0 = time.time()
    g = goicp_amd.GoICP(goicp_amd.cfg_vector(**over)); g.set_icp_one_launch(k & 1)      # odd pairs: ICP3D::Run as one launch (what a batch switches on)
    g.set_clouds(P["t"], P["s"], P["tc"], P["sc"], P["tf"], P["sf"]); g.build_dt()
    rg = g.register(); g.close(); t_gpu = time.time() - t0
    if rg["stats"]["cube_evals"] > 300000:            # the single-threaded oracle would need minutes: not compared
        return k, name, n, None, float("nan"), float(rg["error"]), 0.0, t_gpu, rg["stats"]["cube_evals"]
    t0 = time.time()
    o = oracle_for(P, v); ro = o.register(); t_cpu = time.time() - t0
    ok = (float(rg["error"]) == float(ro["error"]) and rg["comp"] == ro["comp"]
          and np.abs(rg["R"] - ro["R"]).max() < 1e-9 and np.abs(rg["t"] - ro["t"]).max() < 1e-9
          and [(a, float("%g" % b)) for a, b in rg["trace"]] == error_trace(ro["log"]))
    return k, name, n, ok, float(ro["error"]), float(rg["error"]), t_cpu, t_gpu, rg["stats"]["cube_evals"]


if __name__ == "__main__":
    npairs = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    bad = 0
    with ThreadPoolExecutor(max_workers=12) as ex:
        for k, name, n, ok, eo, eg, tc, tg, cev in ex.map(one, range(first, first + npairs)):
            print("pair %3d %-9s N=%3d %s oracle %.6g gpu %.6g  cpu %.2fs gpu %.3fs cube_evals %d" % (k, name, n, "SKIP" if ok is None else ("OK  " if ok else "MISMATCH"), eo, eg, tc, tg, cev), flush=True)
            bad += 1 if ok is False else 0
    print("pairs %d mismatches %d" % (npairs, bad))
    sys.exit(1 if bad else 0)
