"""Randomised parity campaign above 1024 points (run on the GPU box): GoICP::Register() with parity_mode 2 (reference-order sums on rows in HBM;
register-resident clouds up to 5120 points, streamed beyond) vs the plain-C oracle on seeded synthetic pairs -- geometric error, trimmed and
untrimmed, 50^3 / 100^3 grids. One line per pair and a summary; exit status 1 on any mismatch.
Test infrastructure (uses oracle/): python tools/parity_campaign_big.py [n_pairs]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import goicp_amd
from oracle.refbind import cfg_vector, error_trace
from helpers import synth_pair, oracle_for


def one(k):
    rng = np.random.default_rng(77000 + k)
    n = int(rng.choice([1100, 1500, 2048, 2049, 3000, 4096, 5000, 5120, 5121, 7000, 9000]))
    trim = float(rng.choice([0.0, 0.1, 0.2]))
    S = int(rng.choice([50, 100]))
    outl = 0.0 if trim == 0.0 else float(rng.choice([0.05, 0.1]))         # (an untrimmed search on outliers runs for hours, in the reference too)
    over = dict(distTransSize=S, trimFraction=trim, regularization=0.0, ponderation=0, MSEThresh=0.002)
    P = synth_pair(n, seed=88000 + k, noise=0.005, outlier_frac=outl, theta_range=(0.3, np.pi))
    v, _ = cfg_vector(**over)
    t0 = time.time()
    g = goicp_amd.GoICP(goicp_amd.cfg_vector(**over), parity_mode=2); g.set_clouds(P["t"], P["s"]); g.build_dt()
    rg = g.register(); g.close(); t_gpu = time.time() - t0
    if rg["stats"]["cube_evals"] > 120000:
        return k, n, S, trim, None, float("nan"), float(rg["error"]), 0.0, t_gpu, rg["stats"]["cube_evals"]
    t0 = time.time()
    o = oracle_for(P, v); ro = o.register(); t_cpu = time.time() - t0
    ok = (float(rg["error"]) == float(ro["error"]) and np.abs(rg["R"] - ro["R"]).max() < 1e-9 and np.abs(rg["t"] - ro["t"]).max() < 1e-9
          and [(a, float("%g" % b)) for a, b in rg["trace"]] == error_trace(ro["log"]))
    return k, n, S, trim, ok, float(ro["error"]), float(rg["error"]), t_cpu, t_gpu, rg["stats"]["cube_evals"]


if __name__ == "__main__":
    npairs = int(sys.argv[1]) if len(sys.argv) > 1 else 24
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 0          # (the pair number seeds the pair: a second campaign starts where the first ended)
    bad = 0
    for k in range(first, first + npairs):
        k, n, S, trim, ok, eo, eg, tc, tg, cev = one(k)
        print("pair %3d N=%5d S=%3d trim %.1f %s oracle %.9g gpu %.9g  cpu %.1fs gpu %.2fs cube_evals %d" % (k, n, S, trim, "SKIP" if ok is None else ("OK  " if ok else "MISMATCH"), eo, eg, tc, tg, cev), flush=True)
        bad += 1 if ok is False else 0
    print("pairs %d mismatches %d" % (npairs, bad))
    sys.exit(1 if bad else 0)
