"""BASELINE configs[4] end to end (GPU box): GoICP::Register() on the synthetic 50 000-point pair, 500^3 grid, MSEThresh 1e-5, untrimmed -- the
deep-BnB stress shape -- to convergence or to a stated budget, in the default mode (parity_mode 1); then the same registration in parity_mode 2
under the same pop budget as a self-consistency check of the first pops (the CPU cannot finish this shape; the oracle prefix test stays in
tests/test_gpu_parity.py::test_bounded_search_prefix_against_the_oracle).
    python tools/c5_register.py [wall_seconds=120] [search_pops=400000] [points=50000] [dt=500] [mse=1e-5] [outlier_frac=0.1]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import goicp_amd
from goicp_amd.synth import synth_pair

wall = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
spops = int(sys.argv[2]) if len(sys.argv) > 2 else 400000
n = int(sys.argv[3]) if len(sys.argv) > 3 else 50000
S = int(sys.argv[4]) if len(sys.argv) > 4 else 500
mse = float(sys.argv[5]) if len(sys.argv) > 5 else 1e-5
outl = float(sys.argv[6]) if len(sys.argv) > 6 else 0.1

P = synth_pair(n, seed=20260101, theta_range=(2.2, 3.1), outlier_frac=outl)
cfg = goicp_amd.cfg_vector(MSEThresh=mse, trimFraction=0.0, regularization=0.0, regularizationFPFH=0.0, cfpfh=0, ponderation=0, norm=2, distTransSize=S)


def run(pm, wall_s, pops):
    g = goicp_amd.GoICP(cfg, parity_mode=pm, dt_mode=1); g.set_clouds(P["t"], P["s"])
    t0 = time.perf_counter(); g.build_dt(); t_dt = time.perf_counter() - t0
    g.set_budget(wall_s, pops)
    t0 = time.perf_counter(); out = g.register(); t = time.perf_counter() - t0
    g.close()
    rec = dict(parity_mode=pm, points=n, dt=S, MSEThresh=mse, outlier_frac=outl, wall_budget_s=wall_s, search_pop_budget=pops, dt_build_seconds=t_dt,
               register_seconds=t, budget_hit=out["budget_hit"], error=float(out["error"]), trace=out["trace"], stats=out["stats"], search=out["search"],
               kernel_ms=out["kernel_ms"], dR_vs_planted=float(np.abs(out["R"] - P["Rtrue"]).max()), dt_vs_planted=float(np.abs(out["t"] - P["ttrue"]).max()),
               R=out["R"].tolist(), t=out["t"].tolist())
    print(json.dumps(rec), flush=True)
    return rec


a = run(1, wall, spops)
# self-consistency: the verification mode under the same budgets must walk the same rotation nodes and find the same incumbents while both run
b = run(2, wall, spops)
m = min(len(a["trace"]), len(b["trace"]))
same_trace = all(x[0] == y[0] and abs(x[1] - y[1]) <= max(1e-5, n * 2.0 ** -24) * abs(y[1]) for x, y in zip(a["trace"][:m], b["trace"][:m]))
print(json.dumps(dict(summary=True, common_trace_prefix=m, common_prefix_agrees=same_trace, error_pm1=a["error"], error_pm2=b["error"],
                      trans_pops_pm1=a["stats"]["trans_pops"], trans_pops_pm2=b["stats"]["trans_pops"],
                      rot_pops_pm1=a["stats"]["rot_pops"], rot_pops_pm2=b["stats"]["rot_pops"])))
