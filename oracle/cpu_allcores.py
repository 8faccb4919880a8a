"""TEST/BENCH INFRASTRUCTURE ONLY (bench.py's cpu_baseline leg): the CPU path on ALL host cores.

The reference is single-threaded, so "all cores" = W independent worker processes, each holding its own copy of the compiled
reference (oracle/_ref, or the plain-C oracle when that is absent) and replaying a disjoint share of the InnerBnB calls recorded
from the GPU registration (same rotation, level and incumbent => the same searches, `GoICP::InnerBnB`, reference
jly_goicp.cpp:286-579) until a time budget is spent. Prints one JSON line: aggregate cube-evals/s and the worker count.

    python -m oracle.cpu_allcores <jobs.npz> <workers> <seconds>
jobs.npz: cfg[18], model[Nm,3], data[Nd,3], S, geom[8], dist[S^3], R[n,3,3], level[n], E[n]
"""
import json
import multiprocessing as mp
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _rot_cloud(R, d):
    R = np.asarray(R, np.float32); d = np.asarray(d, np.float32)
    return np.stack([(R[k, 0] * d[:, 0] + R[k, 1] * d[:, 1]) + R[k, 2] * d[:, 2] for k in range(3)], 1).astype(np.float32)


def _worker(args):
    path, w, W, seconds = args
    from oracle import refbind
    J = np.load(path)
    kind = "reference" if os.path.exists(refbind.REF_SO) else "port"
    r = refbind.Ref() if kind == "reference" else refbind.Oracle()
    r.set_config(J["cfg"]); r.set_clouds(J["model"], J["data"]); r.load_dt(int(J["S"]), J["geom"], J["dist"])
    r.initialize(); r.counters(reset=True)
    spent = 0.0; calls = 0
    n = len(J["level"])
    k = w
    while spent < seconds and n:
        j = k % n                                    # a worker that runs out of its share wraps around (throughput sample)
        r.set_rotated(_rot_cloud(J["R"][j], J["data"])); r.set_opt_error(float(J["E"][j]))
        t0 = time.perf_counter(); r.inner_bnb(int(J["level"][j]), want_node=False); spent += time.perf_counter() - t0
        calls += 1; k += W
    cnt = r.counters(reset=True)
    if kind == "reference":
        evals = 8 * (cnt[0] - calls + getattr(r, "trans_empty", 0))     # every pop expands 8 cubes except the final threshold break
    else:
        evals = cnt[2]                                                 # the oracle counts cube evaluations itself (n_cube)
    return evals, spent, calls, kind


def main():
    path, W, seconds = sys.argv[1], int(sys.argv[2]), float(sys.argv[3])
    t0 = time.perf_counter()
    with mp.get_context("fork").Pool(W) as pool:
        res = pool.map(_worker, [(path, w, W, seconds) for w in range(W)])
    wall = time.perf_counter() - t0
    evals = sum(r[0] for r in res); busy = max(r[1] for r in res)
    print(json.dumps({"value": evals / busy, "unit": "cube-evals/s", "cores": W, "kind": res[0][3], "cube_evals": int(evals),
                      "calls": int(sum(r[2] for r in res)), "seconds_busy_max": busy, "seconds_wall": wall}))


if __name__ == "__main__":
    main()
