"""Where the wall time of the configs[3] batch goes (GPU box): per-thread-count wall time against the summed Register() /
BuildDT() seconds of the records, the heaviest rows, and the tail (when the last rows finish). Usage:
python tools/c4_breakdown.py [rows] [threads,threads,...]"""
import os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
if os.environ.get("C4_WITH_TORCH"):      # the same batch inside a process that has PyTorch's HIP runtime up (as bench.py does)
    import torch
    torch.zeros(1, device="cuda")
import goicp_amd
from goicp_amd import driver, c4_batch

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 383
tlist = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "1,8,16,32").split(",")]
goicp_amd.load_library()
base = os.path.join(tempfile.gettempdir(), "goicp_c4_bd_%d" % os.getpid())
tsv, cfgp, truth = c4_batch.materialise(base, os.path.join(ROOT, "tests", "golden", "data"), rows=rows)
reg = driver.gpu_register_fn(base)
# warm-up (code objects, pools)
driver.run_batch(base, tsv, cfgp, os.path.join(base, "out_w"), reg, threads=4, limit=16)
for T in tlist:
    done_at = []
    t0 = time.perf_counter()

    def timed(tf, sf, nd, config, output, k, _reg=reg):
        a = time.perf_counter(); r = _reg(tf, sf, nd, config, output, k); b = time.perf_counter()
        r["_wall"] = b - a; r["_end"] = b - t0
        done_at.append((k, a - t0, b - t0, r["seconds"], r["dt_seconds"], r["cube_evals"], r["Nd"]))
        return r
    recs, n = driver.run_batch(base, tsv, cfgp, os.path.join(base, "out_%d" % T), timed, threads=T)
    wall = time.perf_counter() - t0
    A = np.array(done_at)
    print("threads %2d: %d rows in %.3f s = %.1f pairs/s | sum Register %.2f s, sum BuildDT %.2f s, sum call wall %.2f s (threads x wall = %.2f) | cube evals %.2f M"
          % (T, len(recs), wall, len(recs) / wall, A[:, 3].sum(), A[:, 4].sum(), (A[:, 2] - A[:, 1]).sum(), T * wall, A[:, 5].sum() / 1e6), flush=True)
    top = A[np.argsort(-A[:, 3])][:6]
    print("   heaviest rows (k, start, end, Register s, cube evals, Nd): " + "; ".join("%d %.2f-%.2f %.3f %.2fM %d" % (r[0], r[1], r[2], r[3], r[5] / 1e6, r[6]) for r in top))
    ends = np.sort(A[:, 2])
    print("   finish times: 50%% %.2f s, 90%% %.2f s, 99%% %.2f s, last %.2f s; rows still running in the last 20%% of the wall: %d"
          % (ends[len(ends) // 2], ends[int(len(ends) * 0.9)], ends[int(len(ends) * 0.99)], ends[-1], int((A[:, 2] > 0.8 * wall).sum())), flush=True)
