"""configs[2]-shaped registration, repeated (GPU box): Register() seconds, the engine's kernel milliseconds and the ICP iterations of each
repeat, grid built on the device (dt_mode=1) so a repeat costs milliseconds. For A/B runs of the ICP / search side:
GOICP_LIB=<library> python tools/c3_register_perf.py [repeats]. The result (error, R, t) is printed so variants can be compared."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
if os.environ.get("C3_WITH_TORCH"):          # the same inside a process that has PyTorch's HIP runtime up (as bench.py does)
    import torch
    torch.zeros(1, device="cuda")
import goicp_amd
from goicp_amd.synth import synth_pair

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 7
P = synth_pair(5000, seed=20260101, theta_range=(2.2, 3.1))
cfg = goicp_amd.cfg_vector(MSEThresh=0.001, trimFraction=0.1, regularization=0.0, regularizationFPFH=0.0, cfpfh=0, ponderation=0, norm=2, distTransSize=300)
g = goicp_amd.GoICP(cfg, dt_mode=int(os.environ.get("C3_DT_MODE", "1"))); g.set_clouds(P["t"], P["s"]); g.build_dt()
rows = []
for r in range(reps + 1):
    g.clear()
    t0 = time.perf_counter(); out = g.register(); dt = time.perf_counter() - t0
    if r:                                            # the first repeat loads code objects
        rows.append((dt * 1e3, out["kernel_ms"], out["stats"]["icp_iters"], out["stats"]["icp_calls"], out["stats"]["trans_pops"]))
A = np.array(rows)
print("%s: Register %.2f ms (min %.2f) | engine kernels %.2f ms | ICP: %d iterations in %d calls | pops %d | error %.9g R00 %.17g t0 %.17g"
      % (os.path.basename(goicp_amd.binding.LIB_PATH), np.median(A[:, 0]), A[:, 0].min(), np.median(A[:, 1]), A[0, 2], A[0, 3], A[0, 4],
         out["error"], out["R"][0, 0], out["t"][0]))
g.close()
