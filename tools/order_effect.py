"""Why did the cavity batch run 2.3x slower behind the registration legs of bench.py (GPU box)? The same batch three times in one process:
(A) first, (B) right behind 64 configs[2]-shaped registrations in flight, (C) after 20 idle seconds.  python tools/order_effect.py"""
import os, sys, time, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import goicp_amd
from goicp_amd import driver, c4_batch
from goicp_amd.synth import synth_pair
import bench
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
base = os.path.join(tempfile.gettempdir(), "goicp_order_%d" % os.getpid())
tsv, cfgp, _ = c4_batch.materialise(base, os.path.join(ROOT, "tests", "golden", "data"), rows=383)
reg = driver.gpu_register_fn(base)
driver.run_batch(base, tsv, cfgp, os.path.join(base, "w"), reg, threads=16, limit=32, rmsd=False)


def batch(tag):
    t0 = time.perf_counter(); recs, n = driver.run_batch(base, tsv, cfgp, os.path.join(base, "o" + tag), reg, threads=16); t = time.perf_counter() - t0
    print("%s: %d rows in %.3f s = %.1f pairs/s" % (tag, len(recs), t, len(recs) / t), flush=True)


batch("A (first)")
cfg = goicp_amd.cfg_vector(MSEThresh=0.001, trimFraction=0.1, regularization=0.0, regularizationFPFH=0.0, cfpfh=0, ponderation=0, norm=2, distTransSize=300)
r = bench.overlapped_registrations(goicp_amd, synth_pair, cfg, 5000, 64, 0, 1, 0, False, None, dev, torch, dt_mode=1, threads=16, tag="in_flight")
print("64 registrations in flight: %.1f pairs/s" % r["pair_registrations_per_sec_incl_dt_build_in_flight"], flush=True)
batch("B (behind the registrations)")
L = goicp_amd.load_library(); L.goicp_engine_release_cached_memory()
batch("B2 (after releasing the pooled blocks and streams)")
time.sleep(20)
batch("C (after 20 idle seconds)")
