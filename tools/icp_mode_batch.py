"""configs[3] batch with ICP3D::Run as one launch per call (k_icp_loop, GOICP_DEVICE_ICP=1) and as one launch per iteration (=0), and with the
one-launch path limited to small clouds (GOICP_ICP_LOOP_MAX). Run on the GPU box: python tools/icp_mode_batch.py [rows] [threads]"""
import os, sys, time, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import goicp_amd
from goicp_amd import driver, c4_batch

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 383
threads = driver.worker_threads(int(sys.argv[2]) if len(sys.argv) > 2 else 16)
base = os.path.join(tempfile.gettempdir(), "goicp_icp_mode_%d" % os.getpid())
tsv, cfgp, _ = c4_batch.materialise(base, os.path.join(ROOT, "tests", "golden", "data"), rows=rows)
reg = driver.gpu_register_fn(base, device_id=0)


def run(tag, out):
    t0 = time.perf_counter()
    recs, _ = driver.run_batch(base, tsv, cfgp, os.path.join(base, out), reg, threads=threads, rmsd=False)
    dt = time.perf_counter() - t0
    print("%-34s %6.1f pairs/s  failed %d  checksum %.9g" % (tag, len(recs) / dt, sum(1 for r in recs if r[20] != 0), sum(float(r[13]) for r in recs)), flush=True)


print("rows %d, registrations in flight %d" % (rows, threads), flush=True)
run("warm-up", "warm")
for rep in range(2):
    for env in ({"GOICP_DEVICE_ICP": "1"}, {"GOICP_DEVICE_ICP": "0"}, {"GOICP_DEVICE_ICP": "1", "GOICP_ICP_LOOP_MAX": "200"}, {"GOICP_DEVICE_ICP": "1", "GOICP_ICP_LOOP_MAX": "300"}):
        for k in ("GOICP_DEVICE_ICP", "GOICP_ICP_LOOP_MAX"):
            os.environ.pop(k, None)
        os.environ.update(env)
        run(" ".join("%s=%s" % kv for kv in env.items()), "o%d" % rep)
