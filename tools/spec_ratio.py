"""What the speculation of the configs[3] batch costs: pairs/s and queue pops launched per pop consumed (goicp_batch_search_totals) over the
speculation depth (GOICP_SPECULATE) and the pop budget of a speculative job (GOICP_SPEC_POPS), with the batch's usual number of
registrations in flight. Run on the GPU box: python tools/spec_ratio.py [rows] [threads]"""
import os, sys, time, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import goicp_amd
from goicp_amd import driver, c4_batch

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 383
threads = driver.worker_threads(int(sys.argv[2]) if len(sys.argv) > 2 else 16)
base = os.path.join(tempfile.gettempdir(), "goicp_spec_ratio_%d" % os.getpid())
tsv, cfgp, _ = c4_batch.materialise(base, os.path.join(ROOT, "tests", "golden", "data"), rows=rows)
reg = driver.gpu_register_fn(base, device_id=0)


def run(tag, out):
    goicp_amd.batch_search_totals(reset=True)
    t0 = time.perf_counter()
    recs, _ = driver.run_batch(base, tsv, cfgp, os.path.join(base, out), reg, threads=threads, rmsd=False)
    dt = time.perf_counter() - t0
    t = goicp_amd.batch_search_totals()
    print("%-28s %6.1f pairs/s  pops launched/consumed %.2f (%d / %d)  jobs %.2f  launches %d  failed %d" % (
        tag, len(recs) / dt, t["pops_launched"] / max(1, t["pops_consumed"]), t["pops_launched"], t["pops_consumed"],
        t["jobs_launched"] / max(1, t["inner_calls_consumed"]), t["launches"], sum(1 for r in recs if r[20] != 0)), flush=True)


print("rows %d, registrations in flight %d" % (rows, threads), flush=True)
run("warm-up", "warm")
for depth, pops in [(None, None), (16, 20000), (16, 5000), (16, 2000), (16, 1000), (12, 5000), (8, 20000), (8, 5000), (8, 2000), (4, 20000), (2, 20000), (1, 20000),
                    (32, 2000), (128, 20000), (None, None)]:
    for k, v in (("GOICP_SPECULATE", depth), ("GOICP_SPEC_POPS", pops)):
        if v is None: os.environ.pop(k, None)
        else: os.environ[k] = str(v)
    run("depth %s spec_pops %s" % ("driver" if depth is None else depth, "default" if pops is None else pops), "o_%s_%s" % (depth, pops))
