"""Batch driver: the role of the reference's bo1_GoICP.py (one registration per .tsv row, reference bo1_GoICP.py:6-51) with the
pairs sharded over the GPUs of one node -- one process per GPU, no data-path collective; a single all_gather of fixed-size
result records (the RMSD of every pair among them) at the end.

Scheduling. Pair cost varies by orders of magnitude (0.06 s vs 35 s on the reference), so the rows are taken
longest-first by a cheap cost proxy (source atoms x target atoms):
  * schedule="dynamic" (default when a process group is up): every rank's worker threads draw the next row of the sorted
    list from ONE shared counter (an atomic add on the process group's key-value store -- control path only, a few bytes per
    pair), so a rank that finishes early keeps pulling work;
  * schedule="lpt": static greedy longest-processing-time assignment (each row to the least-loaded rank), no communication;
  * schedule="rr": row k -> rank k mod world (kept for reference).
"""
import itertools
import os
import threading
import time

import numpy as np

REC = 24   # floats per result record: pair id, R[9], t[3], error, compat, Nd, seconds, dt_seconds, cube_evals, rmsd, status, rank, pad


def read_tsv_pairs(path):
    """rows -> (source id, target id): whitespace split, source = field 3, target = field 4, stop at the first
    blank line (reference bo1_GoICP.py:11-17)."""
    pairs = []
    with open(path) as f:
        for line in f:
            t = line.split()
            if len(t) < 4:
                break
            pairs.append((t[2], t[3]))
    return pairs


def mol2_atom_count(path):
    """third space-separated token of line 6 (reference bo1_GoICP.py:45-48)."""
    with open(path) as f:
        lines = f.readlines()
    return int(lines[5].split(" ")[2]) if len(lines) > 5 and len(lines[5].split(" ")) > 2 and lines[5].split(" ")[2] else int(lines[5].split()[0])


def shard(n_items, rank, world):
    """static round-robin shard (pair k -> rank k mod world)."""
    return list(range(rank, n_items, world))


def lpt_order(costs):
    """indices by decreasing cost (stable: equal costs keep row order)."""
    return sorted(range(len(costs)), key=lambda i: (-costs[i], i))


def lpt_assign(costs, world):
    """greedy longest-processing-time partition: rows in decreasing cost, each to the least-loaded rank (lowest rank on ties).
    Returns one index list per rank, each in decreasing cost."""
    load = [0.0] * world
    out = [[] for _ in range(world)]
    for i in lpt_order(costs):
        r = min(range(world), key=lambda q: (load[q], q))
        out[r].append(i); load[r] += costs[i]
    return out


_batch_seq = itertools.count()     # run_batch calls of this process; every rank makes them in the same order, so the numbers agree across ranks


class _SharedCounter:
    """next() hands out 0, 1, 2, ... exactly once across all ranks and threads: an atomic add on the process group's store.
    The store keeps a key for the life of the process group, so every run_batch call counts on a key of its own (`key` already
    carries the call's sequence number)."""

    def __init__(self, dist, key):
        from torch.distributed import distributed_c10d as c10d
        self.store = c10d._get_default_store()
        self.key = key
        self.lock = threading.Lock()

    def next(self):
        with self.lock:                              # one outstanding request per process (the store client is not re-entrant)
            return int(self.store.add(self.key, 1)) - 1

    def sum_over_ranks(self, name, value, world, timeout_s=600.0):
        """Every rank adds `value` under `name` and waits (control path, store only -- no collective, any backend) until all `world`
        ranks have done so; returns the total. Used to agree on a failure before anybody enters the record gather."""
        with self.lock:
            self.store.add(self.key + "/" + name, int(value))
            self.store.add(self.key + "/" + name + "/arrived", 1)
            t0 = time.time()
            while int(self.store.add(self.key + "/" + name + "/arrived", 0)) < world:
                if time.time() - t0 > timeout_s:
                    raise RuntimeError("run_batch: a rank did not reach the end of the batch within %.0f s" % timeout_s)
                time.sleep(0.002)
            return int(self.store.add(self.key + "/" + name, 0))


class _LocalCounter:
    def __init__(self):
        self.v = 0; self.lock = threading.Lock()

    def next(self):
        with self.lock:
            v = self.v; self.v += 1
            return v


def pack_record(pair_id, res, rmsd=float("nan"), rank=0):
    r = np.full(REC, np.nan, np.float64)
    r[0] = pair_id; r[1:10] = np.asarray(res["R"]).reshape(9); r[10:13] = res["t"]; r[13] = res["error"]; r[14] = res["compat"]
    r[15] = res["Nd"]; r[16] = res["seconds"]; r[17] = res.get("dt_seconds", np.nan); r[18] = res.get("cube_evals", np.nan); r[19] = rmsd
    r[20] = res.get("status", 0); r[21] = rank
    return r


def gather_records(local_records, n_total, dist=None, device=None):
    """all_gather of the per-rank record blocks; returns [n_registered, REC] ordered by pair id. Under dynamic scheduling a rank may
    hold any number of records, so the blocks are padded to n_total rows (383 x 24 doubles = 74 KB per rank: latency-bound, one
    collective)."""
    import torch
    world = dist.get_world_size() if dist is not None and dist.is_initialized() else 1
    per = max(int(n_total), len(local_records), 1)
    block = np.full((per, REC), np.nan, np.float64)
    for i, r in enumerate(local_records):
        block[i] = r
    t = torch.from_numpy(block)
    if device is not None:
        t = t.to(device)
    if world > 1:
        outs = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(outs, t)
        allb = torch.cat(outs, 0).cpu().numpy()
    else:
        allb = t.cpu().numpy()
    allb = allb[~np.isnan(allb[:, 0])]
    order = np.argsort(allb[:, 0], kind="stable")
    return allb[order]


def protein_paths(base_dir, src, tgt):
    """Files of the protein stage (reference layout: chains/, ref_proteins/<src>.<tgt>/, rot/ -- jly_main.cpp:159-172 (disabled
    there), README.md:20-25); None where the inputs are absent."""
    chain = os.path.join(base_dir, "chains", src + "_protein.mol2")
    aligned = os.path.join(base_dir, "ref_proteins", "%s.%s" % (src, tgt), "aligned_%s_protein.mol2" % src)
    if not (os.path.exists(chain) and os.path.exists(aligned)):
        return None
    os.makedirs(os.path.join(base_dir, "rot"), exist_ok=True)
    return chain, aligned, os.path.join(base_dir, "rot", "rot_%s_protein.mol2" % src)


def rescaled_name(output):
    """"<output stem>_rescaled.txt" as goicp_run_pair_files writes it: the reference cuts the output path at its first '.'
    (jly_main.cpp:155); the cut here is the first '.' of the FILE NAME, so a dot in a directory name cannot map every row to one
    file (same names for the reference's own layout output/similarK.txt)."""
    d, b = os.path.split(output)
    return os.path.join(d, b[:b.find(".")] if "." in b else b) + "_rescaled.txt"


def pair_rmsd(base_dir, src, tgt, output, k=None):
    """RMSD of the registration just written to `output`: the source protein is transformed with the rescaled R, t of THIS run
    (Transformation::applyTransformationProtein, reference transformation.cpp:469-539) and compared with the aligned reference
    protein (Transformation::computeRMSD, :453-464). NaN when the protein files do not exist.
    A source id repeats in up to ten rows of the shipped .tsv and several rows are in flight at once (threads, ranks on one file
    system), so the transformed protein is written under a name of THIS row, measured there, and only then moved (atomically) to
    the reference's name rot/rot_<src>_protein.mol2 -- which therefore always holds one complete file of one of those rows."""
    from . import binding
    pp = protein_paths(base_dir, src, tgt)
    if pp is None:
        return float("nan")
    chain, aligned, rot = pp
    resc = rescaled_name(output)
    mine = "%s.k%s.%d.%d.tmp" % (rot, "x" if k is None else str(k), os.getpid(), threading.get_ident())
    if not os.path.exists(resc) or not binding.apply_transformation_protein(resc, chain, mine):
        return float("nan")
    try:
        return binding.compute_rmsd(aligned, mine)
    finally:
        os.replace(mine, rot)


def gpu_register_fn(base_dir, device_id=0):
    """The registration the batch runs per row: the reference's main() on files through the C ABI (goicp_run_pair_files)."""
    from . import binding

    def reg(tf, sf, nd, config, output, k):
        return binding.run_pair_files(base_dir, tf, sf, nd, config, output, k, device_id=device_id)

    def configure_batch(threads):
        # registrations that share a GPU speculate less (measured at 14 in flight: 128 nodes ahead 549 pairs/s, 16 nodes 608, 4 nodes 518)
        binding.set_batch_speculation(batch_speculation(threads))
        # ICP3D::Run as one launch per call (one workgroup) pays when the GPU is shared: 14 in flight 634 vs 519 pairs/s; alone 5.2 vs 4.3 ms per pair
        binding.set_batch_icp_one_launch(1 if int(threads) >= 3 else 0)
    reg.configure_batch = configure_batch
    return reg


def batch_speculation(threads):
    """Rotation-queue nodes evaluated ahead per launch when `threads` registrations share one GPU: 128 alone, 16 from 16 in flight on."""
    return int(min(128, max(16, 256 // max(1, int(threads)))))


def list_jobs(base_dir, tsv, limit=None):
    """(k, target file, source file, Nd, cost proxy, src id, tgt id) for every row whose cavity files exist (the reference tree
    ships 2 of the 383); k = 1-based row number = the reference's pair number."""
    pairs = read_tsv_pairs(tsv)
    if limit:
        pairs = pairs[:limit]
    jobs = []
    for idx, (src, tgt) in enumerate(pairs):
        sf = os.path.join("cavities", src + "_cavity6.mol2"); tf = os.path.join("cavities", tgt + "_cavity6.mol2")
        if not (os.path.exists(os.path.join(base_dir, sf)) and os.path.exists(os.path.join(base_dir, tf))):
            continue
        nd = mol2_atom_count(os.path.join(base_dir, sf)); nm = mol2_atom_count(os.path.join(base_dir, tf))
        jobs.append((idx + 1, tf, sf, nd, float(nd) * float(nm), src, tgt))
    return jobs, len(pairs)


def host_cpu_budget():
    """CPUs this process may keep busy: the smaller of its affinity set and its cgroup's CFS quota (cpu.max of cgroup v2, cfs_quota_us /
    cfs_period_us of v1). A GPU box hands a job 16 CPUs' worth of QUOTA on 256 visible CPUs; worker threads that spin on the device
    (the default wait) burn the quota, and a cgroup over its quota has ALL its threads stopped until the next period: 16 spinning threads
    plus the runtime's helpers ran the configs[3] batch at 370 pairs/s with 8-10 throttling events, 14 threads at 540 with one
    (tools/throttle_probe.sh)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = min(n, max(1, int(float(q) / float(per))))
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read()); per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0 and per > 0:
                n = min(n, max(1, q // per))
        except (OSError, ValueError):
            pass
    return n


def worker_threads(wanted, world=1, budget=None):
    """Registrations to keep in flight on this rank: `wanted`, but never more spinning threads than this rank's share of the CPU budget
    minus two (the main thread and the HIP runtime's helpers need the rest). Threads that sleep while they wait (GOICP_WAIT=block,
    GOICP_SCHED=yield|block) do not burn the budget: `wanted` stands."""
    if os.environ.get("GOICP_WAIT", "").split(":")[0] in ("block", "nap") or os.environ.get("GOICP_SCHED") in ("yield", "block"):
        return max(1, int(wanted))
    b = host_cpu_budget() if budget is None else budget
    return max(1, min(int(wanted), b // max(1, int(world)) - 2))


def run_batch(base_dir, tsv, config, out_dir, register_fn, rank=0, world=1, limit=None, threads=1, schedule=None, dist=None, rmsd=True,
              counter_key="goicp_batch_next"):
    """register_fn(target_file, source_file, nd, config, output, k) -> result dict. Rows whose cavity files are missing are
    skipped. threads > 1 keeps that many registrations of this rank in flight (one engine + HIP stream each; the C entry points
    release the GIL). Returns (records of this rank in the order they were taken, number of rows)."""
    jobs, n_rows = list_jobs(base_dir, tsv, limit)
    costs = [j[4] for j in jobs]
    seq = next(_batch_seq)
    have_pg = dist is not None and dist.is_initialized() and dist.get_world_size() > 1
    if schedule is None:
        schedule = "dynamic" if (have_pg or world == 1) else "lpt"
    order = lpt_order(costs)
    counter = shared = None
    if schedule == "dynamic":
        if have_pg:
            try:
                counter = shared = _SharedCounter(dist, "%s_%d" % (counter_key, seq))
            except Exception:
                schedule = "lpt"                    # no store access: static longest-first partition instead
        elif world == 1:
            counter = _LocalCounter()
        else:
            schedule = "lpt"
    if schedule == "lpt":
        mine = lpt_assign(costs, world)[rank]
    elif schedule == "rr":
        mine = shard(len(jobs), rank, world)
    else:
        mine = None
    os.makedirs(out_dir, exist_ok=True)

    def one(job):
        k, tf, sf, nd, _, src, tgt = job
        output = os.path.join(out_dir, "similar%d.txt" % k)
        res = register_fn(tf, sf, nd, config, output, k)
        val = pair_rmsd(base_dir, src, tgt, output, k) if (rmsd and int(res.get("status", 0)) == 0) else float("nan")
        return pack_record(k, res, rmsd=val, rank=rank)

    recs = []; lock = threading.Lock()
    if mine is not None:
        local = _LocalCounter()

        def take():
            i = local.next()
            return mine[i] if i < len(mine) else None
    else:
        def take():
            i = counter.next()
            return order[i] if i < len(order) else None

    errors = []

    def worker():
        try:
            while not errors:
                j = take()
                if j is None:
                    return
                r = one(jobs[j])
                with lock:
                    recs.append(r)
        except BaseException as e:                  # surfaced after the join: a failed registration must not pass silently
            errors.append(e)
    nthreads = max(1, int(threads))
    if hasattr(register_fn, "configure_batch"):
        register_fn.configure_batch(nthreads)
    if nthreads > 1:
        ts = [threading.Thread(target=worker) for _ in range(nthreads)]
        for t in ts:
            t.start()
        for t in ts:
            t.join()
    else:
        worker()
    # A rank that failed must not leave the others waiting in the record gather that follows: under a process group all ranks agree
    # on the failure count first (store only), and every rank raises.
    failed_ranks = 0
    if have_pg:
        if shared is None:
            try:
                shared = _SharedCounter(dist, "%s_%d" % (counter_key, seq))
            except Exception:
                shared = None
        if shared is not None:
            failed_ranks = shared.sum_over_ranks("failed", 1 if errors else 0, dist.get_world_size())
    if errors:
        raise errors[0]
    if failed_ranks:
        raise RuntimeError("run_batch: %d other rank(s) failed; no records gathered" % failed_ranks)
    recs.sort(key=lambda r: r[0])
    return recs, n_rows
