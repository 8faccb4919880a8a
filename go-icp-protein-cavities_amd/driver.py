"""Batch driver: the role of the reference's bo1_GoICP.py (one registration per .tsv row) with the pairs
sharded over the GPUs of one node -- one process per GPU, no data-path collective; a single all_gather of
fixed-size result records at the end."""
import os
import time

import numpy as np

REC = 24   # floats per result record: pair id, R[9], t[3], error, compat, Nd, seconds, dt_seconds, cube_evals, rmsd, status, pad


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


def pack_record(pair_id, res, rmsd=float("nan")):
    r = np.full(REC, np.nan, np.float64)
    r[0] = pair_id; r[1:10] = np.asarray(res["R"]).reshape(9); r[10:13] = res["t"]; r[13] = res["error"]; r[14] = res["compat"]
    r[15] = res["Nd"]; r[16] = res["seconds"]; r[17] = res.get("dt_seconds", np.nan); r[18] = res.get("cube_evals", np.nan); r[19] = rmsd
    r[20] = res.get("status", 0)
    return r


def gather_records(local_records, n_total, dist=None, device=None):
    """all_gather of per-rank record blocks padded to the same length; returns [n_total, REC] ordered by pair id."""
    import torch
    world = dist.get_world_size() if dist is not None and dist.is_initialized() else 1
    per = (n_total + world - 1) // world
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


def run_batch(base_dir, tsv, config, out_dir, register_fn, rank=0, world=1, limit=None, threads=1):
    """register_fn(target_file, source_file, nd, config, output, k) -> result dict. Pairs whose cavity files are
    missing are skipped (the reference tree ships 2 of the 383). threads > 1 keeps that many registrations of this rank in
    flight (one engine + HIP stream each; the C entry points release the GIL); records come back in row order either way."""
    pairs = read_tsv_pairs(tsv)
    if limit:
        pairs = pairs[:limit]
    todo = []
    for idx in shard(len(pairs), rank, world):
        src, tgt = pairs[idx]
        k = idx + 1
        sf = os.path.join("cavities", src + "_cavity6.mol2"); tf = os.path.join("cavities", tgt + "_cavity6.mol2")
        if not (os.path.exists(os.path.join(base_dir, sf)) and os.path.exists(os.path.join(base_dir, tf))):
            continue
        todo.append((k, tf, sf, mol2_atom_count(os.path.join(base_dir, sf))))

    def one(job):
        k, tf, sf, nd = job
        return pack_record(k, register_fn(tf, sf, nd, config, os.path.join(out_dir, "similar%d.txt" % k), k))
    if threads > 1 and len(todo) > 1:
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max_workers=threads) as ex:
            recs = list(ex.map(one, todo))
    else:
        recs = [one(j) for j in todo]
    return recs, len(pairs)
