"""BASELINE configs[3] ("C4"): the cavities_similar_BO1_clean.tsv batch under the c-FPFH weighted error, materialised as files.

The reference tree ships the cavity files of only 2 of the 383 similar pairs, so the batch is built as
BASELINE.md 3 defines it: rows 1-2 are the real pairs (their mol2 / cfpfh / protein files are copied from a directory that holds
them), rows 3..383 are seeded synthetic cavity-like pairs (`synth_pair(corr_features=True)`, N in [150, 400]) written in the same
file formats -- cavities/<id>_cavity6.mol2, cfpfh/<id>_cavity6.cfpfh, chains/<id>_protein.mol2 and
ref_proteins/<src>.<tgt>/aligned_<src>_protein.mol2 (the planted pose applied to the source protein, so that the RMSD of a
registration is defined for every row) -- plus the .tsv that lists them and the config file (shipped defaults + cfpfh=1,
regularizationFPFH=5e-6). Everything downstream (driver.run_batch, the file pipeline of the C ABI) treats real and synthetic
rows alike. No part of the oracle is used here."""
import gzip
import os
import shutil

import numpy as np

from .binding import CFG_KEYS, DEFAULT_CFG
from .synth import synth_pair

N_ROWS = 383
SEED = 20260104
PROP_NAMES = {8204959: "OG", 30894: "N", 15219528: "O", 15231913: "NZ", 4646984: "CZ", 16741671: "CA", 7566712: "DU", 0: "OD1"}
C4_OVERRIDES = dict(cfpfh=1, regularizationFPFH=0.000005)


def write_config(path, **over):
    """config.txt in the reference's key=value syntax (reference config.txt:4-53, ConfigMap.cpp:39-60)."""
    d = dict(DEFAULT_CFG)
    d.update(over)
    ints = ("cfpfh", "norm", "ponderation", "distTransSize")
    with open(path, "w") as f:
        f.write("# generated: shipped defaults + %s\n" % ", ".join("%s=%s" % kv for kv in sorted(over.items())))
        for k in CFG_KEYS:
            f.write("%s=%s\n" % (k, ("%d" % d[k]) if k in ints else repr(float(d[k]))))


def write_cavity_mol2(path, xyz, colours, title="CAVITY"):
    """Cavity mol2 as the reference reads it: atom count = third space-separated token of line 6 (bo1_GoICP.py:45-48),
    atom rows `id name x y z ...` after @<TRIPOS>ATOM (transformation.cpp:282-306); the name column carries the property."""
    n = len(xyz)
    with open(path, "w") as f:
        f.write("#    Name: %s\n#    Modification time : synthetic\n\n@<TRIPOS>MOLECULE\n%s\n%5d     0     6     0     0\nPROTEIN\nNO_CHARGES\n\n\n" % (title, title, n))
        f.write("@<TRIPOS>DICT\nPROTEIN macromol\n@<TRIPOS>ATOM\n")
        for i in range(n):
            f.write("%7d %-8s %9.4f %9.4f %9.4f O.3       1 SER1        0.0000 \n" % (i + 1, PROP_NAMES[int(colours[i])], xyz[i, 0], xyz[i, 1], xyz[i, 2]))


def write_cfpfh(path, rows):
    with open(path, "w") as f:
        for r in rows:
            f.write(" ".join("%.6g" % float(v) for v in r) + "\n")      # (the shipped files print full doubles; the reader takes any float text)


def write_protein_mol2(path, xyz):
    """Backbone-only protein chain (N, CA, C, O per residue): the atoms computeRMSD keeps (transformation.cpp:423-464) in
    the 9-column atom rows applyTransformationProtein rewrites (transformation.cpp:469-539)."""
    names = ("N", "CA", "C", "O"); types = ("N.am", "C.3", "C.2", "O.2")
    n = len(xyz)
    with open(path, "w") as f:
        f.write("#\tName:\t\t\tchain\n\n@<TRIPOS>MOLECULE\nchain\n%5d %5d %5d     0     0\nBIOPOLYMER\nNO_CHARGES\n\n\n@<TRIPOS>DICT\nBIOPOLYMER macromol\n@<TRIPOS>ATOM\n" % (n, n - 1, n // 4))
        for i in range(n):
            f.write("%7d %-8s %10.4f %10.4f %10.4f %-6s %4d ALA%-4d %10.4f BACKBONE\n" % (i + 1, names[i % 4], xyz[i, 0], xyz[i, 1], xyz[i, 2], types[i % 4], i // 4 + 1, i // 4 + 1, 0.0))
        f.write("@<TRIPOS>BOND\n")
        for i in range(n - 1):
            f.write("%6d %5d %5d 1\n" % (i + 1, i + 1, i + 2))


def synth_row(k, seed=SEED):
    """Row k (1-based, k >= 3) of the batch: cavity clouds in Angstrom frames of their own + the planted source->target pose."""
    rng = np.random.default_rng(seed + 7919 * k)
    n = int(rng.integers(150, 401))
    P = synth_pair(n, seed=seed + 100003 * k, noise=0.01, outlier_frac=float(rng.choice([0.0, 0.05, 0.1])), theta_range=(0.3, np.pi), corr_features=True)
    s = float(rng.uniform(9.0, 14.0))                       # Angstrom per normalised unit (shipped cavities: max norm 10-15 A)
    cm = rng.uniform(-40, 60, 3); cd = rng.uniform(-40, 60, 3)
    tgt = np.round(P["t"].astype(np.float64) * s + cm, 4); src = np.round(P["s"].astype(np.float64) * s + cd, 4)
    # model ~= Rt*data + tt (normalised)  =>  target_A ~= Rt*source_A + (cm + s*tt - Rt*cd)
    Rt = P["Rtrue"]; tA = cm + s * P["ttrue"] - Rt @ cd
    prot = np.round(cd + rng.normal(scale=9.0, size=(120, 3)), 4)
    aligned = np.round(prot @ Rt.T + tA, 4)
    return dict(n=n, tgt=tgt, src=src, tc=P["tc"], sc=P["sc"], tf=P["tf"], sf=P["sf"], R=Rt, t=tA, prot=prot, aligned=aligned)


def _copy_maybe_gz(src_dir, name, dst):
    p = os.path.join(src_dir, name)
    if os.path.exists(p):
        shutil.copyfile(p, dst); return True
    if os.path.exists(p + ".gz"):
        with gzip.open(p + ".gz", "rb") as fi, open(dst, "wb") as fo:
            shutil.copyfileobj(fi, fo)
        return True
    return False


def materialise(dst, real_dir, rows=N_ROWS, seed=SEED, config_over=None):
    """Writes the batch under `dst` (cavities/ cfpfh/ cavitiesN/ chains/ ref_proteins/ rot/ output/ + c4_batch.tsv + config_c4.txt).
    `real_dir` holds the shipped files of rows 1-2 (cavities/, cfpfh/, proteins/ and the reference's .tsv).
    Returns (tsv path, config path, list of planted poses per row or None for the real rows)."""
    for d in ("cavities", "cfpfh", "cavitiesN", "chains", "ref_proteins", "rot", "output"):
        os.makedirs(os.path.join(dst, d), exist_ok=True)
    real_rows = []
    with open(os.path.join(real_dir, "cavities_similar_BO1_clean.tsv")) as f:
        for line in f:
            if len(line.split()) >= 4:
                real_rows.append(line.rstrip("\n"))
            if len(real_rows) == 2:
                break
    lines = []; truth = []
    for k, line in enumerate(real_rows[:min(2, rows)], start=1):
        t = line.split(); src, tgt = t[2], t[3]
        for cid in (src, tgt):
            shutil.copyfile(os.path.join(real_dir, "cavities", cid + "_cavity6.mol2"), os.path.join(dst, "cavities", cid + "_cavity6.mol2"))
            shutil.copyfile(os.path.join(real_dir, "cfpfh", cid + "_cavity6.cfpfh"), os.path.join(dst, "cfpfh", cid + "_cavity6.cfpfh"))
        pd = os.path.join(real_dir, "proteins")
        if _copy_maybe_gz(pd, src + "_protein.mol2", os.path.join(dst, "chains", src + "_protein.mol2")):
            os.makedirs(os.path.join(dst, "ref_proteins", "%s.%s" % (src, tgt)), exist_ok=True)
            _copy_maybe_gz(pd, "aligned_%s_protein.mol2" % src, os.path.join(dst, "ref_proteins", "%s.%s" % (src, tgt), "aligned_%s_protein.mol2" % src))
        lines.append(line); truth.append(None)
    for k in range(3, rows + 1):
        r = synth_row(k, seed)
        src, tgt = "s%04da_1" % k, "s%04db_1" % k
        write_cavity_mol2(os.path.join(dst, "cavities", src + "_cavity6.mol2"), r["src"], r["sc"])
        write_cavity_mol2(os.path.join(dst, "cavities", tgt + "_cavity6.mol2"), r["tgt"], r["tc"])
        write_cfpfh(os.path.join(dst, "cfpfh", src + "_cavity6.cfpfh"), r["sf"])
        write_cfpfh(os.path.join(dst, "cfpfh", tgt + "_cavity6.cfpfh"), r["tf"])
        write_protein_mol2(os.path.join(dst, "chains", src + "_protein.mol2"), r["prot"])
        os.makedirs(os.path.join(dst, "ref_proteins", "%s.%s" % (src, tgt)), exist_ok=True)
        write_protein_mol2(os.path.join(dst, "ref_proteins", "%s.%s" % (src, tgt), "aligned_%s_protein.mol2" % src), r["aligned"])
        lines.append("SYN%04dA\tSYN%04dB\t%s\t%s\t1.0\tsynthetic\t%d" % (k, k, src, tgt, k)); truth.append((r["R"], r["t"]))
    tsv = os.path.join(dst, "c4_batch.tsv")
    with open(tsv, "w") as f:
        f.write("\n".join(lines) + "\n")
    cfgp = os.path.join(dst, "config_c4.txt")
    write_config(cfgp, **(C4_OVERRIDES if config_over is None else config_over))
    return tsv, cfgp, truth
