#!/usr/bin/env python3
"""Generates the committed known-answer fixtures under tests/golden/ from the COMPILED REFERENCE
(oracle/_ref/libgoicp_ref.so, built by oracle/Makefile from /root/reference where it lies).

Run in the build container only (needs /root/reference):   python tests/golden/make_golden.py
Outputs are data only: inputs (the reference's shipped cavity / c-FPFH / demo data files, normalised clouds)
and expected outputs (DT tables, (cube)->(ub,lb) tables, InnerBnB / ICP / Register results).
"""
import json
import os
import shutil
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle.refbind import (Ref, cfg_vector, read_mol2, normalise_pair, read_cfpfh, read_txt_cloud,  # noqa: E402
                            error_trace, CFG_KEYS)

REF = "/root/reference"
DATA = os.path.join(HERE, "data")


def copy_data():
    for sub, pat in (("cavities", ".mol2"), ("cfpfh", ".cfpfh"), ("cavitiesN", ".xyz"), ("output", ".txt")):
        os.makedirs(os.path.join(DATA, sub), exist_ok=True)
        for f in sorted(os.listdir(os.path.join(REF, sub))):
            if f.endswith(pat):
                shutil.copyfile(os.path.join(REF, sub, f), os.path.join(DATA, sub, f))
    shutil.copyfile(os.path.join(REF, "cavities_similar_BO1_clean.tsv"), os.path.join(DATA, "cavities_similar_BO1_clean.tsv"))
    os.makedirs(os.path.join(DATA, "demo"), exist_ok=True)
    shutil.copyfile(os.path.join(REF, "demo", "output.txt"), os.path.join(DATA, "demo", "output.txt"))
    for f in ("model_rand.txt", "data_rand.txt"):
        shutil.copyfile(os.path.join(REF, "demo", f), os.path.join(DATA, "demo", f))
    # post-processing known answers: protein chain, the reference's transformed copy and the aligned reference
    import gzip
    os.makedirs(os.path.join(DATA, "proteins"), exist_ok=True)
    for src in ("chains/2x86_3_protein.mol2", "rot/rot_2x86_3_protein.mol2", "ref_proteins/2x86_3.1eq2_6/aligned_2x86_3_protein.mol2",
                "rot/rot_2ktd_1_protein.mol2", "ref_proteins/2ktd_1.4imo_2/aligned_2ktd_1_protein.mol2"):
        with open(os.path.join(REF, src), "rb") as fi, gzip.open(os.path.join(DATA, "proteins", os.path.basename(src) + ".gz"), "wb", compresslevel=9) as fo:
            fo.write(fi.read())
    # bunny: model (35947 pts) + first 1000 data points, as float32 arrays (the .txt files are 3.7 MB)
    mb = read_txt_cloud(os.path.join(REF, "demo", "model_bunny.txt"))
    db = read_txt_cloud(os.path.join(REF, "demo", "data_bunny.txt"), limit=1000)
    np.savez_compressed(os.path.join(DATA, "demo", "bunny.npz"), model=mb, data=db)


def load_pair(src, tgt):
    sx, sc = read_mol2(os.path.join(REF, "cavities", src + "_cavity6.mol2"))
    tx, tc = read_mol2(os.path.join(REF, "cavities", tgt + "_cavity6.mol2"))
    s, t, scale, sm, tm = normalise_pair(sx, tx)
    sf = read_cfpfh(os.path.join(REF, "cfpfh", src + "_cavity6.cfpfh"), len(s))
    tf = read_cfpfh(os.path.join(REF, "cfpfh", tgt + "_cavity6.cfpfh"), len(t))
    return dict(s=s, sc=sc, sf=sf, t=t, tc=tc, tf=tf, scale=scale, smean=sm, tmean=tm)


def rot_cloud(R, d):
    R = R.astype(np.float32); d = d.astype(np.float32)
    return np.stack([(R[k, 0] * d[:, 0] + R[k, 1] * d[:, 1]) + R[k, 2] * d[:, 2] for k in range(3)], 1).astype(np.float32)


def axis_angle_f32(v):
    """jly_goicp.cpp:729-747 in numpy float32 (cos/sin via float64 then rounded: fixtures store R, never recompute it)."""
    v = np.asarray(v, np.float32)
    t = np.float32(np.sqrt(np.float32(v[0] * v[0] + v[1] * v[1]) + v[2] * v[2]))
    if t == 0:
        return np.eye(3, dtype=np.float32)
    v = (v / t).astype(np.float32)
    ct = np.float32(np.cos(np.float64(t))); st = np.float32(np.sin(np.float64(t))); ct2 = np.float32(1 - ct)
    R = np.empty((3, 3), np.float32)
    t121 = v[0] * v[1] * ct2; t122 = v[2] * st; t131 = v[0] * v[2] * ct2; t132 = v[1] * st; t231 = v[1] * v[2] * ct2; t232 = v[0] * st
    R[0] = [ct + v[0] * v[0] * ct2, t121 - t122, t131 + t132]
    R[1] = [t121 + t122, ct + v[1] * v[1] * ct2, t231 - t232]
    R[2] = [t131 - t132, t231 + t232, ct + v[2] * v[2] * ct2]
    return R


def new_ref(P, **over):
    v, d = cfg_vector(**over)
    r = Ref(); r.set_config(v)
    r.set_clouds(P["t"], P["s"], P["tc"], P["sc"], P["tf"], P["sf"])
    r.build_dt()
    r.counters(reset=True)
    return r, v


def gen_hotpath(P, name, n_calls, seed, **over):
    """InnerBnB known answers + the (cube)->(ub,lb) pairs the reference pushed while producing them."""
    rng = np.random.default_rng(seed)
    r, v = new_ref(P, **over)
    r.initialize()
    dist, near, cc, _ = r.dt_dump()
    ptr, idx = r.cell_csr(cc)
    calls = []
    cubes = []
    corners = []
    for k in range(n_calls):
        R = axis_angle_f32(rng.uniform(-np.pi, np.pi, 3) * rng.uniform(0.1, 1.0))
        rc = rot_cloud(R, P["s"][:r.Nd])
        r.set_rotated(rc)
        level = int(rng.integers(-1, 6))
        E = np.float32(rng.uniform(8, 60))
        r.set_opt_error(E)
        r.trace_begin(1 << 16)
        val, node = r.inner_bnb(level)
        tr = r.trace_end()[1:]            # [0] is the root push
        calls.append(dict(R=R, level=level, E=E, val=val, node=node[:4].copy()))
        keep = tr[:: max(1, len(tr) // 64)][:64]
        for row in keep:
            cubes.append(np.concatenate([[k], row]))
        for _ in range(4):
            xyz = (rng.integers(-16, 17, 3) / np.float32(32.0)).astype(np.float32)
            cnt = r.check_compatibilities(*xyz)
            fp = r.sum_fpfh(*xyz) if over.get("cfpfh", 0) else 0.0
            corners.append([k, xyz[0], xyz[1], xyz[2], cnt, fp])
    out = dict(cfg=v, weights=r.weights(), maxrotdis=r.maxrotdis(), inlierNum=r.inlier_num(), sse=r.sse_thresh(),
               geom=r.geom, dist=dist, nearest3=near, cellc=cc, cell_ptr=ptr, cell_idx=idx,
               call_R=np.stack([c["R"] for c in calls]), call_level=np.array([c["level"] for c in calls], np.int32),
               call_E=np.array([c["E"] for c in calls], np.float32), call_val=np.array([c["val"] for c in calls], np.float32),
               call_node=np.stack([c["node"] for c in calls]).astype(np.float32),
               cubes=np.array(cubes, np.float32), corners=np.array(corners, np.float64))
    np.savez_compressed(os.path.join(HERE, "hotpath_%s.npz" % name), **out)
    print("hotpath", name, "calls", n_calls, "cube rows", len(cubes))


def gen_icp(P, name, seed, **over):
    rng = np.random.default_rng(seed)
    r, v = new_ref(P, **over)
    r.initialize()
    rows = []
    for k in range(6):
        R = axis_angle_f32(rng.uniform(-0.6, 0.6, 3)).astype(np.float64)
        t = rng.uniform(-0.1, 0.1, 3)
        e, Ro, to, corr, dis = r.icp(R, t)
        rows.append(dict(Rin=R, tin=t, err=e, Rout=Ro, tout=to, corr=corr, dis=dis))
    np.savez_compressed(os.path.join(HERE, "icp_%s.npz" % name), cfg=v,
                        Rin=np.stack([x["Rin"] for x in rows]), tin=np.stack([x["tin"] for x in rows]),
                        err=np.array([x["err"] for x in rows], np.float32), Rout=np.stack([x["Rout"] for x in rows]),
                        tout=np.stack([x["tout"] for x in rows]), corr=np.stack([x["corr"] for x in rows]),
                        dis=np.stack([x["dis"] for x in rows]))
    print("icp", name)


def reg_record(r, over):
    out = r.register()
    tp, rp = r.counters()
    return dict(cfg=over, R=out["R"].tolist(), t=out["t"].tolist(), error=float(out["error"]), comp=out["comp"],
                Nd=out["Nd"], trace=error_trace(out["log"]), trans_pops=tp)


def main():
    os.makedirs(DATA, exist_ok=True)
    copy_data()
    P1 = load_pair("2x86_3", "1eq2_6")
    P2 = load_pair("2ktd_1", "4imo_2")
    for nm, P in (("pair1", P1), ("pair2", P2)):
        np.savez_compressed(os.path.join(HERE, nm + "_clouds.npz"), **{k: np.asarray(v) for k, v in P.items()})

    gen_hotpath(P1, "pair1_default", 24, 11)
    gen_hotpath(P1, "pair1_trim", 16, 12, trimFraction=0.1, regularization=0.0)
    gen_hotpath(P1, "pair1_fpfh", 12, 13, cfpfh=1, regularizationFPFH=0.000005)
    gen_hotpath(P1, "pair1_norm1", 8, 14, norm=1, regularization=0.0, ponderation=0)
    gen_hotpath(P2, "pair2_S50", 12, 15, distTransSize=50)
    gen_icp(P1, "pair1_default", 21)
    gen_icp(P1, "pair1_trim", 22, trimFraction=0.1)
    gen_icp(P1, "pair1_fpfh", 23, cfpfh=1, regularizationFPFH=0.000005)

    regs = {}
    variants = {
        "pair1_default": (P1, {}),
        "pair1_fpfh": (P1, dict(cfpfh=1, regularizationFPFH=0.000005)),
        "pair1_fpfh33": (P1, dict(cfpfh=2, regularizationFPFH=0.000005)),
        "pair1_color8": (P1, dict(cfpfh=3, regularizationFPFH=0.00005, regularization=0.0)),
        "pair1_norm1": (P1, dict(norm=1)),
        "pair1_trim": (P1, dict(trimFraction=0.1, MSEThresh=0.02)),
        "pair1_S50": (P1, dict(distTransSize=50, MSEThresh=0.02)),
        "pair1_S100_geo": (P1, dict(distTransSize=100, MSEThresh=0.05, regularization=0.0, ponderation=0)),
        "pair2_default": (P2, {}),
        "pair1_S300": (P1, dict(distTransSize=300)),          # BASELINE configs[1]: cavity pair 1 on a 300^3 DT
    }
    for nm, (P, over) in variants.items():
        r, v = new_ref(P, **over)
        regs[nm] = reg_record(r, over)
        print("register", nm, regs[nm]["error"], regs[nm]["Nd"] - regs[nm]["comp"], regs[nm]["trans_pops"])

    # upstream demo clouds (3-column txt, no colours): geometric path only
    geo = dict(regularization=0.0, ponderation=0, norm=2, cfpfh=0, regularizationFPFH=0.0, rotMinX=-3.1416, rotMinY=-3.1416,
               rotMinZ=-3.1416, rotWidth=6.2832)
    mr = read_txt_cloud(os.path.join(REF, "demo", "model_rand.txt")); dr = read_txt_cloud(os.path.join(REF, "demo", "data_rand.txt"))
    for nm, over in (("rand_untrimmed", dict(MSEThresh=0.001, trimFraction=0.0, distTransSize=100, **geo)),
                     ("rand_trim", dict(MSEThresh=0.001, trimFraction=0.1, distTransSize=100, **geo))):
        v, _ = cfg_vector(**over)
        r = Ref(); r.set_config(v); r.set_clouds(mr, dr); r.build_dt()
        regs[nm] = reg_record(r, over)
        print("register", nm, regs[nm]["error"], regs[nm]["trans_pops"])
    b = np.load(os.path.join(DATA, "demo", "bunny.npz"))
    over = dict(MSEThresh=0.001, trimFraction=0.0, distTransSize=100, **geo)
    v, _ = cfg_vector(**over)
    r = Ref(); r.set_config(v); r.set_clouds(b["model"], b["data"]); r.build_dt()
    regs["bunny_S100"] = reg_record(r, over)
    print("register bunny_S100", regs["bunny_S100"]["error"], regs["bunny_S100"]["trans_pops"])
    if "--bunny300" in sys.argv:
        over = dict(MSEThresh=0.001, trimFraction=0.0, distTransSize=300, **geo)
        v, _ = cfg_vector(**over)
        r = Ref(); r.set_config(v); r.set_clouds(b["model"], b["data"]); r.build_dt()
        regs["bunny_S300"] = reg_record(r, over)
        print("register bunny_S300", regs["bunny_S300"]["error"], regs["bunny_S300"]["R"], regs["bunny_S300"]["t"])
    else:
        old = os.path.join(HERE, "register.json")
        if os.path.exists(old):
            prev = json.load(open(old))
            if "bunny_S300" in prev:
                regs["bunny_S300"] = prev["bunny_S300"]
    json.dump(regs, open(os.path.join(HERE, "register.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
