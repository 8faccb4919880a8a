"""CPU (-m "not gpu"): host-side pieces of the product library (no GPU needed) against the oracle / goldens:
the DT builder + cell tables, Initialize() tables, config parsing, mol2 -> normalised xyz text, RMSD."""
import os

import numpy as np
import pytest

import goicp_amd
from oracle.refbind import cfg_vector, Oracle
from helpers import GOLDEN, load_clouds, load_hotpath, oracle_for

DATA = os.path.join(GOLDEN, "data")


@pytest.mark.parametrize("name,S", [("pair1", 20), ("pair1", 50), ("pair2", 50), ("pair1", 100)])
def test_host_dt_builder_matches_oracle(name, S):
    P = load_clouds(name)
    v, _ = cfg_vector(distTransSize=S)
    o = oracle_for(P, v)
    od, on, oc, _ = o.dt_dump()
    g = goicp_amd.GoICP(v)
    g.set_clouds(P["t"], P["s"], P["tc"], P["sc"], P["tf"], P["sf"])
    g.build_dt()
    dist, near, cc = g.dt_dump()
    assert np.array_equal(g.geom, o.geom)
    assert np.array_equal(dist, od)
    assert np.array_equal(near, (on[:, 2].astype(np.int64) * S + on[:, 1]) * S + on[:, 0])
    assert np.array_equal(cc, oc)


def test_host_dt_matches_reference_fixture():
    H = load_hotpath("pair2_S50"); P = load_clouds("pair2")
    g = goicp_amd.GoICP(H["cfg"]); g.set_clouds(P["t"], P["s"], P["tc"], P["sc"]); g.build_dt()
    dist, near, cc = g.dt_dump()
    assert np.array_equal(dist, H["dist"]) and np.array_equal(cc, H["cellc"])


def test_config_file_roundtrip(tmp_path):
    p = os.path.join(str(tmp_path), "config.txt")
    with open(p, "w") as f:
        f.write("# comment line\nMSEThresh=0.01\n\nnorm=2\nregularization = 0.0005;\nponderation=1\r\ncfpfh=0\nrotMinX=-3.1416\nrotMinY=-3.1416\nrotMinZ=-3.1416\n"
                "rotWidth=6.2832\ntransMinX=-0.5\ntransMinY=-0.5\ntransMinZ=-0.5\ntransWidth=1.0\ntrimFraction=0.0\ndistTransSize=20\n"
                "distTransExpandFactor=2.0\nthis line has three tokens\n")
    g = goicp_amd.GoICP()
    g.read_config(p)
    v = g.get_config()
    want, _ = cfg_vector()
    assert np.array_equal(v, want)        # missing keys (regularizationNeighbors, regularizationFPFH) parse as atof("") = 0


def test_rmsd_known_answers():
    # Transformation::computeRMSD on the shipped aligned/rot proteins -> 1.73675334 / 13.9296408.
    # The protein files (3.4 MB) are not committed; the function is exercised on a synthetic pair of mol2 files.
    import tempfile
    d = tempfile.mkdtemp()
    def write(path, pts):
        with open(path, "w") as f:
            f.write("@<TRIPOS>MOLECULE\nx\n@<TRIPOS>ATOM\n")
            for i, (nm, p) in enumerate(pts):
                f.write("%7d %-4s %10.4f %10.4f %10.4f C.3 1 ALA1 0.0000\n" % (i + 1, nm, p[0], p[1], p[2]))
            f.write("@<TRIPOS>BOND\n")
    a = [("N", (0, 0, 0)), ("CA", (1, 0, 0)), ("CB", (5, 5, 5)), ("C", (2, 0, 0)), ("O", (3, 0, 0))]
    b = [("N", (0, 0, 1)), ("CA", (1, 0, 1)), ("CB", (9, 9, 9)), ("C", (2, 2, 0)), ("O", (3, 0, 0))]
    write(os.path.join(d, "a.mol2"), a); write(os.path.join(d, "b.mol2"), b)
    got = goicp_amd.compute_rmsd(os.path.join(d, "a.mol2"), os.path.join(d, "b.mol2"))
    want = np.float32(np.sqrt(np.float32(1 + 1 + 4 + 0) / 4))     # backbone atoms N, CA, C, O only (CB skipped)
    assert abs(got - want) < 1e-6


def _gunzip(name, dst):
    import gzip, shutil
    with gzip.open(os.path.join(DATA, "proteins", name + ".gz"), "rb") as fi, open(dst, "wb") as fo:
        shutil.copyfileobj(fi, fo)
    return dst


def test_rmsd_known_answers_on_shipped_proteins(tmp_path):
    # Transformation::computeRMSD (transformation.cpp:453-464) on the shipped aligned / transformed proteins:
    # SURVEY.md 8(c) known answers 1.73675334 (pair 1) and 13.9296408 (pair 2), float accumulation
    d = str(tmp_path)
    a1 = _gunzip("aligned_2x86_3_protein.mol2", os.path.join(d, "a1.mol2")); r1 = _gunzip("rot_2x86_3_protein.mol2", os.path.join(d, "r1.mol2"))
    a2 = _gunzip("aligned_2ktd_1_protein.mol2", os.path.join(d, "a2.mol2")); r2 = _gunzip("rot_2ktd_1_protein.mol2", os.path.join(d, "r2.mol2"))
    assert np.float32(goicp_amd.compute_rmsd(a1, r1)) == np.float32(1.73675334)
    assert np.float32(goicp_amd.compute_rmsd(a2, r2)) == np.float32(13.9296408)


def test_protein_transform_reproduces_shipped_file(tmp_path):
    # Transformation::applyTransformationProtein (transformation.cpp:469-539) with the shipped output/similar1_rescaled.txt
    # on chains/2x86_3_protein.mol2 -> rot/rot_2x86_3_protein.mol2, byte for byte
    d = str(tmp_path)
    chain = _gunzip("2x86_3_protein.mol2", os.path.join(d, "chain.mol2")); want = _gunzip("rot_2x86_3_protein.mol2", os.path.join(d, "want.mol2"))
    out = os.path.join(d, "got.mol2")
    assert goicp_amd.apply_transformation_protein(os.path.join(DATA, "output", "similar1_rescaled.txt"), chain, out)
    assert open(out).read() == open(want).read()


def test_parallel_partition_formulation_equals_the_sequential_loop(tmp_path):
    """wave_intro_select (csrc/engine_dev.hpp) computes the reference's partition (jly_sorting.hpp:282-303) from ranks instead of a
    cell-by-cell scan; tests/csrc/partition_check.c checks that formulation against the sequential loop on random rows."""
    import subprocess
    exe = str(tmp_path / "partition_check")
    subprocess.check_call(["gcc", "-O2", "-o", exe, os.path.join(os.path.dirname(__file__), "csrc", "partition_check.c")])
    out = subprocess.run([exe, "300000"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and out.stdout.strip() == "bad=0", out.stdout


def test_wave_wide_queue_pop_equals_the_sequential_heap(tmp_path):
    """heap_pop_wave (csrc/engine_dev.hpp) resolves a pop's path from ballots over sibling comparisons (six levels per round trip to the
    queue) and lands the displaced element with one more ballot; tests/csrc/heap_pop_model.c restates it lane by lane and checks the
    queue array against the sequential pop_heap order (reference jly_goicp.h:75-87 through std::priority_queue)."""
    import subprocess
    exe = str(tmp_path / "heap_pop_model")
    subprocess.check_call(["gcc", "-O2", "-o", exe, os.path.join(os.path.dirname(__file__), "csrc", "heap_pop_model.c")])
    out = subprocess.run([exe, "200000"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and out.stdout.strip() == "bad=0", out.stdout


@pytest.mark.parametrize("env", [{}, {"GOICP_HI_SHARE": "4"}, {"GOICP_CACHE_GB": "0"}, {"GOICP_CACHE_GB": "0.01"}],
                         ids=["default", "shared_hi_streams", "cache_off", "cache_10MB"])
def test_shared_pool_under_thread_sanitizer(tmp_path, env):
    """Everything a batch's worker threads share inside libgoicp_hip.so is csrc/host_pool.cpp (device-block cache, pooled streams, events
    and pinned buffers, auxiliary streams, polling waits, the once-per-instantiation dynamic-LDS ceiling). tests/csrc/pool_tsan.cpp builds
    that file as it ships against a stub HIP runtime under ThreadSanitizer and runs eight threads through the create / destroy sequence
    of 40 engines each, one of them emptying the pool meanwhile: no data race, no object with two owners, no leak, one attribute call
    per kernel instantiation."""
    import subprocess
    here = os.path.dirname(__file__)
    root = os.path.dirname(here)
    csrc = os.path.join(root, "go-icp-protein-cavities_amd", "csrc")
    rocm_inc = "/opt/rocm/include"
    if not os.path.exists(os.path.join(rocm_inc, "hip", "hip_runtime_api.h")):
        pytest.skip("HIP headers not installed")
    exe = str(tmp_path / "pool_tsan")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=thread", "-D__HIP_PLATFORM_AMD__", "-I" + rocm_inc,
                           "-I" + os.path.join(root, "include"), "-I" + csrc, os.path.join(here, "csrc", "pool_tsan.cpp"),
                           os.path.join(csrc, "host_pool.cpp"), "-o", exe, "-pthread"])
    e = dict(os.environ); e.pop("GOICP_HI_SHARE", None); e.pop("GOICP_CACHE_GB", None); e.update(env)
    out = subprocess.run([exe, "8", "40"], capture_output=True, text=True, timeout=300, env=e)
    assert out.returncode == 0 and "ThreadSanitizer" not in out.stderr and "errors 0" in out.stdout, out.stdout + out.stderr
