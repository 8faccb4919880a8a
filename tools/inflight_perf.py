"""End-to-end registrations/s of configs[2]-shaped pairs (5 000 points, 300^3 grid built on the device) with several registrations in flight on one GPU
(GPU box):  python tools/inflight_perf.py [pairs=64] [threads=16]      (try GPU_MAX_HW_QUEUES=8/16 in the environment: hardware queues per process)"""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import goicp_amd
from goicp_amd.synth import synth_pair
import bench
n_pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 64
threads = int(sys.argv[2]) if len(sys.argv) > 2 else 16
cfg = goicp_amd.cfg_vector(MSEThresh=0.001, trimFraction=0.1, regularization=0.0, regularizationFPFH=0.0, cfpfh=0, ponderation=0, norm=2, distTransSize=300)
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
bench.overlapped_registrations(goicp_amd, synth_pair, cfg, 5000, threads, 0, 1, 0, False, None, dev, torch, dt_mode=1, threads=threads, tag="warm")
r = bench.overlapped_registrations(goicp_amd, synth_pair, cfg, 5000, n_pairs, 0, 1, 0, False, None, dev, torch, dt_mode=1, threads=threads, tag="in_flight")
print("GPU_MAX_HW_QUEUES=%s threads %d:" % (os.environ.get("GPU_MAX_HW_QUEUES", "default"), threads), json.dumps(r))
