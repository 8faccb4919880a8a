import os, sys, time
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import numpy as np
import goicp_amd
from oracle.refbind import cfg_vector
from helpers import synth_pair, oracle_for
P = synth_pair(1000, seed=20260101, theta_range=(2.2, 3.1))
v,_ = cfg_vector(distTransSize=100, trimFraction=0.1, regularization=0.0, ponderation=0, MSEThresh=0.001)
o = oracle_for(P, v); o.initialize()
dist, near3, cc, _ = o.dt_dump()
e = goicp_amd.Engine(S=100, geom4=(o.geom[0], o.geom[2], o.geom[4], o.geom[6]), dist=dist, data_xyz=P["s"], weights=o.weights(),
                     maxrotdis=o.maxrotdis(), model_xyz=P["t"], inlier_num=o.inlier_num(), do_trim=True, norm=2, parity_mode=1)
rng = np.random.default_rng(3)
sse = np.float32(o.sse_thresh())
for it in range(12):
    aa = (rng.uniform(-np.pi, np.pi, 3)).astype(np.float32)
    R = o.axis_angle(aa); o.rotate(R)
    lvl = int(rng.integers(-1, 4)); E = np.float32(rng.uniform(5, 40))
    o.set_opt_error(E); o.counters()
    val, node = o.inner_bnb(lvl); c = o.counters()
    r = e.inner_bnb(R=R, level=lvl, E=E, sse=sse, init=(-0.5,-0.5,-0.5,1.0))
    print(it, lvl, E, 'oracle', val, c[0], 'gpu', r['val'], r['pops'], r['status'], flush=True)
