"""Seeded synthetic inputs: cavity-like shell clouds for the benchmark and the tests.
No part of the oracle is used here."""
import numpy as np


def rot_cloud(R, d):
    """pDataTemp = R*p in float32, left-to-right, no FMA (reference jly_goicp.cpp:750-756)."""
    R = np.asarray(R, np.float32); d = np.asarray(d, np.float32)
    return np.stack([(R[k, 0] * d[:, 0] + R[k, 1] * d[:, 1]) + R[k, 2] * d[:, 2] for k in range(3)], 1).astype(np.float32)


def synth_pair(n, seed=20260101, noise=0.01, outlier_frac=0.10, tmax=0.2, theta_range=(0.3, np.pi), corr_features=False):
    """SURVEY.md 8(d) synthetic generator: shell cloud on 6 overlapping spheres, centred, max-norm 1,
    6 significant digits; data = permuted, rotated, translated, noised copy with outliers."""
    rng = np.random.default_rng(seed)
    centres = rng.uniform(-0.45, 0.45, (6, 3)); radii = rng.uniform(0.35, 0.6, 6)
    which = rng.integers(0, 6, n)
    v = rng.normal(size=(n, 3)); v /= np.linalg.norm(v, axis=1, keepdims=True)
    m = centres[which] + radii[which, None] * v
    m -= m.mean(0); m /= np.linalg.norm(m, axis=1).max()
    m = np.array([[float("%g" % x) for x in row] for row in m])
    rng2 = np.random.default_rng(seed + 1)
    aa = rng2.normal(size=3); aa = aa / np.linalg.norm(aa) * rng2.uniform(*theta_range)
    th = np.linalg.norm(aa); k = aa / th
    K = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    Rt = np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * K @ K
    tt = rng2.uniform(-tmax, tmax, 3)
    perm = rng2.permutation(n)
    d = m[perm]
    # data such that  Rt*d + tt ~= model  <=> d = Rt^T (m - tt)
    d = (d - tt) @ Rt + rng2.normal(scale=noise, size=(n, 3))
    nout = int(outlier_frac * n)
    out_idx = rng2.choice(n, nout, replace=False) if nout else np.zeros(0, int)
    if nout:
        d[out_idx] = rng2.uniform(-1, 1, (nout, 3))
    d = np.array([[float("%g" % x) for x in row] for row in d])
    props = np.array([8204959, 30894, 15219528, 15231913, 4646984, 16741671, 7566712, 0], np.int32)
    mc = props[rng.integers(0, 8, n)]; dc = props[rng2.integers(0, 8, n)]

    def fp(r, k):
        a = r.uniform(0, 1, (k, 33)); a = a / a.sum(1, keepdims=True) * 600
        b = r.uniform(0, 1, (k, 8)); b = b / b.sum(1, keepdims=True) * 200
        return np.concatenate([a, b], 1).astype(np.float32)
    mf = fp(rng, n); df = fp(rng2, n)
    if corr_features:
        # corresponding points carry the model point's colour and (slightly perturbed) descriptor, as real cavity pairs do
        inl = np.ones(n, bool); inl[out_idx] = False
        dc = np.where(inl, mc[perm], dc)
        df = np.where(inl[:, None], (mf[perm] * (1 + rng2.normal(scale=0.02, size=(n, 41)))).astype(np.float32), df)
    return dict(t=m.astype(np.float32), s=d.astype(np.float32), tc=mc, sc=dc, tf=mf, sf=df, Rtrue=Rt, ttrue=tt)
