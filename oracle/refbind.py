"""TEST INFRASTRUCTURE ONLY (never imported by the product path).

ctypes bindings for
  * oracle/_ref/libgoicp_ref.so  -- the unmodified reference compiled in place (oracle/ref_shim.cpp), and
  * oracle/libgoicp_oracle.so    -- the plain-C restatement (oracle/goicp_oracle.c),
plus the small amount of file plumbing the reference's main() does around them
(jly_main.cpp:67-99,272-314; transformation.cpp:282-361), restated in numpy for the tests.
"""
import ctypes as C
import os
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF_SO = os.path.join(HERE, "_ref", "libgoicp_ref.so")
ORACLE_SO = os.path.join(HERE, "libgoicp_oracle.so")

# config vector order shared by ref_set_config / orc_set_config (jly_main.cpp:236-264)
CFG_KEYS = ["MSEThresh", "rotMinX", "rotMinY", "rotMinZ", "rotWidth", "transMinX", "transMinY", "transMinZ",
            "transWidth", "trimFraction", "regularization", "regularizationNeighbors", "regularizationFPFH",
            "cfpfh", "norm", "ponderation", "distTransSize", "distTransExpandFactor"]

# shipped config.txt:4-53
DEFAULT_CFG = dict(MSEThresh=0.01, rotMinX=-3.1416, rotMinY=-3.1416, rotMinZ=-3.1416, rotWidth=6.2832,
                   transMinX=-0.5, transMinY=-0.5, transMinZ=-0.5, transWidth=1.0, trimFraction=0.0,
                   regularization=0.0005, regularizationNeighbors=0.0, regularizationFPFH=0.0, cfpfh=0, norm=2,
                   ponderation=1, distTransSize=20, distTransExpandFactor=2.0)

# transformation.hpp:36
PROPS = dict(OG=8204959, N=30894, O=15219528, NZ=15231913, CZ=4646984, CA=16741671, DU=7566712, OD1=0, C=1)


def cfg_vector(**over):
    d = dict(DEFAULT_CFG)
    for k, v in over.items():
        if k not in d:
            raise KeyError(k)
        d[k] = v
    # the reference parses every value with atof() into a float field
    return np.array([float(d[k]) for k in CFG_KEYS], dtype=np.float64), d


def read_config(path):
    """ConfigMap semantics (ConfigMap.cpp:17-60): '#' comment lines, split on ' =;', exactly two tokens."""
    d = {}
    with open(path) as f:
        for line in f:
            line = line.rstrip("\n").rstrip("\r")
            if line.startswith("#"):
                continue
            toks = [t for t in line.replace("=", " ").replace(";", " ").split(" ") if t]
            if len(toks) == 2:
                d[toks[0]] = toks[1]
    out = {}
    for k in CFG_KEYS:
        v = d.get(k, "")
        try:
            out[k] = float(v)
        except ValueError:
            out[k] = 0.0
    return out


def read_mol2(path):
    """xyz (float64) + colour codes of the @<TRIPOS>ATOM block (transformation.cpp:282-306)."""
    xyz, col = [], []
    atom = False
    with open(path) as f:
        for line in f:
            if not atom:
                if "@<TRIPOS>ATOM" in line:
                    atom = True
                continue
            if line.startswith("@<TRIPOS>"):
                break
            t = line.split()
            if len(t) < 5:
                break
            xyz.append([float(t[2]), float(t[3]), float(t[4])])
            col.append(PROPS.get(t[1], PROPS["OG"]))
    return np.array(xyz, dtype=np.float64), np.array(col, dtype=np.int32)


def _sig6(a):
    """Text round trip of writeNormalizedMolCloudFile (default ostream precision = %g) then `>> float`."""
    flat = np.asarray(a, dtype=np.float64).ravel()
    return np.array([np.float32(float("%g" % v)) for v in flat], dtype=np.float32).reshape(np.shape(a))


def normalise_pair(src_xyz, tgt_xyz):
    """jly_main.cpp:75-93: centre each cloud, divide both by the larger max-norm, 6-digit text round trip."""
    def centre(c):
        n = c.shape[0]
        mean = np.array([np.sum(c[:, k]) for k in range(3)]) / n   # sequential double sums
        # NB: sequential accumulation; numpy's pairwise sum differs in the last ulp, so redo it exactly
        m = [0.0, 0.0, 0.0]
        for p in c:
            m[0] += p[0]; m[1] += p[1]; m[2] += p[2]
        mean = np.array(m) / n
        cc = c - mean
        nrm = np.sqrt(cc[:, 0] ** 2 + cc[:, 1] ** 2 + cc[:, 2] ** 2)
        return cc, mean, float(nrm.max())
    s, smean, sn = centre(src_xyz)
    t, tmean, tn = centre(tgt_xyz)
    scale = sn if sn >= tn else tn
    return _sig6(s / scale), _sig6(t / scale), scale, smean, tmean


def read_cfpfh(path, n):
    v = np.loadtxt(path, dtype=np.float64).astype(np.float32).reshape(-1, 41)
    return np.ascontiguousarray(v[:n])


def read_txt_cloud(path, limit=None):
    """Go-ICP demo format: first line N, then 'x y z' per line."""
    with open(path) as f:
        n = int(f.readline().split()[0])
        if limit is not None:
            n = min(n, limit)
        pts = np.loadtxt(f, dtype=np.float64, max_rows=n)
    return pts.astype(np.float32).reshape(-1, 3)


def _p(a, t):
    return None if a is None else a.ctypes.data_as(C.POINTER(t))


class Ref:
    """The compiled reference (GoICP class, jly_goicp.h:112-218) behind oracle/ref_shim.cpp."""

    def __init__(self):
        self.lib = L = C.CDLL(REF_SO)
        L.ref_new.restype = C.c_void_p
        L.ref_inner_bnb.restype = C.c_float
        L.ref_icp.restype = C.c_float
        L.ref_register.restype = C.c_float
        L.ref_sse_thresh.restype = C.c_float
        L.ref_get_opt_error.restype = C.c_float
        L.ref_sum_fpfh.restype = C.c_float
        L.ref_trace_end.restype = C.c_longlong
        self.h = C.c_void_p(L.ref_new())
        self.Nd = self.Nm = 0

    def close(self):
        if self.h:
            self.lib.ref_free(self.h)
            self.h = None

    def set_config(self, cfgvec):
        self.cfgvec = np.ascontiguousarray(cfgvec, dtype=np.float64)
        self.lib.ref_set_config(self.h, _p(self.cfgvec, C.c_double))

    def set_clouds(self, mxyz, dxyz, mcol=None, dcol=None, mfpfh=None, dfpfh=None):
        mxyz = np.ascontiguousarray(mxyz, dtype=np.float32); dxyz = np.ascontiguousarray(dxyz, dtype=np.float32)
        self.Nm, self.Nd = len(mxyz), len(dxyz)
        f = lambda a, t: None if a is None else np.ascontiguousarray(a, dtype=t)
        mcol, dcol = f(mcol, np.int32), f(dcol, np.int32)
        mfpfh, dfpfh = f(mfpfh, np.float32), f(dfpfh, np.float32)
        self.lib.ref_set_clouds(self.h, self.Nm, _p(mxyz, C.c_float), _p(mcol, C.c_int), _p(mfpfh, C.c_float),
                                self.Nd, _p(dxyz, C.c_float), _p(dcol, C.c_int), _p(dfpfh, C.c_float))

    def set_nd(self, nd):
        self.Nd = nd
        self.lib.ref_set_nd(self.h, nd)

    def build_dt(self):
        self.lib.ref_build_dt(self.h)
        g = np.zeros(8)
        self.lib.ref_dt_geom(self.h, _p(g, C.c_double))
        self.geom = g
        self.S = int(g[7])

    def dt_dump(self, want_vhd=False):
        S = self.S
        dist = np.zeros(S ** 3, np.float32); near = np.zeros(S ** 3 * 3, np.int32); cc = np.zeros(S ** 3, np.int32)
        vhd = np.zeros(S ** 3 * 3, np.int16) if want_vhd else None
        self.lib.ref_dt_dump(self.h, _p(dist, C.c_float), _p(vhd, C.c_short), _p(near, C.c_int), _p(cc, C.c_int))
        return dist, near.reshape(-1, 3), cc, (vhd.reshape(-1, 3) if want_vhd else None)

    def cell_points(self, x, y, z):
        buf = np.zeros(64, np.int32)
        n = self.lib.ref_cell_points(self.h, x, y, z, _p(buf, C.c_int), 64)
        return buf[:n].copy()

    def cell_csr(self, cc):
        """cell -> model point lists as CSR over the S^3 cells (x fastest)."""
        S = self.S
        ptr = np.zeros(S ** 3 + 1, np.int32); idx = []
        occ = np.nonzero(cc != -2)[0]
        counts = np.zeros(S ** 3, np.int32)
        lists = {}
        for k in occ:
            x = int(k % S); y = int((k // S) % S); z = int(k // (S * S))
            pts = self.cell_points(x, y, z)
            lists[int(k)] = pts; counts[k] = len(pts)
        ptr[1:] = np.cumsum(counts)
        for k in occ:
            idx.extend(lists[int(k)].tolist())
        return ptr, np.array(idx, np.int32)

    def distance(self, q):
        q = np.ascontiguousarray(q, dtype=np.float64).reshape(-1, 3)
        out = np.zeros(len(q), np.float32)
        self.lib.ref_distance(self.h, len(q), _p(q, C.c_double), _p(out, C.c_float))
        return out

    def initialize(self):
        self.lib.ref_initialize(self.h)

    def weights(self):
        w = np.zeros(self.Nd, np.float32); self.lib.ref_get_weights(self.h, _p(w, C.c_float)); return w

    def maxrotdis(self):
        m = np.zeros(20 * self.Nd, np.float32); self.lib.ref_get_maxrotdis(self.h, _p(m, C.c_float))
        return m.reshape(20, self.Nd)

    def inlier_num(self):
        return self.lib.ref_inlier_num(self.h)

    def sse_thresh(self):
        return float(self.lib.ref_sse_thresh(self.h))

    def set_sse_thresh(self, v):
        self.lib.ref_set_sse_thresh(self.h, C.c_float(v))

    def set_opt_error(self, e):
        self.lib.ref_set_opt_error(self.h, C.c_float(e))

    def set_init_trans(self, x, y, z, w, lb=0.0):
        self.lib.ref_set_init_trans(self.h, C.c_float(x), C.c_float(y), C.c_float(z), C.c_float(w), C.c_float(lb))

    def set_rotated(self, xyz):
        xyz = np.ascontiguousarray(xyz, dtype=np.float32)
        assert xyz.shape[0] == self.Nd
        self.lib.ref_set_rotated(self.h, _p(xyz, C.c_float))

    def inner_bnb(self, level, want_node=True):
        node = np.zeros(6, np.float32)
        r = self.lib.ref_inner_bnb(self.h, int(level), _p(node, C.c_float), 1 if want_node else 0)
        return np.float32(r), node

    def check_compatibilities(self, x, y, z):
        return int(self.lib.ref_check_compatibilities(self.h, C.c_float(x), C.c_float(y), C.c_float(z)))

    def sum_fpfh(self, x, y, z):
        return np.float32(self.lib.ref_sum_fpfh(self.h, C.c_float(x), C.c_float(y), C.c_float(z)))

    def icp(self, R, t):
        R = np.ascontiguousarray(R, dtype=np.float64).reshape(9).copy(); t = np.ascontiguousarray(t, dtype=np.float64).reshape(3).copy()
        corr = np.zeros(self.Nd * 2, np.int32); dis = np.zeros(self.Nd, np.float64)
        e = self.lib.ref_icp(self.h, _p(R, C.c_double), _p(t, C.c_double), _p(corr, C.c_int), _p(dis, C.c_double))
        return np.float32(e), R.reshape(3, 3), t, corr.reshape(-1, 2), dis

    def register(self):
        out = np.zeros(16, np.float64)
        self.lib.ref_register(self.h, _p(out, C.c_double))
        n = self.lib.ref_last_log(self.h, None, 0)
        buf = C.create_string_buffer(n + 1)
        self.lib.ref_last_log(self.h, buf, n + 1)
        return dict(R=out[:9].reshape(3, 3).copy(), t=out[9:12].copy(), error=np.float32(out[12]), comp=int(out[13]),
                    Nd=int(out[14]), inlierNum=int(out[15]), log=buf.value.decode(errors="replace"))

    def trace_begin(self, cap=1 << 22):
        self.lib.ref_trace_begin(C.c_longlong(cap)); self._cap = cap

    def trace_end(self):
        buf = np.zeros(self._cap * 6, np.float32)
        n = self.lib.ref_trace_end(_p(buf, C.c_float), C.c_longlong(self._cap))
        return buf.reshape(-1, 6)[:min(n, self._cap)].copy()

    def counters(self, reset=True):
        out = np.zeros(3, np.int64)
        self.lib.ref_counters(_p(out, C.c_longlong), 1 if reset else 0)
        self.trans_empty = int(out[2])
        return int(out[0]), int(out[1])

    def load_dt(self, S, geom8, dist):
        """geometric path only (no cell tables): adopt a distance grid built by another builder."""
        g4 = np.array([geom8[0], geom8[2], geom8[4], geom8[6]], np.float64)
        dist = np.ascontiguousarray(dist, dtype=np.float32)
        self.lib.ref_load_dt(self.h, int(S), _p(g4, C.c_double), _p(dist, C.c_float))
        self.geom = np.asarray(geom8, np.float64); self.S = int(S)

    def intro_select(self, data, k):
        d = np.ascontiguousarray(data, dtype=np.float32).copy()
        self.lib.ref_intro_select(_p(d, C.c_float), C.c_longlong(len(d)), C.c_longlong(k))
        return d

    def svd3(self, H):
        H = np.ascontiguousarray(H, dtype=np.float64).reshape(9)
        U = np.zeros(9); W = np.zeros(3); V = np.zeros(9)
        self.lib.ref_svd3(_p(H, C.c_double), _p(U, C.c_double), _p(W, C.c_double), _p(V, C.c_double))
        return U.reshape(3, 3), W, V.reshape(3, 3)


def error_trace(log):
    """'Error*: <v> (<tag>' lines of the reference's stdout -> [(tag, value)]"""
    out = []
    for line in log.splitlines():
        if line.startswith("Error*:") and "(" in line:
            v = float(line.split()[1])
            tag = line.split("(")[1].split()[0].rstrip(")")
            out.append((tag, v))
    return out


class Oracle:
    """The plain-C restatement (oracle/goicp_oracle.c); same surface as Ref so tests read symmetrically."""

    def __init__(self):
        self.lib = L = C.CDLL(ORACLE_SO)
        L.orc_new.restype = C.c_void_p
        for n in ("orc_inner_bnb", "orc_icp", "orc_register", "orc_sse_thresh", "orc_get_opt_error", "orc_sum_fpfh"):
            getattr(L, n).restype = C.c_float
        L.orc_trace_end.restype = C.c_longlong
        self.h = C.c_void_p(L.orc_new())
        self.Nd = self.Nm = 0

    def close(self):
        if self.h:
            self.lib.orc_free(self.h)
            self.h = None

    def set_config(self, cfgvec):
        self.cfgvec = np.ascontiguousarray(cfgvec, dtype=np.float64)
        self.lib.orc_set_config(self.h, _p(self.cfgvec, C.c_double))

    def set_clouds(self, mxyz, dxyz, mcol=None, dcol=None, mfpfh=None, dfpfh=None):
        mxyz = np.ascontiguousarray(mxyz, dtype=np.float32); dxyz = np.ascontiguousarray(dxyz, dtype=np.float32)
        self.Nm, self.Nd = len(mxyz), len(dxyz)
        f = lambda a, t: None if a is None else np.ascontiguousarray(a, dtype=t)
        mcol, dcol = f(mcol, np.int32), f(dcol, np.int32)
        mfpfh, dfpfh = f(mfpfh, np.float32), f(dfpfh, np.float32)
        self.lib.orc_set_clouds(self.h, self.Nm, _p(mxyz, C.c_float), _p(mcol, C.c_int), _p(mfpfh, C.c_float),
                                self.Nd, _p(dxyz, C.c_float), _p(dcol, C.c_int), _p(dfpfh, C.c_float))

    def set_nd(self, nd):
        self.Nd = nd
        self.lib.orc_set_nd(self.h, nd)

    def _geom(self):
        g = np.zeros(8)
        self.lib.orc_dt_geom(self.h, _p(g, C.c_double))
        self.geom = g
        self.S = int(g[7])

    def build_dt(self):
        self.lib.orc_build_dt(self.h)
        self._geom()

    def load_dt(self, S, geom8, dist, nearest3=None, cellc=None, cell_ptr=None, cell_idx=None):
        g4 = np.array([geom8[0], geom8[2], geom8[4], geom8[6]], dtype=np.float64)
        f = lambda a, t: None if a is None else np.ascontiguousarray(a, dtype=t)
        dist = f(dist, np.float32); nearest3 = f(nearest3, np.int32); cellc = f(cellc, np.int32)
        cell_ptr = f(cell_ptr, np.int32); cell_idx = f(cell_idx, np.int32)
        self.lib.orc_load_dt(self.h, int(S), _p(g4, C.c_double), _p(dist, C.c_float), _p(nearest3, C.c_int),
                             _p(cellc, C.c_int), _p(cell_ptr, C.c_int), _p(cell_idx, C.c_int))
        self._geom()

    def dt_dump(self, want_vhd=False):
        S = self.S
        dist = np.zeros(S ** 3, np.float32); near = np.zeros(S ** 3 * 3, np.int32); cc = np.zeros(S ** 3, np.int32)
        vhd = np.zeros(S ** 3 * 3, np.int16) if want_vhd else None
        self.lib.orc_dt_dump(self.h, _p(dist, C.c_float), _p(vhd, C.c_short), _p(near, C.c_int), _p(cc, C.c_int))
        return dist, near.reshape(-1, 3), cc, (vhd.reshape(-1, 3) if want_vhd else None)

    def cell_points(self, x, y, z):
        buf = np.zeros(64, np.int32)
        n = self.lib.orc_cell_points(self.h, x, y, z, _p(buf, C.c_int), 64)
        return buf[:n].copy()

    def distance(self, q):
        q = np.ascontiguousarray(q, dtype=np.float64).reshape(-1, 3)
        out = np.zeros(len(q), np.float32)
        self.lib.orc_distance(self.h, len(q), _p(q, C.c_double), _p(out, C.c_float))
        return out

    def initialize(self):
        self.lib.orc_initialize(self.h)

    def weights(self):
        w = np.zeros(self.Nd, np.float32); self.lib.orc_get_weights(self.h, _p(w, C.c_float)); return w

    def set_weights(self, w):
        w = np.ascontiguousarray(w, dtype=np.float32); self.lib.orc_set_weights(self.h, _p(w, C.c_float))

    def maxrotdis(self):
        m = np.zeros(20 * self.Nd, np.float32); self.lib.orc_get_maxrotdis(self.h, _p(m, C.c_float))
        return m.reshape(20, self.Nd)

    def inlier_num(self):
        return self.lib.orc_inlier_num(self.h)

    def sse_thresh(self):
        return float(self.lib.orc_sse_thresh(self.h))

    def set_sse_thresh(self, v):
        self.lib.orc_set_sse_thresh(self.h, C.c_float(v))

    def set_opt_error(self, e):
        self.lib.orc_set_opt_error(self.h, C.c_float(e))

    def set_init_trans(self, x, y, z, w, lb=0.0):
        self.lib.orc_set_init_trans(self.h, C.c_float(x), C.c_float(y), C.c_float(z), C.c_float(w), C.c_float(lb))

    def set_rotated(self, xyz):
        xyz = np.ascontiguousarray(xyz, dtype=np.float32)
        assert xyz.shape[0] == self.Nd
        self.lib.orc_set_rotated(self.h, _p(xyz, C.c_float))

    def rotate(self, R):
        R = np.ascontiguousarray(R, dtype=np.float32).reshape(9)
        self.lib.orc_rotate(self.h, _p(R, C.c_float))

    def eval_cube(self, level, x, y, z, w):
        ub = C.c_float(); lb = C.c_float()
        self.lib.orc_eval_cube(self.h, int(level), C.c_float(x), C.c_float(y), C.c_float(z), C.c_float(w),
                               C.byref(ub), C.byref(lb))
        return np.float32(ub.value), np.float32(lb.value)

    def eval_cubes(self, level, cubes):
        cubes = np.asarray(cubes, dtype=np.float32).reshape(-1, 4)
        out = np.zeros((len(cubes), 2), np.float32)
        for i, cb in enumerate(cubes):
            out[i] = self.eval_cube(level, *cb)
        return out[:, 0].copy(), out[:, 1].copy()

    def set_pop_budget(self, n):
        self.lib.orc_set_pop_budget(self.h, C.c_longlong(int(n)))

    def inner_bnb(self, level, want_node=True):
        node = np.zeros(6, np.float32)
        r = self.lib.orc_inner_bnb(self.h, int(level), _p(node, C.c_float), 1 if want_node else 0)
        return np.float32(r), node

    def check_compatibilities(self, x, y, z):
        return int(self.lib.orc_check_compatibilities(self.h, C.c_float(x), C.c_float(y), C.c_float(z)))

    def sum_fpfh(self, x, y, z):
        return np.float32(self.lib.orc_sum_fpfh(self.h, C.c_float(x), C.c_float(y), C.c_float(z)))

    def icp(self, R, t):
        R = np.ascontiguousarray(R, dtype=np.float64).reshape(9).copy(); t = np.ascontiguousarray(t, dtype=np.float64).reshape(3).copy()
        corr = np.zeros(self.Nd * 2, np.int32); dis = np.zeros(self.Nd, np.float64)
        e = self.lib.orc_icp(self.h, _p(R, C.c_double), _p(t, C.c_double), _p(corr, C.c_int), _p(dis, C.c_double))
        return np.float32(e), R.reshape(3, 3), t, corr.reshape(-1, 2), dis

    def register(self):
        out = np.zeros(16, np.float64)
        self.lib.orc_register(self.h, _p(out, C.c_double))
        n = self.lib.orc_last_log(self.h, None, 0)
        buf = C.create_string_buffer(n + 1)
        self.lib.orc_last_log(self.h, buf, n + 1)
        return dict(R=out[:9].reshape(3, 3).copy(), t=out[9:12].copy(), error=np.float32(out[12]), comp=int(out[13]),
                    Nd=int(out[14]), inlierNum=int(out[15]), log=buf.value.decode(errors="replace"))

    def trace_begin(self, cap=1 << 22):
        self.lib.orc_trace_begin(self.h, C.c_longlong(cap)); self._cap = cap

    def trace_end(self):
        buf = np.zeros(self._cap * 6, np.float32)
        n = self.lib.orc_trace_end(self.h, _p(buf, C.c_float), C.c_longlong(self._cap))
        return buf.reshape(-1, 6)[:min(n, self._cap)].copy()

    def counters(self, reset=True):
        out = np.zeros(4, np.int64)
        self.lib.orc_counters(self.h, _p(out, C.c_longlong), 1 if reset else 0)
        return tuple(int(v) for v in out)

    def intro_select(self, data, k):
        d = np.ascontiguousarray(data, dtype=np.float32).copy()
        self.lib.orc_intro_select(_p(d, C.c_float), C.c_longlong(len(d)), C.c_longlong(k))
        return d

    def svd3_rot(self, H):
        H = np.ascontiguousarray(H, dtype=np.float64).reshape(9); R = np.zeros(9)
        self.lib.orc_svd3_rot(_p(H, C.c_double), _p(R, C.c_double))
        return R.reshape(3, 3)

    def axis_angle(self, v):
        R = np.zeros(9, np.float32)
        self.lib.orc_axis_angle(C.c_float(v[0]), C.c_float(v[1]), C.c_float(v[2]), _p(R, C.c_float))
        return R.reshape(3, 3)
