"""ctypes bindings over include/goicp_engine.h (hot path) and include/goicp_c.h (GoICP class + file pipeline)."""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("GOICP_LIB") or os.path.join(HERE, "libgoicp_hip.so")      # GOICP_LIB: the diagnosis build (make -C csrc probe) for tools/

GOICP_OK = 0
STATUS = {0: "OK", 1: "ERR_ARG", 2: "ERR_NO_DEVICE", 3: "ERR_HIP", 4: "ERR_OVERFLOW", 5: "ERR_LIMIT"}

# key order of the config vector (reference jly_main.cpp:236-264) and the shipped defaults (reference config.txt:4-53)
CFG_KEYS = ["MSEThresh", "rotMinX", "rotMinY", "rotMinZ", "rotWidth", "transMinX", "transMinY", "transMinZ",
            "transWidth", "trimFraction", "regularization", "regularizationNeighbors", "regularizationFPFH",
            "cfpfh", "norm", "ponderation", "distTransSize", "distTransExpandFactor"]
DEFAULT_CFG = dict(MSEThresh=0.01, rotMinX=-3.1416, rotMinY=-3.1416, rotMinZ=-3.1416, rotWidth=6.2832,
                   transMinX=-0.5, transMinY=-0.5, transMinZ=-0.5, transWidth=1.0, trimFraction=0.0,
                   regularization=0.0005, regularizationNeighbors=0.0, regularizationFPFH=0.0, cfpfh=0, norm=2,
                   ponderation=1, distTransSize=20, distTransExpandFactor=2.0)


def cfg_vector(**over):
    d = dict(DEFAULT_CFG)
    for k, v in over.items():
        if k not in d:
            raise KeyError(k)
        d[k] = v
    return np.array([float(d[k]) for k in CFG_KEYS], dtype=np.float64)


class EngineError(RuntimeError):
    pass


class _Desc(C.Structure):
    _fields_ = [("S", C.c_int32), ("xMin", C.c_double), ("yMin", C.c_double), ("zMin", C.c_double), ("scale", C.c_double),
                ("dt_dist", C.c_void_p), ("nearest_occ", C.c_void_p), ("cell_color", C.c_void_p), ("cell_ptr", C.c_void_p),
                ("cell_idx", C.c_void_p), ("Nm", C.c_int32), ("model_xyz", C.c_void_p), ("model_color", C.c_void_p),
                ("model_fpfh", C.c_void_p), ("Nd", C.c_int32), ("data_xyz", C.c_void_p), ("data_color", C.c_void_p),
                ("data_fpfh", C.c_void_p), ("weights", C.c_void_p), ("maxRotDis", C.c_void_p), ("inlierNum", C.c_int32),
                ("doTrim", C.c_int32), ("norm", C.c_int32), ("cfpfh", C.c_int32), ("regularization", C.c_float),
                ("regularizationNeighbors", C.c_float), ("regularizationFPFH", C.c_float), ("device_id", C.c_int32),
                ("parity_mode", C.c_int32), ("dt_dist_dev", C.c_void_p), ("nearest_occ_dev", C.c_void_p), ("n_occ", C.c_int32),
                ("occ_cells", C.c_void_p), ("occ_color", C.c_void_p), ("occ_ptr", C.c_void_p)]


class BnbJob(C.Structure):
    _fields_ = [("R", C.c_float * 9), ("rot_level", C.c_int32), ("optError_in", C.c_float), ("SSEThresh", C.c_float),
                ("init", C.c_float * 4), ("init_lb", C.c_float), ("max_pops", C.c_uint32), ("flags", C.c_uint32)]


class BnbResult(C.Structure):
    _fields_ = [("optErrorT", C.c_float), ("node", C.c_float * 4), ("found", C.c_int32), ("status", C.c_int32),
                ("pops", C.c_uint32), ("cube_evals", C.c_uint32)]


_lib = None


def load_library():
    """Loads the in-tree HIP library; fails loudly when it has not been built (no fallback path exists)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise EngineError("%s is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(make -C go-icp-protein-cavities_amd/csrc); there is no CPU fallback" % LIB_PATH)
    L = C.CDLL(LIB_PATH)
    L.goicp_engine_last_error.restype = C.c_char_p
    L.goicp_engine_stream.restype = C.c_void_p
    L.goicp_engine_eval_kernel_name.restype = C.c_char_p
    L.goicp_new.restype = C.c_void_p
    L.goicp_get_engine.restype = C.c_void_p
    L.goicp_register.restype = C.c_float
    L.goicp_icp.restype = C.c_float
    L.goicp_sse_thresh.restype = C.c_float
    L.goicp_error_message.restype = C.c_char_p
    L.goicp_compute_rmsd.restype = C.c_float
    _lib = L
    return L


def lib():
    return load_library()


def _f32(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.float32)


def _i32(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.int32)


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class Engine:
    """One engine handle = one GPU + one HIP stream (include/goicp_engine.h)."""

    def __init__(self, *, S, geom4, dist, data_xyz, weights, maxrotdis, model_xyz, inlier_num, do_trim=False, norm=2, cfpfh=0,
                 reg=0.0, reg_fpfh=0.0, nearest=None, cell_color=None, cell_ptr=None, cell_idx=None, model_color=None,
                 data_color=None, model_fpfh=None, data_fpfh=None, device_id=0, parity_mode=1, _borrowed=None):
        self.L = load_library()
        if _borrowed is not None:
            self.h = C.c_void_p(_borrowed); self._own = False; self.Nd = None
            return
        self._keep = k = dict(dist=_f32(dist), dxyz=_f32(data_xyz), w=_f32(weights), mrd=_f32(maxrotdis), mxyz=_f32(model_xyz),
                              near=_i32(nearest), cc=_i32(cell_color), cp=_i32(cell_ptr), ci=_i32(cell_idx), mc=_i32(model_color),
                              dc=_i32(data_color), mf=_f32(model_fpfh), df=_f32(data_fpfh))
        d = _Desc()
        d.S = int(S); d.xMin, d.yMin, d.zMin, d.scale = [float(v) for v in geom4]
        d.dt_dist = _ptr(k["dist"]); d.nearest_occ = _ptr(k["near"]); d.cell_color = _ptr(k["cc"]); d.cell_ptr = _ptr(k["cp"]); d.cell_idx = _ptr(k["ci"])
        d.Nm = len(k["mxyz"]) // 3 if k["mxyz"].ndim == 1 else len(k["mxyz"]); d.model_xyz = _ptr(k["mxyz"]); d.model_color = _ptr(k["mc"]); d.model_fpfh = _ptr(k["mf"])
        d.Nd = len(k["dxyz"]) // 3 if k["dxyz"].ndim == 1 else len(k["dxyz"]); d.data_xyz = _ptr(k["dxyz"]); d.data_color = _ptr(k["dc"]); d.data_fpfh = _ptr(k["df"])
        d.weights = _ptr(k["w"]); d.maxRotDis = _ptr(k["mrd"]); d.inlierNum = int(inlier_num); d.doTrim = int(bool(do_trim)); d.norm = int(norm)
        d.cfpfh = int(cfpfh); d.regularization = float(reg); d.regularizationNeighbors = 0.0; d.regularizationFPFH = float(reg_fpfh)
        d.device_id = int(device_id); d.parity_mode = int(parity_mode)
        self.Nd = d.Nd
        h = C.c_void_p()
        rc = self.L.goicp_engine_create(C.byref(d), C.byref(h))
        if rc != GOICP_OK:
            raise EngineError("goicp_engine_create failed: %s (no CPU fallback exists)" % STATUS.get(rc, rc))
        self.h = h; self._own = True

    @classmethod
    def borrowed(cls, handle, Nd):
        e = cls.__new__(cls); e.L = load_library(); e.h = C.c_void_p(handle); e._own = False; e.Nd = Nd
        return e

    def close(self):
        if getattr(self, "h", None) and self._own:
            self.L.goicp_engine_destroy(self.h)
        self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc, what):
        if rc != GOICP_OK:
            msg = self.L.goicp_engine_last_error(self.h)
            raise EngineError("%s: %s %s" % (what, STATUS.get(rc, rc), msg.decode() if msg else ""))

    def set_rotation(self, slot, R):
        R = np.ascontiguousarray(R, dtype=np.float32).reshape(9)
        self._chk(self.L.goicp_engine_set_rotation(self.h, int(slot), _ptr(R)), "set_rotation")

    def eval_cubes(self, slot, level, cubes):
        cubes = np.ascontiguousarray(cubes, dtype=np.float32).reshape(-1, 4)
        n = len(cubes); ub = np.zeros(n, np.float32); lb = np.zeros(n, np.float32)
        self._chk(self.L.goicp_engine_eval_cubes(self.h, int(slot), int(level), _ptr(cubes), n, _ptr(ub), _ptr(lb)), "eval_cubes")
        return ub, lb

    def eval_cubes_dev(self, slot, level, cubes_ptr, n, ub_ptr, lb_ptr):
        self._chk(self.L.goicp_engine_eval_cubes_dev(self.h, int(slot), int(level), C.c_void_p(cubes_ptr), int(n), C.c_void_p(ub_ptr), C.c_void_p(lb_ptr)), "eval_cubes_dev")

    def eval_cubes_multi_dev(self, slots, levels, cubes_ptr, n, ub_ptr, lb_ptr):
        slots = _i32(slots); levels = _i32(levels)
        self._chk(self.L.goicp_engine_eval_cubes_multi_dev(self.h, _ptr(slots), _ptr(levels), len(slots), C.c_void_p(cubes_ptr), int(n),
                                                           C.c_void_p(ub_ptr), C.c_void_p(lb_ptr)), "eval_cubes_multi_dev")

    def corner_terms(self, slot, xyz):
        xyz = np.ascontiguousarray(xyz, dtype=np.float32).reshape(-1, 3)
        n = len(xyz); cnt = np.zeros(n, np.int32); fp = np.zeros(n, np.float32)
        self._chk(self.L.goicp_engine_corner_terms(self.h, int(slot), _ptr(xyz), n, _ptr(cnt), _ptr(fp)), "corner_terms")
        return cnt, fp

    def inner_bnb_batch(self, jobs):
        """jobs: list of dict(R, level, E, sse, init=(x,y,z,w), init_lb=0)."""
        n = len(jobs)
        J = (BnbJob * n)(); Rr = (BnbResult * n)()
        for i, j in enumerate(jobs):
            R = np.asarray(j["R"], np.float32).reshape(9)
            for k in range(9):
                J[i].R[k] = float(R[k])
            J[i].rot_level = int(j["level"]); J[i].optError_in = float(j["E"]); J[i].SSEThresh = float(j["sse"])
            for k in range(4):
                J[i].init[k] = float(j["init"][k])
            J[i].init_lb = float(j.get("init_lb", 0.0)); J[i].max_pops = int(j.get("max_pops", 0)); J[i].flags = int(j.get("flags", 0))
        rc = self.L.goicp_engine_inner_bnb_batch(self.h, J, n, Rr)
        if rc not in (4, 5):      # ERR_OVERFLOW / ERR_LIMIT are per-job outcomes reported in each result's status
            self._chk(rc, "inner_bnb_batch")
        return [dict(val=np.float32(r.optErrorT), node=np.array(list(r.node), np.float32), found=int(r.found), status=int(r.status),
                     pops=int(r.pops), cube_evals=int(r.cube_evals)) for r in Rr]

    def inner_bnb(self, **job):
        return self.inner_bnb_batch([job])[0]

    def point_eval(self, R, t, want_compat=True):
        R = np.ascontiguousarray(R, dtype=np.float64).reshape(9); t = np.ascontiguousarray(t, dtype=np.float64).reshape(3)
        dist = np.zeros(self.Nd, np.float32); ok = np.zeros(self.Nd, np.int32) if want_compat else None
        self._chk(self.L.goicp_engine_point_eval(self.h, _ptr(R), _ptr(t), _ptr(dist), _ptr(ok)), "point_eval")
        return dist, ok

    def icp_nn(self, R, t, ties=False):
        R = np.ascontiguousarray(R, dtype=np.float64).reshape(9); t = np.ascontiguousarray(t, dtype=np.float64).reshape(3)
        ids = np.zeros(self.Nd, np.int32); d = np.zeros(self.Nd, np.float32)
        if ties:
            nt = np.zeros(self.Nd, np.int32)
            self._chk(self.L.goicp_engine_icp_nn_ties(self.h, _ptr(R), _ptr(t), _ptr(ids), _ptr(d), _ptr(nt)), "icp_nn_ties")
            return ids, d, nt
        self._chk(self.L.goicp_engine_icp_nn(self.h, _ptr(R), _ptr(t), _ptr(ids), _ptr(d)), "icp_nn")
        return ids, d

    def synchronize(self):
        self._chk(self.L.goicp_engine_synchronize(self.h), "synchronize")

    def stream(self):
        return self.L.goicp_engine_stream(self.h)

    def search_stats(self):
        r = C.c_uint64(); c = C.c_uint64()
        self.L.goicp_engine_search_stats(self.h, C.byref(r), C.byref(c))
        return dict(rounds=r.value, cube_evals_launched=c.value)

    def eval_kernel_name(self):
        return self.L.goicp_engine_eval_kernel_name(self.h).decode()

    def last_eval_fused(self):
        """True if the last multi-group sweep evaluated its (no radius, radius) group pairs from one pass of gathers."""
        return bool(self.L.goicp_engine_last_eval_fused(self.h))

    def stats(self):
        ce = C.c_uint64(); lk = C.c_uint64(); ms = C.c_double()
        self.L.goicp_engine_stats(self.h, C.byref(ce), C.byref(lk), C.byref(ms))
        return dict(cube_evals=ce.value, lookups=lk.value, kernel_ms=ms.value)


class GoICP:
    """Registration-level surface (GoICP::BuildDT / Register) through include/goicp_c.h."""

    def __init__(self, cfgvec=None, device_id=0, parity_mode=1, speculate=128, verbose=0, dt_mode=0):
        self.L = load_library()
        self.h = C.c_void_p(self.L.goicp_new())
        self.Nd = self.Nm = 0
        if cfgvec is not None:
            self.set_config(cfgvec)
        self.L.goicp_set_options(self.h, int(device_id), int(parity_mode), int(speculate), int(verbose))
        if dt_mode:
            self.set_dt_mode(dt_mode)

    def set_icp_one_launch(self, on):
        """ICP3D::Run of cavity-sized clouds as one launch per call (include/goicp_c.h: goicp_set_icp_one_launch); off by default."""
        if self.L.goicp_set_icp_one_launch(self.h, int(bool(on))) != GOICP_OK:
            raise EngineError("set_icp_one_launch")

    def set_dt_mode(self, mode):
        """0: the reference's DT3D::Build restated on the host (bit-identical grid); 1: exact transform built on the device (opt-in)."""
        if self.L.goicp_set_dt_mode(self.h, int(mode)) != GOICP_OK:
            raise EngineError("set_dt_mode")

    def close(self):
        if self.h:
            self.L.goicp_free(self.h)
            self.h = None

    def set_budget(self, wall_seconds=0.0, search_pops=0):
        """Budgets for searches that cannot run to convergence (BASELINE configs[4]): register() then returns the best pose so far and
        out["budget_hit"] is True. 0 = none."""
        if self.L.goicp_set_budget(self.h, C.c_double(float(wall_seconds)), C.c_uint32(int(search_pops))) != GOICP_OK:
            raise EngineError("set_budget")

    def set_config(self, cfgvec):
        v = np.ascontiguousarray(cfgvec, dtype=np.float64)
        assert v.shape == (18,)
        self.L.goicp_set_config(self.h, _ptr(v))

    def read_config(self, path):
        if self.L.goicp_read_config(self.h, path.encode()) != GOICP_OK:
            raise EngineError("cannot read config %s" % path)

    def get_config(self):
        v = np.zeros(18); self.L.goicp_get_config(self.h, _ptr(v)); return v

    def set_clouds(self, mxyz, dxyz, mcol=None, dcol=None, mfpfh=None, dfpfh=None):
        mxyz = _f32(mxyz).reshape(-1, 3); dxyz = _f32(dxyz).reshape(-1, 3)
        self.Nm, self.Nd = len(mxyz), len(dxyz)
        mcol, dcol, mfpfh, dfpfh = _i32(mcol), _i32(dcol), _f32(mfpfh), _f32(dfpfh)
        rc = self.L.goicp_set_clouds(self.h, self.Nm, _ptr(mxyz), _ptr(mcol), _ptr(mfpfh), self.Nd, _ptr(dxyz), _ptr(dcol), _ptr(dfpfh))
        if rc != GOICP_OK:
            raise EngineError("set_clouds: bad arguments")

    def set_nd(self, nd):
        if self.L.goicp_set_nd(self.h, int(nd)) != GOICP_OK:
            raise EngineError("set_nd")
        self.Nd = int(nd)

    def build_dt(self):
        self.L.goicp_build_dt(self.h)
        if self.L.goicp_status(self.h) != GOICP_OK:
            raise EngineError("build_dt: %s" % self.L.goicp_error_message(self.h).decode())
        g = np.zeros(8); self.L.goicp_dt_geom(self.h, _ptr(g)); self.geom = g; self.S = int(g[7])

    def load_dt(self, S, geom8, dist, nearest_lin=None, cellc=None, cell_ptr=None, cell_idx=None):
        g4 = np.array([geom8[0], geom8[2], geom8[4], geom8[6]], np.float64)
        dist = _f32(dist); nearest_lin, cellc, cell_ptr, cell_idx = _i32(nearest_lin), _i32(cellc), _i32(cell_ptr), _i32(cell_idx)
        rc = self.L.goicp_load_dt(self.h, int(S), _ptr(g4), _ptr(dist), _ptr(nearest_lin), _ptr(cellc), _ptr(cell_ptr), _ptr(cell_idx))
        if rc != GOICP_OK:
            raise EngineError("load_dt")
        g = np.zeros(8); self.L.goicp_dt_geom(self.h, _ptr(g)); self.geom = g; self.S = int(g[7])

    def dt_dump(self):
        n = self.S ** 3
        dist = np.zeros(n, np.float32); near = np.zeros(n, np.int32); cc = np.zeros(n, np.int32)
        self.L.goicp_dt_dump(self.h, _ptr(dist), _ptr(near), _ptr(cc))
        return dist, near, cc

    def initialize(self):
        rc = self.L.goicp_initialize(self.h)
        if rc != GOICP_OK:
            raise EngineError("initialize: %s: %s" % (STATUS.get(rc, rc), self.L.goicp_error_message(self.h).decode()))

    def engine(self):
        return Engine.borrowed(self.L.goicp_get_engine(self.h), self.Nd)

    def clear(self):
        self.L.goicp_clear(self.h)

    def weights(self):
        w = np.zeros(self.Nd, np.float32); self.L.goicp_get_weights(self.h, _ptr(w)); return w

    def maxrotdis(self):
        m = np.zeros(20 * self.Nd, np.float32); self.L.goicp_get_maxrotdis(self.h, _ptr(m)); return m.reshape(20, self.Nd)

    def inlier_num(self):
        return self.L.goicp_inlier_num(self.h)

    def sse_thresh(self):
        return float(self.L.goicp_sse_thresh(self.h))

    def icp(self, R, t):
        R = np.ascontiguousarray(R, dtype=np.float64).reshape(9).copy(); t = np.ascontiguousarray(t, dtype=np.float64).reshape(3).copy()
        corr = np.zeros((self.Nd, 2), np.int32); dis = np.zeros(self.Nd, np.float64)
        e = self.L.goicp_icp(self.h, _ptr(R), _ptr(t), _ptr(corr))
        self.L.goicp_icp_distances(self.h, _ptr(dis))
        return np.float32(e), R.reshape(3, 3), t, corr, dis

    def stats(self):
        """Counters of this handle so far (goicp_get_stats): rotation pops, InnerBnB calls / launches / jobs, cube evaluations, queue pops, ICP calls / iterations."""
        st = np.zeros(8, np.uint64); ms = C.c_double()
        self.L.goicp_get_stats(self.h, _ptr(st), C.byref(ms))
        keys = ["rot_pops", "inner_calls", "inner_launches", "inner_jobs_launched", "cube_evals", "trans_pops", "icp_calls", "icp_iters"]
        return dict(zip(keys, [int(v) for v in st]))

    def record_jobs(self, on=True):
        self.L.goicp_record_jobs(self.h, 1 if on else 0)

    def job_log(self):
        n = self.L.goicp_get_job_log(self.h, None, None, 0)
        J = (BnbJob * max(n, 1))(); Rr = (BnbResult * max(n, 1))()
        self.L.goicp_get_job_log(self.h, J, Rr, n)
        return [dict(R=np.array(list(J[i].R), np.float32).reshape(3, 3), level=int(J[i].rot_level), E=np.float32(J[i].optError_in),
                     sse=np.float32(J[i].SSEThresh), val=np.float32(Rr[i].optErrorT), cube_evals=int(Rr[i].cube_evals), pops=int(Rr[i].pops))
                for i in range(n)]

    def register(self):
        out = np.zeros(16)
        e = self.L.goicp_register(self.h, _ptr(out))
        rc = self.L.goicp_status(self.h)
        if rc != GOICP_OK or not np.isfinite(e):
            raise EngineError("register: %s: %s" % (STATUS.get(rc, rc), self.L.goicp_error_message(self.h).decode()))
        tags = np.zeros(256, np.int32); vals = np.zeros(256, np.float32)
        n = self.L.goicp_get_trace(self.h, _ptr(tags), _ptr(vals), 256)
        names = ["Init", "ICP", "BNB"]
        st = np.zeros(8, np.uint64); ms = C.c_double()
        self.L.goicp_get_stats(self.h, _ptr(st), C.byref(ms))
        keys = ["rot_pops", "inner_calls", "inner_launches", "inner_jobs_launched", "cube_evals", "trans_pops", "icp_calls", "icp_iters"]
        s6 = np.zeros(6, np.uint64); self.L.goicp_get_search_stats(self.h, _ptr(s6))
        search = dict(pops_launched=int(s6[0]), rounds=int(s6[1]), cube_evals_launched=int(s6[2]), slab_regrowths=int(s6[3]), seconds=float(s6[4]) * 1e-6)
        return dict(R=out[:9].reshape(3, 3).copy(), t=out[9:12].copy(), error=np.float32(out[12]), comp=int(out[13]), Nd=int(out[14]),
                    inlierNum=int(out[15]), trace=[(names[tags[i]], float(vals[i])) for i in range(min(n, 256))],
                    stats=dict(zip(keys, [int(v) for v in st])), kernel_ms=ms.value, search=search, budget_hit=bool(s6[5]))


_RES_KEYS = ["error", "compat", "Nd", "seconds", "dt_seconds"]


def _unpack24(r):
    return dict(R=r[:9].reshape(3, 3).copy(), t=r[9:12].copy(), error=float(r[12]), compat=int(r[13]), Nd=int(r[14]), seconds=float(r[15]),
                dt_seconds=float(r[16]), t_rescaled=r[17:20].copy(), cube_evals=int(r[20]), trans_pops=int(r[21]), rot_pops=int(r[22]), status=int(r[23]))


def set_batch_speculation(depth):
    """Speculation depth of run_pair_files for the whole process (include/goicp_c.h: goicp_set_batch_speculation); -1 = class default."""
    load_library().goicp_set_batch_speculation(int(depth))


def batch_search_totals(reset=False):
    """Totals of the run_pair_files calls of this process (include/goicp_c.h: goicp_batch_search_totals)."""
    out = np.zeros(6, np.uint64)
    load_library().goicp_batch_search_totals(out.ctypes.data_as(C.c_void_p), int(bool(reset)))
    keys = ("registrations", "pops_launched", "pops_consumed", "jobs_launched", "inner_calls_consumed", "launches")
    return {k: int(v) for k, v in zip(keys, out)}


def set_batch_icp_one_launch(on):
    """ICP mode of run_pair_files for the whole process (include/goicp_c.h: goicp_set_batch_icp_one_launch); -1 = class default (off)."""
    load_library().goicp_set_batch_icp_one_launch(int(on))


def run_pair_files(base_dir, target_mol2, source_mol2, nd, config, output, pair, device_id=0, verbose=0):
    """The reference's main() on files (argv order of reference README.md:17)."""
    L = load_library()
    r = np.zeros(24)
    rc = L.goicp_run_pair_files(base_dir.encode(), target_mol2.encode(), source_mol2.encode(), int(nd), config.encode(), output.encode(),
                                int(pair), int(device_id), int(verbose), _ptr(r))
    if rc != GOICP_OK:
        raise EngineError("run_pair_files: %s" % STATUS.get(rc, rc))
    return _unpack24(r)


def run_txt_files(model_txt, data_txt, nd, config, output="", device_id=0, verbose=0):
    L = load_library()
    r = np.zeros(24)
    rc = L.goicp_run_txt_files(model_txt.encode(), data_txt.encode(), int(nd), config.encode(), output.encode(), int(device_id), int(verbose), _ptr(r))
    if rc != GOICP_OK:
        raise EngineError("run_txt_files: %s" % STATUS.get(rc, rc))
    return _unpack24(r)


def compute_rmsd(aligned_mol2, rot_mol2):
    return float(load_library().goicp_compute_rmsd(aligned_mol2.encode(), rot_mol2.encode()))


def apply_transformation_protein(output_txt, protein_mol2, out_mol2):
    return load_library().goicp_apply_transformation_protein(output_txt.encode(), protein_mol2.encode(), out_mol2.encode()) == GOICP_OK
