"""ORACLE (test infrastructure): ctypes bindings to liboracle.so.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

FUNCTION, TRAPEZOIDAL, LGL3, LGL5, LGL7 = 0, 1, 2, 3, 4
MODES = {"Function": 0, "Trapezoidal": 1, "LGL3": 2, "LGL5": 3, "LGL7": 4}
CON, CON_ADJGRAD, JAC, JAC_ADJGRAD, JAC_ADJGRAD_HESS = range(5)

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)


class OdeStruct(C.Structure):
    _fields_ = [("xv", C.c_int), ("uv", C.c_int), ("pv", C.c_int),
                ("f", C.c_void_p), ("fj", C.c_void_p), ("fjgh", C.c_void_p), ("ctx", C.c_void_p)]


def build(force: bool = False) -> str:
    so = os.path.join(_HERE, "liboracle.so")
    if force or not os.path.exists(so):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return so


def cpu_model() -> str:
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown CPU"


def build_native() -> str:
    """The same sources built the way the reference builds its own (-O2 -march=native -ffast-math,
    /root/reference/CMakeLists.txt:55,160) -- for bench.py's cpu_baseline leg only; the parity tests keep the strict-IEEE
    library.  -march=native code must be compiled on the machine that runs it: the file name carries a hash of this
    host's CPU model and flags, so a library built elsewhere is never loaded."""
    import hashlib
    ident = cpu_model()
    try:
        ident += next(ln for ln in open("/proc/cpuinfo") if ln.startswith("flags"))
    except (OSError, StopIteration):
        pass
    so = os.path.join(_HERE, f"liboracle_native_{hashlib.sha256(ident.encode()).hexdigest()[:10]}.so")
    srcs = [os.path.join(_HERE, f) for f in ("defect.cpp", "nlp.cpp", "mesh.cpp", "pathfuncs.cpp", "fullnlp.cpp", "odes.cpp")]
    gen = os.path.join(_HERE, "gen", "odes_gen.c")
    gen4 = os.path.join(_HERE, "gen", "odes_gen4.c")          # the same bodies over four segments at once (batch4.h)
    deps = srcs + [gen, gen4] + [os.path.join(_HERE, f) for f in ("odes.h", "interp_table.h", "ad2.h", "lgl_coeffs.h", "oracle.h", "batch4.h")]
    if os.path.exists(so) and all(os.path.getmtime(so) >= os.path.getmtime(d) for d in deps if os.path.exists(d)):
        return so
    flags = ["-O2", "-march=native", "-ffast-math", "-fPIC"]
    tmp = so + ".build"
    os.makedirs(tmp, exist_ok=True)
    jobs = []
    for src in srcs:
        o = os.path.join(tmp, os.path.basename(src) + ".o")
        jobs.append((["g++", "-std=c++17", "-pthread", "-Wno-unused-function", "-Wno-psabi"] + flags + ["-c", src, "-o", o], o))
    for g in (gen, gen4):
        if os.path.exists(g):
            o = os.path.join(tmp, os.path.basename(g) + ".o")
            jobs.append((["gcc", "-Wno-psabi"] + flags + ["-c", g, "-o", o], o))
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as pool:     # (the two generated files take ~10 s each)
        list(pool.map(lambda j: subprocess.check_call(j[0]), jobs))
    objs = [o for _, o in jobs]
    subprocess.check_call(["g++", "-shared", "-pthread", "-o", so] + objs)
    return so


def use_native():
    """Switch this process to the natively optimised library (bench.py's cpu_baseline leg)."""
    global _LIB
    _LIB = None
    lib(build_native())


def lib(path=None):
    global _LIB
    if _LIB is None:
        L = C.CDLL(path or build())
        L.oracle_get_ode.argtypes = [C.c_char_p, C.c_int, C.POINTER(OdeStruct)]
        L.oracle_set_synthetic32.argtypes = [_dp]
        L.oracle_table_sizes.argtypes = [C.c_int, _ip, _ip, _ip, _ip]
        L.oracle_table_data.argtypes = [C.c_int, _dp, _dp, _dp]
        L.oracle_table_interp.argtypes = [C.c_int, C.c_double, _dp, _dp, _dp]
        L.oracle_defect_sizes.argtypes = [C.c_int] * 5 + [_ip, _ip]
        L.oracle_defect_compute.argtypes = [C.POINTER(OdeStruct), C.c_int, C.c_int, _dp, _dp]
        L.oracle_defect_jacobian.argtypes = [C.POINTER(OdeStruct), C.c_int, C.c_int, _dp, _dp, _dp]
        L.oracle_defect_all.argtypes = [C.POINTER(OdeStruct), C.c_int, C.c_int, _dp, _dp, _dp, _dp, _dp, _dp]
        L.oracle_lgl_table.argtypes = [C.c_int, C.c_char_p]
        L.oracle_lgl_table.restype = _dp
        L.oracle_phase_num_vars.argtypes = [C.c_int] * 7
        L.oracle_phase_defect_index.argtypes = [C.c_int] * 9 + [_ip, _ip]
        L.oracle_nlp_create.argtypes = [C.POINTER(OdeStruct), C.c_int, C.c_int, C.c_int, _ip, _ip, C.c_int, C.c_int,
                                        C.c_int]
        L.oracle_nlp_create.restype = C.c_void_p
        L.oracle_nlp_create_ex.argtypes = [C.POINTER(OdeStruct), C.c_int, C.c_int, C.c_int, _ip, _ip, C.c_int, C.c_int, C.c_int, C.c_int]
        L.oracle_nlp_create_ex.restype = C.c_void_p
        L.oracle_nlp_destroy.argtypes = [C.c_void_p]
        L.oracle_nlp_set_batch4.argtypes = [C.c_void_p, C.c_int]
        L.oracle_nlp_set_batch4.restype = C.c_int
        for fn in ("oracle_nlp_kkt_dim", "oracle_nlp_nnz", "oracle_nlp_num_user_kkt"):
            getattr(L, fn).argtypes = [C.c_void_p]
        L.oracle_nlp_csr.argtypes = [C.c_void_p, _ip, _ip]
        L.oracle_nlp_kkt_locations.argtypes = [C.c_void_p, _ip]
        L.oracle_nlp_kkt_coords.argtypes = [C.c_void_p, _ip, _ip]
        L.oracle_nlp_eval.argtypes = [C.c_void_p, C.c_int, _dp, _dp, _dp, _dp, _dp]
        L.oracle_nlp_eval_blocks.argtypes = [C.c_void_p, C.c_int, _dp, _dp, _dp, _dp, _dp]
        L.oracle_single_mesh_spacing_all.argtypes = [C.c_double, C.c_double, _dp, _dp, _dp, _dp, _dp, _dp]
        L.oracle_lgl_mesh_spacing_all.argtypes = [C.c_int, _dp, _dp, _dp, _dp, _dp, _dp]
        L.oracle_control_spline_all.argtypes = [C.c_int, C.c_int, C.c_int, _dp, _dp, _dp, _dp, _dp, _dp]
        L.oracle_lgl_integral_all.argtypes = [C.POINTER(OdeStruct), C.c_int, C.c_int, C.c_int, _dp, _dp, _dp, _dp, _dp, _dp]
        L.oracle_aux_table.argtypes = [C.c_int, C.c_char_p]
        vp = C.c_void_p
        L.oracle_fullnlp_create.argtypes = [C.c_int] * 3
        L.oracle_fullnlp_create.restype = vp
        L.oracle_fullnlp_destroy.argtypes = [vp]
        L.oracle_fullnlp_add.argtypes = [vp, C.c_int, C.POINTER(OdeStruct), C.c_int, C.c_int, C.c_int, _ip, _ip]
        L.oracle_fullnlp_add_integral.argtypes = [vp, C.c_int, C.POINTER(OdeStruct), C.c_int, C.c_int, C.c_int, C.c_int, _ip, _ip]
        L.oracle_fullnlp_add_mesh_spacing.argtypes = [vp, C.c_int, C.c_int, C.c_int, _ip, _ip]
        L.oracle_fullnlp_add_control_spline.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, _ip, _ip]
        L.oracle_fullnlp_add_single_mesh_spacing.argtypes = [vp, C.c_int, _dp, C.c_double, C.c_int, _ip, _ip]
        for fn in ("analyze", "kkt_dim", "nnz", "num_user_kkt", "num_solver_kkt"):
            getattr(L, "oracle_fullnlp_" + fn).argtypes = [vp]
        L.oracle_fullnlp_csr.argtypes = [vp, _ip, _ip]
        L.oracle_fullnlp_kkt_locations.argtypes = [vp, _ip]
        L.oracle_fullnlp_solver_coeffs.argtypes = [vp]
        L.oracle_fullnlp_solver_coeffs.restype = _dp
        L.oracle_fullnlp_eval.argtypes = [vp, C.c_int, C.c_double, _dp, _dp, _dp, _dp, _dp, _dp, _dp, _dp, _dp]
        L.oracle_aux_table.restype = _dp
        _LIB = L
    return _LIB


def _d(a):
    return None if a is None else a.ctypes.data_as(_dp)


def _i(a):
    return a.ctypes.data_as(_ip)


def synthetic32_coeffs(n: int = 32, seed: int = 32) -> np.ndarray:
    rng = np.random.default_rng(seed)
    return np.concatenate([rng.uniform(0.5, 1.5, n) for _ in range(3)])


def get_ode(name: str, provider: int = 0) -> OdeStruct:
    o = OdeStruct()
    if name == "synthetic32":
        abc = np.ascontiguousarray(synthetic32_coeffs())
        lib().oracle_set_synthetic32(_d(abc))
    rc = lib().oracle_get_ode(name.encode(), provider, C.byref(o))
    if rc:
        raise KeyError(f"oracle ODE {name!r} provider {provider}: rc={rc}")
    return o


def table(which: int):
    """(ts, vs[vlen, n], dvs_dts[vlen, n], even, cubic) of table `which` of the `tabulated` ODE (interp_table.h)."""
    sz = [C.c_int() for _ in range(4)]
    if lib().oracle_table_sizes(which, *[C.byref(x) for x in sz]):
        raise KeyError(which)
    n, vlen, even, cubic = (x.value for x in sz)
    ts, vs, ds = np.empty(n), np.empty((vlen, n)), np.empty((vlen, n))
    lib().oracle_table_data(which, _d(ts), _d(vs), _d(ds))
    return ts, vs, ds, bool(even), bool(cubic)


def table_interp(which: int, t: float):
    """(v, dv/dt, d2v/dt2) of table `which` at t."""
    vlen = table(which)[1].shape[0]
    out = [np.empty(vlen) for _ in range(3)]
    lib().oracle_table_interp(which, float(t), *[_d(x) for x in out])
    return out


def lgl_table(cs: int, which: str) -> np.ndarray:
    p = lib().oracle_lgl_table(cs, which.encode())
    K = cs - 1
    shape = {"tc": (cs,), "s": (K,), "E": (K,)}.get(which)
    if shape is not None:
        return np.array([p[i] for i in range(shape[0])])
    return np.array([[p[i * 4 + j] for j in range(cs)] for i in range(K)])


def defect_sizes(mode: int, ode: OdeStruct, blocked: bool = False):
    ir, orr = C.c_int(), C.c_int()
    rc = lib().oracle_defect_sizes(mode, ode.xv, ode.uv, ode.pv, int(blocked), C.byref(ir), C.byref(orr))
    if rc:
        raise ValueError("bad mode")
    return ir.value, orr.value


def defect_compute(ode, mode, x, blocked=False):
    ir, orr = defect_sizes(mode, ode, blocked)
    x = np.ascontiguousarray(x, dtype=float)
    assert x.size == ir
    fx = np.zeros(orr)
    lib().oracle_defect_compute(C.byref(ode), mode, int(blocked), _d(x), _d(fx))
    return fx


def defect_jacobian(ode, mode, x, blocked=False):
    ir, orr = defect_sizes(mode, ode, blocked)
    x = np.ascontiguousarray(x, dtype=float)
    fx, jx = np.zeros(orr), np.zeros((ir, orr))
    lib().oracle_defect_jacobian(C.byref(ode), mode, int(blocked), _d(x), _d(fx), _d(jx))
    return fx, jx.T.copy()


def defect_all(ode, mode, x, lam, blocked=False):
    """(fx, jx[OR,IR], gx, hx[IR,IR]) -- the reference's ``computeall(x,l)`` tuple."""
    ir, orr = defect_sizes(mode, ode, blocked)
    x = np.ascontiguousarray(x, dtype=float)
    lam = np.ascontiguousarray(lam, dtype=float)
    assert x.size == ir and lam.size == orr
    fx, jx, gx, hx = np.zeros(orr), np.zeros((ir, orr)), np.zeros(ir), np.zeros((ir, ir))
    lib().oracle_defect_all(C.byref(ode), mode, int(blocked), _d(x), _d(lam), _d(fx), _d(jx), _d(gx), _d(hx))
    return fx, jx.T.copy(), gx, hx.T.copy()


def mesh_error_deboor(ode, mode, traj, blocked=False):
    """(tsnd[nb+1], mesh_errors[xv, nb+1], mesh_dist[xv, nb+1]) -- ODEPhase.h:442-585."""
    traj = np.ascontiguousarray(traj, dtype=float)
    nnodes = traj.shape[0]
    cs = 2 if mode == TRAPEZOIDAL else mode
    nb = (nnodes - 1) // (cs - 1)
    tsnd, err, dist = np.zeros(nb + 1), np.zeros((nb + 1, ode.xv)), np.zeros((nb + 1, ode.xv))
    rc = lib().oracle_mesh_error_deboor(C.byref(ode), mode, int(blocked), _d(traj), nnodes, _d(tsnd), _d(err), _d(dist))
    if rc:
        raise ValueError(f"oracle_mesh_error_deboor rc={rc}")
    return tsnd, err.T.copy(), dist.T.copy()


def phase_num_vars(xv, uv, pv, spv, cs, nd, blocked):
    return lib().oracle_phase_num_vars(xv, uv, pv, spv, cs, nd, int(blocked))


def phase_defect_index(xv, uv, pv, spv, cs, nd, blocked, var_offset=0, con_offset=0):
    if blocked:
        ir = cs * (xv + 1) + uv + pv
    else:
        ir = cs * (xv + 1 + uv) + pv
    orr = (cs - 1) * xv
    V = np.zeros((nd, ir), dtype=np.int32)
    Cx = np.zeros((nd, orr), dtype=np.int32)
    got = lib().oracle_phase_defect_index(xv, uv, pv, spv, cs, nd, int(blocked), var_offset, con_offset, _i(V), _i(Cx))
    assert got == ir
    return V, Cx  # row V = application V (i.e. transposed w.r.t. the reference's column-per-application)


class Nlp:
    def __init__(self, ode, mode, blocked, vindex, cindex, primal, equal, threads=1, hessian_sparsity=False):
        self.ode = ode
        self.vindex = np.ascontiguousarray(vindex, dtype=np.int32)
        self.cindex = np.ascontiguousarray(cindex, dtype=np.int32)
        self.nappl, self.ir = self.vindex.shape
        self.orr = self.cindex.shape[1]
        self.primal, self.equal = primal, equal
        self.h = lib().oracle_nlp_create_ex(C.byref(ode), mode, int(blocked), self.nappl, _i(self.vindex),
                                            _i(self.cindex), primal, equal, threads, 1 if hessian_sparsity else 0)
        if not self.h:
            raise ValueError("oracle_nlp_create failed")
        self.kkt_dim = lib().oracle_nlp_kkt_dim(self.h)
        self.nnz = lib().oracle_nlp_nnz(self.h)
        self.num_user_kkt = lib().oracle_nlp_num_user_kkt(self.h)
        self.nkkt = self.num_user_kkt // max(self.nappl, 1)

    def __del__(self):
        h, self.h = getattr(self, "h", None), None
        if h and lib is not None:          # module globals are gone at interpreter exit
            lib().oracle_nlp_destroy(h)

    def set_batch4(self, on=True) -> bool:
        """LGL evalKKT four applications at a time (oracle/batch4.h; the reference's SuperScalar loop).  False when the
        loaded library / ODE provider has no four-wide body."""
        return lib().oracle_nlp_set_batch4(self.h, int(bool(on))) == 0

    def csr(self):
        outer = np.zeros(self.kkt_dim + 1, dtype=np.int32)
        inner = np.zeros(self.nnz, dtype=np.int32)
        lib().oracle_nlp_csr(self.h, _i(outer), _i(inner))
        return outer, inner

    def kkt_locations(self):
        locs = np.zeros(self.num_user_kkt + self.kkt_dim, dtype=np.int32)
        lib().oracle_nlp_kkt_locations(self.h, _i(locs))
        return locs

    def kkt_coords(self):
        r = np.zeros(self.num_user_kkt + self.kkt_dim, dtype=np.int32)
        c = np.zeros_like(r)
        lib().oracle_nlp_kkt_coords(self.h, _i(r), _i(c))
        return r, c

    def eval(self, what, X, LE=None):
        X = np.ascontiguousarray(X, dtype=float)
        LE = None if LE is None else np.ascontiguousarray(LE, dtype=float)
        FXE = np.zeros(self.equal)
        AGX = np.zeros(self.primal)
        vals = np.zeros(self.nnz) if what >= JAC else None
        rc = lib().oracle_nlp_eval(self.h, what, _d(X), _d(LE), _d(FXE), _d(AGX), _d(vals))
        if rc:
            raise RuntimeError(f"oracle_nlp_eval rc={rc}")
        return FXE, AGX, vals

    def eval_blocks(self, what, X, LE=None):
        X = np.ascontiguousarray(X, dtype=float)
        LE = None if LE is None else np.ascontiguousarray(LE, dtype=float)
        fx = np.zeros((self.nappl, self.orr))
        agx = np.zeros((self.nappl, self.ir))
        kkt = np.zeros((self.nappl, self.nkkt)) if what >= JAC else None
        rc = lib().oracle_nlp_eval_blocks(self.h, what, _d(X), _d(LE), _d(fx), _d(agx), _d(kkt))
        if rc:
            raise RuntimeError(f"oracle_nlp_eval_blocks rc={rc}")
        return fx, agx, kkt


# ---- the other per-segment functions of a phase (oracle/pathfuncs.cpp) ------------------------------------------------
def _all(call, irr, orr, x, lam):
    """-> (fx[orr], jx[orr, irr], gx[irr], hx[irr, irr])"""
    x = np.ascontiguousarray(x, dtype=float)
    lam = np.ascontiguousarray(lam, dtype=float)
    assert x.size == irr and lam.size == orr, (x.size, irr, lam.size, orr)
    fx, jx, gx, hx = np.zeros(orr), np.zeros((irr, orr)), np.zeros(irr), np.zeros((irr, irr))
    rc = call(_d(x), _d(lam), _d(fx), _d(jx), _d(gx), _d(hx))
    if rc:
        raise ValueError(f"oracle path function rc={rc}")
    return fx, jx.T.copy(), gx, hx.T.copy()


def single_mesh_spacing_all(cardinal_spacing, x, lam, scale=1.0):
    return _all(lambda *a: lib().oracle_single_mesh_spacing_all(float(cardinal_spacing), float(scale), *a), 3, 1, x, lam)


def lgl_mesh_spacing_all(cs, x, lam):
    return _all(lambda *a: lib().oracle_lgl_mesh_spacing_all(cs, *a), cs, cs - 2, x, lam)


def control_spline_all(cs, usize, x, lam, order=0):
    o = order or cs - 2
    return _all(lambda *a: lib().oracle_control_spline_all(cs, usize, o, *a), (2 * cs - 1) * (usize + 1), usize * o, x, lam)


def lgl_integral_all(integrand: OdeStruct, cs, xv, pv, x, lam):
    return _all(lambda *a: lib().oracle_lgl_integral_all(C.byref(integrand), cs, xv, pv, *a), cs * (xv + 1) + pv, 1, x, lam)


def aux_table(cs, which):
    p = lib().oracle_aux_table(cs, which.encode())
    return np.array([p[i] for i in range(cs)]) if p else None


class FullNlp:
    """Objectives + equalities + inequalities with slacks: the whole KKT layout (oracle/fullnlp.cpp)."""
    OBJ, EQ, IQ = 0, 1, 2

    def __init__(self, primal, equal, inequal):
        self.primal, self.equal, self.inequal = primal, equal, inequal
        self.h = lib().oracle_fullnlp_create(primal, equal, inequal)
        self._keep = []

    def __del__(self):
        h, self.h = getattr(self, "h", None), None
        if h and lib is not None:
            lib().oracle_fullnlp_destroy(h)

    def _tables(self, vindex, cindex):
        v = np.ascontiguousarray(vindex, dtype=np.int32)
        c = np.ascontiguousarray(cindex if cindex is not None else np.zeros((v.shape[0], 1)), dtype=np.int32)
        self._keep += [v, c]
        return v, c

    def add(self, kind, ode, mode, blocked, vindex, cindex=None):
        v, c = self._tables(vindex, cindex)
        self._keep.append(ode)
        assert lib().oracle_fullnlp_add(self.h, kind, C.byref(ode), mode, int(blocked), v.shape[0], _i(v), _i(c)) >= 0

    def add_integral(self, kind, integrand, cs, xv, pv, vindex, cindex=None):
        v, c = self._tables(vindex, cindex)
        self._keep.append(integrand)
        assert lib().oracle_fullnlp_add_integral(self.h, kind, C.byref(integrand), cs, xv, pv, v.shape[0], _i(v), _i(c)) >= 0

    def add_mesh_spacing(self, kind, cs, vindex, cindex):
        v, c = self._tables(vindex, cindex)
        assert lib().oracle_fullnlp_add_mesh_spacing(self.h, kind, cs, v.shape[0], _i(v), _i(c)) >= 0

    def add_control_spline(self, kind, cs, usize, vindex, cindex):
        v, c = self._tables(vindex, cindex)
        assert lib().oracle_fullnlp_add_control_spline(self.h, kind, cs, usize, v.shape[0], _i(v), _i(c)) >= 0

    def add_single_mesh_spacing(self, kind, spacings, vindex, cindex, scale=1.0):
        v, c = self._tables(vindex, cindex)
        sp = np.ascontiguousarray(spacings, dtype=float)
        assert lib().oracle_fullnlp_add_single_mesh_spacing(self.h, kind, _d(sp), float(scale), v.shape[0], _i(v), _i(c)) >= 0

    def analyze(self):
        L = lib()
        L.oracle_fullnlp_analyze(self.h)
        self.kkt_dim, self.nnz = L.oracle_fullnlp_kkt_dim(self.h), L.oracle_fullnlp_nnz(self.h)
        self.num_user_kkt, self.num_solver_kkt = L.oracle_fullnlp_num_user_kkt(self.h), L.oracle_fullnlp_num_solver_kkt(self.h)

    def csr(self):
        outer, inner = np.zeros(self.kkt_dim + 1, dtype=np.int32), np.zeros(self.nnz, dtype=np.int32)
        lib().oracle_fullnlp_csr(self.h, _i(outer), _i(inner))
        return outer, inner

    def kkt_locations(self):
        locs = np.zeros(self.num_user_kkt + self.num_solver_kkt, dtype=np.int32)
        lib().oracle_fullnlp_kkt_locations(self.h, _i(locs))
        return locs

    def set_solver_coeffs(self, coeffs):
        p = lib().oracle_fullnlp_solver_coeffs(self.h)
        for i, v in enumerate(np.asarray(coeffs, dtype=float)):
            p[i] = v

    def eval(self, level, obj_scale, X, LE, LI):
        """-> (val, PGX, AGX, FXE, FXI, vals)"""
        X, LE, LI = (np.ascontiguousarray(a, dtype=float) for a in (X, LE, LI))
        val = C.c_double(0.0)
        PGX, AGX = np.zeros(self.primal), np.zeros(self.primal)
        FXE, FXI, vals = np.zeros(self.equal), np.zeros(max(self.inequal, 1)), np.zeros(self.nnz)
        rc = lib().oracle_fullnlp_eval(self.h, level, float(obj_scale), _d(X), _d(LE), _d(LI),
                                       C.cast(C.byref(val), _dp), _d(PGX), _d(AGX), _d(FXE), _d(FXI), _d(vals))
        assert rc == 0
        return val.value, PGX, AGX, FXE, FXI[: self.inequal], vals
