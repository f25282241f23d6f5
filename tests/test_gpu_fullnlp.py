"""The product's sparse assembly with the whole KKT layout (host/kkt_assembly.h: objectives, equalities, inequalities,
slacks, solver slots) on a phase that hands the solver everything a phase can -- defect equality, path equality,
mesh-spacing equality, pair-wise path inequality, integral objective, all evaluated on the device -- entry by entry
against the oracle's restatement of NonLinearProgram (oracle/fullnlp.cpp), for all five evaluation entry points."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from helpers import FullProblem, rel_err

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class FnDesc(C.Structure):
    _fields_ = [("kind", C.c_int), ("name", C.c_char_p), ("mode", C.c_int), ("blocked", C.c_int), ("ir", C.c_int),
                ("orr", C.c_int), ("nappl", C.c_int), ("vindex", C.POINTER(C.c_int)), ("cindex", C.POINTER(C.c_int)),
                ("consts", C.POINTER(C.c_double)), ("nconst", C.c_int)]


@pytest.fixture(scope="module")
def shim(tmp_path_factory):
    pk = os.path.join(ROOT, "asset_asrl_amd")
    so = str(tmp_path_factory.mktemp("shim") / "shim_driver.so")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-fPIC", "-shared", os.path.join(ROOT, "tests", "host_shim_driver.cpp"),
                           "-o", so, "-L" + pk, "-lasset_host", "-lasset_hip", "-Wl,-rpath," + pk])
    return C.CDLL(so)


@pytest.mark.parametrize("nseg", [13, 1])
def test_full_kkt_layout_matches_the_oracle(oracle, shim, nseg):
    from asset_asrl_amd import _lib, jit
    p = FullProblem(nseg=nseg)
    ref = p.oracle_nlp(oracle)
    names = {"defect": ("reentry", _lib.LGL5)}
    for tag, (fn, jname) in p.product_functions().items():
        names[tag] = (jit.ensure_function(fn, jname), _lib.FUNCTION)
    ip, dp = C.POINTER(C.c_int), C.POINTER(C.c_double)
    keep, descs = [], (FnDesc * len(p.functions))()
    for k, (kind, tag, V, Cx) in enumerate(p.functions):
        v = np.ascontiguousarray(V, dtype=np.int32)
        c = np.ascontiguousarray(Cx if Cx is not None else np.zeros((V.shape[0], 1)), dtype=np.int32)
        keep += [v, c]
        descs[k] = FnDesc(kind, names[tag][0].encode(), names[tag][1], 0, v.shape[1], c.shape[1], v.shape[0],
                          v.ctypes.data_as(ip), c.ctypes.data_as(ip), None, 0)
    r_outer, r_inner = ref.csr()
    r_locs = ref.kkt_locations()
    sc = np.ascontiguousarray(p.solver_coeffs)
    for level in (4, 0, 1, 2, 3):
        outer = np.zeros(ref.kkt_dim + 1, dtype=np.int32)
        inner = np.zeros(ref.nnz + 16, dtype=np.int32)
        locs = np.zeros(r_locs.size + 16, dtype=np.int32)
        val = C.c_double(-1.0)
        PGX, AGX = np.ones(p.n_primal), np.ones(p.n_primal)               # stale data: must be overwritten
        FXE, FXI, vals = np.ones(p.n_equal), np.ones(p.n_inequal), np.ones(ref.nnz)
        err = C.create_string_buffer(512)
        nnz = shim.fullnlp_run(descs, len(p.functions), p.n_primal, p.n_equal, p.n_inequal, level, C.c_double(p.obj_scale),
                               p.X.ctypes.data_as(dp), p.LE.ctypes.data_as(dp), p.LI.ctypes.data_as(dp), sc.ctypes.data_as(dp),
                               outer.ctypes.data_as(ip), inner.ctypes.data_as(ip), inner.size, locs.ctypes.data_as(ip), locs.size,
                               C.byref(val), PGX.ctypes.data_as(dp), AGX.ctypes.data_as(dp), FXE.ctypes.data_as(dp),
                               FXI.ctypes.data_as(dp), vals.ctypes.data_as(dp), err, 512)
        assert nnz == ref.nnz, err.value
        np.testing.assert_array_equal(outer, r_outer)
        np.testing.assert_array_equal(inner[:nnz], r_inner)
        # KKTLocations: the same multiset of user locations (each function lists its slots in the order of its own blocks), the
        # solver's own slots behind them unchanged
        nu = ref.num_user_kkt
        np.testing.assert_array_equal(np.sort(locs[:nu]), np.sort(r_locs[:nu]))
        np.testing.assert_array_equal(locs[nu: r_locs.size], r_locs[nu:])
        rval, rPGX, rAGX, rFXE, rFXI, rvals = ref.eval(level, p.obj_scale, p.X, p.LE, p.LI)
        scale = max(1.0, np.abs(p.X).max())
        assert np.abs(FXE - rFXE).max() / scale < 1e-10 and np.abs(FXI - rFXI).max() / scale < 1e-10
        if level != 2:
            assert abs(val.value - rval) < 1e-10 * max(1.0, abs(rval))
        if level in (1, 3, 4):
            assert rel_err(AGX, rAGX) < 1e-8 and rel_err(PGX, rPGX) < 1e-8
        if level >= 2:
            assert rel_err(vals, rvals) < 1e-8
