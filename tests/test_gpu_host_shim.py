"""GPU test of the C++ host shim: the Concept-style methods scatter into the same CSR values / RHS vectors as the
oracle's NonLinearProgram restatement."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from helpers import Workload, csr_locations, rel_err

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def shim(tmp_path_factory):
    pk = os.path.join(ROOT, "asset_asrl_amd")
    so = str(tmp_path_factory.mktemp("shim") / "shim_driver.so")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-fPIC", "-shared", os.path.join(ROOT, "tests", "host_shim_driver.cpp"),
                           "-o", so, "-L" + pk, "-lasset_host", "-lasset_hip", "-Wl,-rpath," + pk])
    return C.CDLL(so)


def shim_locations(shim, oracle, w, nlp, mode):
    """The constraint's own KKT space -- (row, col) per slot in the order of ITS blocks (BatchedDefectConstraint::getKKTSpace) --
    checked against the oracle's space as a multiset per application (the order is the function's private business:
    SolverInterfaceSpecs.h:41-92, NonLinearProgram.cpp:282-330), and turned into KKTLocations of the oracle's CSR matrix."""
    ip = C.POINTER(C.c_int)
    rows = np.zeros(nlp.num_user_kkt, dtype=np.int32)
    cols = np.zeros(nlp.num_user_kkt, dtype=np.int32)
    err = C.create_string_buffer(512)
    n = shim.shim_space(w.ode.encode(), oracle.MODES[mode], int(w.blocked), w.IR, w.OR, w.nseg, w.vindex.ctypes.data_as(ip),
                        w.cindex.ctypes.data_as(ip), w.n_primal, w.n_equal, rows.ctypes.data_as(ip), cols.ctypes.data_as(ip), err, 512)
    assert n == nlp.num_user_kkt, err.value
    rows_ref, cols_ref = nlp.kkt_coords()
    # analyzeSparsity keeps every slot as (row >= col) and files it in CSR row `col` (NonLinearProgram.cpp:282-307)
    lo, hi = np.minimum(rows, cols).astype(np.int64), np.maximum(rows, cols).astype(np.int64)
    per = nlp.num_user_kkt // w.nseg
    got = np.sort((hi * (1 << 31) + lo).reshape(w.nseg, per), axis=1)
    ref = np.sort((rows_ref[: nlp.num_user_kkt].astype(np.int64) * (1 << 31) + cols_ref[: nlp.num_user_kkt]).reshape(w.nseg, per), axis=1)
    np.testing.assert_array_equal(got, ref)
    outer, inner = nlp.csr()
    locs = nlp.kkt_locations().copy()                     # (the solver's own slots behind the user slots stay)
    locs[: nlp.num_user_kkt] = csr_locations(rows, cols, outer, inner)
    return locs, rows, cols


@pytest.mark.parametrize("ode,mode,blocked,nseg", [("reentry", "LGL7", False, 37), ("twobody_lt", "LGL5", True, 20),
                                                   ("betts_lowthrust", "LGL3", False, 9), ("reentry", "Trapezoidal", False, 11)])
@pytest.mark.parametrize("device_assembly", [False, True])
@pytest.mark.parametrize("shards", [0, 3])
def test_shim_matches_oracle_nlp(oracle, shim, ode, mode, blocked, nseg, device_assembly, shards):
    """shards = 3: the same object built as three in-process shards on device 0 (BatchedDefectConstraint's device-list
    constructor over asset_hip_defect_create_sharded) -- the ByApplication form of ConstraintFunction::thread_split."""
    shim.shim_set_shards(shards)
    w = Workload(ode, mode, nseg, blocked, var_offset=2, con_offset=1, extra_vars=3)
    nlp = w.oracle_nlp(oracle, threads=1)
    locs, rows_sp, cols_sp = shim_locations(shim, oracle, w, nlp, mode)
    ip, dp = C.POINTER(C.c_int), C.POINTER(C.c_double)
    for what in (oracle.JAC_ADJGRAD_HESS, oracle.CON, oracle.CON_ADJGRAD, oracle.JAC, oracle.JAC_ADJGRAD):
        FXE, AGX, vals = np.zeros(w.n_equal), np.zeros(w.n_primal), np.zeros(nlp.nnz)
        rows = np.zeros(nlp.num_user_kkt, dtype=np.int32)
        cols = np.zeros(nlp.num_user_kkt, dtype=np.int32)
        err = C.create_string_buffer(512)
        rc = shim.shim_run(ode.encode(), oracle.MODES[mode], int(w.blocked), w.IR, w.OR, w.nseg,
                           w.vindex.ctypes.data_as(ip), w.cindex.ctypes.data_as(ip), w.n_primal, w.n_equal, what,
                           w.X.ctypes.data_as(dp), w.L.ctypes.data_as(dp), locs.ctypes.data_as(ip),
                           rows.ctypes.data_as(ip), cols.ctypes.data_as(ip), vals.ctypes.data_as(dp),
                           FXE.ctypes.data_as(dp), AGX.ctypes.data_as(dp), err, 512,
                           C.c_longlong(nlp.nnz if device_assembly else 0))
        assert rc == 0, err.value
        np.testing.assert_array_equal(rows, rows_sp)      # (the same space every time it is asked for)
        np.testing.assert_array_equal(cols, cols_sp)
        rFXE, rAGX, rvals = nlp.eval(what, w.X, w.L)
        assert np.abs(FXE - rFXE).max() / max(1.0, np.abs(w.X).max()) < 1e-10
        if what in (oracle.CON_ADJGRAD, oracle.JAC_ADJGRAD, oracle.JAC_ADJGRAD_HESS):
            assert rel_err(AGX, rAGX) < 1e-8
        if what >= oracle.JAC:
            assert rel_err(vals, rvals) < 1e-8


@pytest.mark.parametrize("device_assembly", [False, True])
@pytest.mark.parametrize("shards", [0, 2])
def test_shim_rebind_and_deep_copy(oracle, shim, device_assembly, shards):
    """A constraint created for a smaller mesh, re-bound (the re-meshing step, ODEPhaseBase.cpp:1443-1542) and deep-copied
    (DeepCopySpecs.h:36-60; the original destroyed before the copy evaluates) scatters what the oracle's NLP does."""
    shim.shim_set_shards(shards)
    w = Workload("reentry", "LGL5", 29, var_offset=2, con_offset=1, extra_vars=3)
    nlp = w.oracle_nlp(oracle, threads=1)
    locs, _, _ = shim_locations(shim, oracle, w, nlp, "LGL5")
    ip, dp = C.POINTER(C.c_int), C.POINTER(C.c_double)
    FXE, AGX, vals = np.zeros(w.n_equal), np.zeros(w.n_primal), np.zeros(nlp.nnz)
    err = C.create_string_buffer(512)
    rc = shim.shim_rebind_run(b"reentry", oracle.MODES["LGL5"], 0, w.IR, w.OR, w.nseg, w.vindex.ctypes.data_as(ip),
                              w.cindex.ctypes.data_as(ip), w.n_primal, w.n_equal, 11, w.X.ctypes.data_as(dp),
                              w.L.ctypes.data_as(dp), locs.ctypes.data_as(ip), vals.ctypes.data_as(dp), FXE.ctypes.data_as(dp),
                              AGX.ctypes.data_as(dp), err, 512, C.c_longlong(nlp.nnz if device_assembly else 0))
    assert rc == 0, err.value
    rFXE, rAGX, rvals = nlp.eval(oracle.JAC_ADJGRAD_HESS, w.X, w.L)
    assert np.abs(FXE - rFXE).max() / max(1.0, np.abs(w.X).max()) < 1e-10
    assert rel_err(AGX, rAGX) < 1e-8 and rel_err(vals, rvals) < 1e-8


@pytest.mark.parametrize("ode,mode,blocked,nseg", [("reentry", "LGL7", False, 41), ("twobody_lt", "LGL5", True, 23),
                                                   ("betts_lowthrust", "Trapezoidal", False, 12)])
def test_kkt_assembly_matches_oracle_nlp(oracle, shim, ode, mode, blocked, nseg):
    """Product-side sparsity analysis and eval* drivers (host/kkt_assembly.h) against the oracle's restatement of
    NonLinearProgram: same CSR structure, same KKTLocations, same values for evalOCC / evalRHS / evalSOE / evalAUG / evalKKT."""
    shim.shim_set_shards(0)
    w = Workload(ode, mode, nseg, blocked, var_offset=2, con_offset=1, extra_vars=3)
    nlp = w.oracle_nlp(oracle, threads=1)
    r_outer, r_inner = nlp.csr()
    r_locs = nlp.kkt_locations()
    ip, dp = C.POINTER(C.c_int), C.POINTER(C.c_double)
    for what, okind in ((4, oracle.JAC_ADJGRAD_HESS), (0, oracle.CON), (1, oracle.CON_ADJGRAD), (2, oracle.JAC),
                        (3, oracle.JAC_ADJGRAD)):
        outer = np.zeros(nlp.kkt_dim + 1, dtype=np.int32)
        inner = np.zeros(nlp.nnz + 16, dtype=np.int32)
        locs = np.zeros(r_locs.size, dtype=np.int32)
        FXE, AGX, vals = np.ones(w.n_equal), np.ones(w.n_primal), np.ones(nlp.nnz)   # stale data: must be overwritten
        err = C.create_string_buffer(512)
        nnz = shim.assembly_run(ode.encode(), oracle.MODES[mode], int(w.blocked), w.IR, w.OR, w.nseg,
                                w.vindex.ctypes.data_as(ip), w.cindex.ctypes.data_as(ip), w.n_primal, w.n_equal, what,
                                w.X.ctypes.data_as(dp), w.L.ctypes.data_as(dp), outer.ctypes.data_as(ip),
                                inner.ctypes.data_as(ip), inner.size, locs.ctypes.data_as(ip), vals.ctypes.data_as(dp),
                                FXE.ctypes.data_as(dp), AGX.ctypes.data_as(dp), err, 512)
        assert nnz == nlp.nnz, err.value
        np.testing.assert_array_equal(outer, r_outer)
        np.testing.assert_array_equal(inner[:nnz], r_inner)
        # KKTLocations: the same locations per application, in the order of the function's own space (the device's block layout)
        nu = nlp.num_user_kkt
        np.testing.assert_array_equal(np.sort(locs[:nu].reshape(w.nseg, -1), axis=1), np.sort(r_locs[:nu].reshape(w.nseg, -1), axis=1))
        np.testing.assert_array_equal(locs[nu:], r_locs[nu:])
        rFXE, rAGX, rvals = nlp.eval(okind, w.X, w.L)
        assert np.abs(FXE - rFXE).max() / max(1.0, np.abs(w.X).max()) < 1e-10
        if what in (1, 3, 4):
            assert rel_err(AGX, rAGX) < 1e-8
        if what in (2, 3, 4):
            assert rel_err(vals, rvals) < 1e-8


@pytest.mark.parametrize("ode,blocked,nseg", [("reentry", False, 11), ("twobody_lt", True, 9), ("betts_lowthrust", False, 7)])
@pytest.mark.parametrize("device_assembly", [False, True])
def test_trapezoidal_hessian_sparsity_mask(oracle, shim, ode, blocked, nseg, device_assembly):
    """EnableHessianSparsity of the Trapezoidal defects (TrapezoidalDefects.h:39-141): the cross-node block of the adjoint Hessian
    claims no KKT slots and the fill steps over it -- the shim's KKT space, CSR locations and values against the oracle's
    restatement with the same switch, scattered on the host and assembled on the device."""
    shim.shim_set_shards(0)
    shim.shim_set_hessian_sparsity(1)
    try:
        w = Workload(ode, "Trapezoidal", nseg, blocked, var_offset=2, con_offset=1, extra_vars=3)
        nlp = w.oracle_nlp(oracle, threads=1, hessian_sparsity=True)
        dense = w.oracle_nlp(oracle, threads=1)
        assert nlp.num_user_kkt < dense.num_user_kkt
        locs, rows_sp, cols_sp = shim_locations(shim, oracle, w, nlp, "Trapezoidal")
        ip, dp = C.POINTER(C.c_int), C.POINTER(C.c_double)
        for what in (oracle.JAC_ADJGRAD_HESS, oracle.JAC_ADJGRAD):
            FXE, AGX, vals = np.zeros(w.n_equal), np.zeros(w.n_primal), np.zeros(nlp.nnz)
            rows = np.zeros(nlp.num_user_kkt, dtype=np.int32)
            cols = np.zeros(nlp.num_user_kkt, dtype=np.int32)
            err = C.create_string_buffer(512)
            rc = shim.shim_run(ode.encode(), oracle.MODES["Trapezoidal"], int(w.blocked), w.IR, w.OR, w.nseg,
                               w.vindex.ctypes.data_as(ip), w.cindex.ctypes.data_as(ip), w.n_primal, w.n_equal, what,
                               w.X.ctypes.data_as(dp), w.L.ctypes.data_as(dp), locs.ctypes.data_as(ip),
                               rows.ctypes.data_as(ip), cols.ctypes.data_as(ip), vals.ctypes.data_as(dp),
                               FXE.ctypes.data_as(dp), AGX.ctypes.data_as(dp), err, 512,
                               C.c_longlong(nlp.nnz if device_assembly else 0))
            assert rc == 0, err.value
            np.testing.assert_array_equal(rows, rows_sp)
            np.testing.assert_array_equal(cols, cols_sp)
            rFXE, rAGX, rvals = nlp.eval(what, w.X, w.L)
            assert np.abs(FXE - rFXE).max() / max(1.0, np.abs(w.X).max()) < 1e-10
            assert rel_err(AGX, rAGX) < 1e-8 and rel_err(vals, rvals) < 1e-8
    finally:
        shim.shim_set_hessian_sparsity(0)
