"""In-process sharded constraint at the C ABI (include/asset_hip.h: asset_hip_defect_create_sharded) -- the reference's
thread_split (ConstraintFunction.h:55-62, IndexingData.h:117-146, NonLinearProgram.cpp:71-109, 519-526) with a device handle
per chunk.  One GPU is visible on the test box, so the shards are N handles on device 0: the blocks must be bitwise those of the
single handle, the assembled values bitwise for a phase without parameters and equal to rounding with them."""
import numpy as np
import pytest

from asset_asrl_amd import _lib
from asset_asrl_amd.evaluator import (CON, CON_ADJGRAD, JAC, JAC_ADJGRAD, JAC_ADJGRAD_HESS, DefectEvaluator,
                                      ShardedDefectEvaluator)
from asset_asrl_amd.indexing import kkt_slot_locations, thread_split
from helpers import Workload

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("ode,mode,blocked,nseg,nsh", [("reentry", "LGL7", False, 1001, 3), ("twobody_lt", "LGL5", True, 257, 4),
                                                       ("brachistochrone", "LGL3", False, 5, 8), ("reentry", "Trapezoidal", False, 64, 2)])
def test_sharded_blocks_are_bitwise_the_single_handle(ode, mode, blocked, nseg, nsh):
    w = Workload(ode, mode, nseg, blocked)
    one = DefectEvaluator(ode, mode, blocked, w.vindex, w.cindex, w.n_primal, w.n_equal)
    sh = ShardedDefectEvaluator(ode, mode, blocked, w.vindex, w.cindex, w.n_primal, w.n_equal, [0] * nsh)
    assert [(f, c) for f, c, _ in sh.ranges] == thread_split(nseg, nsh)          # the ByApplication rule, fewer shards than asked when nseg < nsh
    for what in (CON, CON_ADJGRAD, JAC, JAC_ADJGRAD, JAC_ADJGRAD_HESS):
        L = w.L if what in (CON_ADJGRAD, JAC_ADJGRAD, JAC_ADJGRAD_HESS) else None
        a, b = one.eval(what, w.X, L), sh.eval(what, w.X, L)
        for x, y in zip(a, b):
            assert (x is None) == (y is None)
            if x is not None:
                np.testing.assert_array_equal(x, y)
    # page-locked outputs kept across calls (the calling thread enqueues every shard before any copy has run) give the same bits as the
    # pageable ones above (a host thread per shard inside the library); the blocks arrive in the handles' layout
    sh.pin_outputs()
    ref = one.eval(JAC_ADJGRAD_HESS, w.X, w.L)
    got = sh.eval(JAC_ADJGRAD_HESS, w.X, w.L)
    assert got[2].shape == (nseg, sh.KSTRIDE) and sh.eval(CON, w.X)[0] is got[0]
    got = sh.eval(JAC_ADJGRAD_HESS, w.X, w.L)
    for x, y in zip(ref, (got[0], got[1], sh.kkt_to_reference(got[2]))):
        np.testing.assert_array_equal(x, y)
    sh.close()
    one.close()


def test_sharded_calls_leave_the_current_device_alone_and_do_not_serialise():
    """The asset_hip_sharded_* entry points walk the shards' devices (hipSetDevice is per thread) and put the caller's current device
    back; and N handles on ONE device take about the time of one handle for the same phase -- the work and the bytes are the same --
    not N times it (page-locked outputs: every shard is enqueued before any is waited for)."""
    import ctypes as C
    import time
    hip = C.CDLL("libamdhip64.so")
    dev = C.c_int(-1)
    assert hip.hipGetDevice(C.byref(dev)) == 0
    before = dev.value
    w = Workload("reentry", "LGL7", 4000)
    times = {}
    for nsh in (1, 4):
        sh = ShardedDefectEvaluator("reentry", "LGL7", False, w.vindex, w.cindex, w.n_primal, w.n_equal, [0] * nsh).pin_outputs()
        for _ in range(3):
            sh.eval(JAC_ADJGRAD_HESS, w.X, w.L)
        t0 = time.perf_counter()
        for _ in range(10):
            sh.eval(JAC_ADJGRAD_HESS, w.X, w.L)
        times[nsh] = (time.perf_counter() - t0) / 10
        sh.close()
        assert hip.hipGetDevice(C.byref(dev)) == 0 and dev.value == before
    assert times[4] < 2.0 * times[1] + 1e-3, times


@pytest.mark.parametrize("ode,mode,blocked,exact", [("reentry", "LGL5", False, True), ("twobody_lt", "LGL5", True, False)])
def test_sharded_assembly_matches_the_single_handle(ode, mode, blocked, exact):
    nseg = 300
    w = Workload(ode, mode, nseg, blocked)
    locs, nnz = kkt_slot_locations(w.vindex, w.cindex, w.n_primal)
    one = DefectEvaluator(ode, mode, blocked, w.vindex, w.cindex, w.n_primal, w.n_equal)
    one.set_kkt_map(locs, nnz)
    sh = ShardedDefectEvaluator(ode, mode, blocked, w.vindex, w.cindex, w.n_primal, w.n_equal, [0, 0, 0]).set_kkt_map(locs, nnz)
    for what in (JAC_ADJGRAD_HESS, JAC):
        L = w.L if what == JAC_ADJGRAD_HESS else None
        v1, v2 = np.zeros(nnz), np.zeros(nnz)
        r1 = one.eval_assembled(what, w.X, L, v1)
        fx2, agx2 = sh.eval_assembled(what, w.X, L, v2)
        np.testing.assert_array_equal(r1[0], fx2)
        if exact:
            np.testing.assert_array_equal(v1, v2)
        else:       # entries between phase parameters: one running sum there, a partial sum per shard here
            assert np.abs(v1 - v2).max() <= 1e-13 * max(1.0, np.abs(v1).max())
    sh.close()
    one.close()


def test_sharded_create_rejects_bad_arguments():
    w = Workload("brachistochrone", "LGL3", 4)
    with pytest.raises(_lib.AssetHipError):
        ShardedDefectEvaluator("nonexistent", "LGL3", False, w.vindex, w.cindex, w.n_primal, w.n_equal, [0, 0])
    with pytest.raises(_lib.AssetHipError):
        ShardedDefectEvaluator("brachistochrone", "LGL3", False, w.vindex, w.cindex, w.n_primal, w.n_equal, [0, 99])
