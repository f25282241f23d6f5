"""CPU tests of the drop-in boundary: libasset_hip.so loads, exports every symbol include/asset_hip.h declares,
answers the introspection calls, and refuses to evaluate without a device (no CPU fallback)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from asset_asrl_amd import _lib, synth
from asset_asrl_amd.build import dims

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "asset_hip.h")).read()
    declared = set(re.findall(r"\b(asset_hip_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"asset_hip_defect_eval"} - {"asset_hip_defect_eval"}  # keep all
    assert declared == set(_lib.SYMBOLS), (declared ^ set(_lib.SYMBOLS))
    L = C.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(L, name), name


def test_introspection_matches_survey_sizes():
    assert set(_lib.ode_names()) >= {"brachistochrone", "reentry", "twobody_lt", "betts_lowthrust", "synthetic32"}
    for name, sizes in synth.ODE_SIZES.items():
        assert _lib.ode_sizes(name) == sizes
    # SURVEY.md section 8 size table
    for ode, cs, blocked, IR, OR, NKKT in [("brachistochrone", 2, False, 10, 3, 85), ("betts_lowthrust", 3, False, 34, 14, 1071),
                                           ("reentry", 4, False, 32, 15, 1008), ("twobody_lt", 3, True, 24, 12, 588),
                                           ("synthetic32", 4, False, 132, 96, 21450)]:
        d = dims(*synth.ODE_SIZES[ode], cs, blocked)
        assert (d["IR"], d["OR"], d["NKKT"]) == (IR, OR, NKKT)
    assert _lib.has_kernel("reentry", _lib.LGL7, False)
    assert _lib.has_kernel("twobody_lt", _lib.LGL5, True)
    assert not _lib.has_kernel("nonexistent", _lib.LGL3, False)


def test_kkt_block_layouts_without_a_device():
    """asset_hip_kkt_layout (no handle, no device): every compiled (ode, transcription, control mode) exports the order of its KKT
    blocks -- every entry of the block exactly once, padding marked -1 -- equal to the host mirror of the kernels' arithmetic
    (evaluator.kkt_layout_table) and to the sizes build.dims() plans with.  The order is the function's own business
    (DenseFunctionBase.h:1097-1129 against NonLinearProgram.cpp:282-330): narrow shapes J | H with 128-byte-aligned regions,
    wide shapes the reference's order."""
    from asset_asrl_amd.evaluator import kkt_layout_table, reference_slot_order
    seen = set()
    for ode in _lib.ode_names():
        xv, uv, pv = _lib.ode_sizes(ode)
        for mode, mid in _lib.MODES.items():
            for blocked in (False, True):
                if mode == "Function" or not _lib.has_kernel(ode, mid, blocked):
                    continue
                kl, nk, stride, rows, cols = _lib.kkt_layout(ode, mid, blocked)
                d = dims(xv, uv, pv, synth.MODE_CS[mode], blocked, trap=(mode == "Trapezoidal"))
                IR, OR = d["IR"], d["OR"]
                assert (nk, kl, stride) == (d["NKKT"], d["KL"], d["KSTRIDE"])
                st2, r2, c2 = kkt_layout_table(IR, OR, kl)
                assert st2 == stride and np.array_equal(rows, r2) and np.array_equal(cols, c2)
                real = rows >= 0
                rr, cc = reference_slot_order(IR, OR)
                assert sorted(zip(rows[real].tolist(), cols[real].tolist())) == sorted(zip(rr.tolist(), cc.tolist()))
                if kl == 1:
                    assert stride % 16 == 0 and np.all(rows[:IR * OR] >= IR)      # the Jacobian first, on a line of its own
                else:
                    assert stride == nk and np.array_equal(rows, rr) and np.array_equal(cols, cc)
                seen.add(kl)
    assert seen == {0, 1}
    with pytest.raises(_lib.AssetHipError):
        _lib.kkt_layout("nonexistent", _lib.LGL3, False)


def test_lgl_tables_bitwise_equal_to_oracle(oracle):
    for cs in (2, 3, 4):
        for which in ("tc", "s", "A", "B", "U", "C", "D", "E"):
            np.testing.assert_array_equal(_lib.lgl_table(cs, which), oracle.lgl_table(cs, which))
        np.testing.assert_array_equal(_lib.lgl_table(cs, "tc"), synth._TC[cs])


def test_create_rejects_bad_arguments_and_has_no_cpu_fallback():
    from asset_asrl_amd.evaluator import DefectEvaluator
    from helpers import Workload
    w = Workload("brachistochrone", "LGL3", 4)
    with pytest.raises(_lib.AssetHipError, match="no device code"):
        DefectEvaluator("nonexistent", "LGL3", False, w.vindex, w.cindex, w.n_primal, w.n_equal)
    bad = w.vindex.copy()
    bad[2, 3] = w.n_primal
    with pytest.raises(_lib.AssetHipError, match="out of range"):
        DefectEvaluator("brachistochrone", "LGL3", False, bad, w.cindex, w.n_primal, w.n_equal)
    if _lib.device_count() == 0:
        with pytest.raises(_lib.AssetHipError, match="no HIP device"):
            DefectEvaluator("brachistochrone", "LGL3", False, w.vindex, w.cindex, w.n_primal, w.n_equal)


def test_scheduler_choice_by_translation_unit(monkeypatch):
    """asset_asrl_amd/build.py: tu_flags -- the max-ilp scheduler for the Reentry and Trapezoidal units only (measured per unit,
    DESIGN 4.0a), never for capi.hip, and off altogether with ASSET_HIP_NO_MAX_ILP."""
    from asset_asrl_amd import build as b
    monkeypatch.delenv("ASSET_HIP_NO_MAX_ILP", raising=False)
    assert b.tu_flags("/x/gen/tu_reentry_lgl4_0.hip") == b.MAX_ILP
    assert b.tu_flags("/x/gen/tu_twobody_lt_trap_1.hip") == b.MAX_ILP
    assert b.tu_flags("/x/gen/tu_twobody_lt_lgl3_1.hip") == []
    assert b.tu_flags("/x/gen/tu_synthetic32_lgl4_0.hip") == []
    assert b.tu_flags("/x/csrc/capi.hip") == []
    monkeypatch.setenv("ASSET_HIP_NO_MAX_ILP", "1")
    assert b.tu_flags("/x/gen/tu_reentry_lgl4_0.hip") == []
    # the flags are part of an object's digest: a unit is recompiled when its scheduler changes
    monkeypatch.delenv("ASSET_HIP_NO_MAX_ILP")
    import tempfile, os
    with tempfile.NamedTemporaryFile("w", suffix=".h", delete=False) as f:
        f.write("x")
    try:
        assert b._digest([f.name], b.MAX_ILP) != b._digest([f.name], [])
    finally:
        os.unlink(f.name)
