"""The row-wise dense part (csrc/defect_rowdpp.h) issues v_fmac_f64_dpp through inline assembly, where the compiler's hazard
recogniser does not look: the built objects are checked instead (tools/isa_dpp_hazard.py) -- no vector-ALU write of a DPP source
within two wait states of its use, no v_cmpx within five."""
import glob
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def test_no_dpp_hazard_in_the_built_objects():
    import isa_dpp_hazard as H
    objs = sorted(glob.glob(os.path.join(ROOT, "asset_asrl_amd", "csrc", "obj", "tu_*.o")))
    if not objs or not os.path.exists(os.path.join(H.LLVM, "llvm-objdump")):
        pytest.skip("no built objects here (python -c 'import __graft_entry__ as g; g.build()' writes them)")
    total, bad = 0, []
    for o in objs:
        f, n = H.check(H.disassemble(o), os.path.basename(o))
        bad += f
        total += n
    assert not bad, "\n".join(bad[:20])
    assert total > 1000          # the row-wise kernels are in there


def test_no_dpp_hazard_in_the_run_time_compiled_modules():
    """The same inline assembly is instantiated in every module compiled at run time for a user ODE -- another register allocation
    each time: the cached modules of the in-tree cache (what build() compiled for the GPU tests) are scanned like the static objects."""
    import isa_dpp_hazard as H
    mods = [f for f in H.default_files() if f.endswith((".rtc", ".so"))]
    if not mods or not os.path.exists(os.path.join(H.LLVM, "llvm-objdump")):
        pytest.skip("no cached run-time modules here (build() writes them)")
    total, bad = 0, []
    for o in mods:
        f, n = H.check(H.disassemble(o), os.path.relpath(o, ROOT))
        bad += f
        total += n
    assert not bad, "\n".join(bad[:20])
    assert total > 1000          # user ODEs of narrow shapes take the row-wise part too


def test_the_checker_sees_a_planted_hazard():
    import isa_dpp_hazard as H
    text = """0000 <k>:
        v_mov_b32_e32 v0, v9
        v_add_f64 v[2:3], v[4:5], v[6:7]
        v_fmac_f64_dpp v[8:9], v[2:3], v[10:11] row_newbcast:3 row_mask:0xf bank_mask:0xf
        s_nop 1
        v_fmac_f64_dpp v[8:9], v[2:3], v[10:11] row_newbcast:4 row_mask:0xf bank_mask:0xf
"""
    f, n = H.check(text, "planted")
    assert n == 2 and len(f) == 1 and "v_add_f64" in f[0]
