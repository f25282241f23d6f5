"""ORACLE (test infrastructure): write oracle/gen/odes_gen.c -- plain-C analytic ODE derivatives.

These come from the product's code generator and are linked into liboracle.so as ODE provider 1.
They are NOT the parity reference (provider 0, AD2, is, and tests/test_oracle.py checks provider 1
against it); they exist so that bench.py's cpu_baseline times straight-line analytic derivatives
rather than AD2's O(N^2)-per-operation arithmetic, i.e. a CPU path at least as fast as the reference's
expression-tree evaluation.
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))


def main():
    from asset_asrl_amd.ode import ODE_LIBRARY
    from asset_asrl_amd.vf.codegen import emit_c
    os.makedirs(os.path.join(HERE, "gen"), exist_ok=True)
    parts = []
    for name, cls in ODE_LIBRARY.items():
        src = emit_c(cls().derivatives(), f"ode_{name}")
        parts.append(src.replace("#include <math.h>\n", ""))
    text = "#include <math.h>\n" + "\n".join(parts)
    path = os.path.join(HERE, "gen", "odes_gen.c")
    if not os.path.exists(path) or open(path).read() != text:
        open(path, "w").write(text)
    print("wrote", path)


if __name__ == "__main__":
    main()
