"""The other per-segment functions of a phase -- mesh spacing, control spline, segment quadrature
(MeshSpacingConstraints.h:8-193, LGLControlSplines.h:64-315, LGLIntegrals.h:9-52):

* the oracle's closed forms (oracle/pathfuncs.cpp) against tests/golden/pathfuncs.npz -- the value formulas evaluated and
  differentiated exactly in 50-digit arithmetic (tests/golden/make_golden_pathfuncs.py), weights from the parsed reference
  header;
* the product's weight tables (asset_asrl_amd/pathfuncs.py, synth.py) against that header, bit for bit;
* the product's DSL definitions (host evaluation of the value) against the golden values.
The device side is checked against the oracle in tests/test_gpu_function.py."""
import json
import os

import numpy as np
import pytest

from asset_asrl_amd import pathfuncs, synth, vf

HERE = os.path.dirname(os.path.abspath(__file__))
G = np.load(os.path.join(HERE, "golden", "pathfuncs.npz"))
REF = json.load(open(os.path.join(HERE, "golden", "lgl_tables.json")))["tables"]
S4 = REF["4"]["CardinalSpacings"][1]


def _oracle_fn(ob, name):
    if name == "single_mesh_spacing":
        return lambda x, l: ob.single_mesh_spacing_all(S4, x, l, scale=2.5)
    if name.startswith("lgl_mesh_spacing"):
        cs = int(name[-1])
        return lambda x, l: ob.lgl_mesh_spacing_all(cs, x, l)
    if name.startswith("control_spline"):
        cs, usize = int(name[14]), int(name[16])
        return lambda x, l: ob.control_spline_all(cs, usize, x, l)
    cs = int(name[12])
    xv, pv, integ = (2, 0, "integrand_quad2") if name.endswith("quad2") else (3, 1, "integrand_powp")
    o = ob.get_ode(integ, 0)
    return lambda x, l: ob.lgl_integral_all(o, cs, xv, pv, x, l)


def _product_fn(name):
    if name == "single_mesh_spacing":
        return pathfuncs.SingleMeshSpacing(S4, 2.5)
    if name.startswith("lgl_mesh_spacing"):
        return pathfuncs.LGLMeshSpacing(int(name[-1]))
    if name.startswith("control_spline"):
        return pathfuncs.LGLControlSpline(int(name[14]), int(name[16]))
    cs = int(name[12])
    if name.endswith("quad2"):
        g = vf.Arguments(2)
        return pathfuncs.LGLIntegral(g.coeff(1) * g.coeff(1) + g.coeff(0), cs, 2)
    g = vf.Arguments(4)
    integ = g.coeff(3) * g.coeff(0) * g.coeff(0) + vf.sin(g.coeff(1)) * g.coeff(2) \
        + vf.exp(-1.0 * (g.coeff(0) * g.coeff(2))) / (1.0 + g.coeff(3) * g.coeff(3))
    return pathfuncs.LGLIntegral(integ, cs, 3, 1)


CASES = sorted({k[:-2] for k in G.files if k.endswith("_x")})


def _rel(a, b):
    return float(np.abs(a - b).max() / max(1.0, np.abs(b).max()))


@pytest.mark.parametrize("name", CASES)
def test_oracle_closed_forms_match_the_golden_vectors(oracle, name):
    fn = _oracle_fn(oracle, name)
    for k in range(G[name + "_x"].shape[0]):
        fx, jx, gx, hx = fn(G[name + "_x"][k], G[name + "_lam"][k])
        assert _rel(fx, G[name + "_fx"][k]) < 1e-13
        assert _rel(jx, G[name + "_jx"][k]) < 1e-12
        assert _rel(gx, G[name + "_gx"][k]) < 1e-12
        assert _rel(hx, G[name + "_hx"][k]) < 1e-11
        assert np.abs(hx - hx.T).max() < 1e-12 * max(1.0, np.abs(hx).max())


@pytest.mark.parametrize("name", CASES)
def test_product_definitions_give_the_golden_values(name):
    F = _product_fn(name)
    for k in range(G[name + "_x"].shape[0]):
        assert _rel(np.atleast_1d(F.compute(G[name + "_x"][k])), G[name + "_fx"][k]) < 1e-13


def test_product_weight_tables_equal_the_reference_header(oracle):
    for cs in (2, 3, 4):
        t = REF[str(cs)]
        np.testing.assert_array_equal(np.array(pathfuncs._REDUCED_INTEGRAL_WEIGHTS[cs]), np.array(t["Reduced_Integral_Weights"]))
        np.testing.assert_array_equal(oracle.aux_table(cs, "reduced_integral"), np.array(t["Reduced_Integral_Weights"]))
        np.testing.assert_array_equal(synth._TC[cs], np.array(t["CardinalSpacings"]))
        if cs >= 3:
            np.testing.assert_array_equal(np.array(pathfuncs._UONE[cs]), np.array(t["UOneSpline_Weights"]))
            np.testing.assert_array_equal(np.array(pathfuncs._UZERO[cs]), np.array(t["UZeroSpline_Weights"]))
            for j in range(cs - 2):
                np.testing.assert_array_equal(oracle.aux_table(cs, f"uone{j}"), np.array(t["UOneSpline_Weights"][j]))
                np.testing.assert_array_equal(oracle.aux_table(cs, f"uzero{j}"), np.array(t["UZeroSpline_Weights"][j]))
