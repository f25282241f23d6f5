"""The collocation weight tables of the oracle (oracle/lgl_coeffs.h) and of the product (csrc/lgl_tables.h, host copy
through the C ABI) against tests/golden/lgl_tables.json -- the reference's own header
(/root/reference/src/OptimalControl/LGLCoeffs.h) parsed at fixture-generation time by
tests/golden/parse_lglcoeffs.py.  Bit for bit: a transcription error in either table fails here, independently of the
golden vectors (which take their weights from the same fixture, not from the oracle)."""
import json
import os

import numpy as np
import pytest

from asset_asrl_amd import _lib

HERE = os.path.dirname(os.path.abspath(__file__))
REF = json.load(open(os.path.join(HERE, "golden", "lgl_tables.json")))["tables"]
NAMES = {"tc": "CardinalSpacings", "s": "InteriorSpacings", "A": "Cardinal_XInterp_Weights",
         "B": "Cardinal_DXInterp_Weights", "U": "Cardinal_UPoly_Weights", "C": "Cardinal_XDef_Weights",
         "D": "Cardinal_DXDef_Weights", "E": "Interior_DXDef_Weights"}


@pytest.mark.parametrize("cs", [2, 3, 4])
def test_oracle_tables_equal_the_reference_header(oracle, cs):
    for which, name in NAMES.items():
        np.testing.assert_array_equal(oracle.lgl_table(cs, which), np.array(REF[str(cs)][name], dtype=float),
                                      err_msg=f"oracle/lgl_coeffs.h LGLCoeffs<{cs}>::{name}")


@pytest.mark.parametrize("cs", [2, 3, 4])
def test_product_tables_equal_the_reference_header(cs):
    for which, name in NAMES.items():
        np.testing.assert_array_equal(_lib.lgl_table(cs, which), np.array(REF[str(cs)][name], dtype=float),
                                      err_msg=f"csrc/lgl_tables.h LGLCoeffs<{cs}>::{name}")


def test_fixture_holds_the_identities_the_survey_checked():
    """rows of A and U sum to one, rows of C to zero; the full quadrature weights sum to two, the reduced ones to one."""
    for cs in (2, 3, 4):
        t = REF[str(cs)]
        assert np.allclose(np.sum(t["Cardinal_XInterp_Weights"], axis=1), 1.0, atol=2e-15)
        assert np.allclose(np.sum(t["Cardinal_UPoly_Weights"], axis=1), 1.0, atol=2e-15)
        assert np.allclose(np.sum(t["Cardinal_XDef_Weights"], axis=1), 0.0, atol=5e-15)
        assert abs(sum(t["Cardinal_Integral_Weights"]) + sum(t["Interior_Integral_Weights"]) - 2.0) < 1e-14
        assert abs(sum(t["Reduced_Integral_Weights"]) - 1.0) < 1e-13
