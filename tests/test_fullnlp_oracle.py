"""The oracle's full NonLinearProgram restatement (oracle/fullnlp.cpp: objectives, equalities, inequalities, slacks) pinned
two ways on the CPU: the equality-only program reproduces oracle/nlp.cpp bit for bit (structure, locations, values), and
the full program's CSR values equal an independent dense assembly in numpy of the same per-application results."""
import numpy as np

from helpers import FullProblem, Workload


def test_equality_only_program_equals_the_single_constraint_nlp(oracle):
    w = Workload("reentry", "LGL5", 11, var_offset=2, con_offset=1, extra_vars=3)
    ref = w.oracle_nlp(oracle)
    n = oracle.FullNlp(w.n_primal, w.n_equal, 0)
    n.add(n.EQ, oracle.get_ode("reentry", 0), oracle.MODES["LGL5"], False, w.vindex, w.cindex)
    n.analyze()
    for a, b in zip(n.csr(), ref.csr()):
        np.testing.assert_array_equal(a, b)
    np.testing.assert_array_equal(n.kkt_locations(), ref.kkt_locations())
    for level, what in ((4, oracle.JAC_ADJGRAD_HESS), (0, oracle.CON), (1, oracle.CON_ADJGRAD), (2, oracle.JAC), (3, oracle.JAC_ADJGRAD)):
        val, PGX, AGX, FXE, FXI, vals = n.eval(level, 1.0, w.X, w.L, np.zeros(1))
        rFXE, rAGX, rvals = ref.eval(what, w.X, w.L)
        np.testing.assert_array_equal(FXE, rFXE)
        if level in (1, 3, 4):
            np.testing.assert_array_equal(AGX, rAGX)
        if rvals is not None:
            np.testing.assert_array_equal(vals, rvals)
        assert val == 0.0 and not PGX.any()


def test_full_program_values_equal_a_dense_assembly(oracle):
    p = FullProblem(nseg=7)
    n = p.oracle_nlp(oracle)
    P, S, E, I = p.n_primal, p.n_inequal, p.n_equal, p.n_inequal
    assert n.kkt_dim == P + S + E + I and n.num_solver_kkt == S + P + S + E + I
    val, PGX, AGX, FXE, FXI, vals = n.eval(4, p.obj_scale, p.X, p.LE, p.LI)
    # independent assembly: dense symmetric KKT from the per-application results, lower triangle
    K = np.zeros((n.kkt_dim, n.kkt_dim))
    rPGX, rAGX, rFXE, rFXI, rval = np.zeros(P), np.zeros(P), np.zeros(E), np.zeros(I), 0.0
    for kind, tag, V, Cx in p.functions:
        for a in range(V.shape[0]):
            x = p.X[V[a]]
            if tag == "defect":
                lam = p.LE[Cx[a]]
                fx, jx, gx, hx = oracle.defect_all(oracle.get_ode("reentry", 0), oracle.LGL5, x, lam)
            elif tag == "pathcon":
                lam = p.LE[Cx[a]]
                fx, jx, gx, hx = oracle.defect_all(oracle.get_ode("pathcon", 0), oracle.FUNCTION, x, lam)
            elif tag == "meshspacing":
                lam = p.LE[Cx[a]]
                fx, jx, gx, hx = oracle.lgl_mesh_spacing_all(3, x, lam)
            elif tag == "pairprod":
                lam = p.LI[Cx[a]]
                fx, jx, gx, hx = oracle.defect_all(oracle.get_ode("pairprod", 0), oracle.FUNCTION, x, lam)
            else:
                lam = np.array([p.obj_scale])
                fx, jx, gx, hx = oracle.lgl_integral_all(oracle.get_ode("integrand_quad2", 0), 3, 2, 0, x, lam)
            K[np.ix_(V[a], V[a])] += hx
            if kind == 0:
                rval += p.obj_scale * fx[0]
                np.add.at(rPGX, V[a], gx)
                continue
            np.add.at(rAGX, V[a], gx)
            rows = (P + S if kind == 1 else P + S + E) + Cx[a]
            K[np.ix_(rows, V[a])] += jx
            if kind == 1:
                rFXE[Cx[a]] += fx
            else:
                rFXI[Cx[a]] += fx
    sc = p.solver_coeffs
    for i in range(I):
        K[P + S + E + i, P + i] += sc[i]                                   # slack Jacobian ones
    K[np.arange(P), np.arange(P)] += sc[S:S + P]
    K[P + np.arange(S), P + np.arange(S)] += sc[S + P:S + P + S]
    K[P + S + np.arange(E), P + S + np.arange(E)] += sc[S + P + S:S + P + S + E]
    K[P + S + E + np.arange(I), P + S + E + np.arange(I)] += sc[S + P + S + E:]
    outer, inner = n.csr()
    got = np.zeros_like(K)
    for r in range(n.kkt_dim):
        for k in range(outer[r], outer[r + 1]):
            got[inner[k], r] = vals[k]                                      # CSR upper (r, c) = lower (c, r)
    low = np.tril(K)
    assert np.abs(got - low).max() <= 1e-12 * max(1.0, np.abs(low).max())
    assert np.count_nonzero(np.triu(got, 1)) == 0
    np.testing.assert_allclose(PGX, rPGX, rtol=0, atol=1e-12 * max(1.0, np.abs(rPGX).max()))
    np.testing.assert_allclose(AGX, rAGX, rtol=0, atol=1e-11 * max(1.0, np.abs(rAGX).max()))
    np.testing.assert_allclose(FXE, rFXE, rtol=0, atol=1e-13 * max(1.0, np.abs(rFXE).max()))
    np.testing.assert_allclose(FXI, rFXI, rtol=0, atol=1e-13)
    assert abs(val - rval) < 1e-12 * max(1.0, abs(rval))
