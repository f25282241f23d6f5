"""Floating-point operation counts of one defect evaluation (value + Jacobian + adjoint gradient + adjoint Hessian) per mesh segment.

Two counts, both in flops (an FMA = 2):

* :func:`survey_dense_flops` -- SURVEY.md section 8(d)'s figure for the reference's algorithm as it is written
  (/root/reference/src/OptimalControl/LGLDefects.h:414-506): per interior point the dense products ``J_rows += (h E J^) DI``
  (``2 n (q+p) IR``), ``M = (h E H^) DI`` (``2 IR (q+p)^2``) and ``adjhess += DI^T M`` (``2 IR^2 (q+p)``), plus the ODE calls.
* :func:`sparse_useful_flops` -- the same mathematics restated for the SPARSE ``DI_i``: column ``(j, cc)`` of
  ``DI_i = d(x^_i, tau_i, u^_i, P)/dz`` is ``w_i(j, cc) e_cc + h B_ij dfdy_j[:, cc]`` (LGLDefects.h:417-458), so with the
  ODE's structural sparsity (``nnz`` of the columns of ``df/dy`` and of ``lam^T d2f/dy2``) a row of ``H`` costs one FMA per
  structural entry, not per dense entry.  This is what the row-wise kernels (csrc/defect_rows.h for the wide shapes,
  csrc/defect_rowdpp.h for the narrow ones) evaluate, counted for the entries they must produce -- the lower triangle of H, the
  Jacobian, the adjoint gradient -- and nothing for masked lanes, padding or the upper triangle.  It is the numerator of the
  FP64 roofline of a shape whose arithmetic intensity is above the ridge (bench.py: ``roofline.fp64``).
"""
from __future__ import annotations

from typing import Dict


def survey_dense_flops(n: int, m: int, p: int, cs: int, ode_ops: int = 0) -> int:
    """K [2 n (q+p) IR + 2 IR (q+p)^2 + 2 IR^2 (q+p)] + (CS + K) * cost(f, df, lam^T d2f)."""
    K, q = cs - 1, n + 1 + m
    N, IR = q + p, cs * q + p
    return K * (2 * n * N * IR + 2 * IR * N * N + 2 * IR * IR * N) + (cs + K) * int(ode_ops)


def sparse_useful_flops(derivs, cs: int, blocked: bool = False) -> Dict[str, int]:
    """FP64 flops per segment of the node-wise sparse form (see the module docstring).  ``derivs``: the ODE's
    ``vf.codegen.OdeDerivatives`` (structure of J and of the lower triangle of H, operation count of its body)."""
    from .vf.ir import GRAPH as G
    n = derivs.xv
    m, p = (0, derivs.uv + derivs.pv) if blocked else (derivs.uv, derivs.pv)
    K, q = cs - 1, n + 1 + m
    N, IR, OR, P0 = q + p, cs * q + p, (cs - 1) * n, cs * q
    z = G.zero
    nzJ = [[derivs.J[a][b] is not z for b in range(N)] for a in range(n)]
    nzH = [[(derivs.H[max(a, b)][min(a, b)] is not z) for b in range(N)] for a in range(N)]
    colJ = [sum(1 for a in range(n) if nzJ[a][b]) for b in range(N)]         # structural entries of column b of df/dy
    nnzH_full = sum(1 for a in range(N) for b in range(N) if nzH[a][b])
    NZJ = sum(colJ)

    def ycol(c):            # ODE input (column of df/dy) behind block column c, and whether it is a node's column
        if c < P0:
            return c % q, True
        return q + (c - P0), False
    fma = 0
    # ---- rows of H (lane <-> row r of the lower triangle): d_i = DI_i[:, r], M_i = hE_i H^_i d_i, BM_j = sum_i B_ij M_i[0:n]
    for r in range(IR):
        yr, node = ycol(r)
        nd = colJ[yr] * (1 if node else cs)                  # entries of d_i beside the unit entry (a parameter row: every node's column)
        fma += K * (nd + n)                                  # d_i: h B J terms and the time rows
        fma += K * nnzH_full                                 # M_i: one FMA per structural entry of the (symmetric) interior Hessian
        fma += 2 * K * N                                     # the g^ terms: rank-2 row part, and g^ . d_i of the time partial
        fma += cs * K * n + cs * n + 2 * K                   # BM_j, FB, the time sums
        for c in range(r + 1):                               # the row's entries H(r, c), c <= r
            yc, cnode = ycol(c)
            fma += (K + colJ[yc] + 2) if cnode else (cs * colJ[yc] + 2 * cs)
    # ---- rows of [J ; g^T] (OR Jacobian rows + the gradient row): M_i = hE_i J^_i^T l_i, BM_j = sum_i (B_ij M_i + D_ij l_i)
    for _ in range(OR + 1):
        fma += K * NZJ + K * n                               # M_i and the f^ . l sums
        fma += 2 * cs * K * n + cs * n + 2 * K               # BM_j, FB, time sums
        for c in range(IR):
            yc, cnode = ycol(c)
            fma += (2 * K + colJ[yc]) if cnode else cs * colJ[yc]
    fma += OR * (2 * cs + 1)                                 # the defect values
    ode_ops = derivs.stats()["ops_fjgh"]
    algebra = 2 * fma
    return {"algebra": algebra, "ode": (cs + K) * ode_ops, "total": algebra + (cs + K) * ode_ops, "IR": IR, "OR": OR,
            "dense_survey": survey_dense_flops(n, m, p, cs, ode_ops)}
