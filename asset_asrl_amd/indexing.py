"""Phase variable layout and the defect constraint's index tables (host side, set-up time).

Mirror of the parts of the reference's ``PhaseIndexer`` / ``SolverIndexingData`` that the batched
defect evaluator needs (/root/reference/src/OptimalControl/PhaseIndexer.h:63-99,168-197,
PhaseIndexer.cpp:132-189,361-391,412-488; /root/reference/src/VectorFunctions/IndexingData.h:30-210):

* non-blocked  X = [node_0(q) ... node_{S-1}(q) ; P(p) ; StaticP],  q = Xv+1+Uv,  S = (CS-1)*D + 1
* blocked      X = [node_0(Xv+1) ... node_{S-1}(Xv+1) ; u_seg0(Uv) ... u_seg{D-1}(Uv) ; P ; StaticP]
* DefectPath column V = nodes V*(CS-1) ... V*(CS-1)+CS-1 then P; Cindex column V = OR consecutive rows.

Index arrays are int32 and stored application-major (row V = application V), which is the reference's
column-major [rows x applications] matrix seen from C.
"""
from __future__ import annotations

from typing import Sequence

import numpy as np


class PhaseIndexer:
    def __init__(self, Xv: int, Uv: int = 0, OPv: int = 0, SPv: int = 0):
        self.xv, self.uv, self.pv, self.spv = int(Xv), int(Uv), int(OPv), int(SPv)
        self.numDefects = 0
        self.DefectCardinalStates = 0
        self.BlockedControls = False
        self.var_offset = 0
        self.con_offset = 0

    # ---- sizes ---------------------------------------------------------------------
    def XVars(self):
        return self.xv

    def UVars(self):
        return self.uv

    def PVars(self):
        return self.pv

    def StatPVars(self):
        return self.spv

    def XtVars(self):
        return self.xv + 1

    def XtUVars(self):
        return self.xv + 1 + self.uv

    def set_dimensions(self, DCS: int, Dnum: int, BlockCon: bool):
        if DCS < 2 or Dnum < 1:
            raise ValueError("need at least 2 cardinal states and 1 defect")
        self.numDefects = int(Dnum)
        self.DefectCardinalStates = int(DCS)
        self.numStates = (DCS - 1) * Dnum + 1
        self.BlockedControls = bool(BlockCon) and self.uv > 0
        if self.BlockedControls:
            self.numPhaseVars = self.numStates * self.XtVars() + Dnum * self.uv + self.pv + self.spv
        else:
            self.numPhaseVars = self.numStates * self.XtUVars() + self.pv + self.spv
        self.ODEParamLoc0 = self.numPhaseVars - self.pv - self.spv
        self.StaticParamLoc0 = self.numPhaseVars - self.spv
        self.numPhaseEqCons = (DCS - 1) * self.xv * Dnum

    def begin_indexing(self, var_offset: int = 0, con_offset: int = 0):
        self.var_offset, self.con_offset = int(var_offset), int(con_offset)

    def getXTUVarLoc(self, vloc: int, State: int, Defect: int | None = None) -> int:
        o = self.var_offset
        if self.BlockedControls:
            if vloc < self.XtVars():
                return o + vloc + State * self.XtVars()
            if Defect is None:
                Defect = min(State // (self.DefectCardinalStates - 1), self.numDefects - 1)
            return o + self.XtVars() * self.numStates + (vloc - self.XtVars()) + Defect * self.uv
        return o + vloc + State * self.XtUVars()

    # ---- defect constraint tables -------------------------------------------------------
    def defect_sizes(self):
        cs = self.DefectCardinalStates
        ir = cs * self.XtVars() + self.uv + self.pv if self.BlockedControls else cs * self.XtUVars() + self.pv
        return ir, (cs - 1) * self.xv

    def make_defect_Vindex_Cindex(self):
        """(Vindex[nseg, IR], Cindex[nseg, OR]) for DefectPath / BlockDefectPath."""
        cs, D = self.DefectCardinalStates, self.numDefects
        ir, orr = self.defect_sizes()
        o = self.var_offset
        seg = np.arange(D, dtype=np.int64)[:, None]
        p0 = o + self.ODEParamLoc0
        if self.BlockedControls:
            xt = self.XtVars()
            nodes = o + (seg * (cs - 1)) * xt + np.arange(cs * xt)[None, :]
            u = o + xt * self.numStates + seg * self.uv + np.arange(self.uv)[None, :]
            par = np.broadcast_to(p0 + np.arange(self.pv)[None, :], (D, self.pv))
            V = np.concatenate([nodes, u, par], axis=1)
        else:
            xtu = self.XtUVars()
            nodes = o + (seg * (cs - 1)) * xtu + np.arange(cs * xtu)[None, :]
            par = np.broadcast_to(p0 + np.arange(self.pv)[None, :], (D, self.pv))
            V = np.concatenate([nodes, par], axis=1)
        Cx = self.con_offset + seg * orr + np.arange(orr)[None, :]
        assert V.shape == (D, ir)
        return np.ascontiguousarray(V, dtype=np.int32), np.ascontiguousarray(Cx, dtype=np.int32)

    # ---- region tables for plain functions -------------------------------------------------
    REGIONS = ("Front", "Back", "FrontandBack", "BackandFront", "Path", "InnerPath", "NodalPath", "PairWisePath",
               "FrontNodalBackPath", "DefectPath", "DefectPairWisePath", "Params", "ODEParams", "StaticParams")

    def make_Vindex_Cindex(self, region: str, xtuv=(), odepv=(), statpv=(), orows: int = 0, next_cloc: int | None = None):
        """Index tables of a function applied over a phase region (PhaseIndexer.cpp:132-360, PhaseRegionFlags of
        OptimalControlFlags.h:7-25): one application per column of the reference's tables = one row here.  Input of
        an application: the chosen state/time/control variables ``xtuv`` at the region's state(s), then the chosen
        ODE parameters and static parameters.  Returns ``(Vindex[napp, IR], Cindex[napp, orows], next_cloc)``; constraint
        rows are numbered consecutively from ``next_cloc`` (default: after the phase's own rows so far).  A function
        of controls only in a BlockConstant phase is applied once per defect (PhaseIndexer.cpp:153-157, 231-243)."""
        xtuv = [int(v) for v in xtuv]
        odepv = [int(v) for v in odepv]
        statpv = [int(v) for v in statpv]
        if any(v < 0 or v >= self.XtUVars() for v in xtuv) or any(v < 0 or v >= self.pv for v in odepv) or \
                any(v < 0 or v >= self.spv for v in statpv):
            raise ValueError("variable index outside the phase's dimensions")
        cs, D, S = self.DefectCardinalStates, self.numDefects, self.numStates
        only_u = bool(xtuv) and all(v >= self.XtVars() for v in xtuv) and self.BlockedControls
        o = self.var_offset
        par = [o + self.ODEParamLoc0 + v for v in odepv] + [o + self.StaticParamLoc0 + v for v in statpv]

        def at(state):
            return [self.getXTUVarLoc(v, state) for v in xtuv]
        first, last = at(0), at(S - 1)
        nodal = [i * (cs - 1) for i in range(D + 1)]
        if region in ("Params", "ODEParams", "StaticParams"):
            rows = [par]
        elif region == "Front":
            rows = [first + par]
        elif region == "Back":
            rows = [last + par]
        elif region == "FrontandBack":
            rows = [first + last + par]
        elif region == "BackandFront":
            rows = [last + first + par]
        elif region == "Path":
            states = nodal[:D] if only_u else range(S)
            rows = [at(k) + par for k in states]
        elif region == "InnerPath":
            states = nodal[1:D - 1] if only_u else range(1, S - 1)
            rows = [at(k) + par for k in states]
        elif region == "NodalPath":
            states = nodal[:D] if only_u else nodal
            rows = [at(k) + par for k in states]
        elif region == "PairWisePath":
            pairs = list(zip(nodal[:D - 1], nodal[1:D])) if only_u else [(k, k + 1) for k in range(S - 1)]
            rows = [at(a) + at(b) + par for a, b in pairs]
        elif region == "FrontNodalBackPath":
            states = nodal[1:D - 1] if only_u else nodal[1:D]
            rows = [first + at(k) + last + par for k in states]
        elif region == "DefectPath":            # the cs states of every defect (PhaseIndexer.cpp:361-372)
            rows = [[self.getXTUVarLoc(v, i * (cs - 1) + j, i) for j in range(cs) for v in xtuv] + par for i in range(D)]
        elif region == "DefectPairWisePath":    # the 2cs-1 states of every pair of adjacent defects (:392-403)
            rows = [[self.getXTUVarLoc(v, i * (cs - 1) + j, i) for j in range(2 * cs - 1) for v in xtuv] + par
                    for i in range(D - 1)]
        else:
            raise ValueError(f"unknown phase region {region!r}; one of {self.REGIONS}")
        V = np.asarray(rows, dtype=np.int32).reshape(len(rows), -1)
        c0 = self.con_offset + self.numPhaseEqCons if next_cloc is None else int(next_cloc)
        Cx = (c0 + np.arange(len(rows) * orows, dtype=np.int64)).reshape(len(rows), orows).astype(np.int32)
        return np.ascontiguousarray(V), np.ascontiguousarray(Cx), c0 + len(rows) * orows

    # ---- trajectory <-> solver vector ------------------------------------------------------
    def makeSolverInput(self, ActiveTraj: Sequence[np.ndarray], ActiveStaticParams=None) -> np.ndarray:
        T = np.asarray(ActiveTraj, dtype=float)
        if T.shape[0] != self.numStates:
            raise ValueError(f"trajectory has {T.shape[0]} states, phase needs {self.numStates}")
        V = np.zeros(self.numPhaseVars)
        cs, D = self.DefectCardinalStates, self.numDefects
        if self.BlockedControls:
            xt = self.XtVars()
            V[: self.numStates * xt] = T[:, :xt].ravel()
            u0 = self.numStates * xt
            V[u0:u0 + D * self.uv] = T[0:(cs - 1) * D:(cs - 1), xt:xt + self.uv].ravel()
        else:
            xtu = self.XtUVars()
            V[: self.numStates * xtu] = T[:, :xtu].ravel()
        if self.pv:
            V[self.ODEParamLoc0:self.ODEParamLoc0 + self.pv] = T[0, -self.pv:]
        if self.spv:
            V[self.StaticParamLoc0:] = np.asarray(ActiveStaticParams, dtype=float)
        return V

    def collectSolverOutput(self, Vars: np.ndarray):
        cs, D = self.DefectCardinalStates, self.numDefects
        out = np.zeros((self.numStates, self.XtUVars() + self.pv))
        if self.BlockedControls:
            xt = self.XtVars()
            out[:, :xt] = Vars[: self.numStates * xt].reshape(self.numStates, xt)
            U = Vars[self.numStates * xt: self.numStates * xt + D * self.uv].reshape(D, self.uv)
            unum = np.minimum(np.arange(self.numStates) // (cs - 1), D - 1)
            out[:, xt:xt + self.uv] = U[unum]
        else:
            xtu = self.XtUVars()
            out[:, :xtu] = Vars[: self.numStates * xtu].reshape(self.numStates, xtu)
        if self.pv:
            out[:, -self.pv:] = Vars[self.ODEParamLoc0:self.ODEParamLoc0 + self.pv]
        return out, Vars[self.StaticParamLoc0:self.StaticParamLoc0 + self.spv].copy()


def thread_split(nappl: int, parts: int):
    """Contiguous ByApplication split (IndexingData.h:117-146): list of (start, count)."""
    per, rem = divmod(nappl, parts)
    out, start = [], 0
    for i in range(parts if per > 0 else rem):
        cnt = per + (1 if i < rem else 0)
        out.append((start, cnt))
        start += cnt
    return out


def trapezoidal_hessian_mask(xv: int, uv: int, pv: int, blocked: bool) -> np.ndarray:
    """``HessianElemIsNonZero`` of the Trapezoidal defects with ``EnableHessianSparsity`` (TrapezoidalDefects.h:75-130): bool
    [IR, IR], IR = 2 q + p -- true on the two node blocks, on every row / column of a parameter and of the two node times; false
    on the rest of the cross-node block."""
    q = xv + 1 + (0 if blocked else uv)
    p = (uv + pv) if blocked else pv
    IR, T = 2 * q + p, xv
    m = np.zeros((IR, IR), dtype=bool)
    m[:q, :q] = m[q:2 * q, q:2 * q] = True
    m[2 * q:, :] = m[:, 2 * q:] = True
    for k in (T, T + q):
        m[k, :] = m[:, k] = True
    return m


def kkt_slot_locations(vindex, cindex, n_primal: int, con_offset: int = 0, hess_mask=None):
    """``KKTLocations`` of ONE equality constraint alone in its program: for every application V and block slot k (slot order
    ``for i < IR {H(j,i), j >= i ; J(j,i), j < OR}``, DenseFunctionBase.h:1112-1123) the offset of its entry in the value array
    of the upper-triangular row-major CSR KKT matrix (NonLinearProgram.cpp:267-344: lower-triangle triplets are transposed,
    equality rows live behind the primal variables and the slacks -- none here -- so J(c, v) is stored at (v, n_primal + c)).
    Structure only: the matrix holds exactly the entries the constraint's slots name, columns sorted within a row, as
    ``analyzeSparsity`` leaves them.  Returns (locations[nappl, NKKT] int64, nnz).  ``hess_mask[IR, IR]`` (bool, optional:
    ``trapezoidal_hessian_mask``): Hessian slots whose entry is False claim no location -- their map entry is -1, which the
    device assembly drops (include/asset_hip.h: asset_hip_defect_set_kkt_map)."""
    V = np.ascontiguousarray(vindex, dtype=np.int64)
    C = np.ascontiguousarray(cindex, dtype=np.int64)
    nappl, IR = V.shape
    OR = C.shape[1]
    dim = int(n_primal) + int(con_offset) + int(C.max()) + 1
    cols_h, rows_h = [], []
    keys = np.empty((nappl, IR * (IR + 1) // 2 + OR * IR), dtype=np.int64)
    k = 0
    for i in range(IR):
        vi = V[:, i]
        for j in range(i, IR):                      # H(j, i), lower triangle: stored at (min, max)
            vj = V[:, j]
            keys[:, k] = np.minimum(vi, vj) * dim + np.maximum(vi, vj)
            k += 1
        for j in range(OR):                         # J(j, i): row of the variable, column of the constraint
            keys[:, k] = vi * dim + (n_primal + con_offset + C[:, j])
            k += 1
    if hess_mask is None:
        uniq, inv = np.unique(keys.ravel(), return_inverse=True)
        return inv.reshape(keys.shape), int(uniq.size)
    keep = np.ones(keys.shape[1], dtype=bool)
    k = 0
    for i in range(IR):
        for j in range(i, IR):
            keep[k] = bool(hess_mask[j, i])
            k += 1
        k += OR
    uniq, inv = np.unique(keys[:, keep].ravel(), return_inverse=True)
    out = np.full(keys.shape, -1, dtype=np.int64)
    out[:, keep] = inv.reshape(nappl, int(keep.sum()))
    return out, int(uniq.size)
