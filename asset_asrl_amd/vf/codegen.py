"""Straight-line code emission for an ODE right-hand side and its derivatives.

Given ``f : R^N -> R^n`` (N = n+1+m+p, the ODE input ``[x,t,u,p]``) as a DAG, build

* ``J[k,i]  = d f_k / d y_i``                    (forward derivatives on the DAG)
* ``g[i]    = sum_k lam_k d f_k / d y_i``         (one reverse sweep of ``lam . f``)
* ``H[i,j]  = sum_k lam_k d2 f_k / d y_i d y_j``  (forward derivative of ``g``; lower triangle)

and print them as one fully inlined function per "level" (value / value+J /
value+J+g+H).  These three levels are exactly the three entry points the reference
calls on a user ODE from the collocation defects -- ``compute``, ``compute_jacobian``,
``compute_jacobian_adjointgradient_adjointhessian``
(/root/reference/src/OptimalControl/LGLDefects.h:74,148,366-367,383-384).

Two printers share the same node schedule:

* :func:`emit_hip_functor`  -- a ``struct`` of ``__host__ __device__`` templates writing
  through accessor objects, so a kernel can route results straight into LDS tiles /
  registers with its own layout (no intermediate dense arrays).
* :func:`emit_c`            -- plain C with pointer arguments (row-major J, full symmetric H).
"""
from __future__ import annotations

import os
from dataclasses import dataclass
from typing import Dict, List, Sequence, Tuple

from .functions import VectorFunction
import numpy as np

from .ir import COND_OPS, TABLES
from .ir import GRAPH as G
from .ir import Node, topo_order

_COND_C = {"lt": "<", "le": "<=", "gt": ">", "ge": ">=", "and": "&&", "or": "||"}


@dataclass
class OdeDerivatives:
    name: str
    xv: int
    uv: int
    pv: int
    f: List[Node]
    J: List[List[Node]]      # [n][N]
    g: List[Node]            # [N]
    H: List[List[Node]]      # [N][N], only j<=i filled (lower), mirrored on request
    chain_rule = None        # how the derivatives were formed: {"form": "flat" | "block", "ops_flat": ..., "ops_block": ...}

    @property
    def nin(self) -> int:
        return self.xv + 1 + self.uv + self.pv

    def stats(self) -> Dict[str, int]:
        def count(roots):
            return sum(1 for n in topo_order(roots) if n.args)
        z = G.zero
        return {
            "ops_f": count(self.f),
            "ops_fj": count(self.f + [e for r in self.J for e in r]),
            "ops_fjgh": count(self.f + [e for r in self.J for e in r] + self.g
                              + [self.H[i][j] for i in range(self.nin) for j in range(i + 1)]),
            "nnz_J": sum(1 for r in self.J for e in r if e is not z),
            "nnz_H_lower": sum(1 for i in range(self.nin) for j in range(i + 1) if self.H[i][j] is not z),
        }


def differentiate_function(name: str, func: VectorFunction) -> OdeDerivatives:
    """Derivatives of a plain vector function (a path constraint, a spacing relation ...).  The record is the ODE
    one with XV = outputs and the inputs split as XV + 1 + UV (UV = inputs - outputs - 1 may be negative: the
    function kernels only use XV and the input count)."""
    return differentiate(name, func, func.ORows(), func.IRows() - func.ORows() - 1, 0)


def _count_ops(roots) -> int:
    return sum(1 for n in topo_order(roots) if n.args)


def _differentiate_flat(f, ys, lams):
    """Every input direction through the whole (flattened) expression: forward derivatives for J, one reverse sweep of lam . f
    for g, forward derivatives of g for H."""
    N = len(ys)
    J = [[G.d(fk, y) for y in ys] for fk in f]
    g = G.grad(G.dot(lams, f), ys)
    H = [[G.zero] * N for _ in range(N)]
    for i in range(N):
        for j in range(i + 1):
            H[i][j] = G.d(g[i], ys[j])
    return J, g, H


def _differentiate_block(f, ys, lams):
    """The chain rule BLOCK-wise across the cuts of the expression (ir.Graph.cut: the outputs of the inner function of a
    composition, or what a definition marks with ``.cut()``) -- what the reference's NestedFunction does with the Jacobians and
    adjoint Hessians of its two functions (CommonFunctions/NestedFunction.h:140-270), here on the expression graph, any depth:

        J   = f_y + f_c T                                   T = dc/dy (total), f_y / f_c partials with the cuts held as variables
        g   = s_y + T^T s_c                                 s = lam . f
        H   = s_yy + s_yc T + (s_yc T)^T + T^T s_cc T + Hess_y( sum_m w_m c_m(y) ) |_(w = s_c held constant)

    the last term by recursion on the cuts' own expressions.  The partials with respect to the cut variables are formed once and
    shared by every input direction: with a cut of m variables under N inputs the outer expression is differentiated in m + (its
    direct inputs) directions instead of N, and in (m + ...)^2 / 2 second-order pairs instead of N^2 / 2."""
    from .ir import frontier
    N = len(ys)
    zero = G.zero
    yidx = {y.id: i for i, y in enumerate(ys)}
    cuts = [n for n in topo_order(f) if n.op == "cut"]            # inner ones first
    T: Dict[int, List[Node]] = {}

    def total_first(e: Node) -> List[Node]:
        _, cs = frontier([e])
        row = []
        for y in ys:
            acc = G.d(e, y)
            for c in cs:
                t = T[c.id][yidx[y.id]]
                if t is zero:
                    continue
                pc = G.d(e, c)
                if pc is not zero:
                    acc = G.add(acc, G.mul(pc, t))
            row.append(acc)
        return row
    for c in cuts:
        T[c.id] = total_first(c.args[0])
    J = [total_first(fk) for fk in f]

    def low(i, j):
        return (i, j) if i >= j else (j, i)

    def grad_hess(S: Node):
        vs, cs = frontier([S])
        Z = vs + cs
        ny = len(vs)
        gb = G.grad(S, Z)                                         # partial gradient, cuts as variables
        iy = [yidx[v.id] for v in vs]
        g = [zero] * N
        for a in range(ny):
            g[iy[a]] = gb[a]
        for b, c in enumerate(cs):
            if gb[ny + b] is zero:
                continue
            for i in range(N):
                t = T[c.id][i]
                if t is not zero:
                    g[i] = G.add(g[i], G.mul(gb[ny + b], t))
        P = [[G.d(gb[a], Z[b]) if b <= a else None for b in range(len(Z))] for a in range(len(Z))]   # second partials, lower
        H = [[zero] * N for _ in range(N)]
        for a in range(ny):                                       # s_yy
            for b in range(a + 1):
                i, j = low(iy[a], iy[b])
                H[i][j] = G.add(H[i][j], P[a][b])
        for m, c in enumerate(cs):                                # s_yc T + (s_yc T)^T
            Tc = T[c.id]
            for b in range(ny):
                pcy = P[ny + m][b]
                if pcy is zero:
                    continue
                ib = iy[b]
                for j in range(N):
                    if Tc[j] is zero:
                        continue
                    term = G.mul(pcy, Tc[j])
                    if j == ib:
                        term = G.add(term, term)
                    i2, j2 = low(ib, j)
                    H[i2][j2] = G.add(H[i2][j2], term)
        if cs:                                                    # T^T s_cc T
            U = []
            for m in range(len(cs)):
                row = [zero] * N
                for m2, c2 in enumerate(cs):
                    pcc = P[ny + max(m, m2)][ny + min(m, m2)]
                    if pcc is zero:
                        continue
                    for j in range(N):
                        t = T[c2.id][j]
                        if t is not zero:
                            row[j] = G.add(row[j], G.mul(pcc, t))
                U.append(row)
            for i in range(N):
                for j in range(i + 1):
                    acc = H[i][j]
                    for m, c in enumerate(cs):
                        if T[c.id][i] is not zero and U[m][j] is not zero:
                            acc = G.add(acc, G.mul(T[c.id][i], U[m][j]))
                    H[i][j] = acc
            # sum_m (ds/dc_m) Hess_y(c_m): the adjoint weights held constant, the cuts' own expressions one level down
            live = [(m, c) for m, c in enumerate(cs) if gb[ny + m] is not zero]
            if live:
                ws = [G.frozen() for _ in live]
                sigma = G.sum(G.mul(w, c.args[0]) for w, (_, c) in zip(ws, live))
                _, H2 = grad_hess(sigma)
                flat = [H2[i][j] for i in range(N) for j in range(i + 1)]
                flat = G.replace(flat, {w.id: gb[ny + m] for w, (m, _) in zip(ws, live)})
                k = 0
                for i in range(N):
                    for j in range(i + 1):
                        H[i][j] = G.add(H[i][j], flat[k])
                        k += 1
        return g, H
    g, H = grad_hess(G.dot(lams, f))
    return J, g, H


def _differentiate_block_local(f, ys, lams):
    """Block-wise chain rule with LOCAL Jacobians: the cut variables are eliminated group by group from the outermost inwards, the
    second-order terms carried in the coordinates of the variables still alive (inputs and deeper cuts) -- for a group of cuts
    c = e(z) with local Jacobian L = de/dz (barrier partials), weights w_c and the rows / columns A_c. of the running matrix:
        A_uv += sum_m L_mu (A_{c_m v} + sum_m' A_{c_m c_m'} L_m'v) + sum_m A_{u c_m} L_mv + Hess_z(sum_m w_m e_m)_uv ,   w_u += sum_m w_m L_mu
    (the reference's NestedFunction rule, CommonFunctions/NestedFunction.h:140-270, applied at every level of the nest)."""
    from .ir import frontier
    N = len(ys)
    zero = G.zero
    yidx = {y.id: i for i, y in enumerate(ys)}
    cuts = [n for n in topo_order(f) if n.op == "cut"]            # inner ones first
    depth: Dict[int, int] = {}
    for c in cuts:
        _, cs = frontier([c.args[0]])
        depth[c.id] = 1 + max([depth[x.id] for x in cs], default=0)
    # ---- first derivatives (J): total derivatives of the cuts by local composition, then of the outputs
    T: Dict[int, List[Node]] = {}

    def total_first(e: Node) -> List[Node]:
        _, cs = frontier([e])
        row = []
        for y in ys:
            acc = G.d(e, y)
            for c in cs:
                t = T[c.id][yidx[y.id]]
                if t is zero:
                    continue
                pc = G.d(e, c)
                if pc is not zero:
                    acc = G.add(acc, G.mul(pc, t))
            row.append(acc)
        return row
    for c in cuts:
        T[c.id] = total_first(c.args[0])
    J = [total_first(fk) for fk in f]
    # ---- second order: eliminate the cuts, outermost group first
    S = G.dot(lams, f)
    vs, cs = frontier([S])
    Z = vs + cs
    gb = G.grad(S, Z)
    w: Dict[int, Node] = {}
    A: Dict[int, Dict[int, Node]] = {}
    node_of: Dict[int, Node] = {}

    def addA(u: Node, v: Node, val: Node):
        if val is zero:
            return
        node_of[u.id], node_of[v.id] = u, v
        row = A.setdefault(u.id, {})
        row[v.id] = G.add(row.get(v.id, zero), val)
        if u.id != v.id:
            row2 = A.setdefault(v.id, {})
            row2[u.id] = row[v.id]

    def addw(u: Node, val: Node):
        if val is zero:
            return
        node_of[u.id] = u
        w[u.id] = G.add(w.get(u.id, zero), val)
    for a, z in enumerate(Z):
        addw(z, gb[a])
        for b in range(a + 1):
            addA(z, Z[b], G.d(gb[a], Z[b]))
    by_depth: Dict[int, List[Node]] = {}
    for c in cuts:
        by_depth.setdefault(depth[c.id], []).append(c)
    for dep in sorted(by_depth, reverse=True):
        grp = [c for c in by_depth[dep] if c.id in w or c.id in A]
        if not grp:
            continue
        gids = {c.id for c in grp}
        es = [c.args[0] for c in grp]
        zv, zc = frontier(es)
        Zg = zv + zc
        L = [[G.d(e, z) for z in Zg] for e in es]                # local Jacobian (barrier partials)
        # Hess_z( sum_m w_m e_m ), the weights frozen
        live = [m for m, c in enumerate(grp) if w.get(c.id, zero) is not zero]
        Hs = {}
        if live:
            fr = [G.frozen() for _ in live]
            sigma = G.sum(G.mul(x, es[m]) for x, m in zip(fr, live))
            gs = G.grad(sigma, Zg)
            flat = [G.d(gs[a], Zg[b]) for a in range(len(Zg)) for b in range(a + 1)]
            flat = G.replace(flat, {x.id: w[grp[m].id] for x, m in zip(fr, live)})
            k = 0
            for a in range(len(Zg)):
                for b in range(a + 1):
                    Hs[(a, b)] = flat[k]
                    k += 1
        # rows of A that belong to the group
        Arow = {c.id: dict(A.get(c.id, {})) for c in grp}
        # C[m][b] = sum_m' A_{c_m c_m'} L[m'][b]
        Cm = []
        for m, c in enumerate(grp):
            row = [zero] * len(Zg)
            for m2, c2 in enumerate(grp):
                acc = Arow[c.id].get(c2.id, zero)
                if acc is zero:
                    continue
                for b in range(len(Zg)):
                    if L[m2][b] is not zero:
                        row[b] = G.add(row[b], G.mul(acc, L[m2][b]))
            Cm.append(row)
        # drop the group from A and w
        wg = {c.id: w.pop(c.id, zero) for c in grp}
        for c in grp:
            A.pop(c.id, None)
        for r in A.values():
            for cid in gids:
                r.pop(cid, None)
        # gradient
        for m, c in enumerate(grp):
            if wg[c.id] is zero:
                continue
            for b, z in enumerate(Zg):
                if L[m][b] is not zero:
                    addw(z, G.mul(wg[c.id], L[m][b]))
        # second order: pairs inside Zg
        for a in range(len(Zg)):
            for b in range(a + 1):
                acc = Hs.get((a, b), zero)
                for m, c in enumerate(grp):
                    # L[m][a] * (A[c_m][z_b] + C[m][b]) + A[z_a][c_m] * L[m][b]
                    d1 = G.add(Arow[c.id].get(Zg[b].id, zero), Cm[m][b])
                    if L[m][a] is not zero and d1 is not zero:
                        acc = G.add(acc, G.mul(L[m][a], d1))
                    d2 = Arow[c.id].get(Zg[a].id, zero)
                    if d2 is not zero and L[m][b] is not zero:
                        acc = G.add(acc, G.mul(d2, L[m][b]))
                addA(Zg[a], Zg[b], acc)
        # ... and pairs (z in Zg, x alive outside Zg and outside the group)
        zg_ids = {z.id for z in Zg}
        others = set()
        for c in grp:
            others |= {xid for xid in Arow[c.id] if xid not in gids and xid not in zg_ids}
        for xid in others:
            x = node_of[xid]
            for a, z in enumerate(Zg):
                acc = zero
                for m, c in enumerate(grp):
                    v = Arow[c.id].get(xid, zero)
                    if v is not zero and L[m][a] is not zero:
                        acc = G.add(acc, G.mul(L[m][a], v))
                addA(z, x, acc)
    g = [w.get(y.id, zero) for y in ys]
    H = [[zero] * N for _ in range(N)]
    for i in range(N):
        for j in range(i + 1):
            H[i][j] = A.get(ys[i].id, {}).get(ys[j].id, zero)
    return J, g, H


BLOCK_CHAIN_RULE = os.environ.get("ASSET_BLOCK_CHAIN_RULE", "1") == "1"


def differentiate(name: str, ode: VectorFunction, xv: int, uv: int, pv: int) -> OdeDerivatives:
    """f, J, g, H of an ODE right-hand side (or a plain function).  An expression with cuts (ir.Graph.cut -- compositions
    ``F(G)``, ``.cut()``) is differentiated twice, flattened and block-wise across its cuts, and the form with fewer operations in the
    value + Jacobian + adjoint-gradient + adjoint-Hessian body is kept; what is returned holds no cuts."""
    N = xv + 1 + uv + pv
    if ode.IRows() != N:
        raise ValueError(f"ODE input size {ode.IRows()} != XV+1+UV+PV = {N}")
    if ode.ORows() != xv:
        raise ValueError(f"ODE output size {ode.ORows()} != XV = {xv}")
    ys = [G.var(i) for i in range(N)]
    lams = [G.lam(k) for k in range(xv)]
    f_cut = list(ode.outs)
    f = G.strip_cuts(f_cut)

    def pack(J, g, H):
        return f + [e for r in J for e in r] + list(g) + [H[i][j] for i in range(N) for j in range(i + 1)]

    def unpack(flat):
        k = len(f)
        J = [flat[k + r * N:k + (r + 1) * N] for r in range(xv)]
        k += xv * N
        g = flat[k:k + N]
        k += N
        H = [[G.zero] * N for _ in range(N)]
        for i in range(N):
            for j in range(i + 1):
                H[i][j] = flat[k]
                k += 1
        return J, g, H
    J, g, H = _differentiate_flat(f, ys, lams)
    d = OdeDerivatives(name, xv, uv, pv, f, J, g, H)
    d.chain_rule = {"form": "flat", "ops_flat": _count_ops(pack(J, g, H))}
    if BLOCK_CHAIN_RULE and any(n.op == "cut" for n in topo_order(f_cut)):
        best = ("flat", d.chain_rule["ops_flat"], (J, g, H))
        for form, fn in (("block", _differentiate_block), ("block_local", _differentiate_block_local)):
            Jb, gb, Hb = unpack(G.strip_cuts(f_cut + pack(*fn(f_cut, ys, lams))[len(f):]))
            ops = _count_ops(pack(Jb, gb, Hb))
            d.chain_rule["ops_" + form] = ops
            if ops < best[1]:
                best = (form, ops, (Jb, gb, Hb))
        if best[0] != "flat":
            cr = dict(d.chain_rule, form=best[0])
            d = OdeDerivatives(name, xv, uv, pv, f, *best[2])
            d.chain_rule = cr
    return d


# --------------------------------------------------------------------------- printing

def _cnum(v: float) -> str:
    s = repr(float(v))
    if s in ("inf", "-inf", "nan"):
        raise ValueError("non-finite constant in ODE expression")
    if "e" not in s and "." not in s:
        s += ".0"
    return f"({s})" if v < 0 else s


SPLIT_OPS = 1500     # fjgh bodies above this many operations are emitted in two out-of-line parts
# Bodies with many elementary-function calls (the 32-state BASELINE ODE: 33 sincos): a scheduling fence behind each call.  The
# calls are independent of each other, and the compiler's scheduler, which has the whole body as one block, interleaves all
# of them -- ~20 temporaries each -- until the 512 registers of a one-wave-per-SIMD kernel are full and 340 more values
# live in scratch memory (csrc/asset_math.h: ASSET_SCHED_FENCE).  0: never.
TRANS_FENCE = int(os.environ.get("ASSET_TRANS_FENCE", "8"))
LEVEL_ORDER = os.environ.get("ASSET_LEVEL_ORDER", "1") == "1"   # breadth-first statement schedule (experiment switch)
UNIT_LEVEL_ORDER = os.environ.get("ASSET_UNIT_LEVEL_ORDER", "0") == "1"   # the same for the unit bodies of heavy ODEs
UNIT_QUAL = os.environ.get("ASSET_UNIT_QUAL", "__attribute__((always_inline)) inline")   # qualifier of the unit bodies: inlined, their inputs
#   and output pointers stay in registers (out of line they are read back from scratch memory in mid-body, and every such wait
#   also waits for the stores before it: Betts-LGL5 x 1 000: 43.5 -> 41.6 us)

TRANSCENDENTAL = ("sin", "cos", "tan", "exp", "log", "sqrt", "tanh", "sinh", "cosh", "asin", "acos", "atan",
                  "atan2", "powr")


def lower_reciprocals(roots: Sequence[Node]) -> List[Node]:
    """a/d -> a*(1/d) for every denominator d that divides at least twice: one division instead of many.

    Costs at most one extra rounding per quotient (the reference is built with -ffast-math, which licenses the
    same rewrite); both printers apply it, so the CPU baseline and the device evaluate the same expression."""
    order = topo_order(roots)
    uses: Dict[int, int] = {}
    for n in order:
        if n.op == "div" and not n.args[1].is_const():
            uses[n.args[1].id] = uses.get(n.args[1].id, 0) + 1
    memo: Dict[int, Node] = {}
    rcp: Dict[int, Node] = {}
    for n in order:
        if not n.args:
            memo[n.id] = n
            continue
        args = [memo[a.id] for a in n.args]
        if n.op == "div" and uses.get(n.args[1].id, 0) >= 2:
            d = args[1]
            r = rcp.get(d.id)
            if r is None:
                r = rcp[d.id] = G._mk("div", (G.one, d))
            memo[n.id] = G.mul(args[0], r)
        else:
            memo[n.id] = G.rebuild(n, args)
    return [memo[r.id] for r in roots]


def _topo_stop(roots: Sequence[Node], stop) -> List[Node]:
    """topo order of what is reachable from roots without descending below the nodes in `stop`."""
    if not stop:
        return topo_order(roots)
    seen, out = set(), []
    for r in roots:
        stack = [(r, 0)]
        while stack:
            n, i = stack.pop()
            if i == 0:
                if n.id in seen:
                    continue
                seen.add(n.id)
            kids = () if n.id in stop else n.args
            if i < len(kids):
                stack.append((n, i + 1))
                if kids[i].id not in seen:
                    stack.append((kids[i], 0))
            else:
                out.append(n)
    return out


class _Printer:
    """Schedules reachable interior nodes into temporaries t0,t1,...

    ``loaded``: {node id: expression} -- nodes whose value is supplied (not recomputed; nothing below them is scheduled
    unless needed elsewhere).  ``pair_sincos``: emit one ``sincos`` for a sin/cos pair on the same argument (device)."""

    def __init__(self, roots: Sequence[Node], yname="y{}", lname="l{}", loaded=None, pair_sincos=False,
                 level_order=False, tabprefix=""):
        self.tabprefix = tabprefix           # plain C: the table arrays and look-ups are file-scope names of this function's own
        self.names: Dict[int, str] = {}
        self.lines: List[str] = []
        self.used_y = set()
        self.used_l = set()
        self.used_c = set()
        self.yname, self.lname = yname, lname
        self.device_math = pair_sincos      # device functors: csrc/asset_math.h trig (short argument reduction)
        loaded = loaded or {}
        order = _topo_stop(roots, set(loaded))
        if level_order:
            # breadth-first schedule: a node is emitted after everything of smaller depth, so neighbouring statements
            # are independent and a single wave can overlap their latencies (a depth-first order is one long chain)
            depth: Dict[int, int] = {}
            for n in order:
                depth[n.id] = 0 if (not n.args or n.id in loaded) else 1 + max(depth[a.id] for a in n.args)
            order = sorted(order, key=lambda n: depth[n.id])
        partner: Dict[int, Node] = {}
        if pair_sincos:
            by_arg: Dict[int, Dict[str, Node]] = {}
            for n in order:
                if n.op in ("sin", "cos") and n.id not in loaded:
                    by_arg.setdefault(n.args[0].id, {})[n.op] = n
            for pr in by_arg.values():
                if len(pr) == 2:
                    partner[pr["sin"].id] = pr["cos"]
                    partner[pr["cos"].id] = pr["sin"]
        ncalls = sum(1 for n in order if n.op in TRANSCENDENTAL and n.id not in loaded) - len(partner) // 2
        self.fenced = bool(self.device_math and TRANS_FENCE > 0 and ncalls > TRANS_FENCE)
        fence = " ASSET_SCHED_FENCE();" if self.fenced else ""
        self.pos: Dict[int, int] = {}        # node id -> index of the line that defines it (absent: an input, a constant, a loaded value)
        for n in order:
            if n.op == "var":
                self.used_y.add(n.value)
            elif n.op == "lam":
                self.used_l.add(n.value)
            elif n.op == "aconst":
                self.used_c.add(n.value)
            if n.id in loaded:
                self.names[n.id] = loaded[n.id]
                continue
            if not n.args or n.id in self.names:
                continue
            if n.op in COND_OPS:             # a condition is printed where it is used (inside its select): no temporary of its own
                continue
            if n.id in partner:
                other = partner[n.id]
                k = len(self.lines)
                sn, cn = f"t{k}s", f"t{k}c"
                self.lines.append(f"double {sn}, {cn}; asset_sincos({self.ref(n.args[0])}, &{sn}, &{cn});{fence}")
                self.names[n.id] = sn if n.op == "sin" else cn
                self.names[other.id] = cn if n.op == "sin" else sn
                self.pos[n.id] = self.pos[other.id] = len(self.lines) - 1
                continue
            nm = f"t{len(self.lines)}"
            self.lines.append(f"const double {nm} = {self._expr(n)};{fence if n.op in TRANSCENDENTAL else ''}")
            self.names[n.id] = nm
            self.pos[n.id] = len(self.lines) - 1

    def ref(self, n: Node) -> str:
        if n.op == "const":
            return _cnum(n.value)
        if n.op == "var":
            return self.yname.format(n.value)
        if n.op == "lam":
            return self.lname.format(n.value)
        if n.op == "aconst":
            return f"c{n.value}"
        if n.op in COND_OPS:
            a, b = (self.ref(x) for x in n.args)
            return f"({a} {_COND_C[n.op]} {b})"
        return self.names[n.id]

    def _expr(self, n: Node) -> str:
        a = [self.ref(x) for x in n.args]
        op = n.op
        if op == "add":
            return f"{a[0]} + {a[1]}"
        if op == "sub":
            return f"{a[0]} - {a[1]}"
        if op == "mul":
            return f"{a[0]} * {a[1]}"
        if op == "div":
            return f"{a[0]} / {a[1]}"
        if op == "neg":
            return f"-{a[0]}"
        if op == "powi":
            return _powi_expr(a[0], n.value)
        if op == "powr":
            return f"pow({a[0]}, {_cnum(n.value)})"
        if op == "atan2":
            return f"atan2({a[0]}, {a[1]})"
        if op == "select":
            return f"{a[0]} ? {a[1]} : {a[2]}"
        if op == "tabloc":                  # InterpTable1D.h:181-197 (get_telem): evenly spaced abscissae by division, others by bisection
            tab = TABLES[n.value]
            if tab.teven:
                return f"{self.tabprefix or 'asset_'}tab_even({a[0]}, {_cnum(tab.ts[0])}, {_cnum(tab.ts[1] - tab.ts[0])}, {tab.tsize})"
            return f"{self.tabprefix or 'asset_'}tab_find({self.tabprefix}TAB_{n.value}_t, {tab.tsize}, {a[0]})"
        if op == "tabget":
            dg, arr, row, off = n.value
            return f"{self.tabprefix}TAB_{dg}_{arr}[{row * TABLES[dg].tsize + off} + (int){a[0]}]"
        if op == "abs":
            return f"fabs({a[0]})"
        if op == "sign":
            return f"(double)(({a[0]} > 0.0) - ({a[0]} < 0.0))"
        if self.device_math and op in ("sin", "cos", "tan"):
            return f"asset_{op}({a[0]})"
        return f"{op}({a[0]})"


def table_arrays(roots: Sequence[Node]) -> List[Tuple[str, List[float]]]:
    """[(array name, numbers)] of every table array the expressions read (sorted: the same function prints the same code)."""
    need = set()
    for n in topo_order(roots):
        if n.op == "tabget":
            need.add((n.value[0], n.value[1]))
        elif n.op == "tabloc" and not TABLES[n.value].teven:
            need.add((n.value, "t"))
    out = []
    for dg, arr in sorted(need):
        tab = TABLES[dg]
        data = tab.ts if arr == "t" else (tab.vs if arr == "v" else tab.dvs_dts)
        out.append((f"TAB_{dg}_{arr}", [float(v) for v in np.asarray(data, dtype=float).ravel()]))
    return out


# the two look-ups as plain C (the device has them in csrc/asset_math.h)
TABLE_HELPERS_C = """static inline double asset_tab_even(double t, double t0, double step, int n) {
  int e = (int)((t - t0) / step);
  e = e < n - 2 ? e : n - 2;
  return (double)(e > 0 ? e : 0);
}
static inline double asset_tab_find(const double* ts, int n, double t) {
  int lo = 0, hi = n;
  while (lo < hi) { int mid = (lo + hi) >> 1; if (ts[mid] <= t) lo = mid + 1; else hi = mid; }
  int e = lo - 1;
  e = e < n - 2 ? e : n - 2;
  return (double)(e > 0 ? e : 0);
}
"""


def _powi_expr(x: str, n: int) -> str:
    """x**n (n>=3) by binary powering, written out as products."""
    assert n >= 3
    terms = []
    sq = x
    k = n
    # square-and-multiply on strings; depth is tiny for the exponents ODEs use
    while k:
        if k & 1:
            terms.append(sq)
        k >>= 1
        if k:
            sq = f"({sq} * {sq})"
    return " * ".join(terms)


def _level_roots(d: OdeDerivatives, level: int) -> List[Node]:
    roots = list(d.f)
    if level >= 1:
        roots += [e for r in d.J for e in r]
    if level >= 2:
        roots += d.g + [d.H[i][j] for i in range(d.nin) for j in range(i + 1)]
    return roots


def emit_hip_functor(d: OdeDerivatives, struct_name: str) -> str:
    """HIP/C++ functor.  Accessor contracts:

    ``in.y(i)`` / ``in.lam(k)`` return doubles; ``out.f(k,v)``, ``out.J(k,i,v)``, ``out.g(i,v)``,
    ``out.H(i,j,v)`` (called once per lower-triangle entry, j<=i) receive every entry
    including structural zeros, so the caller never has to pre-clear.
    """
    N, n = d.nin, d.xv
    o: List[str] = []
    o.append(f"// generated by asset_asrl_amd/vf/codegen.py -- ODE '{d.name}'  (XV={d.xv}, UV={d.uv}, PV={d.pv})")
    st = d.stats()
    o.append(f"// ops: f={st['ops_f']} f+J={st['ops_fj']} f+J+g+H={st['ops_fjgh']}  "
             f"nnz(J)={st['nnz_J']}/{n * N}  nnz(H lower)={st['nnz_H_lower']}/{N * (N + 1) // 2}")
    o.append(f"struct {struct_name} {{")
    o.append(f"  static constexpr int XV = {d.xv}, UV = {d.uv}, PV = {d.pv}, NIN = {N};")
    o.append(f"  static constexpr int NNZ_J = {st['nnz_J']}, NNZ_H = {st['nnz_H_lower']};")
    o.append(f"  static constexpr int OPS_FJGH = {st['ops_fjgh']};   // operations of the value + J + g + H body (csrc/defect_resident.h: which looped form pays)")
    nac = 1 + max([nd.value for nd in topo_order(_level_roots(d, 2)) if nd.op == "aconst"], default=-1)
    o.append(f"  static constexpr int NACONST = {nac};   // constants of the application the function reads (vf.ApplConst)")
    o.append(f"  static constexpr const char* name() {{ return \"{d.name}\"; }}")
    # structural sparsity: compact position of every J / H (packed lower) entry or -1, and its inverse.  The LGL workspace
    # stores only the non-zeros; the dense accessors ignore these tables.  J is numbered COLUMN by column (input direction
    # by input direction): a column's entries are what one output unit of a heavy right-hand side writes (csrc/defect_units.h)
    # and what one block column of the row-wise dense stage reads, as one run (csrc/defect_rows.h)
    z = G.zero
    jnz = [k * N + i for i in range(N) for k in range(n) if d.J[k][i] is not z]
    # ... and the packed lower Hessian COLUMN by column too (round 4): column j = {H[i][j], i >= j} is what the unit of input
    # direction j writes -- one run of its slot section instead of one entry per row
    hnz = [i * (i + 1) // 2 + j for j in range(N) for i in range(j, N) if d.H[i][j] is not z]
    jpos = {e: c for c, e in enumerate(jnz)}
    hpos = {e: c for c, e in enumerate(hnz)}
    def arr(name, vals):
        vals = list(vals) or [0]
        return f"  static constexpr short {name}[{len(vals)}] = {{{', '.join(str(v) for v in vals)}}};"
    o.append(arr("JPOS", [jpos.get(e, -1) for e in range(n * N)]))
    o.append(arr("HPOS", [hpos.get(e, -1) for e in range(N * (N + 1) // 2)]))
    o.append(arr("JIDX", jnz))
    o.append(arr("HIDX", hnz))
    for nm, vals in table_arrays(_level_roots(d, 2)):       # tabulated data (vf.InterpTable1D): constant arrays of the module
        o.append(f"  static constexpr double {nm}[{len(vals)}] = {{{', '.join(_cnum(v) for v in vals)}}};")
    saved = saved_nodes(d)
    o.insert(-1, f"  static constexpr int NSAVE = {len(saved)};   // transcendental values f_save() hands to fjgh_load()")
    # Large bodies (Betts low-thrust: 7 800 operations for value + J + g + H, ~780 values live at the peak) are emitted
    # as two out-of-line parts -- [f, J, g] and [H] -- so each gets its own register allocation (live sets 221 / 391
    # instead of 783), in depth-first order (the breadth-first order widens live ranges further).  The parts recompute
    # what they share.
    split = st["ops_fjgh"] > SPLIT_OPS
    level_order = LEVEL_ORDER and not split
    sig = "template <class In, class Out> __host__ __device__ static {q} void {name}(const In& __restrict__ in, Out& __restrict__ out)"

    def outputs(level):
        """[(statement format, root)] in the order of _level_roots(d, level)."""
        outs = [(f"out.f({k}, {{}});", d.f[k]) for k in range(n)]
        if level >= 1:
            outs += [(f"out.J({k}, {i}, {{}});", d.J[k][i]) for k in range(n) for i in range(N)]
        if level >= 2:
            outs += [(f"out.g({i}, {{}});", d.g[i]) for i in range(N)]
            outs += [(f"out.H({i}, {j}, {{}});", d.H[i][j]) for i in range(N) for j in range(i + 1)]
        return outs

    def many_calls(outs, extra_roots):
        nodes = topo_order([r for _, r in outs] + list(extra_roots))
        return TRANS_FENCE > 0 and sum(1 for x in nodes if x.op in TRANSCENDENTAL) > 2 * TRANS_FENCE

    def body(name, outs, extra_roots=(), extra_stmt=None, use_saved=False, q="inline", level_order=level_order):
        # STREAMED bodies (many elementary-function calls: the 32-state BASELINE ODE has 33 sincos): the outputs in sharing order,
        # statements depth-first, every output written where its value is complete, a scheduling fence behind it, the saved
        # values loaded where they are first used -- the live set is then the inputs plus one output's sub-expressions.  As one
        # block the compiler hoists every call and every load to the top and sinks every store to the bottom: 512 registers and
        # 1 368 bytes of scratch per lane for the interior body (csrc/asset_math.h: ASSET_SCHED_FENCE).
        stream = many_calls(outs, extra_roots)
        if stream:
            outs, level_order = share_order(outs), False
        roots = [r for _, r in outs] + list(extra_roots)
        if use_saved:
            low = lower_reciprocals(roots + saved)       # lowering rebuilds nodes: locate the saved ones afterwards
            loaded = {low[len(roots) + k].id: f"s{k}" for k in range(len(saved))}
            low = low[:len(roots)]
        else:
            low, loaded = lower_reciprocals(roots), None
        p = _Printer(low, loaded=loaded, pair_sincos=True, level_order=level_order)
        o.append("  " + sig.format(q=q, name=name) + " {")
        for i in sorted(p.used_y):
            o.append(f"    const double y{i} = in.y({i});")
        for k in sorted(p.used_l):
            o.append(f"    const double l{k} = in.lam({k});")
        for k in sorted(p.used_c):
            o.append(f"    const double c{k} = in.aconst({k});")   # constants of the application (vf.ApplConst)
        stmts = [fmt.format(p.ref(node)) for (fmt, _), node in zip(outs, low)]
        nodes = list(low[:len(outs)])
        if extra_stmt:
            stmts += [extra_stmt.format(k, p.ref(node)) for k, node in enumerate(low[len(outs):])]
            nodes += list(low[len(outs):])
        if stream and p.fenced:
            import re as _re
            have = set()

            def need_saved(text):
                for mm in _re.finditer(r"\bs(\d+)\b", text):
                    k = int(mm.group(1))
                    if use_saved and k < len(saved) and k not in have:
                        have.add(k)
                        o.append(f"    const double s{k} = in.saved({k});")
            at = {}
            for st, node in zip(stmts, nodes):
                at.setdefault(p.pos.get(node.id, -1), []).append(st)
            for st in at.get(-1, []):
                need_saved(st)
                o.append("    " + st)
            for i, ln in enumerate(p.lines):
                need_saved(ln)
                o.append("    " + ln)
                for st in at.get(i, []):
                    o.append("    " + st)
                if i in at:
                    o.append("    ASSET_SCHED_FENCE();")
        else:
            if use_saved:
                for k in range(len(saved)):
                    o.append(f"    const double s{k} = in.saved({k});")
            o.extend("    " + ln for ln in p.lines)
            o.extend("    " + st for st in stmts)
        o.append("  }")

    def share_order(outs):
        """Greedy order of a part's outputs: next comes the one that shares the most with what is already computed
        (ties: the one that adds the least).  The statements are emitted depth-first per output, so this keeps the
        values several outputs need close to all of their uses: Betts' Hessian part peaks at 294 live values instead
        of 394 in row-major order (column-major 328, random 410-460)."""
        sub = [set(x.id for x in topo_order([r]) if x.args) for _, r in outs]
        done, rem, order = set(), list(range(len(outs))), []
        while rem:
            k = max(rem, key=lambda k: (len(sub[k] & done), -len(sub[k] - done), -k))
            rem.remove(k)
            order.append(outs[k])
            done |= sub[k]
        return order

    def two_parts(name, use_saved):
        outs = outputs(2)
        nfjg = n + n * N + N
        body(name + "_fjg_", outs[:nfjg], use_saved=use_saved, q="__attribute__((noinline))")   # (ordering it too: no change)
        body(name + "_h_", share_order(outs[nfjg:]), use_saved=use_saved, q="__attribute__((noinline))")
        o.append("  " + sig.format(q="inline", name=name) + " {")
        o.append(f"    {name}_fjg_(in, out);")
        o.append(f"    {name}_h_(in, out);")
        o.append("  }")

    # ---- units: a heavy fjgh body is also emitted as NUNITS bodies that partition its outputs by input direction --
    #      unit "column k" = {J[:,k], g[k], H[i>=k,k]}, the cheap columns together with f -- so that the ODE stage can
    #      give every unit its own WAVE (csrc/defect_units.h): each evaluates all of a group's points for its share of the
    #      outputs, recomputing the forward values it needs.  Betts: 7 792 operations as one body (hundreds of spilled
    #      values) against at most ~2 300 per unit.
    units = plan_units(d, outputs(2)) if split else None
    o.insert(-1, f"  static constexpr int NUNITS = {len(units) if units else 1};   // bodies fjgh_unit<U> (1: none, use fjgh)")

    body("f", outputs(0))
    body("fj", outputs(1))
    if units:
        for u, outs in enumerate(units):
            body(f"fjgh_u{u}_", outs, q=UNIT_QUAL, level_order=UNIT_LEVEL_ORDER)
        o.append("  template <int U, class In, class Out> __host__ __device__ static inline void fjgh_unit(const In& in, Out& out) {")
        for u in range(len(units)):
            o.append(f"    {'if' if u == 0 else 'else if'} constexpr (U == {u}) fjgh_u{u}_(in, out);")
        o.append("  }")
        # the Jacobian kinds (evalSOE / evalAUG) use the same partition: unit u emits f (if it holds it) and its columns of J;
        # UNIT_COLS[u] = the input directions it owns (every direction belongs to one unit: the one without a heavy column of
        # its own goes to the unit that holds f)
        f_fmts = {fmt for fmt, _ in outputs(0)}
        j_fmt = {outputs(1)[n + k * N + i][0]: (k, i) for k in range(n) for i in range(N)}
        masks, owned = [], set()
        for u, outs in enumerate(units):
            l1 = [(fmt, r) for fmt, r in outs if fmt in f_fmts or fmt in j_fmt]
            cols = {j_fmt[fmt][1] for fmt, _ in l1 if fmt in j_fmt}
            masks.append(cols)
            owned |= cols
            body(f"fj_u{u}_", l1, q=UNIT_QUAL, level_order=UNIT_LEVEL_ORDER)
        f_unit = next(u for u, outs in enumerate(units) if any(fmt in f_fmts for fmt, _ in outs))
        masks[f_unit] |= set(range(N)) - owned
        o.append("  template <int U, class In, class Out> __host__ __device__ static inline void fj_unit(const In& in, Out& out) {")
        for u in range(len(units)):
            o.append(f"    {'if' if u == 0 else 'else if'} constexpr (U == {u}) fj_u{u}_(in, out);")
        o.append("  }")
        # (64-bit masks: a plain function with more than 32 inputs -- the LGL7 control spline of four controls has 36 -- is cut into
        #  units too; the unit kernels of defect_units.h are for ODEs, whose N = XV + 1 + UV + PV stays below 64)
        o.append(f"  static constexpr unsigned long long UNIT_COLS[{len(units)}] = {{" +
                 ", ".join(str(sum(1 << c for c in m if c < 64)) + "ull" for m in masks) + "};")
        o.append(f"  static constexpr int F_UNIT = {f_unit};")
        # the state components of g = J^T lam alone (a vector-Jacobian product: about three value bodies).  The cardinal adjoint
        # weights w_j (LGLDefects.h:369-374) need g^_i[0:n] of the interior points and nothing else of them, so a workgroup that
        # evaluates a cardinal unit forms it itself instead of waiting for the interior units of a launch before
        # (csrc/defect_units.h, PHASE 4)
        body("gx", [(f"out.g({i}, {{}});", d.g[i]) for i in range(n)], q=UNIT_QUAL, level_order=UNIT_LEVEL_ORDER)
    # (round 5 also emitted the level-2 body in two HALVES for the pair workgroups of the resident kernel; measured slower and
    #  removed in round 6 -- DESIGN.md 4.0b)
    if split:
        two_parts("fjgh", False)
    else:
        body("fjgh", outputs(2))
    # ---- f_save / fjgh_load: the value pass stores every transcendental sub-expression of f; the second-derivative
    #      pass at the SAME point (cardinal nodes: LGLDefects.h:336 then :383-384) loads them instead of recomputing
    body("f_save", outputs(0), extra_roots=saved, extra_stmt="out.save({}, {});")
    # (round 5) ... and the Jacobian with it: the cardinal value phase of the resident kernel then leaves J_j in the slot, and the rows
    # of [J ; g^T] -- which need nothing of the cardinal second-derivative phase -- can be formed and stored while that phase runs
    body("fj_save", outputs(1), extra_roots=saved, extra_stmt="out.save({}, {});")
    if split:
        two_parts("fjgh_load", True)
    else:
        body("fjgh_load", outputs(2), use_saved=True)
    o.append("};")
    return "\n".join(o) + "\n"


MAX_UNITS = 8


def plan_units(d: OdeDerivatives, outs) -> List[list]:
    """Partition the level-2 outputs ([(statement format, root)] in outputs(2) order) into units by input direction:
    unit "column k" = {J[:, k], g[k], H[i >= k, k]} -- everything differentiated in direction k."""
    N, n = d.nin, d.xv
    f_outs = outs[:n]
    j_outs = outs[n:n + n * N]
    g_outs = outs[n + n * N:n + n * N + N]
    h_outs = outs[n + n * N + N:]
    hpos = {}
    e = 0
    for i in range(N):
        for j in range(i + 1):
            hpos[(i, j)] = e
            e += 1

    def cost(group):
        return sum(1 for x in topo_order(lower_reciprocals([r for _, r in group])) if x.args)

    cols = []
    for k in range(N):
        grp = [j_outs[r * N + k] for r in range(n)] + [g_outs[k]] + [h_outs[hpos[(i, k)]] for i in range(k, N)]
        cols.append((cost(grp), k, grp))
    top = max(c for c, _, _ in cols)
    units, light = [], list(f_outs)
    for c, k, grp in sorted(cols, key=lambda t: -t[0]):
        if c >= 0.5 * top and len(units) < MAX_UNITS - 1:
            units.append(grp)
        else:
            light += grp
    units.append(light)
    return units


def saved_nodes(d: OdeDerivatives) -> List[Node]:
    """Transcendental sub-expressions of the value function f (in schedule order)."""
    return [nd for nd in topo_order(d.f) if nd.op in TRANSCENDENTAL]


def emit_c(d: OdeDerivatives, prefix: str) -> str:
    """Plain C: ``<prefix>_f``, ``<prefix>_fj`` (J row-major n x N), ``<prefix>_fjgh`` (H full N x N)."""
    N, n = d.nin, d.xv
    o: List[str] = ["#include <math.h>",
                    f"/* generated by asset_asrl_amd/vf/codegen.py -- ODE '{d.name}' */"]
    tabs = table_arrays(_level_roots(d, 2))
    if tabs:
        o.append(TABLE_HELPERS_C.replace("asset_tab_", f"{prefix}_tab_"))
        for nm, vals in tabs:
            o.append(f"static const double {prefix}_{nm}[{len(vals)}] = {{{', '.join(_cnum(v) for v in vals)}}};")
    sigs = [
        f"void {prefix}_f(const double* y, double* f)",
        f"void {prefix}_fj(const double* y, double* f, double* J)",
        f"void {prefix}_fjgh(const double* y, const double* lam, double* f, double* J, double* g, double* H)",
    ]
    for level, sig in enumerate(sigs):
        low = lower_reciprocals(_level_roots(d, level))
        p = _Printer(low, yname="y[{}]", lname="lam[{}]", tabprefix=f"{prefix}_" if tabs else "")
        it = iter(low)
        o.append(sig + " {")
        o += ["  " + ln for ln in p.lines]
        for k in range(n):
            o.append(f"  f[{k}] = {p.ref(next(it))};")
        if level >= 1:
            for k in range(n):
                for i in range(N):
                    o.append(f"  J[{k * N + i}] = {p.ref(next(it))};")
        if level >= 2:
            for i in range(N):
                o.append(f"  g[{i}] = {p.ref(next(it))};")
            for i in range(N):
                for j in range(i + 1):
                    r = p.ref(next(it))
                    o.append(f"  H[{i * N + j}] = {r};")
                    if i != j:
                        o.append(f"  H[{j * N + i}] = {r};")
        o.append("}")
    o.append(f"const int {prefix}_sizes[3] = {{{d.xv}, {d.uv}, {d.pv}}};")
    return "\n".join(o) + "\n"
