"""Hash-consed scalar expression DAG with symbolic first/second derivatives.

This is the build's own, deliberately small replacement for the part of the
reference's expression/AD system that the collocation hot path needs: a user
ODE ``f(x,t,u,p)`` must be available to the HIP kernels as *inlinable
straight-line device code* for ``f``, ``df/dy`` and ``lam^T d2f/dy2``.  The
reference evaluates the same three quantities by walking a type-erased
expression tree with analytic chain rules (value / jacobian / adjoint-hessian
triplet of every node, /root/reference/src/VectorFunctions/CommonFunctions/
NestedFunction.h:140-270, CwiseOperators.h:31-...).  Here the tree is
differentiated once, symbolically, on a hash-consed DAG (so common
sub-expressions are shared for free) and emitted as code (see codegen.py).

Nothing here is copied from the reference; only the calculus is the same.
"""
from __future__ import annotations

import hashlib
import math
from typing import Dict, Iterable, List, Sequence, Tuple

# --------------------------------------------------------------------------- nodes

_UNARY = ("neg", "sin", "cos", "tan", "exp", "log", "sqrt", "tanh", "asin", "acos", "atan",
          "sinh", "cosh", "abs", "sign")
_BINARY = ("add", "sub", "mul", "div", "atan2")
# conditions (values 1.0 / 0.0, never differentiated) and the selection between two expressions: the reference's
# ConditionalStatement / IfElseFunction (CommonFunctions/Conditional.h:19-150, 151-260) -- value, Jacobian and adjoint Hessian of an
# ifelse are those of the branch the test picks
_COMPARE = ("lt", "le", "gt", "ge")
_LOGIC = ("and", "or")
COND_OPS = _COMPARE + _LOGIC
# tabulated data (the reference's InterpTable1D, CommonFunctions/InterpTable1D.h): "tabloc" = the element of a table's abscissae its
# argument falls into (get_telem, :181-197), "tabget" = an entry of one of the table's arrays at that element or the next.  Both are
# piecewise constant in the argument -- no derivative -- and the interpolant itself is an ordinary expression built on them
# (functions.py: InterpTable1D), so its first and second derivatives come from the rules of this file
TABLE_OPS = ("tabloc", "tabget")
TABLES: Dict[str, object] = {}      # digest -> table (anything with .ts, .vs, .dvs_dts, .teven, .locate): registered by InterpTable1D


class Node:
    """One DAG vertex.  Immutable, unique per (op, args, value) inside its Graph."""

    __slots__ = ("op", "args", "value", "id", "skey")

    def __init__(self, op: str, args: Tuple["Node", ...], value, nid: int):
        self.op = op
        self.args = args
        self.value = value  # float for const, int index for var/lam, exponent for pow/powi
        self.id = nid
        # structural key: depends only on the expression, not on what else the process built before (ids do); the
        # operands of commutative nodes are ordered by it, so the same function always prints as the same code
        self.skey = hashlib.blake2b(f"{op}|{value!r}|{','.join(a.skey for a in args)}".encode(), digest_size=10).hexdigest()

    def is_const(self) -> bool:
        return self.op == "const"

    def __repr__(self):
        if self.op == "const":
            return f"c({self.value!r})"
        if self.op in ("var", "lam", "aconst"):
            return f"{self.op}{self.value}"
        return f"{self.op}#{self.id}"


class Graph:
    """Hash-consing node factory with local algebraic simplification."""

    def __init__(self):
        self._table: Dict[tuple, Node] = {}
        self._dcache: Dict[Tuple[int, int], Node] = {}
        self.zero = self.const(0.0)
        self.one = self.const(1.0)

    # ---- creation -----------------------------------------------------------------
    def _mk(self, op, args=(), value=None) -> Node:
        key = (op, tuple(a.id for a in args), value)
        n = self._table.get(key)
        if n is None:
            n = Node(op, tuple(args), value, len(self._table))
            self._table[key] = n
        return n

    def const(self, v) -> Node:
        v = float(v)
        if v == 0.0:
            v = 0.0  # collapse -0.0
        return self._mk("const", (), v)

    def var(self, i: int) -> Node:
        return self._mk("var", (), int(i))

    def lam(self, k: int) -> Node:
        """Adjoint (multiplier) input; a second family of leaves."""
        return self._mk("lam", (), int(k))

    def cut(self, a: Node) -> Node:
        """A CUT through the expression: numerically the identity, for differentiation a variable of its own.  Where a function is
        composed, ``f(g(x))``, the outputs of ``g`` are wrapped in cuts, and the derivative builders of codegen.py apply the chain
        rule BLOCK-wise across them -- partials of the outer expression with respect to the cut variables, formed once, times the
        total derivatives of the cuts -- the way the reference's NestedFunction combines the Jacobians and adjoint Hessians of its
        two functions (CommonFunctions/NestedFunction.h:140-270), instead of pushing every input direction through the whole
        flattened expression.  ``d`` and ``grad`` below treat a cut as a LEAF (a barrier): they return partial derivatives.
        Trivial arguments (leaves, a leaf times / plus a constant) are not cut."""
        if not a.args or a.op == "cut":
            return a
        if a.op in ("mul", "add", "sub", "neg", "div") and all((not x.args) for x in a.args):
            return a
        return self._mk("cut", (a,))

    def frozen(self) -> Node:
        """A fresh leaf that stands for an expression held CONSTANT in a differentiation (an adjoint weight of the block-wise
        second-order chain rule); replaced by that expression afterwards (``replace``)."""
        self._nfrozen = getattr(self, "_nfrozen", 0) + 1
        return self._mk("frozen", (), self._nfrozen)

    def aconst(self, k: int) -> Node:
        """Constant of the function APPLICATION (a third family of leaves): a number the caller supplies per application
        beside the solver variables -- e.g. the nodal spacing of the reference's SingleMeshSpacing objects, one object per
        state there.  Never differentiated against; its derivative with respect to every input is zero."""
        return self._mk("aconst", (), int(k))

    # ---- arithmetic with folding --------------------------------------------------
    def add(self, a: Node, b: Node) -> Node:
        if a.is_const() and b.is_const():
            return self.const(a.value + b.value)
        if a.is_const() and a.value == 0.0:
            return b
        if b.is_const() and b.value == 0.0:
            return a
        if b.op == "neg":
            return self.sub(a, b.args[0])
        if a.op == "neg":
            return self.sub(b, a.args[0])
        if a.skey > b.skey:
            a, b = b, a
        return self._mk("add", (a, b))

    def sub(self, a: Node, b: Node) -> Node:
        if a.is_const() and b.is_const():
            return self.const(a.value - b.value)
        if b.is_const() and b.value == 0.0:
            return a
        if a.is_const() and a.value == 0.0:
            return self.neg(b)
        if a is b:
            return self.zero
        if b.op == "neg":
            return self.add(a, b.args[0])
        return self._mk("sub", (a, b))

    def neg(self, a: Node) -> Node:
        if a.is_const():
            return self.const(-a.value)
        if a.op == "neg":
            return a.args[0]
        if a.op == "sub":
            return self.sub(a.args[1], a.args[0])
        return self._mk("neg", (a,))

    def mul(self, a: Node, b: Node) -> Node:
        if a.is_const() and b.is_const():
            return self.const(a.value * b.value)
        for p, q in ((a, b), (b, a)):
            if p.is_const():
                if p.value == 0.0:
                    return self.zero
                if p.value == 1.0:
                    return q
                if p.value == -1.0:
                    return self.neg(q)
        if a.op == "neg" and b.op == "neg":
            return self.mul(a.args[0], b.args[0])
        if a.op == "neg":
            return self.neg(self.mul(a.args[0], b))
        if b.op == "neg":
            return self.neg(self.mul(a, b.args[0]))
        if a.skey > b.skey:
            a, b = b, a
        return self._mk("mul", (a, b))

    def div(self, a: Node, b: Node) -> Node:
        if a.is_const() and b.is_const():
            return self.const(a.value / b.value)
        if a.is_const() and a.value == 0.0:
            return self.zero
        if b.is_const() and b.value == 1.0:
            return a
        if b.is_const() and b.value == -1.0:
            return self.neg(a)
        if a.op == "neg":
            return self.neg(self.div(a.args[0], b))
        return self._mk("div", (a, b))

    def powi(self, a: Node, n: int) -> Node:
        n = int(n)
        if n == 0:
            return self.one
        if n == 1:
            return a
        if a.is_const():
            return self.const(a.value ** n)
        if n == 2:
            return self.mul(a, a)
        if n == -1:
            return self.div(self.one, a)
        if n < 0:
            return self.div(self.one, self.powi(a, -n))
        return self._mk("powi", (a,), n)

    def powr(self, a: Node, c: float) -> Node:
        c = float(c)
        if c == int(c) and abs(c) <= 64:
            return self.powi(a, int(c))
        if c == 0.5:
            return self.unary("sqrt", a)
        if a.is_const():
            return self.const(a.value ** c)
        return self._mk("powr", (a,), c)

    def unary(self, op: str, a: Node) -> Node:
        if op == "neg":
            return self.neg(a)
        if a.is_const():
            return self.const(_EVAL_UNARY[op](a.value))
        return self._mk(op, (a,))

    def compare(self, op: str, a: Node, b: Node) -> Node:
        assert op in _COMPARE
        if a.is_const() and b.is_const():
            return self.const(float(_EVAL_COND[op](a.value, b.value)))
        return self._mk(op, (a, b))

    def logic(self, op: str, a: Node, b: Node) -> Node:
        assert op in _LOGIC
        for x, y in ((a, b), (b, a)):
            if x.is_const():
                t = x.value != 0.0
                return (y if t else self.zero) if op == "and" else (self.one if t else y)
        return self._mk(op, (a, b))

    def select(self, c: Node, a: Node, b: Node) -> Node:
        """a where the condition c holds, b elsewhere."""
        if c.is_const():
            return a if c.value != 0.0 else b
        if a is b:
            return a
        if c.op not in COND_OPS:
            raise ValueError("select: the first argument is a condition (a comparison of two scalar expressions)")
        return self._mk("select", (c, a, b))

    def atan2(self, a: Node, b: Node) -> Node:
        if a.is_const() and b.is_const():
            return self.const(math.atan2(a.value, b.value))
        return self._mk("atan2", (a, b))

    def tabloc(self, digest: str, t: Node) -> Node:
        """The element of table `digest` that t falls into, as a number (0 .. tsize-2)."""
        if t.is_const():
            return self.const(float(TABLES[digest].locate(t.value)))
        return self._mk("tabloc", (t,), digest)

    def tabget(self, digest: str, arr: str, row: int, off: int, loc: Node) -> Node:
        """Entry ``[row, loc + off]`` of the table's array `arr` ('t': abscissae, 'v': values, 'd': nodal derivatives)."""
        if loc.is_const():
            return self.const(float(_tab_entry(TABLES[digest], arr, row, int(loc.value) + off)))
        return self._mk("tabget", (loc,), (digest, arr, int(row), int(off)))

    # ---- n-ary helpers -------------------------------------------------------------
    def sum(self, xs: Iterable[Node]) -> Node:
        acc = self.zero
        for x in xs:
            acc = self.add(acc, x)
        return acc

    def dot(self, a: Sequence[Node], b: Sequence[Node]) -> Node:
        return self.sum(self.mul(x, y) for x, y in zip(a, b))

    # ---- symbolic forward derivative ----------------------------------------------
    def d(self, n: Node, wrt: Node) -> Node:
        """d n / d wrt, where wrt is a 'var' or 'lam' leaf.  Memoised on the DAG."""
        key = (n.id, wrt.id)
        r = self._dcache.get(key)
        if r is not None:
            return r
        r = self._d_impl(n, wrt)
        self._dcache[key] = r
        return r

    def _d_impl(self, n: Node, w: Node) -> Node:
        op = n.op
        if op == "const":
            return self.zero
        if op in ("var", "lam"):
            return self.one if n is w else self.zero
        if op == "cut":                        # a barrier: the partial derivative (the total one is the builder's business)
            return self.one if n is w else self.zero
        if op in ("aconst", "frozen") or op in COND_OPS or op in TABLE_OPS:
            return self.zero
        if op == "select":                     # the derivative of the branch the test picks (Conditional.h:215-250); the test itself has none
            return self.select(n.args[0], self.d(n.args[1], w), self.d(n.args[2], w))
        a = n.args[0]
        da = self.d(a, w)
        if op in _UNARY or op in ("powi", "powr"):
            if da is self.zero:
                return self.zero
            return self.mul(self._dunary(n, a), da)
        b = n.args[1]
        db = self.d(b, w)
        if op == "add":
            return self.add(da, db)
        if op == "sub":
            return self.sub(da, db)
        if op == "mul":
            return self.add(self.mul(da, b), self.mul(a, db))
        if op == "div":
            # d(a/b) = da/b - (a/b) db / b   (re-uses the node n itself)
            t1 = self.div(da, b)
            if db is self.zero:
                return t1
            return self.sub(t1, self.div(self.mul(n, db), b))
        if op == "atan2":
            den = self.add(self.mul(a, a), self.mul(b, b))
            return self.div(self.sub(self.mul(b, da), self.mul(a, db)), den)
        raise ValueError(op)

    def _dunary(self, n: Node, a: Node) -> Node:
        """Local partial d n / d a for a unary node n(a)."""
        op = n.op
        if op == "neg":
            return self.const(-1.0)
        if op == "sin":
            return self.unary("cos", a)
        if op == "cos":
            return self.neg(self.unary("sin", a))
        if op == "tan":
            return self.add(self.one, self.mul(n, n))
        if op == "exp":
            return n
        if op == "log":
            return self.div(self.one, a)
        if op == "sqrt":
            return self.div(self.const(0.5), n)
        if op == "tanh":
            return self.sub(self.one, self.mul(n, n))
        if op == "sinh":
            return self.unary("cosh", a)
        if op == "cosh":
            return self.unary("sinh", a)
        if op == "asin":
            return self.div(self.one, self.unary("sqrt", self.sub(self.one, self.mul(a, a))))
        if op == "acos":
            return self.neg(self.div(self.one, self.unary("sqrt", self.sub(self.one, self.mul(a, a)))))
        if op == "atan":
            return self.div(self.one, self.add(self.one, self.mul(a, a)))
        if op == "abs":
            return self.unary("sign", a)
        if op == "sign":
            return self.zero
        if op == "powi":
            k = n.value
            return self.mul(self.const(float(k)), self.powi(a, k - 1))
        if op == "powr":
            c = n.value
            return self.mul(self.const(c), self.powr(a, c - 1.0))
        raise ValueError(op)

    # ---- reverse sweep: gradient of one scalar wrt a set of leaves ------------------
    def grad(self, s: Node, wrts: Sequence[Node]) -> List[Node]:
        """Reverse-mode gradient of scalar node s with respect to the given leaves."""
        order = topo_order([s])
        adj: Dict[int, Node] = {s.id: self.one}
        for n in reversed(order):
            bar = adj.get(n.id)
            if bar is None or bar is self.zero or not n.args:
                continue
            op = n.op
            if op == "cut":                    # a barrier: its adjoint is read off by the caller, nothing flows into its argument
                continue
            if op in COND_OPS or op in TABLE_OPS:
                continue
            if op == "select":
                c, a, b = n.args
                self._acc(adj, a, self.select(c, bar, self.zero))
                self._acc(adj, b, self.select(c, self.zero, bar))
                continue
            if op in _UNARY or op in ("powi", "powr"):
                a = n.args[0]
                self._acc(adj, a, self.mul(bar, self._dunary(n, a)))
            elif op == "add":
                self._acc(adj, n.args[0], bar)
                self._acc(adj, n.args[1], bar)
            elif op == "sub":
                self._acc(adj, n.args[0], bar)
                self._acc(adj, n.args[1], self.neg(bar))
            elif op == "mul":
                a, b = n.args
                self._acc(adj, a, self.mul(bar, b))
                self._acc(adj, b, self.mul(bar, a))
            elif op == "div":
                a, b = n.args
                t = self.div(bar, b)
                self._acc(adj, a, t)
                self._acc(adj, b, self.neg(self.mul(t, n)))
            elif op == "atan2":
                a, b = n.args
                den = self.add(self.mul(a, a), self.mul(b, b))
                t = self.div(bar, den)
                self._acc(adj, a, self.mul(t, b))
                self._acc(adj, b, self.neg(self.mul(t, a)))
            else:
                raise ValueError(op)
        return [adj.get(w.id, self.zero) for w in wrts]

    def _acc(self, adj, n: Node, v: Node):
        if n.is_const():
            return
        cur = adj.get(n.id)
        adj[n.id] = v if cur is None else self.add(cur, v)

    # ---- substitution (function composition) ---------------------------------------
    def substitute(self, roots: Sequence[Node], var_map: Dict[int, Node]) -> List[Node]:
        """Rebuild roots with every 'var i' leaf replaced by var_map[i]."""
        memo: Dict[int, Node] = {}
        for n in topo_order(roots):
            if n.op == "var":
                if n.value not in var_map:
                    raise IndexError(f"composition: inner function has no output {n.value}")
                memo[n.id] = var_map[n.value]
            elif not n.args:
                memo[n.id] = n
            else:
                memo[n.id] = self.rebuild(n, [memo[a.id] for a in n.args])
        return [memo[r.id] for r in roots]

    def replace(self, roots: Sequence[Node], mapping: Dict[int, Node]) -> List[Node]:
        """Rebuild roots with every node whose id is a key of `mapping` replaced by the mapped node (cuts are kept)."""
        memo: Dict[int, Node] = {}
        for n in topo_order(roots):
            if n.id in mapping:
                memo[n.id] = mapping[n.id]
            elif not n.args:
                memo[n.id] = n
            else:
                memo[n.id] = self.rebuild(n, [memo[a.id] for a in n.args])
        return [memo[r.id] for r in roots]

    def strip_cuts(self, roots: Sequence[Node]) -> List[Node]:
        """The same expressions without their cuts (what is printed: a cut is the identity)."""
        memo: Dict[int, Node] = {}
        for n in topo_order(roots):
            if not n.args:
                memo[n.id] = n
            elif n.op == "cut":
                memo[n.id] = memo[n.args[0].id]
            else:
                memo[n.id] = self.rebuild(n, [memo[a.id] for a in n.args])
        return [memo[r.id] for r in roots]

    def rebuild(self, n: Node, args: Sequence[Node]) -> Node:
        op = n.op
        if op == "cut":
            return self.cut(args[0])
        if op in _COMPARE:
            return self.compare(op, *args)
        if op in _LOGIC:
            return self.logic(op, *args)
        if op == "select":
            return self.select(*args)
        if op == "tabloc":
            return self.tabloc(n.value, args[0])
        if op == "tabget":
            return self.tabget(*n.value, args[0])
        if op == "add":
            return self.add(*args)
        if op == "sub":
            return self.sub(*args)
        if op == "mul":
            return self.mul(*args)
        if op == "div":
            return self.div(*args)
        if op == "atan2":
            return self.atan2(*args)
        if op == "powi":
            return self.powi(args[0], n.value)
        if op == "powr":
            return self.powr(args[0], n.value)
        return self.unary(op, args[0])


def _tab_entry(tab, arr: str, row: int, col: int) -> float:
    if arr == "t":
        return tab.ts[col]
    return (tab.vs if arr == "v" else tab.dvs_dts)[row, col]


_EVAL_COND = {"lt": lambda a, b: a < b, "le": lambda a, b: a <= b, "gt": lambda a, b: a > b, "ge": lambda a, b: a >= b}

_EVAL_UNARY = {
    "sin": math.sin, "cos": math.cos, "tan": math.tan, "exp": math.exp, "log": math.log,
    "sqrt": math.sqrt, "tanh": math.tanh, "asin": math.asin, "acos": math.acos, "atan": math.atan,
    "sinh": math.sinh, "cosh": math.cosh, "abs": abs,
    "sign": lambda v: (1.0 if v > 0 else (-1.0 if v < 0 else 0.0)),
}


def topo_order(roots: Sequence[Node]) -> List[Node]:
    """Children-before-parents order of everything reachable from roots (iterative DFS)."""
    seen = set()
    out: List[Node] = []
    for r in roots:
        if r.id in seen:
            continue
        stack = [(r, 0)]
        while stack:
            n, i = stack.pop()
            if i == 0:
                if n.id in seen:
                    continue
                seen.add(n.id)
            if i < len(n.args):
                stack.append((n, i + 1))
                c = n.args[i]
                if c.id not in seen:
                    stack.append((c, 0))
            else:
                out.append(n)
    return out


def frontier(roots: Sequence[Node]):
    """(var leaves, cut nodes) reachable from roots WITHOUT crossing a cut: what the expressions depend on directly when cuts are
    variables.  Both in a deterministic order (var index / first visit)."""
    seen, vs, cs = set(), {}, []
    stack = list(reversed(list(roots)))
    while stack:
        n = stack.pop()
        if n.id in seen:
            continue
        seen.add(n.id)
        if n.op == "var":
            vs[n.value] = n
        elif n.op == "cut":
            cs.append(n)
        else:
            stack.extend(reversed(n.args))
    return [vs[k] for k in sorted(vs)], cs


def evaluate(roots: Sequence[Node], y: Sequence[float], lam: Sequence[float] = (), aconst: Sequence[float] = ()) -> List[float]:
    """Host-side numeric walk of the DAG (set-up time checks only, never the hot path)."""
    val: Dict[int, float] = {}
    for n in topo_order(roots):
        op = n.op
        if op == "const":
            v = n.value
        elif op == "var":
            v = float(y[n.value])
        elif op == "lam":
            v = float(lam[n.value])
        elif op == "aconst":
            v = float(aconst[n.value])
        elif op == "add":
            v = val[n.args[0].id] + val[n.args[1].id]
        elif op == "sub":
            v = val[n.args[0].id] - val[n.args[1].id]
        elif op == "mul":
            v = val[n.args[0].id] * val[n.args[1].id]
        elif op == "div":
            v = val[n.args[0].id] / val[n.args[1].id]
        elif op == "neg":
            v = -val[n.args[0].id]
        elif op == "powi":
            v = val[n.args[0].id] ** n.value
        elif op == "powr":
            v = val[n.args[0].id] ** n.value
        elif op == "atan2":
            v = math.atan2(val[n.args[0].id], val[n.args[1].id])
        elif op == "cut":
            v = val[n.args[0].id]
        elif op in _COMPARE:
            v = float(_EVAL_COND[op](val[n.args[0].id], val[n.args[1].id]))
        elif op == "and":
            v = float(val[n.args[0].id] != 0.0 and val[n.args[1].id] != 0.0)
        elif op == "or":
            v = float(val[n.args[0].id] != 0.0 or val[n.args[1].id] != 0.0)
        elif op == "select":
            v = val[n.args[1].id] if val[n.args[0].id] != 0.0 else val[n.args[2].id]
        elif op == "tabloc":
            v = float(TABLES[n.value].locate(val[n.args[0].id]))
        elif op == "tabget":
            dg, arr, row, off = n.value
            v = float(_tab_entry(TABLES[dg], arr, row, int(val[n.args[0].id]) + off))
        else:
            v = _EVAL_UNARY[op](val[n.args[0].id])
        val[n.id] = v
    return [val[r.id] for r in roots]


GRAPH = Graph()
"""Process-wide graph: every VectorFunction of a session shares its leaves."""
