"""Python-visible VectorFunction surface (subset) over the expression DAG.

Mirrors the *names and argument meaning* of the reference's ``asset.VectorFunctions``
objects that ODE definitions use (``Arguments``, ``head/tail/segment``, ``norm``,
``normalized``, ``normalized_power3``, ``cross``, ``dot``, ``stack``, ``sum``,
``sin``/``cos``/..., operator overloads, ``F(G)`` composition, ``RowMatrix``) --
see /root/reference/src/VectorFunctions/DenseFunctionBase.h:1644-2101 (operator and
method binds) and /root/reference/asset_asrl/VectorFunctions/__init__.py:7-58.
The objects here do not evaluate anything on the CPU hot path: they only build the
DAG that codegen.py turns into HIP device code.
"""
from __future__ import annotations

from typing import List, Sequence, Union

import numpy as np

from .ir import GRAPH as G
from .ir import Node, evaluate

Number = Union[int, float, np.floating, np.integer]
builtins_abs = abs      # (the module defines ``abs`` as a function of expressions further down)


def _is_num(x) -> bool:
    return isinstance(x, (int, float, np.floating, np.integer))


class VectorFunction:
    """A map R^irows -> R^orows given as DAG roots over the leaves var0..var{irows-1}."""

    __array_ufunc__ = None  # numpy arrays defer to our reflected operators

    def __init__(self, irows: int, outs: Sequence[Node]):
        self._irows = int(irows)
        self.outs: List[Node] = list(outs)

    # ---- sizes (reference names) ---------------------------------------------------
    def IRows(self) -> int:
        return self._irows

    def ORows(self) -> int:
        return len(self.outs)

    def name(self) -> str:
        return f"VectorFunction<{self._irows},{len(self.outs)}>"

    def is_scalar(self) -> bool:
        return len(self.outs) == 1

    # ---- construction helpers ------------------------------------------------------
    def _like(self, outs: Sequence[Node]) -> "VectorFunction":
        return VectorFunction(self._irows, outs)

    def _coerce(self, other, n: int) -> List[Node]:
        """Broadcast a python number / numpy vector / function to n DAG nodes."""
        if isinstance(other, VectorFunction):
            if other._irows != self._irows:
                raise ValueError("Functions must have the same input size")
            if other.ORows() == n:
                return other.outs
            if other.ORows() == 1:
                return other.outs * n
            raise ValueError("Output sizes do not match")
        if _is_num(other):
            return [G.const(other)] * n
        arr = np.asarray(other, dtype=float).ravel()
        if arr.size != n:
            raise ValueError("Vector size does not match function output size")
        return [G.const(v) for v in arr]

    # ---- indexing ------------------------------------------------------------------
    def segment(self, start: int, size: int) -> "VectorFunction":
        if start < 0 or size < 0 or start + size > self.ORows():
            raise ValueError("Segment index out of bounds")
        return self._like(self.outs[start:start + size])

    def head(self, n: int) -> "VectorFunction":
        return self.segment(0, n)

    def tail(self, n: int) -> "VectorFunction":
        return self.segment(self.ORows() - n, n)

    def head2(self):
        return self.head(2)

    def head3(self):
        return self.head(3)

    def tail2(self):
        return self.tail(2)

    def tail3(self):
        return self.tail(3)

    def segment2(self, start: int):
        return self.segment(start, 2)

    def segment3(self, start: int):
        return self.segment(start, 3)

    def coeff(self, i: int) -> "VectorFunction":
        return self.segment(i, 1)

    def __getitem__(self, i):
        if isinstance(i, slice):
            return self._like(self.outs[i])
        if i < 0:
            i += self.ORows()
        return self.coeff(i)

    def tolist(self, spec=None):
        """Scalar components, or sub-vectors for a list of (start,size) pairs."""
        if spec is None:
            return [self.coeff(i) for i in range(self.ORows())]
        return [self.segment(s, n) for (s, n) in spec]

    # ---- comparisons (scalar functions): conditions for ``ifelse`` (ConditionalStatement, CommonFunctions/Conditional.h:19-150) ----
    def _cmp(self, other, op):
        if self.ORows() != 1:
            raise ValueError("LHS and RHS of conditional statement must be scalar functions")
        return Condition(self._irows, G.compare(op, self.outs[0], self._coerce(other, 1)[0]))

    def __lt__(self, other):
        return self._cmp(other, "lt")

    def __le__(self, other):
        return self._cmp(other, "le")

    def __gt__(self, other):
        return self._cmp(other, "gt")

    def __ge__(self, other):
        return self._cmp(other, "ge")

    # ---- composition ---------------------------------------------------------------
    def eval(self, inner: "VectorFunction") -> "VectorFunction":
        """self(inner(x)) -- the reference's ``F.eval(G)`` / ``F(G)``."""
        if inner.ORows() != self._irows:
            raise ValueError("Inner function output size does not match outer input size")
        # (the inner function's outputs become cut variables of the expression: codegen.py differentiates across them block-wise,
        #  the reference's NestedFunction chain rule -- numerically a cut is the identity)
        outs = G.substitute(self.outs, {i: G.cut(n) for i, n in enumerate(inner.outs)})
        return VectorFunction(inner._irows, outs)

    def cut(self) -> "VectorFunction":
        """This function's outputs as cut variables (``ir.Graph.cut``): whatever is built on top of them is differentiated
        block-wise across them.  For an intermediate vector with FEWER components than the inputs it depends on -- a position, a
        perturbing acceleration -- that is cheaper than differentiating the flattened expression; the derivative builder counts
        both and keeps the cheaper form."""
        return self._like([G.cut(n) for n in self.outs])      # (explicit: no size rule)

    def __call__(self, arg):
        if isinstance(arg, VectorFunction):
            return self.eval(arg)
        return self.compute(arg)

    # ---- host-side numeric evaluation (set-up / tests only) ------------------------
    def compute(self, x, appl_consts=()) -> np.ndarray:
        x = np.asarray(x, dtype=float).ravel()
        if x.size != self._irows:
            raise ValueError("Input vector has incorrect size")
        return np.array(evaluate(self.outs, x, (), appl_consts))

    # ---- arithmetic ----------------------------------------------------------------
    def _bin(self, other, fn, swap=False):
        n = self.ORows()
        if isinstance(other, VectorFunction) and other.ORows() != n and n == 1:
            n = other.ORows()
        a = self._coerce(self, n)
        b = self._coerce(other, n)
        if swap:
            a, b = b, a
        return self._like([fn(x, y) for x, y in zip(a, b)])

    def __add__(self, o):
        return self._bin(o, G.add)

    __radd__ = __add__

    def __sub__(self, o):
        return self._bin(o, G.sub)

    def __rsub__(self, o):
        return self._bin(o, G.sub, swap=True)

    def __mul__(self, o):
        return self._bin(o, G.mul)

    __rmul__ = __mul__

    def __truediv__(self, o):
        return self._bin(o, G.div)

    def __rtruediv__(self, o):
        return self._bin(o, G.div, swap=True)

    def __neg__(self):
        return self._like([G.neg(x) for x in self.outs])

    def __pow__(self, p):
        if not _is_num(p):
            raise TypeError("only constant exponents are supported")
        return self._like([G.powr(x, float(p)) for x in self.outs])

    def _map(self, op: str):
        return self._like([G.unary(op, x) for x in self.outs])

    def squared(self):
        return self * self

    def sqrt(self):
        return self._map("sqrt")

    def exp(self):
        return self._map("exp")

    def log(self):
        return self._map("log")

    def sin(self):
        return self._map("sin")

    def cos(self):
        return self._map("cos")

    def tan(self):
        return self._map("tan")

    def tanh(self):
        return self._map("tanh")

    # ---- vector algebra ------------------------------------------------------------
    # (Measured, round 6: cutting the OPERANDS of these vector operations -- the reference's norm / normalized / cross / dot are
    #  functions of their own, nested over whatever produced their arguments -- makes the block-wise derivative builder of
    #  codegen.py WORSE:
    #  the Cartesian-detour low-thrust model of round 4 is 7 792 operations flat, 6 093 with cuts at its F(G) compositions
    #  only, 10 763 with the operands cut as well -- and 9 635 with the level-local elimination (_differentiate_block_local), still
    #  above the flat form: on a hash-consed graph the flat derivative already shares what a fine-grained nest would share, and
    #  every extra cut adds the products that join its two sides.  Cuts are therefore made at compositions and where a definition
    #  says ``.cut()``.)
    @staticmethod
    def _operands(nodes):
        return list(nodes)

    def sum(self):
        return self._like([G.sum(self.outs)])

    def dot(self, other):
        b = self._coerce(other, self.ORows())
        return self._like([G.dot(self._operands(self.outs), self._operands(b))])

    def squared_norm(self):
        a = self._operands(self.outs)
        return self._like([G.dot(a, a)])

    def norm(self):
        a = self._operands(self.outs)
        return self._like([G.unary("sqrt", G.dot(a, a))])

    def inverse_norm(self):
        a = self._operands(self.outs)
        return self._like([G.div(G.one, G.unary("sqrt", G.dot(a, a)))])

    def cubed_norm(self):
        a = self._operands(self.outs)
        n = G.unary("sqrt", G.dot(a, a))
        return self._like([G.powi(n, 3)])

    def normalized(self):
        a = self._operands(self.outs)
        n = G.unary("sqrt", G.dot(a, a))
        return self._like([G.div(x, n) for x in a])

    def _normalized_power(self, k: int):
        a = self._operands(self.outs)
        n = G.unary("sqrt", G.dot(a, a))
        nk = G.powi(n, k)
        return self._like([G.div(x, nk) for x in a])

    def normalized_power2(self):
        return self._normalized_power(2)

    def normalized_power3(self):
        return self._normalized_power(3)

    def normalized_power4(self):
        return self._normalized_power(4)

    def normalized_power5(self):
        return self._normalized_power(5)

    def cross(self, other):
        if self.ORows() != 3:
            raise ValueError("cross requires 3-vectors")
        a = self._operands(self.outs)
        b = self._operands(self._coerce(other, 3))
        return self._like([
            G.sub(G.mul(a[1], b[2]), G.mul(a[2], b[1])),
            G.sub(G.mul(a[2], b[0]), G.mul(a[0], b[2])),
            G.sub(G.mul(a[0], b[1]), G.mul(a[1], b[0])),
        ])

    def cwiseProduct(self, other):
        return self * other

    def cwiseQuotient(self, other):
        return self / other


class Condition:
    """A test on the inputs: a comparison of two scalar functions, or two conditions joined by ``&`` / ``|`` (the reference's
    ConditionalStatement with ANDFlag / ORFlag).  Consumed by :func:`ifelse`."""

    def __init__(self, irows: int, node: Node):
        self._irows, self.node = int(irows), node

    def IRows(self) -> int:
        return self._irows

    def _join(self, other, op):
        if not isinstance(other, Condition) or other._irows != self._irows:
            raise ValueError("conditions over the same inputs can be joined")
        return Condition(self._irows, G.logic(op, self.node, other.node))

    def __and__(self, other):
        return self._join(other, "and")

    def __or__(self, other):
        return self._join(other, "or")

    def compute(self, x) -> bool:
        return bool(evaluate([self.node], np.asarray(x, dtype=float).ravel())[0])


def ifelse(test: Condition, true_func, false_func) -> VectorFunction:
    """``true_func(x)`` where ``test(x)`` holds, ``false_func(x)`` elsewhere (the reference's ``vf.ifelse``, IfElseFunction,
    CommonFunctions/Conditional.h:151-260): value, Jacobian and adjoint Hessian are those of the branch the test picks -- on the device
    a select per emitted quantity, both branches evaluated (the lanes of a wave take different branches)."""
    if not isinstance(test, Condition):
        raise ValueError("ifelse: the test is a comparison of two scalar functions (f > g, f <= 0.0, (a > b) & (c < d))")
    fs = [f for f in (true_func, false_func) if isinstance(f, VectorFunction)]
    if not fs:
        raise ValueError("ifelse: at least one branch is a function")
    n = fs[0].ORows()
    a, b = fs[0]._coerce(true_func, n), fs[0]._coerce(false_func, n)
    if any(f._irows != test._irows for f in fs):
        raise ValueError("Test,True,and False functions in conditional statement must have same number of inputrows.")
    return VectorFunction(test._irows, [G.select(test.node, x, y) for x, y in zip(a, b)])


class MatrixFunction:
    """Row/Col-major matrix view of a VectorFunction (``vf.RowMatrix`` / ``vf.ColMatrix``)."""

    def __init__(self, f: VectorFunction, rows: int, cols: int, rowmajor: bool):
        if f.ORows() != rows * cols:
            raise ValueError("matrix dimensions do not match function output size")
        self.f, self.rows, self.cols, self.rowmajor = f, rows, cols, rowmajor

    def _elem(self, r, c) -> Node:
        k = r * self.cols + c if self.rowmajor else c * self.rows + r
        return self.f.outs[k]

    def __mul__(self, v):
        b = self.f._coerce(v, self.cols)
        outs = [G.sum(G.mul(self._elem(r, c), b[c]) for c in range(self.cols)) for r in range(self.rows)]
        return self.f._like(outs)


# --------------------------------------------------------------------------- free functions

def Arguments(n: int) -> VectorFunction:
    return VectorFunction(n, [G.var(i) for i in range(n)])


def ApplConst(irows: int, k: int = 0) -> VectorFunction:
    """Scalar function (of `irows` inputs) whose value is the k-th constant of the function application: data the caller
    supplies per application beside the solver vector (``FunctionEvaluator(..., appl_consts=array[napp, nconst])``).  It
    has no derivative, so it adds no column to the function's Jacobian / Hessian blocks."""
    return VectorFunction(irows, [G.aconst(k)])


def Segment(irows: int, size: int, start: int) -> VectorFunction:
    return Arguments(irows).segment(start, size)


def Element(irows: int, i: int) -> VectorFunction:
    return Arguments(irows).coeff(i)


def _flatten(args) -> List:
    if len(args) == 1 and isinstance(args[0], (list, tuple)):
        return list(args[0])
    return list(args)


def stack(*args) -> VectorFunction:
    fs = _flatten(args)
    ir = next(f._irows for f in fs if isinstance(f, VectorFunction))
    outs: List[Node] = []
    for f in fs:
        if isinstance(f, VectorFunction):
            if f._irows != ir:
                raise ValueError("Functions must have the same input size")
            outs.extend(f.outs)
        elif _is_num(f):
            outs.append(G.const(f))
        else:
            outs.extend(G.const(v) for v in np.asarray(f, dtype=float).ravel())
    return VectorFunction(ir, outs)


Stack = stack
stack_scalar = stack


def sum(*args) -> VectorFunction:  # noqa: A001  (reference name)
    fs = _flatten(args)
    acc = fs[0]
    for f in fs[1:]:
        acc = acc + f
    return acc


Sum = sum


def _u(op):
    def fn(f: VectorFunction) -> VectorFunction:
        return f._map(op)
    fn.__name__ = op
    return fn


sin, cos, tan, exp, log, sqrt, tanh = (_u(o) for o in ("sin", "cos", "tan", "exp", "log", "sqrt", "tanh"))
sinh, cosh = _u("sinh"), _u("cosh")
arcsin, arccos, arctan = _u("asin"), _u("acos"), _u("atan")
abs = _u("abs")  # noqa: A001
sign = _u("sign")       # -1 / 0 / +1 (the reference's SignFunction.h): piecewise constant, no derivative


def arctan2(y: VectorFunction, x: VectorFunction) -> VectorFunction:
    return y._like([G.atan2(a, b) for a, b in zip(y.outs, x.outs)])


def dot(a: VectorFunction, b) -> VectorFunction:
    return a.dot(b)


def cross(a: VectorFunction, b) -> VectorFunction:
    return a.cross(b)


def cwiseProduct(a, b):
    return a * b


def cwiseQuotient(a, b):
    return a / b


def RowMatrix(f: VectorFunction, rows: int, cols: int) -> MatrixFunction:
    return MatrixFunction(f, rows, cols, True)


def ColMatrix(f: VectorFunction, rows: int, cols: int) -> MatrixFunction:
    return MatrixFunction(f, rows, cols, False)


def IOScaled(func: VectorFunction, input_scales, output_scales) -> VectorFunction:
    """``output_scales * func(input_scales * x)`` (element-wise) -- the reference's IOScaled wrapper
    (/root/reference/src/VectorFunctions/CommonFunctions/IOScaled.h:8-158).  Here it is a composition in the
    expression graph, so the scaled ODE is differentiated and compiled like any other and the scale factors fold
    into the generated device code."""
    n = func.IRows()
    si = np.asarray(input_scales, dtype=float).ravel()
    so = np.asarray(output_scales, dtype=float).ravel()
    if si.size != n or so.size != func.ORows():
        raise ValueError("Incorrect size for input / output scales")
    return func.eval(Arguments(n) * si) * so


def ConstantVector(irows: int, v) -> VectorFunction:
    return VectorFunction(irows, [G.const(x) for x in np.asarray(v, dtype=float).ravel()])


def ConstantScalar(irows: int, v: float) -> VectorFunction:
    return VectorFunction(irows, [G.const(v)])


class InterpTable1D:
    """Tabulated data ``v(t)`` interpolated along one axis, cubic (Hermite, nodal slopes from five-point differences) or linear,
    usable as a number-in / number-out interpolant on the host and as a node of an ODE or function expression -- the reference's
    ``InterpTable1D`` and ``InterpFunction1D`` (/root/reference/src/VectorFunctions/CommonFunctions/InterpTable1D.h:9-320, 322-401).

    Constructors as the reference binds them (:409-424): ``(ts, Vs, axis=0, kind="cubic")`` with ``Vs`` a vector or a matrix whose
    `axis` runs along ``ts`` (0: rows are samples), or ``(Vts, tvar=-1, kind="cubic")`` with a list of value-time vectors.

    In an expression (``tab(tfunc)``, ``tab.sf()``, ``tab.vf()``) the table becomes CONSTANT ARRAYS of the compiled module and two
    piecewise-constant nodes -- the element ``t`` falls into, and entries of the arrays at that element (vf/ir.py: tabloc / tabget);
    the Hermite polynomial on top of them is an ordinary expression, so value, Jacobian and adjoint Hessian of anything built on a
    table come from the same symbolic rules as every other node (the reference hand-writes them: :219-256, :337-400)."""

    def __init__(self, *args, **kw):
        names = ("ts", "Vs", "axis", "kind")
        if args and isinstance(args[0], (list, tuple)) and len(args[0]) and np.ndim(args[0][0]) == 1 and (len(args) < 2 or _is_num(args[1]) or isinstance(args[1], str)) \
                and "Vs" not in kw:
            # (Vts, tvar=-1, kind="cubic"): InterpTable1D.h:43-79
            Vts = [np.asarray(v, dtype=float).ravel() for v in args[0]]
            rest = list(args[1:])
            tvar = kw.get("tvar", rest.pop(0) if rest and _is_num(rest[0]) else -1)
            kind = kw.get("kind", rest.pop(0) if rest else "cubic")
            if len(Vts) == 0:
                raise ValueError("Input is empty")
            m = Vts[0].size
            if m < 2:
                raise ValueError("Invalid sized value-time data.")
            if tvar < 0:
                tvar += m
            if tvar > m - 1 or tvar < 0:
                raise ValueError("Invalid time variable index")
            if any(v.size != m for v in Vts):
                raise ValueError("All value-time vectors must have same size")
            A = np.stack(Vts, axis=1)                         # [m, samples]
            self._set_data(A[tvar], np.delete(A, tvar, axis=0), 1, kind)
            return
        a = dict(zip(names, args))
        a.update(kw)
        Vs = np.asarray(a["Vs"], dtype=float)
        if Vs.ndim == 1:                                        # a vector of values: one output (:39-42)
            self._set_data(a["ts"], Vs[None, :], 1, a.get("kind", "cubic"))
        else:
            self._set_data(a["ts"], Vs, a.get("axis", 0), a.get("kind", "cubic"))

    # ---- InterpTable1D.h:80-131
    def _set_data(self, ts, Vs, axis, kind):
        self.ts = np.asarray(ts, dtype=float).ravel().copy()
        if axis == 1:
            self.vs = np.array(Vs, dtype=float)
        elif axis == 0:
            self.vs = np.array(Vs, dtype=float).T.copy()
        else:
            raise ValueError("Interpolation axis must be 0 or 1")
        self.axis = int(axis)
        if kind in ("cubic", "Cubic"):
            self.kind = "cubic"
        elif kind in ("linear", "Linear"):
            self.kind = "linear"
        else:
            raise ValueError("Unrecognized interpolation type")
        self.tsize = int(self.ts.size)
        self.vlen = int(self.vs.shape[0])
        if self.tsize < 5:
            raise ValueError("t coordinates must be larger than 4")
        if self.tsize != self.vs.shape[1]:
            raise ValueError("Length of t coordinates must match length of interpolation axis")
        if np.any(np.diff(self.ts) < 0):
            raise ValueError("t Coordinates must be in ascending order")
        self.ttotal = float(self.ts[-1] - self.ts[0])
        even = np.linspace(self.ts[0], self.ts[-1], self.tsize)
        self.teven = bool(np.max(np.abs(self.ts - even)) <= builtins_abs(self.ttotal) * 1.0e-12)
        self.WarnOutOfBounds, self.ThrowOutOfBounds = True, False
        self.dvs_dts = self._nodal_slopes() if self.kind == "cubic" else np.zeros_like(self.vs)
        import hashlib
        h = hashlib.blake2b(digest_size=8)
        for part in (self.kind.encode(), self.ts.tobytes(), np.ascontiguousarray(self.vs).tobytes(), str(self.vs.shape).encode()):
            h.update(part)
        self.digest = h.hexdigest()
        from .ir import TABLES
        TABLES[self.digest] = self

    def _nodal_slopes(self):
        """dv/dt at every abscissa from the five-point difference formula exact for quartics on the points around it (centred inside,
        one-sided at the two ends on either side): the weights solve the 5 x 5 moment system in the scaled offsets
        (InterpTable1D.h:134-179)."""
        n = self.tsize
        d = np.empty_like(self.vs)
        rhs = np.array([0.0, 1.0, 0.0, 0.0, 0.0])
        for i in range(n):
            if 2 <= i <= n - 3:
                start = i - 2
            elif i < n - 1 - i:
                start = 0
            else:
                start = n - 5
            step = builtins_abs(self.ts[i + (1 if i < n - 1 else -1)] - self.ts[i])
            off = (self.ts[start:start + 5] - self.ts[i]) / step
            w = np.linalg.solve(np.vander(off, 5, increasing=True).T, rhs)
            d[:, i] = self.vs[:, start:start + 5] @ (w / step)
        return d

    # ---- InterpTable1D.h:181-197
    def locate(self, t: float) -> int:
        if self.teven:
            e = int((t - self.ts[0]) / (self.ts[1] - self.ts[0]))
        else:
            e = int(np.searchsorted(self.ts, t, side="right")) - 1
        return max(min(e, self.tsize - 2), 0)

    def _check_bounds(self, t):
        if self.WarnOutOfBounds or self.ThrowOutOfBounds:
            eps = np.finfo(float).eps * self.ttotal
            if t < self.ts[0] - eps or t > self.ts[-1] + eps:
                msg = f"WARNING: t= {t} falls outside of InterpTable1D time range. Data is being extrapolated!!"
                if self.ThrowOutOfBounds:
                    raise ValueError(msg)
                import warnings
                warnings.warn(msg)

    def _interp(self, t: float, deriv: int):
        """(v, dv/dt, d2v/dt2) up to `deriv`: InterpTable1D.h:199-267."""
        t = float(t)
        self._check_bounds(t)
        e = self.locate(t)
        step = self.ts[e + 1] - self.ts[e]
        x = (t - self.ts[e]) / step
        v0, v1, d0, d1 = self.vs[:, e], self.vs[:, e + 1], self.dvs_dts[:, e], self.dvs_dts[:, e + 1]
        out = []
        if self.kind == "cubic":
            x2, x3 = x * x, x * x * x
            out.append(v0 * (2 * x3 - 3 * x2 + 1) + v1 * (-2 * x3 + 3 * x2) + d0 * ((x3 - 2 * x2 + x) * step) + d1 * ((x3 - x2) * step))
            if deriv > 0:
                out.append(v0 * ((6 * x2 - 6 * x) / step) + v1 * ((-6 * x2 + 6 * x) / step) + d0 * (3 * x2 - 4 * x + 1) + d1 * (3 * x2 - 2 * x))
            if deriv > 1:
                out.append(v0 * ((12 * x - 6) / step ** 2) + v1 * ((-12 * x + 6) / step ** 2) + d0 * ((6 * x - 4) / step) + d1 * ((6 * x - 2) / step))
        else:
            out.append(v0 * (1 - x) + v1 * x)
            if deriv > 0:
                out.append((v1 - v0) / step)
            if deriv > 1:
                out.append(np.zeros(self.vlen))
        return out

    def interp(self, t):
        if np.ndim(t) == 0:
            return self._interp(t, 0)[0]
        return np.stack([self._interp(x, 0)[0] for x in np.asarray(t, dtype=float).ravel()], axis=1)

    def interp_deriv1(self, t):
        return tuple(self._interp(t, 1))

    def interp_deriv2(self, t):
        return tuple(self._interp(t, 2))

    # ---- as an expression: InterpFunction1D (InterpTable1D.h:322-401) and the binds at :434-479
    def _nodes(self, t: Node) -> List[Node]:
        dg = self.digest
        loc = G.tabloc(dg, t)
        t0 = G.tabget(dg, "t", 0, 0, loc)
        if self.teven:
            step = G.const(self.ts[1] - self.ts[0])
        else:
            step = G.sub(G.tabget(dg, "t", 0, 1, loc), t0)
        x = G.div(G.sub(t, t0), step)
        outs = []
        if self.kind == "cubic":
            x2 = G.mul(x, x)
            x3 = G.mul(x2, x)
            c = G.const
            p0 = G.add(G.sub(G.mul(c(2.0), x3), G.mul(c(3.0), x2)), G.one)
            p1 = G.sub(G.mul(c(3.0), x2), G.mul(c(2.0), x3))
            m0 = G.mul(G.add(G.sub(x3, G.mul(c(2.0), x2)), x), step)
            m1 = G.mul(G.sub(x3, x2), step)
        for k in range(self.vlen):
            v0, v1 = G.tabget(dg, "v", k, 0, loc), G.tabget(dg, "v", k, 1, loc)
            if self.kind == "cubic":
                d0, d1 = G.tabget(dg, "d", k, 0, loc), G.tabget(dg, "d", k, 1, loc)
                outs.append(G.sum([G.mul(v0, p0), G.mul(v1, p1), G.mul(d0, m0), G.mul(d1, m1)]))
            else:
                outs.append(G.add(G.mul(v0, G.sub(G.one, x)), G.mul(v1, x)))
        return outs

    def __call__(self, t):
        if isinstance(t, VectorFunction):
            if t.ORows() != 1:
                raise ValueError("InterpTable1D: the argument is a scalar function")
            return VectorFunction(t._irows, self._nodes(t.outs[0]))
        return self.interp(t)

    def sf(self) -> "VectorFunction":
        if self.vlen != 1:
            raise ValueError("InterpTable1D storing Vector data cannot be converted to Scalar Function.")
        return self(Arguments(1))

    def vf(self) -> "VectorFunction":
        return self(Arguments(1))
