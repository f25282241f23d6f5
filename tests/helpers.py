"""Shared workload construction for the parity tests (oracle side lives in oracle/, product in asset_asrl_amd/)."""
from __future__ import annotations

import numpy as np

from asset_asrl_amd import synth
from asset_asrl_amd.indexing import PhaseIndexer


class Workload:
    """A synthetic phase: solver vector X, multipliers L and the defect index tables."""

    def __init__(self, ode: str, mode: str, nseg: int, blocked: bool = False, seed: int = synth.SEED,
                 var_offset: int = 0, con_offset: int = 0, extra_vars: int = 0, sizes=None):
        xv, uv, pv = sizes if sizes is not None else synth.ODE_SIZES[ode]
        self.ode, self.mode, self.nseg, self.blocked = ode, mode, nseg, bool(blocked) and uv > 0
        self.cs = synth.MODE_CS[mode]
        self.traj = synth.make_traj(ode, mode, nseg, seed=seed, sizes=sizes)
        ix = PhaseIndexer(xv, uv, pv, 0)
        ix.set_dimensions(self.cs, nseg, self.blocked)
        ix.begin_indexing(var_offset, con_offset)
        self.indexer = ix
        self.vindex, self.cindex = ix.make_defect_Vindex_Cindex()
        self.IR, self.OR = ix.defect_sizes()
        self.n_primal = var_offset + ix.numPhaseVars + extra_vars
        self.n_equal = con_offset + ix.numPhaseEqCons
        rng = np.random.default_rng(seed + 7)
        self.X = rng.uniform(-1, 1, self.n_primal)
        self.X[var_offset:var_offset + ix.numPhaseVars] = ix.makeSolverInput(self.traj)
        self.L = synth.make_multipliers(self.n_equal, seed=seed + 1)
        self.NKKT = self.IR * (self.IR + 1) // 2 + self.OR * self.IR

    def oracle_nlp(self, ob, threads: int = 1, provider: int = 0, hessian_sparsity: bool = False):
        return ob.Nlp(ob.get_ode(self.ode, provider), ob.MODES[self.mode], self.blocked, self.vindex, self.cindex,
                      self.n_primal, self.n_equal, threads, hessian_sparsity=hessian_sparsity)


def csr_locations(rows, cols, outer, inner):
    """KKTLocations of slots (rows, cols) in an upper-triangular row-major CSR matrix (outer, inner): what
    NonLinearProgram::analyzeSparsity computes (NonLinearProgram.cpp:282-330) -- every slot is kept as (larger, smaller) index
    and filed in CSR row `smaller`; the order of the slots does not enter."""
    rows, cols = np.asarray(rows, dtype=np.int64), np.asarray(cols, dtype=np.int64)
    lo, hi = np.minimum(rows, cols), np.maximum(rows, cols)
    outer, inner = np.asarray(outer, dtype=np.int64), np.asarray(inner, dtype=np.int64)
    dim = outer.size - 1
    key_of = np.repeat(np.arange(dim, dtype=np.int64), np.diff(outer)) * (dim + 1) + inner        # sorted: rows ascending, columns sorted
    want = lo * (dim + 1) + hi
    pos = np.searchsorted(key_of, want)
    assert np.array_equal(key_of[pos], want), "a slot names an entry the matrix does not hold"
    return pos.astype(np.int32)


def rel_err(a, b, floor=1.0):
    a, b = np.asarray(a), np.asarray(b)
    return float(np.max(np.abs(a - b)) / max(floor, float(np.max(np.abs(b)))))


def make_vanderpol():
    """A user-defined ODE (not compiled into libasset_hip.so): forced Van der Pol oscillator with one control and one
    parameter, x0' = x1, x1' = mu (1 - x0^2) x1 - x0 + u exp(-t/10).  The oracle holds the same right-hand side as
    ``vanderpol`` (oracle/odes.h), differentiated independently by AD2."""
    from asset_asrl_amd import vf
    from asset_asrl_amd.ode import ODEArguments, ODEBase

    class VanDerPol(ODEBase):
        def __init__(self):
            a = ODEArguments(2, 1, 1)
            x0, x1 = a.XVec().tolist()
            u, mu, t = a.UVar(0), a.PVar(0), a.TVar()
            super().__init__(vf.stack([x1, mu * (1.0 - x0 * x0) * x1 - x0 + u * vf.exp(-0.1 * t)]), 2, 1, 1,
                             name="vanderpol")

    return VanDerPol()


def make_coupled(n: int = 12):
    """A wider user-defined ODE, (n, 3, 2): x_k' = -x_k/2 + sin(x_{k+1}) x_{k+5} u_{k mod 3} + p0 cos t + p1 x_k x_{k+7}
    (indices mod n).  n = 12: in LGL7 a segment has IR = 66 inputs, so its dense stage is the four-wave kernel -- with
    control rows and parameter columns, which the 32-state BASELINE ODE does not have; n = 16: the BlockConstant form
    (IR = 73) is wide too, with five parameter columns.  The oracle holds the same right-hand sides as ``coupled12`` /
    ``coupled16`` (oracle/odes.h)."""
    from asset_asrl_amd import vf
    from asset_asrl_amd.ode import ODEArguments, ODEBase

    class Coupled(ODEBase):
        def __init__(self):
            a = ODEArguments(n, 3, 2)
            x = a.XVec().tolist()
            t, p0, p1 = a.TVar(), a.PVar(0), a.PVar(1)
            u = [a.UVar(k) for k in range(3)]
            rhs = [-0.5 * x[k] + vf.sin(x[(k + 1) % n]) * x[(k + 5) % n] * u[k % 3] + p0 * vf.cos(t) + p1 * x[k] * x[(k + 7) % n]
                   for k in range(n)]
            super().__init__(vf.stack(rhs), n, 3, 2, name=f"coupled{n}")

    return Coupled()


def make_coupled12():
    return make_coupled(12)


def make_driven(n: int = 14):
    """A wide user-defined ODE with controls and NO parameters, (n, 3, 0): x_k' = -x_k/2 + sin(x_{k+1}) x_{k+5} u_{k mod 3}
    + 0.3 cos(t) x_{k+3} + 0.1 u_{(k+1) mod 3}^2 (indices mod n).  (14, 3, 0) in LGL7 and (20, 3, 0) in LGL5 / LGL7 are wide
    shapes (IR = 72 / 72 / 96) without segment parameters: the row-wise dense stage (csrc/defect_rows.h) with control rows.
    The oracle holds the same right-hand sides as ``driven14`` / ``driven20`` (oracle/odes.h)."""
    from asset_asrl_amd import vf
    from asset_asrl_amd.ode import ODEArguments, ODEBase

    class Driven(ODEBase):
        def __init__(self):
            a = ODEArguments(n, 3, 0)
            x = a.XVec().tolist()
            t = a.TVar()
            u = [a.UVar(k) for k in range(3)]
            rhs = [-0.5 * x[k] + vf.sin(x[(k + 1) % n]) * x[(k + 5) % n] * u[k % 3] + 0.3 * vf.cos(t) * x[(k + 3) % n]
                   + 0.1 * u[(k + 1) % 3] * u[(k + 1) % 3] for k in range(n)]
            super().__init__(vf.stack(rhs), n, 3, 0, name=f"driven{n}")

    return Driven()


SHAPES = [(1, 0, 0), (1, 1, 0), (2, 1, 0), (3, 0, 1), (4, 4, 0), (5, 3, 2), (6, 0, 0), (8, 3, 1), (10, 4, 0), (11, 4, 0)]


def make_shape(n: int, m: int, p: int):
    """One smooth right-hand side for any (states, controls, parameters) -- the oracle holds the members of SHAPES as
    ``shape_n_m_p`` (oracle/odes.h: shape_nmp):
      x_k' = -x_k/2 + sin(x_{k+1}) x_{k+2} [u_{k mod m}] + 0.3 cos(t) x_{k+3} [+ 0.1 u_{(k+1) mod m}^2] [+ p_0 x_k x_{k+1} + p_{p-1} cos t]"""
    from asset_asrl_amd import vf
    from asset_asrl_amd.ode import ODEArguments, ODEBase

    class Shape(ODEBase):
        def __init__(self):
            a = ODEArguments(n, m, p)
            x = a.XVec().tolist() if n > 1 else [a.XVar(0)]
            t = a.TVar()
            u = [a.UVar(k) for k in range(m)]
            par = [a.PVar(k) for k in range(p)]
            rhs = []
            for k in range(n):
                v = vf.sin(x[(k + 1) % n]) * x[(k + 2) % n]
                if m > 0:
                    v = v * u[k % m]
                v = v - 0.5 * x[k] + 0.3 * vf.cos(t) * x[(k + 3) % n]
                if m > 0:
                    v = v + 0.1 * u[(k + 1) % m] * u[(k + 1) % m]
                if p > 0:
                    v = v + par[0] * x[k] * x[(k + 1) % n] + par[p - 1] * vf.cos(t)
                rhs.append(v)
            super().__init__(vf.stack(rhs), n, m, p, name=f"shape_{n}_{m}_{p}")

    return Shape()


class FullProblem:
    """One Reentry LGL5 phase with everything a phase can hand the solver: the defect equality, a path equality at every
    state, the mesh-spacing equality of every segment, a pair-wise path inequality between neighbouring states and an
    integral objective over every segment.  Index tables only (numpy); the product and the oracle both build their
    programs from `functions`: (kind, tag, vindex, cindex) with kind 0 objective / 1 equality / 2 inequality."""

    def __init__(self, nseg: int = 13, seed: int = 77):
        self.w = w = Workload("reentry", "LGL5", nseg, seed=seed)
        ix = w.indexer
        xtu, xv, S, cs = ix.XtUVars(), ix.xv, ix.numStates, w.cs
        self.n_primal = w.n_primal
        rows = w.n_equal
        state = lambda k, v: k * xtu + v                                   # location of variable v of state k

        def take(n):
            nonlocal rows
            r = np.arange(rows, rows + n, dtype=np.int32)
            rows += n
            return r
        fns = [(1, "defect", w.vindex, w.cindex)]
        pv = np.array([[state(k, v) for v in (0, 1, 2, 5, 6, 7)] for k in range(S)], dtype=np.int32)
        fns.append((1, "pathcon", pv, take(2 * S).reshape(S, 2)))
        tv = np.array([[state(s * (cs - 1) + j, xv) for j in range(cs)] for s in range(nseg)], dtype=np.int32)
        fns.append((1, "meshspacing", tv, take(nseg * (cs - 2)).reshape(nseg, cs - 2)))
        self.n_equal = rows
        qv = np.array([[state(k + d, v) for d in (0, 1) for v in (3, 4)] for k in range(S - 1)], dtype=np.int32)
        fns.append((2, "pairprod", qv, np.arange(S - 1, dtype=np.int32).reshape(S - 1, 1)))
        self.n_inequal = S - 1
        ov = np.array([[state(s * (cs - 1) + j, v) for j in range(cs) for v in (2, 0, xv)] for s in range(nseg)], dtype=np.int32)
        fns.append((0, "integral", ov, None))
        self.functions = fns
        rng = np.random.default_rng(seed + 1)
        self.X = w.X
        self.LE = 10.0 * rng.uniform(-1, 1, self.n_equal)
        self.LI = rng.uniform(0.1, 2.0, self.n_inequal)
        self.obj_scale = 0.37
        nsolver = self.n_inequal + self.n_primal + self.n_inequal + self.n_equal + self.n_inequal
        self.solver_coeffs = np.concatenate([np.ones(self.n_inequal), rng.uniform(0.5, 1.5, nsolver - self.n_inequal)])

    def oracle_nlp(self, ob):
        n = ob.FullNlp(self.n_primal, self.n_equal, self.n_inequal)
        for kind, tag, v, c in self.functions:
            if tag == "defect":
                n.add(kind, ob.get_ode("reentry", 0), ob.MODES["LGL5"], False, v, c)
            elif tag == "pathcon":
                n.add(kind, ob.get_ode("pathcon", 0), ob.MODES["Function"], False, v, c)
            elif tag == "meshspacing":
                n.add_mesh_spacing(kind, 3, v, c)
            elif tag == "pairprod":
                n.add(kind, ob.get_ode("pairprod", 0), ob.MODES["Function"], False, v, c)
            else:
                n.add_integral(kind, ob.get_ode("integrand_quad2", 0), 3, 2, 0, v, c)
        n.analyze()
        n.set_solver_coeffs(self.solver_coeffs)
        return n

    def product_functions(self):
        """DSL definitions of the plain functions (the defect is the library ODE): tag -> (vf function, jit name)."""
        from asset_asrl_amd import vf
        from asset_asrl_amd.pathfuncs import LGLIntegral, LGLMeshSpacing
        a = vf.Arguments(6)
        x0, x1, x2, t, u0, u1 = a.tolist()
        b = vf.Arguments(4)
        g = vf.Arguments(2)
        return {"pathcon": (vf.stack([x0 * x0 + x1 * u0 - vf.sin(x2), u0 * u0 + u1 * u1 - 1.0 + t * x0 * vf.exp(-1.0 * x1)]), "pathcon"),
                "meshspacing": (LGLMeshSpacing(3), "lglmeshspacing3"),
                "pairprod": (vf.stack([b[0] * b[2] - b[1] * b[3] - 0.5]), "iq0_pairwisepath"),
                "integral": (LGLIntegral(g.coeff(1) * g.coeff(1) + g.coeff(0), 3, 2), "lglintegral3_quad2")}


def make_nested_orbit(flat: bool = False):
    """A user ODE written the way the reference's users compose functions, ``outer(inner(y))``: the inner function maps two polar
    states and a scaled third one to a position vector, the outer one is a point-mass pull with an oblateness term in that vector
    -- (4, 1, 1): states (rho, phi, z, w), control u, parameter p.  The composition leaves CUTS in the expression graph
    (asset_asrl_amd/vf/ir.py: Graph.cut) and the derivative builder applies the chain rule block-wise across them
    (vf/codegen.py; the reference: CommonFunctions/NestedFunction.h:140-270).  ``flat``: differentiated as one flattened expression."""
    from asset_asrl_amd import vf
    from asset_asrl_amd.ode import ODEArguments, ODEBase

    class NestedOrbit(ODEBase):
        def __init__(self):
            a = ODEArguments(4, 1, 1)
            rho, phi, z, w = a.XVec().tolist()
            t, u, p = a.TVar(), a.UVar(0), a.PVar(0)
            inner = vf.stack([rho * vf.cos(phi), rho * vf.sin(phi), z * (1.0 + 0.1 * p)])       # R^7 -> R^3
            R = vf.Arguments(3)
            r2 = R.squared_norm()
            r = vf.sqrt(r2)
            s = R[2] / r
            pull = R.normalized_power3() * (-1.0)
            obl = vf.stack([R[0] * (5.0 * s * s - 1.0), R[1] * (5.0 * s * s - 1.0), R[2] * (5.0 * s * s - 3.0)]) * (0.01 / (r2 * r2 * r))
            acc = (pull + obl)(inner)                                                             # the composition
            rhs = [w * vf.cos(phi) + acc[0] * 0.5, acc[1] / rho + u * 0.1, w * 0.3 + acc[2] * vf.cos(0.2 * t), acc[0] * acc[2] - 0.05 * w + u]
            super().__init__(vf.stack(rhs), 4, 1, 1, name="nested_orbit")

    ode = NestedOrbit()
    if flat:          # the same function differentiated as ONE flattened expression (the form of rounds 1-5): another device module
        from asset_asrl_amd.vf import codegen
        old = codegen.BLOCK_CHAIN_RULE
        codegen.BLOCK_CHAIN_RULE = False
        try:
            ode.derivatives()
        finally:
            codegen.BLOCK_CHAIN_RULE = old
    return ode


def make_switched():
    """A user ODE with conditionals (2, 1, 0): ``vf.ifelse`` on a joined test, on a test of the time, ``vf.abs`` and ``vf.sign`` --
    the reference's IfElseFunction / ConditionalStatement / SignFunction (CommonFunctions/Conditional.h:19-260).  The oracle holds the
    same right-hand side as ``switched`` (oracle/odes.h), differentiated independently by AD2."""
    from asset_asrl_amd import vf
    from asset_asrl_amd.ode import ODEArguments, ODEBase

    class Switched(ODEBase):
        def __init__(self):
            a = ODEArguments(2, 1, 0)
            x0, x1 = a.XVec().tolist()
            t, u = a.TVar(), a.UVar(0)
            spring = vf.ifelse((x0 > 0.25) & (x1 >= 0.0), 3.0 * x0 + 0.5 * x0 * x0, x0)
            f0 = x1 + vf.ifelse(t < 5.0, 0.1 * vf.sin(x0), 0.0)
            f1 = -1.0 * spring - 0.3 * vf.abs(x1) * x1 - 0.05 * vf.sign(x1) + u * vf.cos(t)
            super().__init__(vf.stack([f0, f1]), 2, 1, 0, name="switched")

    return Switched()


def tabulated_tables():
    """The three tables of the oracle's `tabulated` ODE (oracle/interp_table.h: tabulated_table), built here from the same
    arithmetic as ``vf.InterpTable1D`` objects through three of the reference's constructors (InterpTable1D.h:409-424)."""
    from asset_asrl_amd import vf
    i = np.arange(25, dtype=float)
    h = -1.5 + 0.1 * i + 0.004 * i * i
    density = vf.InterpTable1D(h, 1.2 / (1.0 + 0.5 * (h + 1.0) * (h + 1.0)), kind="cubic")              # (ts, vector)
    i = np.arange(21, dtype=float)
    thrust = vf.InterpTable1D([np.array([1.0 + 0.05 * k - 0.004 * k * k, -0.25 + 0.537 * k]) for k in i], -1, "linear")   # value-time vectors
    s = -2.0 + 0.25 * np.arange(17, dtype=float)
    wind = vf.InterpTable1D(s, np.column_stack([0.3 * s * s - 0.1 * s, 1.0 / (2.5 + s)]), axis=0, kind="cubic")   # (ts, matrix, axis)
    return density, thrust, wind


def make_tabulated():
    """A user ODE on tabulated data (2, 1, 0): a cubic table with uneven abscissae over a state, a linear one over the time, a
    two-valued cubic one over the other state -- the reference's InterpTable1D in an ODE (CommonFunctions/InterpTable1D.h:9-401).
    The oracle holds the same right-hand side as ``tabulated`` (oracle/odes.h) over its own restatement of the table
    (oracle/interp_table.h), differentiated by AD2 with the reference's hand-written dv/dt and d2v/dt2."""
    from asset_asrl_amd import vf
    from asset_asrl_amd.ode import ODEArguments, ODEBase
    density, thrust, wind = tabulated_tables()

    class Tabulated(ODEBase):
        def __init__(self):
            a = ODEArguments(2, 1, 0)
            x0, x1 = a.XVec().tolist()
            t, u = a.TVar(), a.UVar(0)
            w = wind(x1)
            f0 = x1 + 0.1 * w[0] * w[1]
            f1 = u * thrust(t) - 0.05 * density(x0) * x1 * x1 - 1.0
            super().__init__(vf.stack([f0, f1]), 2, 1, 0, name="tabulated")

    return Tabulated()
