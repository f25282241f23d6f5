"""Plain per-application functions of a phase on the device.

``FunctionEvaluator`` batches any vector function written in the expression DSL over index tables, with the same
outputs as the defect evaluator (FX / AGX blocks, KKT block = Jacobian + lower-triangle adjoint Hessian in the
reference's slot order) -- what the reference does for every function handed to ``addEqualCon / addInequalCon /
add*Objective`` through ``ConstraintFunction`` / ``ObjectiveFunction``
(/root/reference/src/VectorFunctions/ComputableBase.h:246-335, DenseFunctionBase.h:1145-1391).

The two mesh relations the reference adds to every LGL phase are provided as DSL functions:
``LGLMeshSpacing(cs)`` and ``SingleMeshSpacing(s)`` (OptimalControl/MeshSpacingConstraints.h:8-98, 101-193), and so is
the segment quadrature behind integral objectives, ``LGLIntegral(integrand, cs, xv, pv)`` (LGLIntegrals.h:9-73), and
the control-spline continuity relation ``LGLControlSpline(cs, usize)`` (LGLControlSplines.h:64-315)."""
from __future__ import annotations

from . import _lib, jit, synth, vf
from .evaluator import DefectEvaluator


def LGLMeshSpacing(cs: int) -> vf.VectorFunction:
    """Inputs: the cs node times of a segment; outputs i = 0..cs-3: tc[i+1] - (t_{i+1} - t_0) / (t_{cs-1} - t_0)
    (MeshSpacingConstraints.h:118-126)."""
    if cs < 3:
        raise ValueError("mesh spacing relations exist for schemes with interior cardinal nodes (LGL5, LGL7)")
    t = vf.Arguments(cs)
    h = t.coeff(cs - 1) - t.coeff(0)
    tc = synth._TC[cs]
    return vf.stack([tc[i + 1] - (t.coeff(1 + i) - t.coeff(0)) / h for i in range(cs - 2)])


def SingleMeshSpacing(cardinal_spacing=None, scale: float = 1.0) -> vf.VectorFunction:
    """Inputs (t_0, t_j, t_f); output scale * (s * (t_f - t_0) - (t_j - t_0))  (MeshSpacingConstraints.h:33-41).
    ``cardinal_spacing=None``: s is constant 0 of the function application (vf.ApplConst) -- one device function for
    all the SingleMeshSpacing objects of a phase, which differ in nothing but s."""
    t = vf.Arguments(3)
    s = vf.ApplConst(3, 0) if cardinal_spacing is None else cardinal_spacing
    return ((t.coeff(2) - t.coeff(0)) * s - (t.coeff(1) - t.coeff(0))) * scale


# Reduced_Integral_Weights of the schemes (LGLCoeffs.h:42, 135, 360-364; expressions kept as the reference writes them)
_REDUCED_INTEGRAL_WEIGHTS = {
    2: (0.5, 0.5),
    3: (1.0 / 6.0, 2.0 / 3.0, 1.0 / 6.0),
    4: (-5.12701665379258 / 4.0 + 10.2540333075852 / 3.0 - 6.12701665379258 / 2.0 + 1.0,
        10.9353308042859 / 4.0 - 18.9665045333251 / 3.0 + 8.03117372903925 / 2.0,
        -10.9353308042859 / 4.0 + 13.8394878795326 / 3.0 - 2.90415707524666 / 2.0,
        5.12701665379258 / 4.0 - 5.12701665379258 / 3.0 + 1.0 / 2.0),
}


def LGLIntegral(integrand: vf.VectorFunction, cs: int, xv: int, pv: int = 0) -> vf.VectorFunction:
    """Quadrature of ``integrand`` over one segment: inputs ``[x_0(xv), t_0, ..., x_{cs-1}(xv), t_{cs-1}, p(pv)]``,
    output ``(t_{cs-1} - t_0) * sum_i w_i integrand([x_i, p])`` (LGLIntegrals.h:9-52).  ``integrand`` takes xv + pv
    inputs."""
    if cs not in _REDUCED_INTEGRAL_WEIGHTS:
        raise ValueError("cs must be 2, 3 or 4")
    if integrand.IRows() != xv + pv:
        raise ValueError(f"integrand takes {integrand.IRows()} inputs, expected xv + pv = {xv + pv}")
    xtv = xv + 1
    a = vf.Arguments(cs * xtv + pv)
    h = a.coeff((cs - 1) * xtv + xv) - a.coeff(xv)
    p = a.tail(pv) if pv else None
    terms = []
    for i, w in enumerate(_REDUCED_INTEGRAL_WEIGHTS[cs]):
        xi = a.segment(i * xtv, xv)
        arg = vf.stack([xi, p]) if pv else xi
        terms.append(integrand.eval(arg) * w)
    return vf.sum(*terms) * h


# UOneSpline_Weights / UZeroSpline_Weights (LGLCoeffs.h:155-158, 372-388; expressions kept as the reference writes them)
_UZERO = {
    3: ((-3.0, 4.0, -1.0),),
    4: ((-6.12701665379258, 8.03117372903925, -2.90415707524666, 1.0),
        (10.2540333075852 * 2.0, -18.9665045333251 * 2.0, +13.8394878795326 * 2.0, -5.12701665379258 * 2.0)),
}
_UONE = {
    3: ((1.0, -4.0, 3.0),),
    4: ((-5.12701665379258 * 3.0 + 10.2540333075852 * 2.0 - 6.12701665379258,
         10.9353308042859 * 3.0 - 18.9665045333251 * 2.0 + 8.03117372903925,
         -10.9353308042859 * 3.0 + 13.8394878795326 * 2.0 - 2.90415707524666,
         5.12701665379258 * 3.0 - 5.12701665379258 * 2.0 + 1.0),
        (-5.12701665379258 * 6.0 + 10.2540333075852 * 2.0, 10.9353308042859 * 6.0 - 18.9665045333251 * 2.0,
         -10.9353308042859 * 6.0 + 13.8394878795326 * 2.0, 5.12701665379258 * 6.0 - 5.12701665379258 * 2.0)),
}


def LGLControlSpline(cs: int, usize: int, order: int | None = None) -> vf.VectorFunction:
    """Continuity of the control polynomial's derivatives across two adjacent segments: inputs are the
    2*cs-1 blocks ``[t, u(usize)]`` of their nodes, outputs ``j = 0..order-1`` (derivative j+1), ``usize`` rows each:
    ``sum_i UOne[j][i] u_i / h0^(j+1) - UZero[j][i] u_{i+cs-1} / h1^(j+1)``  (LGLControlSplines.h:64-108)."""
    if cs not in _UONE:
        raise ValueError("control splines exist for LGL5 (cs=3) and LGL7 (cs=4)")
    order = cs - 2 if order is None else order
    if not 1 <= order <= len(_UONE[cs]):
        raise ValueError("order out of range for this scheme")
    tu = usize + 1
    a = vf.Arguments((2 * cs - 1) * tu)
    t = lambda i: a.coeff(i * tu)
    u = lambda i: a.segment(i * tu + 1, usize)
    h0, h1 = t(cs - 1) - t(0), t(2 * cs - 2) - t(cs - 1)
    outs = []
    for j in range(order):
        acc = None
        for i in range(cs):
            term = u(i) * _UONE[cs][j][i] / h0 ** (j + 1) - u(i + cs - 1) * _UZERO[cs][j][i] / h1 ** (j + 1)
            acc = term if acc is None else acc + term
        outs.append(acc)
    return vf.stack(outs)


class FunctionBundle:
    """Several FunctionEvaluators of one problem evaluated in ONE launch (include/asset_hip.h: asset_hip_bundle_*).  The
    members keep their own index tables, multiplier vectors and outputs; `eval_device` takes one tensor per member."""

    def __init__(self, evaluators):
        import ctypes as C
        self.members = list(evaluators)
        self.name = jit.ensure_bundle([e.device_name for e in self.members])
        hs = (C.c_void_p * len(self.members))(*[e._h for e in self.members])
        self._b = C.c_void_p()
        _lib.check(_lib.lib().asset_hip_bundle_create(self.name.encode(), hs, len(self.members), C.byref(self._b)),
                   "asset_hip_bundle_create")

    def bind_device(self, what: int, X, Ls, fxs, agxs, kkts, stream=None):
        """A zero-argument callable that enqueues the evaluation of every member (arguments converted once)."""
        import ctypes as C
        n = len(self.members)
        VP = C.c_void_p * n

        def arr(ts):
            return VP(*[None if t is None else (t if isinstance(t, int) else t.data_ptr()) for t in ts])
        args = (self._b, what, DefectEvaluator._p(X), arr(Ls), arr(fxs), arr(agxs), arr(kkts),
                None if stream is None else C.c_void_p(stream if isinstance(stream, int) else stream.cuda_stream))
        fn = _lib.lib().asset_hip_bundle_eval_device
        keep = (X, Ls, fxs, agxs, kkts)

        def call(_fn=fn, _args=args, _keep=keep):
            rc = _fn(*_args)
            if rc:
                _lib.check(rc, "asset_hip_bundle_eval_device")
        return call

    def eval_device(self, what: int, X, Ls, fxs, agxs, kkts, stream=None):
        self.bind_device(what, X, Ls, fxs, agxs, kkts, stream)()

    def close(self):
        if getattr(self, "_b", None):
            _lib.lib().asset_hip_bundle_destroy(self._b)
            self._b = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class FunctionEvaluator(DefectEvaluator):
    """``func`` applied to ``X[vindex[V]]`` for every application V, multipliers ``L[cindex[V]]``; device code is
    generated and compiled on first use (jit.ensure_function)."""

    def __init__(self, func: vf.VectorFunction, name: str, vindex, cindex, n_primal: int, n_equal: int, device: int = 0,
                 appl_consts=None):
        self.func = func
        dev_name = jit.ensure_function(func, name)
        self.device_name = dev_name
        super().__init__(dev_name, _lib.FUNCTION, False, vindex, cindex, n_primal, n_equal, device)
        if appl_consts is not None:
            self.set_appl_consts(appl_consts)
