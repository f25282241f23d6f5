"""Plain per-application functions of a phase on the device.

``FunctionEvaluator`` batches any vector function written in the expression DSL over index tables, with the same
outputs as the defect evaluator (FX / AGX blocks, KKT block = Jacobian + lower-triangle adjoint Hessian in the
reference's slot order) -- what the reference does for every function handed to ``addEqualCon / addInequalCon /
add*Objective`` through ``ConstraintFunction`` / ``ObjectiveFunction``
(/root/reference/src/VectorFunctions/ComputableBase.h:246-335, DenseFunctionBase.h:1145-1391).

The two mesh relations the reference adds to every LGL phase are provided as DSL functions:
``LGLMeshSpacing(cs)`` and ``SingleMeshSpacing(s)`` (OptimalControl/MeshSpacingConstraints.h:8-98, 101-193)."""
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


def SingleMeshSpacing(cardinal_spacing: float, scale: float = 1.0) -> vf.VectorFunction:
    """Inputs (t_0, t_j, t_f); output scale * (s * (t_f - t_0) - (t_j - t_0))  (MeshSpacingConstraints.h:33-41)."""
    t = vf.Arguments(3)
    return ((t.coeff(2) - t.coeff(0)) * cardinal_spacing - (t.coeff(1) - t.coeff(0))) * scale


class FunctionEvaluator(DefectEvaluator):
    """``func`` applied to ``X[vindex[V]]`` for every application V, multipliers ``L[cindex[V]]``; device code is
    generated and compiled on first use (jit.ensure_function)."""

    def __init__(self, func: vf.VectorFunction, name: str, vindex, cindex, n_primal: int, n_equal: int, device: int = 0):
        self.func = func
        dev_name = jit.ensure_function(func, name)
        super().__init__(dev_name, _lib.FUNCTION, False, vindex, cindex, n_primal, n_equal, device)
