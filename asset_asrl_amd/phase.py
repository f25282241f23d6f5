"""``Phase`` -- the slice of the reference's ODEPhase API that reaches the accelerated path.

Kept names / argument meaning (host orchestration only; nothing here is timed):
  ``ode.phase(mode, traj, nsegs)``, ``setTraj``, ``setControlMode``, ``switchTranscriptionMode``, ``transcribe``,
  ``get_defect`` (-> object with ``IRows/ORows/compute/jacobian/adjointgradient/adjointhessian/computeall``),
  ``test_threads(i, j, n)`` and ``returnTraj``
(/root/reference/src/OptimalControl/ODEPhaseBase.cpp:536-670,1364-1441; ODEPhase.h:165-341,687-717;
 /root/reference/src/VectorFunctions/DenseFunctionBase.h:1546-1631).
Beside the defects a phase hands the solver the other per-segment functions the reference registers itself when it
transcribes (ODEPhaseBase.cpp:962-1061): the mesh-spacing equalities (``LGLMeshSpacing`` over every defect,
``SingleMeshSpacing`` at every inner nodal state) and, in the spline control modes, the control-spline continuity
equalities (``LGLControlSpline`` over every pair of adjacent defects); user equalities / inequalities over phase regions
(``addEqualCon`` / ``addInequalCon``) and integral objectives (``addIntegralObjective``, ODEPhaseBase.cpp:743-889:
``LGLIntegral`` over every defect).  Each is one device evaluator batched over its applications.  PSIOPT, bounds, links
and the mesh refinement loop stay with the host solver.
"""
from __future__ import annotations

import time
from typing import Sequence

import numpy as np

from . import _lib, jit, synth
from .evaluator import (CON, CON_ADJGRAD, JAC, JAC_ADJGRAD, JAC_ADJGRAD_HESS, DefectEvaluator, unpack_kkt_block)
from .indexing import PhaseIndexer

_MODES = ("LGL3", "LGL5", "LGL7", "Trapezoidal")
_CONTROL_MODES = ("HighestOrderSpline", "FirstOrderSpline", "NoSpline", "BlockConstant")


class DefectFunction:
    """The phase's defect as a VectorFunction of one segment: z[IR] -> d[OR] (``phase.get_defect()``)."""

    def __init__(self, ode_name: str, mode: str, blocked: bool, device: int = 0):
        self._ode, self._mode, self._blocked, self._device = ode_name, mode, blocked, device
        xv, uv, pv = _lib.ode_sizes(ode_name)
        ix = PhaseIndexer(xv, uv, pv, 0)
        ix.set_dimensions(synth.MODE_CS[mode], 1, blocked)
        self._ir, self._or = ix.defect_sizes()
        self._ev = None

    def IRows(self):
        return self._ir

    def ORows(self):
        return self._or

    def name(self):
        return f"{self._mode}Defects<{self._ode}>"

    def _eval(self, what, x, l=None):
        x = np.asarray(x, dtype=float).ravel()
        if x.size != self._ir:
            raise ValueError(f"Input vector has incorrect size: {x.size} != IRows {self._ir}")
        if l is not None:
            l = np.asarray(l, dtype=float).ravel()
            if l.size != self._or:
                raise ValueError(f"Multiplier vector has incorrect size: {l.size} != ORows {self._or}")
        if self._ev is None:
            V = np.arange(self._ir, dtype=np.int32)[None, :]
            Cx = np.arange(self._or, dtype=np.int32)[None, :]
            self._ev = DefectEvaluator(self._ode, self._mode, self._blocked, V, Cx, self._ir, self._or, self._device)
        fx, agx, kkt = self._ev.eval(what, x, l)
        H = J = None
        if kkt is not None:
            H, J = unpack_kkt_block(kkt[0], self._ir, self._or)
        return fx[0], J, (None if agx is None else agx[0]), H

    def compute(self, x):
        return self._eval(CON, x)[0]

    __call__ = compute

    def jacobian(self, x):
        return self._eval(JAC, x)[1]

    def adjointgradient(self, x, l):
        return self._eval(CON_ADJGRAD, x, l)[2]

    def adjointhessian(self, x, l):
        return self._eval(JAC_ADJGRAD_HESS, x, l)[3]

    def computeall(self, x, l):
        """(fx, jx, gx, hx) -- the reference's ``computeall(x, l)`` tuple."""
        return self._eval(JAC_ADJGRAD_HESS, x, l)


class Phase:
    def __init__(self, ode, mode: str = "LGL3", device: int = 0):
        self.ode = ode
        self.device = device
        self.TranscriptionMode = None
        self.ControlMode = "HighestOrderSpline"
        self.switchTranscriptionMode(mode)
        self.ActiveTraj = None
        self.numDefects = 0
        self.DefBinSpacing, self.DefsPerBin = np.array([0.0, 1.0]), np.array([1], dtype=int)
        self._ev = None
        self._ev_kept = None      # (key, evaluator): survives re-meshing, see transcribe()
        self._indexer = None
        self.AutoScaling = False
        # adaptive mesh (ODEPhaseBase.h:96-118): the loop around the de Boor estimate -- checkMesh / updateMesh below
        self.AdaptiveMesh, self.MeshConverged, self.MeshIters = False, False, []
        self.MeshTol, self.MaxMeshIters, self.MaxSegments, self.MinSegments, self.NumExtraSegs = 1.0e-6, 10, 10000, 4, 4
        self.MeshRedFactor, self.MeshIncFactor, self.MeshErrFactor, self.MeshErrorCriteria = 0.5, 5.0, 10.0, "max"
        self.XtUPUnits = np.ones(ode.XtUPVars())
        self._ode_scaled = None
        self._eq_funcs = []       # (region, func, xtuv, opv, spv) -- addEqualCon
        self._iq_funcs = []       # addInequalCon
        self._integral_objs = []  # (integrand, xtuv, opv, spv) -- addIntegralObjective
        self._integral_params = []  # (integrand, xtuv, opv, spv, accumulating static parameter) -- addIntegralParamFunction
        self.ActiveStaticParams = np.zeros(0)   # setStaticParams
        self._eq_evs = []
        self._iq_evs = []
        self._obj_evs = []
        self._ipf_evs = {}        # k -> {"acc": evaluator, "int": evaluator}: addIntegralParamFunction
        self._auto_evs = {}       # "mesh_spacing", "nodal_spacing", "control_spline": evaluators the phase adds itself
        self.EnableHessianSparsity = False   # Trapezoidal only (ODEPhase.h:43, TrapezoidalDefects.h:39): see hessian_mask()
        self.EnableMeshSpacing = True      # (the reference always adds them; the switch is for callers that only want the defects)

    # ---- configuration ---------------------------------------------------------------------
    def switchTranscriptionMode(self, mode: str):
        if mode not in _MODES:
            raise ValueError("Invalid Transcription Method")   # ODEPhase.h:209
        self.TranscriptionMode = mode
        self._ev = None

    setTranscriptionMode = switchTranscriptionMode

    def setControlMode(self, mode: str):
        if mode not in _CONTROL_MODES:
            raise ValueError(f"Unrecognized control mode: {mode}")
        self.ControlMode = mode
        self._ev = None

    # ---- AutoScaling: the dynamics in scaled units (ODEPhase.h:87-109 setUnits, :293-326 transcribe_dynamics) --
    def setUnits(self, XtUPUnits):
        """Units of [x, t, u, p]; the scaled ODE is ``IOScaled(ode, units, units_t / units_x)``."""
        u = np.asarray(XtUPUnits, dtype=float).ravel()
        if u.size != self.ode.XtUPVars():
            raise ValueError("Incorrect size for input units vector")
        from .ode import ODEBase
        from .vf import IOScaled
        xv = self.ode.XVars()
        self.XtUPUnits = u.copy()
        func = IOScaled(self.ode.vf(), u, u[xv] / u[:xv])
        self._ode_scaled = ODEBase(func, xv, self.ode.UVars(), self.ode.PVars(), name=self.ode.ode_name + "_scaled")
        self._ev = None

    def setAutoScaling(self, flag: bool = True):
        self.AutoScaling = bool(flag)
        if self.AutoScaling and self._ode_scaled is None:
            self.setUnits(self.XtUPUnits)
        self._ev = None

    def _active_ode(self):
        return self._ode_scaled if self.AutoScaling else self.ode

    def _blocked(self) -> bool:
        return self.ControlMode == "BlockConstant" and self.ode.UVars() > 0

    # ---- trajectory -------------------------------------------------------------------------
    def setTraj(self, traj: Sequence, nsegs, DefsPerBin=None):
        """Resample `traj` (rows [x,t,u,p]) onto the phase's mesh with the scheme's cardinal spacing (linear interpolation
        in time, as the reference's default LerpIG does).  ``setTraj(traj, nsegs)``: `nsegs` equal segments;
        ``setTraj(traj, DefBinSpacing, DefsPerBin)``: bins with non-dimensional edges DefBinSpacing[0..nbins] (0 ... 1,
        increasing) holding DefsPerBin[i] equal segments each (ODEPhaseBase.cpp:523-620)."""
        T = np.asarray(traj, dtype=float)
        if T.ndim != 2 or T.shape[1] != self.ode.XtUPVars():
            raise ValueError(f"Input trajectory states must have {self.ode.XtUPVars()} columns")
        if not np.all(np.isfinite(T)):
            raise ValueError("NaN or Inf detected in input trajectory")   # ODEPhaseBase.cpp:560-564
        if DefsPerBin is None:
            if int(nsegs) < 1:
                raise ValueError("Number of segments must be positive")
            DBS, DPB = np.array([0.0, 1.0]), np.array([int(nsegs)], dtype=int)
        else:
            DBS, DPB = np.asarray(nsegs, dtype=float).ravel(), np.asarray(DefsPerBin, dtype=int).ravel()
        self._check_bins(DBS, DPB)
        tcol = self.ode.TVar()
        t = T[:, tcol]
        if np.any(np.diff(t) <= 0) and np.any(np.diff(t) >= 0):
            raise ValueError("Trajectory time must be strictly monotonic")
        if t[-1] == t[0]:
            raise ValueError("Trajectory spans no time")
        nodes = self._mesh_times(DBS, DPB, t[0], t[-1])
        order = np.argsort(t)
        out = np.column_stack([np.interp(nodes, t[order], T[order, c]) for c in range(T.shape[1])])
        out[:, tcol] = nodes
        self.ActiveTraj = out
        self.DefBinSpacing, self.DefsPerBin = DBS.copy(), DPB.copy()
        self.numDefects = int(DPB.sum())
        self._ev = None

    @staticmethod
    def _check_bins(DBS, DPB):
        if DBS.size - 1 != DPB.size:                                      # ODEPhaseBase.cpp:664-669
            raise ValueError(f"Size of Defect Bin Spacing({DBS.size}) not consistent with size of Defects Per Bin({DPB.size})")
        if DPB.size < 1 or np.any(DPB < 1) or np.any(np.diff(DBS) <= 0) or not np.all(np.isfinite(DBS)):
            raise ValueError("Defect bins must be increasing and hold at least one defect each")

    def _mesh_times(self, DBS, DPB, t0, tf):
        """Times of the K * numDefects + 1 states of the mesh: bin i spans [DBS[i], DBS[i+1]] of [t0, tf], its DPB[i]
        segments are equal, the states of a segment sit at the scheme's cardinal spacing."""
        cs = synth.MODE_CS[self.TranscriptionMode]
        tc, K = _lib.lgl_table(cs, "tc"), cs - 1
        edges = np.concatenate([np.linspace(DBS[i], DBS[i + 1], DPB[i] + 1)[(1 if i else 0):] for i in range(DPB.size)])
        edges = t0 + (edges - DBS[0]) / (DBS[-1] - DBS[0]) * (tf - t0)
        nodes = np.empty(K * (edges.size - 1) + 1)
        for j in range(K):
            nodes[j:-1:K] = edges[:-1] + tc[j] * (edges[1:] - edges[:-1])
        nodes[-1] = tf
        return nodes

    def refineTrajManual(self, DefBinSpacing, DefsPerBin):
        """Re-distribute the active trajectory on new bins (ODEPhaseBase.cpp:662-677; the adaptive mesh loop's re-meshing step,
        :1443-1542 -- the interpolation is linear here, the reference's is the transcription's own polynomial)."""
        if self.ActiveTraj is None:
            raise RuntimeError("No trajectory set: call setTraj first")
        self.setTraj(self.ActiveTraj, DefBinSpacing, DefsPerBin)
        return self.returnTraj()

    def _nodal_spacing(self):
        """Non-dimensional times of the nodal states (segment boundaries), from the phase's own bins -- whatever the active
        trajectory holds (transcribe_axis_funcs, ODEPhaseBase.cpp:962-970: linspace inside every bin)."""
        DBS, DPB = self.DefBinSpacing, self.DefsPerBin
        c = np.concatenate([np.linspace(DBS[i], DBS[i + 1], DPB[i] + 1)[(1 if i else 0):] for i in range(DPB.size)])
        return (c - DBS[0]) / (DBS[-1] - DBS[0])

    def returnTraj(self):
        return [row.copy() for row in self.ActiveTraj]

    # ---- user constraints over phase regions (ODEPhaseBase.h addEqualCon / addInequalCon) -----------------
    def _add_func(self, store, region, func, xtuv, opv, spv):
        if region not in PhaseIndexer.REGIONS:
            raise ValueError(f"unknown phase region {region!r}")
        xtuv, opv, spv = [int(v) for v in xtuv], [int(v) for v in opv], [int(v) for v in spv]
        states = {"FrontandBack": 2, "BackandFront": 2, "PairWisePath": 2, "FrontNodalBackPath": 3,
                  "Params": 0, "ODEParams": 0, "StaticParams": 0}.get(region, 1)
        need = states * len(xtuv) + len(opv) + len(spv)
        if func.IRows() != need:
            raise ValueError(f"function takes {func.IRows()} inputs, the region and variable lists supply {need}")
        store.append((region, func, xtuv, opv, spv))
        self._ev = None
        return len(store) - 1

    def addEqualCon(self, region: str, func, XtUVars=(), OPVars=(), SPVars=()) -> int:
        """``func(X[region variables]) = 0`` at every application of the region (``"Path"``: every state of the
        mesh, ``"Front"``, ``"Back"``, ``"PairWisePath"`` ...; PhaseIndexer.make_Vindex_Cindex).  The function is a
        vf expression; it is differentiated symbolically, compiled for the device on first use and evaluated for all
        applications at once through the same handle API as the defects (transcription id 0).  Returns its index."""
        return self._add_func(self._eq_funcs, region, func, XtUVars, OPVars, SPVars)

    def addInequalCon(self, region: str, func, XtUVars=(), OPVars=(), SPVars=()) -> int:
        """``func(...) <= 0`` over a region; evaluated like an equality, its rows are numbered in the phase's
        inequality space (PhaseIndexer.cpp:78-92)."""
        return self._add_func(self._iq_funcs, region, func, XtUVars, OPVars, SPVars)

    def addIntegralObjective(self, integrand, XtUVars=(), OPVars=(), SPVars=(), output_scale: float = 1.0) -> int:
        """Minimise ``int integrand(x, t, u, p) dt`` over the phase (ODEPhaseBase.cpp:743-889): the segment quadrature
        ``LGLIntegral`` (LGLIntegrals.h:9-52; Trapezoidal phases use the two-node rule) of ``integrand`` over every defect,
        summed.  ``integrand`` takes the chosen variables of one state and the chosen parameters; it is evaluated, with
        its gradient and Hessian, on the device like any other function (``objective_evaluators``)."""
        xtuv, opv, spv = [int(v) for v in XtUVars], [int(v) for v in OPVars], [int(v) for v in SPVars]
        if integrand.ORows() != 1 or integrand.IRows() != len(xtuv) + len(opv) + len(spv):
            raise ValueError("an integrand has one output and takes the listed state variables and parameters")
        # (the time variable may be among the integrand's inputs: the quadrature appends it again for the node times,
        #  xtrap.head(xp) = XtUVars; xtrap[xp] = TVar, ODEPhaseBase.cpp:786-790 -- the Vindex row then names it twice and the
        #  duplicate-location Hessian entries are summed by the assembly like any other shared location)
        # (output_scale: the reference's OutputScales of the integrand -- with AutoScaling the objective handed to the solver is
        #  output_scale * integrand in the scaled variables, ODEPhaseBase.cpp:796-803; without AutoScaling it is not applied)
        self._integral_objs.append((integrand, xtuv, opv, spv, float(output_scale)))
        self._ev = None
        return len(self._integral_objs) - 1

    def hessian_mask(self):
        """``HessianElemIsNonZero`` of the phase's defects: None (every entry claims a KKT slot) unless the phase is
        Trapezoidal with ``EnableHessianSparsity`` set (ODEPhase.h:43; TrapezoidalDefects.h:75-141) -- then the bool [IR, IR]
        mask for ``indexing.kkt_slot_locations(..., hess_mask=)``, whose dropped slots the device assembly skips."""
        if not self.EnableHessianSparsity or self.TranscriptionMode != "Trapezoidal":
            return None
        from .indexing import trapezoidal_hessian_mask
        return trapezoidal_hessian_mask(self.ode.XVars(), self.ode.UVars(), self.ode.PVars(), self._blocked())

    def setStaticParams(self, params, units=None):
        """Static parameters of the phase: solver variables behind the trajectory and the ODE parameters that user functions may
        name in their ``SPVars`` (ODEPhaseBase.h setStaticParams; PhaseIndexer.cpp: StaticParamLoc).  ``units``: their units under
        AutoScaling (the reference's ``SPUnits``; ones when omitted)."""
        self.ActiveStaticParams = np.asarray(params, dtype=float).ravel().copy()
        self.SPUnits = np.ones(self.ActiveStaticParams.size) if units is None else np.asarray(units, dtype=float).ravel().copy()
        if self.SPUnits.size != self.ActiveStaticParams.size or np.any(self.SPUnits <= 0):
            raise ValueError("one positive unit per static parameter")
        self._ev = None

    def _input_scales(self, xtuv, opv, spv):
        """Units of a user function's inputs, in its argument order (the reference's get_input_scale: the chosen state variables,
        the chosen ODE parameters, the chosen static parameters)."""
        xtu = self.ode.XtUVars()
        sp = getattr(self, "SPUnits", np.ones(len(self.ActiveStaticParams)))
        return [self.XtUPUnits[v] for v in xtuv] + [self.XtUPUnits[xtu + v] for v in opv] + [sp[v] for v in spv]

    def _scaled_integrand(self, integrand, xtuv, opv, spv, output_scale):
        """An integrand as the solver's scaled variables see it (ODEPhaseBase.cpp:796-803, 848-855): IOScaled over the input units with
        the integrand's output scale.  Without AutoScaling the integrand itself."""
        if not self.AutoScaling:
            return integrand
        from .vf import IOScaled
        return IOScaled(integrand, self._input_scales(xtuv, opv, spv), [output_scale])

    def addIntegralParamFunction(self, integrand, XtUVars=(), OPVars=(), SPVars=(), accum_param: int = 0, scale: float = 1.0,
                                 output_scale: float = 1.0) -> int:
        """``int integrand(x, t, u, p) dt - scale * StaticParams[accum_param] = 0`` (ODEPhaseBase.cpp:835-889 with
        PhaseIndexer::addAccumulation, PhaseIndexer.cpp:41-76): the static parameter is made to equal the integral.  TWO equality
        functions share ONE constraint row -- the linear accumulation ``-scale * p`` over the Params region, and the segment
        quadrature of the integrand (the one ``addIntegralObjective`` uses) over every defect, all of whose applications carry
        the accumulation's row: their values, multiplier and Jacobian entries add up in the solver's row.  Returns the index of
        the pair among the phase's integral parameter functions; the two evaluators are
        ``integral_param_evaluators[index] = (accumulation, quadrature)``.  With AutoScaling the solver's variables are in scaled units:
        the integrand is wrapped in IOScaled over its input units with ``output_scale`` and the accumulation becomes
        ``-scale * AccScale * p_scaled``, ``AccScale = SPUnits[accum_param] * output_scale / XtUPUnits[t]`` (ODEPhaseBase.cpp:846-861) --
        the row is then ``(output_scale / t_unit) * (int integrand dt - scale * p)`` of the unscaled problem."""
        xtuv, opv, spv = [int(v) for v in XtUVars], [int(v) for v in OPVars], [int(v) for v in SPVars]
        if integrand.ORows() != 1 or integrand.IRows() != len(xtuv) + len(opv) + len(spv):
            raise ValueError("an integrand has one output and takes the listed state variables and parameters")
        if not 0 <= int(accum_param) < len(self.ActiveStaticParams):
            raise ValueError("accum_param names no static parameter of the phase (setStaticParams first)")
        self._integral_params.append((integrand, xtuv, opv, spv, int(accum_param), float(scale), float(output_scale)))
        self._ev = None
        return len(self._integral_params) - 1

    def removeEqualCon(self, index: int):
        del self._eq_funcs[index]
        self._ev = None

    def removeInequalCon(self, index: int):
        del self._iq_funcs[index]
        self._ev = None

    def _scaled_func(self, region, func, xtuv, opv):
        """With AutoScaling the solver's variables are in scaled units: compose the function with the input units
        (the reference wraps user functions in IOScaled the same way; outputs keep their scale)."""
        if not self.AutoScaling:
            return func
        from .vf import IOScaled
        xtu = self.ode.XtUVars()
        ux = [self.XtUPUnits[v] for v in xtuv]
        states = (func.IRows() - len(opv)) // max(1, len(xtuv)) if xtuv else 0
        units = ux * states + [self.XtUPUnits[xtu + v] for v in opv]
        units += [1.0] * (func.IRows() - len(units))
        return IOScaled(func, units, np.ones(func.ORows()))

    def _function_tables(self, ix, iq_offset: int = 0):
        """Host-only: every function the phase hands the solver beside its defects, with its index tables, in the
        reference's registration order (transcribe_phase, ODEPhaseBase.cpp:1371-1375: dynamics, axis functions, control
        functions, integrals, user functions).  Returns (entries, numPhaseEqCons, numPhaseIqCons); an entry is
        (kind, tag, function, device name, Vindex, Cindex, per-application constants or None), kind in
        {"auto", "equality", "inequality", "objective"}."""
        from .pathfuncs import LGLControlSpline, LGLIntegral, LGLMeshSpacing, SingleMeshSpacing
        next_eq, next_iq = ix.con_offset + ix.numPhaseEqCons, int(iq_offset)
        cs, D, tv = ix.DefectCardinalStates, ix.numDefects, [self.ode.TVar()]
        out = []
        if self.EnableMeshSpacing:
            if self.TranscriptionMode in ("LGL5", "LGL7"):     # transcribe_axis_funcs: LGLMeshSpacing<CS> over DefectPath
                V, Cx, next_eq = ix.make_Vindex_Cindex("DefectPath", tv, (), (), cs - 2, next_eq)
                out.append(("auto", "mesh_spacing", LGLMeshSpacing(cs), f"lglmeshspacing{cs}", V, Cx, None))
            if D >= 2:
                # SingleMeshSpacing(i / D) at the inner nodal states i = 1..D-1 (FrontNodalBackPath; the reference adds one
                # function object per state, addPartitionedEquality): ONE device function whose spacing is a constant of the
                # application (vf.ApplConst), so the D - 1 relations are one batched evaluator
                V, Cx, next_eq = ix.make_Vindex_Cindex("FrontNodalBackPath", tv, (), (), 1, next_eq)
                # the spacing constants come from the phase's DefBinSpacing / DefsPerBin (:963-970), not from the times the
                # active trajectory happens to hold: i / D on the uniform mesh setTraj(traj, n) builds
                cspace = self._nodal_spacing()[1:-1]
                out.append(("auto", "nodal_spacing", SingleMeshSpacing(None), "nodalmeshspacing", V, Cx, cspace[:, None]))
        if self.ode.UVars() > 0 and not self._blocked() and D >= 2:   # transcribe_control_funcs
            order = {("LGL7", "HighestOrderSpline"): 2, ("LGL7", "FirstOrderSpline"): 1, ("LGL5", "HighestOrderSpline"): 1,
                     ("LGL5", "FirstOrderSpline"): 1}.get((self.TranscriptionMode, self.ControlMode))
            if order:
                tu = tv + list(range(self.ode.TVar() + 1, self.ode.TVar() + 1 + self.ode.UVars()))
                F = LGLControlSpline(cs, self.ode.UVars(), order)
                V, Cx, next_eq = ix.make_Vindex_Cindex("DefectPairWisePath", tu, (), (), F.ORows(), next_eq)
                out.append(("auto", "control_spline", F, f"lglcontrolspline{cs}_{self.ode.UVars()}_{order}", V, Cx, None))
        # integral objectives: LGLIntegral over every defect (ODEPhaseBase.cpp:743-889)
        for k, (integrand, xtuv, opv, spv, oscale) in enumerate(self._integral_objs):
            f = LGLIntegral(self._scaled_integrand(integrand, xtuv, opv, spv, oscale), cs, len(xtuv), len(opv) + len(spv))
            V, _, _ = ix.make_Vindex_Cindex("DefectPath", xtuv + tv, opv, spv, 0, 0)
            Cx = np.zeros((V.shape[0], 1), dtype=np.int32)           # every application reads multiplier 0 = ObjScale
            out.append(("objective", f"obj{k}", f, f"obj{k}_integral{cs}", V, Cx, None))
        # integral parameter functions (ODEPhaseBase.cpp:835-889; addAccumulation, PhaseIndexer.cpp:41-76): the accumulation
        # -scale * p over Params claims the row, every application of the quadrature carries that same row
        for k, (integrand, xtuv, opv, spv, acc, scale, oscale) in enumerate(self._integral_params):
            from .vf import Arguments
            acc_scale = 1.0
            if self.AutoScaling:      # AccScale = pscale * output_scale / t_unit (ODEPhaseBase.cpp:857-859)
                acc_scale = getattr(self, "SPUnits", np.ones(len(self.ActiveStaticParams)))[acc] * oscale / self.XtUPUnits[self.ode.TVar()]
            Va, Ca, next_eq = ix.make_Vindex_Cindex("Params", (), (), [acc], 1, next_eq)
            out.append(("equality", f"ipf{k}_acc", Arguments(1) * (-scale * acc_scale), f"ipf{k}_accumulate", Va, Ca, None))
            f = LGLIntegral(self._scaled_integrand(integrand, xtuv, opv, spv, oscale), cs, len(xtuv), len(opv) + len(spv))
            V, _, _ = ix.make_Vindex_Cindex("DefectPath", xtuv + tv, opv, spv, 0, 0)
            Cx = np.full((V.shape[0], 1), int(Ca[0, 0]), dtype=np.int32)
            out.append(("equality", f"ipf{k}_int", f, f"ipf{k}_integral{cs}", V, Cx, None))
        for store, is_eq in ((self._eq_funcs, True), (self._iq_funcs, False)):
            for k, (region, func, xtuv, opv, spv) in enumerate(store):
                f = self._scaled_func(region, func, xtuv, opv)
                V, Cx, nxt = ix.make_Vindex_Cindex(region, xtuv, opv, spv, f.ORows(), next_eq if is_eq else next_iq)
                if is_eq:
                    next_eq = nxt
                else:
                    next_iq = nxt
                out.append(("equality" if is_eq else "inequality", f"{'eq' if is_eq else 'iq'}{k}", f,
                            f"{'eq' if is_eq else 'iq'}{k}_{region.lower()}", V, Cx, None))
        return out, next_eq - ix.con_offset, next_iq - int(iq_offset)

    def _make_function_evaluators(self, ix, build_only: bool = False):
        from .pathfuncs import FunctionEvaluator as _FE

        def FunctionEvaluator(F, name, *args, **kw):      # build_only: device code only (no handle: works without a GPU)
            return jit.ensure_function(F, name) if build_only else _FE(F, name, *args, **kw)
        entries, self.numPhaseEqCons, self.numPhaseIqCons = self._function_tables(ix)
        # every equality evaluator (the defects included) takes the phase's whole equality multiplier vector
        n_eq, n_iq = ix.con_offset + self.numPhaseEqCons, self.numPhaseIqCons
        self._eq_evs, self._iq_evs, self._obj_evs, self._auto_evs = [], [], [], {}
        self._ipf_evs = {}
        for kind, tag, F, name, V, Cx, consts in entries:
            ncon = {"auto": n_eq, "equality": n_eq, "inequality": n_iq, "objective": 1}[kind]
            kw = {"appl_consts": consts} if consts is not None else {}
            ev = FunctionEvaluator(F, name, V, Cx, ix.numPhaseVars, ncon, self.device, **kw)
            if kind == "auto":
                self._auto_evs[tag] = ev
            elif tag.startswith("ipf"):
                self._ipf_evs.setdefault(int(tag[3:].split("_")[0]), {})[tag.split("_")[1]] = ev
            else:
                {"equality": self._eq_evs, "inequality": self._iq_evs, "objective": self._obj_evs}[kind].append(ev)

    def layout(self, Vstart: int = 0, Estart: int = 0, Istart: int = 0):
        """Host-only description of what the phase hands the solver (no device needed): the PhaseIndexer, the defect
        tables and the entries of ``_function_tables``, with the phase's variables starting at ``Vstart`` of the solver
        vector, its equality rows at ``Estart`` and its inequality rows at ``Istart`` -- the arguments of the reference's
        ``transcribe_phase(Vstart, Estart, Istart, ...)`` (OptimalControlProblem.cpp:131-146).  -> (indexer, (Vindex, Cindex)
        of the defects, entries, numPhaseEqCons, numPhaseIqCons)."""
        if self.ActiveTraj is None:
            raise RuntimeError("No trajectory set: call setTraj first")
        ix = PhaseIndexer(self.ode.XVars(), self.ode.UVars(), self.ode.PVars(), len(self.ActiveStaticParams))
        ix.set_dimensions(synth.MODE_CS[self.TranscriptionMode], self.numDefects, self._blocked())
        ix.begin_indexing(Vstart, Estart)
        entries, neq, niq = self._function_tables(ix, Istart)
        return ix, ix.make_defect_Vindex_Cindex(), entries, neq, niq

    @property
    def equality_evaluators(self):
        """Device evaluators of the functions added with addEqualCon, in order (after ``evaluator``, the defects)."""
        if self._ev is None:
            self.transcribe()
        return list(self._eq_evs)

    @property
    def integral_param_evaluators(self):
        """[(accumulation evaluator, quadrature evaluator)] of the integral parameter functions: two equality functions on
        one constraint row each (addIntegralParamFunction)."""
        if self._ev is None:
            self.transcribe()
        return [(self._ipf_evs[k]["acc"], self._ipf_evs[k]["int"]) for k in sorted(self._ipf_evs)]

    @property
    def objective_evaluators(self):
        """Device evaluators of the integral objectives (one output per defect; multiplier vector = [ObjScale])."""
        if self._ev is None:
            self.transcribe()
        return list(self._obj_evs)

    @property
    def phase_function_evaluators(self):
        """The equalities the phase adds itself: {"mesh_spacing", "nodal_spacing", "control_spline"} -> evaluator."""
        if self._ev is None:
            self.transcribe()
        return dict(self._auto_evs)

    @property
    def inequality_evaluators(self):
        if self._ev is None:
            self.transcribe()
        return list(self._iq_evs)

    def function_bundle(self):
        """Everything the phase hands the solver beside its defects -- the equalities it adds itself, the user's path
        equalities and inequalities, the integral objectives -- as ONE launch (pathfuncs.FunctionBundle).  Returns
        (bundle, members) with members = [(kind, evaluator)], kind in {"equality", "inequality", "objective"}: the
        multiplier vector of an equality is LE, of an inequality LI, of an objective [ObjScale]."""
        from .pathfuncs import FunctionBundle
        if self._ev is None:
            self.transcribe()
        members = [("equality", e) for e in self._auto_evs.values()] + [("equality", e) for e in self._eq_evs] + \
                  [("inequality", e) for e in self._iq_evs] + [("objective", e) for e in self._obj_evs]
        if not members:
            raise RuntimeError("the phase has no function beside its defects")
        return FunctionBundle([e for _, e in members]), members

    # ---- transcription ------------------------------------------------------------------------
    def transcribe(self):
        if self.ActiveTraj is None:
            raise RuntimeError("No trajectory set: call setTraj first")
        # library ODEs are compiled into libasset_hip.so; any other ODEBase gets device code on first use (jit.py)
        name = jit.ensure_kernel(self._active_ode(), self.TranscriptionMode, self._blocked())
        ix = PhaseIndexer(self.ode.XVars(), self.ode.UVars(), self.ode.PVars(), len(self.ActiveStaticParams))
        ix.set_dimensions(synth.MODE_CS[self.TranscriptionMode], self.numDefects, self._blocked())
        ix.begin_indexing(0, 0)
        V, Cx = ix.make_defect_Vindex_Cindex()
        self._indexer = ix
        self._make_function_evaluators(ix)
        # a re-meshed phase (refineTrajManual: another number of segments, same dynamics and transcription) keeps its device
        # handle and gives it new index tables (asset_hip_defect_rebind) instead of creating one: the re-meshing step of the
        # adaptive mesh loop, ODEPhaseBase.cpp:1443-1542
        key = (name, self.TranscriptionMode, self._blocked(), self.device)
        kept = self._ev_kept
        self._ev = None
        if kept is not None and kept[0] == key and kept[1]._h:
            # NOTE the aliasing: the kept evaluator is re-pointed IN PLACE -- a reference handed out earlier through ``phase.evaluator``
            # (a pre-refinement evaluator kept for comparison, a handle given to an adapter) now evaluates the NEW mesh and has lost its
            # KKT map, per-application constants and pinned outputs.  A handle that cannot be re-bound (it is a member of a function
            # bundle: EINVAL) or a failed re-bind gets a fresh evaluator instead; the old object is left to its holders.
            try:
                self._ev = kept[1].rebind(V, Cx, ix.numPhaseVars, self.numPhaseEqCons)
            except Exception:
                self._ev_kept = None
        if self._ev is None:
            if kept is not None and self._ev_kept is not None:
                kept[1].close()
            self._ev = DefectEvaluator(name, self.TranscriptionMode, self._blocked(), V, Cx, ix.numPhaseVars,
                                       self.numPhaseEqCons, self.device)
            self._ev_kept = (key, self._ev)
        return self

    def prebuild_device_code(self):
        """Generate and compile (or find in the in-tree cache) the device code of everything this phase would hand the
        solver -- defects, the functions it registers itself, user functions, objectives -- without creating a device
        handle: runs where there is a compiler but no GPU (the build step), so that a GPU box only loads plugins."""
        if self.ActiveTraj is None:
            raise RuntimeError("No trajectory set: call setTraj first")
        jit.ensure_kernel(self._active_ode(), self.TranscriptionMode, self._blocked())
        ix = PhaseIndexer(self.ode.XVars(), self.ode.UVars(), self.ode.PVars(), len(self.ActiveStaticParams))
        ix.set_dimensions(synth.MODE_CS[self.TranscriptionMode], self.numDefects, self._blocked())
        ix.begin_indexing(0, 0)
        self._make_function_evaluators(ix, build_only=True)      # (build_only: the lists hold device names, not evaluators)
        names = list(self._auto_evs.values()) + list(self._eq_evs) + list(self._iq_evs) + list(self._obj_evs)
        if 1 <= len(names) <= 8:
            jit.ensure_bundle(names)                              # function_bundle(): the same list in one launch
        self._eq_evs, self._iq_evs, self._obj_evs, self._auto_evs, self._ipf_evs = [], [], [], {}, {}
        return self

    def solver_input(self) -> np.ndarray:
        if self._indexer is None:
            self.transcribe()
        traj = self.ActiveTraj / self.XtUPUnits if self.AutoScaling else self.ActiveTraj   # variables in scaled units
        sp = self.ActiveStaticParams
        if self.AutoScaling and len(sp):
            sp = sp / getattr(self, "SPUnits", np.ones(len(sp)))
        return self._indexer.makeSolverInput(traj, sp if len(sp) else None)

    @property
    def evaluator(self) -> DefectEvaluator:
        if self._ev is None:
            self.transcribe()
        return self._ev

    # ---- mesh error (device de Boor estimator; the refinement loop itself stays with the host) ----------------
    def get_meshinfo_deboor(self):
        """(tsnd, mesh_errors[XV, nb+1], mesh_dist[XV, nb+1]) -- ODEPhase.h:442-585.  The estimate is taken on the
        unscaled dynamics and trajectory; with AutoScaling the results are converted to scaled units the way
        ODEPhase.h:551-559 does (y / x-units * t-unit^Order, h / t-unit)."""
        from . import mesh
        name = jit.ensure_kernel(self.ode, self.TranscriptionMode, self._blocked())
        tsnd, err, dist = mesh.mesh_error_deboor(name, self.TranscriptionMode, self.ActiveTraj, self._blocked(),
                                                 self.device)[:3]
        if self.AutoScaling:
            xv = self.ode.XVars()
            ux, ut = self.XtUPUnits[:xv, None], self.XtUPUnits[xv]
            order = {"Trapezoidal": 2.0, "LGL3": 3.0, "LGL5": 5.0, "LGL7": 7.0}[self.TranscriptionMode]
            err = err / ux
            dist = dist * (ut ** (order + 1) / ux) ** (1.0 / (order + 1))
        return tsnd, err, dist

    def getMeshInfo(self, integ: bool = False, n: int = 100):
        """(tsnd, bins, error) -- ODEPhaseBase.h:1355-1399; only the de Boor estimator is provided."""
        if integ:
            raise NotImplementedError("the integrator-based estimator needs the ODE integrator, which stays on the host")
        from . import mesh
        tsnd, err, dist = self.get_meshinfo_deboor()
        return mesh.bins_from_density(tsnd, np.abs(err).max(axis=0), np.abs(dist).max(axis=0), n)

    # ---- the adaptive mesh loop's two steps (ODEPhaseBase.cpp:1443-1494 checkMesh, :1496-1542 updateMesh).  The solver between
    #      them is the caller's: solve, checkMesh(); while not converged: updateMesh(), solve again, checkMesh()
    def setAdaptiveMesh(self, flag: bool = True):
        self.AdaptiveMesh = bool(flag)

    def setMeshTol(self, tol: float):
        self.MeshTol = abs(float(tol))

    def checkMesh(self, meshinfo=None) -> bool:
        """Estimate the error of the active trajectory (the device de Boor estimator; `meshinfo`: another source of
        (tsnd, mesh_errors[XV, nb+1], mesh_dist[XV, nb+1]) -- a test's oracle), record the iterate and compare the criterion
        ("max", "avg", "geometric": MeshIterateInfo.h:42-47) with MeshTol."""
        from .mesh import MeshIterateInfo
        tsnd, err, dist = meshinfo() if meshinfo is not None else self.get_meshinfo_deboor()
        it = MeshIterateInfo(self.numDefects, self.MeshTol, tsnd, np.abs(err).max(axis=0), np.abs(dist).max(axis=0))
        if self.MeshErrorCriteria not in ("max", "avg", "geometric"):
            raise ValueError("Unknown mesh error criteria")      # ("endtoend" needs the integrator, which stays on the host)
        crit = {"max": it.max_error, "avg": it.avg_error, "geometric": it.gmean_error}[self.MeshErrorCriteria]
        it.converged = self.MeshConverged = bool(crit < self.MeshTol)
        self.MeshIters.append(it)
        return self.MeshConverged

    def updateMesh(self):
        """The next mesh from the last iterate: per segment (err * MeshErrFactor / MeshTol)^(1 / (Order + 1)) new segments (at least
        MeshRedFactor), summed, plus NumExtraSegs, kept between MeshRedFactor and MeshIncFactor times the current number and
        between MinSegments and MaxSegments; their edges equidistribute the error density; the trajectory is re-distributed
        (control-switch detection, :1511-1536, is not built)."""
        if not self.MeshIters:
            raise RuntimeError("checkMesh first")
        it = self.MeshIters[-1]
        order = {"Trapezoidal": 2.0, "LGL3": 3.0, "LGL5": 5.0, "LGL7": 7.0}[self.TranscriptionMode]
        per = np.maximum(self.MeshRedFactor, (it.error[:-1] * self.MeshErrFactor / self.MeshTol) ** (1.0 / (order + 1.0)))
        n = int(np.ceil(per.sum())) + self.NumExtraSegs
        n = min(max(n, int(self.numDefects * self.MeshRedFactor)), int(self.numDefects * self.MeshIncFactor))
        n = min(max(n, self.MinSegments), self.MaxSegments)
        bins = it.calc_bins(n)
        it.up_numsegs = bins.size - 1
        return self.refineTrajManual(bins, np.ones(bins.size - 1, dtype=int))

    def get_defect(self) -> DefectFunction:
        name = jit.ensure_kernel(self._active_ode(), self.TranscriptionMode, self._blocked())
        return DefectFunction(name, self.TranscriptionMode, self._blocked(), self.device)

    # ---- the reference's built-in benchmark of the path ---------------------------------------------
    def test_threads(self, i: int = 1, j: int = 1, n: int = 100, verbose: bool = True):
        """NLPTest analogue (NonLinearProgram.cpp:686-820): n x {evalKKT, evalOCC} of the defect constraint with
        multipliers 100*U(-1,1); prints mean ms per call.  The thread counts are accepted for signature
        compatibility; the device evaluates all segments in one launch."""
        ev = self.evaluator
        X = self.solver_input()
        rng = np.random.default_rng(0)
        t_kkt = t_occ = 0.0
        for _ in range(n):
            L = 100.0 * rng.uniform(-1, 1, ev.n_equal)
            t0 = time.perf_counter()
            ev.eval(JAC_ADJGRAD_HESS, X, L)
            t_kkt += time.perf_counter() - t0
            t0 = time.perf_counter()
            ev.eval(CON, X)
            t_occ += time.perf_counter() - t0
        res = {"evalKKT_ms": 1e3 * t_kkt / n, "evalOCC_ms": 1e3 * t_occ / n, "segments": ev.nseg}
        if verbose:
            print(f"{res['evalKKT_ms']} ms\n{res['evalOCC_ms']} ms")
        return res
