"""``OptimalControlProblem`` -- the part of the reference's multi-phase container that places phases in ONE solver
vector (host only; links between phases, the solve / optimize drivers and the mesh loop stay with the host solver):

  ``addPhase`` / ``addPhases`` / ``Phase(i)`` / ``getPhaseNum``         OptimalControlProblem.h (phase list)
  ``transcribe_phases``                                                  OptimalControlProblem.cpp:115-155

``transcribe_phases`` gives phase i the offsets ``Vstart = sum(numPhaseVars[:i])``, ``Estart = sum(numPhaseEqCons[:i])``,
``Istart = sum(numPhaseIqCons[:i])`` -- every phase's variables, equality rows (its defects first, then what it registers
beside them) and inequality rows one after the other.  Phases are independent given X and L, so a multi-GPU job deals
whole phases to its ranks (``distributed.PhaseShardedEvaluator``, BASELINE.json configs[3]); this class is what feeds it
the index tables instead of hand-rolled offsets.
"""
from __future__ import annotations

from typing import List, Sequence

import numpy as np


class OptimalControlProblem:
    def __init__(self):
        self.phases: List = []
        self.numPhaseVars: List[int] = []
        self.numPhaseEqCons: List[int] = []
        self.numPhaseIqCons: List[int] = []
        self._layouts = None

    # ---- phase list ---------------------------------------------------------------------------------------
    def addPhase(self, phase) -> int:
        if any(p is phase for p in self.phases):
            raise ValueError("Attempting to add the same phase to the problem twice")
        self.phases.append(phase)
        self._layouts = None
        return len(self.phases) - 1

    def addPhases(self, phases: Sequence) -> List[int]:
        return [self.addPhase(p) for p in phases]

    def Phase(self, i: int):
        return self.phases[i]

    def getPhaseNum(self, phase) -> int:
        for i, p in enumerate(self.phases):
            if p is phase:
                return i
        raise ValueError("phase does not belong to this problem")

    # ---- layout -------------------------------------------------------------------------------------------
    def transcribe_phases(self):
        """-> list over phases of (indexer, (Vindex, Cindex) of the defects, entries, numPhaseEqCons, numPhaseIqCons), the
        tuples of ``Phase.layout`` at the phase's offsets.  Also fills numPhaseVars / numPhaseEqCons / numPhaseIqCons."""
        self.numPhaseVars, self.numPhaseEqCons, self.numPhaseIqCons, out = [], [], [], []
        for ph in self.phases:
            lay = ph.layout(sum(self.numPhaseVars), sum(self.numPhaseEqCons), sum(self.numPhaseIqCons))
            out.append(lay)
            self.numPhaseVars.append(lay[0].numPhaseVars)
            self.numPhaseEqCons.append(lay[3])
            self.numPhaseIqCons.append(lay[4])
        self._layouts = out
        return out

    @property
    def layouts(self):
        return self._layouts if self._layouts is not None else self.transcribe_phases()

    @property
    def n_primal(self) -> int:
        self.layouts
        return int(sum(self.numPhaseVars))

    @property
    def n_equal(self) -> int:
        self.layouts
        return int(sum(self.numPhaseEqCons))

    @property
    def n_inequal(self) -> int:
        self.layouts
        return int(sum(self.numPhaseIqCons))

    def defect_tables(self):
        """[(Vindex, Cindex)] of every phase's defects, indices into the problem's X and L."""
        return [lay[1] for lay in self.layouts]

    def solver_input(self) -> np.ndarray:
        """The phases' trajectories as one solver vector."""
        X = np.zeros(self.n_primal)
        for ph, lay in zip(self.phases, self.layouts):
            ix = lay[0]
            traj = ph.ActiveTraj / ph.XtUPUnits if ph.AutoScaling else ph.ActiveTraj
            v = ix.var_offset
            ix.begin_indexing(0, ix.con_offset)
            X[v:v + ix.numPhaseVars] = ix.makeSolverInput(traj)
            ix.begin_indexing(v, ix.con_offset)
        return X

    # ---- multi-GPU: whole phases per rank -------------------------------------------------------------------
    def phase_sharded_evaluator(self, rank=None, world=None, device: int = 0, group=None, evaluator_factory=None):
        """``distributed.PhaseShardedEvaluator`` over this problem's phases (same ODE, transcription, control mode and
        segment count in every phase -- the configs[3] shape)."""
        from . import jit
        from .distributed import PhaseShardedEvaluator
        p0 = self.phases[0]
        key = lambda p: (p.ode.ode_name, p.TranscriptionMode, p._blocked(), p.numDefects)
        if any(key(p) != key(p0) for p in self.phases):
            raise ValueError("phase_sharded_evaluator: the phases differ in ODE, transcription, control mode or size")
        name = p0.ode.ode_name if evaluator_factory is not None else jit.ensure_kernel(p0._active_ode(), p0.TranscriptionMode, p0._blocked())
        return PhaseShardedEvaluator(name, p0.TranscriptionMode, p0._blocked(), self.defect_tables(), self.n_primal,
                                     self.n_equal, rank=rank, world=world, device=device, group=group,
                                     evaluator_factory=evaluator_factory)
