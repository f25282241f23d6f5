"""Shuttle-reentry phase on an MI355X: what a host solver would call per iteration.

    python examples/reentry_phase.py [nsegs]

Builds the LGL7 transcription of the reentry dynamics (reference: examples/Reentry.py:30-97) on a synthetic
trajectory, evaluates the defect constraint's value / adjoint gradient / Jacobian / adjoint Hessian for every segment
(the evalKKT-equivalent), assembles the KKT entries on the device into an upper-triangular CSR value array, and
estimates the mesh error.  Needs a GPU: there is no CPU fallback."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from asset_asrl_amd import synth                                   # noqa: E402
from asset_asrl_amd.evaluator import JAC_ADJGRAD_HESS, unpack_kkt_block   # noqa: E402
from asset_asrl_amd.ode import ShuttleReentry                     # noqa: E402

nsegs = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
ode = ShuttleReentry()
phase = ode.phase("LGL7", synth.make_traj("reentry", "LGL7", nsegs), nsegs)
ev = phase.evaluator                                              # transcribes: index tables + device handle
X = phase.solver_input()                                          # NLP variables [node states ..., parameters]
L = synth.make_multipliers(ev.n_equal)                            # equality multipliers, 100*U(-1,1) like NLPTest

t0 = time.perf_counter()
fx, agx, kkt = ev.eval(JAC_ADJGRAD_HESS, X, L)                     # host pointers in, blocks out
dt = time.perf_counter() - t0
H, J = unpack_kkt_block(kkt[0], ev.IR, ev.OR)
print(f"{nsegs} segments: defect blocks {fx.shape}, adjoint-gradient blocks {agx.shape}, KKT blocks {kkt.shape} "
      f"({1e3 * dt:.2f} ms incl. PCIe)")
print(f"segment 0: |d| = {np.abs(fx[0]).max():.3e}, J {J.shape}, H {H.shape}, |J^T lam - g| = "
      f"{np.abs(J.T @ L[:ev.OR] - agx[0]).max():.1e}")

tsnd, bins, err = phase.getMeshInfo(False, nsegs)
print(f"de Boor mesh error: max {err.max():.3e} at t/T = {tsnd[err.argmax()]:.3f}")

defect = phase.get_defect()                                       # one segment as a VectorFunction: z[IR] -> d[OR]
fx1, jx1, gx1, hx1 = defect.computeall(X[ev.vindex[0]], L[:ev.OR])
print(f"get_defect().computeall: fx {fx1.shape} jx {jx1.shape} gx {gx1.shape} hx {hx1.shape}")

# a user path constraint over every state of the mesh (reference: phase.addInequalCon("Path", ...), Reentry.py heating
# constraint): written in the expression DSL, compiled for the device on first use, evaluated for all states at once
from asset_asrl_amd import vf                                     # noqa: E402
a = vf.Arguments(3)                                               # altitude, velocity, angle of attack (scaled units)
qdot = 0.2 * vf.exp(-2.0 * a[0]) * a[1] * a[1] * a[1] * (1.0 + a[2] * a[2]) - 70.0
phase.addInequalCon("Path", vf.stack([qdot]), [0, 3, 6])
(iq,) = phase.inequality_evaluators                               # re-transcribes: one device evaluator per function
X = phase.solver_input()
fq, gq, kq = iq.eval(JAC_ADJGRAD_HESS, X, np.ones(phase.numPhaseIqCons))
print(f"path inequality at {iq.nseg} states: values {fq.shape}, gradient blocks {gq.shape}, KKT blocks {kq.shape}, "
      f"max value {fq.max():.3f}")
