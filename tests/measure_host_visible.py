"""Host-visible evaluation rates of the bench workload: what a solver on the HOST pays per
evaluation through each host-pointer entry point (PCIe included), next to the device-resident time.

  python tests/measure_host_visible.py            # on the GPU box

Lives under tests/ because it borrows the oracle's sparsity analysis in place of the host solver that would own it
(only tests/, smoke() and bench.py's cpu_baseline leg may touch oracle/); nothing of the oracle is timed.
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import Workload  # noqa: E402

from asset_asrl_amd import _lib  # noqa: E402
from asset_asrl_amd.evaluator import JAC_ADJGRAD_HESS, DefectEvaluator  # noqa: E402
from oracle import bindings as ob  # noqa: E402  (the sparsity analysis a host solver owns; not timed)


def timeit(f, n=20):
    for _ in range(3):
        f()
    t0 = time.perf_counter()
    for _ in range(n):
        f()
    return (time.perf_counter() - t0) / n * 1e3


def main():
    w = Workload("reentry", "LGL7", 10000)
    ev = DefectEvaluator("reentry", "LGL7", False, w.vindex, w.cindex, w.n_primal, w.n_equal)
    out = {"workload": "reentry LGL7 x 10000"}
    out["blocks_pageable_ms"] = timeit(lambda: ev.eval(JAC_ADJGRAD_HESS, w.X, w.L))
    ev.pin_outputs()
    out["blocks_pinned_ms"] = timeit(lambda: ev.eval(JAC_ADJGRAD_HESS, w.X, w.L))
    nlp = w.oracle_nlp(ob, threads=8)
    locs = nlp.kkt_locations()[:nlp.num_user_kkt].reshape(w.nseg, ev.NKKT)
    ev.set_kkt_map(locs, nlp.nnz)
    vals = np.zeros(nlp.nnz)
    _lib.check(_lib.lib().asset_hip_host_register(vals.ctypes.data, vals.nbytes))

    def assembled():
        vals.fill(0.0)
        ev.eval_assembled(JAC_ADJGRAD_HESS, w.X, w.L, vals)

    def assembled_zeroed():
        ev.eval_assembled(JAC_ADJGRAD_HESS, w.X, w.L, vals, target_zeroed=True)
    out["assembled_add_ms"] = timeit(assembled)
    out["assembled_zeroed_pinned_ms"] = timeit(assembled_zeroed)
    out["nnz"] = int(nlp.nnz)
    out["bytes_blocks"] = int(w.nseg * (ev.NKKT + ev.IR + ev.OR) * 8)
    out["bytes_values"] = int(nlp.nnz * 8)
    _lib.lib().asset_hip_host_unregister(vals.ctypes.data)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
