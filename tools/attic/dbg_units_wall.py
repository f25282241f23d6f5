"""Phase stamps (100 MHz wall clock) of every workgroup of the heavy-ODE unit kernels, from a -DASSET_WALLCLOCK build.

  python tools/build_one.py tu_betts_lowthrust_lgl3_0 build_dbg/uw/lib.so -DASSET_WALLCLOCK
  ASSET_HIP_TUNING=1 ASSET_HIP_SKIP_DENSE=1 ASSET_HIP_LIB=build_dbg/uw/lib.so python tools/dbg_units_wall.py [nseg]
"""
import sys, os, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from helpers import Workload
from asset_asrl_amd.evaluator import DefectEvaluator
nseg = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
mode = sys.argv[2] if len(sys.argv) > 2 else "LGL5"
w = Workload("betts_lowthrust", mode, nseg, False)
ev = DefectEvaluator("betts_lowthrust", mode, False, w.vindex, w.cindex, w.n_primal, w.n_equal)
NU, CS = 7, {"LGL3": 2, "LGL5": 3, "LGL7": 4}[mode]
gp = min(max((nseg * NU + 1023) // 1024, 1), 64 // CS)
ng = (nseg + gp - 1) // gp
names = ["start", "gathered", "P1 done", "P2 done", "flag seen", "P3 done"]
for rep in range(3):
    fx, agx, kkt = ev.eval(4, w.X, w.L)
    st = np.asarray(agx).ravel()[:NU * ng * 8]
    st = (st.reshape(NU, ng, 8) if os.environ.get('ASSET_HIP_NO_UNITS_FUSE') else st.reshape(ng, NU, 8).transpose(1, 0, 2))[:, :, :6]
    base = st[:, :, 0].min()
    us = (st - base) / 100.0
    print(f"rep {rep}: gp {gp} groups {ng}")
    d = us[:, :, 3] - us[:, :, 2]
    print("  P2 duration percentiles 50/90/99/max:", np.percentile(d, [50, 90, 99, 100]).round(2), " per unit max:", d.max(axis=1).round(1))
    slow = np.argwhere(d > 1.5 * np.median(d))
    print("  slow workgroups (unit, group):", [tuple(x) for x in slow[:40]], len(slow))
    for r in range(NU):
        print(f"  unit {r}: " + " | ".join(f"{names[k]} {np.median(us[r, :, k]):6.2f} (max {us[r, :, k].max():6.2f})" for k in range(6)))
