"""Start / end wall-clock (100 MHz) of every workgroup of the fused launch, from a -DASSET_WALLCLOCK build.

  python tools/build_one.py tu_reentry_lgl4_0 build_dbg/libdbgW.so -DASSET_WALLCLOCK
  ASSET_HIP_LIB=build_dbg/libdbgW.so python tools/dbg_wall.py [nseg]
"""
import sys, os, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0,ROOT); sys.path.insert(0,os.path.join(ROOT,'tests'))
from helpers import Workload
from asset_asrl_amd.evaluator import *
nseg=int(sys.argv[1]) if len(sys.argv)>1 else 10000
w=Workload("reentry","LGL7",nseg,False)
ev=DefectEvaluator("reentry","LGL7",False,w.vindex,w.cindex,w.n_primal,w.n_equal)
G=min(nseg,2048); per=nseg//G; rem=nseg%G
first=np.array([s*per+min(s,rem) for s in range(G)])
for rep in range(4):
    fx,agx,kkt=ev.eval(4,w.X,w.L)
    fx=fx.reshape(nseg,-1)
    t0=fx[first,0]; t1=fx[first,1]
    base=t0.min()
    st=(t0-base)/100.0; en=(t1-base)/100.0      # us
    print(f"rep {rep}: kernel span {en.max():.1f} us | start: median {np.median(st):.2f} p90 {np.percentile(st,90):.2f} max {st.max():.2f} | "
          f"duration: min {np.min(en-st):.1f} median {np.median(en-st):.1f} p90 {np.percentile(en-st,90):.1f} max {np.max(en-st):.1f} | "
          f"end: min {en.min():.1f} median {np.median(en):.1f} max {en.max():.1f}")
    if rep==3:
        d=en-st; cnt=np.array([per+(1 if s<rem else 0) for s in range(G)])
        for c in sorted(set(cnt)): print(f"   workgroups with {c} segments: {np.sum(cnt==c)}; duration median {np.median(d[cnt==c]):.1f} max {d[cnt==c].max():.1f}")
        hw=fx[first,2].astype(np.int64); xcc=fx[first,3].astype(np.int64)&15
        cu=(hw>>8)&15; sh=(hw>>12)&1; se=(hw>>13)&7; simd=(hw>>4)&3
        print("   HW_ID sample", [hex(v) for v in hw[:4]], "xcc", xcc[:16])
        for x in sorted(set(xcc)):
            m=xcc==x
            print(f"   xcc {x}: n {m.sum()} start median {np.median(st[m]):.2f} max {st[m].max():.2f} | duration median {np.median(d[m]):.1f} max {d[m].max():.1f} | end median {np.median(en[m]):.1f} max {en[m].max():.1f}")
        cuid=xcc*1000+se*100+sh*50+cu
        per_cu=[(c,(cuid==c).sum(),en[cuid==c].max(),d[cuid==c].mean()) for c in sorted(set(cuid))]
        n=np.array([p[1] for p in per_cu]); print("   CUs seen", len(per_cu), "waves per CU: min", n.min(), "max", n.max(), "hist", np.bincount(n))
        for k in sorted(set(n)):
            sel=[p for p in per_cu if p[1]==k]
            print(f"   CUs with {k} workgroups: {len(sel)}; mean duration {np.mean([p[3] for p in sel]):.1f}; latest end {np.max([p[2] for p in sel]):.1f}")
        idx=np.arange(G)
        for lo in range(0,G,256): m=(idx>=lo)&(idx<lo+256); print(f"   wg [{lo},{lo+256}): start median {np.median(st[m]):.2f}  end median {np.median(en[m]):.1f} max {en[m].max():.1f}")
