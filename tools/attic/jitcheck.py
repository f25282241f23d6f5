"""One-off (round 6): does a module pre-built by build() here hit the cache on the GPU box (a snapshot under another root)?"""
import glob, os, sys, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
from asset_asrl_amd import jit, build
from helpers import make_tabulated, make_vanderpol
print("cwd", os.getcwd(), "CSRC", build.CSRC)
before = set(glob.glob(os.path.join(jit.JIT_DIR, "*", "*.rtc")))
for ode, mode, blocked in ((make_tabulated(), "LGL5", False), (make_vanderpol(), "LGL7", True)):
    t = time.time()
    name = jit.ensure_kernel(ode, mode, blocked)
    print("ensure_kernel", name, mode, blocked, round(time.time() - t, 2), "s")
print("cached modules before:", len(before), "new files:", sorted(set(glob.glob(os.path.join(jit.JIT_DIR, "*", "*.rtc"))) - before))
