"""Experiment helper: build a library holding ONE translation unit (+ the C ABI) with extra hipcc flags.

  python tools/build_one.py tu_reentry_lgl4_0 build_dbg/libdbg.so -DASSET_TIMING
  ASSET_HIP_LIB=build_dbg/libdbg.so python tools/dbg_time.py
"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from asset_asrl_amd import build as B

tu, out = sys.argv[1], os.path.join(ROOT, sys.argv[2])
extra = sys.argv[3:]
# stamps and elimination experiments are refused by the headers unless the build says it is a measurement build
if any(f.startswith(("-DASSET_TIMING", "-DASSET_WALLCLOCK", "-DASSET_FUNC_TIMING", "-DASSET_EXP_")) for f in extra):
    extra.append("-DASSET_TUNING_BUILD")
B.generate(verbose=False)
os.makedirs(os.path.dirname(out), exist_ok=True)
objs = []
for src in (os.path.join(B.GEN, tu + ".hip"), os.path.join(B.CSRC, "capi.hip"), os.path.join(B.CSRC, "capi_sharded.hip")):
    obj = os.path.join(os.path.dirname(out), os.path.basename(src) + ".o")
    subprocess.check_call([B.HIPCC] + B.FLAGS + B.tu_flags(src) + extra + ["-c", src, "-o", obj])
    objs.append(obj)
subprocess.check_call([B.HIPCC, "--offload-arch=" + B.ARCH, "-shared", "-fPIC", "-o", out] + objs + ["-lhiprtc"])
print(out)
