"""Run-time device code for user-defined ODEs.

The reference takes any VectorFunction as ODE right-hand side (``oc.ode_x_u_p.ode(vf, Xv, Uv, Pv)``,
/root/reference/src/OptimalControl/ODE.h:128-187, pybind/OptimalControl/GenericODESBuildPart1-6.cpp) and walks its
expression tree at every evaluation.  Here the expression graph of an :class:`~asset_asrl_amd.ode.ODEBase` is
differentiated symbolically, printed as a HIP functor (``vf/codegen.py``) and compiled for gfx950 together with the
defect kernels of the requested transcription -- one small shared object per (ODE, mode, control mode), built with
``hipcc`` on first use, cached in-tree next to the other generated sources, and handed to the library through
``asset_hip_load_plugin`` (include/asset_hip.h).  There is no interpreter and no CPU fallback: without ``hipcc`` the
call fails.
"""
from __future__ import annotations

import ctypes as C
import hashlib
import os
import re
import subprocess

from . import _lib, build
from .vf.codegen import emit_hip_functor, saved_nodes

JIT_DIR = os.path.join(build.GEN, "jit")
_MODE_CS = {"LGL3": 2, "LGL5": 3, "LGL7": 4}
_loaded: set = set()


def _ident(s: str) -> str:
    return re.sub(r"[^0-9a-zA-Z_]", "_", s)


def device_name(ode) -> str:
    """Name the ODE is registered under on the device: library ODEs keep theirs, user ODEs get a content hash."""
    from .ode import ODE_LIBRARY
    if ode.ode_name in ODE_LIBRARY and type(ode) is ODE_LIBRARY[ode.ode_name]:
        return ode.ode_name
    cached = getattr(ode, "_device_name", None)
    if cached:
        return cached
    body = emit_hip_functor(ode.derivatives(), "OdeUser")       # the name does not enter the hash, the maths does
    body = "\n".join(ln for ln in body.splitlines() if "name()" not in ln and not ln.startswith("// generated"))
    h = hashlib.sha256(body.encode()).hexdigest()[:10]
    ode._device_name = f"{_ident(ode.ode_name)}_{h}"
    return ode._device_name


def ensure_kernel(ode, mode: str, blocked: bool) -> str:
    """Make sure device code for (ode, mode, blocked) is registered; returns the device-side ODE name."""
    name = device_name(ode)
    blocked = bool(blocked) and ode.UVars() > 0
    mode_id = _lib.MODES[mode]
    if _lib.has_kernel(name, mode_id, blocked):
        return name
    from .ode import ODE_LIBRARY
    if name in ODE_LIBRARY:
        raise _lib.AssetHipError(f"library ODE '{name}' has no {mode}{' BlockConstant' if blocked else ''} kernel: its "
                                 "working set exceeds one CU's LDS")
    xv, uv, pv = ode.XVars(), ode.UVars(), ode.PVars()
    d = ode.derivatives()
    d.name = name
    st = d.stats()
    if mode == "Trapezoidal":
        G = build.pick_trap_group(xv, uv, pv, blocked, len(saved_nodes(d)), st["nnz_J"], st["nnz_H_lower"])
        reg = f"ASSET_REGISTER_TRAP({{S}}, {int(blocked)}, {G})"
    else:
        cs = _MODE_CS[mode]
        G = build.pick_group(xv, uv, pv, cs, blocked, len(saved_nodes(d)), st["nnz_J"], st["nnz_H_lower"])
        reg = f"ASSET_REGISTER_LGL({{S}}, {cs}, {int(blocked)}, {G})"
    if G == 0:
        raise _lib.AssetHipError(f"ODE '{ode.ode_name}' ({xv},{uv},{pv}) with {mode}: per-segment working set exceeds "
                                 "one CU's LDS; no kernel can be instantiated")
    sname = "Ode_" + _ident(name)
    hdr = ("#pragma once\n#include <math.h>\n#include \"" + os.path.join(build.CSRC, "asset_math.h") + "\"\n"
           + emit_hip_functor(d, sname))
    tag = f"{mode.lower()}_{int(blocked)}"
    src = (f'#include "ode.h"\n#include "{os.path.join(build.CSRC, "registry.h")}"\n'
           + reg.replace("{S}", sname) + "\nASSET_PLUGIN_EXPORT()\n")
    return _build_and_load(name, "ode.h", hdr, tag, src, mode_id, blocked, f"user ODE '{ode.ode_name}'")


def _build_and_load(name, hdr_name, hdr, tag, src, mode_id, blocked, what) -> str:
    """Compile one plugin translation unit (cached by content) and register it; returns ``name``."""
    deps = [os.path.join(build.CSRC, f) for f in sorted(os.listdir(build.CSRC)) if f.endswith(".h")]
    key = hashlib.sha256((hdr + src + " ".join(build.FLAGS)).encode()
                         + b"".join(open(p, "rb").read() for p in deps)).hexdigest()[:16]
    wd = os.path.join(JIT_DIR, name)
    os.makedirs(wd, exist_ok=True)
    so = os.path.join(wd, f"plugin_{tag}_{key}.so")
    if not os.path.exists(so):
        build._write_if_changed(os.path.join(wd, hdr_name), hdr)
        tu = os.path.join(wd, f"tu_{tag}.hip")
        build._write_if_changed(tu, src)
        if not os.path.exists(build.HIPCC):
            raise _lib.AssetHipError(f"{build.HIPCC} not found: {what} needs the HIP compiler at run time")
        cmd = [build.HIPCC] + build.FLAGS + ["-DASSET_PLUGIN", "-shared", "-I", os.path.join(build.HERE, "..", "include"),
                                             tu, "-o", so + ".tmp"]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise _lib.AssetHipError(f"hipcc failed for {what}:\n{r.stderr[-3000:]}")
        os.replace(so + ".tmp", so)
        for f in os.listdir(wd):            # plugins of this unit built against older sources
            if f.startswith(f"plugin_{tag}_") and f.endswith(".so") and os.path.join(wd, f) != so:
                os.remove(os.path.join(wd, f))
    if so not in _loaded:
        rc = _lib.lib().asset_hip_load_plugin(so.encode())
        if rc < 0:
            _lib.check(rc, "asset_hip_load_plugin")
        _loaded.add(so)
    if not _lib.has_kernel(name, mode_id, blocked):
        raise _lib.AssetHipError(f"plugin {so} did not register ({name}, mode {mode_id}, blocked={blocked})")
    return name


def ensure_function(func, name: str) -> str:
    """Device code for a plain vector function batched over applications (transcription id 0, csrc/func_kernels.h):
    ``DefectEvaluator(ensure_function(f, "my_con"), "Function", False, vindex, cindex, ...)`` then evaluates it like a
    defect -- FX / AGX blocks and the KKT block (Jacobian + lower-triangle adjoint Hessian) of every application."""
    from .vf.codegen import differentiate_function
    d = differentiate_function(name, func)
    body = emit_hip_functor(d, "FnUser")
    body = "\n".join(ln for ln in body.splitlines() if "name()" not in ln and not ln.startswith("// generated"))
    dev = f"{_ident(name)}_{hashlib.sha256(body.encode()).hexdigest()[:10]}"
    if _lib.has_kernel(dev, _lib.FUNCTION, False):
        return dev
    d.name = dev
    sname = "Fn_" + _ident(dev)
    hdr = ("#pragma once\n#include <math.h>\n#include \"" + os.path.join(build.CSRC, "asset_math.h") + "\"\n"
           + emit_hip_functor(d, sname))
    src = (f'#include "fn.h"\n#include "{os.path.join(build.CSRC, "registry.h")}"\n'
           f"ASSET_REGISTER_FUNC({sname})\nASSET_PLUGIN_EXPORT()\n")
    return _build_and_load(dev, "fn.h", hdr, "function_0", src, _lib.FUNCTION, False, f"function '{name}'")
