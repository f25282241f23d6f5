"""Run-time device code for user-defined ODEs and functions.

The reference takes any VectorFunction as ODE right-hand side (``oc.ode_x_u_p.ode(vf, Xv, Uv, Pv)``,
/root/reference/src/OptimalControl/ODE.h:128-187, pybind/OptimalControl/GenericODESBuildPart1-6.cpp) and walks its
expression tree at every evaluation.  Here the expression graph of an :class:`~asset_asrl_amd.ode.ODEBase` is
differentiated symbolically, printed as a HIP functor (``vf/codegen.py``) and compiled for gfx950 together with the
defect kernels of the requested transcription -- one module per (ODE, mode, control mode), compiled IN PROCESS with
hiprtc on first use (``asset_hip_jit_plugin``, include/asset_hip.h: the library calls the compiler library itself, no
compiler driver is started), cached in-tree next to the other generated sources as the code object plus the lowered
kernel names.  ``ASSET_HIP_JIT=hipcc`` selects the older route instead (a shared object built by the ``hipcc`` driver,
``asset_hip_load_plugin``); it is a switch for debugging, not a fallback: whichever route is selected, a failure raises.
There is no interpreter and no CPU fallback.
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
_PARTITION = None      # (worker, workers): compile-only builds spread over processes -- a module is compiled by the worker its key names


def set_partition(worker=None, workers=None):
    """compile_only_mode builds only: this process compiles the cache misses whose content key is `worker` modulo `workers` and
    skips the others (None: all) -- __graft_entry__.build() runs its pre-build once per worker, then once more serially."""
    global _PARTITION
    _PARTITION = None if worker is None else (int(worker), int(workers))


_TOUCHED: set = set()     # cache files this process compiled or found (prune_unused drops the rest)
_FUNCTORS: dict = {}        # device name of a plain function -> (functor struct name, its derivatives)
# The build step (__graft_entry__.build) runs where there is no device: it compiles and caches the modules the test suite
# will ask for and registers nothing.  Set through compile_only_mode().
_COMPILE_ONLY = False


def compile_only_mode(on: bool = True):
    global _COMPILE_ONLY
    _COMPILE_ONLY = bool(on)


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


def ensure_kernel(ode, mode: str, blocked: bool, compile_only=None) -> str:
    """Make sure device code for (ode, mode, blocked) is registered; returns the device-side ODE name.
    compile_only: compile and cache without registering (the build step on a machine without a device)."""
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
    cs_id = 1 if mode == "Trapezoidal" else _MODE_CS[mode]
    return _build_and_load(name, "ode.h", hdr, tag, reg.replace("{S}", sname), mode_id, blocked,
                           f"user ODE '{ode.ode_name}'", rtc=(sname, 1, cs_id, G, f"ASSET_RTC_LGL({sname}, {cs_id}, {int(blocked)}, {G})"),
                           compile_only=compile_only)


def jit_route() -> str:
    r = os.environ.get("ASSET_HIP_JIT", "hiprtc")
    if r not in ("hiprtc", "hipcc"):
        raise _lib.AssetHipError(f"ASSET_HIP_JIT={r!r}: expected 'hiprtc' or 'hipcc'")
    return r


def rtc_options():
    """hiprtc options: the flags of the static build plus where the compiler library finds its own headers (it has the HIP
    device headers built in, but neither the resource directory of its clang nor <hip/hip_runtime.h> as a file)."""
    import glob
    rocm = os.path.dirname(os.path.dirname(os.path.realpath(build.HIPCC)))
    res = sorted(glob.glob(os.path.join(rocm, "lib", "llvm", "lib", "clang", "*", "include")))
    opts = [f for f in build.FLAGS if f != "-fPIC"] + ["-Wno-cuda-compat", "-Wno-pragma-once-outside-header"]
    opts += ["-I" + res[-1]] if res else []
    opts += ["-I" + os.path.join(rocm, "include")]
    return opts


def _build_and_load(name, hdr_name, hdr, tag, reg_line, mode_id, blocked, what, rtc, compile_only=None) -> str:
    """Compile one generated translation unit (cached by content) and register it; returns ``name``.
    rtc = (functor, kind, mode id, segments per group, the ASSET_RTC_* line)."""
    if compile_only is None:
        compile_only = _COMPILE_ONLY
    deps = [os.path.join(build.CSRC, f) for f in sorted(os.listdir(build.CSRC)) if f.endswith(".h")]
    route = jit_route()
    wd = os.path.join(JIT_DIR, name)
    os.makedirs(wd, exist_ok=True)

    def key_of(src, flags):
        # (the key names WHAT is compiled, not WHERE the tree lies: the generated text carries absolute include paths, and the GPU box
        #  runs a snapshot of this tree under another root -- with the paths in the key every module pre-built here was compiled again
        #  there, round after round)
        root = os.path.dirname(build.HERE)
        text = (hdr + src + " ".join(flags)).replace(root, "$ROOT")
        return hashlib.sha256(text.encode() + b"".join(open(p, "rb").read() for p in deps)).hexdigest()[:16]

    def drop_older(prefix, suffix, keep):
        for f in os.listdir(wd):            # builds of this unit against older sources
            if f.startswith(prefix) and f.endswith(suffix) and os.path.join(wd, f) != keep:
                os.remove(os.path.join(wd, f))

    if route == "hiprtc":
        functor, kind, cs_id, G, rtc_line = rtc
        opts = rtc_options()
        src = hdr + f'\n#include "{os.path.join(build.CSRC, "rtc_device.h")}"\n{rtc_line}\n'
        mod = os.path.join(wd, f"module_{tag}_{key_of(src, opts)}.rtc")
        _TOUCHED.add(mod)
        copts = (C.c_char_p * len(opts))(*[o.encode() for o in opts])
        args = (src.encode(), functor.encode(), kind, cs_id, int(blocked), G, copts, len(opts), mod.encode())
        if compile_only:                    # (the build step, which has no device: compile and cache)
            if not os.path.exists(mod):
                if _PARTITION is not None and int(key_of(src, opts), 16) % _PARTITION[1] != _PARTITION[0]:
                    return name             # (another worker of a parallel pre-build compiles this one)
                _lib.check(_lib.lib().asset_hip_jit_compile(*args), f"asset_hip_jit_compile ({what})")
                drop_older(f"module_{tag}_", ".rtc", mod)
            return name
        fresh = not os.path.exists(mod)
        _lib.check(_lib.lib().asset_hip_jit_plugin(name.encode(), *args), f"asset_hip_jit_plugin ({what})")
        if fresh:
            drop_older(f"module_{tag}_", ".rtc", mod)
    else:
        src = (f'#include "{hdr_name}"\n#include "{os.path.join(build.CSRC, "registry.h")}"\n' + reg_line
               + "\nASSET_PLUGIN_EXPORT()\n")
        so = os.path.join(wd, f"plugin_{tag}_{key_of(src, build.FLAGS)}.so")
        _TOUCHED.add(so)
        if not os.path.exists(so):
            if compile_only and _PARTITION is not None and int(key_of(src, build.FLAGS), 16) % _PARTITION[1] != _PARTITION[0]:
                return name                 # (another worker of a parallel pre-build compiles this one)
            build._write_if_changed(os.path.join(wd, hdr_name), hdr)
            tu = os.path.join(wd, f"tu_{tag}.hip")
            build._write_if_changed(tu, src)
            if not os.path.exists(build.HIPCC):
                raise _lib.AssetHipError(f"{build.HIPCC} not found: {what} needs the HIP compiler driver (ASSET_HIP_JIT=hipcc)")
            cmd = [build.HIPCC] + build.FLAGS + ["-DASSET_PLUGIN", "-shared", "-I", os.path.join(build.HERE, "..", "include"),
                                                 tu, "-o", f"{so}.{os.getpid()}.tmp"]     # (a name of this process's own: two
            r = subprocess.run(cmd, capture_output=True, text=True)                                          #  builders never share a half-written file)
            if r.returncode != 0:
                raise _lib.AssetHipError(f"hipcc failed for {what}:\n{r.stderr[-3000:]}")
            os.replace(f"{so}.{os.getpid()}.tmp", so)
            drop_older(f"plugin_{tag}_", ".so", so)
        if compile_only:
            return name
        if so not in _loaded:
            rc = _lib.lib().asset_hip_load_plugin(so.encode())
            if rc < 0:
                _lib.check(rc, "asset_hip_load_plugin")
            _loaded.add(so)
    if not _lib.has_kernel(name, mode_id, blocked):
        raise _lib.AssetHipError(f"{what}: the compiled module did not register ({name}, mode {mode_id}, blocked={blocked})")
    return name


def ensure_function(func, name: str, compile_only=None) -> str:
    """Device code for a plain vector function batched over applications (transcription id 0, csrc/func_kernels.h):
    ``DefectEvaluator(ensure_function(f, "my_con"), "Function", False, vindex, cindex, ...)`` then evaluates it like a
    defect -- FX / AGX blocks and the KKT block (Jacobian + lower-triangle adjoint Hessian) of every application."""
    from .vf.codegen import differentiate_function
    d = differentiate_function(name, func)
    body = emit_hip_functor(d, "FnUser")
    body = "\n".join(ln for ln in body.splitlines() if "name()" not in ln and not ln.startswith("// generated"))
    dev = f"{_ident(name)}_{hashlib.sha256(body.encode()).hexdigest()[:10]}"
    d.name = dev
    sname = "Fn_" + _ident(dev)
    _FUNCTORS[dev] = (sname, d)                 # (a bundle is assembled from the functors of its members: ensure_bundle)
    if _lib.has_kernel(dev, _lib.FUNCTION, False):
        return dev
    hdr = ("#pragma once\n#include <math.h>\n#include \"" + os.path.join(build.CSRC, "asset_math.h") + "\"\n"
           + emit_hip_functor(d, sname))
    return _build_and_load(dev, "fn.h", hdr, "function_0", f"ASSET_REGISTER_FUNC({sname})", _lib.FUNCTION, False,
                           f"function '{name}'", rtc=(sname, 2, 0, 0, f"ASSET_RTC_FUNC({sname})"), compile_only=compile_only)


def ensure_bundle(dev_names, compile_only=None) -> str:
    """One module that evaluates the plain functions `dev_names` (device names returned by ensure_function, in this order)
    in a single launch (csrc/func_kernels.h: func_bundle_kernel; include/asset_hip.h: asset_hip_bundle_*).  hiprtc only."""
    if not 1 <= len(dev_names) <= 8:
        raise ValueError("a bundle holds 1..8 functions")
    if jit_route() != "hiprtc":
        raise _lib.AssetHipError("function bundles are compiled in process (unset ASSET_HIP_JIT=hipcc)")
    missing = [n for n in dev_names if n not in _FUNCTORS]
    if missing:
        raise _lib.AssetHipError(f"ensure_bundle: {missing} were not produced by ensure_function in this process")
    snames = [_FUNCTORS[n][0] for n in dev_names]
    name = "bundle_" + hashlib.sha256(",".join(dev_names).encode()).hexdigest()[:12]
    if _lib.has_kernel(name, _lib.FUNCTION, False):
        return name
    body, seen = [], set()
    for n in dev_names:
        sname, d = _FUNCTORS[n]
        if sname not in seen:
            seen.add(sname)
            body.append(emit_hip_functor(d, sname))
    hdr = ("#include <math.h>\n#include \"" + os.path.join(build.CSRC, "asset_math.h") + "\"\n" + "\n".join(body))
    flist = ", ".join(snames)
    return _build_and_load(name, "bundle.h", hdr, "bundle_0", "", _lib.FUNCTION, False, f"bundle of {len(dev_names)} functions",
                           rtc=(flist, 3, 0, 0, f"ASSET_RTC_BUNDLE({flist})"), compile_only=compile_only)


def prune_unused(verbose: bool = False, everything: bool = False) -> int:
    """Remove stale code objects from the in-tree module cache: the cache is keyed by content, so builds against older sources
    are never loaded again, but they travel with every snapshot of the tree.  By default only inside the directories of units THIS
    process asked for (older ``module_*`` / ``plugin_*`` builds of the same unit); ``everything=True`` also removes the directories of
    units this process never asked for -- cached modules of ODEs some other script compiled.  ``__graft_entry__.build()`` passes it,
    after asking for everything the GPU tests use.  Returns the number of files removed."""
    import shutil
    removed = 0
    if not os.path.isdir(JIT_DIR):
        return 0
    for d in sorted(os.listdir(JIT_DIR)):
        wd = os.path.join(JIT_DIR, d)
        if not os.path.isdir(wd):
            continue
        files = [os.path.join(wd, f) for f in os.listdir(wd)]
        keep = [f for f in files if f in _TOUCHED]
        if not keep:
            if everything:
                removed += len(files)
                shutil.rmtree(wd, ignore_errors=True)
            continue
        for f in files:
            if f not in _TOUCHED and (f.endswith(".rtc") or (f.endswith(".so") and os.path.basename(f).startswith("plugin_"))):
                os.remove(f)
                removed += 1
    if verbose:
        print(f"[asset_hip] module cache: {removed} stale files removed, {len(_TOUCHED)} in use", flush=True)
    return removed
