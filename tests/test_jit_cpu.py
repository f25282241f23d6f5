"""Run-time compilation of a user-defined ODE (asset_asrl_amd/jit.py): code generation and compilation for gfx950 work
without a GPU -- in process through hiprtc (the default route: compile and cache; loading the module needs a device, so
registration is checked in test_gpu_jit.py), and through the hipcc driver (ASSET_HIP_JIT=hipcc: a shared object, which
registers on a CPU box too).  The numerical checks live in test_gpu_jit.py."""
import glob
import os

import pytest

from asset_asrl_amd import _lib, jit, vf
from asset_asrl_amd.ode import ODEArguments, ODEBase, ShuttleReentry
from helpers import make_vanderpol


def test_user_ode_is_compiled_in_process_and_cached(monkeypatch):
    monkeypatch.delenv("ASSET_HIP_JIT", raising=False)
    ode = make_vanderpol()
    name = jit.device_name(ode)
    assert name.startswith("vanderpol_") and name not in ("vanderpol",)
    assert jit.ensure_kernel(ode, "LGL5", False, compile_only=True) == name
    mods = glob.glob(os.path.join(jit.JIT_DIR, name, "module_lgl5_0_*.rtc"))
    assert len(mods) == 1                                            # one cache file per (ODE, mode, control mode), keyed by content
    head = open(mods[0], "rb").read(4096).split(b"\n")
    assert head[0] == b"ASSET-HIP-RTC-1" and int(head[1]) >= 10      # code object + the lowered names of the kernel slots
    assert any(b"lgl_defect_kernel" in ln and name.encode() in ln for ln in head[2:2 + int(head[1])])
    stamp = os.path.getmtime(mods[0])
    assert jit.ensure_kernel(make_vanderpol(), "LGL5", False, compile_only=True) == name   # same maths -> same name,
    assert os.path.getmtime(mods[0]) == stamp                                                # nothing rebuilt
    if not os.path.exists("/dev/kfd"):                               # no device: loading the module must fail loudly
        with pytest.raises(_lib.AssetHipError):
            jit.ensure_kernel(ode, "LGL5", False)


def test_module_cache_key_does_not_depend_on_where_the_tree_lies(monkeypatch, tmp_path):
    """The GPU box runs a snapshot of this tree under another root.  The generated translation units carry absolute include paths; with
    them in the cache key every module build() had compiled here was compiled again there (round 6 found it).  The same tree seen through
    a symbolic link elsewhere must find the cached module: no new file, nothing rebuilt."""
    import subprocess
    import sys
    monkeypatch.delenv("ASSET_HIP_JIT", raising=False)
    ode = make_vanderpol()
    name = jit.ensure_kernel(ode, "LGL3", False, compile_only=True)
    mods = sorted(glob.glob(os.path.join(jit.JIT_DIR, name, "module_lgl3_0_*.rtc")))
    assert len(mods) == 1
    stamp = os.path.getmtime(mods[0])
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    link = tmp_path / "elsewhere"
    os.symlink(root, link)
    code = (f"import os, sys; L = {str(link)!r}; sys.path.insert(0, L); sys.path.insert(0, os.path.join(L, 'tests'));"
            "from asset_asrl_amd import jit, build; from helpers import make_vanderpol;"
            "assert build.CSRC.startswith(L), build.CSRC;"                       # (abspath keeps the link: another root, as on the GPU box)
            "print(jit.ensure_kernel(make_vanderpol(), 'LGL3', False, compile_only=True))")
    env = {k: v for k, v in os.environ.items() if k != "PYTHONPATH"}
    r = subprocess.run([sys.executable, "-c", code], cwd=str(link), env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    assert r.stdout.strip().splitlines()[-1] == name
    assert sorted(glob.glob(os.path.join(jit.JIT_DIR, name, "module_lgl3_0_*.rtc"))) == mods and os.path.getmtime(mods[0]) == stamp


def test_compile_errors_come_back_with_the_compilers_log(monkeypatch, tmp_path):
    import ctypes as C
    opts = jit.rtc_options()
    copts = (C.c_char_p * len(opts))(*[o.encode() for o in opts])
    rc = _lib.lib().asset_hip_jit_compile(b"struct Broken { int x }\n", b"Broken", 2, 0, 0, 0, copts, len(opts),
                                          str(tmp_path / "broken.rtc").encode())
    assert rc == -5 and b"error" in _lib.lib().asset_hip_last_error()
    assert not (tmp_path / "broken.rtc").exists()


def test_user_ode_through_the_compiler_driver_registers(monkeypatch):
    monkeypatch.setenv("ASSET_HIP_JIT", "hipcc")
    ode = make_vanderpol()
    name = jit.device_name(ode)
    assert jit.ensure_kernel(ode, "LGL5", False) == name          # compiles (or finds the cached plugin) and registers
    assert _lib.has_kernel(name, _lib.LGL5, False)
    assert not _lib.has_kernel(name, _lib.LGL7, False)              # only what was asked for
    assert _lib.ode_sizes(name) == (2, 1, 1)
    assert name in _lib.ode_names()
    assert jit.ensure_kernel(ode, "Trapezoidal", True) == name     # BlockConstant control form
    assert _lib.has_kernel(name, _lib.TRAPEZOIDAL, True)
    assert jit.ensure_kernel(make_vanderpol(), "LGL5", False) == name   # same maths -> same device name, nothing rebuilt


def test_device_name_follows_the_dynamics_not_the_class_name():
    a = ODEArguments(2, 1, 1)
    x0, x1 = a.XVec().tolist()
    other = ODEBase(vf.stack([x1, 2.0 * a.PVar(0) * x1 - x0 + a.UVar(0)]), 2, 1, 1, name="vanderpol")
    assert jit.device_name(other) != jit.device_name(make_vanderpol())


def test_library_odes_keep_their_names_and_are_not_recompiled():
    assert jit.device_name(ShuttleReentry()) == "reentry"
    assert jit.ensure_kernel(ShuttleReentry(), "LGL7", False) == "reentry"


def test_load_plugin_rejects_a_non_plugin():
    rc = _lib.lib().asset_hip_load_plugin(_lib.LIB_PATH.encode())
    assert rc < 0 and b"asset_hip_plugin_entries" in _lib.lib().asset_hip_last_error()


def test_plain_function_gets_device_code(monkeypatch):
    """Transcription id 0: any DSL vector function batched over applications (csrc/func_kernels.h)."""
    from asset_asrl_amd.pathfuncs import LGLMeshSpacing, SingleMeshSpacing
    monkeypatch.delenv("ASSET_HIP_JIT", raising=False)
    dev = jit.ensure_function(LGLMeshSpacing(3), "lglmeshspacing3", compile_only=True)      # in process, cached
    assert glob.glob(os.path.join(jit.JIT_DIR, dev, "module_function_0_*.rtc"))
    monkeypatch.setenv("ASSET_HIP_JIT", "hipcc")                                             # driver route: registers here
    name = jit.ensure_function(LGLMeshSpacing(3), "lglmeshspacing3")
    assert name == dev
    assert _lib.has_kernel(name, _lib.FUNCTION, False) and not _lib.has_kernel(name, _lib.LGL3, False)
    assert jit.ensure_function(LGLMeshSpacing(3), "lglmeshspacing3") == name
    assert jit.ensure_function(SingleMeshSpacing(0.25), "single_spacing") != name
    with pytest.raises(ValueError):
        LGLMeshSpacing(2)


def test_lgl_integral_quadrature_is_exact_for_cubics(monkeypatch):
    monkeypatch.delenv("ASSET_HIP_JIT", raising=False)
    """LGLIntegral (LGLIntegrals.h:9-52): h * sum_i w_i integrand(x_i); the LGL7 reduced weights integrate
    t^2 + t^3 exactly.  Its device code is generated like any other function's."""
    import numpy as np
    from asset_asrl_amd.pathfuncs import LGLIntegral
    g = vf.Arguments(2)
    F = LGLIntegral(g.coeff(1) * g.coeff(1) + g.coeff(0), 4, 2)
    assert (F.IRows(), F.ORows()) == (12, 1)
    t0, h = 0.3, 1.7
    z = []
    for c in (0.0, 2.65575603264643e-1, 7.34424396735357e-1, 1.0):
        t = t0 + c * h
        z += [t ** 3, t, t]
    prim = lambda t: t ** 3 / 3 + t ** 4 / 4
    assert abs(F.compute(np.array(z))[0] - (prim(t0 + h) - prim(t0))) < 1e-12
    name = jit.ensure_function(F, "lglintegral_test", compile_only=True)     # (registration: test_gpu_function.py)
    assert glob.glob(os.path.join(jit.JIT_DIR, name, "module_function_0_*.rtc"))


def test_control_spline_vanishes_on_a_smooth_control():
    """LGLControlSpline (LGLControlSplines.h:64-108): derivative continuity of the control polynomial across two
    segments; a control that is a single cubic (LGL7) / quadratic (LGL5) over both segments satisfies it."""
    import numpy as np
    from asset_asrl_amd.pathfuncs import LGLControlSpline
    tc = np.array([0.0, 2.65575603264643e-1, 7.34424396735357e-1, 1.0])
    t0, h0, h1 = 0.2, 1.3, 0.7
    ts = np.concatenate([t0 + tc * h0, (t0 + h0) + tc[1:] * h1])
    z = np.concatenate([[t, 1 + 2 * t - 0.5 * t ** 2 + 0.3 * t ** 3, t ** 3 - t] for t in ts])
    F = LGLControlSpline(4, 2)
    assert (F.IRows(), F.ORows()) == (21, 4)
    assert np.abs(F.compute(z)).max() < 1e-10                      # 15-digit weight literals
    zbad = z.copy()
    zbad[-1] += 0.1                                                # a kink in the last node's control
    assert np.abs(F.compute(zbad)).max() > 1e-2
    ts3 = np.array([0.0, 0.5, 1.0, 1.4, 1.8])
    assert np.abs(LGLControlSpline(3, 1).compute(np.concatenate([[t, 2 - t + 0.7 * t * t] for t in ts3]))).max() < 1e-13
    with pytest.raises(ValueError):
        LGLControlSpline(2, 1)
