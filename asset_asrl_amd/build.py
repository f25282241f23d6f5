"""Build libasset_hip.so: generate the ODE functors, compile every kernel TU for gfx950, link.

``python -m asset_asrl_amd.build`` (or ``__graft_entry__.build()``).  hipcc cross-compiles without a GPU.
The shared object is written in-tree (asset_asrl_amd/libasset_hip.so) so it travels with the snapshot.
"""
from __future__ import annotations

import hashlib
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
GEN = os.path.join(CSRC, "gen")
OBJ = os.path.join(CSRC, "obj")
LIB = os.path.join(HERE, "libasset_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
ARCH = "gfx950"
FLAGS = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value"]
# Scheduler by translation unit.  The segment loops of the resident kernel are single basic blocks bound by LDS / matrix-instruction
# latency; the scheduler that goes for instruction-level parallelism (instead of the lowest register pressure) hides more of it in
# the shapes with registers to spare -- measured per unit, all 1 000-step bench lines: Reentry-LGL7 29.9 -> 29.3 us (5 000 segments
# 21.5 -> 21.1), Trapezoidal 11.4 -> 11.3 / 17.1 -> 16.5 -- and costs where there are none: TwoBody-LGL5 33.5 -> 34.3, the row-wise
# stage of the 32-state ODE 1.19 -> 1.27 ms; Betts within 0.4 %.  So: the Reentry units and every Trapezoidal unit.
MAX_ILP = ["-mllvm", "-amdgpu-sched-strategy=max-ilp"]


def tu_flags(src):
    b = os.path.basename(src)
    if os.environ.get("ASSET_HIP_NO_MAX_ILP") or not b.startswith("tu_"):
        return []
    return MAX_ILP if (b.startswith("tu_reentry_") or "_trap_" in b) else []

LDS_BUDGET = 160 * 1024
LDS_TARGET = 40 * 1024          # keep >= 4 single-wave workgroups per CU when the sizes allow it


def dims(xv, uv, pv, cs, blocked, nsave=0, nzj=None, nzh=None, trap=False):
    """Mirror of ``Dims<>`` in csrc/defect_dims.h (sizes + LDS plan).  ``nzj`` / ``nzh`` are the structural
    non-zero counts of the ODE Jacobian / packed-lower Hessian (dense when omitted)."""
    n = xv
    m, p = (0, uv + pv) if blocked else (uv, pv)
    K = cs - 1
    q = n + 1 + m
    N = q + p
    IR, OR = cs * q + p, K * n
    NKKT = IR * (IR + 1) // 2 + OR * IR
    NH = N * (N + 1) // 2
    nzj = n * N if nzj is None else nzj
    nzh = NH if nzh is None else nzh
    IRP, ORP = (IR + 15) // 16 * 16, (OR + 15) // 16 * 16
    NP = (N + 3) // 4 * 4
    WSLOTD = IR + OR + cs * n + cs * nzj + cs * N + cs * nzh + K * n + K * nzj + K * N + K * nzh
    WSLOT = WSLOTD + cs * nsave
    LDM = K * NP + 1
    SCRATCH = max(K * n * IRP, IRP * LDM) + K * (NP - n) * IRP + ORP * (IRP + 4) + 4 * IRP + 2
    wide = (70 + WSLOTD + SCRATCH) * 8 > LDS_BUDGET or (IR >= 64 and not trap)
    # KKT block layout (Dims::KL): wide shapes keep the reference's order, the others write J | H with padded regions
    KSTRIDE = NKKT if wide else (OR * IR + 15) // 16 * 16 + (IR * (IR + 1) // 2 + 15) // 16 * 16
    if wide:     # Dims::WIDE (csrc/defect_wide.h): DI resident, M in registers, no DC tile
        SCRATCH = K * n * IRP + K * (N - n) * IRP + (ORP + cs * n + 2) + 4 * IRP + 2 \
            + (n * N + 3) // 4 + (NH + 3) // 4
    DENSE = WSLOTD + SCRATCH
    STG_LD = (nzj + nzh) | 1
    budget = 64 * 1024
    staged = 16 * STG_LD * 8 <= budget
    LC = 64 if (not staged or 64 * STG_LD * 8 <= budget) else (32 if 32 * STG_LD * 8 <= budget else 16)
    BODY = max(LC * STG_LD if staged else 0, DENSE)

    def lds_bytes(G=0):
        return (70 + BODY) * 8
    return dict(n=n, m=m, p=p, q=q, N=N, IR=IR, OR=OR, NKKT=NKKT, KSTRIDE=KSTRIDE, KL=0 if wide else 1, SLOT=WSLOT, LC=LC, lds_bytes=lds_bytes,
                lds_bytes_ode=(70 + (LC * STG_LD if staged else 0)) * 8, lds_bytes_dense=(70 + DENSE) * 8)


def pick_group(xv, uv, pv, cs, blocked, nsave=0, nzj=None, nzh=None, trap=False):
    d = dims(xv, uv, pv, cs, blocked, nsave, nzj, nzh, trap)
    return 64 // cs if d["lds_bytes"]() <= LDS_BUDGET else 0


def pick_trap_group(xv, uv, pv, blocked, nsave=0, nzj=None, nzh=None):
    """Trapezoidal runs through the LGL kernels as a two-node scheme (csrc/defect_dims.h, Dims::TRAP)."""
    return pick_group(xv, uv, pv, 2, blocked, nsave, nzj, nzh, trap=True)


def _struct_name(name: str) -> str:
    return "Ode" + "".join(w.capitalize() for w in name.split("_"))


def generate(verbose=True):
    """Write csrc/gen/ode_<name>.h and csrc/gen/tu_<name>.hip for every library ODE."""
    from .ode import ODE_LIBRARY
    from .vf.codegen import emit_hip_functor, saved_nodes
    os.makedirs(GEN, exist_ok=True)
    tus = []
    for name, cls in ODE_LIBRARY.items():
        ode = cls()
        sn = _struct_name(name)
        hdr = "#pragma once\n#include <math.h>\n#include \"../asset_math.h\"\n" + emit_hip_functor(ode.derivatives(), sn)
        _write_if_changed(os.path.join(GEN, f"ode_{name}.h"), hdr)
        xv, uv, pv = ode.XVars(), ode.UVars(), ode.PVars()
        # one translation unit per (transcription, control mode): they compile in parallel
        units = []
        for cs in (2, 3, 4):
            for blocked in ((0, 1) if uv > 0 else (0,)):
                st = ode.derivatives().stats()
                G = pick_group(xv, uv, pv, cs, bool(blocked), len(saved_nodes(ode.derivatives())), st["nnz_J"],
                               st["nnz_H_lower"])
                if G == 0:
                    continue  # working set exceeds one CU's LDS -- not instantiated (asset_hip_has_kernel says so)
                units.append((f"lgl{cs}_{blocked}", f"ASSET_REGISTER_LGL({sn}, {cs}, {blocked}, {G})"))
        for blocked in ((0, 1) if uv > 0 else (0,)):
            st = ode.derivatives().stats()
            G = pick_trap_group(xv, uv, pv, bool(blocked), len(saved_nodes(ode.derivatives())), st["nnz_J"], st["nnz_H_lower"])
            if G:
                units.append((f"trap_{blocked}", f"ASSET_REGISTER_TRAP({sn}, {blocked}, {G})"))
        for tag, line in units:
            tu = os.path.join(GEN, f"tu_{name}_{tag}.hip")
            _write_if_changed(tu, f'#include "ode_{name}.h"\n#include "../registry.h"\n{line}\n')
            tus.append(tu)
        if verbose:
            print(f"[asset_hip] generated {name}: {ode.derivatives().stats()}", flush=True)
    return tus


def _write_if_changed(path, text):
    if os.path.exists(path) and open(path).read() == text:
        return
    with open(path, "w") as f:
        f.write(text)


def _digest(paths, extra=[]):
    h = hashlib.sha256()
    for p in sorted(paths):
        h.update(open(p, "rb").read())
    h.update(" ".join(FLAGS + extra).encode())
    return h.hexdigest()


def _compile(src):
    os.makedirs(OBJ, exist_ok=True)
    obj = os.path.join(OBJ, os.path.basename(src).rsplit(".", 1)[0] + ".o")
    deps = [src] + [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    for f in os.listdir(GEN):
        if f.startswith("ode_") and f.endswith(".h") and f'#include "{f}"' in open(src).read():
            deps.append(os.path.join(GEN, f))
    deps.append(os.path.join(HERE, "..", "include", "asset_hip.h"))
    stamp = obj + ".sha"
    dg = _digest(deps, tu_flags(src))
    if os.path.exists(obj) and os.path.exists(stamp) and open(stamp).read() == dg:
        return obj, False
    cmd = [HIPCC] + FLAGS + tu_flags(src) + ["-c", src, "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed for {src}:\n{r.stderr[-4000:]}")
    with open(stamp, "w") as f:
        f.write(dg)
    return obj, True


def build(verbose=True, jobs=None) -> str:
    tus = generate(verbose)
    keep = {os.path.basename(t) for t in tus}
    for f in os.listdir(GEN):       # drop translation units of an older layout
        if f.startswith("tu_") and f not in keep:
            os.remove(os.path.join(GEN, f))
    if os.path.isdir(OBJ):
        for f in os.listdir(OBJ):
            if f.startswith("tu_") and f.split(".")[0] + ".hip" not in keep:
                os.remove(os.path.join(OBJ, f))
    srcs = tus + [os.path.join(CSRC, "capi.hip"), os.path.join(CSRC, "capi_sharded.hip")]
    jobs = jobs or min(len(srcs), max(1, (os.cpu_count() or 2)))
    with ThreadPoolExecutor(jobs) as ex:
        res = list(ex.map(_compile, srcs))
    objs = [o for o, _ in res]
    if any(ch for _, ch in res) or not os.path.exists(LIB):
        cmd = [HIPCC, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", LIB] + objs + ["-lhiprtc"]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n" + r.stderr[-4000:])
    # host-side C++ shim (plain g++; links against the C ABI only)
    host_srcs = [os.path.join(HERE, "host", f) for f in ("batched_defect_constraint.cpp", "kkt_assembly.cpp")]
    host_deps = host_srcs + [os.path.join(HERE, "host", f) for f in ("batched_defect_constraint.h", "kkt_assembly.h")]
    host_lib = os.path.join(HERE, "libasset_host.so")
    if (not os.path.exists(host_lib)
            or os.path.getmtime(host_lib) < max([os.path.getmtime(p) for p in host_deps] + [os.path.getmtime(LIB)])):
        cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-shared"] + host_srcs + ["-o", host_lib, "-L" + HERE, "-lasset_hip",
               "-Wl,-rpath,$ORIGIN"]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("host shim build failed:\n" + r.stderr[-4000:])
    if verbose:
        print(f"[asset_hip] {LIB} ({os.path.getsize(LIB) // 1024} KiB), {host_lib}", flush=True)
    return LIB


if __name__ == "__main__":
    sys.exit(0 if build() else 1)
