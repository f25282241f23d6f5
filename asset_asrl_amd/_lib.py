"""ctypes binding of the C ABI in include/asset_hip.h (libasset_hip.so).

There is deliberately no fallback: if the HIP library is missing or no device is visible, every
evaluation entry point raises.  The library is built in-tree by ``asset_asrl_amd.build``.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("ASSET_HIP_LIB") or os.path.join(_HERE, "libasset_hip.so")  # env override: kernel experiments

FUNCTION, TRAPEZOIDAL, LGL3, LGL5, LGL7 = 0, 1, 2, 3, 4
MODES = {"Function": FUNCTION, "Trapezoidal": TRAPEZOIDAL, "LGL3": LGL3, "LGL5": LGL5, "LGL7": LGL7}
CON, CON_ADJGRAD, JAC, JAC_ADJGRAD, JAC_ADJGRAD_HESS = range(5)
KEEP_HESSIAN_SLOTS = 0x100   # OR into JAC / JAC_ADJGRAD: Hessian slots of the blocks left untouched (include/asset_hip.h)

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int32)


class DefectDesc(C.Structure):
    _fields_ = [("mode", C.c_int), ("blocked", C.c_int), ("ode", C.c_char_p), ("nseg", C.c_int),
                ("vindex", _ip), ("cindex", _ip), ("n_primal", C.c_int), ("n_equal", C.c_int),
                ("device", C.c_int)]


# every symbol include/asset_hip.h declares: name -> (restype, argtypes)
SYMBOLS = {
    "asset_hip_defect_create": (C.c_int, [C.POINTER(DefectDesc), C.POINTER(C.c_void_p)]),
    "asset_hip_defect_destroy": (None, [C.c_void_p]),
    "asset_hip_defect_rebind": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.c_int, C.c_int]),
    "asset_hip_defect_sizes": (C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "asset_hip_defect_kkt_layout": (C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "asset_hip_kkt_layout": (C.c_int, [C.c_char_p, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int32),
                                      C.POINTER(C.c_int32)]),
    "asset_hip_defect_eval": (C.c_int, [C.c_void_p, C.c_int, _dp, _dp, _dp, _dp, _dp]),
    "asset_hip_defect_eval_device": (C.c_int, [C.c_void_p, C.c_int] + [C.c_void_p] * 6),
    "asset_hip_jit_compile": (C.c_int, [C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_char_p),
                                       C.c_int, C.c_char_p]),
    "asset_hip_jit_plugin": (C.c_int, [C.c_char_p, C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                      C.POINTER(C.c_char_p), C.c_int, C.c_char_p]),
    "asset_hip_bundle_create": (C.c_int, [C.c_char_p, C.POINTER(C.c_void_p), C.c_int, C.POINTER(C.c_void_p)]),
    "asset_hip_bundle_eval_device": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p),
                                              C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.c_void_p]),
    "asset_hip_bundle_destroy": (None, [C.c_void_p]),
    "asset_hip_defect_time_device": (C.c_int, [C.c_void_p, C.c_int] + [C.c_void_p] * 5 + [C.c_int, C.c_int,
                                                                                         C.POINTER(C.c_float)]),
    "asset_hip_defect_set_appl_consts": (C.c_int, [C.c_void_p, _dp, C.c_int]),
    "asset_hip_defect_set_kkt_map": (C.c_int, [C.c_void_p, C.POINTER(C.c_int32), C.c_longlong, C.c_int]),
    "asset_hip_defect_eval_assembled": (C.c_int, [C.c_void_p, C.c_int, _dp, _dp, _dp, _dp, _dp]),
    "asset_hip_defect_eval_assembled_zeroed": (C.c_int, [C.c_void_p, C.c_int, _dp, _dp, _dp, _dp, _dp]),
    "asset_hip_defect_eval_assembled_device": (C.c_int, [C.c_void_p, C.c_int] + [C.c_void_p] * 6),
    "asset_hip_defect_eval_kkt_device": (C.c_int, [C.c_void_p, C.c_int] + [C.c_void_p] * 6),
    "asset_hip_defect_create_sharded": (C.c_int, [C.POINTER(DefectDesc), C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_void_p)]),
    "asset_hip_sharded_destroy": (None, [C.c_void_p]),
    "asset_hip_sharded_shards": (C.c_int, [C.c_void_p]),
    "asset_hip_sharded_range": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "asset_hip_sharded_handle": (C.c_void_p, [C.c_void_p, C.c_int]),
    "asset_hip_sharded_eval": (C.c_int, [C.c_void_p, C.c_int, _dp, _dp, _dp, _dp, _dp]),
    "asset_hip_sharded_set_kkt_map": (C.c_int, [C.c_void_p, C.POINTER(C.c_int32), C.c_longlong]),
    "asset_hip_sharded_eval_assembled": (C.c_int, [C.c_void_p, C.c_int, _dp, _dp, _dp, _dp, _dp]),
    "asset_hip_mesh_error_deboor": (C.c_int, [C.c_char_p, C.c_int, C.c_int, _dp, C.c_int, _dp, _dp, _dp, _dp, _dp, C.c_int]),
    "asset_hip_host_register": (C.c_int, [C.c_void_p, C.c_size_t]),
    "asset_hip_host_unregister": (C.c_int, [C.c_void_p]),
    "asset_hip_num_odes": (C.c_int, []),
    "asset_hip_ode_name": (C.c_char_p, [C.c_int]),
    "asset_hip_ode_sizes": (C.c_int, [C.c_char_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "asset_hip_has_kernel": (C.c_int, [C.c_char_p, C.c_int, C.c_int]),
    "asset_hip_load_plugin": (C.c_int, [C.c_char_p]),
    "asset_hip_lgl_table": (C.c_int, [C.c_int, C.c_char_p, _dp, C.c_int]),
    "asset_hip_device_count": (C.c_int, []),
    "asset_hip_last_error": (C.c_char_p, []),
    "asset_hip_version": (C.c_char_p, []),
}

_LIB = None


class AssetHipError(RuntimeError):
    pass


def lib():
    """Load libasset_hip.so (fails loudly when it has not been built)."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise AssetHipError(f"{LIB_PATH} not found: run `python -m asset_asrl_amd.build` "
                                "(the HIP evaluator has no CPU fallback)")
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(L, name)
            fn.restype, fn.argtypes = res, args
        _LIB = L
    return _LIB


def check(rc: int, what: str = ""):
    if rc != 0:
        msg = lib().asset_hip_last_error().decode(errors="replace")
        raise AssetHipError(f"{what or 'asset_hip'} failed (rc={rc}): {msg}")


def ode_names():
    L = lib()
    return [L.asset_hip_ode_name(i).decode() for i in range(L.asset_hip_num_odes())]


def ode_sizes(name: str):
    xv, uv, pv = C.c_int(), C.c_int(), C.c_int()
    check(lib().asset_hip_ode_sizes(name.encode(), C.byref(xv), C.byref(uv), C.byref(pv)), "asset_hip_ode_sizes")
    return xv.value, uv.value, pv.value


def kkt_layout(name: str, mode: int, blocked: bool):
    """(layout id, NKKT, stride, rows, cols) of the KKT blocks of a compiled (ode, mode, blocked): asset_hip_kkt_layout -- no
    handle, no device."""
    nk, st = C.c_int(), C.c_int()
    kl = lib().asset_hip_kkt_layout(name.encode(), mode, int(blocked), C.byref(nk), C.byref(st), None, None)
    if kl < 0:
        check(kl, "asset_hip_kkt_layout")
    rows, cols = np.empty(st.value, dtype=np.int32), np.empty(st.value, dtype=np.int32)
    kl = lib().asset_hip_kkt_layout(name.encode(), mode, int(blocked), None, None, rows.ctypes.data_as(_ip), cols.ctypes.data_as(_ip))
    if kl < 0:
        check(kl, "asset_hip_kkt_layout")
    return kl, nk.value, st.value, rows, cols


def has_kernel(name: str, mode: int, blocked: bool) -> bool:
    return bool(lib().asset_hip_has_kernel(name.encode(), mode, int(blocked)))


def lgl_table(cs: int, which: str) -> np.ndarray:
    buf = np.zeros(12)
    n = lib().asset_hip_lgl_table(cs, which.encode(), buf.ctypes.data_as(_dp), buf.size)
    if n < 0:
        check(n, "asset_hip_lgl_table")
    out = buf[:n].copy()
    return out.reshape(cs - 1, cs) if which in "ABUCD" else out


def device_count() -> int:
    return lib().asset_hip_device_count()
