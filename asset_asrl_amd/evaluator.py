"""``DefectEvaluator`` -- thin Python handle over the C ABI (one phase shard on one GPU)."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from ._lib import CON, CON_ADJGRAD, JAC, JAC_ADJGRAD, JAC_ADJGRAD_HESS, KEEP_HESSIAN_SLOTS, MODES  # noqa: F401


def _dptr(a):
    return None if a is None else a.ctypes.data_as(C.POINTER(C.c_double))


def reference_slot_order(IR: int, OR: int):
    """(rows, cols) of the canonical numbering of a block's entries -- the reference's order,
    ``for i: {H(j,i), j>=i ; J(j,i)}`` (DenseFunctionBase.h:1112-1123); Jacobian rows are IR + j."""
    rows, cols = [], []
    for i in range(IR):
        rows += list(range(i, IR)) + [IR + j for j in range(OR)]
        cols += [i] * (IR - i + OR)
    return np.asarray(rows, dtype=np.int32), np.asarray(cols, dtype=np.int32)


def kkt_layout_table(IR: int, OR: int, kl: int):
    """Host mirror of the block layouts the kernels write (csrc/defect_dims.h: Dims::KL, hcol / jcol): (stride, rows, cols),
    rows / cols as asset_hip_defect_kkt_layout returns them.  kl 0: the reference's order; kl 1: J column-major, then the packed
    lower triangle of H column-major, each region padded to a multiple of 16 doubles."""
    if not kl:
        r, c = reference_slot_order(IR, OR)
        return r.size, r, c
    jreg, hreg = (OR * IR + 15) // 16 * 16, (IR * (IR + 1) // 2 + 15) // 16 * 16
    rows = np.full(jreg + hreg, -1, dtype=np.int32)
    cols = np.full(jreg + hreg, -1, dtype=np.int32)
    for c in range(IR):
        rows[c * OR:(c + 1) * OR], cols[c * OR:(c + 1) * OR] = np.arange(IR, IR + OR), c
        h0 = jreg + c * IR - c * (c - 1) // 2
        rows[h0:h0 + IR - c], cols[h0:h0 + IR - c] = np.arange(c, IR), c
    return jreg + hreg, rows, cols


class _KktLayout:
    """The order of a handle's KKT blocks (asset_hip_defect_kkt_layout): ``KSTRIDE`` doubles per block, ``kkt_rows`` /
    ``kkt_cols`` per slot (-1: padding), ``kkt_perm[k]`` = offset in the block of canonical entry k."""

    def _read_layout(self, handle):
        L = _lib.lib()
        st = C.c_int()
        kl = L.asset_hip_defect_kkt_layout(handle, C.byref(st), None, None)
        if kl < 0:
            _lib.check(kl, "asset_hip_defect_kkt_layout")
        self.KSTRIDE, self.kkt_layout = st.value, kl
        self.kkt_rows = np.empty(self.KSTRIDE, dtype=np.int32)
        self.kkt_cols = np.empty(self.KSTRIDE, dtype=np.int32)
        ip = C.POINTER(C.c_int32)
        kl = L.asset_hip_defect_kkt_layout(handle, None, self.kkt_rows.ctypes.data_as(ip), self.kkt_cols.ctypes.data_as(ip))
        if kl < 0:
            _lib.check(kl, "asset_hip_defect_kkt_layout")
        self._set_layout(kl, self.KSTRIDE, self.kkt_rows, self.kkt_cols)

    def _set_layout(self, kl, stride, rows, cols):
        """(needs self.IR / OR / NKKT)"""
        self.kkt_layout, self.KSTRIDE, self.kkt_rows, self.kkt_cols = int(kl), int(stride), rows, cols
        where = {(int(r), int(c)): k for k, (r, c) in enumerate(zip(self.kkt_rows, self.kkt_cols)) if r >= 0}
        rr, cc = reference_slot_order(self.IR, self.OR)
        if len(where) != self.NKKT or any((int(r), int(c)) not in where for r, c in zip(rr, cc)):
            raise _lib.AssetHipError("asset_hip_defect_kkt_layout does not enumerate every entry of the block exactly once")
        self.kkt_perm = np.asarray([where[(int(r), int(c))] for r, c in zip(rr, cc)], dtype=np.int64)
        self._perm_identity = self.KSTRIDE == self.NKKT and bool(np.all(self.kkt_perm == np.arange(self.NKKT)))

    def kkt_to_reference(self, kkt):
        """Blocks as the handle writes them ([nseg, KSTRIDE] or flat; numpy or a torch tensor) -> [nseg, NKKT] in the canonical
        (reference) order.  A host-side convenience of tests and scripts: a solver-side consumer walks ``kkt_rows`` /
        ``kkt_cols`` instead (host/batched_defect_constraint.cpp)."""
        if hasattr(kkt, "detach"):
            kkt = kkt.detach().cpu().numpy()
        a = np.asarray(kkt).reshape(-1, self.KSTRIDE)
        return a if self._perm_identity else a[:, self.kkt_perm]


_STREAM_LEGACY = 1      # include/asset_hip.h: ASSET_HIP_STREAM_LEGACY = hipStreamLegacy, the null stream named explicitly


def _stream_arg(stream):
    """The `stream` argument of the C ABI: None -> NULL (the handle's own stream); a raw hipStream_t (int) as it is; a
    torch stream by its handle -- except that torch's DEFAULT stream has handle 0, which the C ABI would read as "no
    stream given" and launch on the handle's private stream, where nothing torch enqueues afterwards is ordered behind
    it.  The null stream is therefore passed by its explicit name."""
    if stream is None:
        return None
    h = stream if isinstance(stream, int) else stream.cuda_stream
    return C.c_void_p(h if h else _STREAM_LEGACY)


class DefectEvaluator(_KktLayout):
    """Evaluates the defect constraint of ``nseg`` mesh segments on a HIP device.

    ``vindex[nseg, IR]`` / ``cindex[nseg, OR]`` are the application-major index tables
    (see indexing.PhaseIndexer.make_defect_Vindex_Cindex).  KKT blocks: the device-pointer entry points (``eval_device``,
    ``bind_device``, ``time_device``) and the pinned outputs take / return them in the HANDLE'S layout -- ``nseg * KSTRIDE``
    doubles, order ``kkt_rows`` / ``kkt_cols`` --; ``eval`` converts to the canonical order unless told ``native=True``.
    """

    def __init__(self, ode: str, mode, blocked: bool, vindex, cindex, n_primal: int, n_equal: int,
                 device: int = 0):
        L = _lib.lib()
        self.ode = ode
        self.mode = MODES[mode] if isinstance(mode, str) else int(mode)
        self.blocked = bool(blocked)
        self.vindex = np.ascontiguousarray(vindex, dtype=np.int32)
        self.cindex = np.ascontiguousarray(cindex, dtype=np.int32)
        if self.vindex.ndim != 2 or self.cindex.ndim != 2 or self.vindex.shape[0] != self.cindex.shape[0]:
            raise ValueError("vindex/cindex must be [nseg, IR] / [nseg, OR]")
        self.nseg = self.vindex.shape[0]
        self.n_primal, self.n_equal = int(n_primal), int(n_equal)
        desc = _lib.DefectDesc(self.mode, int(self.blocked), ode.encode(), self.nseg,
                               self.vindex.ctypes.data_as(C.POINTER(C.c_int32)),
                               self.cindex.ctypes.data_as(C.POINTER(C.c_int32)),
                               self.n_primal, self.n_equal, int(device))
        self._h = C.c_void_p()
        self._pinned = None
        _lib.check(L.asset_hip_defect_create(C.byref(desc), C.byref(self._h)), "asset_hip_defect_create")
        ir, orr, nk = C.c_int(), C.c_int(), C.c_int()
        _lib.check(L.asset_hip_defect_sizes(self._h, C.byref(ir), C.byref(orr), C.byref(nk)))
        self.IR, self.OR, self.NKKT = ir.value, orr.value, nk.value
        self._read_layout(self._h)
        if self.vindex.shape[1] != self.IR or self.cindex.shape[1] != self.OR:
            self.close()
            raise ValueError(f"index tables are [{self.vindex.shape[1]}],[{self.cindex.shape[1]}] wide, "
                             f"the defect needs IR={self.IR}, OR={self.OR}")

    def rebind(self, vindex, cindex, n_primal: int, n_equal: int):
        """New index tables for the same (ODE, transcription, control mode): the re-meshing step of the adaptive mesh loop
        (asset_hip_defect_rebind; ODEPhaseBase.cpp:1443-1542).  Keeps the device code, the per-lane tables, the stream and
        every buffer that still fits; a KKT map, per-application constants and pinned outputs must be set again."""
        v = np.ascontiguousarray(vindex, dtype=np.int32)
        c = np.ascontiguousarray(cindex, dtype=np.int32)
        if v.ndim != 2 or c.ndim != 2 or v.shape[0] != c.shape[0] or v.shape[1] != self.IR or c.shape[1] != self.OR:
            raise ValueError(f"vindex/cindex must be [nseg, {self.IR}] / [nseg, {self.OR}]")
        for b in (self._pinned or ()):
            _lib.lib().asset_hip_host_unregister(b.ctypes.data)
        self._pinned = None
        _lib.check(_lib.lib().asset_hip_defect_rebind(self._h, v.shape[0], v.ctypes.data_as(C.POINTER(C.c_int32)),
                                                      c.ctypes.data_as(C.POINTER(C.c_int32)), int(n_primal), int(n_equal)),
                   "asset_hip_defect_rebind")
        self.vindex, self.cindex, self.nseg = v, c, v.shape[0]
        self.n_primal, self.n_equal = int(n_primal), int(n_equal)
        self._nvalues = -1
        return self

    def close(self):
        h, self._h = getattr(self, "_h", None), None
        if h and _lib is not None and getattr(_lib, "lib", None) is not None:   # module globals vanish at interpreter exit
            for b in (getattr(self, "_pinned", None) or ()):
                _lib.lib().asset_hip_host_unregister(b.ctypes.data)
            self._pinned = None
            _lib.lib().asset_hip_defect_destroy(h)

    __del__ = close

    def IRows(self):
        return self.IR

    def ORows(self):
        return self.OR

    def numKKTEles(self, dojac: bool = True, dohess: bool = True):
        return (self.IR * (self.IR + 1) // 2 if dohess else 0) + (self.OR * self.IR if dojac else 0)

    # ---- host-pointer evaluation -----------------------------------------------------------
    def eval(self, what: int, X, L=None, native: bool = False):
        """Returns (fx[nseg,OR], agx[nseg,IR] or None, kkt or None).  kkt: [nseg, NKKT] in the canonical (reference) order --
        a host-side re-ordering of what the C ABI returned -- or, with ``native=True`` and always with pinned outputs
        (``pin_outputs``), the blocks as the C ABI returns them: [nseg, KSTRIDE] in the handle's layout."""
        X = np.ascontiguousarray(X, dtype=np.float64)
        if X.size != self.n_primal:
            raise ValueError(f"X has {X.size} entries, expected {self.n_primal}")
        if L is not None:
            L = np.ascontiguousarray(L, dtype=np.float64)
            if L.size != self.n_equal:
                raise ValueError(f"L has {L.size} entries, expected {self.n_equal}")
        kind = what & 0xFF                 # (what may carry KEEP_HESSIAN_SLOTS)
        if self._pinned is not None:       # persistent page-locked outputs (pin_outputs): views, overwritten by the next call
            fx, agx, kkt = self._pinned
            agx = agx if kind in (CON_ADJGRAD, JAC_ADJGRAD, JAC_ADJGRAD_HESS) else None
            kkt = kkt if kind >= JAC else None
        else:
            fx = np.empty((self.nseg, self.OR))
            agx = np.empty((self.nseg, self.IR)) if kind in (CON_ADJGRAD, JAC_ADJGRAD, JAC_ADJGRAD_HESS) else None
            kkt = np.empty((self.nseg, self.KSTRIDE)) if kind >= JAC else None
            if kkt is not None and self.KSTRIDE != self.NKKT:
                kkt.fill(np.nan)           # (padding slots are never written: make a consumer that reads them stand out)
        _lib.check(_lib.lib().asset_hip_defect_eval(self._h, what, _dptr(X), _dptr(L), _dptr(fx), _dptr(agx),
                                                    _dptr(kkt)), "asset_hip_defect_eval")
        if kkt is not None and not native and self._pinned is None:
            kkt = self.kkt_to_reference(kkt)
        return fx, agx, kkt

    def set_appl_consts(self, consts):
        """Constants of every application of a plain function that reads some (vf.ApplConst): array [nseg, nconst]."""
        c = np.ascontiguousarray(consts, dtype=np.float64).reshape(self.nseg, -1)
        _lib.check(_lib.lib().asset_hip_defect_set_appl_consts(self._h, _dptr(c), c.shape[1]), "asset_hip_defect_set_appl_consts")

    def pin_outputs(self):
        """Allocate the block arrays once and page-lock them (asset_hip_host_register): ``eval`` then returns these
        arrays, overwritten by every call, and the copies out run at PCIe rate instead of through pageable staging."""
        if self._pinned is None:
            bufs = (np.empty((self.nseg, self.OR)), np.empty((self.nseg, self.IR)), np.empty((self.nseg, self.KSTRIDE)))
            for b in bufs:
                _lib.check(_lib.lib().asset_hip_host_register(b.ctypes.data, b.nbytes), "asset_hip_host_register")
            self._pinned = bufs
        return self

    # ---- on-device KKT assembly -----------------------------------
    def set_kkt_map(self, slot_locations, nvalues: int, accumulate: bool = False):
        """slot_locations[V, k] = KKTLocations[InnerKKTStarts[V] + k]: where entry k (canonical numbering: the reference's
        order, whatever the layout of the handle's blocks) of application V lives in the
        solver's CSR value array of length ``nvalues`` (uploaded once per sparsity analysis).  ``accumulate``: the
        device-pointer evaluation adds into whatever the value array holds (all atomics) instead of expecting zeros
        at this constraint's locations."""
        m = np.ascontiguousarray(slot_locations, dtype=np.int32)
        if m.size != self.nseg * self.NKKT:
            raise ValueError(f"kkt map has {m.size} entries, expected nseg*NKKT = {self.nseg * self.NKKT}")
        _lib.check(_lib.lib().asset_hip_defect_set_kkt_map(self._h, m.ctypes.data_as(C.POINTER(C.c_int32)),
                                                           int(nvalues), int(accumulate)),
                   "asset_hip_defect_set_kkt_map")
        self._nvalues = int(nvalues)

    def eval_assembled(self, what: int, X, L, kkt_values, target_zeroed: bool = False):
        """Like :meth:`eval`, but the KKT entries are ADDED into ``kkt_values`` (the solver's value array) on the
        device; returns (fx blocks, agx blocks or None).  ``target_zeroed``: the array was just cleared and this is the
        first function to fill its range -- the range is overwritten by one device-to-host copy, no host add."""
        X = np.ascontiguousarray(X, dtype=np.float64)
        L = None if L is None else np.ascontiguousarray(L, dtype=np.float64)
        if X.size != self.n_primal or (L is not None and L.size != self.n_equal):
            raise ValueError("X / L size mismatch")
        if (not isinstance(kkt_values, np.ndarray) or kkt_values.dtype != np.float64 or not kkt_values.flags.c_contiguous
                or kkt_values.size != getattr(self, "_nvalues", -1)):
            raise ValueError("kkt_values must be the contiguous float64 value array the map was built for")
        fx = np.empty((self.nseg, self.OR))
        agx = np.empty((self.nseg, self.IR)) if what in (JAC_ADJGRAD, JAC_ADJGRAD_HESS) else None
        fn = _lib.lib().asset_hip_defect_eval_assembled_zeroed if target_zeroed else _lib.lib().asset_hip_defect_eval_assembled
        _lib.check(fn(self._h, what, _dptr(X), _dptr(L), _dptr(fx), _dptr(agx), _dptr(kkt_values)),
                   "asset_hip_defect_eval_assembled")
        return fx, agx

    def eval_assembled_device(self, what: int, X, L, fx, agx, kkt_values, stream=None):
        st = _stream_arg(stream)
        _lib.check(_lib.lib().asset_hip_defect_eval_assembled_device(self._h, what, self._p(X), self._p(L), self._p(fx),
                                                                     self._p(agx), self._p(kkt_values), st),
                   "asset_hip_defect_eval_assembled_device")

    def eval_kkt_device(self, what: int, X, L, FXE, AGX, kkt_values, stream=None):
        """Constraint values ADDED into FXE[n_equal], adjoint gradient into AGX[n_primal], KKT entries into kkt_values
        (as eval_assembled_device): the constraint's whole share of an evalKKT on the device, bitwise repeatable."""
        st = _stream_arg(stream)
        _lib.check(_lib.lib().asset_hip_defect_eval_kkt_device(self._h, what, self._p(X), self._p(L), self._p(FXE),
                                                               self._p(AGX), self._p(kkt_values), st),
                   "asset_hip_defect_eval_kkt_device")

    # ---- device-pointer evaluation (torch tensors or raw ints) ------------------------------
    @staticmethod
    def _p(t):
        if t is None:
            return None
        return C.c_void_p(t if isinstance(t, int) else t.data_ptr())

    def eval_device(self, what: int, X, L, fx, agx, kkt, stream=None):
        st = _stream_arg(stream)
        _lib.check(_lib.lib().asset_hip_defect_eval_device(self._h, what, self._p(X), self._p(L), self._p(fx),
                                                           self._p(agx), self._p(kkt), st),
                   "asset_hip_defect_eval_device")

    def bind_device(self, what: int, X, L, fx, agx, kkt, stream=None):
        """A zero-argument callable that enqueues this evaluation: the ctypes arguments are converted once, so a solver
        loop (or bench.py) pays about a microsecond of host time per call instead of the ~20 us of `eval_device`.  The
        tensors must stay alive (and in place) while the callable is used."""
        fn = _lib.lib().asset_hip_defect_eval_device
        st = _stream_arg(stream)
        args = (self._h, what, self._p(X), self._p(L), self._p(fx), self._p(agx), self._p(kkt), st)
        keep = (X, L, fx, agx, kkt)

        def call(_fn=fn, _args=args, _keep=keep):
            rc = _fn(*_args)
            if rc:
                _lib.check(rc, "asset_hip_defect_eval_device")
        return call

    def time_device(self, what: int, X, L, fx, agx, kkt, warmup: int = 3, iters: int = 20) -> float:
        ms = C.c_float()
        _lib.check(_lib.lib().asset_hip_defect_time_device(self._h, what, self._p(X), self._p(L), self._p(fx),
                                                           self._p(agx), self._p(kkt), warmup, iters, C.byref(ms)),
                   "asset_hip_defect_time_device")
        return float(ms.value)


def unpack_kkt_block(blk: np.ndarray, IR: int, OR: int):
    """A block in the canonical (reference) order -> (H lower-triangular filled symmetric [IR,IR], J [OR,IR])."""
    H = np.zeros((IR, IR))
    J = np.zeros((OR, IR))
    k = 0
    for i in range(IR):
        H[i:, i] = blk[k:k + IR - i]
        k += IR - i
        J[:, i] = blk[k:k + OR]
        k += OR
    H = H + np.tril(H, -1).T
    return H, J


class ShardedDefectEvaluator(_KktLayout):
    """One constraint as several device handles in ONE process (include/asset_hip.h: asset_hip_defect_create_sharded): the
    reference's ``ConstraintFunction.thread_split`` (ConstraintFunction.h:55-62, IndexingData.h:117-146) with a device per
    chunk instead of a CPU thread.  ``devices``: one HIP ordinal per shard; naming a device several times gives several
    handles on it.  Host-pointer evaluation only: every shard's blocks land in the caller's arrays over its own PCIe link.
    ``pin_outputs()`` keeps ONE set of page-locked output arrays across calls (``eval`` then returns views of them, overwritten by
    the next call, the KKT blocks in the handles' layout) -- what a solver does with its RHS and KKT arrays; without it the
    outputs are fresh pageable arrays and the library drives every shard from a host thread of its own."""

    def __init__(self, ode: str, mode, blocked: bool, vindex, cindex, n_primal: int, n_equal: int, devices):
        L = _lib.lib()
        self.mode = MODES[mode] if isinstance(mode, str) else int(mode)
        self.vindex = np.ascontiguousarray(vindex, dtype=np.int32)
        self.cindex = np.ascontiguousarray(cindex, dtype=np.int32)
        self.nseg = self.vindex.shape[0]
        self.n_primal, self.n_equal = int(n_primal), int(n_equal)
        devs = (C.c_int * len(devices))(*[int(d) for d in devices])
        desc = _lib.DefectDesc(self.mode, int(bool(blocked)), ode.encode(), self.nseg,
                               self.vindex.ctypes.data_as(C.POINTER(C.c_int32)), self.cindex.ctypes.data_as(C.POINTER(C.c_int32)),
                               self.n_primal, self.n_equal, 0)
        self._s = C.c_void_p()
        _lib.check(L.asset_hip_defect_create_sharded(C.byref(desc), len(devices), devs, C.byref(self._s)), "asset_hip_defect_create_sharded")
        ir, orr, nk = C.c_int(), C.c_int(), C.c_int()
        _lib.check(L.asset_hip_defect_sizes(C.c_void_p(L.asset_hip_sharded_handle(self._s, 0)), C.byref(ir), C.byref(orr), C.byref(nk)))
        self.IR, self.OR, self.NKKT = ir.value, orr.value, nk.value
        self._read_layout(C.c_void_p(L.asset_hip_sharded_handle(self._s, 0)))
        self._nvalues = -1
        self._pinned = None

    def pin_outputs(self):
        if self._pinned is None:
            bufs = (np.empty((self.nseg, self.OR)), np.empty((self.nseg, self.IR)), np.empty((self.nseg, self.KSTRIDE)))
            for b in bufs:
                _lib.check(_lib.lib().asset_hip_host_register(b.ctypes.data, b.nbytes), "asset_hip_host_register")
            self._pinned = bufs
        return self

    @property
    def ranges(self):
        """[(first application, count, device)] of the shards (the ByApplication split)."""
        L, out = _lib.lib(), []
        for i in range(L.asset_hip_sharded_shards(self._s)):
            f, c, d = C.c_int(), C.c_int(), C.c_int()
            _lib.check(L.asset_hip_sharded_range(self._s, i, C.byref(f), C.byref(c), C.byref(d)))
            out.append((f.value, c.value, d.value))
        return out

    def eval(self, what: int, X, L=None, native: bool = False):
        """As DefectEvaluator.eval: kkt in the canonical order unless ``native`` (or pinned outputs)."""
        X = np.ascontiguousarray(X, dtype=np.float64)
        L = None if L is None else np.ascontiguousarray(L, dtype=np.float64)
        kind = what & 0xFF
        if self._pinned is not None:
            fx, agx, kkt = self._pinned
            agx = agx if kind in (CON_ADJGRAD, JAC_ADJGRAD, JAC_ADJGRAD_HESS) else None
            kkt = kkt if kind >= JAC else None
        else:
            fx = np.empty((self.nseg, self.OR))
            agx = np.empty((self.nseg, self.IR)) if kind in (CON_ADJGRAD, JAC_ADJGRAD, JAC_ADJGRAD_HESS) else None
            kkt = np.empty((self.nseg, self.KSTRIDE)) if kind >= JAC else None
        _lib.check(_lib.lib().asset_hip_sharded_eval(self._s, what, _dptr(X), _dptr(L), _dptr(fx), _dptr(agx), _dptr(kkt)),
                   "asset_hip_sharded_eval")
        if kkt is not None and not native and self._pinned is None:
            kkt = self.kkt_to_reference(kkt)
        return fx, agx, kkt

    def set_kkt_map(self, slot_locations, nvalues: int):
        m = np.ascontiguousarray(slot_locations, dtype=np.int32).reshape(self.nseg, self.NKKT)
        _lib.check(_lib.lib().asset_hip_sharded_set_kkt_map(self._s, m.ctypes.data_as(C.POINTER(C.c_int32)), int(nvalues)),
                   "asset_hip_sharded_set_kkt_map")
        self._nvalues = int(nvalues)
        return self

    def eval_assembled(self, what: int, X, L, values):
        """``values[nvalues]`` is accumulated into; returns (fx, agx or None)."""
        X = np.ascontiguousarray(X, dtype=np.float64)
        L = None if L is None else np.ascontiguousarray(L, dtype=np.float64)
        if values.dtype != np.float64 or not values.flags.c_contiguous or values.size != self._nvalues:
            raise ValueError("values must be a contiguous float64 array of the map's length")
        kind = what & 0xFF
        if self._pinned is not None:
            fx, agx = self._pinned[0], (self._pinned[1] if kind in (JAC_ADJGRAD, JAC_ADJGRAD_HESS) else None)
        else:
            fx = np.empty((self.nseg, self.OR))
            agx = np.empty((self.nseg, self.IR)) if kind in (JAC_ADJGRAD, JAC_ADJGRAD_HESS) else None
        _lib.check(_lib.lib().asset_hip_sharded_eval_assembled(self._s, what, _dptr(X), _dptr(L), _dptr(fx), _dptr(agx), _dptr(values)),
                   "asset_hip_sharded_eval_assembled")
        return fx, agx

    def close(self):
        s, self._s = getattr(self, "_s", None), None
        if s and _lib is not None and getattr(_lib, "lib", None) is not None:
            for b in (getattr(self, "_pinned", None) or ()):
                _lib.lib().asset_hip_host_unregister(b.ctypes.data)
            self._pinned = None
            _lib.lib().asset_hip_sharded_destroy(s)

    __del__ = close
