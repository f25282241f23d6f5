"""Multi-GPU sharding of a phase's defect constraint: one process per GPU (torch.distributed, RCCL on ROCm).

Segments are independent given X and L (every application reads only its own Vindex/Cindex column,
/root/reference/src/VectorFunctions/DenseFunctionBase.h:1296-1312), so the evaluation itself needs no
collective: rank r evaluates the contiguous range the reference's ByApplication rule would give thread r
(/root/reference/src/VectorFunctions/IndexingData.h:117-146; the worker launch it replaces:
/root/reference/src/Solvers/NonLinearProgram.cpp:519-526).  The one exchange step of a solver iteration is the
gather of the per-shard FX / AGX / KKT blocks to the rank that owns the host KKT system: disjoint blocks, no
reduction -- boundary-node Hessian entries are emitted by both neighbours and summed by the scatter, exactly as on
one device.

Device-resident path (``alloc_device`` / ``eval_device`` / ``gather_device``): every rank owns ONE flat buffer in
HBM, ``[fx | agx | kkt]`` sized for the largest shard, that the evaluation kernels write in place; the exchange is a
single ``dist.gather`` of that buffer into the root's ``[world, slot]`` receive buffer (RCCL: grouped send/recv over
xGMI, every peer on its own link).  Nothing touches the host.  Shards differ by at most one segment, so the padding is
at most one segment per rank; ``shard_blocks_on_root`` returns per-shard views of the receive buffer (no copy),
``blocks_on_root`` the phase-order concatenation.

Host-visible path (``HostSharedBlocks``): the reference's solver keeps the KKT system on the host, and on one GPU the
blocks reach it over ONE PCIe link (84 MB for a 10 000-segment LGL7 phase: 1.7 ms, 40x the evaluation).  With one
process per GPU every rank has a link of its own: each rank copies its flat buffer straight into its range of one
page-locked host buffer that all ranks map (a file in /dev/shm), and the only synchronisation is a barrier -- the shards
never cross xGMI and never funnel through the root's link.  This is where more GPUs buy the host solver time even when
the evaluation itself is tens of microseconds.

``PhaseShardedEvaluator`` is the multi-phase form (BASELINE.json configs[3]: eight linked phases): whole phases are
dealt to the ranks round-robin -- phases are independent given X and L too
(/root/reference/src/OptimalControl/OptimalControlProblem.cpp:115-155) -- and gathered the same way.
"""
from __future__ import annotations

from typing import Callable, List, Optional, Sequence

import numpy as np

from .indexing import thread_split

_ADJ = (1, 3, 4)     # evaluation kinds that produce an adjoint gradient (CON_ADJGRAD, JAC_ADJGRAD, JAC_ADJGRAD_HESS)


def _sizes(ode, mode, blocked):
    """(IR, OR, NKKT, KSTRIDE) of a rank that owns no segment (it still takes part in the exchange): from the library's tables."""
    from . import _lib, synth
    from .build import dims
    d = dims(*_lib.ode_sizes(ode), synth.MODE_CS[mode] if isinstance(mode, str) else max(mode, 2), blocked)
    mode_id = _lib.MODES[mode] if isinstance(mode, str) else int(mode)
    _, nk, stride, _, _ = _lib.kkt_layout(ode, mode_id, bool(blocked))
    assert nk == d["NKKT"]
    return d["IR"], d["OR"], nk, stride


def _al16(n: int) -> int:
    return (n + 15) // 16 * 16          # (a block array starts on a 128-byte line)


class HostSharedBlocks:
    """One host buffer of ``world * slot_doubles`` doubles that every rank of the job maps (POSIX shared memory) and
    page-locks; rank r owns the range [r * slot_doubles, (r + 1) * slot_doubles).  ``push`` enqueues the D2H copy of a
    rank's flat device buffer into its range; ``wait`` completes it and meets the other ranks at a barrier, after which
    the root (any rank) reads every shard from ``.buf`` -- host memory, no collective in the data path."""

    def __init__(self, slot_doubles: int, rank: int, world: int, group=None, barrier_group=None, tag: Optional[str] = None,
                 total_doubles: Optional[int] = None):
        import os
        import torch
        import torch.distributed as dist
        self.rank, self.world, self.slot = rank, world, int(slot_doubles)
        self.group, self.barrier_group = group, barrier_group
        self._dist = dist if (dist.is_available() and dist.is_initialized()) else None
        if tag is None:                       # one name for the whole job, chosen by rank 0
            box = [f"asset_hip_blocks_{os.getpid()}_{int.from_bytes(os.urandom(4), 'little'):08x}"]
            if self._dist is not None and world > 1:
                self._dist.broadcast_object_list(box, src=0, group=group)
            tag = box[0]
        self.path = os.path.join("/dev/shm", tag)
        n = self.slot * world if total_doubles is None else int(total_doubles)   # (total_doubles: a layout of the caller's own)
        self.buf, self._registered = None, False
        # Every local step that can fail is followed by an agreement among the ranks (an all-reduce of a success flag in
        # place of a bare barrier): a failure on one rank raises on ALL of them at the same point, nobody is left waiting
        # in a collective, and rank 0 removes the /dev/shm name whatever happened.
        try:
            err = None
            if rank == 0:
                try:
                    with open(self.path, "wb") as f:
                        f.truncate(n * 8)
                except OSError as exc:
                    err = exc
            self._agree(err, "creating the shared block buffer")
            try:
                self.buf = torch.from_file(self.path, shared=True, size=n, dtype=torch.float64)
                if torch.cuda.is_available():     # page-lock the mapping: the D2H copies then run at the link's rate
                    rc = torch.cuda.cudart().cudaHostRegister(self.buf.data_ptr(), n * 8, 0)
                    if int(rc) != 0:
                        raise RuntimeError(f"cudaHostRegister of the shared block buffer failed (rc={int(rc)})")
                    self._registered = True
            except Exception as exc:              # noqa: BLE001 -- whatever it is, the other ranks must hear of it
                err = exc
            self._agree(err, "mapping / page-locking the shared block buffer")
            self.mine = self.buf[rank * self.slot:(rank + 1) * self.slot] if total_doubles is None else None
        except Exception:
            self.close()
            raise
        finally:
            if rank == 0:                         # every rank holds its mapping (or the construction failed everywhere)
                try:
                    os.unlink(self.path)
                except OSError:
                    pass

    def _agree(self, err, what: str):
        """All ranks learn whether any of them failed; raises on every rank if so."""
        ok = 0 if err is not None else 1
        if self._dist is not None and self.world > 1:
            import torch
            g = self.barrier_group if self.barrier_group is not None else self.group
            on_gpu = self._dist.get_backend(g) == "nccl"
            flag = torch.tensor([ok], dtype=torch.int32, device="cuda" if on_gpu else "cpu")
            self._dist.all_reduce(flag, op=self._dist.ReduceOp.MIN, group=g)
            ok = int(flag.item())
        if not ok:
            raise RuntimeError(f"HostSharedBlocks: {what} failed on " + ("this rank: " + repr(err) if err is not None else "another rank"))

    def _barrier(self):
        if self._dist is not None and self.world > 1:
            self._dist.barrier(group=self.barrier_group if self.barrier_group is not None else self.group)

    def push(self, local_flat):
        """Enqueue (on the current stream of `local_flat`'s device) the copy of this rank's flat buffer into its range."""
        self.mine.copy_(local_flat, non_blocking=True)

    def wait(self, stream=None):
        import torch
        if torch.cuda.is_available():
            (stream if stream is not None else torch.cuda.current_stream()).synchronize()
        self._barrier()

    def shard(self, r: int):
        return self.buf[r * self.slot:(r + 1) * self.slot]

    def close(self):
        import torch
        if self._registered:
            torch.cuda.cudart().cudaHostUnregister(self.buf.data_ptr())
            self._registered = False
        self.buf = self.mine = None           # drops the mapping


class _StreamOrder:
    """Orders the exchange after the evaluation whatever streams the caller uses.  ``eval_device(stream=None)`` enqueues on
    torch's CURRENT stream (not on the handle's private stream, which torch knows nothing about); every evaluation records
    an event on the stream it went to, and ``gather_device`` / ``push_host`` make the stream they run on wait for it -- a
    no-op on the same stream, a cross-stream dependency otherwise.  (c10d orders a collective after the current stream
    only.)"""
    _eval_event = None
    _ordered = False          # set once there is an exchange to order (a receive buffer or a host-shared buffer exists)

    def _eval_stream(self, stream):
        import torch
        if not torch.cuda.is_available():
            return stream
        return torch.cuda.current_stream() if stream is None else stream

    def _mark_evaluated(self, stream):
        import torch
        if not self._ordered or not torch.cuda.is_available() or stream is None or isinstance(stream, int):
            return                # (no exchange, nothing to order: an event record costs ~4 us per evaluation)
        if self._eval_event is None:
            self._eval_event = torch.cuda.Event()
        self._eval_event.record(stream)

    def _after_evaluation(self):
        import torch
        if self._eval_event is not None and torch.cuda.is_available():
            torch.cuda.current_stream().wait_event(self._eval_event)


class ShardedDefectEvaluator(_StreamOrder):
    def __init__(self, ode: str, mode, blocked: bool, vindex, cindex, n_primal: int, n_equal: int,
                 rank: Optional[int] = None, world: Optional[int] = None, device: int = 0, group=None,
                 evaluator_factory: Optional[Callable] = None):
        import torch.distributed as dist
        self.dist = dist
        self.group = group
        self.rank = dist.get_rank(group) if rank is None else rank
        self.world = dist.get_world_size(group) if world is None else world
        vindex = np.ascontiguousarray(vindex, dtype=np.int32)
        cindex = np.ascontiguousarray(cindex, dtype=np.int32)
        self.nseg_total = vindex.shape[0]
        self.shards = thread_split(self.nseg_total, self.world)
        self.shards += [(self.nseg_total, 0)] * (self.world - len(self.shards))   # fewer segments than ranks
        self.start, self.count = self.shards[self.rank]
        self.max_count = max(c for _, c in self.shards)
        self.ev = None
        if self.count > 0:
            if evaluator_factory is None:
                from .evaluator import DefectEvaluator
                evaluator_factory = DefectEvaluator
            self.ev = evaluator_factory(ode, mode, blocked, vindex[self.start:self.start + self.count],
                                        cindex[self.start:self.start + self.count], n_primal, n_equal, device)
            self.IR, self.OR, self.NKKT = self.ev.IR, self.ev.OR, self.ev.NKKT
            self.KSTRIDE = getattr(self.ev, "KSTRIDE", self.NKKT)      # blocks travel in the handle's layout (DefectEvaluator.kkt_rows / kkt_cols)
        else:
            self.IR, self.OR, self.NKKT, self.KSTRIDE = _sizes(ode, mode, blocked)
        self._local = self._recv = None

    def kkt_to_reference(self, kkt):
        """KKT blocks as the ranks write and exchange them ([n, KSTRIDE]) -> [n, NKKT] in the canonical (reference) order
        (host-side; needs a rank that owns segments -- DefectEvaluator.kkt_to_reference)."""
        if self.ev is not None and hasattr(self.ev, "kkt_to_reference"):
            return self.ev.kkt_to_reference(kkt)
        return kkt

    # ---- local evaluation (host arrays) -----------------------------------------------------
    def eval_local(self, what: int, X, L=None):
        """Host-pointer evaluation of this rank's shard -> (fx, agx, kkt) numpy blocks (or empty arrays)."""
        if self.ev is None:
            return (np.zeros((0, self.OR)), np.zeros((0, self.IR)), np.zeros((0, self.NKKT)))
        return self.ev.eval(what, X, L)

    # ---- device-resident evaluation + exchange ---------------------------------------------
    @property
    def slot_doubles(self) -> int:
        """Length of a rank's flat output buffer: [fx | agx | kkt] for max_count segments."""
        return _al16(self.max_count * (self.OR + self.IR)) + self.max_count * self.KSTRIDE

    def _views(self, flat, count=None):
        m = self.max_count
        o1, o2, o3 = m * self.OR, m * (self.OR + self.IR), _al16(m * (self.OR + self.IR))
        fx, agx, kkt = flat[:o1].view(m, self.OR), flat[o1:o2].view(m, self.IR), flat[o3:].view(m, self.KSTRIDE)
        if count is not None:
            fx, agx, kkt = fx[:count], agx[:count], kkt[:count]
        return fx, agx, kkt

    def alloc_device(self, device, dst: int = 0, always_exchange: bool = False):
        """Allocate the flat output buffer on `device` (and the receive buffer on rank `dst`).  The padding segment of
        a shorter shard is zero and stays zero.  `always_exchange`: run the gather even in a world of one rank (tests)."""
        import torch
        self._dst = dst
        self._local = torch.zeros(self.slot_doubles, dtype=torch.float64, device=device)
        self.fx, self.agx, self.kkt = self._views(self._local)
        if self.rank == dst and (self.world > 1 or always_exchange):
            self._recv = torch.empty((self.world, self.slot_doubles), dtype=torch.float64, device=device)
        self._ordered = self.world > 1 or always_exchange
        return self

    def eval_device(self, what: int, X, L=None, stream=None):
        """Evaluate this rank's shard into its flat buffer (enqueued on `stream`, not synchronised).  X / L: device
        tensors (the whole solver vectors).  Returns the (fx, agx, kkt) views of the local shard."""
        if self._local is None:
            raise RuntimeError("call alloc_device() first")
        stream = self._eval_stream(stream)
        if self.ev is not None:
            self.ev.eval_device(what, X, L, self.fx, self.agx if what in _ADJ else None,
                                self.kkt if what >= 2 else None, stream)
        self._mark_evaluated(stream)
        return self.fx[:self.count], self.agx[:self.count], self.kkt[:self.count]

    def gather_device(self, async_op: bool = False):
        """The exchange step: one gather of every rank's flat buffer to the root, device to device.  Ordered after the
        work already enqueued on the current stream (c10d semantics); returns the work handle when `async_op`."""
        if self.world == 1 and self._recv is None:
            return None
        self._after_evaluation()
        out = list(self._recv.unbind(0)) if self.rank == self._dst else None
        return self.dist.gather(self._local, out, dst=self._dst, group=self.group, async_op=async_op)

    def alloc_host_shared(self, barrier_group=None, tag: Optional[str] = None):
        """The host-visible exchange: a page-locked host buffer shared by the ranks (HostSharedBlocks).  After
        ``eval_device(...); push_host(); wait_host()`` every rank's (fx, agx, kkt) blocks are in host memory;
        ``host_shard_blocks()`` returns them per shard as numpy views."""
        self._host = HostSharedBlocks(self.slot_doubles, self.rank, self.world, self.group, barrier_group, tag)
        self._ordered = True
        return self

    def push_host(self):
        self._after_evaluation()
        self._host.push(self._local)

    def wait_host(self, stream=None):
        self._host.wait(stream)

    def host_shard_blocks(self):
        return [tuple(v.numpy() for v in self._views(self._host.shard(r), c)) for r, (_, c) in enumerate(self.shards)]

    def shard_blocks_on_root(self):
        """Per rank, the (fx, agx, kkt) views of its shard in the root's receive buffer after `gather_device` -- no
        copy; shard r covers segments shards[r][0] ... of the phase.  None on the other ranks."""
        if self._recv is None:
            return [self._views(self._local, self.count)] if self.world == 1 else None
        return [self._views(self._recv[r], c) for r, (_, c) in enumerate(self.shards)]

    def blocks_on_root(self):
        """(fx[nseg_total, OR], agx[nseg_total, IR], kkt[nseg_total, KSTRIDE], handle layout) on the root (None elsewhere): the shards
        of `shard_blocks_on_root` concatenated per kind (one device copy; the scatter can as well walk the shards)."""
        import torch
        per = self.shard_blocks_on_root()
        if per is None:
            return None
        if len(per) == 1:
            return per[0]
        return tuple(torch.cat([p[k] for p in per], dim=0) for k in range(3))

    # ---- sharded on-device assembly (SURVEY section 8 rows f-1 x e) ---------------------------------
    # Every rank assembles ITS shard into a compact value array on its device -- the locations its slots name, sorted -- and
    # pushes it over its own PCIe link straight into the solver's value array in shared page-locked host memory: segments are
    # contiguous in the variable order, so a shard's locations are one long run of the CSR value array
    # (NonLinearProgram.cpp:316-330) plus a few short pieces around the nodes it shares with its neighbours and in the rows of
    # phase parameters.  Long runs that only this rank touches are copied to their place directly; everything else (short runs,
    # locations two or more ranks contribute to) travels as one packed side vector per rank, and the root adds those -- a few
    # hundred entries per shard boundary -- after the barrier.  The host-side scatter of the reference
    # (DenseFunctionBase.h:1449-1465, one indexed += per slot) is gone; no block array leaves a device.
    SIDE_RUN = 4096          # runs of fewer locations than this go through the packed side vector

    def set_kkt_map(self, slot_locations, nvalues: int):
        """slot_locations[nseg_total, NKKT] = KKTLocations[InnerKKTStarts[V] + k] of the WHOLE constraint (every rank passes the
        same table, as it passes the same index tables), nvalues = length of the solver's value array; -1 drops a slot."""
        m = np.ascontiguousarray(slot_locations, dtype=np.int64).reshape(self.nseg_total, self.NKKT)
        self._nvalues = int(nvalues)
        locs = []
        for s, c in self.shards:
            l = np.unique(m[s:s + c]) if c > 0 else np.zeros(0, dtype=np.int64)
            locs.append(l[l >= 0])
        cnt = np.zeros(self._nvalues, dtype=np.int16)            # ranks that contribute to a location
        for l in locs:
            cnt[l] += 1
        plans = []
        for l in locs:                                           # the same plan on every rank: the root needs all of them
            if l.size == 0:
                plans.append((np.zeros((0, 3), dtype=np.int64), np.zeros(0, dtype=np.int64)))
                continue
            ex = cnt[l] == 1
            brk = np.nonzero((np.diff(l) != 1) | (ex[1:] != ex[:-1]))[0] + 1
            a, b = np.r_[0, brk], np.r_[brk, l.size]
            direct = ex[a] & (b - a >= self.SIDE_RUN)
            runs = np.stack([a[direct], b[direct], l[a[direct]]], axis=1) if direct.any() else np.zeros((0, 3), dtype=np.int64)
            side = np.ones(l.size, dtype=bool)
            for pa, pb, _ in runs:
                side[pa:pb] = False
            plans.append((runs, np.nonzero(side)[0]))
        self._asm_locs, self._asm_plans = locs, plans
        self._side_off = np.r_[0, np.cumsum([p[1].size for p in plans])].astype(np.int64)
        mine = locs[self.rank]
        if self.ev is not None:
            mr = m[self.start:self.start + self.count]
            lm = np.full(mr.shape, -1, dtype=np.int32)
            sel = mr >= 0
            lm[sel] = np.searchsorted(mine, mr[sel]).astype(np.int32)
            self.ev.set_kkt_map(lm, int(mine.size))
        return self

    def alloc_assembled(self, device, barrier_group=None, tag: Optional[str] = None):
        """Device buffers of the assembled evaluation -- this rank's compact value array, its FX / AGX blocks -- and the shared
        page-locked host layout [values (nvalues) | side vectors | FX, AGX blocks of every rank]."""
        import torch
        mine, (runs, side_pos) = self._asm_locs[self.rank], self._asm_plans[self.rank]
        self._vals = torch.zeros(max(1, mine.size), dtype=torch.float64, device=device)
        self._side_pos = torch.from_numpy(side_pos).to(device)
        self._side_dev = torch.zeros(max(1, side_pos.size), dtype=torch.float64, device=device)
        self._fa = torch.zeros(self.max_count * (self.OR + self.IR), dtype=torch.float64, device=device)
        self._afx = self._fa[:self.max_count * self.OR].view(self.max_count, self.OR)
        self._aagx = self._fa[self.max_count * self.OR:].view(self.max_count, self.IR)
        self._fa_off = self._nvalues + int(self._side_off[-1])
        total = self._fa_off + self.world * self._fa.numel()
        self._hostv = HostSharedBlocks(0, self.rank, self.world, self.group, barrier_group, tag, total_doubles=total)
        self._ordered = True
        return self

    def eval_assembled_device(self, what: int, X, L=None, stream=None):
        """This rank's shard of an evalKKT: KKT entries summed into its compact value array, FX / AGX blocks beside them."""
        stream = self._eval_stream(stream)
        if self.ev is not None:
            import torch
            ctx = torch.cuda.stream(stream) if (stream is not None and not isinstance(stream, int) and torch.cuda.is_available()) else None
            if ctx is not None:
                with ctx:
                    self._vals.zero_()
            else:
                self._vals.zero_()
            self.ev.eval_assembled_device(what, X, L, self._afx, self._aagx if what in _ADJ else None, self._vals, stream)
        self._mark_evaluated(stream)

    def push_assembled(self):
        """Enqueue this rank's copies into the shared host layout (on the current stream, after the evaluation)."""
        import torch
        self._after_evaluation()
        host = self._hostv.buf
        runs, side_pos = self._asm_plans[self.rank]
        for pa, pb, loc in runs:                                 # long runs only this rank touches: straight to their place
            host[int(loc):int(loc) + int(pb - pa)].copy_(self._vals[int(pa):int(pb)], non_blocking=True)
        if side_pos.size:
            self._side_dev[:side_pos.size].copy_(self._vals.index_select(0, self._side_pos))   # one packed vector per rank
            o = self._nvalues + int(self._side_off[self.rank])
            host[o:o + side_pos.size].copy_(self._side_dev[:side_pos.size], non_blocking=True)
        o = self._fa_off + self.rank * self._fa.numel()
        host[o:o + self._fa.numel()].copy_(self._fa, non_blocking=True)

    def wait_assembled(self, stream=None, dst: int = 0):
        """Completes this rank's copies, meets the other ranks, and -- on `dst` -- adds the side vectors into the value array.
        Afterwards ``host_values()`` holds the constraint's whole contribution at its locations (other locations untouched)."""
        self._hostv.wait(stream)
        if self.rank == dst:
            host = self._hostv.buf.numpy()
            vals = host[:self._nvalues]
            sides = [(self._asm_locs[r][self._asm_plans[r][1]], host[self._nvalues + int(self._side_off[r]):self._nvalues + int(self._side_off[r + 1])])
                     for r in range(self.world) if self._asm_plans[r][1].size]
            for l, _ in sides:
                vals[l] = 0.0
            for l, v in sides:                                   # (a rank's side locations are distinct: a plain indexed +=)
                vals[l] += v
        self._hostv._barrier()                                   # (nobody reads, nobody overwrites, before the root is done)

    def host_values(self):
        return self._hostv.buf.numpy()[:self._nvalues]

    def host_assembled_blocks(self):
        """[(fx, agx)] per rank from the shared host layout."""
        out, n = [], self._fa.numel()
        host = self._hostv.buf.numpy()
        for r, (_, c) in enumerate(self.shards):
            f = host[self._fa_off + r * n:self._fa_off + (r + 1) * n]
            out.append((f[:self.max_count * self.OR].reshape(self.max_count, self.OR)[:c],
                        f[self.max_count * self.OR:].reshape(self.max_count, self.IR)[:c]))
        return out

    # ---- exchange of host blocks (kept for callers that evaluate through host pointers) -----
    def gather_blocks(self, blocks, dst: int = 0, device=None):
        """Gather one kind of block ([count, width] array/tensor) from every rank to `dst`.

        Shards differ by at most one segment: each rank pads to the largest shard, `dst` trims.  Returns the
        concatenated [nseg_total, width] tensor on `dst`, None elsewhere."""
        import torch
        t = torch.as_tensor(blocks)
        if device is not None:
            t = t.to(device)
        width = t.shape[1]
        pad = torch.zeros((self.max_count, width), dtype=t.dtype, device=t.device)
        pad[: t.shape[0]] = t
        out = [torch.empty_like(pad) for _ in range(self.world)] if self.rank == dst else None
        self.dist.gather(pad, out, dst=dst, group=self.group)
        if self.rank != dst:
            return None
        return torch.cat([o[:c] for o, (_, c) in zip(out, self.shards)], dim=0)


class PhaseShardedEvaluator(_StreamOrder):
    """Several phases of one problem (same ODE / transcription / size), whole phases per rank: phase k belongs to rank
    k % world.  `phases` = list of (vindex, cindex) tables into the problem's X and L.

    A rank's phases are evaluated as ONE function: the defect evaluator takes a variable / constraint index table per
    application and nothing in it knows about phases, so the tables of the local phases are concatenated and the whole
    rank is one launch (measured on one MI355X, 8 phases x 1 250 Reentry-LGL7 segments: 141.9 us as eight launches on one
    stream, 138.9 us on eight streams joined by events, 168.5 us forked and joined inside one C call -- a 1 250-segment
    launch is latency-bound at ~17 us and cross-stream events cost more than they hide -- against one merged launch,
    DESIGN.md section 6)."""

    def __init__(self, ode: str, mode, blocked: bool, phases: Sequence, n_primal: int, n_equal: int,
                 rank: Optional[int] = None, world: Optional[int] = None, device: int = 0, group=None,
                 evaluator_factory: Optional[Callable] = None):
        import torch.distributed as dist
        self.dist, self.group = dist, group
        self.rank = dist.get_rank(group) if rank is None else rank
        self.world = dist.get_world_size(group) if world is None else world
        self.nphases = len(phases)
        self.nseg = int(np.asarray(phases[0][0]).shape[0])
        if any(np.asarray(v).shape[0] != self.nseg for v, _ in phases):
            raise ValueError("PhaseShardedEvaluator: phases must have the same number of segments")
        self.owner = [k % self.world for k in range(self.nphases)]
        self.mine = [k for k in range(self.nphases) if self.owner[k] == self.rank]
        self.per_rank = -(-self.nphases // self.world)       # slots per rank (the last ranks may leave one unused)
        if evaluator_factory is None:
            from .evaluator import DefectEvaluator
            evaluator_factory = DefectEvaluator
        self.ev = None
        if self.mine:
            vix = np.concatenate([np.asarray(phases[k][0]) for k in self.mine], axis=0)
            cix = np.concatenate([np.asarray(phases[k][1]) for k in self.mine], axis=0)
            self.ev = evaluator_factory(ode, mode, blocked, vix, cix, n_primal, n_equal, device)
            self.IR, self.OR, self.NKKT = self.ev.IR, self.ev.OR, self.ev.NKKT
            self.KSTRIDE = getattr(self.ev, "KSTRIDE", self.NKKT)
        else:
            self.IR, self.OR, self.NKKT, self.KSTRIDE = _sizes(ode, mode, blocked)
        self._local = self._recv = None

    def kkt_to_reference(self, kkt):
        if self.ev is not None and hasattr(self.ev, "kkt_to_reference"):
            return self.ev.kkt_to_reference(kkt)
        return kkt

    def _flat_doubles(self, nlocal: int) -> int:
        m = nlocal * self.nseg
        return _al16(m * (self.OR + self.IR)) + m * self.KSTRIDE

    def _nlocal(self, rank: int) -> int:
        return len(range(rank, self.nphases, self.world))

    def _views(self, flat, nlocal: int):
        """(fx, agx, kkt) of a rank's flat buffer: [fx | agx | kkt], each over that rank's nlocal * nseg applications."""
        m = nlocal * self.nseg
        o1, o2, o3 = m * self.OR, m * (self.OR + self.IR), _al16(m * (self.OR + self.IR))
        return flat[:o1].view(m, self.OR), flat[o1:o2].view(m, self.IR), flat[o3:o3 + m * self.KSTRIDE].view(m, self.KSTRIDE)

    def _phase_views(self, flat, nlocal: int, slot: int):
        lo, hi = slot * self.nseg, (slot + 1) * self.nseg
        return tuple(v[lo:hi] for v in self._views(flat, nlocal))

    def alloc_device(self, device, dst: int = 0, always_exchange: bool = False):
        import torch
        self._dst = dst
        self._local = torch.zeros(self._flat_doubles(self.per_rank), dtype=torch.float64, device=device)
        if self.rank == dst and (self.world > 1 or always_exchange):
            self._recv = torch.empty((self.world, self._flat_doubles(self.per_rank)), dtype=torch.float64, device=device)
        self._ordered = self.world > 1 or always_exchange
        return self

    def eval_device(self, what: int, X, L=None, stream=None):
        stream = self._eval_stream(stream)
        if self.ev is not None:
            fx, agx, kkt = self._views(self._local, len(self.mine))
            self.ev.eval_device(what, X, L, fx, agx if what in _ADJ else None, kkt if what >= 2 else None, stream)
        self._mark_evaluated(stream)

    def gather_device(self, async_op: bool = False):
        if self.world == 1 and self._recv is None:
            return None
        self._after_evaluation()
        out = list(self._recv.unbind(0)) if self.rank == self._dst else None
        return self.dist.gather(self._local, out, dst=self._dst, group=self.group, async_op=async_op)

    def blocks_on_root(self):
        """List over phases of (fx, agx, kkt) views on the root (None elsewhere)."""
        if self._recv is None:
            if self.world != 1:
                return None
            return [self._phase_views(self._local, self.nphases, k) for k in range(self.nphases)]
        return [self._phase_views(self._recv[self.owner[k]], self._nlocal(self.owner[k]), k // self.world)
                for k in range(self.nphases)]
