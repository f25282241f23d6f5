"""Multi-GPU sharding of a phase's defect constraint: one process per GPU (torch.distributed, RCCL on ROCm).

Segments are independent given X and L (every application reads only its own Vindex/Cindex column,
/root/reference/src/VectorFunctions/DenseFunctionBase.h:1296-1312), so the evaluation itself needs no
collective: rank r evaluates the contiguous range the reference's ByApplication rule would give thread r
(/root/reference/src/VectorFunctions/IndexingData.h:117-146).  The only exchange step is the optional gather of
the per-shard FX / AGX / KKT blocks to the rank that owns the host KKT system (`gather_blocks`): disjoint
blocks, no reduction -- boundary-node Hessian entries are emitted by both neighbours and summed by the
scatter, exactly as on one device.
"""
from __future__ import annotations

from typing import Callable, Optional

import numpy as np

from .indexing import thread_split


class ShardedDefectEvaluator:
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
        else:
            from .build import dims
            from . import _lib, synth
            d = dims(*_lib.ode_sizes(ode), synth.MODE_CS[mode] if isinstance(mode, str) else max(mode, 2), blocked)
            self.IR, self.OR, self.NKKT = d["IR"], d["OR"], d["NKKT"]

    # ---- local evaluation ------------------------------------------------------------------
    def eval_local(self, what: int, X, L=None):
        """Host-pointer evaluation of this rank's shard -> (fx, agx, kkt) numpy blocks (or empty arrays)."""
        if self.ev is None:
            return (np.zeros((0, self.OR)), np.zeros((0, self.IR)), np.zeros((0, self.NKKT)))
        return self.ev.eval(what, X, L)

    # ---- exchange --------------------------------------------------------------------------
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
