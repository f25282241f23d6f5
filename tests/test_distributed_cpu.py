"""world_size-2 gloo test (CPU) of the N>1 path: shard rule, per-rank index slicing and the gather of disjoint blocks.
The per-shard evaluator is replaced by the oracle (tests may use it) because no GPU exists here."""
import os
import socket

import numpy as np
import pytest

from helpers import Workload


class _OracleShard:
    """Stands in for DefectEvaluator on CPU: same constructor/eval signature, blocks from the oracle."""

    def __init__(self, ode, mode, blocked, vindex, cindex, n_primal, n_equal, device=0):
        from oracle import bindings as ob
        self.ob = ob
        self.nlp = ob.Nlp(ob.get_ode(ode, 0), ob.MODES[mode], blocked, vindex, cindex, n_primal, n_equal, 1)
        self.IR, self.OR, self.NKKT = self.nlp.ir, self.nlp.orr, self.nlp.nkkt

    def eval(self, what, X, L=None):
        return self.nlp.eval_blocks(what, X, L)


def _worker(rank, world, port, nseg, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from asset_asrl_amd.distributed import ShardedDefectEvaluator
    w = Workload("reentry", "LGL5", nseg)
    sh = ShardedDefectEvaluator("reentry", "LGL5", False, w.vindex, w.cindex, w.n_primal, w.n_equal,
                                evaluator_factory=_OracleShard)
    fx, agx, kkt = sh.eval_local(4, w.X, w.L)
    got = [sh.gather_blocks(b, dst=0) for b in (fx, agx, kkt)]
    if rank == 0:
        q.put([g.numpy() for g in got] + [sh.shards])
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("nseg", [9, 1])
def test_two_rank_shard_and_gather(oracle, nseg):
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, nseg, q)) for r in range(2)]
    for p in procs:
        p.start()
    fx, agx, kkt, shards = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    w = Workload("reentry", "LGL5", nseg)
    rfx, ragx, rkkt = w.oracle_nlp(oracle).eval_blocks(oracle.JAC_ADJGRAD_HESS, w.X, w.L)
    assert [c for _, c in shards] == ([5, 4] if nseg == 9 else [1, 0])
    np.testing.assert_array_equal(fx, rfx)
    np.testing.assert_array_equal(agx, ragx)
    np.testing.assert_array_equal(kkt, rkkt)
