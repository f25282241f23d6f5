"""world_size-2 gloo tests (CPU) of the N>1 path.  Everything of asset_asrl_amd/distributed.py runs for real -- the shard
rule, per-rank index slicing, the flat [fx | agx | kkt] buffers the kernels write in place, the single gather of those
buffers to the root, the per-shard views and the phase-order concatenation, the round-robin deal of whole phases --
except the kernel launch itself: no GPU exists here, so the per-shard evaluator is a stand-in with DefectEvaluator's
constructor / eval / eval_device signature whose blocks come from the oracle (tests may use it)."""
import os
import socket

import numpy as np
import pytest

from helpers import Workload


from asset_asrl_amd.evaluator import _KktLayout


class _OracleShard(_KktLayout):
    """Stands in for DefectEvaluator on CPU: same constructor / eval / eval_device signatures, blocks from the oracle -- written into
    the device-side arrays in the layout the library's kernels for this shape use (asset_hip_kkt_layout: no device needed)."""

    def __init__(self, ode, mode, blocked, vindex, cindex, n_primal, n_equal, device=0):
        from oracle import bindings as ob
        from asset_asrl_amd import _lib
        self.ob = ob
        self.nlp = ob.Nlp(ob.get_ode(ode, 0), ob.MODES[mode], blocked, vindex, cindex, n_primal, n_equal, 1)
        self.IR, self.OR, self.NKKT = self.nlp.ir, self.nlp.orr, self.nlp.nkkt
        self.nseg = self.nlp.nappl
        kl, nk, stride, rows, cols = _lib.kkt_layout(ode, _lib.MODES[mode], blocked)
        assert nk == self.NKKT
        self._set_layout(kl, stride, rows, cols)

    def eval(self, what, X, L=None):
        return self.nlp.eval_blocks(what, X, L)

    def eval_device(self, what, X, L, fx, agx, kkt, stream=None):
        """Writes the blocks IN PLACE into the (here: CPU) tensors it is handed, like the kernels do."""
        import torch
        rfx, ragx, rkkt = self.nlp.eval_blocks(what, X.numpy(), None if L is None else L.numpy())
        fx[: self.nseg].copy_(torch.from_numpy(rfx))
        if agx is not None:
            agx[: self.nseg].copy_(torch.from_numpy(ragx))
        if kkt is not None and rkkt is not None:
            native = np.zeros((self.nseg, self.KSTRIDE))
            native[:, self.kkt_perm] = rkkt
            kkt[: self.nseg].copy_(torch.from_numpy(native))


    # on-device assembly, stand-in: the oracle's blocks added into the (compact) value array through the map the sharded
    # evaluator hands over -- what asset_hip_defect_set_kkt_map / _eval_assembled_device do on a GPU
    def set_kkt_map(self, local_map, nvalues):
        self._map, self._nv = np.asarray(local_map), int(nvalues)

    def eval_assembled_device(self, what, X, L, fx, agx, vals, stream=None):
        import torch
        rfx, ragx, rkkt = self.nlp.eval_blocks(what, X.numpy(), None if L is None else L.numpy())
        fx[: self.nseg].copy_(torch.from_numpy(rfx))
        if agx is not None:
            agx[: self.nseg].copy_(torch.from_numpy(ragx))
        v = vals.numpy()
        assert not v[: self._nv].any()                               # (the caller hands over zeros)
        sel = self._map >= 0
        np.add.at(v, self._map[sel], rkkt[sel])


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _init(rank, world, port):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    return dist


def _worker(rank, world, port, nseg, q):
    import torch
    dist = _init(rank, world, port)
    from asset_asrl_amd.distributed import ShardedDefectEvaluator
    w = Workload("reentry", "LGL5", nseg)
    sh = ShardedDefectEvaluator("reentry", "LGL5", False, w.vindex, w.cindex, w.n_primal, w.n_equal,
                                evaluator_factory=_OracleShard)
    # host-block path
    fx, agx, kkt = sh.eval_local(4, w.X, w.L)
    got = [sh.gather_blocks(b, dst=0) for b in (fx, agx, kkt)]
    # device-resident path (CPU tensors under gloo): in-place evaluation into the flat buffer, ONE gather
    sh.alloc_device(torch.device("cpu"))
    X, L = torch.from_numpy(w.X), torch.from_numpy(w.L)
    for _ in range(2):                                    # twice: the buffers are reused every solver iteration
        lfx, lagx, lkkt = sh.eval_device(4, X, L)
        sh.gather_device()
    assert lfx.shape[0] == sh.count and lkkt.shape == (sh.count, sh.KSTRIDE)
    assert sh._local.numel() == (sh.max_count * (sh.OR + sh.IR) + 15) // 16 * 16 + sh.max_count * sh.KSTRIDE
    shards = sh.shard_blocks_on_root()
    full = sh.blocks_on_root()
    # host-visible exchange: every rank copies its flat buffer into its range of one shared host buffer, then a barrier
    sh.alloc_host_shared()
    sh.eval_device(4, X, L)
    sh.push_host()
    sh.wait_host()
    hs = sh.host_shard_blocks()
    assert len(hs) == world and not os.path.exists(sh._host.path)     # (the name is unlinked once every rank has mapped it)
    host_full = [np.concatenate([h[k] for h in hs], axis=0) for k in range(3)]   # read on EVERY rank: it is host memory
    for got_h, loc in zip(hs[rank], (lfx, lagx, lkkt)):
        np.testing.assert_array_equal(got_h, loc.numpy())
    if rank == 0:
        host_full[2] = sh.kkt_to_reference(host_full[2])                  # (the blocks travel in the kernels' layout)
        q.put(("host", host_full))
    sh._host.close()
    if rank == 0:
        assert len(shards) == world and all(s[2].shape[0] == c for s, (_, c) in zip(shards, sh.shards))
        assert shards[0][0].data_ptr() == sh._recv[0].data_ptr()          # views of the receive buffer, no copy
        q.put([g.numpy() for g in got] + [sh.shards] + [full[0].numpy(), full[1].numpy(), sh.kkt_to_reference(full[2])])
    else:
        assert shards is None and full is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("nseg,world", [(9, 2), (1, 2), (10, 4), (3, 4)])   # uneven shards; fewer segments than ranks
def test_shard_and_gather(oracle, nseg, world):
    import torch.multiprocessing as mp
    from asset_asrl_amd.indexing import thread_split
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, nseg, q)) for r in range(world)]
    for p in procs:
        p.start()
    tagged = q.get(timeout=120)
    assert tagged[0] == "host"
    hfx, hagx, hkkt = tagged[1]
    fx, agx, kkt, shards, dfx, dagx, dkkt = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    w = Workload("reentry", "LGL5", nseg)
    rfx, ragx, rkkt = w.oracle_nlp(oracle).eval_blocks(oracle.JAC_ADJGRAD_HESS, w.X, w.L)
    expect = [c for _, c in thread_split(nseg, world)]
    assert [c for _, c in shards] == expect + [0] * (world - len(expect))
    assert {(9, 2): [5, 4], (1, 2): [1, 0], (10, 4): [3, 3, 2, 2], (3, 4): [1, 1, 1, 0]}[(nseg, world)] == [c for _, c in shards]
    for got, ref in ((fx, rfx), (agx, ragx), (kkt, rkkt), (dfx, rfx), (dagx, ragx), (dkkt, rkkt), (hfx, rfx), (hagx, ragx),
                     (hkkt, rkkt)):
        np.testing.assert_array_equal(got, ref)


def _phase_worker(rank, world, port, nphases, nseg, q):
    import torch
    dist = _init(rank, world, port)
    from asset_asrl_amd.distributed import PhaseShardedEvaluator
    # the phases placed in one solver vector by the OptimalControlProblem mirror (OptimalControlProblem.cpp:115-155) ...
    from asset_asrl_amd.ocp import OptimalControlProblem
    from asset_asrl_amd.ode import TwoBody
    ocp, ws, voff, coff = OptimalControlProblem(), [], 0, 0
    for k in range(nphases):
        wk = Workload("twobody_lt", "LGL5", nseg, True, seed=100 + k, var_offset=voff, con_offset=coff)
        ws.append(wk)
        voff, coff = wk.n_primal, wk.n_equal
        ph = TwoBody().phase("LGL5", wk.traj, nseg)
        ph.setControlMode("BlockConstant")
        ph.EnableMeshSpacing = False
        ocp.addPhase(ph)
    n_primal, n_equal = ocp.n_primal, ocp.n_equal
    assert (n_primal, n_equal) == (ws[-1].n_primal, ws[-1].n_equal)     # ... where offsets rolled by hand put them
    for wk, (V, Cx) in zip(ws, ocp.defect_tables()):
        np.testing.assert_array_equal(V, wk.vindex)
        np.testing.assert_array_equal(Cx, wk.cindex)
    X, L = ocp.solver_input(), np.zeros(n_equal)
    for wk in ws:
        c0 = wk.indexer.con_offset
        L[c0:c0 + wk.indexer.numPhaseEqCons] = wk.L[c0:c0 + wk.indexer.numPhaseEqCons]
    sh = ocp.phase_sharded_evaluator(evaluator_factory=_OracleShard)
    assert sh.mine == list(range(rank, nphases, world))
    sh.alloc_device(torch.device("cpu"))
    sh.eval_device(4, torch.from_numpy(X), torch.from_numpy(L))
    sh.gather_device()
    blocks = sh.blocks_on_root()
    if rank == 0:
        from oracle import bindings as ob
        for k, wk in enumerate(ws):
            ref = ob.Nlp(ob.get_ode("twobody_lt", 0), ob.MODES["LGL5"], True, wk.vindex, wk.cindex, n_primal, n_equal,
                         1).eval_blocks(4, X, L)
            for got, r in zip((blocks[k][0].numpy(), blocks[k][1].numpy(), sh.kkt_to_reference(blocks[k][2])), ref):
                np.testing.assert_array_equal(got, r)
        q.put("ok")
    else:
        assert blocks is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("nphases,world", [(3, 2), (5, 4), (3, 4)])
def test_phase_deal_and_gather(oracle, nphases, world):
    """BASELINE.json configs[3] shape: linked phases dealt round-robin (three phases on two ranks: the last slot of
    rank 1 stays unused; five on four; fewer phases than ranks)."""
    import torch.multiprocessing as mp
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_phase_worker, args=(r, world, port, nphases, 6, q)) for r in range(world)]
    for p in procs:
        p.start()
    assert q.get(timeout=120) == "ok"
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0


def _failing_worker(rank, world, port, q):
    """HostSharedBlocks when ONE rank cannot map the buffer: every rank raises, nobody hangs, the name is removed."""
    import torch
    dist = _init(rank, world, port)
    from asset_asrl_amd import distributed as D
    if rank == 1:
        real = torch.from_file

        def broken(*a, **k):
            raise OSError("no mapping for you")
        torch.from_file = broken
    try:
        D.HostSharedBlocks(64, rank, world, tag=f"asset_hip_test_fail_{port}")
        q.put((rank, "constructed"))
    except RuntimeError as exc:
        q.put((rank, "raised", str(exc)))
    if rank == 1:
        torch.from_file = real
    dist.barrier()
    dist.destroy_process_group()


def test_host_shared_blocks_fail_on_every_rank_together():
    import torch.multiprocessing as mp
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_failing_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert [g[1] for g in got] == ["raised", "raised"], got
    assert "another rank" in got[0][2] and "this rank" in got[1][2]
    assert not os.path.exists(f"/dev/shm/asset_hip_test_fail_{port}")


def _asm_worker(rank, world, port, nseg, side_run, q):
    import torch
    dist = _init(rank, world, port)
    from asset_asrl_amd.distributed import ShardedDefectEvaluator
    from asset_asrl_amd.indexing import kkt_slot_locations
    w = Workload("reentry", "LGL5", nseg)
    sh = ShardedDefectEvaluator("reentry", "LGL5", False, w.vindex, w.cindex, w.n_primal, w.n_equal,
                                evaluator_factory=_OracleShard)
    sh.SIDE_RUN = side_run
    locs, nnz = kkt_slot_locations(w.vindex, w.cindex, w.n_primal)
    nvalues = nnz + 11                                                # (the solver's own slots behind the constraint's)
    sh.set_kkt_map(locs, nvalues).alloc_assembled(torch.device("cpu"))
    X, L = torch.from_numpy(w.X), torch.from_numpy(w.L)
    for it in range(2):                                               # twice: every solver iteration re-uses the layout
        if rank == 0:
            sh.host_values()[:] = 7.0                                 # what other functions / the solver left there
        dist.barrier()
        sh.eval_assembled_device(4, X, L)
        sh.push_assembled()
        sh.wait_assembled()
    if rank == 0:
        q.put((sh.host_values().copy(), [tuple(a.copy() for a in b) for b in sh.host_assembled_blocks()], sh.shards,
               [int(p[0].shape[0]) for p in sh._asm_plans]))
    sh._hostv.close()
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("nseg,world,side_run", [(9, 2, 4096), (10, 4, 16), (3, 4, 4), (40, 4, 64)])
def test_sharded_assembly_on_the_device_is_bitwise_the_single_scatter(oracle, nseg, world, side_run):
    """SURVEY section 8 rows f-1 x e: every rank assembles its shard, pushes its long runs straight into the shared value array
    and the rest as one side vector; the root's result is bit for bit the scatter of all blocks by one process
    (NonLinearProgram.cpp:316-330, DenseFunctionBase.h:1449-1465), other locations keep what they held."""
    import torch.multiprocessing as mp
    from asset_asrl_amd.indexing import kkt_slot_locations
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_asm_worker, args=(r, world, port, nseg, side_run, q)) for r in range(world)]
    for p in procs:
        p.start()
    vals, blocks, shards, ndirect = q.get(timeout=180)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    w = Workload("reentry", "LGL5", nseg)
    rfx, ragx, rkkt = w.oracle_nlp(oracle).eval_blocks(oracle.JAC_ADJGRAD_HESS, w.X, w.L)
    locs, nnz = kkt_slot_locations(w.vindex, w.cindex, w.n_primal)
    ref = np.full(nnz + 11, 7.0)
    ref[np.unique(locs)] = 0.0
    np.add.at(ref, locs.ravel(), rkkt.ravel())                        # one process, segment order
    np.testing.assert_array_equal(vals, ref)
    np.testing.assert_array_equal(np.concatenate([b[0] for b in blocks]), rfx)
    np.testing.assert_array_equal(np.concatenate([b[1] for b in blocks]), ragx)
    if side_run <= 64 and nseg >= 10:
        assert sum(ndirect) > 0                                       # the direct copies were exercised too
