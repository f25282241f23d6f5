"""The device-resident sharded path on a real GPU: one rank, RCCL backend, exchange forced (a gather in a world of one
rank still goes through c10d's NCCL path), against the oracle.  Runs in a child process so that the process group
does not outlive the test."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_CHILD = r"""
import os, sys, json
import numpy as np, torch
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests"))
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", RANK="0", WORLD_SIZE="1")
import torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
from helpers import Workload, rel_err
from asset_asrl_amd.distributed import ShardedDefectEvaluator
from oracle import bindings as ob
w = Workload("reentry", "LGL7", 333)
sh = ShardedDefectEvaluator("reentry", "LGL7", False, w.vindex, w.cindex, w.n_primal, w.n_equal, device=0)
sh.alloc_device(torch.device("cuda", 0), always_exchange=True)
X, L = torch.from_numpy(w.X).cuda(), torch.from_numpy(w.L).cuda()
st = torch.cuda.Stream()
with torch.cuda.stream(st):
    for _ in range(2):
        sh.eval_device(4, X, L, st)
        sh.gather_device()
torch.cuda.synchronize()
fx, agx, kkt = [t.cpu().numpy() for t in sh.blocks_on_root()]
rfx, ragx, rkkt = w.oracle_nlp(ob, threads=4).eval_blocks(4, w.X, w.L)
# evaluation and exchange on DIFFERENT streams, no stream argument: the exchange must still see the finished blocks
sA, sB = torch.cuda.Stream(), torch.cuda.Stream()
sh._local.zero_(); sh._recv.zero_()
torch.cuda.synchronize()
with torch.cuda.stream(sA):
    sh.eval_device(4, X, L)
with torch.cuda.stream(sB):
    sh.gather_device()
torch.cuda.synchronize()
cross = bool(all(np.array_equal(a.cpu().numpy(), b) for a, b in zip(sh.blocks_on_root(), (fx, agx, kkt))))
# the same on torch's DEFAULT stream (handle 0, which the C ABI reads as "the handle's own stream": the evaluator names
# the null stream explicitly, hipStreamLegacy) -- evaluation and gather with no stream argument and no stream context
sh._local.zero_(); sh._recv.zero_()
torch.cuda.synchronize()
for _ in range(3):
    sh.eval_device(4, X, L)
    sh.gather_device()
torch.cuda.synchronize()
default_stream = bool(all(np.array_equal(a.cpu().numpy(), b) for a, b in zip(sh.blocks_on_root(), (fx, agx, kkt))))
# host-visible exchange: the flat buffer copied into this rank's range of the shared page-locked host buffer
sh.alloc_host_shared()
with torch.cuda.stream(st):
    sh.eval_device(4, X, L, st)
    sh.push_host()
    sh.wait_host(st)
hfx, hagx, hkkt = sh.host_shard_blocks()[0]
host_same = bool(np.array_equal(hfx, fx) and np.array_equal(hagx, agx) and np.array_equal(hkkt, kkt))
sh._host.close()
print(json.dumps({{"fx": float(np.abs(fx - rfx).max() / max(1.0, np.abs(w.X).max())), "agx": rel_err(agx, ragx),
                  "kkt": rel_err(sh.kkt_to_reference(kkt), rkkt), "shape": list(kkt.shape), "stride": sh.KSTRIDE, "host_same": host_same, "cross_stream": cross,
                  "default_stream": default_stream}}))
dist.destroy_process_group()
"""


@pytest.mark.gpu
def test_sharded_device_path_with_rccl_gather_matches_the_oracle(oracle):
    r = subprocess.run([sys.executable, "-c", _CHILD.format(root=ROOT)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert out["shape"] == [333, out["stride"]] and out["stride"] >= 1008     # (the blocks travel in the kernels' layout)
    assert out["fx"] < 1e-10 and out["agx"] < 1e-8 and out["kkt"] < 1e-8, out
    assert out["host_same"]                  # the host-shared exchange delivers the same bits
    assert out["cross_stream"]               # evaluation on one stream, gather on another: ordered by the evaluator
    assert out["default_stream"]             # ... and both on torch's default stream


@pytest.mark.gpu
def test_bench_runs_the_exchange_path_with_one_rank():
    env = dict(os.environ, ASSET_BENCH_FORCE_DIST="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "5", "--warmup", "2", "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    assert r.stdout.strip().splitlines()[-1].startswith("{"), r.stdout[-500:]      # the JSON line is the last line
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert out["scaling"] == "strong" and out["n_gpus"] == 1 and out["value"] > 0
    assert "exchange_ms" in out and out["exchange_bytes_into_root"] > 8e7
    assert out["host_visible"]["ms_per_step"] > out["ms_per_step"] and out["host_visible"]["bytes_per_rank"] > 8e7
    assert len(out["per_rank_roofline_frac"]) == 1 and 0.05 < out["per_rank_roofline_frac"][0] < 1.0


@pytest.mark.gpu
def test_two_ranks_sharing_one_gpu_match_the_oracle(oracle):
    """World size 2 with real kernels: two processes on GPU 0 (gloo; the shard rule gives 167 + 166 segments), the device
    gather and the host-shared exchange both deliver the oracle's blocks."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                        "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "tests", "two_rank_one_gpu.py")],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert out["shards"] == [[0, 167], [167, 166]]
    assert out["host_fx"] < 1e-10 and out["host_agx"] < 1e-8 and out["host_kkt"] < 1e-8, out
    assert out.get("gather_kkt", 1.0) < 1e-8, out
    # sharded on-device assembly (f-1 x e): bitwise the single-GPU assembled values, direct copies and side vectors both used
    assert out["asm_same"], out
    assert sum(out["asm_direct_runs"]) >= 2
