"""Child of tests/test_gpu_distributed.py: launched by torch.distributed.run with two processes that share GPU 0 (gloo
backend -- RCCL refuses two ranks on one device): real kernels on real shards, then both exchanges against the oracle."""
import os, sys, json
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch.distributed as dist
rank=int(os.environ["RANK"]); world=int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
dist.init_process_group("gloo")
from helpers import Workload, rel_err
from asset_asrl_amd.distributed import ShardedDefectEvaluator
from oracle import bindings as ob
w = Workload("reentry", "LGL7", 333)
sh = ShardedDefectEvaluator("reentry", "LGL7", False, w.vindex, w.cindex, w.n_primal, w.n_equal, device=0)
sh.alloc_device(torch.device("cuda", 0))
X, L = torch.from_numpy(w.X).cuda(), torch.from_numpy(w.L).cuda()
sh.eval_device(4, X, L)
torch.cuda.synchronize()
try:
    sh.gather_device()
    torch.cuda.synchronize()
    blocks = sh.blocks_on_root()
    how="gloo gather of device tensors"
except Exception as e:
    print("gather of CUDA tensors under gloo failed:", str(e)[:200]); blocks=None; how="n/a"
sh.alloc_host_shared()
sh.push_host(); sh.wait_host()
hs = sh.host_shard_blocks()
if rank == 0:
    rfx, ragx, rkkt = w.oracle_nlp(ob, threads=4).eval_blocks(4, w.X, w.L)
    hk = np.concatenate([h[2] for h in hs]); hf=np.concatenate([h[0] for h in hs]); ha=np.concatenate([h[1] for h in hs])
    hk = sh.kkt_to_reference(hk)                      # (the blocks travel in the kernels' layout)
    out={"host_kkt": rel_err(hk, rkkt), "host_agx": rel_err(ha, ragx), "host_fx": float(np.abs(hf-rfx).max()), "shards": sh.shards, "how": how}
    if blocks is not None:
        out["gather_kkt"]=rel_err(sh.kkt_to_reference(blocks[2].cpu().numpy()), rkkt)
sh._host.close()
# sharded on-device assembly: every rank's compact value array pushed into the shared value array, against ONE device's
# asset_hip_defect_eval_assembled_zeroed, bit for bit
from asset_asrl_amd.indexing import kkt_slot_locations
locs, nnz = kkt_slot_locations(w.vindex, w.cindex, w.n_primal)
sh.SIDE_RUN = 1024
sh.set_kkt_map(locs, nnz).alloc_assembled(torch.device("cuda", 0))
for _ in range(2):
    sh.eval_assembled_device(4, X, L)
    sh.push_assembled()
    sh.wait_assembled()
if rank == 0:
    from asset_asrl_amd.evaluator import DefectEvaluator
    ev = DefectEvaluator("reentry", "LGL7", False, w.vindex, w.cindex, w.n_primal, w.n_equal)
    ev.set_kkt_map(locs, nnz)
    ref = np.zeros(nnz)
    ev.eval_assembled(4, w.X, w.L, ref, target_zeroed=True)
    got = sh.host_values()
    out["asm_same"] = bool(np.array_equal(got, ref))
    out["asm_err"] = rel_err(got, ref)
    out["asm_direct_runs"] = [int(p[0].shape[0]) for p in sh._asm_plans]
    print(json.dumps(out))
sh._hostv.close()
dist.barrier(); dist.destroy_process_group()
