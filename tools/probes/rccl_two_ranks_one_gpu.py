"""Probe (GPU box, one device): does RCCL accept TWO ranks on the same MI355X?  If it does, the packed all-gather of the
sharded step can be shown with world_size 2 over the real backend; if it refuses (NCCL's "Duplicate GPU detected"), the
message is the evidence.  Run: python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1
--master-port 29631 tools/probes/rccl_two_ranks_one_gpu.py"""
import os
import sys

import torch
import torch.distributed as dist

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
try:
    dist.init_process_group("nccl", device_id=dev)
    x = torch.full((1024,), rank + 1, dtype=torch.int32, device=dev)
    out = torch.empty(world * 1024, dtype=torch.int32, device=dev)
    dist.all_gather_into_tensor(out, x)
    torch.cuda.synchronize()
    ok = all(int(out[r * 1024]) == r + 1 for r in range(world))
    print(f"rank {rank}: all_gather_into_tensor over RCCL with {world} ranks on one device: {'ok' if ok else 'WRONG'}", flush=True)
    dist.destroy_process_group()
except Exception as e:  # noqa: BLE001 - the message is the result
    print(f"rank {rank}: RCCL refused: {type(e).__name__}: {str(e)[:600]}", flush=True)
    sys.exit(3)
