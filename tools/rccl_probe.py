"""Can two ranks share ONE GPU under RCCL on this box?  (torchrun --nproc-per-node 2 tools/rccl_probe.py)
If yes, the two-rank tests of the overlapped gradient exchange can run on the real backend instead of gloo."""
import os
import torch
import torch.distributed as dist

rank = int(os.environ["RANK"])
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=rank, world_size=int(os.environ["WORLD_SIZE"]))
t = torch.full([1 << 20], float(rank + 1), device="cuda", dtype=torch.bfloat16)
dist.all_reduce(t)
torch.cuda.synchronize()
print(f"[rccl_probe] rank {rank}: sum = {float(t[0])}", flush=True)
dist.destroy_process_group()
