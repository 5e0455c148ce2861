"""Can two ranks share one GPU under RCCL?  (1-GPU boxes are all we can launch; if this works the multi-rank device path can be
exercised for real.)  torchrun --nproc-per-node 2 scripts/rccl_same_gpu_probe.py"""
import os, torch, torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda:0"))
x = torch.full((1024,), dist.get_rank() + 1, device="cuda:0", dtype=torch.int32)
dist.all_reduce(x)
torch.cuda.synchronize()
print("rank", dist.get_rank(), "sum", int(x[0]))
dist.destroy_process_group()
