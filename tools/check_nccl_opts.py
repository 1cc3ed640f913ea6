"""One-rank RCCL process group with the high-priority stream option: accepted by this torch build, an all-reduce runs."""
import os, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29511")
torch.cuda.set_device(0)
opts = dist.ProcessGroupNCCL.Options(is_high_priority_stream=True)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0), pg_options=opts)
x = torch.ones(1 << 20, device="cuda")
w = dist.all_reduce(x, async_op=True); w.wait(); torch.cuda.synchronize()
print("ok", float(x.sum()), torch.cuda.nccl.version())
dist.destroy_process_group()
