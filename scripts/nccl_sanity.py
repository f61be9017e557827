"""World-size-1 sanity run of the torch.distributed (backend nccl = RCCL) calls the multi-GPU path makes outside the
halo pushes: process-group creation with device_id, uint8 all_gather (hipIpc handle swap), int32 MIN all_reduce
(transport votes), float64 all_reduce (global_sum), barrier.  Point-to-point needs >= 2 ranks and is not covered."""
import os
import torch
import torch.distributed as dist

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29544")
os.environ.setdefault("RANK", "0")
os.environ.setdefault("WORLD_SIZE", "1")
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
mine = torch.arange(128, dtype=torch.uint8).cuda()
parts = [torch.empty_like(mine)]
dist.all_gather(parts, mine)
assert bytes(parts[0].cpu().numpy().tobytes()) == bytes(range(128))
ok = torch.tensor([1], dtype=torch.int32, device="cuda")
dist.all_reduce(ok, op=dist.ReduceOp.MIN)
assert int(ok.item()) == 1
t = torch.tensor([2.5], dtype=torch.float64, device="cuda")
dist.all_reduce(t)
dist.all_reduce(t, op=dist.ReduceOp.MAX)
assert float(t.item()) == 2.5
dist.barrier()
torch.cuda.synchronize()
dist.destroy_process_group()
print("nccl sanity ok")
