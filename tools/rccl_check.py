import os, torch, torch.distributed as dist
os.environ.setdefault('MASTER_ADDR','127.0.0.1'); os.environ.setdefault('MASTER_PORT','29511')
torch.cuda.set_device(0)
dist.init_process_group('nccl', rank=0, world_size=1)
x=torch.ones(1<<20, device='cuda')
h=dist.all_reduce(x[:1000], op=dist.ReduceOp.SUM, async_op=True); h.wait()
dist.barrier(); torch.cuda.synchronize()
print('rccl ok', float(x.sum()))
dist.destroy_process_group()
