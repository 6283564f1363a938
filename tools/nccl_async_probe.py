import os, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR","127.0.0.1"); os.environ.setdefault("MASTER_PORT","29533")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
g = torch.ones(1<<20, device="cuda")
w1 = dist.all_reduce(g[1000:], async_op=True); w2 = dist.all_reduce(g[:1000], async_op=True)
w1.wait(); w2.wait(); torch.cuda.synchronize(); print("nccl async ok", float(g.sum()))
dist.destroy_process_group()
