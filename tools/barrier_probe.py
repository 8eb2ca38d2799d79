import os, time, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR","127.0.0.1"); os.environ.setdefault("MASTER_PORT","29533")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda",0))
t=torch.ones(1,device="cuda"); dist.all_reduce(t); torch.cuda.synchronize()
for i in range(5):
    torch.cuda.synchronize(); t0=time.perf_counter(); dist.barrier(); t1=time.perf_counter(); torch.cuda.synchronize(); t2=time.perf_counter()
    print(f"barrier {1e3*(t1-t0):.3f} ms  (+sync {1e3*(t2-t1):.3f})")
x=torch.ones(64,40,device="cuda"); o=torch.empty(64,40,device="cuda")
for i in range(3):
    torch.cuda.synchronize(); t0=time.perf_counter(); dist.all_gather_into_tensor(o,x); torch.cuda.synchronize(); print(f"all_gather+sync {1e3*(time.perf_counter()-t0):.3f} ms")
dist.destroy_process_group()
