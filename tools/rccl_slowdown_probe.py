"""Why is a captured forward slower once RCCL is initialised (one rank, no collective in the step)?
python tools/rccl_slowdown_probe.py [none|init|init_destroy|gloo]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch, torch.distributed as dist
mode = sys.argv[1] if len(sys.argv) > 1 else "none"
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29534")
torch.cuda.set_device(0)
if mode in ("init", "init_destroy"):
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    t = torch.ones(1, device="cuda"); dist.all_reduce(t); torch.cuda.synchronize()
    if mode == "init_destroy":
        dist.destroy_process_group()
elif mode == "gloo":
    dist.init_process_group("gloo", rank=0, world_size=1)
import bench
spec = dict(bench.WORKLOADS[1])
r = bench.run_config(1, spec, 20, 5, graph=True, kernel_pass=False, announce=False, pipeline="prefetch", extra_blocks=2)
print(mode, os.environ.get("TORCH_NCCL_ENABLE_MONITORING"), [round(v, 4) for v in r["block_ms"]], flush=True)
