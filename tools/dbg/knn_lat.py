import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from pointasnl_amd import _hip
if len(sys.argv) > 1: _hip.LIB_PATH = os.path.abspath(sys.argv[1])
_hip.lib()
g = torch.Generator(device="cuda").manual_seed(1)
for (b, n, m, k) in [(64, 1024, 512, 32), (16, 8192, 1024, 32)]:
    sup = torch.rand((b, n, 3), device="cuda", generator=g); qry = sup[:, :m].contiguous()
    idx = torch.empty((b, m, k), dtype=torch.int32, device="cuda")
    run = lambda: _hip.launch("pasnl_knn_batch", "knn_batch", b, n, m, k, _hip.ptr(sup), _hip.ptr(qry), _hip.ptr(idx), 0, None)
    for _ in range(5): run()
    torch.cuda.synchronize()
    ts = []
    for _ in range(20):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); run(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) * 1e3)
    print(os.path.basename(_hip.LIB_PATH), (b, n, m, k), f"isolated launch: median {sorted(ts)[10]:.1f} us")
