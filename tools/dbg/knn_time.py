import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from pointasnl_amd import _hip
_hip.lib()
g = torch.Generator(device="cuda").manual_seed(1)
for (b, n, m, k) in [(64, 1024, 512, 32), (64, 512, 128, 64), (8, 1280, 320, 32), (8, 320, 320, 32), (8, 80, 80, 32), (16, 1024, 256, 32), (16, 1024, 1024, 16)]:
    sup = torch.rand((b, n, 3), device="cuda", generator=g); qry = sup[:, :m].contiguous()
    idx = torch.empty((b, m, k), dtype=torch.int32, device="cuda")
    run = lambda: _hip.launch("pasnl_knn_batch", "knn_batch", b, n, m, k, _hip.ptr(sup), _hip.ptr(qry), _hip.ptr(idx), 0, None)
    for _ in range(5): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): run()
    e1.record(); torch.cuda.synchronize()
    print((b, n, m, k), f"{e0.elapsed_time(e1) * 1e3 / 50:.1f} us")
