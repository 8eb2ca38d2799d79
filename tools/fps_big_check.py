"""FPS on large clouds: exactness vs the C oracle + time."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import bench as B, oracle
from conftest import clouds
import pointasnl_amd as P
oracle.build()
def t(x, m, iters=10):
    for _ in range(2): out = P.tf_sampling.farthest_point_sample(m, x)
    torch.cuda.synchronize(); ts=[]
    for _ in range(iters):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); out = P.tf_sampling.farthest_point_sample(m, x); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1)*1e3)
    return out.cpu().numpy(), float(np.median(ts))
cases = [("ball 16x8192->1024", B.synth_clouds(1, 16, 8192), 1024), ("scannet 16x8192->1024", B.synth_scannet(2, 16, 8192)[..., :3].copy(), 1024),
         ("kitti 8x10240->1280", B.synth_kitti(3, 8, 10240), 1280), ("lattice 4x8192->1024", clouds(4, 4, 8192, "lattice"), 1024),
         ("ball 4x4096->512", B.synth_clouds(5, 4, 4096), 512), ("ball 3x5000->700", B.synth_clouds(6, 3, 5000), 700),
         ("dup 2x6000->800", np.repeat(B.synth_clouds(7, 2, 3000), 2, axis=1), 800), ("ball 2x2500->2500", B.synth_clouds(8, 2, 2500), 2500)]
for name, x_np, m in cases:
    got, us = t(torch.from_numpy(np.ascontiguousarray(x_np)).cuda(), m)
    want = oracle.ops.farthest_point_sample(m, x_np[:2])
    print(f"{name:26s} {us:9.1f} us  exact={bool((got[:2] == want).all())}", flush=True)
