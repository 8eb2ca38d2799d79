"""pasnl_knn_batch_ref against pasnl_knn_batch_ws at the models' search shapes: time, flagged queries per call.
python tools/knn_ref_probe.py   (under rocprofv3 --kernel-trace --stats for the per-kernel split)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench as B
import pointasnl_amd as P

def t(fn, it=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e3

shapes = [("cls L1", B.synth_clouds(1, 64, 1024), 512, 32), ("cls L2", B.synth_clouds(2, 64, 512), 128, 64),
          ("scannet L1", B.synth_scannet(3, 16, 8192)[..., :3].copy(), 1024, 32), ("scannet self", B.synth_scannet(3, 16, 8192)[..., :3].copy(), 8192, 32),
          ("kitti self", B.synth_kitti(4, 8, 10240), 10240, 32), ("kitti L1", B.synth_kitti(4, 8, 10240), 1280, 32)]
for name, sup, m, k in shapes:
    s = torch.from_numpy(np.ascontiguousarray(sup)).cuda(); q = s[:, :m].contiguous()
    stats = []
    P.nearest_neighbors.knn_batch(s, q, k, dtype=torch.int32, stats=stats)
    nf = stats[0].cpu().numpy()
    a = t(lambda: P.nearest_neighbors.knn_batch(s, q, k, dtype=torch.int32, tie_order="index"))
    r = t(lambda: P.nearest_neighbors.knn_batch(s, q, k, dtype=torch.int32))
    print(f"{name:14s} canonical {a:8.1f} us   reference {r:8.1f} us   flagged queries {int(nf.sum())} in {int((nf > 0).sum())} clouds of {len(nf)}", flush=True)
