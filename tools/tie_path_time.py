"""Kernel times of the default kNN on the cls layer-1 shape with one chance tie (run under rocprofv3 --kernel-trace --stats)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench as B
import pointasnl_amd as P
shape = sys.argv[1] if len(sys.argv) > 1 else "cls"
if shape == "cls":
    sup = B.synth_clouds(1, 64, 1024); m = 512
elif shape == "scannet":
    sup = np.ascontiguousarray(B.synth_scannet(3, 16, 8192)[..., :3]); m = 1024
else:
    sup = np.ascontiguousarray(B.synth_kitti(4, 8, 10240)[..., :3]); m = 1280
s = torch.from_numpy(sup).cuda(); q = s[:, :m].contiguous()
for _ in range(20):
    stats = []
    P.nearest_neighbors.knn_batch(s, q, 32, dtype=torch.int32, stats=stats)
torch.cuda.synchronize()
print("listed", stats[0].cpu().numpy().tolist(), "left", stats[1].cpu().numpy().tolist())
