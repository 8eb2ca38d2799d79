"""The grid-pruned kNN (knn_grid.hip, global workspace) on the SMALL clouds it is not used for (tuning build:
PASNL_KNN_GRID_MIN_N), against the brute-force kernels: cls layer1 / layer2, ScanNet and KITTI deep levels."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench as B
from pointasnl_amd import _hip
_hip.LIB_PATH = os.path.join(os.path.dirname(_hip.LIB_PATH), "libpasnl_hip_tuning.so")
import pointasnl_amd as P
for (b, n, m, k) in [(64, 1024, 1024, 32), (64, 1024, 512, 32), (64, 512, 128, 64), (16, 1024, 256, 32), (16, 1024, 1024, 16),
                     (8, 1280, 320, 32), (8, 1280, 1280, 32), (64, 2048, 2048, 32)]:
    x = torch.from_numpy(B.synth_clouds(n + m, b, n)).cuda()
    q = x[:, :m].contiguous()
    res = {}
    for mode, minn in (("brute", "100000"), ("grid", "256")):
        os.environ["PASNL_KNN_GRID_MIN_N"] = minn
        for _ in range(3):
            out = P.nearest_neighbors.knn_batch(x, q, k, dtype=torch.int32)
        torch.cuda.synchronize()
        ts = []
        for _ in range(9):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); out = P.nearest_neighbors.knn_batch(x, q, k, dtype=torch.int32); e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3)
        res[mode] = (float(np.median(ts)), out.clone())
    same = torch.equal(res["brute"][1], res["grid"][1])
    print(f"b={b} n={n} m={m} k={k}: brute {res['brute'][0]:7.1f} us, grid {res['grid'][0]:7.1f} us, identical {same}", flush=True)
