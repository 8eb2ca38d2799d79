"""Tuning build: grid kNN time vs points-per-cell target (PASNL_KNN_RHO = rho / K)."""
import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench as B
from pointasnl_amd import _hip
_hip.LIB_PATH = os.path.join(os.path.dirname(_hip.LIB_PATH), "libpasnl_hip_tuning.so")
import pointasnl_amd as P
def t(x, k):
    for _ in range(2): P.nearest_neighbors.knn_batch(x, x, k, dtype=torch.int32)
    torch.cuda.synchronize(); ts=[]
    for _ in range(6):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); P.nearest_neighbors.knn_batch(x, x, k, dtype=torch.int32); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1)*1e3)
    return float(np.median(ts))
data = {"ball8192": torch.from_numpy(B.synth_clouds(1, 16, 8192)).cuda(), "scannet8192": torch.from_numpy(B.synth_scannet(2, 16, 8192)[..., :3].copy()).cuda(),
        "kitti10240": torch.from_numpy(B.synth_kitti(3, 8, 10240)).cuda()}
os.environ["PASNL_KNN_REFINE"] = sys.argv[1]
for rho in ("0.55", "0.7", "0.85", "1.0"):
    os.environ["PASNL_KNN_RHO"] = rho
    print(rho, {f"{n} K={k}": round(t(x, k)) for n, x in data.items() for k in (16, 32)}, flush=True)
