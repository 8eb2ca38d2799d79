"""Picks per round of the large-cloud sampler (tuning build: PASNL_FPS_K = 0 (one pick per round, fps_pruned_kernel), 2, 3, 4):
HIP-event medians at the two model shapes, on ball / lidar-like / indoor-block clouds."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench as B
import pointasnl_amd as P
from pointasnl_amd import _hip
_hip.LIB_PATH = os.path.join(os.path.dirname(_hip.LIB_PATH), "libpasnl_hip_tuning.so")
cases = [("ball 16x8192->1024", B.synth_clouds(1, 16, 8192), 1024), ("scannet 16x8192->1024", B.synth_scannet(2, 16, 8192)[..., :3].copy(), 1024),
         ("kitti 8x10240->1280", B.synth_kitti(3, 8, 10240), 1280), ("ball 8x10240->1280", B.synth_clouds(4, 8, 10240), 1280),
         ("ball 16x4096->512", B.synth_clouds(5, 16, 4096), 512)]
for name, x, m in cases:
    xd = torch.from_numpy(x).cuda()
    ref = None
    row = []
    for k in ("0", "2", "3", "4"):
        os.environ["PASNL_FPS_K"] = k
        for _ in range(2):
            out = P.tf_sampling.farthest_point_sample(m, xd)
        torch.cuda.synchronize()
        ts = []
        for _ in range(7):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); out = P.tf_sampling.farthest_point_sample(m, xd); e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3)
        if ref is None:
            ref = out.clone()
        row.append(f"K={k}: {np.median(ts):7.1f} us{'' if torch.equal(out, ref) else ' MISMATCH'}")
    print(f"{name:24s} " + "  ".join(row), flush=True)
