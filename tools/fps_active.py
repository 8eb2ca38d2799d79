"""Tuning build: fraction of (wave, round) pairs of fps_pruned_kernel that are ACTIVE (not skipped by the box test)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench as B
from pointasnl_amd import _hip
_hip.LIB_PATH = os.path.join(os.path.dirname(_hip.LIB_PATH), "libpasnl_hip_tuning.so")
import pointasnl_amd as P
buf = (ctypes.c_ulonglong * 4)()
for name, x, m in [("ball 16x8192->1024", B.synth_clouds(1, 16, 8192), 1024), ("scannet", B.synth_scannet(2, 16, 8192)[..., :3].copy(), 1024),
                   ("kitti 8x10240->1280", B.synth_kitti(3, 8, 10240), 1280)]:
    xt = torch.from_numpy(x).cuda()
    P.tf_sampling.farthest_point_sample(m, xt); torch.cuda.synchronize()
    _hip.lib().pasnl_fps_dbg_read(buf)
    P.tf_sampling.farthest_point_sample(m, xt); torch.cuda.synchronize()
    _hip.lib().pasnl_fps_dbg_read(buf)
    print(name, "active fraction", buf[0] / buf[1], "active waves per round", 16 * buf[0] / buf[1])
