"""Tuning build: how well the large-cloud sampler (fps_pruned_kernel) prunes -- active wave-rounds / wave-rounds -- and its time,
on the bench's clouds (ball 16x8192 -> 1024, ScanNet-like block, lidar-like 8x10240 -> 1280)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench as B
from pointasnl_amd import _hip
_hip.LIB_PATH = os.path.join(os.path.dirname(_hip.LIB_PATH), "libpasnl_hip_tuning.so")
import pointasnl_amd as P
buf = (ctypes.c_ulonglong * 8)()
cases = [("ball 16x8192->1024", B.synth_clouds(1, 16, 8192), 1024), ("kitti 8x10240->1280", B.synth_kitti(3, 8, 10240), 1280)]
if hasattr(B, "synth_scannet"):
    cases.append(("scannet 16x8192->1024", B.synth_scannet(2, 16, 8192)[..., :3].copy(), 1024))
for name, x, m in cases:
    xt = torch.from_numpy(x).cuda()
    P.tf_sampling.farthest_point_sample(m, xt); torch.cuda.synchronize()
    _hip.lib().pasnl_fps_dbg_read(buf)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); P.tf_sampling.farthest_point_sample(m, xt); e1.record(); torch.cuda.synchronize()
    _hip.lib().pasnl_fps_dbg_read(buf)
    print(f"{name}: {e0.elapsed_time(e1) * 1e3:.0f} us (tuning build, counters on), active wave-rounds {buf[0]} / {buf[1]} = {buf[0] / max(1, buf[1]):.3f}"
          + (f", rounds with k active waves on the busiest SIMD: {[buf[2], buf[3], buf[4], buf[5], buf[6]]}" if buf[2] + buf[3] + buf[4] else ""), flush=True)
