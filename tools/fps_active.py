"""Tuning build: where a round of the large-cloud sampler (fps_multi_kernel) goes -- touched waves per round, picks per round,
cycles of wave 1 before the first barrier, of the merging wave between the barriers, of a whole round (workgroup 0)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench as B
from pointasnl_amd import _hip
_hip.LIB_PATH = os.path.join(os.path.dirname(_hip.LIB_PATH), "libpasnl_hip_tuning.so")
import pointasnl_amd as P
buf = (ctypes.c_ulonglong * 8)()
for k in ("2", "3", "4"):
    os.environ["PASNL_FPS_K"] = k
    for name, x, m in [("ball 16x8192->1024", B.synth_clouds(1, 16, 8192), 1024), ("kitti 8x10240->1280", B.synth_kitti(3, 8, 10240), 1280)]:
        xt = torch.from_numpy(x).cuda()
        P.tf_sampling.farthest_point_sample(m, xt); torch.cuda.synchronize()
        _hip.lib().pasnl_fps_dbg_read(buf)
        P.tf_sampling.farthest_point_sample(m, xt); torch.cuda.synchronize()
        _hip.lib().pasnl_fps_dbg_read(buf)
        r = max(1, buf[3])
        print(f"K={k} {name}: rounds {buf[3]}, picks/round {buf[7] / r:.2f}, touched waves/round {buf[2] / r:.2f}, cycles/round: "
              f"wave1 before barrier {buf[4] / r:.0f}, merge (wave 0) {buf[5] / r:.0f}, whole round {buf[6] / r:.0f}", flush=True)
