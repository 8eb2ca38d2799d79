"""Tuning build only: phase cycles per round of fps_small_kernel (workgroup 0, wave 0; s_memtime marks)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench as B
from pointasnl_amd import _hip
_hip.LIB_PATH = os.path.join(os.path.dirname(_hip.LIB_PATH), "libpasnl_hip_tuning.so")
import pointasnl_amd as P
x = torch.from_numpy(B.synth_clouds(3, 64, 1024)).cuda()
dbg = torch.zeros(8, dtype=torch.int64, device="cuda")
os.environ["PASNL_FPS_PROBE"] = hex(dbg.data_ptr())
for cfg in sys.argv[1:] or ["s4,4", "s2,8", "s1,16"]:
    os.environ["PASNL_FPS_CFG"] = cfg
    for _ in range(3):
        P.tf_sampling.farthest_point_sample(512, x)
    torch.cuda.synchronize()
    t = dbg.cpu().tolist()
    names = ["(loop overhead)", "dist+tournament", "wave max+slot write", "barrier", "slot reads+reduce"]
    print(cfg, "cycles/round (100 MHz s_memtime ticks x clock ratio!):", ", ".join(f"{n} {t[i] / 511:.1f}" for i, n in enumerate(names)), " total", sum(t[:5]) / 511)
