"""Diagnostics: phase clocks of workgroup 0 of the grid ball query (PASNL_BALL_PROBE; tuning build only:
make -C pointasnl_amd/csrc tuning -> libpasnl_hip_tuning.so, loaded here instead of the product library)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench as B
import pointasnl_amd as P
from pointasnl_amd import _hip
_hip.LIB_PATH = os.path.join(os.path.dirname(_hip.LIB_PATH), "libpasnl_hip_tuning.so")
for b in (64, 4096):
    x = torch.from_numpy(B.synth_clouds(1, b, 1024)).cuda()
    q = x[:, :512].contiguous()
    dbg = torch.zeros(8, dtype=torch.int64, device="cuda")
    os.environ["PASNL_BALL_PROBE"] = hex(dbg.data_ptr())
    for _ in range(3):
        P.tf_grouping.query_ball_point(0.2, 32, x, q)
    torch.cuda.synchronize()
    t = dbg.cpu().tolist()
    names = ["A load+bbox", "B/C grid build", "D search r1", "E emit r1", "F copy r1", "D search r2", "E emit r2"]
    print(f"B={b}: " + ", ".join(f"{n} {t[i+1]-t[i]}" for i, n in enumerate(names)), " | F copy r2", t[0] - t[7])
