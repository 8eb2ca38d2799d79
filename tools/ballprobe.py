"""Diagnostics: phase cycles of one wave of the grid ball query (tuning build only: make -C pointasnl_amd/csrc tuning ->
libpasnl_hip_tuning.so, loaded here instead of the product library; pasnl_ball_probe_read is not in the product library)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench as B
import pointasnl_amd as P
from pointasnl_amd import _hip
_hip.LIB_PATH = os.path.join(os.path.dirname(_hip.LIB_PATH), "libpasnl_hip_tuning.so")
lib = _hip.lib()
names = ["build", "-", "setup+table", "walk", "network+rows out", "tier2/round end", "tier2 rounds", "steps"]
for b in (64, 1024, 4096):
    x = torch.from_numpy(B.synth_clouds(1, min(b, 256), 1024)).cuda()
    if b > 256:
        x = x.repeat(b // 256, 1, 1).contiguous()
    q = x[:, :512].contiguous()
    P.tf_grouping.query_ball_point(0.2, 32, x, q)
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * 8)()
    lib.pasnl_ball_probe_read(buf)
    reps = 5
    for _ in range(reps):
        P.tf_grouping.query_ball_point(0.2, 32, x, q)
    torch.cuda.synchronize()
    lib.pasnl_ball_probe_read(buf)
    t = [v / reps for v in buf]
    print(f"B={b}: " + ", ".join(f"{n} {t[i]:.0f}" for i, n in enumerate(names)), flush=True)
