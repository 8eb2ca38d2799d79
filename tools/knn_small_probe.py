"""Phases of knn_tree_small_kernel (tuning build): python tools/knn_small_probe.py [n]"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pointasnl_amd import _hip
_hip.LIB_PATH = os.path.join(ROOT, "pointasnl_amd", "csrc", "libpasnl_hip_tuning.so")
import numpy as np, torch
import bench as B
import pointasnl_amd as P
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
sup = B.synth_clouds(1, 64, n)
sup[5, 7] = sup[5, 3]          # ONE duplicated point: the queries around it are flagged
s = torch.from_numpy(sup).cuda(); q = s[:, :n // 2].contiguous()
for _ in range(3):
    stats = []
    P.nearest_neighbors.knn_batch(s, q, 32, dtype=torch.int32, stats=stats)
torch.cuda.synchronize()
print("flagged:", stats[0].cpu().numpy().sum(), "in clouds", np.nonzero(stats[0].cpu().numpy())[0])
buf = (ctypes.c_ulonglong * 32)()
assert _hip.lib().pasnl_knn_small_probe_read(buf) == 0
t = np.array(list(buf), dtype=np.float64)
ns = lambda a, b: (t[b] - t[a]) * 10.0  # s_memtime ticks at 100 MHz
print(f"load+box {ns(0,1):.0f} ns")
prev = 1
for l in range(1, 20):
    if t[1 + l] > t[prev]:
        print(f"level {l - 1}: {ns(prev, 1 + l):.0f} ns")
        prev = 1 + l
print(f"build total {ns(0,30):.0f} ns, searches {ns(30,31):.0f} ns")
