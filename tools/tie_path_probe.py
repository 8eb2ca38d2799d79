"""Phases of knn_tie_path_kernel's workgroup 0 (tuning build): python tools/tie_path_probe.py [cls|scannet]"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pointasnl_amd import _hip
_hip.LIB_PATH = os.path.join(ROOT, "pointasnl_amd", "csrc", "libpasnl_hip_tuning.so")
import numpy as np, torch
import bench as B
import pointasnl_amd as P
shape = sys.argv[1] if len(sys.argv) > 1 else "cls"
if shape == "cls":
    sup = B.synth_clouds(1, 64, 1024); m = 512
else:
    sup = np.ascontiguousarray(B.synth_scannet(3, 16, 8192)[..., :3]); m = 1024
if len(sys.argv) > 2:
    sup[0, 7] = sup[0, 3]          # a duplicated point (never separated: the path runs down to a leaf)
s = torch.from_numpy(sup).cuda(); q = s[:, :m].contiguous()
for _ in range(3):
    stats = []
    P.nearest_neighbors.knn_batch(s, q, 32, dtype=torch.int32, stats=stats)
torch.cuda.synchronize()
print("listed:", int(stats[0].sum()), "left:", int(stats[1].sum()))
buf = (ctypes.c_ulonglong * 32)()
assert _hip.lib().pasnl_knn_small_probe_read(buf) == 0
t = np.array(list(buf), dtype=np.float64)
us = lambda a, b: (t[b] - t[a]) / 100.0  # units of 100 s_memtime ticks (~2.2 GHz here: 100 ticks = 0.045 us)
if t[1] > t[0]:
    print(f"count {us(0,1):.2f} us, load {us(1,2):.2f}, tied points {us(2,3):.2f}")
else:
    print(f"(small kernel) entry to the tied points listed {us(0,3):.2f}")
prev = 3
for i in range(24):
    if t[4 + i] > t[prev]:
        print(f"split {i}: {us(prev, 4 + i):.2f} us")
        prev = 4 + i
print(f"row {us(prev,30):.2f} us; total {us(0,30):.2f} us (units of 100 cycles)")
