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

# workgroup timeline of one launch at B = 4096: how many workgroups a CU really holds at a time
import ctypes as C, collections
print("runtime occupancy (workgroups per CU, n = 1024):", lib.pasnl_ball_occupancy(1024))
x = torch.from_numpy(B.synth_clouds(1, 256, 1024)).cuda().repeat(16, 1, 1).contiguous()
q = x[:, :512].contiguous()
P.tf_grouping.query_ball_point(0.2, 32, x, q)
torch.cuda.synchronize()
nwg = 4096
buf = (C.c_ulonglong * (4 * nwg))()
lib.pasnl_ball_trace_read(buf, nwg)
per_cu = collections.defaultdict(list)
t0 = min(buf[4 * i] for i in range(nwg))
for i in range(nwg):
    st, en, hw, xcc = buf[4 * i], buf[4 * i + 1], buf[4 * i + 2], buf[4 * i + 3]
    cu = (xcc & 0xF, (hw >> 13) & 0x7, (hw >> 12) & 1, (hw >> 8) & 0xF)  # xcc, se, sh, cu
    per_cu[cu].append((st - t0, en - t0))
conc = []
for cu, iv in per_cu.items():
    ev = sorted([(a, 1) for a, _ in iv] + [(b, -1) for _, b in iv])
    cur = mx = 0
    for _, d in ev:
        cur += d
        mx = max(mx, cur)
    conc.append(mx)
dur = [b - a for iv in per_cu.values() for a, b in iv]
import numpy as np
print(f"CUs seen {len(per_cu)}, workgroups per CU {nwg / len(per_cu):.1f}, max concurrent workgroups per CU: min {min(conc)} median {int(np.median(conc))} max {max(conc)}")
print(f"workgroup duration (100 MHz ticks): median {np.median(dur):.0f}, p10 {np.percentile(dur, 10):.0f}, p90 {np.percentile(dur, 90):.0f}; span {max(b for iv in per_cu.values() for _, b in iv)} ticks")

# slot utilisation and dispatch gaps per CU
util, gaps = [], []
for cu, iv in per_cu.items():
    iv = sorted(iv)
    first, lastend = iv[0][0], max(b for _, b in iv)
    util.append(sum(b - a for a, b in iv) / (3.0 * (lastend - first)))
    ends = sorted(b for _, b in iv)
    starts = sorted(a for a, _ in iv)
    # the k-th start (k >= 3) follows the (k-3)-th end: the slot's idle time between two workgroups
    for k in range(3, len(starts)):
        gaps.append(starts[k] - ends[k - 3])
starts_all = sorted(a for iv in per_cu.values() for a, _ in iv)
print(f"slot utilisation per CU: median {np.median(util):.2f} min {min(util):.2f}; gap between a workgroup's end and the next start in its slot (ticks): median {np.median(gaps):.0f} p90 {np.percentile(gaps, 90):.0f}")
print("start time of the k-th workgroup (ticks): " + ", ".join(f"{k}:{starts_all[k]}" for k in (0, 255, 511, 767, 1023, 2047, 4095)))
print("per-CU first start -> last end (ticks): median", int(np.median([max(b for _, b in iv) - min(a for a, _ in iv) for iv in per_cu.values()])))
