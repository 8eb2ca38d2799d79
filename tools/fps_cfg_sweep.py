"""(waves, points-per-lane) sweep of fps_kernel at given shapes, one process per configuration (tuning build:
PASNL_FPS_CFG is read at launch).   python tools/fps_cfg_sweep.py   (drives itself)"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
if len(sys.argv) > 1 and sys.argv[1] == "--one":
    b, n, m = map(int, sys.argv[2:5])
    sys.path.insert(0, os.path.dirname(HERE))
    import torch
    from pointasnl_amd import _hip
    _hip.LIB_PATH = _hip.LIB_PATH.replace("libpasnl_hip.so", "libpasnl_hip_tuning.so")
    import pointasnl_amd
    x = torch.rand(b, n, 3, device="cuda")
    f = lambda: pointasnl_amd.tf_sampling.farthest_point_sample(m, x)
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        f()
    e1.record(); torch.cuda.synchronize()
    print(f"{e0.elapsed_time(e1) * 100:.1f}")
    sys.exit(0)

for b, n, m in [(8, 1280, 320), (16, 2048, 512), (16, 1536, 384), (64, 1024, 512)]:
    row = []
    for w, p in [(2, 16), (4, 8), (8, 4), (4, 16), (8, 8), (16, 2), (16, 4), (8, 2), (4, 4)]:
        if w * 64 * p < n:
            continue
        env = dict(os.environ, PASNL_FPS_CFG=f"{w},{p}", PASNL_FPS_NOPRUNE="1")
        out = subprocess.run([sys.executable, __file__, "--one", str(b), str(n), str(m)], env=env, capture_output=True, text=True)
        row.append(f"({w},{p}) {out.stdout.strip() or 'ERR'}")
    print(f"B={b} n={n} m={m}:  " + "  ".join(row), flush=True)
