"""fps_kernel at cls layer-1 (1024 -> 512) over batch sizes for the workgroup shapes (waves, points per lane) that cover 1024
points: does a single barrier-free wave per cloud, (1,16), beat (4,4) once the CUs are full?  (tuning build; drives itself)"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
for b in (64, 256, 1024, 2048):
    row = []
    for w, p in [(4, 4), (2, 8), (1, 16), (8, 2)]:
        env = dict(os.environ, PASNL_FPS_CFG=f"{w},{p}")
        out = subprocess.run([sys.executable, os.path.join(HERE, "fps_cfg_sweep.py"), "--one", str(b), "1024", "512"], env=env,
                             capture_output=True, text=True)
        row.append(f"({w},{p}) {out.stdout.strip() or 'ERR'} us")
    print(f"B={b} 1024->512:  " + "  ".join(row), flush=True)
