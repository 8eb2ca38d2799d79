"""Tuning build: the two-tile non-local attention kernel (cb = 32) against the one-tile kernel (PASNL_NL_PAIR=0), per key split."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1:
    sys.path.insert(0, ROOT)
    import numpy as np, torch
    from pointasnl_amd import _hip
    _hip.LIB_PATH = os.path.join(os.path.dirname(_hip.LIB_PATH), os.environ.get("PASNL_AB_LIB", "libpasnl_hip_tuning.so"))
    from pointasnl_amd.utils import pointasnl_util as U
    out = []
    for (b, p, n, cb, name) in [(64, 512, 1024, 32, "cls-L1"), (16, 1024, 8192, 32, "scannet-L1"), (8, 1280, 10240, 32, "kitti-L1_1"), (32, 1024, 4096, 32, "mid")]:
        q = torch.randn((b, p, cb), device="cuda"); kv = torch.randn((b, n, 2 * cb), device="cuda")
        for _ in range(3): U.nl_attention(q, kv, variant=2)
        torch.cuda.synchronize()
        ts = []
        for _ in range(20):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); U.nl_attention(q, kv, variant=2); e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3)
        out.append(f"{name} {np.median(ts):7.1f}")
    print(f"pair={os.environ.get('PASNL_NL_PAIR', '1')} split={os.environ.get('PASNL_NL_SPLIT', 'auto'):>4s}", " | ".join(out), flush=True)
else:
    for pair in ("0", "1"):
        for sp in ("", "2", "4", "8"):
            env = dict(os.environ, PASNL_NL_PAIR=pair)
            if sp: env["PASNL_NL_SPLIT"] = sp
            else: env.pop("PASNL_NL_SPLIT", None)
            subprocess.run([sys.executable, __file__, "run"], env=env)
