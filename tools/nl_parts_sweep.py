"""nl_attention: keys over workgroups (pasnl_nl_attention_ws) -- sweep of (kparts, waves per workgroup) with the tuning build.
python tools/nl_parts_sweep.py"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pointasnl_amd import _hip
_hip.LIB_PATH = os.path.join(ROOT, "pointasnl_amd", "csrc", "libpasnl_hip_tuning.so")
import numpy as np, torch
from pointasnl_amd.utils import pointasnl_util as U

def t(fn, it=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e3

for b, p, n in [(8, 1280, 10240), (8, 320, 1280), (16, 256, 1024), (4, 1024, 8192), (16, 1024, 8192)]:
    g = torch.Generator(device="cuda").manual_seed(1)
    q = torch.randn((b, p, 32), device="cuda", generator=g); kv = torch.randn((b, n, 64), device="cuda", generator=g)
    U.NL_KEY_PARTS = False
    os.environ.pop("PASNL_NL_PARTS", None)
    base = U.nl_attention(q, kv); t0 = t(lambda: U.nl_attention(q, kv))
    U.NL_KEY_PARTS = True
    auto = U.nl_attention(q, kv); ta = t(lambda: U.nl_attention(q, kv))
    ref = torch.softmax((q.double() @ kv[..., :32].double().transpose(1, 2)) / np.sqrt(32.0), -1) @ kv[..., 32:].double()
    print(f"[{b},{p},{n}] plain {t0:.1f} us  auto {ta:.1f} us  err plain {float((base - ref).abs().max()):.2e} auto {float((auto - ref).abs().max()):.2e}", flush=True)
    if int(_hip.lib().pasnl_nl_attention_workspace_bytes(b, p, n, 32)) == 0:
        continue
    row = []
    for split in (8, 4, 2):
        for k in (2, 3, 4, 5, 6, 8, 10, 16):
            if k * split * 2 > n // 32: continue
            os.environ["PASNL_NL_PARTS"] = f"{k},{split}"
            try:
                o = U.nl_attention(q, kv)
            except Exception as e:
                continue
            row.append((t(lambda: U.nl_attention(q, kv)), k, split, float((o - ref).abs().max())))
    row.sort()
    print("   best:", " | ".join(f"k={k} w={s}: {tt:.1f}" for tt, k, s, _ in row[:8]), flush=True)
