"""Where the cycles of the MFMA non-local attention kernel go (probe build of the library, see tools/sa_cell_probe.py).
    python tools/nl_probe.py <tag> [launches]"""
import ctypes, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pointasnl_amd import _hip
tag = sys.argv[1] if len(sys.argv) > 1 else ""
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 100
_hip.LIB_PATH = os.path.join(os.path.dirname(_hip.LIB_PATH), f"libpasnl_hip_probe{tag}.so")
lib = _hip.lib()
read = lib.pasnl_nl_probe_read
read.argtypes = [ctypes.POINTER(ctypes.c_ulonglong)]
names = ["staging", "S", "softmax", "PV", "blocks", "loop_total", "waves"]
g = torch.Generator(device="cuda").manual_seed(1)
for (b, p, n, cb, name) in [(64, 512, 1024, 32, "cls-L1"), (64, 128, 512, 64, "cls-L2"), (16, 1024, 8192, 32, "scannet-L1")]:
    q = torch.randn((b, p, cb), device="cuda", generator=g)
    kv = torch.randn((b, n, 2 * cb), device="cuda", generator=g)
    out = torch.empty_like(q)
    run = lambda: _hip.launch("pasnl_nl_attention", "nl_attention", b, p, n, cb, _hip.ptr(q), _hip.ptr(kv), _hip.ptr(out), 0)
    for _ in range(reps):
        run()
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * 16)()
    read(buf)
    _hip.PROFILE = []
    for _ in range(reps):
        run()
    torch.cuda.synchronize()
    us = sorted(e0.elapsed_time(e1) * 1e3 for sym, ints, e0, e1 in _hip.PROFILE)
    _hip.PROFILE = None
    read(buf)
    v = dict(zip(names, [x / reps for x in list(buf)[:7]]))
    blocks, waves = max(1.0, v["blocks"]), max(1.0, v["waves"])
    row = {"lib": tag, "shape": name, "kernel_us_median": round(us[len(us) // 2], 1), "waves": round(waves),
           "blocks_per_wave": round(blocks / waves, 2), "loop_cycles_per_wave": round(v["loop_total"] / waves)}
    for k in ("staging", "S", "softmax", "PV"):
        row[k + "_per_block"] = round(v[k] / blocks)
    print(json.dumps(row), flush=True)
