"""Where the cycles of pasnl_sa_cell go.  Runs a probe build of the library (make -C pointasnl_amd/csrc probe PROBE=<level>
ABLATE=<mask> TAG=<tag>) on the cls shapes under sustained load and prints, per tile and wave, the s_memtime cycles
between the kernel's phase marks, plus the clock implied by (cycles per wave) / (launch time).
    python tools/sa_cell_probe.py <tag> [launches]"""
import ctypes, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pointasnl_amd import _hip
tag = sys.argv[1] if len(sys.argv) > 1 else ""
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 100
_hip.LIB_PATH = os.path.join(os.path.dirname(_hip.LIB_PATH), f"libpasnl_hip_probe{tag}.so")
from pointasnl_amd.utils import pointasnl_util as U, tf_util

lib = _hip.lib()
read = lib.pasnl_sa_cell_probe_read
read.argtypes = [ctypes.POINTER(ctypes.c_ulonglong)]
names = ["prologue", "start_wait", "conv0", "chunk_wait", "conv1", "epilogue", "tiles", "wave_total", "waves", "staging"]
out = []
tf_util.set_store(tf_util.VariableStore(seed=5))
g = torch.Generator(device="cuda").manual_seed(1)
for (b, n, c, m, k, c1, name) in [(64, 1024, 3, 512, 32, 64, "cls-L1"), (64, 512, 128, 128, 64, 128, "cls-L2")]:
    xyz = torch.rand((b, n, 3), device="cuda", generator=g)
    feat = torch.randn((b, n, c), device="cuda", generator=g)
    idx = torch.randint(0, n, (b, m, k), device="cuda", dtype=torch.int32, generator=g)
    nx = xyz[:, :m].contiguous()
    with tf_util.variable_scope(name):
        run = lambda: U.sa_cell(xyz, feat, idx, nx, [c1, c1, 2 * c1], False, None, None, True)
        for _ in range(reps):
            run()
        torch.cuda.synchronize()
        buf = (ctypes.c_ulonglong * 16)()
        read(buf)  # reset
        _hip.PROFILE = []  # HIP events around every C-ABI launch (as bench_ops.py)
        for _ in range(reps):
            run()
        torch.cuda.synchronize()
        us_all = sorted(e0.elapsed_time(e1) * 1e3 for sym, ints, e0, e1 in _hip.PROFILE if sym == "pasnl_sa_cell")
        _hip.PROFILE = None
        read(buf)
    v = dict(zip(names, [x / reps for x in list(buf)[:10]]))
    tiles, waves = max(1, v["tiles"]), max(1, v["waves"])
    us = us_all[len(us_all) // 2]
    row = {"lib": tag, "shape": name, "kernel_us_median": round(us, 1), "waves": round(waves),
           "tiles_per_wave": round(tiles / waves, 2), "wave_total_cycles": round(v["wave_total"] / waves), "staging_cycles": round(v["staging"] / waves),
           "implied_MHz": round(v["wave_total"] / waves / us)}
    for kname in ("start_wait", "conv0", "chunk_wait", "conv1"):
        row[kname + "_per_tile"] = round(v[kname] / tiles)
    groups = b * m
    row["prologue_per_group"] = round(v["prologue"] / groups)
    row["epilogue_per_group"] = round(v["epilogue"] / groups)
    print(json.dumps(row), flush=True)
    out.append(row)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open(f"gpurun_out/sa_cell_probe{tag}.json", "w"), indent=1)
