"""Tuning build only: FPS kernel shapes (PASNL_FPS_CFG = "<waves>,<ppl>" old kernel, "s<waves>,<ppl>" small-cloud kernel),
bit-exactness against the oracle + median time.   make -C pointasnl_amd/csrc tuning; python tools/fps_sweep.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch
import bench as B
import oracle
from conftest import clouds
from pointasnl_amd import _hip
_hip.LIB_PATH = os.path.join(os.path.dirname(_hip.LIB_PATH), "libpasnl_hip_tuning.so")
import pointasnl_amd as P

oracle.build()
def run(cfg, x, m, iters=20):
    if cfg: os.environ["PASNL_FPS_CFG"] = cfg
    else: os.environ.pop("PASNL_FPS_CFG", None)
    for _ in range(3): out = P.tf_sampling.farthest_point_sample(m, x)
    torch.cuda.synchronize()
    ts = []
    for _ in range(iters):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); out = P.tf_sampling.farthest_point_sample(m, x); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    return out.cpu().numpy(), float(np.median(ts))

for (b, n, m, cfgs) in [(64, 1024, 512, ["", "4,4", "8,2", "16,2", "2,8", "8,4"]), (64, 512, 128, ["", "1,8", "s1,8", "s2,4", "s4,2", "s4,4"]),
                        (256, 1024, 512, ["", "s4,4", "s2,8"]), (16, 2048, 256, ["", "s4,8", "s2,16"]), (16, 4096, 512, ["", "s4,16"])]:
    for kind in ("ball", "lattice"):
        x_np = B.synth_clouds(3, b, n) if kind == "ball" else clouds(7, b, n, "lattice")
        want = oracle.ops.farthest_point_sample(m, x_np[:8])
        x = torch.from_numpy(x_np).cuda()
        for cfg in cfgs:
            got, us = run(cfg, x, m)
            print(f"B={b} n={n} m={m} {kind:8s} cfg={cfg or 'default':8s} {us:8.1f} us   exact={bool((got[:8] == want).all())}", flush=True)
