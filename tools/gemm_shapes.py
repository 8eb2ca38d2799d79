"""Times every dense layer (tf_util._dense = one hipBLASLt GEMM + epilogue) of the cls forward in isolation:
(scope, M, K, N, us, TFLOP/s).   python tools/gemm_shapes.py [--AS]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench as B
from pointasnl_amd.utils import tf_util
from pointasnl_amd.models import pointasnl_cls

x = torch.from_numpy(B.synth_clouds(1235, 64, 1024)).cuda()
tf_util.set_store(tf_util.VariableStore(seed=1234))
rows = []
orig = tf_util._dense
def timed(inputs, cout, scope, bn, act, weight_decay=None):
    out = orig(inputs, cout, scope, bn, act, weight_decay)
    if timed.on:
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ts = []
        for _ in range(10):
            e0.record(); orig(inputs, cout, scope, bn, act, weight_decay); e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3)
        m, k = inputs.numel() // inputs.shape[-1], inputs.shape[-1]
        us = sorted(ts)[len(ts) // 2]
        rows.append((tf_util.store().path(scope), m, k, cout, us, 2.0 * m * k * cout / us / 1e6))
    return out
timed.on = False
tf_util._dense = timed
with torch.no_grad():
    pointasnl_cls.get_model(x, is_training=False, adaptive_sample="--AS" in sys.argv)
    timed.on = True
    pointasnl_cls.get_model(x, is_training=False, adaptive_sample="--AS" in sys.argv)
tot = 0
for r in sorted(rows, key=lambda r: -r[4]):
    print(f"{r[0]:44s} M={r[1]:7d} K={r[2]:5d} N={r[3]:5d} {r[4]:8.1f} us {r[5]:7.1f} TF/s")
    tot += r[4]
print("total", round(tot, 1), "us")
