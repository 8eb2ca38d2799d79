"""What torch's TunableOp finds for the long-contraction GEMMs (weights handed over as a transposed view, as tf_util._dense
does): times every shape with the default heuristic, lets TunableOp tune it, times again and writes the results file.
    python tools/tunable_probe.py out.csv [cls|all]"""
import sys

import torch

which = sys.argv[2] if len(sys.argv) > 2 else "cls"
shapes = [(32768, 2048, 128, True), (8192, 4096, 256, True)]
if which == "all":
    shapes += [(131072, 4192, 128, True), (16384, 8288, 256, True), (4096, 16480, 256, True), (1024, 16480, 512, True),
               (320, 16384, 512, True), (2560, 4096, 128, True), (640, 8192, 256, True), (320, 8192, 512, True), (2560, 2048, 128, True),
               (640, 4096, 256, True), (10240, 2048, 64, True), (81920, 2048, 32, True)]


def timed(fn, n=40):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(5):
                fn()
        g.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n // 5):
            g.replay()
        e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (n // 5 * 5)


ops = []
for M, K, N, relu in shapes:
    a = torch.randn(M, K, device="cuda"); wt = torch.randn(N, K, device="cuda"); b = torch.randn(N, device="cuda")
    ops.append((lambda a=a, wt=wt, b=b: torch._addmm_activation(b, a, wt.t())))
base = [timed(f) for f in ops]
import torch.cuda.tunable as T
T.enable(True); T.tuning_enable(True); T.set_max_tuning_duration(300); T.set_max_tuning_iterations(100)
T.set_filename(sys.argv[1])
for f in ops:
    f()
torch.cuda.synchronize()
T.tuning_enable(False)
tuned = [timed(f) for f in ops]
T.enable(False)
for (M, K, N, relu), t0, t1 in zip(shapes, base, tuned):
    print(f"M={M:6d} K={K:5d} N={N:5d}: default {t0:7.1f} us  tuned {t1:7.1f} us  ({2*M*K*N/t0/1e6:5.1f} -> {2*M*K*N/t1/1e6:5.1f} TF)")
print("sum", round(sum(base), 1), "->", round(sum(tuned), 1))
