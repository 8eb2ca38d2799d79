"""What torch's TunableOp finds for the dense shapes of the classifier: time every shape with the default heuristic, then
let TunableOp tune it and time again.    python tools/tunable_probe.py [out.csv]"""
import sys

import torch

shapes = [  # (M, K, N, relu) of pointasnl_cls at B = 64 (tools/gemm_audit.py)
    (32768, 2048, 128, True), (8192, 4096, 256, True), (32768, 256, 512, True), (8192, 512, 1024, True),
    (32768, 128, 256, True), (8192, 256, 512, True), (32768, 132, 128, True), (32768, 128, 128, False),
    (8192, 260, 256, True), (8192, 131, 64, False),
]


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
    a = torch.randn(M, K, device="cuda"); w = torch.randn(K, N, device="cuda"); b = torch.randn(N, device="cuda")
    ops.append((lambda a=a, w=w, b=b, relu=relu: torch._addmm_activation(b, a, w) if relu else torch.addmm(b, a, w)))
base = [timed(f) for f in ops]
import torch.cuda.tunable as T
T.enable(True); T.tuning_enable(True); T.set_max_tuning_duration(200); T.set_max_tuning_iterations(50)
if len(sys.argv) > 1:
    T.set_filename(sys.argv[1])
for f in ops:
    f()
torch.cuda.synchronize()
T.tuning_enable(False)
tuned = [timed(f) for f in ops]
for (M, K, N, relu), t0, t1 in zip(shapes, base, tuned):
    print(f"M={M:6d} K={K:5d} N={N:5d} relu={int(relu)}: default {t0:7.1f} us  tuned {t1:7.1f} us  ({2*M*K*N/t0/1e6:5.1f} -> {2*M*K*N/t1/1e6:5.1f} TF)")
print("sum", round(sum(base), 1), "->", round(sum(tuned), 1))
if len(sys.argv) > 1:
    T.write_file(sys.argv[1])
