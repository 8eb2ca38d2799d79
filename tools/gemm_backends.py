"""The big GEMM shapes of the cls forward under hipBLASLt vs rocBLAS (torch.backends.cuda.preferred_blas_library), with and
without the fused bias+ReLU epilogue; fp32."""
import os, sys
import torch
shapes = [("after_conv L1", 32768, 2048, 128), ("after_conv L2", 8192, 4096, 256), ("layer3_1 conv2", 32768, 256, 512),
          ("layer3_1 conv1", 32768, 128, 256), ("layer3_2 conv2", 8192, 512, 1024), ("layer3_2 conv1", 8192, 256, 512),
          ("fc1", 64, 1536, 512), ("agg L1", 32768, 128, 128), ("conv_kv L2", 32768, 128, 128)]
def t(fn, it=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); ts = []
    for _ in range(it):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) * 1e3)
    return sorted(ts)[len(ts) // 2]
for lib in ("cublaslt", "cublas"):
    torch.backends.cuda.preferred_blas_library(lib)
    for name, m, k, n in shapes:
        a, w, b = torch.randn(m, k, device="cuda"), torch.randn(k, n, device="cuda") * 0.05, torch.randn(n, device="cuda")
        u1 = t(lambda: torch._addmm_activation(b, a, w))
        u2 = t(lambda: torch.addmm(b, a, w))
        u3 = t(lambda: torch.mm(a, w))
        fl = 2.0 * m * k * n
        print(f"{lib:9s} {name:16s} M={m:6d} K={k:5d} N={n:5d}  addmm+relu {u1:7.1f} us ({fl/u1/1e6:6.1f} TF)  addmm {u2:7.1f}  mm {u3:7.1f}", flush=True)
