"""pasnl_dense_bf16x3 (fp32 operands as three bf16 terms, six products, csrc/dense_bf16x3.hip) against the vendor fp32 GEMM on the
long products of the models: time (a replayed HIP graph of 5 calls) and worst error against fp64.   python tools/bf16x3_gemm_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pointasnl_amd.utils import tf_util


def timed(fn, n=30):
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


tf_util.set_store(tf_util.VariableStore(seed=1))
for (M, K, N) in [(131072, 4192, 128), (16384, 8288, 256), (4096, 16480, 256), (32768, 2048, 128), (8192, 4096, 256), (2560, 4096, 128), (300, 1024, 128)]:
    a = torch.randn(M, K, device="cuda"); w = torch.randn(K, N, device="cuda") * 0.05; b = torch.randn(N, device="cuda")
    wt = w.t().contiguous().t()
    got = tf_util._dense_bf16x3(a, w, b, True)
    ref32 = torch._addmm_activation(b, a, wt)
    rows = slice(0, min(M, 512))
    want = torch.relu(a[rows].double() @ w.double() + b.double())
    scale = float(want.abs().max())
    e_bx = float((got[rows].double() - want).abs().max()) / scale
    e_32 = float((ref32[rows].double() - want).abs().max()) / scale
    t_v = timed(lambda: torch._addmm_activation(b, a, wt))
    t_b = timed(lambda: tf_util._dense_bf16x3(a, w, b, True))
    fl = 2 * M * K * N / 1e6
    print(f"M={M:6d} K={K:6d} N={N:4d}  vendor fp32 {t_v:7.1f} us {fl / t_v:6.1f} TF (err {e_32:.1e}) | bf16x3 {t_b:7.1f} us {fl / t_b:6.1f} fp32-equivalent TF "
          f"(err {e_bx:.1e})  x{t_v / t_b:.2f}", flush=True)
