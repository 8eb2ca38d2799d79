"""pasnl_dense_splitk against the vendor GEMM (plain and with transposed weights) on the thin, long products of the
segmentation models, each alone in a replayed HIP graph (as tools/gemm_audit.py times them)."""
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


for (M, K, N) in [(320, 16384, 512), (2560, 4096, 128), (640, 8192, 256), (320, 8192, 512), (2560, 2048, 128), (640, 4096, 256),
                  (4096, 384, 256), (4096, 16480, 256), (1024, 16480, 512), (512, 8192, 512), (4096, 2048, 128), (1024, 4096, 256),
                  (16384, 8288, 256), (8192, 4096, 256), (10240, 2048, 64)]:
    a = torch.randn(M, K, device="cuda"); w = torch.randn(K, N, device="cuda"); b = torch.randn(N, device="cuda")
    wt = w.t().contiguous().t()
    t_v = timed(lambda: torch._addmm_activation(b, a, w))
    t_t = timed(lambda: torch._addmm_activation(b, a, wt))
    t_s = timed(lambda: tf_util._dense_splitk(a, w, b, True))
    fl = 2 * M * K * N / 1e6
    print(f"M={M:6d} K={K:6d} N={N:4d}  vendor {t_v:7.1f} us {fl / t_v:6.1f} TF | vendor W^T {t_t:7.1f} us {fl / t_t:6.1f} TF | "
          f"splitk {t_s:7.1f} us {fl / t_s:6.1f} TF", flush=True)
