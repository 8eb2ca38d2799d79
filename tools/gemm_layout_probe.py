"""Does the vendor GEMM run faster when the weight matrix is handed over transposed (a (N,K) tensor viewed as (K,N))?
    python tools/gemm_layout_probe.py"""
import torch


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


if __name__ != "__main__":
    raise SystemExit
for M, K, N in [(32768, 2048, 128), (8192, 4096, 256), (32768, 256, 512), (8192, 512, 1024), (32768, 128, 256), (131072, 4192, 128),
                (16384, 8288, 256), (2560, 4096, 128), (320, 16384, 512)]:
    x = torch.randn(M, K, device="cuda"); w = torch.randn(K, N, device="cuda"); b = torch.randn(N, device="cuda")
    wt = w.t().contiguous()  # (N, K)
    t_nn = timed(lambda: torch._addmm_activation(b, x, w))
    t_nt = timed(lambda: torch._addmm_activation(b, x, wt.t()))
    t_lin = timed(lambda: torch.relu_(torch.nn.functional.linear(x, wt, b)))
    print(f"M={M:6d} K={K:5d} N={N:4d}: W (K,N) {t_nn:7.1f} us   W^T view {t_nt:7.1f} us   linear+relu {t_lin:7.1f} us   ({2*M*K*N/min(t_nn,t_nt)/1e6:5.1f} TF best)", flush=True)
