import torch
def timed(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph(); s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        with torch.cuda.graph(g, stream=s):
            for _ in range(10): fn()
        g.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n // 10): g.replay()
        e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (n // 10 * 10)
M = 98304
for K, N in [(134, 195), (136, 195), (134, 208), (136, 208), (136, 224), (136, 256), (144, 208), (160, 256), (128, 192)]:
    x = torch.randn(M, K, device="cuda"); w = torch.randn(K, N, device="cuda"); b = torch.randn(N, device="cuda")
    t = timed(lambda: torch.addmm(b, x, w))
    print(f"M={M} K={K} N={N}: {t:7.1f} us  {2*M*K*N/t/1e6:6.1f} TF (useful {2*M*134*195/t/1e6:6.1f})")
# strided variants: x (M,136) view of width 134? output into padded buffer
x = torch.randn(M, 136, device="cuda"); w = torch.randn(134, 208, device="cuda"); b = torch.randn(208, device="cuda")
t = timed(lambda: torch.addmm(b, x[:, :134], w)); print("A lda=136 K=134, N=208:", round(t, 1))
